#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu5.log 2>&1; tail -6 $OUT/pytest_gpu5.log
for v in 14 15 14 15; do
  timeout 300 python bench.py --steps 5 --warmup 1 --variant $v --no-cpu > $OUT/sweep_v$v.json 2>> $OUT/sweep.err
  python -c "
import json; d=json.load(open('$OUT/sweep_v$v.json')); print('variant', $v, 'map_ms', d['kernel_ms']['map'], 'frac', d['roofline']['frac'], 'value', d['value'])" | tee -a $OUT/sweep_strict_window.txt
done
timeout 600 python bench.py --raw --steps 3 --warmup 1 --no-cpu > $OUT/bench_raw_40m_c.json 2> $OUT/bench_raw_40m_c.err; python -c "
import json; d=json.load(open('$OUT/bench_raw_40m_c.json')); print('raw', d['value'], d['kernel_ms'])"
timeout 600 python bench.py --meta --fingerprint --steps 3 --warmup 1 --no-cpu > $OUT/bench_meta_fp_100m.json 2> $OUT/bench_meta_fp_100m.err; python -c "
import json; d=json.load(open('$OUT/bench_meta_fp_100m.json')); print('meta', d['kernel_ms'], d.get('meta')); print('fp', d.get('fingerprint'))"; tail -2 $OUT/bench_meta_fp_100m.err
timeout 600 python scripts/host_batch_latency.py > $OUT/host_batch_latency.json 2> $OUT/host_batch_latency.err; cat $OUT/host_batch_latency.json; tail -2 $OUT/host_batch_latency.err
