#!/bin/bash
# session-4 second GPU pass: parity tests, default bench, raw bench (decode counters fixed), meta bench, 1B stream
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu4.log 2>&1; tail -8 $OUT/pytest_gpu4.log
timeout 500 python bench.py --steps 5 --warmup 1 > $OUT/bench_default3.json 2> $OUT/bench_default3.err; cut -c1-400 $OUT/bench_default3.json; echo; python -c "
import json; d=json.load(open('$OUT/bench_default3.json')); print(d['value'], d['roofline']['frac'], d['kernel_ms'], d.get('cpu_baseline'), d.get('parity_vs_oracle_on_sample'))"
tail -2 $OUT/bench_default3.err
timeout 600 python bench.py --raw --steps 3 --warmup 1 --no-cpu > $OUT/bench_raw_40m_b.json 2> $OUT/bench_raw_40m_b.err; python -c "
import json; d=json.load(open('$OUT/bench_raw_40m_b.json')); print('raw', d['value'], d['kernel_ms'])"; tail -2 $OUT/bench_raw_40m_b.err
timeout 600 python bench.py --meta --steps 3 --warmup 1 --no-cpu > $OUT/bench_meta_100m.json 2> $OUT/bench_meta_100m.err; python -c "
import json; d=json.load(open('$OUT/bench_meta_100m.json')); print('meta', d['value'], d['kernel_ms'], d.get('meta'))"; tail -2 $OUT/bench_meta_100m.err
timeout 900 python bench.py --stream 1000000000 > $OUT/bench_stream_1b.json 2> $OUT/bench_stream_1b.err; tail -4 $OUT/bench_stream_1b.err; cat $OUT/bench_stream_1b.json
