#!/bin/bash
# mixed corpus: parity, then the bench (and the raw-entry form)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_entries.py -m gpu -x -q 2>&1 | tail -2
for v in 15 14; do
timeout 400 python bench.py --mixed --variant $v --steps 5 --warmup 1 --no-cpu > $OUT/bench_mixed_v$v.json 2> $OUT/bench_mixed.err; python -c "
import json; d=json.load(open('$OUT/bench_mixed_v$v.json')); print('mixed v$v', d['value'], d['kernel_ms'], d['roofline']['frac'], d['config']['mean_der_bytes'], d['result'])"
done
timeout 400 python bench.py --mixed --raw --steps 3 --warmup 1 --no-cpu > $OUT/bench_mixed_raw.json 2>> $OUT/bench_mixed.err; python -c "
import json; d=json.load(open('$OUT/bench_mixed_raw.json')); print('mixed raw', d['value'], d['kernel_ms'])"
tail -2 $OUT/bench_mixed.err
