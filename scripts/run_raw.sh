#!/bin/bash
# raw get-entries bench (decode + Chain[0] match + map) after the N2 parity tests
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_entries.py tests/test_storage_gpu.py -m gpu -x -q 2>&1 | tail -2
for k in 1 2; do
timeout 600 python bench.py --raw --steps 3 --warmup 1 --no-cpu > $OUT/bench_raw_$1_$k.json 2> $OUT/bench_raw_$1.err; python -c "
import json; d=json.load(open('$OUT/bench_raw_$1_$k.json')); print('raw $1', d['value'], d['kernel_ms'])"
done
