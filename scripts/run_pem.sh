#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pem; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_pem.py tests/test_gpu_entries.py tests/test_storage_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python bench.py --pem --steps 2 --warmup 1 --no-cpu > $OUT/bench_pem_b.json 2> $OUT/bench_pem_b.err; python -c "
import json; d=json.load(open('$OUT/bench_pem_b.json')); print('pem', d.get('pem'))"
