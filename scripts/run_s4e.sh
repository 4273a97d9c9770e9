#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu7.log 2>&1; tail -6 $OUT/pytest_gpu7.log
timeout 600 python bench.py --pem --steps 3 --warmup 1 --no-cpu > $OUT/bench_pem_100m.json 2> $OUT/bench_pem_100m.err; python -c "
import json; d=json.load(open('$OUT/bench_pem_100m.json')); print('pem', d.get('pem'), d['kernel_ms'])"; tail -2 $OUT/bench_pem_100m.err
timeout 600 python bench.py --global-dedup --steps 3 --warmup 1 --no-cpu > $OUT/bench_global_dedup_n1.json 2> $OUT/bench_global_dedup_n1.err; python -c "
import json; d=json.load(open('$OUT/bench_global_dedup_n1.json')); print('gd', d['value'], d['ms_per_step'], d['result'])"; tail -2 $OUT/bench_global_dedup_n1.err
