#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu9.log 2>&1; tail -6 $OUT/pytest_gpu9.log
timeout 600 python bench.py --meta --pem --steps 3 --warmup 1 --no-cpu > $OUT/bench_meta_pem_100m.json 2> $OUT/bench_meta_pem_100m.err; python -c "
import json; d=json.load(open('$OUT/bench_meta_pem_100m.json')); print('meta', d['kernel_ms'], d.get('meta')); print('pem', d.get('pem'))"; tail -2 $OUT/bench_meta_pem_100m.err
timeout 600 python bench.py --global-dedup --steps 3 --warmup 1 --no-cpu > $OUT/bench_global_dedup_n1_b.json 2> $OUT/bench_global_dedup_n1_b.err; python -c "
import json; d=json.load(open('$OUT/bench_global_dedup_n1_b.json')); print('gd', d['value'], d['ms_per_step'])"
