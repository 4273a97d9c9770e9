#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu6.log 2>&1; tail -6 $OUT/pytest_gpu6.log
timeout 600 python bench.py --meta --steps 3 --warmup 1 --no-cpu > $OUT/bench_meta_100m_b.json 2> $OUT/bench_meta_100m_b.err; python -c "
import json; d=json.load(open('$OUT/bench_meta_100m_b.json')); print('meta', d['kernel_ms'], d.get('meta'))"; tail -2 $OUT/bench_meta_100m_b.err
bash scripts/prof_bench.sh s4/v15 100000000 15 2>&1 | tail -12
