#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu10.log 2>&1; tail -6 $OUT/pytest_gpu10.log
for v in 15 14 15; do
  timeout 300 python bench.py --steps 5 --warmup 1 --variant $v --no-cpu > $OUT/sweep2_v$v.json 2>> $OUT/sweep2.err
  python -c "
import json; d=json.load(open('$OUT/sweep2_v$v.json')); print('variant', $v, 'map_ms', d['kernel_ms']['map'], 'frac', d['roofline']['frac'], 'value', d['value'])" | tee -a $OUT/sweep_tail_prefetch.txt
done
