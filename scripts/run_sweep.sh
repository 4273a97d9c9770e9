#!/bin/bash
# usage: run_sweep.sh TAG "v1 v2 ..." — parity tests, then the default bench per map variant (no CPU leg)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/sweep; mkdir -p $OUT
cd $R
TAG=$1; VARS=$2
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1; tail -3 $OUT/pytest_$TAG.log
for v in $VARS; do
  timeout 300 python bench.py --steps 5 --warmup 1 --variant $v --no-cpu > $OUT/sw_${TAG}_v$v.json 2>> $OUT/sw_$TAG.err
  python -c "
import json; d=json.load(open('$OUT/sw_${TAG}_v$v.json')); print('$TAG variant', $v, 'map_ms', d['kernel_ms']['map'], 'frac', d['roofline']['frac'], 'value', d['value'])" | tee -a $OUT/sweep_$TAG.txt
done
