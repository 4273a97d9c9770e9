#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/probe; mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -2
for tag in base linear base linear base linear; do
  lib=$R/ct_mapreduce_amd/libctmr.so
  [ $tag != base ] && lib=$R/ct_mapreduce_amd/libctmr_sweep_$tag.so
  CTMR_LIB=$lib timeout 300 python bench.py --no-cpu --traffic off --steps 8 --warmup 2 > $OUT/b_$tag.json 2> $OUT/b_$tag.err
  python3 -c "
import json; d=json.load(open('$OUT/b_$tag.json')); print('$tag', 'map_ms', round(d['kernel_ms']['map'],3), 'insert2', round(d['kernel_ms']['insert'],3), 'step', round(d['ms_per_step'],2))" | tee -a $OUT/summary.txt
done
