#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/occ; mkdir -p $OUT
cd $R
for x in 0 3072 8192 17408 48000 0; do
  CTMR_EXTRA_LDS=$x CTMR_LIB=$R/ct_mapreduce_amd/libctmr_sweep.so timeout 300 python bench.py --no-cpu --traffic off --steps 5 --warmup 1 > $OUT/b_$x.json 2> $OUT/b_$x.err
  python3 -c "
import json; d=json.load(open('$OUT/b_$x.json')); print('extra_lds', $x, 'map_ms', round(d['kernel_ms']['map'],3))" | tee -a $OUT/summary.txt
done
