#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/tbl; mkdir -p $OUT
cd $R
for t in 28 29 30 28; do
  timeout 300 python bench.py --no-cpu --traffic off --steps 5 --warmup 1 --table-slots-log2 $t > $OUT/b_$t.json 2> $OUT/b_$t.err
  python3 -c "
import json; d=json.load(open('$OUT/b_$t.json')); print('table 2^$t', 'map_ms', round(d['kernel_ms']['map'],3), 'insert2', round(d['kernel_ms']['insert'],3), 'step', round(d['ms_per_step'],2))" | tee -a $OUT/summary.txt
done
