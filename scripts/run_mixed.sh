#!/bin/bash
# mixed-corpus map kernel time of build variants: TAGS="base nosubj ..." (tag X = libctmr_sweep_X.so)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/mixed; mkdir -p $OUT
cd $R
for tag in ${TAGS:-base}; do
  lib=$R/ct_mapreduce_amd/libctmr.so
  [ $tag != base ] && lib=$R/ct_mapreduce_amd/libctmr_sweep_$tag.so
  CTMR_LIB=$lib timeout 600 python bench.py --mixed --no-cpu --traffic off --steps 6 > $OUT/bench_mixed_$tag.json 2> $OUT/bench_mixed_$tag.err; python -c "
import json; d=json.loads([l for l in open('$OUT/bench_mixed_$tag.json').read().splitlines() if l.startswith('{')][-1]); print('$tag mixed', round(d['value']/1e9,3), round(d['ms_per_step'],2), round(d['kernel_ms']['map'],3))" | tee -a $OUT/summary.txt
done
