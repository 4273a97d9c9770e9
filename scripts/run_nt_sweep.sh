#!/bin/bash
# map-kernel build variants against the default build, same box, same run (bench.py, no CPU legs, no PMC)
# usage: TAGS="base ntmore base ntmore" run_nt_sweep.sh   (tag X = ct_mapreduce_amd/libctmr_sweep_X.so, base = libctmr.so)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/nt; mkdir -p $OUT
cd $R
for tag in ${TAGS:-base}; do
  lib=$R/ct_mapreduce_amd/libctmr.so
  [ $tag != base ] && lib=$R/ct_mapreduce_amd/libctmr_sweep_$tag.so
  CTMR_LIB=$lib timeout 300 python bench.py --no-cpu --traffic off --steps 8 --warmup 2 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python3 -c "
import json; d=json.load(open('$OUT/bench_$tag.json')); print('$tag', 'map_ms', round(d['kernel_ms']['map'],3), 'avg', round(d['roofline']['avg_launch_ms'],3), 'step', round(d['ms_per_step'],3))" | tee -a $OUT/summary2.txt
done
