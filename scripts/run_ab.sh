#!/bin/bash
# A/B of map-kernel builds on one box: TAGS="base X base X" (tag X = ct_mapreduce_amd/libctmr_sweep_X.so, base = libctmr.so)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ab; mkdir -p $OUT
cd $R
for tag in ${TAGS:-base}; do
  lib=$R/ct_mapreduce_amd/libctmr.so
  [ $tag != base ] && lib=$R/ct_mapreduce_amd/libctmr_sweep_$tag.so
  CTMR_LIB=$lib timeout 300 python bench.py --no-cpu --traffic off --steps 8 --warmup 2 ${BENCH_ARGS:-} > $OUT/b_$tag.json 2> $OUT/b_$tag.err
  python3 -c "
import json; d=json.load(open('$OUT/b_$tag.json')); print('$tag', 'map_ms', round(d['kernel_ms']['map'],3), 'step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items() if k in ('decode','match')})" | tee -a $OUT/summary.txt
done
[ -n "${PYTEST:-}" ] && timeout 900 python -m pytest $PYTEST -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -2
