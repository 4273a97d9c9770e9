#!/bin/bash
# round 4: what the key parse and the new OID / Name-value rules cost the map kernel — A/B builds on one box
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4b; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
run() { # tag, bench args…
  tag=$1; shift
  lib=$R/ct_mapreduce_amd/libctmr.so
  [ $tag != base ] && lib=$R/ct_mapreduce_amd/libctmr_sweep_$tag.so
  name=$tag$(echo "$*" | tr -d ' -')
  CTMR_LIB=$lib timeout 300 python bench.py --no-cpu --no-secondary --traffic off --steps 8 --warmup 2 "$@" > $OUT/b_$name.json 2> $OUT/b_$name.err
  python3 -c "
import json; d=json.loads([l for l in open('$OUT/b_$name.json').read().splitlines() if l.startswith('{')][-1]); print('$tag $*', 'map', round(d['kernel_ms']['map'],3), 'insert', round(d['kernel_ms']['insert'],3), 'step', round(d['ms_per_step'],2), d['checks']['entries_disagreeing_with_generator'])" | tee -a $OUT/summary.txt || tail -3 $OUT/b_$name.err
}
for rep in 1 2; do
  run base
  run base --no-strict-spki
  run nooid
  run nonv
  run nonvoid
done
run base --mixed
run base --mixed --no-strict-spki
timeout 900 python -m pytest tests/test_gpu_spki.py tests/test_gpu_parity.py tests/test_gpu_exchange.py tests/test_gpu_bloom.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -2
