#!/bin/bash
# round 4: the RSA key shape taken the short way — same-box A/B against the previous build, then the suite and the key fuzz
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4i; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
for tag in prev base prev base prev base; do
  lib=$R/ct_mapreduce_amd/libctmr.so
  [ $tag != base ] && lib=$R/ct_mapreduce_amd/libctmr_sweep_$tag.so
  for m in "" "--no-strict-spki"; do
  CTMR_LIB=$lib timeout 300 python bench.py $m --no-cpu --no-secondary --traffic off --steps 6 --warmup 2 > $OUT/b.json 2> $OUT/b.err
  python3 -c "
import json; d=json.loads([l for l in open('$OUT/b.json').read().splitlines() if l.startswith('{')][-1]); print('$tag', '$m', 'map', round(d['kernel_ms']['map'],3), 'step', round(d['ms_per_step'],2), d['checks']['entries_disagreeing_with_generator'])" | tee -a $OUT/ab.txt
  done
done
timeout 600 python scripts/fuzz_gpu.py 3000000 20260928 > $OUT/fuzz_gpu_certificates.txt 2>&1; tail -2 $OUT/fuzz_gpu_certificates.txt
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
