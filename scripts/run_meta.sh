#!/bin/bash
# IssuerMetadata memo pre-check inside the map kernel (k_map_fused<16, true>) against the build before it.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/meta; mkdir -p $OUT; rm -f $OUT/*
cd $R
timeout ${PYTEST_TIMEOUT:-300} python -m pytest ${PYTEST:-tests/test_gpu_meta.py tests/test_gpu_parity.py tests/test_storage_gpu.py} -m gpu -x -q -p no:cacheprovider 2>&1 | tail -12 | tee -a $OUT/summary.txt
grep -q "failed\|error" $OUT/summary.txt && exit 1
for tag in ${TAGS:-base before base before}; do
  lib=$R/ct_mapreduce_amd/libctmr.so
  [ $tag != base ] && lib=$R/ct_mapreduce_amd/libctmr_sweep_$tag.so
  CTMR_LIB=$lib timeout 300 python bench.py --meta --no-cpu --traffic off --steps 5 --warmup 1 > $OUT/b_$tag.json 2> $OUT/b_$tag.err || { echo "$tag failed"; tail -3 $OUT/b_$tag.err; exit 1; }
  python3 -c "
import json; d=json.loads([l for l in open('$OUT/b_$tag.json').read().splitlines() if l.startswith('{')][-1]); print('$tag', 'certs/s', round(d['value']/1e9,3), 'ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items()}, d.get('meta'))" | tee -a $OUT/summary.txt
done
