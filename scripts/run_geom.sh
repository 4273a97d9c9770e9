#!/bin/bash
# window geometry / occupancy experiments of the map kernel; every variant also runs the parity file of the GPU suite
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/geom; mkdir -p $OUT
cd $R
for tag in ${TAGS:-base w14x w16x base w14x}; do
  lib=$R/ct_mapreduce_amd/libctmr.so
  [ $tag != base ] && lib=$R/ct_mapreduce_amd/libctmr_sweep_$tag.so
  CTMR_LIB=$lib timeout 300 python bench.py --no-cpu --traffic off --steps 6 --warmup 2 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python3 -c "
import json; d=json.load(open('$OUT/bench_$tag.json')); print('$tag', 'map_ms', round(d['kernel_ms']['map'],3), 'step', round(d['ms_per_step'],3))" | tee -a $OUT/summary.txt
done
for tag in w14x; do
  CTMR_LIB=$R/ct_mapreduce_amd/libctmr_sweep_$tag.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed" | sed "s/^/$tag parity: /"
  CTMR_LIB=$R/ct_mapreduce_amd/libctmr_sweep_$tag.so timeout 300 python bench.py --mixed --no-cpu --traffic off --steps 4 > $OUT/mixed_$tag.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$OUT/mixed_$tag.json')); print('$tag mixed map_ms', round(d['kernel_ms']['map'],3))"
done
bash scripts/run_occ.sh
