#!/bin/bash
# round 2, second GPU call: the -m gpu suite with the native group layer, the two global-dedup bench modes at N = 1
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2b; mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.txt
tail -30 $OUT/pytest_gpu.txt
for m in owner bloom; do
  timeout 600 python bench.py --global-dedup $m --no-cpu --steps 3 > $OUT/bench_gd_$m.json 2> $OUT/bench_gd_$m.err; echo "bench $m rc $?"; cut -c1-1500 $OUT/bench_gd_$m.json; tail -3 $OUT/bench_gd_$m.err
done
