#!/bin/bash
# round 2, first GPU call: the whole -m gpu suite, smoke, the default bench line (with its own traffic measurement)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2a; mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.txt
tail -25 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc $?" >> $OUT/smoke.txt; tail -3 $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"; cat $OUT/bench_default.json; tail -5 $OUT/bench_default.err
