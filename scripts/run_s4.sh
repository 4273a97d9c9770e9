#!/bin/bash
# session-4 GPU pass: parity tests, default bench (regression check of the map kernel), raw-entry bench with
# rocprofv3 kernel stats
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu3.log 2>&1; tail -15 $OUT/pytest_gpu3.log
timeout 400 python bench.py --steps 5 --warmup 1 > $OUT/bench_default2.json 2> $OUT/bench_default2.err; cat $OUT/bench_default2.json | cut -c1-1500
timeout 600 python bench.py --raw --steps 3 --warmup 1 --no-cpu > $OUT/bench_raw_40m.json 2> $OUT/bench_raw_40m.err; cat $OUT/bench_raw_40m.json; tail -3 $OUT/bench_raw_40m.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_raw -o kt --output-format csv -- python $R/bench.py --raw --entries 20000000 --steps 3 --warmup 1 --no-cpu > $OUT/kt_raw.log 2>&1
find $OUT -name "*kernel_trace.csv" -size +1M -delete
find $OUT/kt_raw -name "*kernel_stats.csv" | head -1 | xargs head -12 | cut -c1-160
