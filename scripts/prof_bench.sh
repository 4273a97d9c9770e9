#!/bin/bash
# usage: prof_bench.sh TAG ENTRIES — rocprofv3 kernel-trace stats of the bench command, then separate
# PMC passes (FETCH_SIZE / WRITE_SIZE) for the HBM traffic of the dominant kernel.
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; E=$2
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
CMD="python $R/bench.py --entries $E --steps 5 --warmup 1"
timeout 600 $CMD > $OUT/bench.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- $CMD --no-cpu > $OUT/kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $OUT/$c -o pmc --output-format csv -- $CMD --no-cpu > $OUT/$c.log 2>&1
done
python $R/scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -size +1M -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete
cat $OUT/bench.json; find $OUT -name "*kernel_stats.csv" | head -1 | xargs head -12; grep -E "k_map" $OUT/pmc_summary.txt
