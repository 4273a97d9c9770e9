#!/bin/bash
# usage: prof_bench.sh TAG ENTRIES [VARIANT] — bench line, rocprofv3 kernel-trace stats of the same command, then
# separate PMC passes (FETCH_SIZE / WRITE_SIZE; no trace flags) for the HBM traffic of the map kernel.
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; E=$2; V=${3:-0}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
CMD="python $R/bench.py --entries $E --steps 5 --warmup 1 --variant $V"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- $CMD --no-cpu > $OUT/kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $OUT/$c -o pmc --output-format csv -- $CMD --no-cpu > $OUT/$c.log 2>&1
done
python $R/scripts/make_traffic.py $OUT $E ${V/#0/15} > $OUT/traffic.json 2> $OUT/traffic.err
python $R/scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -size +1M -delete; find $OUT -name "*counter_collection.csv" -size +1M -delete
timeout 600 $CMD --traffic-file $OUT/traffic.json > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json; cat $OUT/traffic.json; find $OUT -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-200
