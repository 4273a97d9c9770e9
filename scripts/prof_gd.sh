#!/bin/bash
# rocprofv3 kernel stats of the two global-dedup modes at N=1 (100 M entries, 10 % duplicates)
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s5; mkdir -p $OUT
for mode in owner bloom; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$mode -o gd_$mode --output-format csv -- python $R/bench.py --no-cpu --steps 3 --warmup 1 --global-dedup $mode > $OUT/prof_$mode.json 2> $OUT/prof_$mode.err
  f=$(find $OUT/prof_$mode -name "*kernel_stats.csv" | head -1)
  head -16 "$f" | cut -d, -f1-8 | tee $OUT/prof_${mode}_kernel_stats.txt
  find $OUT/prof_$mode -name "*.csv" -size +1M -delete
done
