#!/bin/bash
# round 4: do chunk c's key records travel while chunk c + 1 is mapped?  kernel + memory-copy trace of a LOCAL group
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4w; mkdir -p $OUT; rm -rf $OUT/*
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/prof -o trace --output-format csv -- python $R/scripts/chunk_overlap_trace.py run > $OUT/run.txt 2>&1; tail -3 $OUT/run.txt
python $R/scripts/chunk_overlap_trace.py report $OUT/prof | tee $OUT/chunk_overlap_report.txt
f=$(find $OUT/prof -name "*memory_copy_trace.csv" | head -1); head -3 "$f" | cut -c1-300
find $OUT/prof -name "*.csv" -size +2M -delete
