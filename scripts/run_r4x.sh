#!/bin/bash
# round 4, last run: the whole GPU suite at HEAD, the chunked-round timeline, one rank's cost at a world of 8
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4x; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -B5 -A30 "^E  " $OUT/pytest.log | head -60
( cd /tmp; timeout 300 rocprofv3 --kernel-trace -d $OUT/prof -o trace --output-format csv -- python $R/scripts/chunk_overlap_trace.py run > $OUT/run.txt 2>&1; grep chunks $OUT/run.txt
  python $R/scripts/chunk_overlap_trace.py report $OUT/prof > $OUT/chunk_report.txt; head -40 $OUT/chunk_report.txt; find $OUT/prof -name "*.csv" -size +2M -delete )
timeout 300 python scripts/rank_cost_at_world.py 12500000 8 > $OUT/rank_cost_12500000_w8.json 2>&1; tail -1 $OUT/rank_cost_12500000_w8.json | cut -c1-500
