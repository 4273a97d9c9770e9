#!/bin/bash
# round 4: arena compaction (test + suite), and what the apply phase of the owner stand-in spends its time on
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4o; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -B2 -A30 "^E  " $OUT/pytest.log | head -60
timeout 600 python scripts/rank_cost_at_world.py 12500000 8 > $OUT/rank_cost.json 2>&1; tail -1 $OUT/rank_cost.json | cut -c1-900
( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o rc --output-format csv -- python $R/scripts/rank_cost_at_world.py 12500000 8 1 > /dev/null 2>&1
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); head -25 "$f" | cut -c1-150; find $OUT/prof -name "*.csv" -size +1M -delete )
timeout 600 python bench.py --no-cpu --no-secondary --traffic off --steps 6 --warmup 2 > $OUT/bench_d.json 2> $OUT/bench_d.err; python -c "
import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1]); print('default', d['value'], d['ms_per_step'], d['kernel_ms'])" $OUT/bench_d.json
