#!/bin/bash
# round 4: counters that serialised on one address (k_apply_lost, the map's pending flag) — suite, rank cost, mixed + default lines
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4p; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
J='import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])'
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -B2 -A30 "^E  " $OUT/pytest.log | head -60
for spec in "50000000 2" "25000000 4" "12500000 8"; do
  set -- $spec
  timeout 600 python scripts/rank_cost_at_world.py $1 $2 > $OUT/rank_cost_$1_w$2.json 2>&1; tail -1 $OUT/rank_cost_$1_w$2.json | cut -c1-700
done
for m in "--mixed" "" "--mixed" ""; do
  tag=$(echo "d$m" | tr -d ' -')
  timeout 600 python bench.py $m --no-cpu --no-secondary --traffic off --steps 6 --warmup 2 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; python -c "$J
print('$m', d['value'], d['ms_per_step'], d['kernel_ms'], d['checks']['entries_disagreeing_with_generator'])" $OUT/bench_$tag.json | tee -a $OUT/lines.txt
done
timeout 600 python scripts/fuzz_gpu_groups.py 1500 > $OUT/fuzz_gpu_groups.txt 2>&1; tail -2 $OUT/fuzz_gpu_groups.txt
