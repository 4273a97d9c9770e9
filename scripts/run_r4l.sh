#!/bin/bash
# round 4: opening control row of exact rounds, k_insert2 taking each thread's first DEFER entry — suite, group fuzz, bench
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4l; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
J='import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])'
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -B2 -A30 "^E  " $OUT/pytest.log | head -60
timeout 600 python scripts/fuzz_gpu_groups.py 2500 > $OUT/fuzz_gpu_groups.txt 2>&1; tail -2 $OUT/fuzz_gpu_groups.txt
for rep in 1 2 3; do
  timeout 600 python bench.py --no-cpu --no-secondary --traffic off --steps 6 --warmup 2 > $OUT/bench_d_$rep.json 2> $OUT/bench_d_$rep.err; python -c "$J
print('default', d['value'], d['ms_per_step'], d['kernel_ms'], d['checks']['entries_disagreeing_with_generator'])" $OUT/bench_d_$rep.json | tee -a $OUT/lines.txt || tail -3 $OUT/bench_d_$rep.err
done
