#!/bin/bash
# round 4: the index + arena table — whole GPU suite, the fuzz campaigns, then the bench lines
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4g; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
J='import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])'
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -B2 -A30 "^E  " $OUT/pytest.log | head -80
[ -n "${TESTS_ONLY:-}" ] && exit 0
timeout 600 python scripts/fuzz_gpu_groups.py 2500 > $OUT/fuzz_gpu_groups.txt 2>&1; tail -2 $OUT/fuzz_gpu_groups.txt
timeout 600 python scripts/fuzz_gpu.py 4000000 20260927 > $OUT/fuzz_gpu_certificates.txt 2>&1; tail -2 $OUT/fuzz_gpu_certificates.txt
for rep in 1 2; do
for m in "" "--mixed" "--no-strict-spki"; do
  tag=$(echo "d$m" | tr -d ' -')_$rep
  timeout 600 python bench.py $m --no-cpu --no-secondary --traffic off --steps 8 --warmup 2 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; python -c "$J
print('$m', d['value'], d['ms_per_step'], d['kernel_ms'], d['checks']['entries_disagreeing_with_generator'])" $OUT/bench_$tag.json || tail -3 $OUT/bench_$tag.err
done
done
