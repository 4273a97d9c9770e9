#!/bin/bash
# round 4: k_resolve with four entries per thread, the string sets checked four octets at a time — tests, lines, fuzz
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4m; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
J='import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])'
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -B2 -A30 "^E  " $OUT/pytest.log | head -60
for rep in 1 2; do
for m in "" "--strict-strings" "--mixed --strict-strings"; do
  tag=$(echo "d$m" | tr -d ' -')_$rep
  timeout 600 python bench.py $m --no-cpu --no-secondary --traffic off --steps 6 --warmup 2 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; python -c "$J
print('$m', d['value'], d['ms_per_step'], d['kernel_ms'], d['checks']['entries_disagreeing_with_generator'])" $OUT/bench_$tag.json | tee -a $OUT/lines.txt || tail -3 $OUT/bench_$tag.err
done
done
timeout 600 python scripts/fuzz_gpu.py 3000000 20260930 > $OUT/fuzz_gpu_certificates.txt 2>&1; tail -2 $OUT/fuzz_gpu_certificates.txt
