#!/bin/bash
# round 4: k_ec_resolve split by curve class (P-256 with 8-limb registers / the rest) — key tests, mixed line, raw line, fuzz
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4q; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
J='import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])'
timeout 900 python -m pytest tests/test_gpu_spki.py tests/test_gpu_parity.py tests/test_gpu_entries.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3
for m in "--mixed" "--mixed" "--raw"; do
  tag=$(echo "d$m" | tr -d ' -')
  timeout 600 python bench.py $m --no-cpu --no-secondary --traffic off --steps 6 --warmup 2 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; python -c "$J
print('$m', d['value'], d['ms_per_step'], d['kernel_ms'], d['checks']['entries_disagreeing_with_generator'] if 'checks' in d else '')" $OUT/bench_$tag.json | tee -a $OUT/lines.txt
done
timeout 600 python scripts/fuzz_gpu.py 2000000 20261001 > $OUT/fuzz_gpu_certificates.txt 2>&1; tail -2 $OUT/fuzz_gpu_certificates.txt
timeout 600 python scripts/fuzz_gpu_groups.py 800 > $OUT/fuzz_gpu_groups.txt 2>&1; tail -1 $OUT/fuzz_gpu_groups.txt
