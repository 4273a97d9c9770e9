#!/bin/bash
# round 3: the reworked bench.py — default line (with secondary.mixed), the multi-rank tests, the secondary lines
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3b; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bench_multirank.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_multirank.log 2>&1; tail -30 $OUT/pytest_multirank.log | cut -c1-300
[ -n "${QUICK:-}" ] && exit 0
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err; python -c "
import json; d=json.loads([l for l in open('$OUT/bench_default.json').read().splitlines() if l.startswith('{')][-1]); r=d['roofline']
print('default', d['value'], d['ms_per_step'], 'frac', r['frac'], 'alg', r['frac_algorithmic'], d['kernel_ms'], d['checks'], d['parity_vs_oracle_on_sample'], d.get('secondary'))"
for m in "--raw" "--meta" "--global-dedup owner" "--global-dedup bloom" "--mixed"; do
  tag=$(echo $m | tr -d ' -'); timeout 900 python bench.py $m --no-cpu --steps 3 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; tail -2 $OUT/bench_$tag.err; python -c "
import json; d=json.loads([l for l in open('$OUT/bench_$tag.json').read().splitlines() if l.startswith('{')][-1]); r=d['roofline']
print('$m', d['value'], d['ms_per_step'], 'frac', r['frac'], 'alg', r['frac_algorithmic'], r.get('invalid'), d['kernel_ms'], d.get('exchange'), d.get('roofline_decode_match'))"
done
