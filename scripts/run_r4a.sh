#!/bin/bash
# round 4, first GPU call: strict_spki (the public key is parsed, on by default) — the whole -m gpu suite, then what the
# key parse costs: A/B/A/B of the default line with and without it, and the same on the mixed corpus (half EC keys).
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4a; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
J='import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])'
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
for rep in 1 2; do
  for m in "" "--no-strict-spki" "--mixed" "--mixed --no-strict-spki"; do
    tag=$(echo "d$m" | tr -d ' -')_$rep
    timeout 600 python bench.py $m --no-cpu --no-secondary --traffic off --steps 10 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; python -c "$J
print('$m', d['value'], d['ms_per_step'], d['kernel_ms'], d['checks'])" $OUT/bench_$tag.json || tail -3 $OUT/bench_$tag.err
  done
done
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "$J
r=d['roofline']
print('default', d['value'], d['ms_per_step'], 'frac', r['frac'], 'alg', r['frac_algorithmic'], 'B/cert', r['traffic_measurement'] and r['traffic_measurement']['traffic_bytes_per_cert'], d['kernel_ms'], d['cpu_baseline']['value'], d['parity_vs_oracle_on_sample'], d['checks'])
for k,v in d.get('secondary',{}).items(): print('  ', k, {a:b for a,b in v.items() if a not in ('workload','note')})" $OUT/bench_default.json; tail -2 $OUT/bench_default.err
