#!/bin/bash
# round 4: the default line with its traffic measurement, then the strict_strings and mixed lines
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4j; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
J='import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])'
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "$J
r=d['roofline']
print('default', d['value'], d['ms_per_step'], 'frac', r['frac'], 'alg', r['frac_algorithmic'], 'traffic', r['traffic_measurement'], 'overfetch', r.get('over_fetch_vs_needed_lines'), d['kernel_ms'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['parity_vs_oracle_on_sample'], d['checks'])
for k,v in d.get('secondary',{}).items(): print('  ', k, {a:b for a,b in v.items() if a not in ('workload','note')})" $OUT/bench_default.json; tail -2 $OUT/bench_default.err
for m in "--strict-strings" "--mixed --strict-strings"; do
  tag=$(echo $m | tr -d ' -'); timeout 600 python bench.py $m --no-cpu --no-secondary --traffic off --steps 6 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; python -c "$J
print('$m', d['value'], d['ms_per_step'], d['kernel_ms'])" $OUT/bench_$tag.json
done
