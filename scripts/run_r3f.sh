#!/bin/bash
# round 3: aligned layout variant + the default line with both secondary legs
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3f; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err; python -c "
import json; d=json.loads([l for l in open('$OUT/bench_default.json').read().splitlines() if l.startswith('{')][-1]); r=d['roofline']
print('default', d['value'], d['ms_per_step'], 'frac', r['frac'], 'alg', r['frac_algorithmic'], d['kernel_ms'], d['checks'], d['parity_vs_oracle_on_sample'])
for k,v in d.get('secondary',{}).items(): print(k, {a:b for a,b in v.items() if a!='workload'})"
