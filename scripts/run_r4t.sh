#!/bin/bash
# the driver's round-end sequence at HEAD: build check + smoke, the GPU suite, the default bench line
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4t; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
J='import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])'
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "$J
r=d['roofline']
print('default', d['value'], d['ms_per_step'], 'frac', r['frac'], 'alg', r['frac_algorithmic'], d['kernel_ms'], d['cpu_baseline']['value'], d['parity_vs_oracle_on_sample'], d['checks']['entries_disagreeing_with_generator'])" $OUT/bench_default.json; tail -2 $OUT/bench_default.err
