#!/bin/bash
# what the driver does at round end: build check, GPU tests, smoke, default bench
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4; mkdir -p $OUT
cd $R
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_final.log 2>&1; tail -3 $OUT/pytest_final.log
timeout 600 python bench.py --pem > $OUT/bench_final.json 2> $OUT/bench_final.err; python -c "
import json; d=json.load(open('$OUT/bench_final.json')); print(d['value'], d['roofline']['frac'], d['roofline']['frac_physical'], d['kernel_ms'], d['cpu_baseline']['value'], d['parity_vs_oracle_on_sample']); print('pem', d.get('pem'))"; tail -2 $OUT/bench_final.err
