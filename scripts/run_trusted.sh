#!/bin/bash
# CTMR_CHAIN0_TRUSTED_LOG against the exact Chain[0] match: the raw-entry tests, then bench.py --raw in both modes.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trusted; mkdir -p $OUT; rm -f $OUT/*
cd $R
timeout ${PYTEST_TIMEOUT:-300} python -m pytest tests/test_gpu_entries.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 | tee -a $OUT/summary.txt
grep -q "failed\|error" $OUT/summary.txt && exit 1
for mode in ${MODES:-exact trusted exact trusted}; do
  extra=""; [ $mode = trusted ] && extra="--trusted-chain"
  timeout 200 python bench.py --raw --no-cpu --steps 5 --warmup 1 $extra > $OUT/b_$mode.json 2> $OUT/b_$mode.err || { echo "$mode failed"; tail -3 $OUT/b_$mode.err; exit 1; }
  python3 -c "
import json; d=json.loads([l for l in open('$OUT/b_$mode.json').read().splitlines() if l.startswith('{')][-1]); print('$mode', d['value'], 'ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items()}, d['result']['n_new'] if 'result' in d else '')" | tee -a $OUT/summary.txt
done
