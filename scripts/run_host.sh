#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 600 python scripts/host_batch_latency.py > $OUT/host_batch_latency3.json 2> $OUT/host_batch_latency3.err; python -c "
import json; d=json.load(open('$OUT/host_batch_latency3.json'))
for r in d['results']: print(r)"
