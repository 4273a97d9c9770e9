#!/bin/bash
# parity tests, then the default bench on the homogeneous and on the mixed corpus (no CPU leg), twice each
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/mixed; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for k in 1 2; do
  for extra in "" "--mixed"; do
    timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu $extra > $OUT/b.json 2>> $OUT/b.err
    python -c "
import json; d=json.loads([l for l in open('$OUT/b.json').read().splitlines() if l.startswith('{')][-1]); print('[$extra]', 'map_ms', round(d['kernel_ms']['map'],3), 'value', round(d['value']))" | tee -a $OUT/mixed.txt
  done
done
