#!/bin/bash
# parity tests that exercise the reduce tail (insert2 / scan / compact), then the default bench
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s5; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_collisions.py tests/test_gpu_exchange.py tests/test_gpu_bloom.py tests/test_gpu_entries.py -x -q -m gpu > $OUT/pytest_tail.txt 2>&1
tail -5 $OUT/pytest_tail.txt
timeout 900 python bench.py --no-cpu --steps 8 > $OUT/bench_tail.json 2> $OUT/bench_tail.err
python - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_tail.json").read().splitlines() if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["kernel_ms"], d["result"]["n_new"])
PY
