#!/bin/bash
# Bloom pre-filter variant of the global dedup: GPU parity tests, the N=1 bench of both global-dedup modes, and the
# N=2 code path with two ranks sharing the one GPU of a gpurun box over gloo (functional check, not a measurement)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s5; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_bloom.py tests/test_gpu_exchange.py -x -q -m gpu > $OUT/pytest_bloom.txt 2>&1
tail -15 $OUT/pytest_bloom.txt
for mode in owner bloom; do
  timeout 900 python bench.py --no-cpu --global-dedup $mode > $OUT/bench_gd_$mode.json 2> $OUT/bench_gd_$mode.err
  tail -2 $OUT/bench_gd_$mode.err; cut -c1-400 $OUT/bench_gd_$mode.json
  python - <<PY
import json
d = json.load(open("$OUT/bench_gd_$mode.json"))
print("$mode", d["ms_per_step"], d["value"], d["result"])
PY
done
for mode in owner bloom; do
  CTMR_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2951$([ $mode = owner ] && echo 3 || echo 4) bench.py --gpus 2 --steps 2 --warmup 1 --entries 4000000 --no-cpu --global-dedup $mode > $OUT/bench_n2_gloo_$mode.json 2> $OUT/bench_n2_gloo_$mode.err
  tail -3 $OUT/bench_n2_gloo_$mode.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_n2_gloo_$mode.json"))
print("n2 $mode", d["ms_per_step"], d["value"], d["result"])
PY
done
