#!/bin/bash
# the N>1 code path of bench.py (sharding by log index, count all-reduce, barrier, max-over-ranks timing; both global-dedup
# drivers) with two ranks sharing the one GPU of a gpurun box over gloo — a functional check, not a measurement
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/n2; mkdir -p $OUT
cd $R
i=0
for extra in "" "--global-dedup owner" "--global-dedup bloom" "--raw"; do
  i=$((i+1))
  CTMR_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2951$i bench.py --gpus 2 --steps 2 --warmup 1 --entries 2000000 $extra > $OUT/bench_n2_$i.json 2> $OUT/bench_n2_$i.err
  echo "rc=$? [$extra]"
  python - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_n2_$i.json").read().splitlines() if l.startswith("{")][-1])
print(d["n_gpus"], round(d["ms_per_step"], 2), round(d["value"]), d["result"], d["config"]["parallelism"])
PY
done
