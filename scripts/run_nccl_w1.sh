#!/bin/bash
# RCCL code path with a world of ONE rank (the only world a 1-GPU gpurun box allows under nccl): process-group init with
# device_id, count all-reduce, and both global-dedup drivers (all_gather_into_tensor of the Bloom filter included)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s5; mkdir -p $OUT
cd $R
i=0
for extra in "" "--global-dedup owner" "--global-dedup bloom"; do
  i=$((i+1))
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$i bench.py --gpus 2 --steps 2 --warmup 1 --entries 4000000 --no-cpu $extra > $OUT/nccl_w1_$i.json 2> $OUT/nccl_w1_$i.err
  echo "rc=$? [$extra]"; tail -2 $OUT/nccl_w1_$i.err
  python - <<PY
import json
d = json.loads([l for l in open("$OUT/nccl_w1_$i.json").read().splitlines() if l.startswith("{")][-1])
print(d["n_gpus"], d["ms_per_step"], d["value"], d["result"], d["config"]["parallelism"])
PY
done
