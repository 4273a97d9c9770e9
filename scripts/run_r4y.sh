#!/bin/bash
# 8 rank processes under torch.distributed.run on the one GPU over the stand-in librccl: owner-computes, shards whole and in 4 chunks
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4y; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
python -c "from tests.harness import build_fake_rccl; print(build_fake_rccl())" > $OUT/fake.path
J='import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])'
for k in 1 4; do
CTMR_RCCL_LIB=$(cat $OUT/fake.path) timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 2954$k bench.py --gpus 8 --dedup owner --chunks $k --steps 3 --warmup 1 > $OUT/bench_torchrun_n8_owner_chunks$k.json 2> $OUT/bench_torchrun_n8_owner_chunks$k.err; python -c "$J
print('torchrun gpus 8 owner chunks $k', d['value'], d['ms_per_step'], d['checks'], d['parity_vs_oracle_on_sample'], d['exchange']['ms_phase_rank0'], d['exchange']['wire_bytes_sent_by_rank0_per_step'])" $OUT/bench_torchrun_n8_owner_chunks$k.json || tail -12 $OUT/bench_torchrun_n8_owner_chunks$k.err
done
