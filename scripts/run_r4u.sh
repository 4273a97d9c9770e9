#!/bin/bash
# the driver's exact N = 8 command — torch.distributed.run, 8 ranks — on the one GPU over the stand-in librccl, final build
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4u; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
python -c "from tests.harness import build_fake_rccl; print(build_fake_rccl())" > $OUT/fake.path
J='import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])'
CTMR_RCCL_LIB=$(cat $OUT/fake.path) timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 8 --steps 3 --warmup 1 > $OUT/bench_torchrun_n8.json 2> $OUT/bench_torchrun_n8.err; python -c "$J
print('torchrun gpus 8 (one GPU, stand-in librccl)', d['value'], d['ms_per_step'], d['scaling'], d['config'].get('dedup'), d['checks'], d['parity_vs_oracle_on_sample'], d['exchange']['ms_phase_rank0'], d['exchange']['wire_bytes_sent_by_rank0_per_step'])" $OUT/bench_torchrun_n8.json || tail -12 $OUT/bench_torchrun_n8.err
