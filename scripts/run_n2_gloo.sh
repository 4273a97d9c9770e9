#!/bin/bash
# the N>1 code path of bench.py (sharding by log index, count all-reduce, barrier, max-over-ranks timing) with two
# ranks sharing the one GPU of a gpurun box over gloo — a functional check, not a measurement
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4; mkdir -p $OUT
cd $R
CTMR_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --entries 2000000 > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2_gloo.err
tail -3 $OUT/bench_n2_gloo.err; cat $OUT/bench_n2_gloo.json | cut -c1-900
CTMR_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --entries 2000000 --global-dedup > $OUT/bench_n2_gloo_gd.json 2> $OUT/bench_n2_gloo_gd.err
tail -3 $OUT/bench_n2_gloo_gd.err; cat $OUT/bench_n2_gloo_gd.json | cut -c1-600
