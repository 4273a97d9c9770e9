#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2c; mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_errors.py tests/test_gpu_exchange.py -m gpu -q --maxfail=10 -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.txt
tail -30 $OUT/pytest_gpu.txt
timeout 600 python scripts/host_batch_latency.py > $OUT/host_batch_latency.jsonl 2> $OUT/host_batch_latency.err; echo "latency rc $?"; cat $OUT/host_batch_latency.jsonl; tail -3 $OUT/host_batch_latency.err
timeout 600 python bench.py --global-dedup owner --no-cpu --steps 3 > $OUT/bench_gd_owner.json 2> $OUT/bench_gd_owner.err; python -c "
import json; d=json.load(open('$OUT/bench_gd_owner.json')); print('owner', d['value'], d['ms_per_step'], d['result'])"
