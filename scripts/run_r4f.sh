#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4f; mkdir -p $OUT; rm -rf $OUT/*
cd $R
VERBOSE=1 timeout 900 python scripts/fuzz_gpu_groups.py ${TRIALS:-4000} > $OUT/fuzz_gpu_groups.txt 2>&1; grep -c FAILED $OUT/fuzz_gpu_groups.txt; grep -A12 MISMATCH $OUT/fuzz_gpu_groups.txt | head -30; tail -2 $OUT/fuzz_gpu_groups.txt
timeout 600 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_rccl_transport.py tests/test_gpu_spki.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -2
