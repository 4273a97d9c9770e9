#!/bin/bash
# round 4: owner-computes rounds mapped in chunks (ctmr_group_set_chunks) — the new tests, the group suites, the group fuzz
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4v; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_rccl_transport.py tests/test_gpu_bench_multirank.py tests/test_gpu_bloom.py ${EXTRA:-} -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -B5 -A40 "^E  " $OUT/pytest.log | head -120
timeout 300 python scripts/fuzz_gpu_groups.py ${TRIALS:-600} > $OUT/fuzz_gpu_groups.txt 2>&1; tail -3 $OUT/fuzz_gpu_groups.txt
