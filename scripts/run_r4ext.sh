#!/bin/bash
# round 4, last GPU seconds: strict_extensions on the GPU + the suites its plumbing touches (issuer registration, strict_leaf)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4ext; mkdir -p $OUT; rm -rf $OUT/*
cd $R
timeout 400 python -m pytest tests/test_gpu_ext.py tests/test_gpu_parity.py tests/test_gpu_strings.py tests/test_gpu_spki.py tests/test_gpu_entries.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -B5 -A25 "^E  " $OUT/pytest.log | head -70
