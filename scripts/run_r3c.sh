#!/bin/bash
# round 3: strict_leaf, the N2 fixtures, the hand-written scan, trusted-log twin fix — whole -m gpu suite + a strict fuzz campaign
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3c; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_entries.py tests/test_gpu_pem.py tests/test_gpu_scale.py tests/test_gpu_pipeline.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_new.log 2>&1; tail -25 $OUT/pytest_new.log | cut -c1-250
[ -n "${QUICK:-}" ] && exit 0
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -20
STRICT_LEAF=1 timeout 900 python scripts/fuzz_gpu_entries.py ${FUZZ_N:-2000000} 7001 > $OUT/fuzz_gpu_entries_strict_leaf.txt 2>&1; tail -3 $OUT/fuzz_gpu_entries_strict_leaf.txt
