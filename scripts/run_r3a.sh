#!/bin/bash
# round 3, first GPU call: the restructured reduce (round steps), the owner-computes / Bloom redesign and the LDS-DMA fill variant
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3a; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_bloom.py tests/test_gpu_exchange.py tests/test_gpu_collisions.py tests/test_gpu_rccl_transport.py tests/test_gpu_many_issuers.py tests/test_gpu_pipeline.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_xchg.log 2>&1; tail -40 $OUT/pytest_xchg.log | cut -c1-220
[ -n "${QUICK:-}" ] && exit 0
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -5; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -30
for m in owner bloom; do
  timeout 600 python bench.py --global-dedup $m --no-cpu --steps 3 > $OUT/bench_gd_$m.json 2> $OUT/bench_gd_$m.err; python -c "
import json; d=json.load(open('$OUT/bench_gd_$m.json')); print('$m', d['value'], d['ms_per_step'], d['result']['global_dedup'])" || tail -5 $OUT/bench_gd_$m.err
done
TAGS="base glds base glds" bash scripts/run_ab.sh 2>&1 | tail -6
cp gpurun_out/ab/summary.txt $OUT/ab_glds.txt
# parity of the glds build: the parity matrix against the oracle with the variant library
CTMR_LIB=$R/ct_mapreduce_amd/libctmr_sweep_glds.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_meta.py tests/test_gpu_entries.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -2
