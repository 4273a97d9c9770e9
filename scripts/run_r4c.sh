#!/bin/bash
# round 4: k_ec_resolve with pass-1 inserts + LDS compaction, OID whitelist in alg_id — tests, then default / mixed with and without the key parse
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4c; mkdir -p $OUT; rm -rf $OUT/*
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -3
run() {
  name=b$(echo "$*" | tr -d ' -')_$rep
  timeout 300 python bench.py --no-cpu --no-secondary --traffic off --steps 8 --warmup 2 "$@" > $OUT/$name.json 2> $OUT/$name.err
  python3 -c "
import json; d=json.loads([l for l in open('$OUT/$name.json').read().splitlines() if l.startswith('{')][-1]); print('$*', 'map', round(d['kernel_ms']['map'],3), 'insert', round(d['kernel_ms']['insert'],3), 'step', round(d['ms_per_step'],2), d['checks']['entries_disagreeing_with_generator'])" | tee -a $OUT/summary.txt || tail -3 $OUT/$name.err
}
for rep in 1 2; do
  run
  run --no-strict-spki
  run --mixed
  run --mixed --no-strict-spki
done
