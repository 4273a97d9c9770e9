#!/bin/bash
# Raw get-entries front end: fused decode+match (default) against the two-kernel form (sweep build, CTMR_RAW_SEPARATE=1).
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/raw; mkdir -p $OUT; rm -f $OUT/*
cd $R
timeout ${PYTEST_TIMEOUT:-300} python -m pytest ${PYTEST:-tests/test_gpu_entries.py} -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 | tee -a $OUT/summary.txt
grep -q "failed\|error" $OUT/summary.txt && exit 1
export CTMR_LIB=$R/ct_mapreduce_amd/libctmr_sweep.so
for mode in ${MODES:-fused separate fused separate}; do
  if [ $mode = separate ]; then export CTMR_RAW_SEPARATE=1; else unset CTMR_RAW_SEPARATE; fi
  timeout 200 python bench.py --raw --no-cpu --steps 5 --warmup 1 ${BENCH_ARGS:-} > $OUT/b_$mode.json 2> $OUT/b_$mode.err || { echo "$mode failed"; tail -3 $OUT/b_$mode.err; exit 1; }
  python3 -c "
import json; d=json.loads([l for l in open('$OUT/b_$mode.json').read().splitlines() if l.startswith('{')][-1]); print('$mode', d['value'], 'ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items()})" | tee -a $OUT/summary.txt
done
if [ -n "${FUZZ_TOTAL:-}" ]; then
  timeout ${FUZZ_TIMEOUT:-240} python scripts/fuzz_gpu_entries.py $FUZZ_TOTAL ${FUZZ_SEED:-9000} > $OUT/fuzz_entries_fused.log 2>&1; tail -3 $OUT/fuzz_entries_fused.log | tee -a $OUT/summary.txt
fi
