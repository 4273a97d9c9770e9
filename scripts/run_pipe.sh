#!/bin/bash
# Pipelined-probe map kernel (sweep variants 17/18/19 = 4/2/8 groups per wave) against the default (15), one box.
# Fails fast: correctness first (a small parity suite on the variant), then the timings.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pipe; mkdir -p $OUT; rm -f $OUT/*
cd $R
export CTMR_LIB=$R/ct_mapreduce_amd/libctmr_sweep.so
if [ -n "${PYTEST:-}" ]; then
  CTMR_MAP_VARIANT=${PYTEST_VARIANT:-17} timeout ${PYTEST_TIMEOUT:-240} python -m pytest $PYTEST -m gpu -x -q -p no:cacheprovider 2>&1 | tail -8 | tee -a $OUT/summary.txt
  grep -q "failed\|error" $OUT/summary.txt && exit 1
fi
for v in ${VARIANTS:-15 17 18 19 15 17}; do
  timeout 120 python bench.py --no-cpu --traffic off --steps 8 --warmup 2 --variant $v ${BENCH_ARGS:-} > $OUT/b_$v.json 2> $OUT/b_$v.err || { echo "variant $v failed"; tail -3 $OUT/b_$v.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$OUT/b_$v.json')); print('variant $v', 'map_ms', round(d['kernel_ms']['map'],3), 'step', round(d['ms_per_step'],2))" | tee -a $OUT/summary.txt
done
if [ -n "${PARITY_VARIANT:-}" ]; then
  timeout 200 python bench.py --traffic off --steps 3 --warmup 1 --variant $PARITY_VARIANT > $OUT/parity_$PARITY_VARIANT.json 2> $OUT/parity_$PARITY_VARIANT.err
  python3 -c "
import json; d=json.load(open('$OUT/parity_$PARITY_VARIANT.json')); print('variant $PARITY_VARIANT sample parity', d.get('parity_vs_oracle_on_sample'), d.get('parity_sample'))" | tee -a $OUT/summary.txt
fi
