#!/bin/bash
# round 4: what the index size is worth now that a slot is 8 bytes — 2^27 / 2^28 (default) / 2^29 slots for the 100 M batch
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4r; mkdir -p $OUT; rm -rf $OUT/*
cd $R
J='import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])'
for s in 28 27 29 28 27; do
  timeout 600 python bench.py --table-slots-log2 $s --no-cpu --no-secondary --traffic off --steps 6 --warmup 2 > $OUT/b_$s.json 2> $OUT/b_$s.err; python -c "$J
print('slots 2^$s', d['value'], d['ms_per_step'], d['kernel_ms'], d['checks']['entries_disagreeing_with_generator'])" $OUT/b_$s.json | tee -a $OUT/lines.txt || tail -3 $OUT/b_$s.err
done
