#!/bin/bash
# is the first big run on a fresh box slower?  the default line three times, first thing on the box, per-launch times kept
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4n; mkdir -p $OUT; rm -rf $OUT/*
cd $R
J='import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])'
for rep in 1 2 3 4; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | head -6 > $OUT/smi_before_$rep.txt
  timeout 600 python bench.py --no-cpu --no-secondary --traffic off --steps 8 --warmup ${WARM:-2} > $OUT/b_$rep.json 2> $OUT/b_$rep.err; python -c "$J
print('run $rep', d['ms_per_step'], d['kernel_ms']['map'], d['roofline']['launch_ms'])" $OUT/b_$rep.json | tee -a $OUT/lines.txt
done
cat $OUT/smi_before_1.txt $OUT/smi_before_2.txt
