#!/bin/bash
# is the map kernel's slow mode a matter of WHERE the buffers lie?  only informative on a box that shows the slow mode
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4z; mkdir -p $OUT; rm -rf $OUT/*
cd $R
J='import json,sys; d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])'
run() {  # tag, env
  env $2 timeout 300 python bench.py --no-cpu --no-secondary --traffic off --steps 5 --warmup 2 > $OUT/b_$1.json 2> $OUT/b_$1.err
  python -c "$J
print('$1', round(d['ms_per_step'],2), d['roofline']['launch_ms'])" $OUT/b_$1.json | tee -a $OUT/lines.txt
}
run first X=1
slow=$(python -c "
import json; d=json.loads([l for l in open('$OUT/b_first.json').read().splitlines() if l.startswith('{')][-1]); print(1 if d['roofline']['avg_launch_ms'] > 23.0 else 0)")
if [ "$slow" = "0" ]; then echo "fast box: nothing to learn here"; exit 0; fi
run again X=1
run pad_1g CTMR_BENCH_PAD_KIB=1048576
run pad_odd CTMR_BENCH_PAD_KIB=3145772
run again2 X=1
