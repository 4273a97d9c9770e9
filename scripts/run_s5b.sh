#!/bin/bash
# default bench with the all-core cpu_baseline leg, then both global-dedup modes at N=1 again (persistent buffers)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s5; mkdir -p $OUT
cd $R
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/bench_default.err
python - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_default.json").read().splitlines() if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
print(json.dumps(d["cpu_baseline"], indent=1))
print(d.get("parity_vs_oracle_on_sample"))
PY
for mode in bloom owner; do
  timeout 900 python bench.py --no-cpu --global-dedup $mode > $OUT/bench_gd2_$mode.json 2> $OUT/bench_gd2_$mode.err
  python - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_gd2_$mode.json").read().splitlines() if l.startswith("{")][-1])
print("$mode", d["ms_per_step"], d["value"], d["result"]["global_dedup"])
PY
done
