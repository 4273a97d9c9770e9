#!/bin/bash
# how the cpu_baseline leg scales over host threads on the GPU box
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s5; mkdir -p $OUT
cd $R
for f in /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us /sys/fs/cgroup/cpuset.cpus.effective; do echo "$f: $(cat $f 2>&1)"; done | tee $OUT/cpu_scaling.txt
nproc | tee -a $OUT/cpu_scaling.txt; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)" | tee -a $OUT/cpu_scaling.txt
for T in 8 32 64 128 256; do
  timeout 600 python bench.py --entries 20000000 --steps 2 --warmup 1 --cpu-sample 500000 --cpu-threads $T --cpu-sample-mt 8000000 > $OUT/cpu_T$T.json 2> $OUT/cpu_T$T.err
  python - <<PY | tee -a $OUT/cpu_scaling.txt
import json
d = json.loads([l for l in open("$OUT/cpu_T$T.json").read().splitlines() if l.startswith("{")][-1])
c = d["cpu_baseline"]
print("threads", c["cores"], "certs/s", round(c["value"]), "one core", round(c["one_core"]["value"]), c["sample"][-20:])
PY
done
