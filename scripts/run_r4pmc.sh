#!/bin/bash
# round 4: instruction counters of the default map kernel (one --pmc pass, no trace flags), 10 M entries
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4pmc; mkdir -p $OUT; rm -rf $OUT/*
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/p1 -o p1 --output-format csv -- python $R/bench.py --total-entries 10000000 --no-cpu --no-secondary --traffic off --steps 2 --warmup 1 > $OUT/p1.log 2>&1
f=$(find $OUT/p1 -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT/pmc_map_kernel_instruction_counts.txt
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    if "k_map_fused" not in k: continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,c in acc.items():
    launches=n[(k,"SQ_WAVES")]
    w=c["SQ_WAVES"]/launches
    print(k[:60], "launches", launches, "waves per launch", int(w))
    for name in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR"):
        print(f"  {name:18s} per wave: {c[name]/launches/w:9.1f}")
    print(f"  wave cycles per wave: {c['SQ_WAVE_CYCLES']/launches/w:9.0f}   SQ busy cycles per launch: {c['SQ_BUSY_CYCLES']/launches:.3e}")
PY
find $OUT -name "*.csv" -size +2M -delete
