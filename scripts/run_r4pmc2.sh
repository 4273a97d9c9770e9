#!/bin/bash
# round 4: the map kernel's instruction counters per configuration (one --pmc pass each, no trace flags), 10 M entries
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4pmc2; mkdir -p $OUT; rm -rf $OUT/*
i=0
for m in "" "--no-strict-spki" "--strict-strings" "--mixed" "--mixed --no-strict-spki" "--meta"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES -d $OUT/p$i -o p$i --output-format csv -- python $R/bench.py $m --total-entries 10000000 --no-cpu --no-secondary --traffic off --steps 2 --warmup 1 > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" "default $m" <<'PY' | tee -a $OUT/pmc_map_kernel_by_configuration.txt
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    if "k_map_fused" not in k and "k_ec_resolve" not in k: continue
    k=k.split("(")[0].replace("void ctmr::","")
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,c in sorted(acc.items()):
    L=n[(k,"SQ_WAVES")]; w=c["SQ_WAVES"]/L
    if w < 1: continue
    print(f"{sys.argv[2]:28s} {k:40s} waves {int(w):8d}  VALU {c['SQ_INSTS_VALU']/L/w:8.1f}  SALU {c['SQ_INSTS_SALU']/L/w:8.1f}  LDS {c['SQ_INSTS_LDS']/L/w:6.1f}  VMEM_RD {c['SQ_INSTS_VMEM_RD']/L/w:6.1f}  cycles/wave {c['SQ_WAVE_CYCLES']/L/w:8.0f}")
PY
  find $OUT/p$i -name "*.csv" -size +1M -delete
done
