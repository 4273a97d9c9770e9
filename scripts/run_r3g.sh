#!/bin/bash
# where the 167 written bytes per certificate go: PMC WRITE_SIZE / FETCH_SIZE of the map kernel with the slot image store
# removed, and with the whole table access removed (measurement builds: results are wrong on purpose)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3g; mkdir -p $OUT; rm -rf $OUT/*
cd /tmp; export TMPDIR=/tmp
for tag in base noimg noprobe; do
  lib=$R/ct_mapreduce_amd/libctmr.so; [ $tag != base ] && lib=$R/ct_mapreduce_amd/libctmr_sweep_$tag.so
  for c in FETCH_SIZE WRITE_SIZE; do
    CTMR_LIB=$lib CTMR_BENCH_CHILD=1 timeout 300 rocprofv3 --pmc $c -d $OUT/$tag-$c -o pmc --output-format csv -- python $R/bench.py --entries 10000000 --steps 2 --warmup 1 --no-cpu --traffic off --no-secondary > /dev/null 2>&1
    python3 - <<PY
import csv,glob
vals=[]
for f in glob.glob("$OUT/$tag-$c/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="$c" and "k_map_fused" in r["Kernel_Name"]: vals.append(float(r["Counter_Value"]))
v=sum(vals)/len(vals)*1024/1e7*(2 if "$c"=="FETCH_SIZE" else 1)
print("$tag $c bytes/cert", round(v,1), "launches", len(vals))
open("$OUT/summary.txt","a").write("$tag $c bytes_per_cert %.1f (launches %d)\n" % (v,len(vals)))
PY
  done
  CTMR_LIB=$lib timeout 300 python $R/bench.py --entries 100000000 --steps 4 --warmup 1 --no-cpu --traffic off --no-secondary 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print('$tag map_ms', round(d['kernel_ms']['map'],3)); open('$OUT/summary.txt','a').write('$tag map_ms %.3f\n' % d['kernel_ms']['map'])"
  find $OUT -name "*.csv" -size +1M -delete
done
