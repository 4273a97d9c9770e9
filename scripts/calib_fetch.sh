#!/bin/bash
# usage: calib_fetch.sh OUTNAME — builds scripts/calib_fetch (if needed), runs it plain (timings) and under
# rocprofv3 --pmc FETCH_SIZE (separate pass, no trace flags), prints counter per dispatch in launch order.
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; mkdir -p $OUT
[ -x $R/scripts/calib_fetch ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/scripts/calib_fetch $R/scripts/calib_fetch.hip
$R/scripts/calib_fetch > $OUT/plain.jsonl 2> $OUT/plain.err
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- $R/scripts/calib_fetch > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- $R/scripts/calib_fetch > $OUT/write.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0], float(r["Counter_Value"])))
rows.sort()
rows = [r for r in rows if r[1].startswith(("k_stream", "void k_win", "k_win"))]
plain = [json.loads(l) for l in open(out + "/plain.jsonl") if l.startswith("{") and "kernel" in l]
# every case is launched twice (warm-up + timed); keep the second dispatch of each pair
disp = rows[1::2]
with open(out + "/calib_summary.jsonl", "w") as fo:
    for p, (d, k, v) in zip(plain, disp):
        p["FETCH_SIZE_KB"] = v
        p["fetch_bytes"] = v * 1024
        ref = p.get("unique128", p.get("bytes"))
        p["fetch_over_unique128"] = v * 1024 / ref
        if "unique64" in p:
            p["fetch_over_unique64"] = v * 1024 / p["unique64"]
        fo.write(json.dumps(p) + "\n")
        print(json.dumps(p))
PY
python3 - $OUT <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/write/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "WRITE_SIZE":
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0], float(r["Counter_Value"])))
rows.sort()
plain = {}
for l in open(out + "/plain.jsonl"):
    if l.startswith("{") and "kernel" in l:
        d = json.loads(l)
        plain.setdefault(d["kernel"], []).append(d)
with open(out + "/calib_summary.jsonl", "a") as fo:
    fills = [r for r in rows if "k_fill" in r[1]]
    if fills:
        d = dict(plain["k_fill"][0]); d["WRITE_SIZE_KB"] = fills[-1][2]; d["write_over_bytes"] = fills[-1][2] * 1024 / d["bytes"]
        fo.write(json.dumps(d) + "\n"); print(json.dumps(d))
    rmw = [r for r in rows if "k_rand_rmw" in r[1]]
    for d, r in zip(plain.get("k_rand_rmw", []), rmw):
        d = dict(d); d["WRITE_SIZE_KB"] = r[2]; d["write_bytes_per_key"] = r[2] * 1024 / d["keys"]
        fo.write(json.dumps(d) + "\n"); print(json.dumps(d))
PY
find $OUT -name "*.csv" -size +1M -delete
