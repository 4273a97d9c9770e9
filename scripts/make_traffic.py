#!/usr/bin/env python3
"""make_traffic.py OUTDIR ENTRIES VARIANT — turn the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
the bench command into the per-launch HBM traffic of the map kernel.

Corrections (MI355X_MICROARCH.md §HBM + scripts/calib_fetch.hip, profiles/r01/calib_fetch_s2.jsonl): on gfx950
FETCH_SIZE counts 128-byte line fetches at 64 bytes each, for wide coalesced streams and for the map kernel's
per-lane window bursts alike (measured 0.500-0.518 of the unique 128-B lines on 9 known patterns) -> x2.
WRITE_SIZE is taken as is: the map's record stores are whole 64-B sectors (32 B/cert expected, compare)."""
import csv
import glob
import json
import sys

out, entries, variant = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == c and "k_map" in row["Kernel_Name"]:
                v.append(float(row["Counter_Value"]))
    vals[c] = v
res = {"entries": entries, "map_variant": variant, "launches": len(vals["FETCH_SIZE"])}
if vals["FETCH_SIZE"]:
    f = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"])
    res["FETCH_SIZE_KB_per_launch"] = f
    res["fetch_bytes"] = 2.0 * f * 1024.0
if vals["WRITE_SIZE"]:
    w = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"])
    res["WRITE_SIZE_KB_per_launch"] = w
    res["write_bytes"] = w * 1024.0
if "fetch_bytes" in res and "write_bytes" in res:
    res["traffic_bytes"] = res["fetch_bytes"] + res["write_bytes"]
    res["traffic_bytes_per_cert"] = res["traffic_bytes"] / entries
print(json.dumps(res))
