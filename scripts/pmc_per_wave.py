#!/usr/bin/env python3
"""Per-wave view of a pmc_summary.py file: every SQ counter of the named kernel divided by SQ_WAVES.
   python scripts/pmc_per_wave.py FILE [kernel substring]"""
import sys
f, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_map_fused")
rows = {}
for line in open(f):
    if sub not in line:
        continue
    if "duration_ns" in line:
        name = line.split(" duration_ns")[0].strip()
        rows.setdefault(name, {})["duration_us"] = float(line.split("mean=")[1].split()[0]) / 1e3
        continue
    head, rest = line.split(" n=", 1)
    name, ctr = head.rsplit(None, 1)
    rows.setdefault(name.strip(), {})[ctr] = float(rest.split("mean=")[1].split()[0])
for name, c in rows.items():
    w = c.get("SQ_WAVES")
    print(name, "waves", w, "duration_us", c.get("duration_us"))
    if not w:
        continue
    for k in sorted(c):
        if k.startswith("SQ") and k != "SQ_WAVES":
            print("   %-24s %12.1f per wave" % (k, c[k] / w))
