#!/usr/bin/env python3
"""Static instruction counts of one kernel per source line, from `hipcc -S -gline-tables-only` output.
   python scripts/asm_by_line.py FILE.s KERNEL_SUBSTRING [top N [CHAIN_FILTER]]
Counts every instruction under the innermost `.loc` that precedes it (inlined code keeps its own lines); prints the
heaviest (file, line) pairs with VALU / SALU / LDS / VMEM splits and a per-file total.  Static counts: a loop body counts
once — read it next to the PMC per-wave totals (scripts/pmc_per_wave.py)."""
import re
import sys
from collections import defaultdict

path, needle = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
only = sys.argv[4] if len(sys.argv) > 4 else None   # count only instructions whose inline chain holds this "file:line:" (e.g. the hot call site)
keep = True
files, cur, inside = {}, None, False
cnt = defaultdict(lambda: [0, 0, 0, 0, 0])
outer = defaultdict(lambda: [0, 0, 0, 0, 0])
cur_outer = None
for line in open(path, errors="replace"):
    s = line.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    if s.startswith("_Z") and needle in s and s.split(";")[0].strip().endswith(":"):
        inside = True
        continue
    if inside and s.startswith(".Lfunc_end"):
        break
    if not inside:
        continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        # the inline chain in the comment: innermost first.  `outer` = the outermost frame inside der_walk.h (the line of
        # walk_cert — or of the function walk_cert calls — the instruction was inlined into), else the innermost frame
        chain = re.findall(r"([A-Za-z_0-9]+\.(?:h|hip|inc)):(\d+):\d+", s)
        dw = [(f, int(l)) for f, l in chain if f == "der_walk.h"]
        cur_outer = dw[-1] if dw else ((chain[-1][0], int(chain[-1][1])) if chain else None)
        keep = only is None or (only in s)
        continue
    if not s or s[0] in ".;" or s.endswith(":"):
        continue
    if not keep:
        continue
    op = s.split()[0]
    k = 0 if op.startswith("v_") else 1 if op.startswith("s_") else 2 if op.startswith("ds_") else 3 if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else 4
    cnt[cur][k] += 1
    outer[cur_outer][k] += 1
tot = defaultdict(lambda: [0, 0, 0, 0, 0])
for (f, l), c in cnt.items():
    for i in range(5):
        tot[files.get(f, f)][i] += c[i]
print("per file: VALU SALU LDS VMEM other")
for f, c in sorted(tot.items(), key=lambda t: -sum(t[1])):
    print("  %-22s %6d %6d %5d %5d %5d" % (f, *c))
print("heaviest lines:")
for (f, l), c in sorted(cnt.items(), key=lambda t: -sum(t[1]))[:top]:
    print("  %-22s:%-5d %5d %5d %4d %4d" % (files.get(f, f), l, c[0], c[1], c[2], c[3]))
print("by the outermost der_walk.h frame (the line of walk_cert the code was inlined into), else the innermost frame:")
for key, c in sorted(outer.items(), key=lambda t: -sum(t[1]))[:top]:
    print("  %-28s %5d %5d %4d %4d" % ("%s:%s" % key if key else "?", c[0], c[1], c[2], c[3]))
