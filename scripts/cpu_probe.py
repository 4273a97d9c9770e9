"""How the oracle leg scales over host threads on this box (cgroup quota, affinity): python scripts/cpu_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from ct_mapreduce_amd import synth  # noqa: E402

for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
          "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(f, open(f).read().strip())
    except OSError as ex:
        print(f, "-", ex.__class__.__name__)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
cfg = synth.config(seed=5, n_issuers=8, dup_permille=100)
iss = synth.issuers(cfg)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
t0 = time.perf_counter()
b = synth.host_batch(cfg, 0, n)
print("host batch", n, round(time.perf_counter() - t0, 1), "s")
pay = np.concatenate([b.payload, np.zeros(64, np.uint8)])
arr = (pay, b.offsets.astype(np.uint64), b.issuer_idx.astype(np.uint32))
for T in (1, 4, 16, 64, 128, 256):
    v = max(bench.cpu_baseline_threads(arr, iss, b"", synth.BASE_TIME, n, T)[0] for _ in range(2))
    print("threads", T, "certs/s", round(v))
