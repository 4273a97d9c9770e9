"""Multi-GPU driver: one process per GPU, the CT-entry stream sharded by log-index range.

Mirrors how the reference splits work between processes (`-offset/-limit`,
cmd/ct-fetch/ct-fetch.go:288-305, sharing state through Redis): rank r maps entries
[r·E, (r+1)·E) into its own in-HBM known-certificate table; the per-issuer unique counts
(cmd/storage-statistics/storage-statistics.go:44-53) are merged with ONE all-reduce (RCCL over xGMI
with backend "nccl"; "gloo" in the CPU tests).  Exact whenever no key spans two shards (BASELINE
config 4); the cross-shard key exchange for global dedup is the next row.

`map_fn` is injectable so that the shard arithmetic and the merge can be tested without a GPU
(tests/test_distributed_cpu.py passes an oracle-backed map_fn; production passes Engine.map_batch).
"""
from dataclasses import dataclass

import numpy as np


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous log-index range of `rank` — the remainder goes to the lowest ranks."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@dataclass
class ShardResult:
    lo: int
    hi: int
    local_counts: np.ndarray     # u64[n_issuers] unique (expDate, serial) per issuer in this shard
    global_counts: np.ndarray    # after the all-reduce
    n_new_local: int
    n_new_global: int


def merge_counts(local_counts: np.ndarray, device=None):
    """All-reduce (sum) of the per-issuer count vector over the default process group."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(local_counts).astype(np.int64))
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(np.uint64)


def run_sharded(n_total: int, n_issuers: int, make_batch, map_fn, counts_fn, device=None) -> ShardResult:
    """make_batch(lo, hi) → batch; map_fn(batch) → n_new; counts_fn() → u64[n_issuers]."""
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_range(n_total, rank, world)
    n_new = int(map_fn(make_batch(lo, hi)))
    local = np.asarray(counts_fn(), dtype=np.uint64)[:n_issuers]
    glob = merge_counts(local, device)
    tot = merge_counts(np.array([n_new], dtype=np.uint64), device)
    return ShardResult(lo, hi, local, glob, n_new, int(tot[0]))
