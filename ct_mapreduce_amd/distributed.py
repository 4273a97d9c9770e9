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


# ------------------------------------------------------------------------------------------------
# Global dedup across GPUs: owner-computes key exchange (BASELINE config 5).
#
#   phase 1  export   map the shard, write one 64-byte key record per PASS entry, partitioned by
#                     owner = hash(key) mod world                       (Engine.exchange_export)
#   exchange A        partitions → owners (RCCL send/recv; sender-rank order = global log order)
#   phase 2  insert   owner inserts what it received, one "was unknown" byte per key
#                                                                        (Engine.exchange_insert)
#   exchange B        bytes → senders
#   phase 3  apply    sender flags its records, compacts new_idx        (Engine.exchange_apply)
#   counts            owner-local per-issuer counters, all-reduced as in the shard-local mode
#
# The per-rank phases are methods so that a test can drive several "ranks" inside one process
# (tests/test_gpu_exchange.py: two engines on one GPU); run_global_dedup() drives one rank over
# torch.distributed.

def _share_stream(engine, torch_device):
    """Run the engine's kernels on torch's current stream of that device: the buffers torch allocates/zeroes and the
    collectives' waits are ordered on that stream, so the engine's reads and writes of them must be too."""
    if getattr(torch_device, "type", "cpu") == "cuda" and hasattr(engine, "set_stream"):
        import torch
        engine.set_stream(torch.cuda.current_stream(torch_device).cuda_stream)


class _Buffers:
    """Grow-only byte buffers kept across rounds (a fresh multi-GB torch.empty per round would put allocator work and
    first-touch page faults inside every step)."""

    def _buf(self, name, nbytes):
        import torch
        have = getattr(self, "_b_" + name, None)
        if have is None or have.numel() < nbytes:
            have = None
            setattr(self, "_b_" + name, None)              # release before growing
            have = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=self.dev)
            setattr(self, "_b_" + name, have)
        return have[:max(nbytes, 1)]


class GlobalDedupRank(_Buffers):
    KEY = 64

    def __init__(self, engine, rank, world, torch_device):
        self.eng, self.rank, self.world, self.dev = engine, rank, world, torch_device
        _share_stream(engine, torch_device)

    def export(self, d_payload, d_offsets, d_iss, d_et, n, d_records):
        import torch
        self.n, self.d_records = n, d_records
        self.keys = self._buf("keys", max(n, 1) * self.KEY)
        self.send_counts = self.eng.exchange_export(d_payload, d_offsets, d_iss, d_et, n, d_records,
                                                    self.world, self.keys.data_ptr())
        self.n_keys = sum(self.send_counts)
        return self.send_counts

    def partition(self, owner):
        """Key records destined to `owner` (a view of the export buffer)."""
        lo = sum(self.send_counts[:owner]) * self.KEY
        return self.keys[lo:lo + self.send_counts[owner] * self.KEY]

    def insert(self, received, recv_counts):
        """received: key records concatenated in sender-rank order."""
        import torch
        self.recv_counts = list(recv_counts)
        nrecv = sum(recv_counts)
        self.flags_out = self._buf("flags_out", nrecv)       # the insert writes every byte
        self.n_new_owned = self.eng.exchange_insert(received.data_ptr(), nrecv, self.flags_out.data_ptr()) \
            if nrecv else 0
        return self.flags_out

    def flags_for(self, sender):
        lo = sum(self.recv_counts[:sender])
        return self.flags_out[lo:lo + self.recv_counts[sender]]

    def apply(self, flags_mine, d_new_idx=0):
        """flags_mine: one byte per exported key, in export (owner-major) order."""
        return self.eng.exchange_apply(self.d_records, self.n, self.keys.data_ptr(),
                                       flags_mine.data_ptr(), self.n_keys, d_new_idx)


def run_global_dedup(rank_obj: GlobalDedupRank, d_payload, d_offsets, d_iss, d_et, n, d_records,
                     d_new_idx=0):
    """One rank's step over torch.distributed (backend nccl = RCCL)."""
    import torch
    import torch.distributed as dist
    world, rank, dev = rank_obj.world, rank_obj.rank, rank_obj.dev
    send_counts = rank_obj.export(d_payload, d_offsets, d_iss, d_et, n, d_records)
    if world == 1:
        flags = rank_obj.insert(rank_obj.partition(0), [send_counts[0]])
        return rank_obj.apply(flags, d_new_idx)
    sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    allc = [torch.empty_like(sc) for _ in range(world)]
    dist.all_gather(allc, sc)
    recv_counts = [int(allc[s][rank].item()) for s in range(world)]
    K = GlobalDedupRank.KEY
    recv = rank_obj._buf("recv", sum(recv_counts) * K)
    ops, off = [], 0
    for s in range(world):                      # exchange A: key partitions
        if s == rank:                           # own partition: a local copy, no self send/recv
            if recv_counts[s]:
                recv[off * K:(off + recv_counts[s]) * K].copy_(rank_obj.partition(s))
            off += recv_counts[s]
            continue
        if recv_counts[s]:
            ops.append(dist.P2POp(dist.irecv, recv[off * K:(off + recv_counts[s]) * K], s))
        off += recv_counts[s]
        if send_counts[s]:
            ops.append(dist.P2POp(dist.isend, rank_obj.partition(s), s))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    rank_obj.insert(recv, recv_counts)
    flags_mine = rank_obj._buf("flags_mine", sum(send_counts))
    ops, off = [], 0
    for s in range(world):                      # exchange B: flags back, export order = owner-major
        if s == rank:
            if send_counts[s]:
                flags_mine[off:off + send_counts[s]].copy_(rank_obj.flags_for(s))
            off += send_counts[s]
            continue
        if send_counts[s]:
            ops.append(dist.P2POp(dist.irecv, flags_mine[off:off + send_counts[s]], s))
        off += send_counts[s]
        if recv_counts[s]:
            ops.append(dist.P2POp(dist.isend, rank_obj.flags_for(s), s))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    return rank_obj.apply(flags_mine, d_new_idx)


def run_simulated(rank_objs, shards, new_idx_ptrs=None):
    """Drive all ranks of a world inside ONE process (tests): shards[r] = (d_payload, d_offsets,
    d_iss, d_et, n, d_records).  Same data movement as run_global_dedup, with tensor slicing
    instead of send/recv."""
    import torch
    world = len(rank_objs)
    counts = [r.export(*shards[k]) for k, r in enumerate(rank_objs)]
    for o, r in enumerate(rank_objs):
        parts = [rank_objs[s].partition(o) for s in range(world)]
        r.insert(torch.cat(parts) if world > 1 else parts[0], [counts[s][o] for s in range(world)])
    out = []
    for k, r in enumerate(rank_objs):
        fl = [rank_objs[o].flags_for(k) for o in range(world)]
        out.append(r.apply(torch.cat(fl) if world > 1 else fl[0],
                           new_idx_ptrs[k] if new_idx_ptrs else 0))
    return out


# ------------------------------------------------------------------------------------------------
# Global dedup across GPUs, Bloom pre-filter variant (BASELINE north_star "all-gather of per-GPU Bloom fingerprints",
# SURVEY.md §8(e)(i)) — exact, identical results to the owner-computes exchange above.
#
#   phase 1  map      the ordinary fused map + insert into the rank's OWN table     (Engine.map_batch_device)
#   phase 2  add      locally-new keys → this rank's cumulative Bloom filter        (Engine.bloom_add)
#   all-gather        the filters (n_words × 8 bytes per rank) — the only traffic for keys no other rank holds
#   phase 3  probe    locally-new keys × the other ranks' filters → one key record per (key, peer that may hold it)
#                                                                                   (Engine.bloom_probe)
#   exchange A        key records → those peers
#   phase 4  lookup   exact, read-only: "known here before you" byte per record     (Engine.bloom_lookup)
#   exchange B        bytes → askers
#   phase 5  apply    flagged entries lose WasUnknown and leave the per-issuer count (Engine.bloom_apply)
#   counts            per-issuer counters all-reduced as in the shard-local mode
#
# Against the owner-computes exchange: every key record travels there (64 B × (G−1)/G of all PASS entries); here the
# filters travel (2 B per key held, to each of the G−1 peers) plus records for cross-rank duplicates and ≈0.5 % false
# positives per peer.  Fewer bytes for big single rounds and the first rounds of a stream; a long stream's cumulative
# filter outgrows the per-round key traffic after about three equal rounds (DESIGN.md §8 has the arithmetic).

class BloomDedupRank(_Buffers):
    KEY = 64

    def __init__(self, engine, rank, world, torch_device, bloom_bits):
        import torch
        self.eng, self.rank, self.world, self.dev = engine, rank, world, torch_device
        _share_stream(engine, torch_device)
        self.n_words = bloom_bits // 64
        # all-gather buffer, rank-major; this rank's filter IS row `rank` (caller-owned filter memory)
        self.filters = torch.zeros((world, self.n_words), dtype=torch.int64, device=torch_device)
        engine.bloom_config(bloom_bits, self.filters[rank].data_ptr())

    def own_filter(self):
        return self.filters[self.rank]

    def map(self, d_payload, d_offsets, d_iss, d_et, n, d_records, d_ends=0, order_base=0, view=None,
            blob_bytes=0):
        """Phases 1 + 2.  Packed batch, or an entry view (view = N.EntryView over d_payload = the blob)."""
        self.batch = (d_payload, d_offsets, d_ends, n, d_records)
        self.order_base = order_base
        if view is not None:
            st = self.eng.map_view_device(d_payload, blob_bytes, view, n, d_records)
        elif n:
            st = self.eng.map_batch_device(d_payload, d_offsets, d_iss, d_et, n, d_records)
        else:
            st = None
        self.eng.bloom_add(d_payload, d_offsets, d_ends, n, d_records)
        return st

    def probe(self):
        """Phase 3 (after the all-gather filled self.filters) → counts per peer."""
        import torch
        d_payload, d_offsets, d_ends, n, d_records = self.batch
        cap = max(1024, n // 4, getattr(self, "_cap", 0))
        while True:
            self.keys = self._buf("keys", cap * self.KEY)
            counts, fits = self.eng.bloom_probe(d_payload, d_offsets, d_ends, n, d_records,
                                                self.filters.data_ptr(), self.world, self.rank, self.order_base,
                                                self.keys.data_ptr(), cap)
            if fits:
                break
            cap = sum(counts)
        self._cap = cap
        self.send_counts = counts
        self.n_keys = sum(counts)
        return counts

    def partition(self, peer):
        lo = sum(self.send_counts[:peer]) * self.KEY
        return self.keys[lo:lo + self.send_counts[peer] * self.KEY]

    def lookup(self, received, recv_counts):
        """Phase 4.  received: key records concatenated in asker-rank order."""
        import torch
        self.recv_counts = list(recv_counts)
        nrecv = sum(recv_counts)
        self.flags_out = self._buf("flags_out", nrecv)       # the lookup writes every byte
        if nrecv:
            self.eng.bloom_lookup(received.data_ptr(), nrecv, self.order_base, self.flags_out.data_ptr())
        return self.flags_out

    def flags_for(self, asker):
        lo = sum(self.recv_counts[:asker])
        return self.flags_out[lo:lo + self.recv_counts[asker]]

    def apply(self, flags_mine, d_new_idx=0):
        """Phase 5.  flags_mine: one byte per exported record, in export (peer-major) order."""
        _, _, _, n, d_records = self.batch
        return self.eng.bloom_apply(d_records, n, self.keys.data_ptr(), flags_mine.data_ptr(), self.n_keys,
                                    d_new_idx)


def _all_to_all_v(rank, world, part_fn, send_counts, recv_counts, out, unit):
    """part_fn(p) → tensor for peer p (send_counts[p] × unit bytes); out ← what the peers sent, sender-rank order."""
    import torch.distributed as dist
    ops, off = [], 0
    for s in range(world):
        if s == rank:                               # own partition: a local copy, no self send/recv
            if recv_counts[s]:
                out[off * unit:(off + recv_counts[s]) * unit].copy_(part_fn(s))
            off += recv_counts[s]
            continue
        if recv_counts[s]:
            ops.append(dist.P2POp(dist.irecv, out[off * unit:(off + recv_counts[s]) * unit], s))
        off += recv_counts[s]
        if send_counts[s]:
            ops.append(dist.P2POp(dist.isend, part_fn(s), s))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()


def run_bloom_dedup(rank_obj: BloomDedupRank, d_payload, d_offsets, d_iss, d_et, n, d_records, d_new_idx=0,
                    order_base=0, d_ends=0, view=None, blob_bytes=0):
    """One rank's round over torch.distributed (backend nccl = RCCL; gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist
    world, rank, dev = rank_obj.world, rank_obj.rank, rank_obj.dev
    rank_obj.map(d_payload, d_offsets, d_iss, d_et, n, d_records, d_ends, order_base, view, blob_bytes)
    if world > 1:                                   # the Bloom all-gather
        mine = rank_obj.own_filter().clone()
        if "nccl" in str(dist.get_backend()):
            dist.all_gather_into_tensor(rank_obj.filters, mine)
        else:
            rows = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(rows, mine)
            for p in range(world):
                if p != rank:
                    rank_obj.filters[p].copy_(rows[p])
    send_counts = rank_obj.probe()
    if world == 1:
        return rank_obj.apply(rank_obj._buf("flags_mine", 1), d_new_idx)
    sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    allc = [torch.empty_like(sc) for _ in range(world)]
    dist.all_gather(allc, sc)
    recv_counts = [int(allc[s][rank].item()) for s in range(world)]
    K = BloomDedupRank.KEY
    recv = rank_obj._buf("recv", sum(recv_counts) * K)
    _all_to_all_v(rank, world, rank_obj.partition, send_counts, recv_counts, recv, K)      # exchange A
    rank_obj.lookup(recv, recv_counts)
    flags_mine = rank_obj._buf("flags_mine", sum(send_counts))
    _all_to_all_v(rank, world, rank_obj.flags_for, recv_counts, send_counts, flags_mine, 1)  # exchange B
    return rank_obj.apply(flags_mine, d_new_idx)


def run_simulated_bloom(rank_objs, shards, new_idx_ptrs=None, order_bases=None):
    """All ranks of a world inside ONE process (tests): shards[r] = (d_payload, d_offsets, d_iss, d_et, n,
    d_records).  Same data movement as run_bloom_dedup with tensor copies instead of collectives."""
    import torch
    world = len(rank_objs)
    for k, r in enumerate(rank_objs):
        r.map(*shards[k], order_base=order_bases[k] if order_bases else 0)
    for r in rank_objs:                             # "all-gather"
        for p, q in enumerate(rank_objs):
            if q is not r:
                r.filters[p].copy_(q.own_filter())
    counts = [r.probe() for r in rank_objs]
    for o, r in enumerate(rank_objs):
        parts = [rank_objs[s].partition(o) for s in range(world)]
        r.lookup(torch.cat(parts) if world > 1 else parts[0], [counts[s][o] for s in range(world)])
    out = []
    for k, r in enumerate(rank_objs):
        fl = [rank_objs[o].flags_for(k) for o in range(world)]
        fl = torch.cat(fl) if world > 1 else fl[0]
        if fl.numel() == 0:
            fl = torch.zeros(1, dtype=torch.uint8, device=r.dev)
        out.append(r.apply(fl, new_idx_ptrs[k] if new_idx_ptrs else 0))
    return out


# ------------------------------------------------------------------------------------------------
# Raw get-entries shards + global dedup: the key records carry issuer INDICES, so every rank's issuer table must
# list the same certificates in the same order.  Engines run with issuer auto-registration off; a decode that meets
# unregistered Chain[0] certificates fails with E_NOTFOUND and lists them; the lists of all ranks are gathered and
# the union is registered everywhere in one agreed order (bytewise), then the decode is repeated.  New issuers are
# rare (hundreds per log), so the extra round is too.
def union_in_agreed_order(pending_lists):
    """Deterministic registration order for the union of the ranks' pending lists."""
    return sorted(set(d for lst in pending_lists for d in lst))


def decode_synchronised(engines_or_engine, decode_calls, gather=None):
    """decode_calls[r]() runs rank r's decode (it may raise CtmrError E_NOTFOUND); gather(list) → list of every
    rank's list (default: torch.distributed.all_gather_object).  With a list of engines (tests: several ranks in one
    process) no collective is used.  Returns each call's result."""
    from . import _native as N
    from .engine import CtmrError
    single = not isinstance(engines_or_engine, (list, tuple))
    engines = [engines_or_engine] if single else list(engines_or_engine)
    calls = [decode_calls] if single else list(decode_calls)
    for _ in range(64):
        results, pending = [], []
        for eng, call in zip(engines, calls):
            try:
                results.append(call())
                pending.append([])
            except CtmrError as ex:
                if ex.code != N.E_NOTFOUND:
                    raise
                results.append(None)
                pending.append(eng.pending_issuers())
        if single:
            if gather is None:
                import torch.distributed as dist
                box = [None] * dist.get_world_size()
                dist.all_gather_object(box, pending[0])
                everyone = box
            else:
                everyone = gather(pending[0])
        else:
            everyone = pending
        fresh = union_in_agreed_order(everyone)
        if not fresh:
            return results[0] if single else results
        for eng in engines:
            eng.add_issuers(fresh)
    raise RuntimeError("issuer synchronisation does not converge")
