"""Multi-GPU groups: a thin ctypes binding of the native ctmr_group_* layer (include/ctmr.h, csrc/engine/group.inc).

The reference splits a log between ct-fetch processes by `-offset/-limit` (cmd/ct-fetch/ct-fetch.go:288-305) and makes
their dedup global through one Redis server (storage/rediscache.go:21-65).  Here rank r maps the entries of its
log-index shard on its GPU and ONE native call per round runs the shard maps and the exchange that makes the dedup
global — owner-computes key exchange or the all-gather of per-GPU Bloom filters as an exact pre-filter — with the same
kernels whether the ranks live in one process (`Group.local`: device-to-device copies; what the -m gpu tests drive,
several engines on the one reachable GPU) or one process per GPU (`Group.rccl`: RCCL over xGMI).  Nothing in here
moves data: no torch, no torch.distributed.  The only thing a multi-process host has to carry between its ranks itself
is the 128-byte group id (`Group.unique_id()` on one rank → every rank, over any channel it has).
"""
import ctypes as C

import numpy as np

from . import _native as N
from .engine import CtmrError

DEDUP_LOCAL, DEDUP_OWNER, DEDUP_BLOOM = N.DEDUP_LOCAL, N.DEDUP_OWNER, N.DEDUP_BLOOM
MODES = {"local": DEDUP_LOCAL, "owner": DEDUP_OWNER, "bloom": DEDUP_BLOOM}


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous log-index range of `rank` — the remainder goes to the lowest ranks."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(d_payload, d_offsets, d_issuer_idx, d_entry_type, n, d_records, d_new_idx=0, order_base=0, d_ends=0,
          blob_bytes=0) -> N.Shard:
    """One rank's input of a round (device pointers as ints; ctmr_shard)."""
    return N.Shard(d_payload=d_payload or None, d_offsets=d_offsets or None, d_ends=d_ends or None,
                   d_issuer_idx=d_issuer_idx or None, d_entry_type=d_entry_type or None, n=n, blob_bytes=blob_bytes,
                   order_base=order_base, d_records=d_records or None, d_new_idx=d_new_idx or None)


class Group:
    def __init__(self, handle, engines):
        self._lib = N.lib()
        self._h = handle
        self.engines = list(engines)      # the LOCAL ranks' engines, rank order (kept alive with the group)

    # ---- construction
    @staticmethod
    def local(engines) -> "Group":
        """Every rank in this process: one engine per device, or several engines on one device."""
        lib = N.lib()
        arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
        h = C.c_void_p()
        rc = lib.ctmr_group_create_local(arr, len(engines), C.byref(h))
        if rc != 0:
            raise CtmrError(rc, "ctmr_group_create_local failed")
        return Group(h, engines)

    @staticmethod
    def unique_id() -> bytes:
        """ncclGetUniqueId through the library: call on ONE rank, hand the bytes to all of them."""
        buf = C.create_string_buffer(N.GROUP_ID_BYTES)
        rc = N.lib().ctmr_group_unique_id(buf)
        if rc != 0:
            raise CtmrError(rc, "ctmr_group_unique_id failed (librccl not loadable?)")
        return buf.raw

    @staticmethod
    def rccl(engine, group_id: bytes, rank: int, world: int) -> "Group":
        """One process per GPU: RCCL transport."""
        assert len(group_id) == N.GROUP_ID_BYTES
        h = C.c_void_p()
        rc = N.lib().ctmr_group_create_rccl(engine._h, group_id, rank, world, C.byref(h))
        if rc != 0:
            raise CtmrError(rc, (N.lib().ctmr_last_error(engine._h) or b"").decode(errors="replace"))
        return Group(h, [engine])

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ctmr_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001  (interpreter shutdown)
            pass

    def _check(self, rc):
        if rc != 0:
            raise CtmrError(rc, (self._lib.ctmr_group_last_error(self._h) or b"").decode(errors="replace"))

    # ---- the data path
    def bloom_config(self, bits: int):
        self._check(self._lib.ctmr_group_bloom_config(self._h, bits))

    def set_chunks(self, chunks: int):
        """Owner-computes rounds map every shard in `chunks` pieces, chunk c's key records travelling while chunk c + 1
        is walked (include/ctmr.h: ctmr_group_set_chunks); results are those of the unchunked round."""
        self._check(self._lib.ctmr_group_set_chunks(self._h, int(chunks)))

    def map_batch(self, mode, shards):
        """One round.  shards: one N.Shard per local rank (rank order).  Returns the per-rank BatchStats."""
        mode = MODES.get(mode, mode)
        arr = (N.Shard * len(shards))(*shards)
        stats = (N.BatchStats * len(shards))()
        self._check(self._lib.ctmr_group_map_batch(self._h, mode, arr, stats))
        return list(stats)

    def issuer_counts(self, n: int) -> np.ndarray:
        """Σ over ALL ranks of the per-issuer unique counts (cmd/storage-statistics/storage-statistics.go:44-53)."""
        out = np.zeros(max(n, 1), np.uint64)
        self._check(self._lib.ctmr_group_issuer_counts(self._h, out.ctypes.data, n))
        return out[:n]

    def total_count(self) -> int:
        v = C.c_uint64(0)
        self._check(self._lib.ctmr_group_total_count(self._h, C.byref(v)))
        return v.value

    def info(self) -> N.GroupStats:
        st = N.GroupStats()
        self._check(self._lib.ctmr_group_info(self._h, C.byref(st)))
        return st

    # ---- what a multi-process host needs besides the data path
    def all_reduce_u64(self, values, op_max=False) -> np.ndarray:
        a = np.ascontiguousarray(values, dtype=np.uint64).copy()
        self._check(self._lib.ctmr_group_all_reduce_u64(self._h, a.ctypes.data, len(a), int(op_max)))
        return a

    def barrier(self):
        self._check(self._lib.ctmr_group_barrier(self._h))


# ------------------------------------------------------------------------------------------------
# Raw get-entries shards + global dedup: the key records carry issuer INDICES, so every rank's issuer table must
# list the same certificates in the same order.  Engines run with issuer auto-registration off; a decode that meets
# unregistered Chain[0] certificates fails with E_NOTFOUND and lists them; the lists of all ranks are gathered (by the
# host: they are a few certificates per log) and the union is registered everywhere in one agreed order (bytewise),
# then the decode is repeated.  New issuers are rare (hundreds per log), so the extra round is too.
def union_in_agreed_order(pending_lists):
    """Deterministic registration order for the union of the ranks' pending lists."""
    return sorted(set(d for lst in pending_lists for d in lst))


def decode_synchronised(engines_or_engine, decode_calls, gather=None):
    """decode_calls[r]() runs rank r's decode (it may raise CtmrError E_NOTFOUND).  With a list of engines (several
    ranks in one process) no communication is needed; with ONE engine (one process per GPU) `gather(my_list)` must
    return every rank's list — the host's own channel.  Returns each call's result."""
    single = not isinstance(engines_or_engine, (list, tuple))
    engines = [engines_or_engine] if single else list(engines_or_engine)
    calls = [decode_calls] if single else list(decode_calls)
    if single and gather is None:
        raise ValueError("one engine per process: pass gather(list) -> list of every rank's list")
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
        everyone = gather(pending[0]) if single else pending
        fresh = union_in_agreed_order(everyone)
        if not fresh:
            return results[0] if single else results
        for eng in engines:
            eng.add_issuers(fresh)
    raise RuntimeError("issuer synchronisation does not converge")
