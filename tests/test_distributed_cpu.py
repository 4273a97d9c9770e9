"""world_size-2 gloo test of the multi-GPU host logic (shard ranges + count all-reduce).
The per-rank map is the ORACLE here (tests may use it as the checker/stand-in); on GPUs the same
driver is fed Engine.map_batch (bench.py --gpus N)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from ct_mapreduce_amd.distributed import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_the_stream():
    for n in (0, 1, 7, 100, 1001):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def _worker(rank, world, port, n_total, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ct_mapreduce_amd import synth
    from ct_mapreduce_amd.distributed import run_sharded
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = synth.config(seed=77, n_issuers=8, dup_permille=0, ca_permille=20, expired_permille=20)
    issuers = synth.issuers(cfg)
    ids = [orc.issuer_id(d[orc.parse_cert(d).spki_off:][:orc.parse_cert(d).spki_len]) for d in issuers]
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    blob = np.frombuffer(b"".join(issuers), np.uint8)
    eng = orc.Engine(b"Synth Issuer 00", False, synth.BASE_TIME)

    def map_fn(b):
        st, unk, eh = eng.batch(b.payload, b.offsets, b.issuer_idx, blob, io)
        return int(unk.sum())

    res = run_sharded(n_total, len(issuers), lambda lo, hi: synth.host_batch(cfg, lo, hi - lo), map_fn,
                      lambda: np.array([eng.issuer_count(i) for i in ids], dtype=np.uint64))
    if rank == 0:
        np.save(out, np.concatenate([[res.n_new_global], res.global_counts]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gloo_counts_match_single_process(tmp_path):
    from ct_mapreduce_amd import synth
    from oracle import oracle as orc
    n_total = 3001
    out = str(tmp_path / "counts.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, n_total, out), nprocs=2, join=True)
    got = np.load(out)
    # single process over the whole stream
    cfg = synth.config(seed=77, n_issuers=8, dup_permille=0, ca_permille=20, expired_permille=20)
    issuers = synth.issuers(cfg)
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    eng = orc.Engine(b"Synth Issuer 00", False, synth.BASE_TIME)
    b = synth.host_batch(cfg, 0, n_total)
    st, unk, eh = eng.batch(b.payload, b.offsets, b.issuer_idx, np.frombuffer(b"".join(issuers), np.uint8), io)
    ids = [orc.issuer_id(d[orc.parse_cert(d).spki_off:][:orc.parse_cert(d).spki_len]) for d in issuers]
    want = [int(unk.sum())] + [eng.issuer_count(i) for i in ids]
    assert list(got) == want


# ------------------------------------------------------------------------------------------------
# The cross-rank key exchange driver (distributed.run_global_dedup: counts all-gather, key partitions to their
# owners in sender-rank order, flags back in export order) under world_size-3 gloo.  The per-rank engine is a
# CPU stand-in built on the ORACLE's field extraction with the same three calls as Engine.exchange_* — what is
# under test is the data movement between processes, which cannot run on GPUs here (one GPU per gpurun box).
class _FakeExchangeEngine:
    KEY = 64

    def __init__(self, filt, now):
        from oracle import oracle as orc
        self.orc, self.filt, self.now = orc, filt, now
        self.known = set()
        self.flags_by_entry = None

    @staticmethod
    def _view(ptr, nbytes):
        import ctypes
        return np.ctypeslib.as_array((ctypes.c_uint8 * max(nbytes, 1)).from_address(ptr))

    def exchange_export(self, batch, _o, _i, _e, n, _r, world, d_keys_out):
        import zlib
        orc = self.orc
        parts = [[] for _ in range(world)]
        for i in range(n):
            der = batch.cert(i)
            c = orc.parse_cert(der)
            if not c.ok or orc.is_filtered_out(der, c, self.filt, False, self.now) != orc.ST_PASS:
                continue
            key = b"%d|%d|" % (orc.exp_hour(c.not_after), int(batch.issuer_idx[i])) + der[c.serial_off:c.serial_off + c.serial_len]
            rec = i.to_bytes(8, "little") + len(key).to_bytes(2, "little") + key
            assert len(rec) <= self.KEY
            parts[zlib.crc32(key) % world].append(rec.ljust(self.KEY, b"\0"))
        out = self._view(d_keys_out, n * self.KEY)
        blob = b"".join(b"".join(p) for p in parts)
        out[:len(blob)] = np.frombuffer(blob, np.uint8)
        self.n = n
        return [len(p) for p in parts]

    def exchange_insert(self, d_keys, n_keys, d_flags):
        keys, flags = self._view(d_keys, n_keys * self.KEY), self._view(d_flags, n_keys)
        new = 0
        for k in range(n_keys):
            rec = keys[k * self.KEY:(k + 1) * self.KEY].tobytes()
            key = rec[10:10 + int.from_bytes(rec[8:10], "little")]
            flags[k] = key not in self.known
            new += int(flags[k])
            self.known.add(key)
        return new

    def exchange_apply(self, _records, n, d_keys_sent, d_flags, n_keys, _new_idx):
        keys, flags = self._view(d_keys_sent, n_keys * self.KEY), self._view(d_flags, n_keys)
        self.flags_by_entry = np.zeros(n, np.uint8)
        for k in range(n_keys):
            self.flags_by_entry[int.from_bytes(keys[k * self.KEY:k * self.KEY + 8].tobytes(), "little")] = flags[k]
        return int(self.flags_by_entry.sum())


def _exchange_worker(rank, world, port, n_total, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from ct_mapreduce_amd import synth
    from ct_mapreduce_amd.distributed import GlobalDedupRank, run_global_dedup, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = synth.config(seed=78, n_issuers=8, dup_permille=300, ca_permille=20, expired_permille=20)
    eng = _FakeExchangeEngine(b"Synth Issuer 00", synth.BASE_TIME)
    ro = GlobalDedupRank(eng, rank, world, torch.device("cpu"))
    got = []
    for lo_all, hi_all in ((0, n_total // 2), (n_total // 2, n_total)):          # two waves: the owners' sets persist
        lo, hi = shard_range(hi_all - lo_all, rank, world)
        b = synth.host_batch(cfg, lo_all + lo, hi - lo)
        run_global_dedup(ro, b, 0, 0, 0, b.n, 0)
        got.append((lo_all + lo, eng.flags_by_entry.copy()))
    np.save(os.path.join(outdir, f"flags_{rank}.npy"), np.array(got, dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_three_rank_gloo_key_exchange_matches_single_process(tmp_path):
    from ct_mapreduce_amd import synth
    from oracle import oracle as orc
    n_total, world = 2400, 3
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_exchange_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    flags = np.zeros(n_total, np.uint8)
    for r in range(world):
        for first, f in np.load(tmp_path / f"flags_{r}.npy", allow_pickle=True):
            flags[first:first + len(f)] = f
    # single process, whole stream in log order: the reference loop
    cfg = synth.config(seed=78, n_issuers=8, dup_permille=300, ca_permille=20, expired_permille=20)
    issuers = synth.issuers(cfg)
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    eng = orc.Engine(b"Synth Issuer 00", False, synth.BASE_TIME)
    b = synth.host_batch(cfg, 0, n_total)
    st, unk, eh = eng.batch(b.payload, b.offsets, b.issuer_idx, np.frombuffer(b"".join(issuers), np.uint8), io)
    assert 0 < unk.sum() < (st == 0).sum()                     # there ARE cross-shard duplicates
    assert (flags == unk).all()
