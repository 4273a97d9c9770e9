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


# ------------------------------------------------------------------------------------------------
# The Bloom pre-filter variant of the global dedup (distributed.run_bloom_dedup: local insert → filter all-gather →
# probe → key records to the peers whose filter matched → exact lookup → flags back → apply) under world_size-3 gloo,
# three rounds.  The per-rank engine is a CPU stand-in with the same calls as Engine.map_batch_device / bloom_*,
# built on the ORACLE's field extraction; what is under test is the protocol and the data movement between processes.
class _FakeBloomEngine:
    KEY = 64

    def __init__(self, filt, now):
        from oracle import oracle as orc
        self.orc, self.filt, self.now = orc, filt, now
        self.known = {}           # key → (epoch, batch index)
        self.epoch = 0
        self.exchanged = 0

    _view = staticmethod(_FakeExchangeEngine._view)

    @staticmethod
    def _pos(key, n_words):
        import hashlib
        g = int.from_bytes(hashlib.sha256(key).digest()[:8], "little")
        bits = 0
        for sh in (40, 46, 52, 58):
            bits |= 1 << ((g >> sh) & 63)
        return g % n_words, bits

    def bloom_config(self, bits, d_words):
        import ctypes
        self.n_words = bits // 64
        self.words = np.ctypeslib.as_array((ctypes.c_uint64 * self.n_words).from_address(d_words))
        self.words[:] = 0

    def map_batch_device(self, batch, _o, _i, _e, n, _r):
        orc = self.orc
        self.epoch += 1
        self.keys_of = [None] * n
        self.flags_by_entry = np.zeros(n, np.uint8)
        for i in range(n):
            der = batch.cert(i)
            c = orc.parse_cert(der)
            if not c.ok or orc.is_filtered_out(der, c, self.filt, False, self.now) != orc.ST_PASS:
                continue
            key = b"%d|%d|" % (orc.exp_hour(c.not_after), int(batch.issuer_idx[i])) + der[c.serial_off:c.serial_off + c.serial_len]
            if key not in self.known:
                self.known[key] = (self.epoch, i)
                self.keys_of[i] = key
                self.flags_by_entry[i] = 1

    def bloom_add(self, _p, _o, _e, n, _r):
        self.round_epoch = self.epoch if n else 0
        for i in range(n):
            if self.keys_of[i] is not None:
                w, bits = self._pos(self.keys_of[i], self.n_words)
                self.words[w] |= np.uint64(bits)

    def bloom_probe(self, _p, _o, _e, n, _r, d_filters, world, rank, order_base, d_keys_out, cap):
        import ctypes
        filters = np.ctypeslib.as_array((ctypes.c_uint64 * (world * self.n_words)).from_address(d_filters))
        parts = [[] for _ in range(world)]
        for i in range(n):
            key = self.keys_of[i]
            if key is None:
                continue
            w, bits = self._pos(key, self.n_words)
            for p in range(world):
                if p != rank and (int(filters[p * self.n_words + w]) & bits) == bits:
                    rec = i.to_bytes(8, "little") + (order_base + i).to_bytes(8, "little") + len(key).to_bytes(2, "little") + key
                    assert len(rec) <= self.KEY
                    parts[p].append(rec.ljust(self.KEY, b"\0"))
        counts = [len(p) for p in parts]
        if sum(counts) > cap:
            return counts, False
        blob = b"".join(b"".join(p) for p in parts)
        self._view(d_keys_out, cap * self.KEY)[:len(blob)] = np.frombuffer(blob, np.uint8)
        self.exchanged += sum(counts)
        return counts, True

    def bloom_lookup(self, d_keys, n_keys, order_base, d_flags):
        keys, flags = self._view(d_keys, n_keys * self.KEY), self._view(d_flags, n_keys)
        for k in range(n_keys):
            rec = keys[k * self.KEY:(k + 1) * self.KEY].tobytes()
            order = int.from_bytes(rec[8:16], "little")
            key = rec[18:18 + int.from_bytes(rec[16:18], "little")]
            hit = self.known.get(key)
            flags[k] = hit is not None and (hit[0] != self.round_epoch or order_base + hit[1] < order)

    def bloom_apply(self, _records, n, d_keys_sent, d_flags, n_keys, _new_idx):
        keys, flags = self._view(d_keys_sent, n_keys * self.KEY), self._view(d_flags, n_keys)
        for k in range(n_keys):
            if flags[k]:
                self.flags_by_entry[int.from_bytes(keys[k * self.KEY:k * self.KEY + 8].tobytes(), "little")] = 0
        return int(self.flags_by_entry.sum())


def _bloom_worker(rank, world, port, n_total, bits, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from ct_mapreduce_amd import synth
    from ct_mapreduce_amd.distributed import BloomDedupRank, run_bloom_dedup, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = synth.config(seed=79, n_issuers=8, dup_permille=300, ca_permille=20, expired_permille=20)
    eng = _FakeBloomEngine(b"Synth Issuer 00", synth.BASE_TIME)
    ro = BloomDedupRank(eng, rank, world, torch.device("cpu"), bits)
    got = []
    third = n_total // 3
    for lo_all, hi_all in ((0, third), (third, 2 * third), (2 * third, n_total)):   # tables and filters persist
        lo, hi = shard_range(hi_all - lo_all, rank, world)
        # rank 2 sits out the second round: an empty batch must not disturb the protocol
        n = 0 if (rank == 2 and lo_all == third) else hi - lo
        b = synth.host_batch(cfg, lo_all + lo, n)
        run_bloom_dedup(ro, b, 0, 0, 0, b.n, 0, order_base=lo_all + lo)
        got.append((lo_all + lo, eng.flags_by_entry.copy() if n else np.zeros(0, np.uint8), n))
    np.save(os.path.join(outdir, f"bloom_{rank}.npy"), np.array(got, dtype=object), allow_pickle=True)
    with open(os.path.join(outdir, f"bloom_exchanged_{rank}.txt"), "w") as f:
        f.write(str(eng.exchanged))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("bits", [1 << 16, 1 << 9])
def test_three_rank_gloo_bloom_dedup_matches_single_process(tmp_path, bits):
    from ct_mapreduce_amd import synth
    from oracle import oracle as orc
    n_total, world = 2400, 3
    port = 33500 + (os.getpid() % 2000) + (1 if bits == 1 << 9 else 0)
    mp.spawn(_bloom_worker, args=(world, port, n_total, bits, str(tmp_path)), nprocs=world, join=True)
    flags = np.zeros(n_total, np.uint8)
    seen = np.zeros(n_total, bool)
    for r in range(world):
        for first, f, n in np.load(tmp_path / f"bloom_{r}.npy", allow_pickle=True):
            flags[first:first + n] = f
            seen[first:first + n] = True
    # single process, the entries the ranks saw, in log order: the reference loop
    cfg = synth.config(seed=79, n_issuers=8, dup_permille=300, ca_permille=20, expired_permille=20)
    issuers = synth.issuers(cfg)
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    eng = orc.Engine(b"Synth Issuer 00", False, synth.BASE_TIME)
    b = synth.host_batch(cfg, 0, n_total)
    unk = np.zeros(n_total, np.uint8)
    idx = np.nonzero(seen)[0]
    assert 0 < len(idx) < n_total
    n_pass = 0
    for i in idx:                                              # skipped entries never reach the reference loop
        st_i, unk[i], _ = eng.entry(b.cert(int(i)), issuers[int(b.issuer_idx[i])])
        n_pass += st_i == orc.ST_PASS
    assert 0 < unk.sum()
    assert (flags == unk).all(), np.nonzero(flags != unk)[0][:10]
    exchanged = sum(int(open(tmp_path / f"bloom_exchanged_{r}.txt").read()) for r in range(world))
    if bits == 1 << 16:
        assert 0 < exchanged < n_pass // 2                     # most keys never leave their rank
    else:
        assert exchanged > n_pass // 2                         # 8-word filter, saturated: nearly everything is checked exactly
