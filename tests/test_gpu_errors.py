"""-m gpu: the C ABI fails loudly (code + message, nothing half-done) on bad input and on exhausted capacity."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402,F401

import ct_mapreduce_amd as ctmr
from oracle import oracle as orc
from ct_mapreduce_amd import synth, _native as N
from ct_mapreduce_amd.engine import Batch, RawEntries

NOW = synth.BASE_TIME


def code(fn):
    with pytest.raises(ctmr.CtmrError) as ei:
        fn()
    return ei.value.code


def test_table_full_fails_before_anything_is_applied():
    """max_table_slots caps the growth; a batch that cannot fit is refused BEFORE any insert (a partially
    applied batch lost its new certificates' write-back): the sets, the counters and a retry with a smaller batch are
    what they would be had the failed call never happened."""
    cfg = synth.config(seed=1, n_issuers=2)
    issuers = synth.issuers(cfg)
    eng = ctmr.Engine(device=0, table_slots=1 << 10, pair_slots=1 << 10, max_table_slots=1 << 10)
    eng.add_issuers(issuers)
    eng.set_filter(b"", True, NOW)
    small = synth.host_batch(cfg, 0, 300)
    r1 = eng.map_batch(small)
    keys_before, total_before = sorted(eng.keys(b"serials::*")), eng.total_count()
    assert code(lambda: eng.map_batch(synth.host_batch(cfg, 300, 3000))) == N.E_FULL     # 3 000 keys, 1 024 slots
    assert sorted(eng.keys(b"serials::*")) == keys_before and eng.total_count() == total_before == r1.stats.n_new
    nxt = synth.host_batch(cfg, 300, 300)
    r2 = eng.map_batch(nxt)                                                            # … and the engine still works
    o = orc.Engine(b"", True, NOW)
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    blob = np.frombuffer(b"".join(issuers), np.uint8)
    o.batch(small.payload, small.offsets, small.issuer_idx, blob, io, entry_type=small.entry_type)
    st, unk, _ = o.batch(nxt.payload, nxt.offsets, nxt.issuer_idx, blob, io, entry_type=nxt.entry_type)
    assert (((r2.records["flags"] & 2) != 0) == (unk != 0)).all() and eng.total_count() == o.total_count()
    key = b"serials::2030-01-01-00::" + eng.issuer_id(0).encode()

    def point_inserts():                               # SetInsert meets the same limit, one member at a time
        for k in range(2000):
            eng.set_insert(key, bytes([k >> 8, k & 255, 7]))
    assert code(point_inserts) == N.E_FULL
    assert 100 < eng.set_cardinality(key) < 500 and eng.total_count() == o.total_count() + eng.set_cardinality(key)
    eng.close()


def test_table_grows_and_recovers_tombstones():
    """Redis grows until OOM (storage/rediscache.go:57-65): the table is rebuilt on the GPU (k_rehash) when a batch
    could push the load past 3/4 — larger when members were added, the same size when expiry sweeps left tombstones."""
    cfg = synth.config(seed=3, n_issuers=8, dup_permille=200)
    issuers = synth.issuers(cfg)
    eng = ctmr.Engine(device=0, table_slots=1 << 10, pair_slots=1 << 14)
    eng.add_issuers(issuers)
    eng.set_filter(b"", True, NOW)
    o = orc.Engine(b"", True, NOW)
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    blob = np.frombuffer(b"".join(issuers), np.uint8)
    first = 0
    for n in (500, 700, 3000, 20000, 1000):          # 1 024 slots → 65 536: several rebuilds, duplicates across them
        b = synth.host_batch(cfg, first, n)
        res = eng.map_batch(b)
        st, unk, _ = o.batch(b.payload, b.offsets, b.issuer_idx, blob, io, entry_type=b.entry_type)
        assert (((res.records["flags"] & 2) != 0) == (unk != 0)).all(), n
        first += n - 100                              # overlap: the old members must still be found after a rebuild
    assert eng.total_count() == o.total_count()
    assert sorted(eng.keys(b"serials::*")) == [k for k in o.keys() if k.startswith(b"serials::")]
    counts = eng.issuer_counts()
    for k in range(len(issuers)):
        assert int(counts[k]) == o.issuer_count(eng.issuer_id(k))
    key = sorted(eng.keys(b"serials::*"))[len(o.keys()) // 2]
    assert eng.set_list(key) == o.members(key)
    # tombstones: expire everything, then fill again and again under a cap that only holds one round
    eng2 = ctmr.Engine(device=0, table_slots=1 << 13, pair_slots=1 << 14, max_table_slots=1 << 13)
    eng2.add_issuers(issuers)
    eng2.set_filter(b"", True, NOW)
    for rnd in range(6):                              # 6 × 4 000 members through 8 192 slots
        b = synth.host_batch(cfg, 100000 + rnd * 5000, 5000)
        res = eng2.map_batch(b)
        assert res.stats.n_new == eng2.total_count() > 3500
        assert eng2.expire_sweep(NOW + 400 * 86400) == res.stats.n_new and eng2.total_count() == 0
    eng2.close()
    eng.close()


def test_the_arena_squeezes_out_the_cells_of_known_certificates():
    """A batch takes one key cell per ENTRY; the cells of entries that were not new are garbage.  A deployment that keeps
    meeting known certificates (a restart that re-reads a log) must not grow for it: before a round that does not fit the
    arena any more the live cells are compacted (k_arena_compact) — same answers, same sets, no growth."""
    cfg = synth.config(seed=31, n_issuers=6, dup_permille=100)
    issuers = synth.issuers(cfg)
    eng = ctmr.Engine(device=0, table_slots=1 << 13, pair_slots=1 << 14)      # arena: 4 096 cells to begin with
    eng.add_issuers(issuers)
    eng.set_filter(b"", True, NOW)
    o = orc.Engine(b"", True, NOW)
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    blob = np.frombuffer(b"".join(issuers), np.uint8)
    assert eng.table_info().arena_cells == 4096
    b = synth.host_batch(cfg, 0, 1500)
    for rnd in range(14):                             # the same 1 500 entries again and again: 21 000 cells without squeezing
        res = eng.map_batch(b)
        st, unk, _ = o.batch(b.payload, b.offsets, b.issuer_idx, blob, io, entry_type=b.entry_type)
        assert (((res.records["flags"] & 2) != 0) == (unk != 0)).all(), rnd
        assert (res.records["status"] == st).all()
    ti = eng.table_info()
    assert ti.arena_cells == 4096 and ti.arena_compactions >= 3 and ti.arena_growths == 0 and ti.rebuilds == 0
    assert ti.arena_used <= ti.occupied + 1500 and ti.occupied == eng.total_count() == o.total_count()
    # new members after compactions: cells are handed out behind the squeezed ones; removal and re-insert still work
    b2 = synth.host_batch(cfg, 5000, 1200)
    res = eng.map_batch(b2)
    st, unk, _ = o.batch(b2.payload, b2.offsets, b2.issuer_idx, blob, io, entry_type=b2.entry_type)
    assert (((res.records["flags"] & 2) != 0) == (unk != 0)).all()
    assert eng.total_count() == o.total_count()
    assert sorted(eng.keys(b"serials::*")) == [k for k in o.keys() if k.startswith(b"serials::")]
    key = sorted(eng.keys(b"serials::*"))[3]
    assert eng.set_list(key) == o.members(key)
    counts = eng.issuer_counts()
    for k in range(len(issuers)):
        assert int(counts[k]) == o.issuer_count(eng.issuer_id(k))
    # a batch larger than what squeezing frees: the arena grows as before
    b3 = synth.host_batch(cfg, 20000, 9000)
    res = eng.map_batch(b3)
    st, unk, _ = o.batch(b3.payload, b3.offsets, b3.issuer_idx, blob, io, entry_type=b3.entry_type)
    assert (((res.records["flags"] & 2) != 0) == (unk != 0)).all()
    ti = eng.table_info()
    assert ti.arena_growths >= 1 and ti.rebuilds >= 1 and eng.total_count() == o.total_count()
    eng.close()


def test_members_inserted_before_their_issuer_is_registered_are_migrated():
    """a Redis restore (storage.redis_load) or a RemoteCache.SetInsert ahead of add_issuers put
    `serials::` members of a not-yet-registered issuer ID into the host-side store, where the map's dedup never
    looked.  Registration now moves them into the table."""
    cfg = synth.config(seed=9, n_issuers=3, dup_permille=0)
    issuers = synth.issuers(cfg)
    b = synth.host_batch(cfg, 0, 400)
    ref = ctmr.Engine(device=0, table_slots=1 << 12, pair_slots=1 << 12)
    ref.add_issuers(issuers)
    ref.set_filter(b"", True, NOW)
    ref.map_batch(b)
    dump = {k: ref.set_list(k) for k in ref.keys(b"serials::*")}
    eng = ctmr.Engine(device=0, table_slots=1 << 12, pair_slots=1 << 12)
    eng.set_filter(b"", True, NOW)
    long_member = bytes(range(1, 60))                 # > CTMR_MAX_SERIAL: stays host-side, but starts to count
    some_key = sorted(dump)[0]
    for k, members in dump.items():                   # restore BEFORE any issuer is known
        for m in members:
            assert eng.set_insert(k, m)
    assert eng.set_insert(some_key, long_member)
    assert sorted(eng.keys(b"serials::*")) == sorted(dump) and eng.total_count() == 0      # no issuer yet: nothing to count under
    eng.add_issuers(issuers)
    assert eng.total_count() == ref.total_count() + 1
    assert (eng.issuer_counts()[:3].sum() == ref.issuer_counts()[:3].sum() + 1)
    for k, members in dump.items():
        assert eng.set_list(k) == sorted(members + ([long_member] if k == some_key else []))
        assert eng.set_cardinality(k) == len(members) + (k == some_key)
    res = eng.map_batch(b)                            # the restored certificates are KNOWN to the map now
    assert res.stats.n_new == 0 and res.stats.n_dup == int(res.stats.by_status[0])
    assert not eng.set_insert(some_key, dump[some_key][0])
    ref.close()
    eng.close()


def test_issuer_table_full_and_bad_arguments():
    cfg = synth.config(seed=1, n_issuers=4)
    eng = ctmr.Engine(device=0, table_slots=1 << 12, pair_slots=1 << 10, max_issuers=3)
    assert code(lambda: eng.add_issuers(synth.issuers(cfg))) == N.E_FULL
    eng.add_issuers(synth.issuers(cfg)[:2])
    b = synth.host_batch(cfg, 0, 10)
    bad = Batch(b.payload, b.offsets.copy(), b.issuer_idx, b.entry_type)
    bad.offsets[3], bad.offsets[4] = bad.offsets[4], bad.offsets[3]
    assert code(lambda: eng.map_batch(bad)) == N.E_INVAL                                  # offsets not monotone
    assert code(lambda: eng.set_filter(b",".join(b"x" for _ in range(70)), False, 0)) == N.E_INVAL   # > 64 pieces
    assert code(lambda: eng.issuer_info(99)) == N.E_NOTFOUND
    # device entry point: payload must be 16-byte aligned
    dev = torch.device("cuda:0")
    d_pay = torch.zeros(b.payload.size + 64, dtype=torch.uint8, device=dev)
    d_off = torch.from_numpy(b.offsets.astype(np.int64)).to(dev)
    d_iss = torch.zeros(10, dtype=torch.int32, device=dev)
    assert code(lambda: eng.map_batch_device(d_pay.data_ptr() + 1, d_off.data_ptr(), d_iss.data_ptr(), 0, 10)) == N.E_INVAL
    # raw entries: bounds must start at 0 (device form) and be monotone (host form)
    raw = synth.host_entries(cfg, 0, 5)
    rb = RawEntries(raw.blob, raw.bounds.copy())
    rb.bounds[2], rb.bounds[3] = rb.bounds[3], rb.bounds[2]
    assert code(lambda: eng.map_entries(rb)) == N.E_INVAL
    d_blob = torch.zeros(raw.blob.size + 64, dtype=torch.uint8, device=dev)
    d_b = torch.from_numpy((raw.bounds + np.uint64(16)).astype(np.int64)).to(dev)
    assert code(lambda: eng.map_entries_device(d_blob.data_ptr(), d_b.data_ptr(), 5)) == N.E_INVAL
    # the engine is still usable after every failure
    eng.set_filter(b"", True, NOW)
    assert eng.map_batch(b).stats.n == 10
    eng.close()


def test_pinned_buffers_round_trip():
    cfg = synth.config(seed=2, n_issuers=2)
    eng = ctmr.Engine(device=0, table_slots=1 << 14, pair_slots=1 << 10)
    eng.add_issuers(synth.issuers(cfg))
    eng.set_filter(b"", True, NOW)
    b = synth.host_batch(cfg, 0, 2000)
    want = eng.map_batch(b).records.copy()
    eng.reset_known()
    p = eng.pinned_array(b.payload.nbytes)
    p[:] = b.payload
    got = eng.map_batch(Batch(p, b.offsets, b.issuer_idx, b.entry_type)).records
    assert (got == want).all()
    eng.close()


def test_concurrent_callers_share_one_engine():
    """RemoteCache methods are called from numThreads goroutines at once (ct-fetch.go:140-145): every entry point
    takes the engine mutex.  Four host threads map disjoint windows of one stream and hammer the point operations;
    the final state equals the single-threaded oracle's."""
    import threading
    from oracle import oracle as orc
    cfg = synth.config(seed=3, n_issuers=4, dup_permille=0)
    issuers = synth.issuers(cfg)
    eng = ctmr.Engine(device=0, table_slots=1 << 16, pair_slots=1 << 15)   # 90 d × 24 h × 4 issuers of (expDate, issuer) pairs
    eng.add_issuers(issuers)
    eng.set_filter(b"", True, NOW)
    batches = [synth.host_batch(cfg, k * 3000, 3000) for k in range(8)]
    errors, news = [], [0] * 4

    def worker(t):
        try:
            for k in (t, t + 4):
                news[t] += eng.map_batch(batches[k]).stats.n_new
                for j in range(50):
                    m = bytes([1 + t, j, k])
                    assert eng.set_insert(b"crl::thread%d" % t, m) and eng.set_contains(b"crl::thread%d" % t, m)
                eng.issuer_counts()
                eng.keys(b"crl::*")
        except Exception as ex:          # noqa: BLE001
            errors.append(ex)
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    o = orc.Engine(b"", True, NOW)
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    blob = np.frombuffer(b"".join(issuers), np.uint8)
    want = 0
    for b in batches:
        st, unk, eh = o.batch(b.payload, b.offsets, b.issuer_idx, blob, io)
        want += int(unk.sum())
    assert sum(news) == want == eng.total_count() == o.total_count()
    assert sorted(eng.keys(b"serials::*")) == [k for k in o.keys() if k.startswith(b"serials::")]
    for t in range(4):
        assert eng.set_cardinality(b"crl::thread%d" % t) == 100
    eng.close()
