"""-m gpu: the C ABI fails loudly (code + message, nothing half-done) on bad input and on exhausted capacity."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402,F401

import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth, _native as N
from ct_mapreduce_amd.engine import Batch, RawEntries

NOW = synth.BASE_TIME


def code(fn):
    with pytest.raises(ctmr.CtmrError) as ei:
        fn()
    return ei.value.code


def test_known_certificate_table_full_is_reported():
    cfg = synth.config(seed=1, n_issuers=2)
    eng = ctmr.Engine(device=0, table_slots=1 << 10, pair_slots=1 << 10)
    eng.add_issuers(synth.issuers(cfg))
    eng.set_filter(b"", True, NOW)
    assert code(lambda: eng.map_batch(synth.host_batch(cfg, 0, 3000))) == N.E_FULL       # 3 000 keys, 1 024 slots
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
