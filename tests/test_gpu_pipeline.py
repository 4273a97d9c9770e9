"""-m gpu: asynchronous host ingestion (ctmr_submit_batch / ctmr_wait / ctmr_flush, csrc/engine/pipeline.inc) — the
GPU-side counterpart of the reference's entryChan (cmd/ct-fetch/ct-fetch.go:132,191): get-entries-sized batches
(≤ 1 001 entries, :417-424) submitted without waiting coalesce into super-batches and come back, ticket by ticket,
exactly as a sequence of synchronous calls over the same stream would answer — including which duplicate is "first"
when the two copies of a key sit in different batches that are in flight together."""
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402,F401

import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth, _native as N
from tests.gpu_common import run_oracle, assert_records_equal

NOW = synth.BASE_TIME
FILT = b"Synth Issuer 0,Synth Issuer 1"


def make(issuers, **kw):
    kw.setdefault("table_slots", 1 << 20)
    kw.setdefault("pair_slots", 1 << 14)
    e = ctmr.Engine(device=0, **kw)
    e.add_issuers(issuers)
    e.set_filter(FILT, False, NOW)
    return e


def arrays(b):
    pay = np.concatenate([b.payload, np.zeros(32, np.uint8)])
    return pay, b.offsets.astype(np.uint64), b.issuer_idx.astype(np.uint32), b.entry_type.astype(np.uint8)


@pytest.mark.parametrize("size,count", [(1001, 150), (257, 40), (70000, 3)])
def test_tickets_answer_like_synchronous_calls(size, count):
    cfg = synth.config(seed=91, n_issuers=64, zipf=1, dup_permille=200, ca_permille=10, expired_permille=10)
    issuers = synth.issuers(cfg)
    eng = make(issuers)
    batches = [synth.host_batch(cfg, k * size, size) for k in range(count)]
    keep, tickets = [], []
    for b in batches:                                  # submit everything, collect nothing yet …
        a = arrays(b)
        keep.append(a)
        tickets.append(eng.submit_batch(a[0], a[1], a[2], a[3], b.n))
        if len(tickets) - sum(1 for _ in ()) > 0 and len(tickets) % 120 == 0:
            eng.flush()
    o = None
    for b, t in zip(batches, tickets):                 # … then collect in order: the oracle sees ONE stream
        res = eng.wait(t, b.n)
        o, st, unk, eh = run_oracle(b, issuers, FILT, False, NOW, engine=o)
        assert_records_equal(res, b, st, unk, eh)
        assert res.stats.n_dup == int(((st == 0) & (unk == 0)).sum())
        assert res.stats.payload_bytes == int(b.offsets[-1])
    assert eng.total_count() == o.total_count()
    with pytest.raises(ctmr.CtmrError) as ei:          # a ticket is collected once
        eng.wait(tickets[0], batches[0].n)
    assert ei.value.code == N.E_NOTFOUND
    # the synchronous entry point still works on the same engine, and sees what the pipeline inserted
    again = eng.map_batch(batches[0])
    assert again.stats.n_new == 0
    eng.close()


def test_out_of_order_collection_empty_batches_and_pinned_memory():
    cfg = synth.config(seed=92, n_issuers=8, dup_permille=300)
    issuers = synth.issuers(cfg)
    eng = make(issuers)
    sizes = [1001, 0, 5, 1001, 0, 300]
    batches, first = [], 0
    for n in sizes:
        batches.append(synth.host_batch(cfg, first, n))
        first += n
    keep, tickets = [], []
    for k, b in enumerate(batches):
        pay, off, iss, et = arrays(b)
        if k % 2 == 0:                                 # every other payload in page-locked memory: straight DMA
            p = eng.pinned_array(pay.nbytes)
            p[:] = pay
            pay = p
        keep.append((pay, off, iss, et))
        tickets.append(eng.submit_batch(pay, off, iss, None if k == 2 else et, b.n))
    # the oracle processes the stream in SUBMISSION order whatever the collection order is
    want, o = [], None
    for k, b in enumerate(batches):
        if k == 2:
            b.entry_type[:] = 0                        # submitted with entry_type = NULL: all X509
        o, st, unk, eh = run_oracle(b, issuers, FILT, False, NOW, engine=o)
        want.append((st, unk, eh))
    for k in (5, 0, 3, 1, 2, 4):
        res = eng.wait(tickets[k], batches[k].n)
        if batches[k].n:
            assert_records_equal(res, batches[k], *want[k])
        else:
            assert res.stats.n == 0 and len(res.new_idx) == 0
    eng.close()


def test_pipeline_full_is_reported_not_deadlocked():
    cfg = synth.config(seed=93, n_issuers=4)
    issuers = synth.issuers(cfg)
    eng = make(issuers)
    b = synth.host_batch(cfg, 0, 70000)                # one submit = one whole super-batch
    a = arrays(b)
    tickets = [eng.submit_batch(a[0], a[1], a[2], a[3], b.n) for _ in range(4)]
    with pytest.raises(ctmr.CtmrError) as ei:          # four uncollected super-batches: the fifth is refused, not blocked
        for _ in range(3):
            eng.submit_batch(a[0], a[1], a[2], a[3], b.n)
    assert ei.value.code == N.E_RANGE
    r0 = eng.wait(tickets[0], b.n)
    assert r0.stats.n_new > 60000
    t5 = eng.submit_batch(a[0], a[1], a[2], a[3], b.n)  # a slot is free again
    for t in tickets[1:] + [t5]:
        assert eng.wait(t, b.n).stats.n_new == 0        # the same certificates again: all known
    eng.close()


def test_several_threads_submit_and_wait():
    cfg = synth.config(seed=94, n_issuers=16, dup_permille=0)
    issuers = synth.issuers(cfg)
    eng = make(issuers)
    per, rounds, T = 1001, 30, 4
    news, errors = [0] * T, []

    def worker(t):
        try:
            for r in range(rounds):
                b = synth.host_batch(cfg, (t * rounds + r) * per, per)
                a = arrays(b)
                tk = eng.submit_batch(a[0], a[1], a[2], a[3], b.n)
                news[t] += eng.wait(tk, b.n).stats.n_new
        except Exception as ex:                          # noqa: BLE001
            errors.append(ex)
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    whole = synth.host_batch(cfg, 0, per * rounds * T)
    o, st, unk, eh = run_oracle(whole, issuers, FILT, False, NOW)
    assert sum(news) == int(unk.sum()) == eng.total_count()   # disjoint keys: the interleaving does not matter
    eng.close()


def test_submitters_blocked_on_a_full_pipeline_never_orphan_a_super_batch():
    """Back-pressure with several submitters (round-2 advisor finding): packed and raw submits alternate, so nearly every
    submit closes the open super-batch and needs a slot of its own; four threads keep three tickets outstanding each
    against four slots, so submitters sleep in pipe_open_for with no FREE slot and wake one after the other.  A
    submitter that woke up and took a second slot without looking at p->open again left the first one in PS_OPEN for
    ever — its tickets never completed.  Every ticket must come back and the union must be the oracle's."""
    cfg = synth.config(seed=96, n_issuers=16, dup_permille=0)
    issuers = synth.issuers(cfg)
    eng = make(issuers)
    per, rounds, T, keep_out = 301, 24, 4, 3
    news, errors = [0] * T, []

    def collect(t, item):
        raw, tk, n, _keep = item
        res = eng.wait_entries(tk, n) if raw else eng.wait(tk, n)
        news[t] += res.stats.n_new

    def worker(t):
        raw = t % 2 == 1
        out = []
        try:
            for r in range(rounds):
                first = (t * rounds + r) * per
                while True:
                    try:
                        if raw:
                            e = synth.host_entries(cfg, first, per)
                            item = (True, eng.submit_entries(e.blob, e.bounds, e.n), e.n, e)
                        else:
                            b = synth.host_batch(cfg, first, per)
                            a = arrays(b)
                            item = (False, eng.submit_batch(a[0], a[1], a[2], a[3], b.n), b.n, a)
                        break
                    except ctmr.CtmrError as ex:           # every slot holds uncollected results: collect, retry
                        if ex.code != N.E_RANGE:
                            raise
                        if out:
                            collect(t, out.pop(0))
                        else:                              # … all of them other threads' results: they will collect
                            time.sleep(0.002)
                out.append(item)
                while len(out) > keep_out:
                    collect(t, out.pop(0))
            while out:
                collect(t, out.pop(0))
        except Exception as ex:                          # noqa: BLE001
            errors.append(ex)
    ths = [threading.Thread(target=worker, args=(t,), daemon=True) for t in range(T)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(120)
    assert not any(th.is_alive() for th in ths), "a ticket never completed: a super-batch was orphaned"
    assert not errors, errors
    whole = synth.host_batch(cfg, 0, per * rounds * T)
    o, st, unk, eh = run_oracle(whole, issuers, FILT, False, NOW)
    assert sum(news) == int(unk.sum()) == eng.total_count()
    eng.close()


def test_table_growth_happens_under_the_pipeline():
    """The known-certificate table starts at 4 096 slots; the super-batches the worker runs make it grow (k_rehash)
    between them — results stay those of the oracle over the whole stream."""
    cfg = synth.config(seed=95, n_issuers=32, zipf=1, dup_permille=250, ca_permille=10, expired_permille=10)
    issuers = synth.issuers(cfg)
    eng = make(issuers, table_slots=1 << 12)
    per, count = 1001, 120
    batches = [synth.host_batch(cfg, k * per, per) for k in range(count)]
    keep, pending = [], []
    o = None

    def collect(k, t):
        nonlocal o
        res = eng.wait(t, batches[k].n)
        o, st, unk, eh = run_oracle(batches[k], issuers, FILT, False, NOW, engine=o)
        assert (res.records["status"] == st).all(), k
        assert (((res.records["flags"] & 2) != 0) == (unk != 0)).all(), k

    for k, b in enumerate(batches):
        a = arrays(b)
        keep.append(a)
        pending.append((k, eng.submit_batch(a[0], a[1], a[2], a[3], b.n)))
        if k % 17 == 16:
            eng.flush()                                # several small super-batches: several rebuilds
        while len(pending) > 30:                       # at most two flushed super-batches uncollected
            collect(*pending.pop(0))
    for k, t in pending:
        collect(k, t)
    assert eng.total_count() == o.total_count() > 60000
    assert sorted(eng.keys(b"serials::*")) == [k for k in o.keys() if k.startswith(b"serials::")]
    eng.close()


# ---------------------------------------------------------------------------------------------------------------
# raw get-entries responses through the same pipeline (ctmr_submit_entries / ctmr_wait_entries)
from oracle import oracle as orc  # noqa: E402


def _raw_arrays(raw):
    blob = np.concatenate([np.ascontiguousarray(raw.blob, dtype=np.uint8), np.zeros(32, np.uint8)])
    return blob, np.ascontiguousarray(raw.bounds, dtype=np.uint64)


@pytest.mark.parametrize("size,count", [(1001, 90), (333, 25), (70000, 2)])
def test_raw_tickets_answer_like_synchronous_raw_calls(size, count):
    """Responses of `size` entries submitted without waiting; each ticket must answer like the oracle's
    LogEntryFromLeaf + insertCTWorker over the ONE stream they form — records, NEW list, timestamps, decode statistics;
    the engine registers the issuers on the way (no add_issuers here)."""
    cfg = synth.config(seed=93, n_issuers=48, zipf=1, dup_permille=200, ca_permille=10, expired_permille=10)
    eng = ctmr.Engine(device=0, table_slots=1 << 20, pair_slots=1 << 14)
    eng.set_filter(FILT, False, NOW)
    raws = [synth.host_entries(cfg, k * size, size) for k in range(count)]
    keep, tickets = [], []
    for raw in raws:
        a = _raw_arrays(raw)
        keep.append(a)
        tickets.append(eng.submit_entries(a[0], a[1], raw.n))
    o = orc.Engine(FILT, False, NOW)
    for raw, t in zip(raws, tickets):
        res = eng.wait_entries(t, raw.n)
        st, unk, eh, ts = o.raw_batch(raw.blob, raw.bounds)
        assert (res.records["status"] == st).all()
        assert (((res.records["flags"] & 2) != 0) == (unk != 0)).all()
        assert (res.timestamp == ts).all()
        assert (res.new_idx == np.nonzero(unk)[0]).all()
        assert res.decode.n == raw.n and res.decode.n_x509 + res.decode.n_precert == raw.n
        assert res.decode.blob_bytes == int(raw.bounds[-1]) == res.stats.payload_bytes
    assert eng.total_count() == o.total_count()
    assert 0 < eng.issuer_count() <= 48
    eng.close()


def test_raw_and_packed_submits_share_the_pipeline_in_order():
    """Alternating forms: every change of form closes the open super-batch, the order of the stream is kept — a key's
    first copy in a packed batch makes its later copy in a raw batch a known duplicate, and the other way round."""
    cfg = synth.config(seed=94, n_issuers=16, zipf=1, dup_permille=300)
    issuers = synth.issuers(cfg)
    eng = make(issuers)
    o = orc.Engine(FILT, False, NOW)
    size, pending, keep = 700, [], []

    def collect(form, x, t):
        if form == "packed":
            res = eng.wait(t, x.n)
            _, st, unk, _ = run_oracle(x, issuers, FILT, False, NOW, engine=o)
        else:
            res = eng.wait_entries(t, x.n)
            st, unk, _, _ = o.raw_batch(x.blob, x.bounds)
        assert (res.records["status"] == st).all(), form
        assert (((res.records["flags"] & 2) != 0) == (unk != 0)).all(), form

    for k in range(12):
        if k % 2 == 0:
            b = synth.host_batch(cfg, k * size, size)
            a = arrays(b)
            keep.append(a)
            pending.append(("packed", b, eng.submit_batch(a[0], a[1], a[2], a[3], b.n)))
        else:
            raw = synth.host_entries(cfg, k * size, size)
            a = _raw_arrays(raw)
            keep.append(a)
            pending.append(("raw", raw, eng.submit_entries(a[0], a[1], raw.n)))
        if len(pending) == 3:          # at most four super-batches may be unfinished or uncollected
            collect(*pending.pop(0))
    while pending:
        collect(*pending.pop(0))
    assert eng.total_count() == o.total_count()
    eng.close()


def test_raw_submit_with_autoregistration_off_reports_the_pending_issuers():
    cfg = synth.config(seed=95, n_issuers=5)
    raw = synth.host_entries(cfg, 0, 2000)
    eng = ctmr.Engine(device=0, table_slots=1 << 14, pair_slots=1 << 12)
    eng.set_filter(b"", True, NOW)
    eng.set_issuer_autoregister(False)
    a = _raw_arrays(raw)
    t = eng.submit_entries(a[0], a[1], raw.n)
    with pytest.raises(ctmr.CtmrError) as ei:
        eng.wait_entries(t, raw.n)
    assert ei.value.code == N.E_NOTFOUND
    pend = eng.pending_issuers()
    assert sorted(pend) == sorted(set(synth.issuers(cfg)[int(i)] for i in synth.host_batch(cfg, 0, 2000).issuer_idx))
    eng.add_issuers(sorted(pend))
    t = eng.submit_entries(a[0], a[1], raw.n)
    res = eng.wait_entries(t, raw.n)
    o = orc.Engine(b"", True, NOW)
    st, unk, _, _ = o.raw_batch(raw.blob, raw.bounds)
    assert (res.records["status"] == st).all() and (((res.records["flags"] & 2) != 0) == (unk != 0)).all()
    eng.close()
