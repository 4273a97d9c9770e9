"""-m gpu: BASELINE.json's configs[1] and configs[2] at their full sizes.

configs[1]  1 M synthetic ~1.5 KB certificates, single issuer, known-certificate dedup — bit-exact against the oracle on
            every entry (the batch is generated in HBM and copied back for the CPU run).
configs[2]  10 M certificates, 256 issuers, issuerCN prefix filter + per-issuer unique counts — too large for the
            oracle to run whole inside a test, so checked (i) through size-independent properties of the generator and
            of set insertion: new ⇔ PASS ∧ first carrier of its key, per-issuer counts = histogram of the new entries,
            idempotent replay, the NEW list; and (ii) bit-exact against the oracle on a strided, duplicate-closed
            sample of 1.1 M entries (bench.py's parity leg).
(configs[3]/[4] — 100 M-entry batch, 1 B-entry stream — run in bench.py / bench.py --stream with the same checks.)"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth, _native as N
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_is_dup, oracle_sample_check  # noqa: E402  (the generator's duplicate structure; bench.py's oracle sample)

NOW = synth.BASE_TIME


def device_batch(eng, cfg, first, n, dev):
    d_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    total = eng.synth_device(cfg, first, n, d_off.data_ptr(), 0, 0, 0, 0)
    d_pay = torch.empty(total + N.PAYLOAD_PAD + 16, dtype=torch.uint8, device=dev)
    d_iss = torch.empty(n, dtype=torch.int32, device=dev)
    d_et = torch.empty(n, dtype=torch.uint8, device=dev)
    eng.synth_device(cfg, first, n, d_off.data_ptr(), d_pay.data_ptr(), d_pay.numel(), d_iss.data_ptr(), d_et.data_ptr())
    return d_off, d_pay, d_iss, d_et, total


def test_config1_one_million_single_issuer_bit_exact():
    n, dev = 1_000_000, torch.device("cuda:0")
    cfg = synth.config(seed=20260921 + 1, n_issuers=1, dup_permille=50)
    issuers = synth.issuers(cfg)
    eng = ctmr.Engine(device=0, table_slots=1 << 21, pair_slots=1 << 16)
    eng.add_issuers(issuers)
    eng.set_filter(b"", False, NOW)
    d_off, d_pay, d_iss, d_et, total = device_batch(eng, cfg, 0, n, dev)
    d_rec = torch.empty(n * 32, dtype=torch.uint8, device=dev)
    d_new = torch.empty(n, dtype=torch.int64, device=dev)
    st = eng.map_batch_device(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), n,
                              d_rec.data_ptr(), d_new.data_ptr())
    # the reference loop on the same bytes
    o = orc.Engine(b"", False, NOW)
    io = np.array([0, len(issuers[0])], np.uint64)
    ost, ounk, oeh = o.batch(d_pay.cpu().numpy(), d_off.cpu().numpy().astype(np.uint64),
                             d_iss.cpu().numpy().astype(np.uint32), np.frombuffer(issuers[0], np.uint8), io)
    rec = d_rec.cpu().numpy().view(ctmr.engine.RECORD_DTYPE)
    assert (rec["status"] == ost).all()
    assert (((rec["flags"] & 2) != 0) == (ounk != 0)).all()
    assert (rec["exp_hour"][ost != orc.ST_PARSE_ERROR] == oeh[ost != orc.ST_PARSE_ERROR]).all()
    assert (d_new[:st.n_new].cpu().numpy() == np.nonzero(ounk)[0]).all()
    assert st.n_new == int(ounk.sum()) == o.total_count() == eng.total_count()
    assert int(eng.issuer_counts()[0]) == o.issuer_count(eng.issuer_id(0))
    assert sorted(eng.keys(b"serials::*")) == [k for k in o.keys() if k.startswith(b"serials::")]
    assert 0 < st.n_dup < st.n_new
    eng.close()


def test_config2_ten_million_256_issuers_properties():
    n, dev = 10_000_000, torch.device("cuda:0")
    cfg = synth.config(seed=20260921 + 2, n_issuers=256, zipf=1, dup_permille=100, ca_permille=10, expired_permille=10)
    eng = ctmr.Engine(device=0, table_slots=1 << 25, pair_slots=1 << 22)
    eng.add_issuers(synth.issuers(cfg))
    eng.set_filter(b"Synth Issuer 0,Synth Issuer 1", False, NOW)        # passes issuers 000-199
    d_off, d_pay, d_iss, d_et, total = device_batch(eng, cfg, 0, n, dev)
    d_rec = torch.empty(n * 32, dtype=torch.uint8, device=dev)
    d_new = torch.empty(n, dtype=torch.int64, device=dev)
    st = eng.map_batch_device(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), n,
                              d_rec.data_ptr(), d_new.data_ptr())
    rec = d_rec.view(-1, 32)
    status = rec[:, 0].cpu().numpy()
    flags = rec[:, 1].cpu().numpy()
    iss = d_iss.cpu().numpy()
    # filter semantics: CN filter ⇔ issuer index ≥ 200 (unless an earlier filter fired), nothing fails to parse
    assert (status != N.ST_PARSE_ERROR).all() and st.by_status[N.ST_PARSE_ERROR] == 0
    cn = status == N.ST_FILTERED_CN
    assert (iss[cn] >= 200).all() and (iss[status == N.ST_PASS] < 200).all()
    # dedup: an entry is new ⇔ it reached Store and is the first carrier of its key (the generator re-emits an
    # EARLIER entry's key exactly for the entries synth_is_dup marks)
    dup = synth_is_dup(cfg.seed, 0, n, 100, np)
    new = (flags & N.FL_WAS_UNKNOWN) != 0
    assert (new == ((status == N.ST_PASS) & ~dup)).all()
    assert st.n_new == int(new.sum()) and st.n_dup == int(((status == N.ST_PASS) & dup).sum())
    new_idx = d_new[:st.n_new].cpu().numpy()
    assert (new_idx == np.nonzero(new)[0]).all()                         # ascending, complete: checksum-free equality
    # per-issuer unique counts (storage-statistics.go:44-53) = histogram of the new entries' issuers
    counts = eng.issuer_counts().astype(np.int64)
    assert (counts == np.bincount(iss[new], minlength=256)).all() and counts[200:].sum() == 0
    assert eng.total_count() == st.n_new
    # Σ SCARD over the keys of one issuer = its count (the statistics tool's own arithmetic)
    k0 = [k for k in eng.keys(b"serials::*::" + eng.issuer_id(7).encode())]
    assert sum(eng.set_cardinality(k) for k in k0) == int(counts[7])
    # … and against the ORACLE (round 3): 40 equally spaced slices of 25 000 entries over the whole batch + every entry
    # outside them whose key a sampled duplicate repeats, in log order — the closed set on which the oracle's answer is
    # the whole batch's answer (tests/test_bench_helpers_cpu.py); ≈ 1.1 M entries, status and WasUnknown bit-exact
    filt = b"Synth Issuer 0,Synth Issuer 1"
    _, pinfo, _, _ = oracle_sample_check(np, torch, ctmr, synth, N, cfg, 100, synth.issuers(cfg), filt, NOW, dev, 0, n,
                                         d_off, d_pay, d_iss, d_et, d_rec, 40, 25_000)
    assert pinfo["status_mismatches"] == 0 and pinfo["was_unknown_mismatches"] == 0
    assert pinfo["entries"] > 1_000_000 and pinfo["known_duplicates"] > 50_000 and pinfo["sources_outside_the_slices"] > 50_000
    # idempotence: replaying the batch finds every stored entry known
    st2 = eng.map_batch_device(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), n,
                               d_rec.data_ptr(), d_new.data_ptr())
    assert st2.n_new == 0 and st2.n_dup == int((status == N.ST_PASS).sum())
    assert (eng.issuer_counts().astype(np.int64) == counts).all()
    eng.close()


def test_aligned_entry_view_is_the_same_certificates_at_aligned_offsets():
    """ctmr_synth_view_device (bench.py --aligned, secondary.aligned128): the synthetic certificates laid at multiples of
    128 bytes as an entry view — byte-identical certificates, identical records / NEW list / counts through
    ctmr_map_view_device."""
    n, dev = 200_000, torch.device("cuda:0")
    cfg = synth.config(seed=20260921 + 4, n_issuers=64, zipf=1, dup_permille=50, ca_permille=10, expired_permille=10)
    results = []
    for align in (0, 128):
        eng = ctmr.Engine(device=0, table_slots=1 << 19, pair_slots=1 << 16)
        eng.add_issuers(synth.issuers(cfg))
        eng.set_filter(b"Synth Issuer 0", False, NOW)
        d_rec = torch.empty(n * 32, dtype=torch.uint8, device=dev)
        d_new = torch.empty(n, dtype=torch.int64, device=dev)
        if align:
            d_st = torch.empty(n + 1, dtype=torch.int64, device=dev)
            d_en = torch.empty(n, dtype=torch.int64, device=dev)
            total = eng.synth_view_device(cfg, 0, n, align, d_st.data_ptr(), d_en.data_ptr(), 0, 0, 0, 0)
            d_pay = torch.zeros(total + N.PAYLOAD_PAD + 16, dtype=torch.uint8, device=dev)
            d_iss = torch.empty(n, dtype=torch.int32, device=dev)
            d_et = torch.empty(n, dtype=torch.uint8, device=dev)
            eng.synth_view_device(cfg, 0, n, align, d_st.data_ptr(), d_en.data_ptr(), d_pay.data_ptr(), d_pay.numel(),
                                  d_iss.data_ptr(), d_et.data_ptr())
            assert int((d_st[:n] % align).abs().sum().item()) == 0 and int(d_st[n].item()) == total
            view = N.EntryView(cert_start=d_st.data_ptr(), cert_end=d_en.data_ptr(), issuer_idx=d_iss.data_ptr(),
                               entry_type=d_et.data_ptr(), timestamp=None, chain0_start=None, chain0_len=None)
            st = eng.map_view_device(d_pay.data_ptr(), total, view, n, d_rec.data_ptr(), d_new.data_ptr())
            lens = (d_en - d_st[:n]).cpu().numpy()
            first = d_pay[int(d_st[5].item()):int(d_en[5].item())].cpu().numpy().tobytes()
        else:
            d_off, d_pay, d_iss, d_et, total = device_batch(eng, cfg, 0, n, dev)
            st = eng.map_batch_device(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), n,
                                      d_rec.data_ptr(), d_new.data_ptr())
            lens = np.diff(d_off.cpu().numpy())
            first = d_pay[int(d_off[5].item()):int(d_off[6].item())].cpu().numpy().tobytes()
        results.append((d_rec.cpu().numpy().copy(), d_new[:st.n_new].cpu().numpy().copy(), int(st.n_new),
                        [int(x) for x in st.by_status], eng.issuer_counts().copy(), lens, first, d_iss.cpu().numpy().copy()))
        eng.close()
    a, b = results
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and a[2] == b[2] > 0 and a[3] == b[3]
    assert (a[4] == b[4]).all() and (a[5] == b[5]).all() and a[6] == b[6] == synth.leaf(cfg, 5)[0] and (a[7] == b[7]).all()
