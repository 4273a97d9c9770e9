"""-m gpu, N2: raw get-entries → k_decode_match (→ k_chain0_match for retry rounds) → the map/reduce, through the C
ABI, bit for bit against the oracle's LogEntryFromLeaf + insertCTWorker restatement (orc_engine_raw_batch)."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402,F401

import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth, _native as N
from ct_mapreduce_amd.engine import RawEntries
from oracle import oracle as orc
from tests.test_entry_decode_cpu import mutate_entry, x509_leaf, precert_leaf, chain, asn1cert

NOW = synth.BASE_TIME


def check_against_oracle(eng, raw, o, res):
    st, unk, eh, ts = o.raw_batch(raw.blob if len(raw.blob) else np.zeros(1, np.uint8), raw.bounds)
    r = res.records
    assert (r["status"] == st).all(), np.nonzero(r["status"] != st)[0][:10]
    assert (((r["flags"] & 2) != 0) == (unk != 0)).all()
    parsed = (st != orc.ST_PARSE_ERROR) & (st != orc.ST_ENTRY_DECODE_ERROR)
    assert (r["exp_hour"][parsed] == eh[parsed]).all()
    assert (res.timestamp == ts).all()
    assert (res.new_idx == np.nonzero(unk)[0]).all()
    for k in range(8):
        assert res.stats.by_status[k] == int((st == k).sum()), k
    assert res.decode.n_decode_error == int((st == orc.ST_ENTRY_DECODE_ERROR).sum())
    okeys = [k for k in o.keys() if k.startswith(b"serials::")]
    assert sorted(eng.keys(b"serials::*")) == okeys
    assert eng.total_count() == o.total_count()
    return st, unk


def test_synthetic_raw_entries_bit_exact_and_issuers_self_register():
    cfg = synth.config(seed=20260921 + 11, n_issuers=64, dup_permille=100, ca_permille=10, expired_permille=10)
    raw = synth.host_entries(cfg, 0, 20000)
    filt = b"Synth Issuer 00,Synth Issuer 01,Synth Issuer 02,Synth Issuer 03,Synth Issuer 04,Synth Issuer 05"
    eng = ctmr.Engine(device=0, table_slots=1 << 17, pair_slots=1 << 16)
    eng.set_filter(filt, False, NOW)
    assert eng.issuer_count() == 0                                  # the host registers nothing
    res = eng.map_entries(raw)
    o = orc.Engine(filt, False, NOW)
    st, unk = check_against_oracle(eng, raw, o, res)
    assert res.decode.n_x509 + res.decode.n_precert == 20000 and res.decode.n_precert > 5000
    assert 0 < res.decode.n_issuers_added <= 64 and eng.issuer_count() == res.decode.n_issuers_added
    assert (st == 0).sum() > 1000 and 0 < unk.sum() < (st == 0).sum()
    # per-issuer unique counts (storage-statistics.go:44-53), issuers identified by the ids the GPU computed
    counts = eng.issuer_counts()
    for k in range(eng.issuer_count()):
        assert int(counts[k]) == o.issuer_count(eng.issuer_id(k)), k
    # the same job from the packed form on a second engine: identical records
    b = synth.host_batch(cfg, 0, 20000)
    eng2 = ctmr.Engine(device=0, table_slots=1 << 17, pair_slots=1 << 16)
    eng2.add_issuers(synth.issuers(cfg))
    eng2.set_filter(filt, False, NOW)
    res2 = eng2.map_batch(b)
    for f in ("status", "flags", "serial_len", "exp_hour", "serial"):
        assert (res.records[f] == res2.records[f]).all(), f
    assert (res.new_idx == res2.new_idx).all()
    # PEM write-back of the new certificates straight out of the raw blob
    pems = eng.pem_new()
    assert len(pems) == len(res.new_idx)
    for pem, i in list(zip(pems, res.new_idx))[::37]:
        assert pem == orc.pem_encode(b.cert(int(i)))
    # a second window overlapping the first: no issuer is added again, duplicates are known
    raw2 = synth.host_entries(cfg, 15000, 10000)
    res3 = eng.map_entries(raw2)
    check_against_oracle(eng, raw2, o, res3)
    assert res3.decode.n_issuers_added <= 64 - res.decode.n_issuers_added
    eng.close()
    eng2.close()


def test_mutated_and_edge_entries():
    rng = random.Random(20260923)
    cfg = synth.config(seed=5, n_issuers=8, dup_permille=50)
    raw = synth.host_entries(cfg, 0, 3000)
    pairs = []
    for i in range(raw.n):
        leaf, extra = raw.leaf_input(i), raw.extra_data(i)
        if rng.random() < 0.4:
            leaf, extra = mutate_entry(rng, leaf, extra)
        pairs.append((leaf, extra))
    cert = synth.leaf(cfg, 7)[0]
    iss = synth.issuer(cfg, 3)
    pairs += [
        (x509_leaf(cert), chain([])),                                   # len(Chain) < 1
        (precert_leaf(b"\x30\x00"), asn1cert(cert) + chain([])),
        (x509_leaf(cert), chain([b"\x30\x03abc"])),                     # Chain[0] does not parse
        (x509_leaf(cert), chain([iss[:-1]])),                           # … a truncated issuer certificate
        (x509_leaf(cert, ext=b"\x00" * 5), chain([iss, iss, iss])),
        (x509_leaf(b"\x30\x00"), chain([iss])),                         # leaf certificate does not parse
        (b"", b""),
        (x509_leaf(cert), b""),
        (x509_leaf(cert, entry_type=3), chain([iss])),
    ]
    rng.shuffle(pairs)
    mixed = RawEntries.from_pairs(pairs)
    mixed.blob = np.concatenate([mixed.blob, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    eng = ctmr.Engine(device=0, table_slots=1 << 15, pair_slots=1 << 14)
    eng.set_filter(b"", True, NOW)
    res = eng.map_entries(mixed)
    o = orc.Engine(b"", True, NOW)
    st, _ = check_against_oracle(eng, mixed, o, res)
    for code in (orc.ST_ENTRY_DECODE_ERROR, orc.ST_NO_ISSUER, orc.ST_ISSUER_PARSE_ERROR, orc.ST_PARSE_ERROR, orc.ST_PASS):
        assert (st == code).sum() > 0, code
    # empty batch
    empty = eng.map_entries(RawEntries(np.zeros(0, np.uint8), np.zeros(1, np.uint64)))
    assert empty.stats.n == 0 and len(empty.new_idx) == 0
    eng.close()


def test_device_generator_and_device_path_match_the_host_path():
    cfg = synth.config(seed=20260921 + 12, n_issuers=256, dup_permille=100)
    n = 50000
    dev = torch.device("cuda:0")
    eng = ctmr.Engine(device=0, table_slots=1 << 18, pair_slots=1 << 16, profile=True)
    eng.set_filter(b"Synth Issuer 0,Synth Issuer 1", False, NOW)
    d_bounds = torch.empty(2 * n + 1, dtype=torch.int64, device=dev)
    total = eng.synth_entries_device(cfg, 0, n, d_bounds.data_ptr(), 0, 0)
    d_blob = torch.zeros(total + N.PAYLOAD_PAD + 16, dtype=torch.uint8, device=dev)
    assert eng.synth_entries_device(cfg, 0, n, d_bounds.data_ptr(), d_blob.data_ptr(), d_blob.numel()) == total
    raw = synth.host_entries(cfg, 0, n)
    assert (d_bounds.cpu().numpy().astype(np.uint64) == raw.bounds).all()
    assert (d_blob[:total].cpu().numpy() == raw.blob[:total]).all()          # host and device emit identical bytes
    d_rec = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
    d_new = torch.zeros(n, dtype=torch.int64, device=dev)
    d_ts = torch.zeros(n, dtype=torch.int64, device=dev)
    st, ds = eng.map_entries_device(d_blob.data_ptr(), d_bounds.data_ptr(), n, d_rec.data_ptr(), d_new.data_ptr(),
                                    d_ts.data_ptr())
    assert ds.n_issuers_added == eng.issuer_count() <= 256 and ds.blob_bytes == total
    o = orc.Engine(b"Synth Issuer 0,Synth Issuer 1", False, NOW)
    ost, ounk, oeh, ots = o.raw_batch(raw.blob, raw.bounds)
    rec = d_rec.cpu().numpy().view(ctmr.engine.RECORD_DTYPE)
    assert (rec["status"] == ost).all() and (((rec["flags"] & 2) != 0) == (ounk != 0)).all()
    assert (d_ts.cpu().numpy().astype(np.uint64) == ots).all()
    assert (d_new[:st.n_new].cpu().numpy() == np.nonzero(ounk)[0]).all()
    # explicit two-step form with a caller-owned view
    eng.reset_known()
    v_start = torch.zeros(n, dtype=torch.int64, device=dev)
    v_end = torch.zeros(n, dtype=torch.int64, device=dev)
    v_iss = torch.zeros(n, dtype=torch.int32, device=dev)
    v_et = torch.zeros(n, dtype=torch.uint8, device=dev)
    view = N.EntryView(cert_start=v_start.data_ptr(), cert_end=v_end.data_ptr(), issuer_idx=v_iss.data_ptr(),
                       entry_type=v_et.data_ptr(), timestamp=None, chain0_start=None, chain0_len=None)
    ds2 = eng.decode_entries_device(d_blob.data_ptr(), d_bounds.data_ptr(), n, view)
    assert ds2.n_issuers_added == 0 and ds2.n_x509 + ds2.n_precert == n
    st2 = eng.map_view_device(d_blob.data_ptr(), total, view, n, d_rec.data_ptr(), d_new.data_ptr())
    rec2 = d_rec.cpu().numpy().view(ctmr.engine.RECORD_DTYPE)
    assert (rec2["status"] == ost).all() and st2.n_new == st.n_new
    # issuer indices name the right certificates
    b = synth.host_batch(cfg, 0, n)
    iss_idx = v_iss.cpu().numpy().astype(np.uint32)
    ids = {k: eng.issuer_id(k) for k in range(eng.issuer_count())}
    want = {k: orc.issuer_id(_spki(c)) for k, c in enumerate(synth.issuers(cfg))}
    for i in range(0, n, 997):
        assert ids[int(iss_idx[i])] == want[int(b.issuer_idx[i])]
    eng.close()


def _spki(der):
    c = orc.parse_cert(der)
    return der[c.spki_off:c.spki_off + c.spki_len]


def test_pem_over_entry_view_device():
    """N1 over N2: PEM of the new certificates straight out of the raw blob through a caller-owned entry view."""
    cfg = synth.config(seed=31, n_issuers=4, dup_permille=100)
    n = 3000
    dev = torch.device("cuda:0")
    raw = synth.host_entries(cfg, 0, n)
    b = synth.host_batch(cfg, 0, n)
    eng = ctmr.Engine(device=0, table_slots=1 << 14, pair_slots=1 << 12)
    eng.set_filter(b"", True, NOW)
    d_blob = torch.from_numpy(raw.blob.copy()).to(dev)
    d_bounds = torch.from_numpy(raw.bounds.astype(np.int64)).to(dev)
    v_start = torch.zeros(n, dtype=torch.int64, device=dev)
    v_end = torch.zeros(n, dtype=torch.int64, device=dev)
    v_iss = torch.zeros(n, dtype=torch.int32, device=dev)
    v_et = torch.zeros(n, dtype=torch.uint8, device=dev)
    view = N.EntryView(cert_start=v_start.data_ptr(), cert_end=v_end.data_ptr(), issuer_idx=v_iss.data_ptr(),
                       entry_type=v_et.data_ptr(), timestamp=None, chain0_start=None, chain0_len=None)
    eng.decode_entries_device(d_blob.data_ptr(), d_bounds.data_ptr(), n, view)
    d_rec = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
    d_new = torch.zeros(n, dtype=torch.int64, device=dev)
    st = eng.map_view_device(d_blob.data_ptr(), int(raw.bounds[-1]), view, n, d_rec.data_ptr(), d_new.data_ptr())
    m = int(st.n_new)
    d_po = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    total = eng.pem_encode_view_device(d_blob.data_ptr(), view, d_new.data_ptr(), m, 0, 0, d_po.data_ptr())
    d_pem = torch.zeros(total, dtype=torch.uint8, device=dev)
    eng.pem_encode_view_device(d_blob.data_ptr(), view, d_new.data_ptr(), m, d_pem.data_ptr(), total, d_po.data_ptr())
    po, pem, new = d_po.cpu().numpy(), d_pem.cpu().numpy().tobytes(), d_new[:m].cpu().numpy()
    for k in range(0, m, 41):
        assert pem[po[k]:po[k + 1]] == orc.pem_encode(b.cert(int(new[k])))
    eng.close()


def test_many_chain0_certificates_sharing_length_head_and_tail_register_in_few_rounds():
    """Chain[0] candidates are selected by (length, first 16, last 16 bytes) and unregistered certificates are reported
    once per such hash per round — 300 DISTINCT issuer certificates that agree in all of that (the same certificate
    damaged in the middle: a corrupted or hostile log) used to need one registration round each and failed with
    "does not converge"; now the engine lists every unregistered entry once the first rounds did not settle it."""
    cfg = synth.config(seed=31, n_issuers=2)
    iss = synth.issuer(cfg, 0)
    cert = synth.leaf(cfg, 5)[0]
    pairs = []
    for k in range(300):
        bad = bytearray(iss)
        bad[200 + k] ^= 0x5a                        # inside the certificate, away from both ends
        pairs.append((x509_leaf(cert), chain([bytes(bad)])))
    pairs += [(x509_leaf(synth.leaf(cfg, 6 + k)[0]), chain([iss])) for k in range(50)]
    raw = RawEntries.from_pairs(pairs)
    raw.blob = np.concatenate([raw.blob, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    eng = ctmr.Engine(device=0, table_slots=1 << 12, pair_slots=1 << 10)
    eng.set_filter(b"", True, NOW)
    res = eng.map_entries(raw)
    o = orc.Engine(b"", True, NOW)
    check_against_oracle(eng, raw, o, res)
    assert eng.issuer_count() == 301 and res.decode.n_issuers_added == 301
    eng.close()


def test_trusted_log_chain0_match():
    """CTMR_CHAIN0_TRUSTED_LOG (include/ctmr.h): identical to the exact mode — and to the oracle — on everything a log
    that serves the chains it validated can produce, mutated FRAMING included; the one difference is a Chain[0] that
    copies a registered certificate except strictly between its first and last 16 bytes."""
    rng = random.Random(7)
    cfg = synth.config(seed=20260921 + 13, n_issuers=48, dup_permille=100, ca_permille=10, expired_permille=10)
    raw = synth.host_entries(cfg, 0, 30000)
    pairs = []
    for i in range(raw.n):
        leaf, extra = raw.leaf_input(i), raw.extra_data(i)
        if rng.random() < 0.1:
            leaf, extra = mutate_entry(rng, leaf, extra)          # framing damage: lengths, types, cuts, insertions
        pairs.append((leaf, extra))
    mixed = RawEntries.from_pairs(pairs)
    mixed.blob = np.concatenate([mixed.blob, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    o = orc.Engine(b"", False, NOW)
    eng = ctmr.Engine(device=0, table_slots=1 << 17, pair_slots=1 << 16)
    eng.set_filter(b"", False, NOW)
    eng.set_chain0_match(N.CHAIN0_TRUSTED_LOG)
    res = eng.map_entries(mixed)
    check_against_oracle(eng, mixed, o, res)
    n_iss = eng.issuer_count()
    # a second call: every issuer is registered, none of its bytes beyond head and tail is read again
    raw2 = synth.host_entries(cfg, 30000, 30000)
    res2 = eng.map_entries(raw2)
    check_against_oracle(eng, raw2, o, res2)
    assert eng.issuer_count() >= n_iss
    # the documented difference.  40 000 good entries of ONE issuer, then copies of that issuer certificate damaged in
    # the middle: the exact mode gives the oracle's answer (the damaged certificate is another issuer, or does not
    # parse); the trusted mode attributes them to the certificate registered for that (length, head, tail) — the one
    # with the lowest log index, whatever order the waves ran in — and differs NOWHERE else.
    cfg1 = synth.config(seed=77, n_issuers=1)
    good = synth.host_entries(cfg1, 0, 40000)
    iss = synth.issuer(cfg1, 0)
    pairs = [(good.leaf_input(i), good.extra_data(i)) for i in range(good.n)]
    n_bad = 64
    src = [k for k in range(400) if good.leaf_input(k)[10:12] == b"\x00\x00"][:n_bad]   # x509 entries: extra_data = the chain
    for t, k in enumerate(src):
        dmg = bytearray(iss)
        dmg[200 + 3 * t] ^= 0x40                                    # inside the body, far from head and tail
        pairs.append((good.leaf_input(k), chain([bytes(dmg)])))
    both = RawEntries.from_pairs(pairs)
    both.blob = np.concatenate([both.blob, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    out = {}
    for mode in (N.CHAIN0_EXACT, N.CHAIN0_TRUSTED_LOG):
        e2 = ctmr.Engine(device=0, table_slots=1 << 17, pair_slots=1 << 16)
        e2.set_filter(b"", False, NOW)
        e2.set_chain0_match(mode)
        out[mode] = e2.map_entries(both)
        if mode == N.CHAIN0_EXACT:
            check_against_oracle(e2, both, orc.Engine(b"", False, NOW), out[mode])
        e2.close()
    ex, tr = out[N.CHAIN0_EXACT].records, out[N.CHAIN0_TRUSTED_LOG].records
    head = slice(0, good.n)
    for f in ("status", "flags", "serial_len", "exp_hour", "serial", "issuer_idx"):
        assert (ex[f][head] == tr[f][head]).all(), f
    # trusted: the damaged copies behave exactly like the entries whose leaves they repeat (0..63), as known duplicates
    assert (tr["status"][good.n:] == tr["status"][src]).all()
    was_pass = tr["status"][src] == orc.ST_PASS
    assert was_pass.sum() > n_bad // 2
    assert (tr["issuer_idx"][good.n:][was_pass] == tr["issuer_idx"][src][was_pass]).all()
    assert ((tr["flags"][good.n:][was_pass] & 2) == 0).all()
    # exact: wherever the leaf gets as far as the issuer, it is NOT the good certificate's
    ex_tail = ex[good.n:][was_pass]
    assert ((ex_tail["status"] != orc.ST_PASS) | (ex_tail["issuer_idx"] != ex["issuer_idx"][src][was_pass])).all()
    eng.close()


def test_trusted_log_never_trusts_registered_certificates_that_share_a_candidate_hash():
    """Two REGISTERED certificates that agree in length, first and last 16 bytes are always compared bytewise: the
    trusted mode then equals the exact mode even on such a pair."""
    cfg = synth.config(seed=78, n_issuers=1)
    good = synth.host_entries(cfg, 0, 6000)
    iss = synth.issuer(cfg, 0)
    twin = bytearray(iss)
    twin[len(iss) // 2] ^= 0x01                                      # the signature no longer verifies; nobody checks it
    twin = bytes(twin)
    pairs = []
    for i in range(good.n):
        pairs.append((good.leaf_input(i), chain([twin if i % 3 == 0 else iss])))
    raw = RawEntries.from_pairs(pairs)
    raw.blob = np.concatenate([raw.blob, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    recs = {}
    for mode in (N.CHAIN0_EXACT, N.CHAIN0_TRUSTED_LOG):
        eng = ctmr.Engine(device=0, table_slots=1 << 15, pair_slots=1 << 14)
        eng.set_filter(b"", False, NOW)
        eng.set_chain0_match(mode)
        eng.add_issuers([iss, twin])
        res = eng.map_entries(raw)
        check_against_oracle(eng, raw, orc.Engine(b"", False, NOW), res)
        recs[mode] = res.records
        eng.close()
    for f in ("status", "flags", "issuer_idx", "serial"):
        assert (recs[N.CHAIN0_EXACT][f] == recs[N.CHAIN0_TRUSTED_LOG][f]).all(), f


def test_trusted_log_compares_bytes_when_two_registered_certificates_share_tag_and_length():
    """Round-2 advisor finding: the trusted-log match accepts a candidate on the UPPER HALF of the candidate hash (what a
    table word carries) + the length, but only certificates with equal FULL hashes were marked as twins.  Two registered
    certificates of one length whose hashes agree in the upper 32 bits and in the table's home slot — found here by
    search, with a numpy port of cert_quick_hash over the signature's last 16 bytes — must still be told apart: every
    entry's Chain[0] is the SECOND one, the first one sits in front of it in the probe run."""
    U = np.uint64

    def qh_mix(z):
        z = (z ^ (z >> U(30))) * U(0xbf58476d1ce4e5b9)
        z = (z ^ (z >> U(27))) * U(0x94d049bb133111eb)
        return z ^ (z >> U(31))

    cfg = synth.config(seed=79, n_issuers=1)
    iss = synth.issuer(cfg, 0)
    L = len(iss)
    rng = np.random.default_rng(20260923)
    n = 1 << 22
    tails = rng.integers(0, 1 << 32, (n, 4), dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = qh_mix(U(0x9e3779b97f4a7c15) + U(L))
        hd = np.frombuffer(iss[:16], "<u4").astype(np.uint64)
        for k in range(4):
            h = qh_mix(h ^ (hd[k] << U(1) | U(1)))
        h = np.full(n, h, np.uint64)
        for k in range(4):
            h = qh_mix(h ^ (tails[:, k] << U(1)))
    slots = 1024                                                   # idb_ht_size of an engine with max_issuers <= 256
    sig = ((h >> U(32)) << U(10)) | (h & U(slots - 1))
    order = np.argsort(sig, kind="stable")
    same = np.nonzero(sig[order][1:] == sig[order][:-1])[0]
    assert len(same) >= 1, "search space too small"
    a, b = int(order[same[0]]), int(order[same[0] + 1])
    assert h[a] != h[b]                                             # different hashes: round 2 saw no twins here
    certs = [iss[:-16] + tails[k].astype("<u4").tobytes() for k in (a, b)]
    assert certs[0] != certs[1] and len(certs[0]) == L
    from tests import harness                                       # the numpy port IS the product's hash (host build)
    harness.product_decode_entry(b"\0" * 16, b"\0" * 4)              # (builds and binds the harness library)
    for k, c in zip((a, b), certs):
        assert harness._entry.harness_quick_hash(c, len(c)) == int(h[k])
    good = synth.host_entries(cfg, 0, 3000)
    raw = RawEntries.from_pairs([(good.leaf_input(i), chain([certs[1]])) for i in range(good.n)])
    raw.blob = np.concatenate([raw.blob, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    recs = {}
    for mode in (N.CHAIN0_EXACT, N.CHAIN0_TRUSTED_LOG):
        eng = ctmr.Engine(device=0, table_slots=1 << 15, pair_slots=1 << 14, max_issuers=256)
        eng.set_filter(b"", False, NOW)
        eng.set_chain0_match(mode)
        eng.add_issuers(certs)
        res = eng.map_entries(raw)
        recs[mode] = res.records
        assert (res.records["issuer_idx"][res.records["status"] == 0] == 1).all(), mode   # the second certificate, bytewise
        eng.close()
    for f in ("status", "flags", "issuer_idx", "serial"):
        assert (recs[N.CHAIN0_EXACT][f] == recs[N.CHAIN0_TRUSTED_LOG][f]).all(), f


def test_raw_entries_wrapped_around_the_reference_certificates_on_the_gpu():
    """The committed N2 fixtures (tests/golden/entries_from_reference_pems.json: the reference's own test certificates
    as RFC 6962 leaves / extra_data, encoded independently, expectations derived from the reference's goldens) through
    the product: decode + Chain[0] match + map + reduce on the GPU, in both Chain[0] modes."""
    from tests.test_oracle_golden import load_entry_fixture, STATUS_NAMES
    fx, pairs, raw = load_entry_fixture()
    for mode in (N.CHAIN0_EXACT, N.CHAIN0_TRUSTED_LOG):
        eng = ctmr.Engine(device=0, table_slots=1 << 12, pair_slots=1 << 10)
        eng.set_filter(fx["filter"].encode(), fx["log_expired"], fx["now"])
        eng.set_chain0_match(mode)
        res = eng.map_entries(raw)
        for i, e in enumerate(fx["entries"]):
            r = res.records[i]
            assert int(r["status"]) == STATUS_NAMES[e["status"]] and bool(r["flags"] & 2) == e["was_unknown"], e["name"]
            assert int(res.timestamp[i]) == e["timestamp"] and bool(r["flags"] & N.FL_PRECERT) == (e["entry_type"] == 1)
            if e.get("serial_hex"):
                assert bytes(r["serial"][:int(r["serial_len"])]).hex() == e["serial_hex"]
                assert orc.exp_date_id(int(r["exp_hour"])) == e["exp_date"]
                assert eng.issuer_id(int(r["issuer_idx"])) == e["issuer_id"]
        assert sorted(k.decode() for k in eng.keys(b"serials::*")) == fx["final_keys"]
        assert eng.total_count() == fx["final_total_count"] and eng.issuer_count() == 2     # kEmptySPKI, kRealSPKI registered on the way
        for k in fx["final_keys"]:
            assert eng.set_list(k.encode()) == [bytes.fromhex("00aa")]
        check_against_oracle(eng, raw, orc.Engine(fx["filter"].encode(), fx["log_expired"], fx["now"]), res)
        eng.close()


def test_strict_leaf_walks_the_leaf_tbs_of_precertificate_entries():
    """ctmr_set_strict_leaf(1): the leaf TBSCertificate of every precertificate entry is parsed as ct.LogEntryFromLeaf
    does (ct-fetch.go:452); entries whose TBS fails are undecodable and their Chain[0] is never registered.  Synthetic
    entries with the leaf damaged at random (the precert entries' TBS lies in leaf_input) + hand-built cases, against the
    oracle in strict mode; the default mode on the same input keeps its round-2 answers."""
    from tests.test_walk_cpu import tbs_of, mutate
    rng = random.Random(20260925)
    cfg = synth.config(seed=15, n_issuers=8, dup_permille=50)
    raw = synth.host_entries(cfg, 0, 4000)
    only = synth.issuer(synth.config(seed=99, n_issuers=1), 0)          # an issuer that ONLY broken-leaf entries name
    cert = synth.leaf(cfg, 7)[0]
    pairs = []
    for i in range(raw.n):
        leaf, extra = raw.leaf_input(i), raw.extra_data(i)
        if leaf[10:12] == b"\x00\x01" and rng.random() < 0.5:           # damage inside the leaf's TBSCertificate only
            tbs = mutate(rng, leaf[47:-2])
            leaf = leaf[:44] + len(tbs).to_bytes(3, "big") + tbs + leaf[-2:]
        pairs.append((leaf, extra))
    bad = bytearray(tbs_of(cert)); bad[1] = 0x84
    pairs += [(precert_leaf(bytes(bad), ts=7), asn1cert(cert) + chain([only])),
              (precert_leaf(tbs_of(cert) + b"\x00", ts=8), asn1cert(cert) + chain([only])),
              (precert_leaf(b"\x30\x00", ts=9), asn1cert(cert) + chain([only]))]
    rng.shuffle(pairs)
    mixed = RawEntries.from_pairs(pairs)
    mixed.blob = np.concatenate([mixed.blob, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    seen = {}
    for strict in (True, False):
        eng = ctmr.Engine(device=0, table_slots=1 << 15, pair_slots=1 << 14)
        eng.set_filter(b"", True, NOW)
        eng.set_strict_leaf(strict)
        res = eng.map_entries(mixed)
        o = orc.Engine(b"", True, NOW)
        o.set_strict_leaf(strict)
        st, _ = check_against_oracle(eng, mixed, o, res)
        seen[strict] = (int((st == orc.ST_ENTRY_DECODE_ERROR).sum()), eng.issuer_count())
        eng.close()
    assert seen[True][0] > seen[False][0] + 200                          # the damaged leaves are what the strict mode drops
    assert seen[False][1] == seen[True][1] + 1                           # … and their Chain[0] was never registered
