"""-m gpu: the HIP path through the C ABI vs the CPU oracle, bit for bit."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402,F401  (loaded first so that libctmr binds to the same HIP runtime)

import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth, _native as N
from ct_mapreduce_amd.engine import Batch
from oracle import oracle as orc
from tests import der as D
from tests.gpu_common import run_oracle, assert_records_equal, assert_state_equal
from tests.test_walk_cpu import mutate, edge_seeds

NOW = synth.BASE_TIME


# 15 = k_map_fused (the default: map + pass 1 of the insert in one kernel), 13 = k_map_winc + k_insert (what the
# exchange modes run).  The retired designs live in the sweep build only (scripts/sweep.py).
@pytest.fixture(params=[(13, 0, 0), (15, 0, 0)], ids=["winc_separate_insert", "fused"])
def variant(request):
    return request.param


def make_engine(variant, **kw):
    v, c, lds = variant
    kw.setdefault("table_slots", 1 << 18)
    kw.setdefault("pair_slots", 1 << 16)
    return ctmr.Engine(device=0, map_variant=v, certs_per_tile=c, lds_tile_bytes=lds, **kw)


def test_synthetic_batch_bit_exact(variant):
    cfg = synth.config(seed=20260921 + 3, n_issuers=256, dup_permille=100, ca_permille=10, expired_permille=10)
    batch = synth.host_batch(cfg, 0, 30000)
    issuers = synth.issuers(cfg)
    filt = b"Synth Issuer 0,Synth Issuer 1"
    eng = make_engine(variant)
    assert eng.add_issuers(issuers) == 0
    eng.set_filter(filt, False, NOW)
    res = eng.map_batch(batch)
    o, st, unk, eh = run_oracle(batch, issuers, filt, False, NOW)
    assert (st == orc.ST_FILTERED_CN).sum() > 0 and (st == orc.ST_FILTERED_CA).sum() > 0
    assert (st == orc.ST_FILTERED_EXPIRED).sum() > 0 and 0 < unk.sum() < (st == 0).sum()
    assert_records_equal(res, batch, st, unk, eh)
    assert_state_equal(eng, o, len(issuers))
    # second batch overlapping the first: cross-batch duplicates
    batch2 = synth.host_batch(cfg, 20000, 20000)
    res2 = eng.map_batch(batch2)
    _, st2, unk2, eh2 = run_oracle(batch2, issuers, engine=o)
    assert_records_equal(res2, batch2, st2, unk2, eh2)
    assert_state_equal(eng, o, len(issuers))
    # idempotence: replaying a batch finds nothing new
    res3 = eng.map_batch(batch)
    assert res3.stats.n_new == 0 and res3.stats.n_dup == int((st == 0).sum())
    eng.close()


def test_issuer_ids_match_oracle_and_goldens(golden_certs, variant):
    cfg = synth.config(n_issuers=8)
    issuers = synth.issuers(cfg) + [golden_certs["kEmptySPKI"], golden_certs["kRealSPKI"],
                                    golden_certs["kLeadingZeroes"], b"\x30\x00", b""]
    eng = make_engine(variant)
    eng.add_issuers(issuers)
    for k, der in enumerate(issuers):
        c = orc.parse_cert(der)
        info = eng.issuer_info(k)
        assert bool(info.valid) == bool(c.ok)
        if c.ok:
            assert info.issuer_id.decode() == orc.issuer_id(der[c.spki_off:c.spki_off + c.spki_len])
    # G2/G4 share one SPKI → one canonical issuer
    assert eng.issuer_id(8) == eng.issuer_id(10) == "VCIlmPM9NkgFQtrs4Oa5TeFcDu6MWRTKSNdePEhOgD8="
    assert eng.issuer_info(10).canonical_idx == 8
    assert eng.issuer_id(9) == "d_Kor69hknpIfroNumzs6NkLxxCUNhMn46dzck_SZSQ="
    eng.close()


def test_golden_entry_through_gpu(golden_certs, variant):
    """Derived expectation of SURVEY §8(c): leaf G2 issued by G4."""
    eng = make_engine(variant)
    eng.add_issuers([golden_certs["kEmptySPKI"]])
    eng.set_filter(b"", False, 1546300800)   # 2019-01-01
    b = Batch.from_certs([golden_certs["kLeadingZeroes"], golden_certs["kRealSPKI"],
                          golden_certs["kLeadingZeroes"]], [0, 0, 0], [0, 1, 1])
    b.payload = np.concatenate([b.payload, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    res = eng.map_batch(b)
    assert list(res.records["status"]) == [0, orc.ST_FILTERED_CA, 0]
    assert list(res.records["flags"]) == [2, 1, 1]
    assert bytes(res.records["serial"][0][:2]) == b"\x00\xaa" and res.records["serial_len"][0] == 2
    key = b"serials::2020-02-05-00::VCIlmPM9NkgFQtrs4Oa5TeFcDu6MWRTKSNdePEhOgD8="
    assert eng.keys(b"serials::*") == [key]
    assert eng.set_list(key) == [b"\x00\xaa"]
    assert eng.set_cardinality(key) == 1 and eng.exists(key)
    # expiry = the truncated hour (knowncertificates.go:98-104): sweeping just before keeps, at drops
    assert eng.expire_sweep(1580860800 - 1) == 0
    assert eng.expire_sweep(1580860800) == 1
    assert eng.keys(b"serials::*") == [] and eng.total_count() == 0
    eng.close()


def test_edge_cases_and_fuzz(golden_certs, variant):
    rng = random.Random(99)
    cfg = synth.config(seed=5, n_issuers=4, ca_permille=100, expired_permille=100)
    issuers = synth.issuers(cfg) + [b"\x30\x03\x02\x01\x00", synth.issuer(cfg, 1),   # 4 invalid, 5 dup SPKI of 1
                                    D.cert(serial=b"\x00\x01", exts=[D.BC_CA])]         # 6: a non-fatal finding is an err for Chain[0]
    certs, iss, ets = [], [], []

    def add(c, i=0, et=0):
        certs.append(c); iss.append(i); ets.append(et)

    seeds = list(golden_certs.values()) + [synth.leaf(cfg, i)[0] for i in range(20)]
    for r in range(3000):
        add(mutate(rng, seeds[r % len(seeds)]), rng.randrange(4), rng.randrange(2))
    # the Go-specific rules (numeric zones, lax INTEGERs and negative serials — kept for X509 entries, dropped for
    # precertificates —, unique ids, high tag numbers, wrappers that lie about their length), plain and mutated
    edges = edge_seeds()
    for c in edges:
        add(c, 0, 0); add(c, 0, 1)
    for r in range(3000):
        c = mutate(rng, edges[r % len(edges)])
        add(mutate(rng, c) if r % 3 == 0 and len(c) > 1 else c, rng.randrange(4), rng.randrange(2))
    add(b"")                                            # empty record
    add(b"\x30")
    add(synth.leaf(cfg, 1)[0], N.NO_ISSUER)             # chain empty
    add(synth.leaf(cfg, 2)[0], 4)                       # issuer cert does not parse
    add(synth.leaf(cfg, 3)[0], 77)                      # out-of-range index behaves as no issuer
    a = synth.leaf(cfg, 6)[0]
    add(a, 1); add(a, 5)                                # same SPKI through two issuer entries → duplicate
    add(synth.leaf(cfg, 7)[0], 6)                       # issuer certificate with a lax INTEGER: ISSUER_PARSE_ERROR
    for s in (b"\x00\xaa", bytes(range(1, 21)), bytes(range(1, 22)), bytes(range(1, 31)),
              bytes(range(1, 41)), bytes(range(1, 42)), b"\x05" * 45, b"\x06" * 300, bytes(range(1, 31)),
              b"\x05" * 45):
        add(D.cert(serial=s, issuer=D.name(D.rdn(3, b"Synth Issuer 000"))), 2)
    add(D.cert(exts=[D.ext(0x11, D.seq(D.tlv(0x82, b"a" * 70000)))], serial=b"\x77"), 2)   # > LDS tile
    add(bytes(rng.randrange(256) for _ in range(100000)), 1)                                # junk > LDS
    add(D.cert(not_after=D.gentime("20491231235959Z"), serial=b"\x78"), 3)
    add(D.cert(not_after=D.gentime("19500101000000Z"), serial=b"\x79"), 3)
    for r in range(200):
        add(synth.leaf(cfg, 100 + r)[0], r % 4, r % 2)
    batch = Batch.from_certs(certs, iss, ets)
    batch.payload = np.concatenate([batch.payload, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    for filt, log_exp in ((b"", False), (b"Synth Issuer 000,Test", False), (b"zz,", True), (b"Synth Issuer 00", True)):
        eng = make_engine(variant)
        eng.add_issuers(issuers)
        eng.set_filter(filt, log_exp, NOW)
        res = eng.map_batch(batch)
        o, st, unk, eh = run_oracle(batch, issuers, filt, log_exp, NOW)
        assert_records_equal(res, batch, st, unk, eh)
        assert_state_equal(eng, o, len(issuers), sample_keys=1000)
        assert res.stats.n_host_set == sum(1 for i in range(batch.n) if st[i] == 0 and
                                           orc.parse_cert(batch.cert(i)).serial_len > 40)
        eng.close()


def test_empty_and_tiny_batches(variant):
    cfg = synth.config(n_issuers=1)
    eng = make_engine(variant)
    eng.add_issuers(synth.issuers(cfg))
    eng.set_filter(b"", False, NOW)
    res = eng.map_batch(Batch.from_certs([], []))
    assert res.stats.n == 0 and len(res.new_idx) == 0
    for n in (1, 2, 63, 64, 65, 1023, 1024, 1025):
        eng.reset_known()
        b = synth.host_batch(cfg, 1000, n)
        res = eng.map_batch(b)
        o, st, unk, eh = run_oracle(b, synth.issuers(cfg), b"", False, NOW)
        assert_records_equal(res, b, st, unk, eh)
    eng.close()


def test_remote_cache_set_semantics(variant):
    """storage/knowncertificates_test.go:11-83 and storage/mockcache.go semantics through the ABI."""
    cfg = synth.config(n_issuers=2)
    eng = make_engine(variant)
    eng.add_issuers(synth.issuers(cfg))
    iid = eng.issuer_id(0)
    for key in (b"serials::2029-01-30::test issuer",                    # host-side store (unregistered issuer)
                ("serials::2029-01-30-07::" + iid).encode()):          # in-HBM table
        for s in (b"\x01", b"\x02", b"\x03", b"\x04"):
            assert eng.set_insert(key, s)
        for s in (b"\x01", b"\x02", b"\x03", b"\x04"):
            assert not eng.set_insert(key, s)                          # known
        assert eng.set_insert(key, b"\x05") and not eng.set_insert(key, b"\x05")
        assert eng.set_list(key) == [b"\x01", b"\x02", b"\x03", b"\x04", b"\x05"]
        assert eng.set_cardinality(key) == 5
        assert eng.set_contains(key, b"\x03") and not eng.set_contains(key, b"\x06")
        assert eng.set_contains(key, b"\x00\x03") is False             # length is part of the member
        assert eng.set_remove(key, b"\x03") and not eng.set_remove(key, b"\x03")
        assert eng.set_list(key) == [b"\x01", b"\x02", b"\x04", b"\x05"]
        assert eng.set_insert(key, b"\x03")
        assert eng.set_cardinality(key) == 5 and eng.exists(key)
        long = b"\x07" * 45                                            # > CTMR_MAX_SERIAL → host part
        assert eng.set_insert(key, long) and not eng.set_insert(key, long)
        assert eng.set_cardinality(key) == 6 and eng.set_contains(key, long)
        assert eng.set_remove(key, long) and eng.set_cardinality(key) == 5
    assert sorted(eng.keys(b"serials::*")) == sorted([b"serials::2029-01-30::test issuer",
                                                      ("serials::2029-01-30-07::" + iid).encode()])
    assert eng.keys(b"serials::2029-01-30-0?::*") == [("serials::2029-01-30-07::" + iid).encode()]
    assert int(eng.issuer_counts()[0]) == 5 and eng.total_count() == 5
    # other RemoteCache keys (IssuerMetadata's crl::/issuer:: sets) live host-side
    assert eng.set_insert(b"crl::" + iid.encode(), b"http://::1/file.crl")
    assert not eng.set_insert(b"crl::" + iid.encode(), b"http://::1/file.crl")
    assert eng.set_list(b"crl::" + iid.encode()) == [b"http://::1/file.crl"]
    eng.expire_at(b"serials::2029-01-30::test issuer", 100)
    assert eng.expire_sweep(99) == 0 and eng.expire_sweep(100) == 5
    assert not eng.exists(b"serials::2029-01-30::test issuer")
    eng.close()


def test_device_generator_matches_host_and_large_properties(variant):
    """BASELINE config 2 shape at 1M entries: size-independent properties, plus device-generated
    bytes == host-generated bytes on a slice."""
    import torch
    n = 1_000_000
    cfg = synth.config(seed=20260921 + 2, n_issuers=1, dup_permille=0)
    eng = make_engine(variant, table_slots=1 << 22, profile=True)
    issuers = synth.issuers(cfg)
    eng.add_issuers(issuers)
    eng.set_filter(b"", False, NOW)
    dev = torch.device("cuda:0")
    d_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    total = eng.synth_device(cfg, 0, n, d_off.data_ptr(), 0, 0, 0, 0)
    d_pay = torch.empty(total + N.PAYLOAD_PAD + 16, dtype=torch.uint8, device=dev)
    d_iss = torch.empty(n, dtype=torch.int32, device=dev)
    d_et = torch.empty(n, dtype=torch.uint8, device=dev)
    assert eng.synth_device(cfg, 0, n, d_off.data_ptr(), d_pay.data_ptr(), d_pay.numel(),
                            d_iss.data_ptr(), d_et.data_ptr()) == total
    hb = synth.host_batch(cfg, 0, 5000)
    off = d_off[:5001].cpu().numpy().astype(np.uint64)
    assert (off == hb.offsets).all()
    assert (d_pay[:int(off[-1])].cpu().numpy() == hb.payload[:int(off[-1])]).all()
    assert (d_iss[:5000].cpu().numpy().astype(np.uint32) == hb.issuer_idx).all()
    assert (d_et[:5000].cpu().numpy() == hb.entry_type).all()
    d_rec = torch.empty(n * 32, dtype=torch.uint8, device=dev)
    d_new = torch.empty(n, dtype=torch.int64, device=dev)
    st = eng.map_batch_device(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), n,
                              d_rec.data_ptr(), d_new.data_ptr())
    assert st.n == n and sum(st.by_status) == n and st.by_status[1] == 0
    assert st.n_new + st.n_dup == st.by_status[0]
    assert st.n_new == eng.total_count() == int(eng.issuer_counts()[0])
    new = d_new[:st.n_new].cpu().numpy()
    assert (np.diff(new) > 0).all()                                     # ascending log index
    rec = d_rec.cpu().numpy().view(ctmr.engine.RECORD_DTYPE)
    assert ((rec["flags"] & 2) != 0).sum() == st.n_new
    assert (np.nonzero(rec["flags"] & 2)[0] == new).all()
    # the first 5000 records equal the oracle's
    o, ost, unk, eh = run_oracle(hb, issuers, b"", False, NOW)
    assert (rec["status"][:5000] == ost).all() and (rec["exp_hour"][:5000][ost != 1] == eh[ost != 1]).all()
    # replay: everything that passed is now known
    st2 = eng.map_batch_device(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), n)
    assert st2.n_new == 0 and st2.n_dup == st.by_status[0] and eng.total_count() == st.n_new
    assert st.ms_map > 0
    eng.close()


def test_real_world_roots_as_issuers_and_leaves():
    """The CA bundles shipped in the image (RSA/EC keys, UTCTime/GeneralizedTime, names without CN, v1 certificates):
    issuer IDs computed on the GPU ≡ hashlib over OpenSSL's SPKI; as leaves, every record field ≡ oracle."""
    import base64
    import hashlib
    from tests import harness
    from tests.test_real_certs_cpu import bundle_ders
    ders = bundle_ders()
    if len(ders) < 50:
        pytest.skip("no CA bundle in this image")
    eng = ctmr.Engine(device=0, table_slots=1 << 12, pair_slots=1 << 10)
    eng.add_issuers(ders)
    for k, d in enumerate(ders):
        o = harness.ossl_extract(d)
        want = base64.urlsafe_b64encode(hashlib.sha256(bytes(o.spki[:o.spki_len])).digest()).decode()
        info = eng.issuer_info(k)
        assert info.valid and info.issuer_id.decode() == want, k
    for log_expired, now in ((True, NOW), (False, NOW), (False, 0)):
        eng.reset_known()
        eng.set_filter(b"", log_expired, now)
        batch = Batch.from_certs(ders, list(range(len(ders))))
        batch.payload = np.concatenate([batch.payload, np.zeros(N.PAYLOAD_PAD, np.uint8)])
        res = eng.map_batch(batch)
        o, st, unk, eh = run_oracle(batch, ders, b"", log_expired, now)
        assert_records_equal(res, batch, st, unk, eh)
        assert_state_equal(eng, o, len(ders))
    # every certificate of the bundle is a CA (or expired): certIsFilteredOut stops them all, nothing reaches Store
    assert ((st == orc.ST_FILTERED_CA) | (st == orc.ST_FILTERED_EXPIRED)).all() and eng.total_count() == 0
    eng.close()


def test_long_names_and_mixed_keys(variant):
    """OV/EV-like certificates: issuer and subject names of up to several hundred bytes push the SPKI header and
    the extension block past the front window (refills, not the slow path), EC and RSA keys mixed in one wave."""
    rng = random.Random(31337)

    def long_name(n_attrs, width):
        return D.name(*[D.rdn(a, bytes(rng.choice(b"abcdefghij KLMNOP") for _ in range(width)))
                        for a in ([6, 8, 7, 10, 11, 3] * 3)[:n_attrs]])
    rsa_spki = bytes.fromhex("30820122300d06092a864886f70d01010105000382010f003082010a0282010100") + \
        bytes(rng.randrange(256) for _ in range(256)) + bytes.fromhex("0203010001")
    cfg = synth.config(seed=8, n_issuers=3)
    issuers = synth.issuers(cfg)
    certs, iss = [], []
    for k in range(700):
        exts = [D.ext(0x0f, D.tlv(0x03, b"\x05\xa0"), True), D.BC_NOT_CA if k % 7 else D.BC_CA,
                D.ext(0x0e, D.tlv(0x04, bytes(20))),
                D.ext(0x1f, D.seq(D.seq(D.tlv(0xa0, D.tlv(0xa0, D.tlv(0x86, b"http://crl.example.com/x.crl")))))),
                D.ext(0x11, D.seq(*[D.tlv(0x82, b"host%d.example.com" % j) for j in range(rng.randrange(1, 40))]))]
        certs.append(D.cert(serial=bytes([1] + [rng.randrange(256) for _ in range(rng.randrange(1, 19))]),
                            issuer=long_name(rng.randrange(1, 7), rng.randrange(4, 60)),
                            subject=long_name(rng.randrange(1, 10), rng.randrange(4, 70)),
                            spki=rsa_spki if k % 2 else D.EC_SPKI, exts=exts if k % 11 else None,
                            not_after=D.gentime("20300101000000Z") if k % 5 == 0 else None))
        iss.append(k % 3)
    batch = Batch.from_certs(certs, iss)
    batch.payload = np.concatenate([batch.payload, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    eng = make_engine(variant)
    eng.add_issuers(issuers)
    eng.set_filter(b"a,b,c,d,e,f,g,h,i,j,K,L", False, NOW)          # CommonName first letters: about half pass
    res = eng.map_batch(batch)
    o, st, unk, eh = run_oracle(batch, issuers, b"a,b,c,d,e,f,g,h,i,j,K,L", False, NOW)
    assert (st == 0).sum() > 40 and (st == orc.ST_FILTERED_CN).sum() > 50 and (st == orc.ST_FILTERED_CA).sum() > 50
    assert_records_equal(res, batch, st, unk, eh)
    assert_state_equal(eng, o, len(issuers))
    eng.close()


def test_mixed_synthetic_corpus_bit_exact(variant):
    """The mixed corpus (profile=1): lanes of one wave walk EC and RSA certificates, 38-byte and 250-byte subjects,
    UTCTime and GeneralizedTime side by side."""
    cfg = synth.config(seed=20260921 + 9, n_issuers=64, dup_permille=100, ca_permille=10, expired_permille=10, profile=1)
    batch = synth.host_batch(cfg, 0, 20000)
    issuers = synth.issuers(cfg)
    filt = b"Synth Issuer 00,Synth Issuer 01,Synth Issuer 02"
    eng = make_engine(variant)
    eng.add_issuers(issuers)
    eng.set_filter(filt, False, NOW)
    res = eng.map_batch(batch)
    o, st, unk, eh = run_oracle(batch, issuers, filt, False, NOW)
    assert (st == 0).sum() > 5000 and 0 < unk.sum() < (st == 0).sum()
    assert_records_equal(res, batch, st, unk, eh)
    assert_state_equal(eng, o, len(issuers))
    eng.close()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 2047, 4093, 4099])
def test_batch_sizes_around_the_reduce_tail_boundaries(n):
    """k_insert2 / k_compact handle four entries per thread and 1 024 per block: every size class around those
    boundaries, with in-batch duplicates (DEFER entries) and a second batch of pure duplicates."""
    cfg = synth.config(seed=777 + n, n_issuers=4, dup_permille=300, ca_permille=50, expired_permille=50)
    issuers = synth.issuers(cfg)
    eng = make_engine((15, 0, 0), table_slots=1 << 15, pair_slots=1 << 15)
    eng.add_issuers(issuers)
    eng.set_filter(b"", False, NOW)
    o = None
    for first in (0, 0, n):                       # the batch, the same batch again (all known), the next entries
        b = synth.host_batch(cfg, first, n)
        res = eng.map_batch(b)
        o, st, unk, eh = run_oracle(b, issuers, b"", False, NOW, engine=o)
        assert_records_equal(res, b, st, unk, eh)
    assert_state_equal(eng, o, len(issuers))
    eng.close()
