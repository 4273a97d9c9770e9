"""-m gpu, N3: IssuerMetadata.Accumulate's memo on the GPU (k_meta_new) — the first sightings it
reports are exactly those the reference's per-issuer maps (storage/issuermetadata.go:92-138) would produce from the
newly unknown certificates, computed here with the oracle's field extraction (orc_cert_meta)."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402,F401

import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth, _native as N
from tests import storage_mirror as S
from ct_mapreduce_amd.engine import Batch
from oracle import oracle as orc
from tests import der as D

NOW = synth.BASE_TIME


def dp_ext(*points, critical=None):
    return D.ext(0x1f, D.seq(*points), critical)


def dp(*general_names, reasons=None, crl_issuer=None):
    parts = []
    if general_names:
        parts.append(D.tlv(0xa0, D.tlv(0xa0, b"".join(general_names))))
    if reasons:
        parts.append(D.tlv(0x81, reasons))
    if crl_issuer:
        parts.append(D.tlv(0xa2, crl_issuer))
    return D.seq(*parts)


def uri(s):
    return D.tlv(0x86, s)


def expected_first_sightings(certs, issuer_canon, new_idx, exp_hours):
    """The reference's memo semantics over the new certificates, in log order."""
    seen, out = set(), set()
    for i in new_idx:
        meta = orc.cert_meta(certs[i])
        assert meta is not None
        name, uris, m = meta
        c = issuer_canon[i]
        host = (m.n_crl_ext > 1 or len(name) > 4096 or any(len(u) > 4096 for u in uris) or m.n_crl > 4
                or len(certs[i]) > 0xfffe)
        items = [(N.MK_EXPDATE, c, int(exp_hours[i]), b"")]
        if host:
            out.add((N.MK_HOST, i))
        else:
            items += [(N.MK_DN, c, 0, name)] + [(N.MK_CRL, c, 0, u) for u in uris]
        for it in items:
            if it not in seen:
                seen.add(it)
                out.add(it)
    return out


def got_first_sightings(eng, items):
    out = set()
    for kind, entry, idx, exp_hour, b in items:
        c = eng.issuer_info(idx).canonical_idx
        if kind == N.MK_HOST:
            out.add((kind, entry))
        elif kind == N.MK_EXPDATE:
            out.add((kind, c, exp_hour, b""))
        else:
            out.add((kind, c, 0, b))
    return out


def test_first_sightings_on_synthetic_batches():
    cfg = synth.config(seed=20260921 + 21, n_issuers=32, dup_permille=100)
    issuers = synth.issuers(cfg)
    eng = ctmr.Engine(device=0, table_slots=1 << 17, pair_slots=1 << 14, collect_meta=True)
    eng.add_issuers(issuers)
    eng.set_filter(b"", True, NOW)
    seen_total = set()
    for first, n in ((0, 20000), (15000, 20000)):
        b = synth.host_batch(cfg, first, n)
        res = eng.map_batch(b)
        certs = [b.cert(i) for i in range(n)]
        exp = expected_first_sightings(certs, [int(k) for k in b.issuer_idx], [int(i) for i in res.new_idx],
                                       res.records["exp_hour"])
        exp -= seen_total                                   # the memo persists across batches
        items = eng.meta_new()
        got = got_first_sightings(eng, items)
        assert len(items) == len(got)                       # nothing reported twice
        assert got == exp
        assert eng.meta_new() == items                      # idempotent until the next batch
        seen_total |= exp
    kinds = {k[0] for k in seen_total}
    assert kinds == {N.MK_EXPDATE, N.MK_CRL, N.MK_DN}
    assert len([k for k in seen_total if k[0] == N.MK_DN]) == 32
    # after a reset everything is a first sighting again
    eng.meta_reset()
    res = eng.map_batch(synth.host_batch(cfg, 0, 10))
    assert res.stats.n_new == 0 and eng.meta_new() == []
    eng.close()


def test_crl_and_dn_edge_cases():
    iss_cert = D.cert(serial=b"\x11", exts=[D.BC_CA], subject=D.name(D.rdn(3, b"Edge CA")))
    other = D.cert(serial=b"\x12", exts=[D.BC_CA], subject=D.name(D.rdn(3, b"Edge CA 2")), spki=D.EC_SPKI_2)
    n1 = D.name(D.rdn(6, b"US", 0x13), D.rdn(10, b"Edge Org"), D.rdn(3, b"Edge CA"))
    n2 = D.name(D.rdn(3, b"Edge CA"), D.rdn(10, b"Edge Org"))          # same attributes, other order: other DN
    big = D.name(D.rdn(10, b"x" * 5000), D.rdn(3, b"Edge CA"))
    u1, u2 = b"http://crl.example/a.crl", b" http://crl.example/a.crl  "
    ldap = b"ldap://dir.example/cn=x?certificateRevocationList"
    certs = [
        D.cert(serial=b"\x01", issuer=n1, exts=[D.BC_NOT_CA, dp_ext(dp(uri(u1)))]),
        D.cert(serial=b"\x02", issuer=n1, exts=[D.BC_NOT_CA, dp_ext(dp(uri(u1)), dp(uri(ldap), uri(u2)))]),
        D.cert(serial=b"\x03", issuer=n2, exts=[dp_ext(dp(D.tlv(0x82, b"dns.example"), uri(b"https://x.example/c")), critical=False)]),
        D.cert(serial=b"\x04", issuer=n1, exts=[dp_ext(dp(reasons=b"\x01\x80"), dp(crl_issuer=D.tlv(0xa4, n1)))]),   # no names
        D.cert(serial=b"\x05", issuer=n1, exts=[D.ext(0x1f, D.seq(D.tlv(0x30, b"\xa0\x05\xa0\x09\x86\x01")))]),      # malformed → none
        D.cert(serial=b"\x06", issuer=n1, exts=[dp_ext(dp(uri(b"http://one.example/"))), dp_ext(dp(uri(b"http://two.example/")))]),  # twice → host
        D.cert(serial=b"\x07", issuer=big, exts=[dp_ext(dp(uri(u1)))]),                                              # 5 KB DN → host
        D.cert(serial=b"\x08", issuer=n1),                                                                           # no extensions
        D.cert(serial=b"\x09", issuer=n1, exts=[dp_ext(dp(uri(b"")))]),                                              # empty URI
        D.cert(serial=b"\x0a", issuer=n1, exts=[dp_ext(dp(uri(b"u" * 300)))]),
        D.cert(serial=b"\x0b", issuer=n2, exts=[dp_ext(dp(*[uri(b"http://many.example/%d" % k) for k in range(5)]))]),  # 5 URIs → host
        D.cert(serial=b"\x0c", issuer=n1, exts=[dp_ext(dp(uri(b"http://a.example/" + b"y" * 5000)))]),                 # 5 KB URI → host
        D.cert(serial=b"\x0d", issuer=n1, exts=[dp_ext(dp(*[uri(b"http://four.example/%d" % k) for k in range(4)]))]),
        # expDate hours outside the (issuer, hour) bitmap's 1970–2089 range take the hash-set path: first and repeat
        D.cert(serial=b"\x0e", issuer=n1, not_after=D.gentime("21500601120000Z")),
        D.cert(serial=b"\x0f", issuer=n1, not_after=D.gentime("21500601123000Z")),
        D.cert(serial=b"\x10", issuer=n1, not_before=D.gentime("19600101000000Z"), not_after=D.gentime("19650601120000Z")),
        D.cert(serial=b"\x11", issuer=n1, not_before=D.gentime("19600101000000Z"), not_after=D.gentime("19650601125959Z")),
    ]
    idx = [0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 1, 0, 0, 0, 0, 0, 0]
    eng = ctmr.Engine(device=0, table_slots=1 << 12, pair_slots=1 << 10, collect_meta=True)
    eng.set_profile("fast")     # certificate 05's malformed distribution point is a parse error under the (default) reference profile
    eng.add_issuers([iss_cert, other])
    eng.set_filter(b"", True, 0)
    res = eng.map_batch(Batch.from_certs(certs, idx))
    assert (res.records["status"] == 0).all() and res.stats.n_new == len(certs)
    exp = expected_first_sightings(certs, idx, list(range(len(certs))), res.records["exp_hour"])
    raw_items = eng.meta_new()
    got = got_first_sightings(eng, raw_items)
    assert got == exp
    assert {(N.MK_HOST, 5), (N.MK_HOST, 6), (N.MK_HOST, 10), (N.MK_HOST, 11)} <= got
    assert (N.MK_CRL, 0, 0, b"http://four.example/3") in got
    assert (N.MK_CRL, 0, 0, u2) in got and (N.MK_CRL, 0, 0, ldap) in got and (N.MK_CRL, 0, 0, b"") in got
    assert not any(k[0] == N.MK_CRL and k[3].startswith(b"\x86") for k in got)
    hours = res.records["exp_hour"]
    assert hours[13] == hours[14] > (1 << 20) and hours[15] == hours[16] < 0
    far = [it for it in raw_items if it[0] == N.MK_EXPDATE and it[3] in (int(hours[13]), int(hours[15]))]
    assert len(far) == 2                                       # ONE first sighting per out-of-range hour
    near = [it for it in raw_items if it[0] == N.MK_EXPDATE and it[3] == int(hours[0])]
    assert len(near) == 2                                      # … and per (issuer, hour) bit: two issuers share hours[0]
    eng.close()


def test_device_variant_and_small_buffer():
    dev = torch.device("cuda:0")
    cfg = synth.config(seed=4, n_issuers=8)
    n = 4000
    b = synth.host_batch(cfg, 0, n)
    eng = ctmr.Engine(device=0, table_slots=1 << 14, pair_slots=1 << 12, collect_meta=True)
    eng.add_issuers(synth.issuers(cfg))
    eng.set_filter(b"", True, NOW)
    d_pay = torch.from_numpy(b.payload.copy()).to(dev)
    d_off = torch.from_numpy(b.offsets.astype(np.int64)).to(dev)
    d_iss = torch.from_numpy(b.issuer_idx.astype(np.int32)).to(dev)
    d_et = torch.from_numpy(b.entry_type.copy()).to(dev)
    d_rec = torch.zeros(n * 32, dtype=torch.uint8, device=dev)
    d_new = torch.zeros(n, dtype=torch.int64, device=dev)
    st = eng.map_batch_device(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), n,
                              d_rec.data_ptr(), d_new.data_ptr())
    d_items = torch.zeros(4 * n * 32, dtype=torch.uint8, device=dev)
    with pytest.raises(ctmr.CtmrError) as ei:                # too small: fails loudly, memo cleared
        eng.meta_new_device(d_pay.data_ptr(), d_off.data_ptr(), 0, d_rec.data_ptr(), d_new.data_ptr(), st.n_new,
                            d_items.data_ptr(), 3)
    assert ei.value.code == N.E_RANGE
    got = eng.meta_new_device(d_pay.data_ptr(), d_off.data_ptr(), 0, d_rec.data_ptr(), d_new.data_ptr(), st.n_new,
                              d_items.data_ptr(), 4 * n)
    items = np.frombuffer(d_items[:got * 32].cpu().numpy().tobytes(),
                          dtype=np.dtype([("entry", "<u8"), ("kind", "<u4"), ("issuer_idx", "<u4"), ("exp_hour", "<i4"),
                                          ("off", "<u4"), ("len", "<u4"), ("pad", "<u4")]))
    assert (items["kind"] == N.MK_DN).sum() == 8 and (items["kind"] == N.MK_CRL).sum() == 8
    for it in items[items["kind"] == N.MK_CRL]:
        der = b.cert(int(it["entry"]))
        assert der[it["off"]:it["off"] + it["len"]] == orc.cert_meta(der)[1][0]
    # without collect_meta the call refuses
    e2 = ctmr.Engine(device=0, table_slots=1 << 10, pair_slots=1 << 10)
    with pytest.raises(ctmr.CtmrError):
        e2.meta_new_device(d_pay.data_ptr(), d_off.data_ptr(), 0, d_rec.data_ptr(), d_new.data_ptr(), 1,
                           d_items.data_ptr(), 10)
    e2.close()
    eng.close()


def test_precheck_against_a_warm_memo():
    """Round 2: with collect_meta the map kernel looks the memo of EARLIER calls up while the bytes are in its LDS window
    and k_meta_new skips every certificate that brings nothing new.  A warm memo, a batch of mostly-seen certificates, and
    among them the ones that must NOT be skipped: a new expDate hour, a second CRL distribution point and another issuer
    Name for a known issuer, two URIs, a repeated extension (host), a Name/URI longer than the fast paths — all reported
    exactly as the reference's memo semantics say, nothing else reported."""
    cfg = synth.config(seed=20260921 + 23, n_issuers=24, dup_permille=50)
    issuers = synth.issuers(cfg)
    eng = ctmr.Engine(device=0, table_slots=1 << 17, pair_slots=1 << 14, collect_meta=True)
    eng.add_issuers(issuers)
    eng.set_filter(b"", True, NOW)
    seen_total = set()

    def run(certs, idx):
        b = Batch.from_certs(certs, idx)
        res = eng.map_batch(b)
        exp = expected_first_sightings(certs, [int(k) for k in idx], [int(i) for i in res.new_idx], res.records["exp_hour"])
        exp -= seen_total
        items = eng.meta_new()
        got = got_first_sightings(eng, items)
        assert len(items) == len(got) and got == exp
        seen_total.update(k for k in exp if k[0] != N.MK_HOST)   # a host-routed certificate is handed over every time
        return exp, res

    b1 = synth.host_batch(cfg, 0, 20000)
    exp1, _ = run([b1.cert(i) for i in range(b1.n)], b1.issuer_idx)
    assert len(exp1) > 24 * 2
    # what issuer 0's leaves look like, to build relatives of them
    i0 = int(np.nonzero(b1.issuer_idx == 0)[0][0])
    name0, uris0, _ = orc.cert_meta(b1.cert(i0))
    assert len(uris0) == 1
    other_name = D.name(D.rdn(10, b"Another Org"), D.rdn(3, b"Same key, other Name"))
    long_name = D.name(D.rdn(10, b"o" * 150), D.rdn(3, b"long"))
    special = [
        D.cert(serial=b"\x51", issuer=name0, exts=[dp_ext(dp(uri(uris0[0])))]),                                  # all seen but the hour
        D.cert(serial=b"\x52", issuer=name0, exts=[dp_ext(dp(uri(uris0[0])))], not_after=D.gentime("20440601120000Z")),
        D.cert(serial=b"\x53", issuer=name0, exts=[dp_ext(dp(uri(b"http://crl.example/second-shard.crl")))]),      # second CRL DP of the issuer
        D.cert(serial=b"\x54", issuer=other_name, exts=[dp_ext(dp(uri(uris0[0])))]),                             # another Name, same issuer
        D.cert(serial=b"\x55", issuer=name0, exts=[dp_ext(dp(uri(uris0[0]), uri(b"ldap://x.example/y")))]),       # two URIs
        D.cert(serial=b"\x56", issuer=name0, exts=[dp_ext(dp(uri(uris0[0]))), dp_ext(dp(uri(uris0[0])))]),        # extension twice → host
        D.cert(serial=b"\x57", issuer=long_name, exts=[dp_ext(dp(uri(b"http://crl.example/" + b"p" * 90)))]),    # beyond the 128 / 64-byte fast paths
        D.cert(serial=b"\x58", issuer=name0),                                                                    # no CRL DP at all
        D.cert(serial=b"\x59", issuer=name0, exts=[dp_ext(dp(uri(b"http://crl.example/second-shard.crl")))]),      # … now seen (same batch)
        D.cert(serial=b"\x5a", issuer=long_name, exts=[dp_ext(dp(uri(b"http://crl.example/" + b"p" * 90)))]),
    ]
    b2 = synth.host_batch(cfg, 20000, 20000)
    certs2 = [b2.cert(i) for i in range(b2.n)]
    idx2 = [int(k) for k in b2.issuer_idx]
    rng = random.Random(5)
    for c in special:                                             # scattered through the batch
        at = rng.randrange(len(certs2))
        certs2.insert(at, c)
        idx2.insert(at, 0)
    exp2, res2 = run(certs2, idx2)
    kinds = sorted(k[0] for k in exp2)
    assert (N.MK_CRL, 0, 0, b"http://crl.example/second-shard.crl") in exp2
    assert (N.MK_DN, 0, 0, other_name) in exp2 and (N.MK_DN, 0, 0, long_name) in exp2
    assert kinds.count(N.MK_HOST) == 1
    # third pass over the same certificates after the table is cleared: every certificate is new again, nothing is a
    # first sighting — the steady state the pre-check exists for
    eng.reset_known()
    exp3, res3 = run(certs2, idx2)
    assert res3.stats.n_new >= res2.stats.n_new > 0
    assert exp3 == {k for k in exp3 if k[0] == N.MK_HOST}         # host-routed certificates are handed over every time
    eng.close()


@pytest.mark.parametrize("profile", ["reference", "fast"])
def test_precheck_when_the_subject_alt_name_precedes_the_distribution_points(profile):
    """Round 6: under the reference profile the subjectAltName is walked by the whole wave through windows of its own, and
    its last round loads only the chunks up to the value's end — what lies behind them in the window is stale.  A
    cRLDistributionPoints extension BEHIND a long subjectAltName lies exactly there: the memo pre-check must not read it
    from the window (WinReaderC::part).  Warm memo, then the same issuer's certificates with SAN lengths that end a round
    anywhere in a window, known and new distribution points: first sightings as the reference's memo semantics say."""
    cfg = synth.config(seed=20260921 + 61, n_issuers=8, dup_permille=0, ca_permille=0, expired_permille=0)
    issuers = synth.issuers(cfg)
    eng = ctmr.Engine(device=0, table_slots=1 << 16, pair_slots=1 << 12, collect_meta=True)
    eng.set_profile(profile)
    eng.add_issuers(issuers)
    eng.set_filter(b"", True, NOW)
    seen_total = set()

    def run(certs, idx):
        res = eng.map_batch(Batch.from_certs(certs, idx))
        assert (res.records["status"] == N.ST_PASS).all()
        exp = expected_first_sightings(certs, [int(k) for k in idx], [int(i) for i in res.new_idx], res.records["exp_hour"])
        exp -= seen_total
        items = eng.meta_new()
        got = got_first_sightings(eng, items)
        assert len(items) == len(got) and got == exp
        seen_total.update(k for k in exp if k[0] != N.MK_HOST)
        return exp

    b1 = synth.host_batch(cfg, 0, 4000)
    run([b1.cert(i) for i in range(b1.n)], b1.issuer_idx)
    i0 = int(np.nonzero(b1.issuer_idx == 0)[0][0])
    name0, uris0, _ = orc.cert_meta(b1.cert(i0))
    known, fresh = uris0[0], b"http://crl.example/behind-the-names.crl"

    def san(n_bytes, rng):
        names, total = [], 0
        while total < n_bytes:
            nm = b"h%d.example.org" % rng.randrange(10 ** rng.randrange(1, 9))
            names.append(D.tlv(0x82, nm))
            total += len(names[-1])
        return D.ext(0x11, D.seq(*names))

    rng = random.Random(61)
    certs = []
    for k in range(1500):
        u = known if k % 3 else fresh
        certs.append(D.cert(serial=b"\x61" + k.to_bytes(3, "big"), issuer=name0, spki=D.rsa_spki(),
                            exts=[D.BC_NOT_CA, san(rng.randrange(40, 900), rng), dp_ext(dp(uri(u)))]))
    exp = run(certs, [0] * len(certs))
    assert (N.MK_CRL, 0, 0, fresh) in exp and (N.MK_CRL, 0, 0, known) not in exp
    eng.reset_known()
    assert not any(k[0] == N.MK_CRL for k in run(certs, [0] * len(certs)))   # steady state: everything seen
    eng.close()
