"""CPU-side checks of the walk: oracle vs OpenSSL (independent extractor) on goldens + synthetic
certs, and the PRODUCT's device walk (host build, tests/harness) vs the oracle on valid, mutated
and truncated inputs.  No GPU needed; the GPU parity tests repeat the fuzz through the C ABI."""
import random

import pytest

from ct_mapreduce_amd import synth
from oracle import oracle as orc
from tests import harness

FIELDS = ("nonfatal", "serial_off", "serial_len", "not_before", "not_after", "cn_off", "cn_len", "bc_valid",
          "is_ca", "spki_off", "spki_len")


def same(der):
    o = orc.parse_cert(der)
    p = harness.product_walk(der, 0xA5)
    p2 = harness.product_walk(der, 0x30)   # different garbage after the cert: must not matter
    assert bool(p.ok) == bool(p2.ok)
    assert bool(o.ok) == bool(p.ok), (o.ok, o.err_site, p.ok)
    if o.ok:
        for f in FIELDS:
            assert getattr(o, f) == getattr(p, f) == getattr(p2, f), f
        s20 = der[o.serial_off:o.serial_off + min(o.serial_len, 20)].ljust(20, b"\0")
        assert b"".join(int(w).to_bytes(4, "little") for w in p.serial_w) == s20
        assert p.cn_match == 1
        # strict_strings: the character sets of the Names' string values (stdlib rules), product vs oracle
        assert harness.product_name_strings(der, 0xA5) == harness.product_name_strings(der, 0x30) == (1 if o.string_findings == 0 else 0)
        meta = orc.cert_meta(der)               # IssuerMetadata's inputs come from the same walk: RawIssuer is the Name it validated
        assert meta is not None and meta[0] == der[o.issuer_off:o.issuer_off + o.issuer_len] and meta[0][:1] == b"\x30"
        cn = der[o.cn_off:o.cn_off + o.cn_len]
        for filt in (b"Synth Issuer 0,Synth Issuer 1", b"c", b"ca", b"cab,x", b"x, ca", b"", b",", cn, cn + b"x", cn[:5] + b",zz"):
            pf = harness.product_walk(der, 0x11, filt)
            want = orc.is_filtered_out(der, o, filt, True, 0) != orc.ST_FILTERED_CN or filt == b""
            if o.bc_valid and o.is_ca:
                want = any(cn.startswith(piece) for piece in filt.split(b","))
            assert bool(pf.cn_match) == want, (filt, cn)
    return bool(o.ok)


def check_against_openssl(der):
    o = orc.parse_cert(der)
    x = harness.ossl_extract(der)
    assert o.ok == 1 and x is not None
    assert o.not_after == x.not_after and o.not_before == x.not_before
    assert bool(o.bc_valid) == bool(x.has_bc) and bool(o.is_ca) == bool(x.is_ca)
    raw = der[o.serial_off:o.serial_off + o.serial_len]
    mag = bytes(x.serial[:x.serial_len])
    assert not x.serial_neg
    assert raw.lstrip(b"\x00") == mag.lstrip(b"\x00")          # OpenSSL normalises; we keep raw
    assert der[o.cn_off:o.cn_off + o.cn_len] == bytes(x.cn[:x.cn_len])
    assert der[o.spki_off:o.spki_off + o.spki_len] == bytes(x.spki[:x.spki_len])


def test_goldens_against_openssl(golden_certs):
    for der in golden_certs.values():
        check_against_openssl(der)
        assert same(der)


def test_synthetic_certs_against_openssl():
    cfg = synth.config(seed=7, n_issuers=256, dup_permille=100, ca_permille=100, expired_permille=100)
    for i in range(400):
        der, iss, et = synth.leaf(cfg, i)
        check_against_openssl(der)
        assert same(der)
    for k in (0, 1, 17, 255):
        der = synth.issuer(cfg, k)
        check_against_openssl(der)
        assert same(der)
        c = orc.parse_cert(der)
        assert c.is_ca == 1


def test_synthetic_length_distribution_and_features():
    cfg = synth.config(seed=11, n_issuers=256, dup_permille=0, ca_permille=10, expired_permille=10)
    lens, cas, exps, lead0 = [], 0, 0, 0
    for i in range(3000):
        der, iss, et = synth.leaf(cfg, i)
        c = orc.parse_cert(der)
        assert c.ok
        lens.append(len(der))
        cas += c.is_ca
        exps += c.not_after < synth.BASE_TIME
        lead0 += der[c.serial_off] == 0
        assert der[c.cn_off:c.cn_off + c.cn_len] == b"Synth Issuer %03d" % iss
    assert 1200 <= min(lens) and max(lens) <= 2010
    assert 1500 < sum(lens) / len(lens) < 1570
    assert 5 <= cas <= 70 and 5 <= exps <= 70
    assert 1200 < lead0 < 1800          # ≈50 % carry the leading 00 (G2 semantics exercised)


def mutate(rng, der):
    b = bytearray(der)
    kind = rng.randrange(6)
    if kind == 0:      # flip a byte in the first 300 (headers live there)
        i = rng.randrange(min(len(b), 300))
        b[i] ^= 1 << rng.randrange(8)
    elif kind == 1:    # flip anywhere
        i = rng.randrange(len(b))
        b[i] = rng.randrange(256)
    elif kind == 2:    # truncate
        b = b[:rng.randrange(len(b))]
    elif kind == 3:    # extend
        b += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 8)))
    elif kind == 4:    # overwrite a short run
        i = rng.randrange(len(b))
        for k in range(i, min(len(b), i + rng.randrange(1, 6))):
            b[k] = rng.choice((0x00, 0x30, 0x80, 0x81, 0x82, 0x83, 0x84, 0xff, 0xa0, 0xa3, 0x02, 0x17, 0x18))
    else:              # two flips
        for _ in range(2):
            i = rng.randrange(len(b))
            b[i] ^= 1 << rng.randrange(8)
    return bytes(b)


def test_product_walk_equals_oracle_on_mutations(golden_certs):
    rng = random.Random(20260921)
    cfg = synth.config(seed=3, n_issuers=16, ca_permille=200, expired_permille=50)
    seeds = list(golden_certs.values()) + [synth.leaf(cfg, i)[0] for i in range(40)]
    accepted = 0
    for r in range(6000):
        der = mutate(rng, seeds[r % len(seeds)])
        accepted += same(der)
    assert 300 < accepted < 5700      # both outcomes are exercised


@pytest.mark.parametrize("t,ok,exp", [
    (b"\x17\x0d" + b"260101000000Z", True, 1767225600),
    (b"\x17\x0b" + b"2601010000Z", True, 1767225600),
    (b"\x18\x0f" + b"20260101000000Z", True, 1767225600),
    (b"\x17\x0d" + b"491231235959Z", True, 2524607999),
    (b"\x17\x0d" + b"500101000000Z", True, -631152000),
    (b"\x18\x0f" + b"19691231235959Z", True, -1),
    (b"\x18\x0f" + b"00010101000000Z", True, -62135596800),
    (b"\x17\x0d" + b"260230000000Z", False, 0),       # Feb 30
    (b"\x17\x0d" + b"240229000000Z", True, 1709164800),
    (b"\x17\x0d" + b"250229000000Z", False, 0),
    (b"\x17\x0d" + b"261301000000Z", False, 0),
    (b"\x17\x0d" + b"260101240000Z", False, 0),
    (b"\x17\x0d" + b"260101006000Z", False, 0),
    (b"\x17\x0d" + b"260101000060Z", False, 0),
    (b"\x17\x0d" + b"26010100000 Z", False, 0),
    (b"\x17\x0d" + b"260101000000+", False, 0),
    (b"\x17\x11" + b"260101000000+0100", True, 1767225600 - 3600),   # numeric zones: Go's "Z0700" layout element
    (b"\x17\x0f" + b"2601010000-0530", True, 1767225600 + 19800),
    (b"\x18\x13" + b"20260101000000-0800", True, 1767225600 + 28800),
    (b"\x17\x11" + b"260101000000+0000", False, 0),   # offset 0 prints back as "Z": rejected
    (b"\x17\x11" + b"260101000000+0060", False, 0),   # prints back as +0100
    (b"\x18\x13" + b"20260101000000+00:3", False, 0),
    (b"\x18\x0d" + b"260101000000Z", False, 0),
    (b"\x16\x0d" + b"260101000000Z", False, 0),
])
def test_time_forms(golden_certs, t, ok, exp):
    """Swap notAfter of the leading-zeroes golden (GeneralizedTime at a fixed offset)."""
    der = golden_certs["kLeadingZeroes"]
    c = orc.parse_cert(der)
    # find notAfter TLV: second time inside validity; golden uses 18 0f … twice
    i = der.index(b"\x18\x0f20200205000000Z")
    old = der[i:i + 17]
    body = der[:i] + t + der[i + 17:]
    delta = len(t) - len(old)
    # patch the three enclosing lengths (validity SEQ short form, TBS and Certificate long form)
    b = bytearray(body)
    vi = der.index(b"\x30\x22\x18\x0f")
    b[vi + 1] = 0x22 + delta
    tbs = int.from_bytes(der[6:8], "big") + delta
    tot = int.from_bytes(der[2:4], "big") + delta
    b[6:8] = tbs.to_bytes(2, "big")
    b[2:4] = tot.to_bytes(2, "big")
    der2 = bytes(b)
    o = orc.parse_cert(der2)
    assert bool(o.ok) == ok
    if ok:
        assert o.not_after == exp
    assert same(der2) == ok


def test_mixed_synthetic_corpus_against_openssl():
    """profile=1: EC P-256 and RSA keys, OV-like subjects, longer issuer names, GeneralizedTime — oracle, product
    walk (host build) and OpenSSL agree, and the corpus really is mixed."""
    cfg = synth.config(seed=21, n_issuers=9, profile=1, ca_permille=30, expired_permille=30)
    n_ec = n_gt = n_ov = 0
    for i in range(600):
        der, iss, et = synth.leaf(cfg, i)
        assert same(der)
        c = orc.parse_cert(der)
        o = harness.ossl_extract(der)
        assert c.ok and o is not None
        assert (c.not_before, c.not_after) == (o.not_before, o.not_after)
        assert der[c.cn_off:c.cn_off + c.cn_len] == bytes(o.cn[:o.cn_len]) == b"Synth Issuer %03d" % iss
        assert der[c.spki_off:c.spki_off + c.spki_len] == bytes(o.spki[:o.spki_len])
        n_ec += c.spki_len == 91
        n_gt += b"\x18\x0f20" in der
        n_ov += b"San Francisco" in der
    assert n_ec > 200 and n_gt > 80 and n_ov > 150
    for k in range(9):
        assert same(synth.issuer(cfg, k))


def edge_seeds():
    """Small hand-built certificates that sit on the Go-specific rules (numeric zones, lax INTEGERs, unique ids,
    high tag numbers, odd basicConstraints, explicit wrappers that lie about their length): mutations of these land on
    headers far more often than mutations of a 1.5 KB certificate do."""
    from tests import der as D
    # (an algorithm parsePublicKey does not know: this seed is about the STRUCTURE around the key; tests/test_spki_cpu.py
    #  holds the seeds whose keys are parsed)
    ec = D.seq(D.oid(0x2a, 0x86, 0x48, 0xce, 0x3d, 2, 2), D.oid(0x2a, 0x86, 0x48, 0xce, 0x3d, 3, 1, 7))
    hi = D.seq(D.tlv(0x31, D.seq(D.oid(0x55, 4, 3), b"\x5f\x28\x01x")), D.rdn(3, b"cn"), D.rdn(10, b"o", tag=0x13))
    return [
        D.cert(exts=[D.BC_NOT_CA]),
        D.cert(not_after=D.utctime("270101010000+0100"), not_before=D.gentime("20250101000000-0800"), exts=[D.BC_CA]),
        D.cert(not_after=D.utctime("2701010100+0100"), serial=b"\x00\x7f", version=D.tlv(0xa0, D.tlv(0x02, b"\x00\x02"))),
        D.cert(serial=b"\xff\x80", issuer=hi, subject=hi, exts=[D.ext(0x13, D.seq(D.tlv(0x02, b"\x00\x05")))]),
        D.cert(extra_tbs=D.tlv(0x81, b"\x01\x02") + D.tlv(0x82, b"\x00\x02") + D.tlv(0xa3, D.seq(D.BC_CA) + b"\xff")),
        D.cert(extra_tbs=b"\xa3\x01" + D.seq(D.ext(0x13, D.seq(D.tlv(0x01, b"\xff"), D.tlv(0x05, b""), b"\x05\x7f")))),
        D.cert(version=b"\xa0\x05" + D.tlv(0x02, b"\x02"), spki=D.seq(ec, D.tlv(0x03, b"\x03" + bytes(31) + b"\xf8"), D.tlv(0x05, b"")),
               tbs_sigalg=D.seq(D.oid(0x2a, 3)), outer_sigalg=D.seq(D.oid(0x2a, 3), D.tlv(0x30, b""), D.tlv(0x05, b""))),
        D.cert(extra_tbs=D.tlv(0x1f, b"")[:1] + b"\x28\x00" + b"\x05\x00", version=False, sig=b"\x07\x80"),
        D.cert(exts=[D.seq(D.oid(0x2b, 6, 1, 4, 1, 0x82, 0x37), D.tlv(0x01, b"\xff"), D.tlv(0x04, b"abc")), D.BC_CA, D.BC_NOT_CA],
               issuer=D.seq(D.tlv(0x31, D.seq(D.oid(0x55, 4, 3), D.tlv(0x13, b"a")) + D.seq(D.oid(0x55, 4, 3), D.tlv(0x16, b"b"), D.tlv(0x05, b"")))),
               subject=D.seq(D.tlv(0x31, b""))),
    ]


def test_product_walk_equals_oracle_on_mutated_edge_certificates():
    rng = random.Random(20260923)
    seeds = edge_seeds()
    accepted = 0
    for der in seeds:
        assert same(der), der.hex()
    for r in range(12000):
        der = mutate(rng, seeds[r % len(seeds)])
        if rng.randrange(3) == 0 and len(der) > 1:
            der = mutate(rng, der)
        accepted += same(der)
    assert 1000 < accepted < 11000


# ---- strict_leaf: the bare TBSCertificate of a precertificate entry's MerkleTreeLeaf (round 3) -----------------------
def tbs_of(der):
    """The TBSCertificate TLV of a well-formed certificate (tests only: two definite-length headers)."""
    def hdr(p):
        ln = der[p + 1]
        if ln < 0x80:
            return p + 2, p + 2 + ln
        k = ln & 0x7f
        return p + 2 + k, p + 2 + k + int.from_bytes(der[p + 2:p + 2 + k], "big")
    c0, _ = hdr(0)
    _, t1 = hdr(c0)
    return der[c0:t1]


def same_tbs(tbs):
    o = orc.parse_tbs(tbs)
    p = harness.product_walk_tbs(tbs, 0xA5)
    p2 = harness.product_walk_tbs(tbs, 0x30)
    assert bool(o.ok) == bool(p.ok) == bool(p2.ok), (o.ok, o.err_site, p.ok, p2.ok)
    if o.ok:
        for f in FIELDS:
            assert getattr(o, f) == getattr(p, f) == getattr(p2, f), f
    return bool(o.ok)


def test_tbs_walk_is_the_certificate_walk_without_wrapper_and_signature(golden_certs):
    cfg = synth.config(seed=5, n_issuers=8, profile=1)
    for der in list(golden_certs.values()) + [synth.leaf(cfg, i)[0] for i in range(60)] + edge_seeds():
        whole = orc.parse_cert(der)
        if not whole.ok:
            continue
        tbs = tbs_of(der)
        assert same_tbs(tbs)
        t = orc.parse_tbs(tbs)
        off = der.index(tbs)                      # every position moves by the outer header
        assert (t.serial_off + off, t.serial_len, t.not_after, t.bc_valid, t.is_ca) == \
               (whole.serial_off, whole.serial_len, whole.not_after, whole.bc_valid, whole.is_ca)
        assert not same_tbs(tbs + b"\x00")        # x509.ParseTBSCertificate: trailing data
        assert not same_tbs(tbs[:-1]) and not same_tbs(der)     # a whole certificate is not a TBSCertificate


def test_product_tbs_walk_equals_oracle_on_mutations(golden_certs):
    rng = random.Random(20260924)
    cfg = synth.config(seed=6, n_issuers=16, ca_permille=200, expired_permille=50)
    seeds = [tbs_of(d) for d in list(golden_certs.values()) + [synth.leaf(cfg, i)[0] for i in range(40)] + edge_seeds()
             if orc.parse_cert(d).ok]
    accepted = 0
    for r in range(8000):
        accepted += same_tbs(mutate(rng, seeds[r % len(seeds)]))
    assert 400 < accepted < 7600


def test_the_walk_behind_register_held_outer_headers_decides_like_the_plain_walk(golden_certs):
    """Round 6: the fast kernels' first window begins BEHIND the Certificate and TBSCertificate headers, which the walk
    reads from sixteen octets held in registers (kernels/readers.h WinGeo::SKIP, der_walk.h HeadView).  The device's
    bookkeeping simulated on the host (harness.walk_window, skip = 8): whenever no read left the window and the head view
    did not hand the certificate over (on the GPU both mean: the exact reader repeats it), the verdict is the plain
    walk's — for damage anywhere, and for damage aimed at the first sixteen octets (high tag numbers, lengths of one to
    four octets, headers that overrun)."""
    rng = random.Random(20260930)
    cfg = synth.config(seed=9, n_issuers=16, ca_permille=100, expired_permille=50)
    seeds = list(golden_certs.values()) + [synth.leaf(cfg, i)[0] for i in range(30)]
    decided = handed_over = accepted = 0
    for r in range(6000):
        der = seeds[r % len(seeds)]
        if r % 3 == 0:
            der = mutate(rng, der)
        else:                                # the outer headers
            b = bytearray(der)
            for _ in range(rng.randrange(1, 4)):
                b[rng.randrange(min(16, len(b)))] = rng.choice((0x30, 0x1f, 0x3f, 0x80, 0x81, 0x82, 0x83, 0x84, 0x00, 0xff, 0xa0,
                                                                rng.randrange(256)))
            der = bytes(b)
        phase = rng.randrange(128)
        ok, refills, misses, _, _, coop, deferred = harness.walk_window(der, phase, 216, False, False, skip=8)
        if misses or deferred:
            handed_over += 1
            continue
        decided += 1
        assert ok == bool(harness.product_walk(der).ok), (r, der[:24].hex())
        accepted += ok
    assert decided > 1500 and handed_over > 500 and accepted > 200
