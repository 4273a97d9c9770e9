"""Hand-built edge-case certificates: oracle vs the product walk (host build) — CPU only."""
from oracle import oracle as orc
from tests import der as D
from tests import harness


def both(c):
    o = orc.parse_cert(c)
    p = harness.product_walk(c)
    assert bool(o.ok) == bool(p.ok), (o.ok, o.err_site, p.ok)
    if o.ok:
        for f in ("serial_off", "serial_len", "not_before", "not_after", "cn_off", "cn_len",
                  "bc_valid", "is_ca", "spki_off", "spki_len"):
            assert getattr(o, f) == getattr(p, f), f
    return o


def cn_of(c, o):
    return c[o.cn_off:o.cn_off + o.cn_len]


def test_basic_and_openssl_agree():
    c = D.cert(exts=[D.BC_NOT_CA])
    o = both(c)
    assert o.ok and o.bc_valid and not o.is_ca and cn_of(c, o) == b"Test CA"
    x = harness.ossl_extract(c)
    assert x is not None and x.not_after == o.not_after


def test_last_cn_wins_and_non_string_cn_skipped():
    iss = D.name(D.rdn(3, b"first"), D.rdn(10, b"org"), D.rdn(3, b"second"))
    c = D.cert(issuer=iss)
    assert cn_of(c, both(c)) == b"second"
    iss = D.name(D.rdn(3, b"first"), D.rdn(3, b"\x00s\x00e", tag=0x1e))   # BMPString: not a Go string here
    c = D.cert(issuer=iss)
    assert cn_of(c, both(c)) == b"first"
    # multi-valued RDN
    iss = D.seq(D.tlv(0x31, D.seq(D.oid(0x55, 4, 3), D.tlv(0x13, b"a")) + D.seq(D.oid(0x55, 4, 3), D.tlv(0x16, b"b"))))
    c = D.cert(issuer=iss)
    assert cn_of(c, both(c)) == b"b"
    # no CN at all
    c = D.cert(issuer=D.name(D.rdn(10, b"org only")))
    assert both(c).cn_len == 0
    # empty issuer
    c = D.cert(issuer=D.seq())
    assert both(c).ok and both(c).cn_len == 0


def test_serial_forms():
    for s, ok in ((b"\x00\xaa", True), (b"\x00\x7f", False), (b"\xff\x80", False), (b"\xff\x7f", True),
                  (b"\x80", True), (b"", False), (b"\x00", True), (bytes(range(1, 21)), True),
                  (bytes(range(1, 31)), True), (bytes(range(1, 41)), True), (bytes(range(1, 46)), True),
                  (b"\x01" * 200, True)):
        c = D.cert(serial=s)
        o = both(c)
        assert bool(o.ok) == ok, s
        if ok:
            assert c[o.serial_off:o.serial_off + o.serial_len] == s


def test_versionless_and_unique_ids_and_no_extensions():
    assert both(D.cert(version=False)).ok
    c = D.cert(extra_tbs=b"")  # no [3]
    assert both(c).bc_valid == 0
    # issuerUniqueID / subjectUniqueID then extensions: built by hand in extra_tbs
    c = D.cert(extra_tbs=D.tlv(0x81, b"\x00\x01") + D.tlv(0x82, b"\x00\x02") + D.tlv(0xa3, D.seq(D.BC_CA)))
    o = both(c)
    assert o.ok and o.is_ca


def test_basic_constraints_variants():
    def bc(value, critical=True):
        return D.cert(exts=[D.ext(0x0f, D.tlv(0x03, b"\x05\xa0")), D.ext(0x13, value, critical)])
    assert both(bc(D.seq())).is_ca == 0
    assert both(bc(D.seq(D.tlv(0x01, b"\xff")))).is_ca == 1
    assert both(bc(D.seq(D.tlv(0x01, b"\x00")))).is_ca == 0
    assert both(bc(D.seq(D.tlv(0x01, b"\x01")))).ok == 0            # invalid DER boolean
    assert both(bc(D.seq(D.tlv(0x01, b"\xff"), D.tlv(0x02, b"\x00")))).is_ca == 1
    assert both(bc(D.seq(D.tlv(0x02, b"\x03")))).is_ca == 0          # pathLen only
    assert both(bc(D.seq(D.tlv(0x01, b"\xff")) + b"\x05\x00")).ok == 0   # trailing data in OCTET STRING
    assert both(bc(D.seq(D.tlv(0x01, b"\xff")), critical=None)).is_ca == 1
    # repeated extension: last wins
    c = D.cert(exts=[D.BC_CA, D.BC_NOT_CA])
    assert both(c).is_ca == 0
    c = D.cert(exts=[D.BC_NOT_CA, D.BC_CA])
    assert both(c).is_ca == 1
    # critical flag with a bad boolean, and a non-OCTET value
    bad = D.seq(D.oid(0x55, 0x1d, 0x13), D.tlv(0x01, b"\x02"), D.tlv(0x04, D.seq()))
    assert both(D.cert(exts=[bad])).ok == 0
    bad = D.seq(D.oid(0x55, 0x1d, 0x13), D.tlv(0x05, b""))
    assert both(D.cert(exts=[bad])).ok == 0


def test_length_encodings():
    c = D.cert(exts=[D.ext(0x11, D.seq(D.tlv(0x82, b"a" * 70000)))])     # 3-byte long form
    o = both(c)
    assert o.ok and len(c) > 70000
    # non-minimal long form on the outer header
    good = D.cert()
    assert good[1] == 0x82
    bad = good[:1] + b"\x83\x00" + good[2:4] + good[4:]
    assert both(bad).ok == 0
    # indefinite length
    bad = good[:1] + b"\x80" + good[4:]
    assert both(bad).ok == 0
    # outer length one short / one long
    n = int.from_bytes(good[2:4], "big")
    assert both(good[:2] + (n - 1).to_bytes(2, "big") + good[4:]).ok == 0
    assert both(good[:2] + (n + 1).to_bytes(2, "big") + good[4:]).ok == 0
    assert both(b"").ok == 0 and both(b"\x30").ok == 0 and both(b"\x30\x00").ok == 0


def test_signature_bit_string_rules():
    assert both(D.cert(sig=b"")).ok == 0
    assert both(D.cert(sig=b"\x00")).ok == 1
    assert both(D.cert(sig=b"\x01")).ok == 0
    assert both(D.cert(sig=b"\x08\xff")).ok == 0
    assert both(D.cert(sig=b"\x03\xf8")).ok == 1
    assert both(D.cert(sig=b"\x03\xfc")).ok == 0
