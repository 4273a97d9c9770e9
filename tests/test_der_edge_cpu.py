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
                  "bc_valid", "is_ca", "spki_off", "spki_len", "nonfatal"):
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
    # raw content octets verbatim (types.go:165-178).  Empty: fatal.  Not minimally encoded: only CT-go's lax re-parse
    # accepts it → a non-fatal finding; negative: "x509: negative serial number", non-fatal in CT-go.
    LAX, NEG = orc.NF_LAX_INTEGER, orc.NF_NEGATIVE_SERIAL
    for s, ok, nf in ((b"\x00\xaa", True, 0), (b"\x00\x7f", True, LAX), (b"\xff\x80", True, LAX | NEG),
                      (b"\xff\x7f", True, NEG), (b"\x80", True, NEG), (b"", False, 0), (b"\x00", True, 0),
                      (bytes(range(1, 21)), True, 0), (bytes(range(1, 31)), True, 0), (bytes(range(1, 41)), True, 0),
                      (bytes(range(1, 46)), True, 0), (b"\x01" * 200, True, 0)):
        c = D.cert(serial=s)
        o = both(c)
        assert bool(o.ok) == ok, s
        if ok:
            assert c[o.serial_off:o.serial_off + o.serial_len] == s and o.nonfatal == nf, s


def test_non_fatal_findings_depend_on_the_entry_type():
    # ct-fetch.go:452-459 (X509 entry: dropped only on x509.IsFatal) vs :202-209 (precertificate: any err) vs
    # :221-225 (Chain[0]: any err)
    issuer = D.cert(exts=[D.BC_CA])
    for serial in (b"\x00\x7f", b"\x80\x01"):
        leaf = D.cert(serial=serial, exts=[D.BC_NOT_CA])
        e = orc.Engine(b"", True, 0)
        assert e.entry(leaf, issuer, 0)[0] == orc.ST_PASS
        assert e.entry(leaf, issuer, 1)[0] == orc.ST_PARSE_ERROR
        assert e.entry(D.cert(exts=[D.BC_NOT_CA]), D.cert(serial=serial, exts=[D.BC_CA]), 0)[0] == orc.ST_ISSUER_PARSE_ERROR
        assert e.entry(D.cert(serial=b"\x05", exts=[D.BC_NOT_CA]), issuer, 1)[0] == orc.ST_PASS


def test_time_zone_offsets():
    # Go asn1 parseUTCTime / parseGeneralizedTime: time.Parse with "Z0700", then the time must print back identically
    def t(nb=None, na=None):
        return both(D.cert(not_before=nb, not_after=na))
    base = t(na=D.utctime("270101000000Z")).not_after
    assert t(na=D.utctime("270101010000+0100")).not_after == base
    assert t(na=D.utctime("261231230000-0100")).not_after == base
    assert t(na=D.utctime("2701010100+0100")).not_after == base            # layout without seconds
    assert t(na=D.utctime("2701010530+0530")).not_after == base
    assert t(na=D.gentime("20270101010000+0100")).not_after == base
    assert t(na=D.gentime("20261231183000-0530")).not_after == base
    assert t(na=D.utctime("270101000000+9900")).not_after == base - 99 * 3600   # hh is not range-checked (Go 1.13)
    for bad in ("270101000000+0000", "270101000000-0000", "270101000000+0060", "270101000000+01", "270101000000+010",
                "270101000000+01000", "270101000000 0100", "270101000000+0a00", "270101000000z", "2701010000Z0",
                "27010100000Z", "270101000000.5Z", "270101000000+01:00"):
        assert not t(na=D.utctime(bad)).ok, bad
    for bad in ("20270101000000+0000", "202701010000Z", "202701010000+0100", "20270101000000+0160", "20270101000000",
                "20270101000000.0Z"):
        assert not t(na=D.gentime(bad)).ok, bad
    # the year of a UTCTime is read in its own zone: 49 → 2049, 50 → 1950, whatever the offset moves the instant to
    assert t(na=D.utctime("491231235959-0100")).not_after == both(D.cert(not_after=D.gentime("20500101005959Z"))).not_after
    assert t(nb=D.utctime("500101000000+0100")).not_before == both(D.cert(not_before=D.gentime("19491231230000Z"))).not_before
    # notBefore goes through the same parser
    assert not t(nb=D.utctime("250101000000+0000")).ok


def test_subject_name_structure_is_validated():
    ok = D.name(D.rdn(3, b"leaf"), D.tlv(0x31, D.seq(D.oid(0x55, 4, 10), D.tlv(0x13, b"o")) + D.seq(D.oid(0x2a, 3), D.tlv(0x02, b"\x01"))))
    assert both(D.cert(subject=ok)).ok
    assert both(D.cert(subject=D.seq())).ok                                          # empty RDNSequence
    bad = [D.tlv(0x31, b""),                                                        # Name is not a SEQUENCE
           D.seq(D.seq(D.seq(D.oid(0x55, 4, 3), D.tlv(0x0c, b"x")))),              # RDN is not a SET
           D.seq(D.tlv(0x31, D.tlv(0x31, D.oid(0x55, 4, 3) + D.tlv(0x0c, b"x")))),   # AttributeTypeAndValue is not a SEQUENCE
           D.seq(D.tlv(0x31, D.seq(D.tlv(0x0c, b"x"), D.tlv(0x0c, b"x")))),         # type is not an OID
           D.seq(D.tlv(0x31, D.seq(D.oid(), D.tlv(0x0c, b"x")))),                   # zero length OBJECT IDENTIFIER
           D.seq(D.tlv(0x31, D.seq(D.oid(0x55, 4, 0x83), D.tlv(0x0c, b"x")))),      # truncated base 128 integer
           D.seq(D.tlv(0x31, D.seq(D.oid(0x55, 4, 3)))),                            # no value
           D.seq(D.tlv(0x31, D.seq(D.oid(0x55, 4, 3), b"\x0c\x05x"))),             # value longer than the attribute
           D.seq(D.tlv(0x31, D.seq(D.oid(0x55, 4, 3), D.tlv(0x0c, b"x"))) + b"\x31\x05")]  # RDN longer than the Name
    for sub in bad:
        assert not both(D.cert(subject=sub)).ok, sub.hex()
        assert not both(D.cert(issuer=sub)).ok, sub.hex()
    # bytes behind the value inside an AttributeTypeAndValue are ignored; an empty SET is an empty RDN
    assert both(D.cert(subject=D.seq(D.tlv(0x31, D.seq(D.oid(0x55, 4, 3), D.tlv(0x0c, b"x"), D.tlv(0x05, b"")))))).ok
    assert both(D.cert(subject=D.seq(D.tlv(0x31, b"")))).ok
    # a value with a high tag number is an ANY like every other: [APPLICATION 40] = 0x5f 0x28
    hi = D.seq(D.tlv(0x31, D.seq(D.oid(0x55, 4, 3), b"\x5f\x28\x01x")))
    assert both(D.cert(subject=hi)).ok and both(D.cert(issuer=hi)).cn_len == 0
    for bad_tag in (b"\x5f\x1e\x01x", b"\x5f\x80\x28\x01x", b"\x5f\x88\x80\x80\x80\x00\x01x", b"\x5f\xa8"):
        assert not both(D.cert(subject=D.seq(D.tlv(0x31, D.seq(D.oid(0x55, 4, 3), bad_tag))))).ok, bad_tag
    assert both(D.cert(subject=D.seq(D.tlv(0x31, D.seq(D.oid(0x55, 4, 3), b"\x5f\x87\xff\xff\xff\x7f\x01x"))))).ok  # tag 2^31-1


def test_version_wrapper_and_int_rules():
    def with_version(v):
        return D.cert(version=v)
    assert both(with_version(D.tlv(0xa0, D.tlv(0x02, b"\x02")))).ok
    assert not both(with_version(D.tlv(0xa0, b""))).ok                               # zero length explicit tag
    assert not both(with_version(D.tlv(0xa0, D.tlv(0x04, b"\x02")))).ok              # not an INTEGER
    assert not both(with_version(D.tlv(0xa0, D.tlv(0x02, b"")))).ok
    assert not both(with_version(D.tlv(0xa0, D.tlv(0x02, b"\x01\x00\x00\x00\x00")))).ok   # does not fit int32
    assert both(with_version(D.tlv(0xa0, D.tlv(0x02, b"\x7f\xff\xff\xff")))).ok
    o = both(with_version(D.tlv(0xa0, D.tlv(0x02, b"\x00\x00\x00\x00\x02"))))       # lax: decoded, fits
    assert o.ok and o.nonfatal == orc.NF_LAX_INTEGER
    assert not both(with_version(D.tlv(0xa0, D.tlv(0x02, b"\x00" * 8 + b"\x02")))).ok   # more than 8 octets
    assert not both(with_version(D.tlv(0xa0, D.tlv(0x02, b"\x00\x80\x00\x00\x00\x00")))).ok   # lax, but 2^39
    # Go resumes behind the INNER integer: a wrapper that claims more than its element swallows nothing …
    good = both(D.cert())
    c = with_version(b"\xa0\x05" + D.tlv(0x02, b"\x02"))     # wrapper says 5, element is 3: serial is read right behind the element
    o = both(c)
    assert o.ok and o.serial_len == good.serial_len
    # … and one that claims less does not cut its element short
    assert both(with_version(b"\xa0\x01" + D.tlv(0x02, b"\x02"))).ok


def test_algorithm_identifiers_and_spki_structure():
    def spki(alg, key=D.tlv(0x03, b"\x00" + bytes(65))):
        return D.seq(alg, key)
    # the STRUCTURE of publicKeyInfo, under an algorithm parsePublicKey does not know (1.2.840.10045.2.2: the key bits
    # are not looked at; tests/test_spki_cpu.py has the keys of the algorithms it does know)
    ec = D.seq(D.oid(0x2a, 0x86, 0x48, 0xce, 0x3d, 2, 2), D.oid(0x2a, 0x86, 0x48, 0xce, 0x3d, 3, 1, 7))
    assert both(D.cert(spki=spki(ec))).ok
    assert both(D.cert(spki=spki(D.seq(D.oid(0x2a, 3))))).ok                          # parameters are optional
    assert both(D.cert(spki=spki(D.seq(D.oid(0x2a, 3), D.tlv(0x05, b""), D.tlv(0x05, b""))))).ok   # extra elements ignored
    assert both(D.cert(spki=D.seq(ec, D.tlv(0x03, b"\x00" + bytes(65)), D.tlv(0x05, b"")))).ok
    for bad in (spki(D.seq()), spki(D.seq(D.tlv(0x05, b""))), spki(D.seq(D.oid())), spki(D.seq(D.oid(0x2a, 0x86))),
                spki(D.seq(D.oid(0x2a, 3), b"\x05\x01")), spki(D.tlv(0x31, D.oid(0x2a, 3))), spki(ec, D.tlv(0x04, b"\x00k")),
                spki(ec, D.tlv(0x03, b"")), spki(ec, D.tlv(0x03, b"\x08k")), spki(ec, D.tlv(0x03, b"\x01")),
                spki(ec, D.tlv(0x03, b"\x01" + bytes(64) + b"\x01")), D.seq(ec), D.seq()):
        assert not both(D.cert(spki=bad)).ok, bad.hex()
    assert both(D.cert(spki=spki(ec, D.tlv(0x03, b"\x01" + bytes(64) + b"\x02")))).ok   # pad bits zero

    def with_sigalgs(tbs_alg=D.SIGALG, outer_alg=D.SIGALG):
        return D.cert(tbs_sigalg=tbs_alg, outer_sigalg=outer_alg)
    assert both(with_sigalgs()).ok
    for bad in (D.seq(), D.seq(D.tlv(0x05, b"")), D.seq(D.oid(0x2a, 0x86)), D.tlv(0x31, D.oid(0x2a, 3)),
                D.seq(D.oid(0x2a, 3), b"\x05\x02\x00")):
        assert not both(with_sigalgs(tbs_alg=bad)).ok, bad.hex()
        assert not both(with_sigalgs(outer_alg=bad)).ok, bad.hex()
    assert both(with_sigalgs(tbs_alg=D.seq(D.oid(0x2a, 3)), outer_alg=D.seq(D.oid(0x2a, 3), D.tlv(0x30, b""), D.tlv(0x05, b"")))).ok


def test_versionless_and_unique_ids_and_no_extensions():
    assert both(D.cert(version=False)).ok
    c = D.cert(extra_tbs=b"")  # no [3]
    assert both(c).bc_valid == 0
    # issuerUniqueID / subjectUniqueID then extensions: built by hand in extra_tbs
    c = D.cert(extra_tbs=D.tlv(0x81, b"\x00\x01") + D.tlv(0x82, b"\x00\x02") + D.tlv(0xa3, D.seq(D.BC_CA)))
    o = both(c)
    assert o.ok and o.is_ca
    # they are BIT STRINGs (asn1.BitString `optional,tag:1|2`)
    for bad in (D.tlv(0x81, b""), D.tlv(0x81, b"\x08\x00"), D.tlv(0x82, b"\x01"), D.tlv(0x82, b"\x01\x01"), b"\x81\x7f\x00"):
        assert not both(D.cert(extra_tbs=bad)).ok, bad.hex()
    assert both(D.cert(extra_tbs=D.tlv(0x82, b"\x01\x02"))).ok
    # optional fields with another tag are skipped and the rest of the TBSCertificate is ignored — but the header at
    # that position must parse
    assert both(D.cert(extra_tbs=D.tlv(0x05, b""))).ok
    assert both(D.cert(extra_tbs=D.tlv(0x82, b"\x00") + D.tlv(0x81, b"junk that is never parsed"))).ok
    assert both(D.cert(extra_tbs=D.tlv(0xa1, b"\x08"))).ok                          # constructed [1]: not the BIT STRING
    assert both(D.cert(extra_tbs=b"\x05\x7f")).ok                                  # header parses; length is not checked
    for bad in (b"\x05", b"\x05\x80", b"\x05\x81\x01", b"\x05\x82\x00\x80", b"\x1f", b"\x1f\x1e\x00"):
        assert not both(D.cert(extra_tbs=bad)).ok, bad.hex()
    # [3]: empty is an error, an inner element that is not a SEQUENCE means "no extensions", and parsing never looks
    # at the wrapper's own length
    assert not both(D.cert(extra_tbs=D.tlv(0xa3, b""))).ok
    assert not both(D.cert(extra_tbs=b"\x83\x00")).ok
    assert both(D.cert(extra_tbs=D.tlv(0x83, b"\x01"))).ok
    o = both(D.cert(extra_tbs=D.tlv(0xa3, D.tlv(0x31, D.BC_CA))))
    assert o.ok and not o.bc_valid
    o = both(D.cert(extra_tbs=b"\xa3\x01" + D.seq(D.BC_CA)))                       # wrapper claims 1 byte
    assert o.ok and o.is_ca
    o = both(D.cert(extra_tbs=D.tlv(0xa3, D.seq(D.BC_CA) + b"\xff\xff")))           # junk behind the inner SEQUENCE
    assert o.ok and o.is_ca
    assert not both(D.cert(extra_tbs=D.tlv(0xa3, b"\x30\x7f"))).ok                 # inner SEQUENCE longer than the TBS


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
    # Go's struct { IsCA bool `optional`; MaxPathLen int `optional,default:-1` }: an element of another type leaves
    # the optional field unset, and whatever follows the last field is ignored
    o = both(bc(D.seq(D.tlv(0x01, b"\xff"), D.tlv(0x05, b""))))
    assert o.ok and o.is_ca == 1
    o = both(bc(D.seq(D.tlv(0x04, b"x"), D.tlv(0x01, b"\xff"))))     # cA is not first: never seen
    assert o.ok and o.bc_valid and o.is_ca == 0
    o = both(bc(D.seq(D.tlv(0x01, b"\xff"), D.tlv(0x02, b"\x00"), D.tlv(0x02, b""), b"\xff")))
    assert o.ok and o.is_ca == 1
    assert both(bc(D.seq(D.tlv(0x02, b"\x01"), D.tlv(0x01, b"\x07")))).ok          # a BOOLEAN behind pathLen is never parsed
    assert not both(bc(D.seq(D.tlv(0x01, b"\xff"), b"\x05"))).ok                   # … but the next header must parse
    assert both(bc(D.seq(D.tlv(0x01, b"\xff"), b"\x05\x7f"))).ok                  # (its length is not checked)
    assert not both(bc(D.seq(D.tlv(0x01, b"\xff"), b"\x02\x7f"))).ok              # a matching INTEGER must fit
    assert not both(bc(D.seq(D.tlv(0x02, b"")))).ok
    assert not both(bc(D.seq(D.tlv(0x02, b"\x01\x00\x00\x00\x00")))).ok           # pathLen does not fit int32
    o = both(bc(D.seq(D.tlv(0x02, b"\x00\x05"))))                                 # lax INTEGER
    assert o.ok and o.nonfatal == orc.NF_LAX_INTEGER
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
    # extnID must be a well-formed OID; bytes behind extnValue are ignored
    assert not both(D.cert(exts=[D.seq(D.oid(0x55, 0x1d, 0x93), D.tlv(0x04, b""))])).ok
    assert not both(D.cert(exts=[D.seq(D.oid(), D.tlv(0x04, b""))])).ok
    assert not both(D.cert(exts=[D.seq(D.oid(0x2b, 6, 1, 4, 1, 0x82), D.tlv(0x04, b""))])).ok
    assert both(D.cert(exts=[D.seq(D.oid(0x2b, 6, 1, 4, 1, 0x82, 0x37), D.tlv(0x04, b""), D.tlv(0x05, b""))])).ok


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


def _attr(oid_tlv, value_tlv):
    return D.seq(D.tlv(0x31, D.seq(oid_tlv, value_tlv)))


def test_object_identifier_arcs_follow_parse_base128_int():
    """parseObjectIdentifier / parseBase128Int: an arc is at most 5 octets, at most 2^31 - 1 (the first arc pair
    included), complete, and not led by 0x80 ("integer is not minimally encoded")."""
    val = D.tlv(0x0c, b"x")
    good = [D.oid(0x55, 4, 3),
            D.oid(0x2a, 0x87, 0xff, 0xff, 0xff, 0x7f),         # arc 2^31 - 1
            D.oid(0x87, 0xff, 0xff, 0xff, 0x7f),               # the first octets may be a long arc too (2.x)
            D.oid(0x2a, 0x03, 0x04, 0x05, 0x06, 0x07, 0x08, 0x09, 0x0a, 0x0b, 0x0c, 0x0d, 0x0e, 0x0f, 0x10)]
    bad = [D.oid(0x2a, 0x88, 0x80, 0x80, 0x80, 0x00),          # 2^31: "base 128 integer too large"
           D.oid(0x2a, 0x8f, 0xff, 0xff, 0xff, 0x7f),          # 5 octets, above int32
           D.oid(0x2a, 0x81, 0x80, 0x80, 0x80, 0x80, 0x00),    # 6 octets
           D.oid(0x88, 0x80, 0x80, 0x80, 0x00),                # the same in first position
           D.oid(0x2a, 0x03, 0x81),                            # truncated
           D.oid(0x2a, 0x80, 0x01),                            # padded arc: not minimally encoded
           D.oid(0x80, 0x2a),                                  # … in first position
           D.oid()]                                            # empty
    for o in good:
        for where in ("subject", "issuer"):
            assert both(D.cert(**{where: _attr(o, val)})).ok, (o.hex(), where)
        assert both(D.cert(exts=[D.seq(o, D.tlv(0x04, b""))])).ok, o.hex()        # extnID
    for o in bad:
        for where in ("subject", "issuer"):
            assert not both(D.cert(**{where: _attr(o, val)})).ok, (o.hex(), where)
        assert not both(D.cert(exts=[D.seq(o, D.tlv(0x04, b""))])).ok, o.hex()
    # AlgorithmIdentifier OIDs obey the same rule (tbs signature and the SPKI algorithm)
    assert both(D.cert(spki=D.spki(bytes.fromhex("2a87ffffff7f"), D.NULL, b"\x01\x02"))).ok
    assert not both(D.cert(spki=D.spki(bytes.fromhex("2a8880808000"), D.NULL, b"\x01\x02"))).ok


def test_name_values_of_the_types_go_decodes():
    """An AttributeTypeAndValue's Value is an interface{}: INTEGER, BIT STRING, OBJECT IDENTIFIER and the two time types
    are decoded by their tag and a malformed one fails the Name; every other type is taken as it is."""
    t = D.oid(0x55, 4, 5)
    good = [D.tlv(0x02, b"\x01"), D.tlv(0x02, b"\x7f" + b"\xff" * 7), D.tlv(0x02, b"\xff"),          # int64
            D.tlv(0x03, b"\x00"), D.tlv(0x03, b"\x07\x80"), D.tlv(0x03, b"\x00\xa5\x5a"),
            D.tlv(0x06, b"\x2a\x03"), D.tlv(0x06, b"\x2a\x87\xff\xff\xff\x7f"),
            D.utctime("250101000000Z"), D.gentime("20250101000000Z"), D.utctime("2501010000Z"),
            D.tlv(0x04, b"\xff\xff"), D.tlv(0x01, b"\x42"), D.tlv(0x05, b"x"), D.tlv(0x30, b"\xff"),    # not looked into
            D.tlv(0x1e, b"\x00"), D.tlv(0x14, b"\xff"), D.tlv(0x82, b"\xff"), D.tlv(0x0a, b"")]
    bad = [D.tlv(0x02, b""), D.tlv(0x02, b"\x01" + b"\x00" * 8),                                      # empty; 9 octets
           D.tlv(0x03, b""), D.tlv(0x03, b"\x08\x00"), D.tlv(0x03, b"\x01\x01"), D.tlv(0x03, b"\x01"),
           D.tlv(0x06, b""), D.tlv(0x06, b"\x2a\x81"), D.tlv(0x06, b"\x2a\x88\x80\x80\x80\x00"),
           D.utctime("250101000000"), D.utctime("251301000000Z"), D.gentime("20250101000000+0000"),
           D.tlv(0x17, b""), D.tlv(0x18, b"2025")]
    for v in good:
        for where in ("subject", "issuer"):
            assert both(D.cert(**{where: _attr(t, v)})).ok, (v.hex(), where)
    for v in bad:
        for where in ("subject", "issuer"):
            assert not both(D.cert(**{where: _attr(t, v)})).ok, (v.hex(), where)
    # a not minimally encoded INTEGER is what only the lax re-parse takes: a non-fatal finding, as for the serial number
    lax = both(D.cert(subject=_attr(t, D.tlv(0x02, b"\x00\x01"))))
    assert lax.ok and lax.nonfatal & 2
    # the same value types behind a long-form attribute (the walk's general path) and inside a multi-valued RDN
    multi = D.seq(D.tlv(0x31, D.seq(t, D.tlv(0x0c, b"x")) + D.seq(t, D.tlv(0x03, b"\x08\x00"))))
    assert not both(D.cert(subject=multi)).ok
    long_form = D.seq(D.tlv(0x31, D.seq(t, b"\x02\x81\x01\x05")))
    assert not both(D.cert(subject=long_form)).ok          # the value's length is not minimally encoded
