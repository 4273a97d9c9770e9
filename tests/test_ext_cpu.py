"""strict_extensions (opt-in): the bodies of the extensions Go's parseCertificate unmarshals by plain struct rules — one
hand-built certificate per rule, oracle vs the product walk (host build), OpenSSL as a third opinion where it has one.
CPU only."""
import random

from oracle import oracle as orc
from tests import der as D
from tests import harness

AIA_OID = D.tlv(0x06, bytes.fromhex("2b06010505070101"))
OCSP = D.tlv(0x06, bytes.fromhex("2b06010505073001"))
POL = D.tlv(0x06, bytes.fromhex("6086480186f8420101"))
EKU_SRV = D.tlv(0x06, bytes.fromhex("2b06010505070301"))
EKU_CLI = D.tlv(0x06, bytes.fromhex("2b06010505070302"))


def x(oid_last, value):
    return D.ext(oid_last, value)


def aia(value):
    return D.seq(AIA_OID, D.tlv(0x04, value))


def verdicts(c):
    """(structure parses, strict_extensions accepts) — asserted equal between the oracle and the product."""
    o = orc.parse_cert(c)
    harness.product_set_ext(False)
    p0 = harness.product_walk(c)
    harness.product_set_ext(True)
    p1 = harness.product_walk(c)
    harness.product_set_ext(False)
    assert bool(o.ok) == bool(p0.ok), (o.ok, o.err_site, p0.ok)
    strict_ok = bool(o.ok) and not o.ext_fatal
    assert strict_ok == bool(p1.ok), (o.ok, o.ext_fatal, p1.ok)
    return bool(o.ok), strict_ok


def good(e):
    assert verdicts(D.cert(exts=[e])) == (True, True), e.hex()


def bad(e):
    assert verdicts(D.cert(exts=[e])) == (True, False), e.hex()      # only the switch rejects it


def test_key_usage_is_one_bit_string():
    good(x(15, D.tlv(0x03, b"\x05\xa0")))
    good(x(15, D.tlv(0x03, b"\x00")))                                # no bits at all
    good(x(15, D.tlv(0x03, b"\x07\x80")))
    for v in (D.tlv(0x03, b""), D.tlv(0x03, b"\x08\x00"), D.tlv(0x03, b"\x01"), D.tlv(0x03, b"\x05\xa1"),   # parseBitString
              D.tlv(0x04, b"\x05\xa0"), D.tlv(0x23, D.tlv(0x03, b"\x05\xa0")),                             # wrong / constructed tag
              D.tlv(0x03, b"\x05\xa0") + b"\x00", b"", b"\x03\x05\x05\xa0"):                               # trailing; empty; truncated
        bad(x(15, v))


def test_subject_key_identifier_is_one_octet_string():
    good(x(14, D.tlv(0x04, b"\x11" * 20)))
    good(x(14, D.tlv(0x04, b"")))
    for v in (D.tlv(0x03, b"\x00\x11"), D.tlv(0x24, D.tlv(0x04, b"\x11")), D.tlv(0x04, b"\x11") + b"\x05\x00", b"",
              b"\x04\x81\x01\x11"):                                   # … and a length that is not minimal
        bad(x(14, v))


def test_ext_key_usage_is_a_sequence_of_oids():
    good(x(37, D.seq(EKU_SRV, EKU_CLI)))
    good(x(37, D.seq()))
    for v in (D.seq(EKU_SRV, D.tlv(0x0c, b"x")),                      # sequence tag mismatch
              D.seq(D.tlv(0x06, b"")), D.seq(D.tlv(0x06, b"\x2a\x81")), D.seq(D.tlv(0x06, b"\x2a\x88\x80\x80\x80\x00")),
              D.tlv(0x31, EKU_SRV), D.seq(EKU_SRV) + b"\x00\x00", D.seq(EKU_SRV)[:-1] + b"", EKU_SRV):
        bad(x(37, v))


def test_authority_key_identifier_optional_first_field():
    kid = D.tlv(0x80, b"\x22" * 20)
    good(x(35, D.seq(kid)))
    good(x(35, D.seq()))
    good(x(35, D.seq(kid, D.tlv(0xa1, b"\xff\xff"), D.tlv(0x82, b"\x01"))))   # whatever follows the first field is ignored
    good(x(35, D.seq(D.tlv(0xa1, D.tlv(0x86, b"u")), D.tlv(0x82, b"\x01"))))   # no keyIdentifier: the field is skipped
    good(x(35, D.seq(D.tlv(0xa0, b"\x01"))))                                   # [0] but constructed: does not match, skipped
    good(x(35, D.seq(b"\x81\x7f")))                                            # a skipped element need not even fit
    for v in (D.seq(b"\x80\x7f\x01"),                                          # the keyIdentifier itself must fit
              D.seq(b"\x9f"), D.seq(b"\x80\x81\x01\x00"),                      # its header must parse (truncated tag; length not minimal)
              D.tlv(0x31, kid), D.seq(kid) + b"\x00", b""):
        bad(x(35, v))


def test_certificate_policies_is_a_sequence_of_sequences_led_by_an_oid():
    q = D.seq(D.tlv(0x06, bytes.fromhex("2b06010505070201")), D.tlv(0x16, b"http://cps"))
    good(x(32, D.seq(D.seq(POL), D.seq(POL, D.seq(q)))))
    good(x(32, D.seq(D.seq(POL, b"\xff\xff"))))                       # what follows the OID is not looked at
    good(x(32, D.seq()))
    for v in (D.seq(POL), D.seq(D.seq()), D.seq(D.seq(D.tlv(0x0c, b"x"))), D.seq(D.seq(D.tlv(0x06, b"\x2a\x81"))),
              D.seq(D.seq(POL), D.tlv(0x31, POL)), D.seq(D.seq(POL)) + b"\x05\x00", D.tlv(0x30, b"\x30")):
        bad(x(32, v))


def test_authority_info_access_elements_carry_an_oid_and_a_location():
    loc = D.tlv(0x86, b"http://ocsp.example")
    good(aia(D.seq(D.seq(OCSP, loc), D.seq(OCSP, D.tlv(0x0c, b"anything")))))
    good(aia(D.seq(D.seq(OCSP, loc, b"\xff"))))
    good(aia(D.seq()))
    for v in (D.seq(D.seq(OCSP)),                                     # Location is not optional: "sequence truncated"
              D.seq(D.seq(loc, OCSP)), D.seq(D.seq(OCSP, b"\x86\x05ab")), D.seq(OCSP, loc),
              D.seq(D.seq(OCSP, loc)) + b"\x00", D.tlv(0x31, D.seq(OCSP, loc))):
        bad(aia(v))


def test_other_extensions_and_the_switch_off():
    # subjectAltName, nameConstraints, cRLDistributionPoints and unknown extensions are not looked into
    for e in (x(17, b"\xff\xff"), x(30, b""), x(31, b"\x30\x80"), D.seq(D.tlv(0x06, b"\x2a\x03\x04"), D.tlv(0x04, b"\xff"))):
        good(e)
    # a repeated extension: every occurrence is checked
    c = D.cert(exts=[x(15, D.tlv(0x03, b"\x05\xa0")), x(15, D.tlv(0x03, b"\x08\x00"))])
    assert verdicts(c) == (True, False)
    # the critical flag and long-form headers in front of the value do not matter
    c = D.cert(exts=[D.ext(15, D.tlv(0x03, b"\x08\x00"), critical=True)])
    assert verdicts(c) == (True, False)
    long_form = D.seq(D.oid(0x55, 0x1d, 14), b"\x04\x81\x03" + D.tlv(0x04, b"\x11") )
    assert verdicts(D.cert(exts=[long_form]))[0] is False              # the extnValue length itself is not minimal: structure
    # the TBSCertificate walk (strict_leaf) applies the same rule
    c = D.cert(exts=[x(14, D.tlv(0x03, b"\x00\x11"))])
    o = orc.parse_cert(c)
    tbs = c[o.tbs_off:o.tbs_off + o.tbs_len]
    harness.product_set_ext(True)
    assert harness.product_walk_tbs(tbs).ok == 0
    harness.product_set_ext(False)
    assert harness.product_walk_tbs(tbs).ok == 1 and orc.parse_tbs(tbs).ext_fatal != 0


def test_real_world_certificates_pass_the_switch(golden_certs=None):
    """Every certificate of the reference's goldens and of the system CA bundle that parses also passes strict_extensions."""
    import glob
    import ssl
    ders = []
    for p in glob.glob("tests/golden/*.pem"):
        for blk in open(p).read().split("-----BEGIN CERTIFICATE-----")[1:]:
            ders.append(ssl.PEM_cert_to_DER_cert("-----BEGIN CERTIFICATE-----" + blk.split("-----END CERTIFICATE-----")[0] + "-----END CERTIFICATE-----\n"))
    for path in ("/etc/ssl/certs/ca-certificates.crt",):
        try:
            txt = open(path).read()
        except OSError:
            continue
        for blk in txt.split("-----BEGIN CERTIFICATE-----")[1:]:
            ders.append(ssl.PEM_cert_to_DER_cert("-----BEGIN CERTIFICATE-----" + blk.split("-----END CERTIFICATE-----")[0] + "-----END CERTIFICATE-----\n"))
    assert len(ders) >= 3
    n_ok = 0
    for d in ders:
        a, b = verdicts(d)
        if a:
            assert b, "a real certificate fails strict_extensions"
            n_ok += 1
    assert n_ok >= 3


def test_product_equals_oracle_on_mutated_extension_bodies():
    rng = random.Random(20261012)
    seeds = [D.cert(exts=[x(15, D.tlv(0x03, b"\x05\xa0")), x(14, D.tlv(0x04, b"\x11" * 20)), x(37, D.seq(EKU_SRV, EKU_CLI)),
                          x(35, D.seq(D.tlv(0x80, b"\x22" * 20))), x(32, D.seq(D.seq(POL))),
                          aia(D.seq(D.seq(OCSP, D.tlv(0x86, b"http://o.example")))), D.BC_NOT_CA])]
    o = orc.parse_cert(seeds[0])
    lo, hi = o.exts_off, o.exts_end
    n = rejected = 0
    for _ in range(30000):
        c = bytearray(seeds[0])
        for _k in range(rng.choice((1, 1, 2, 3))):
            p = rng.randrange(lo, hi)
            c[p] = rng.choice((c[p] ^ (1 << rng.randrange(8)), rng.randrange(256), 0x00, 0x80, 0x30, 0x06, 0x04, 0x03))
        a, b = verdicts(bytes(c))
        n += 1
        rejected += a and not b
    assert rejected > 1000                                             # the switch had something to say
