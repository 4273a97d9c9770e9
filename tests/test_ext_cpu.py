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


NF_STRING, NF_EXT = 4, 16


def verdicts(c, findings=None):
    """(structure parses, strict_extensions accepts) — asserted equal between the oracle and the product, and with them
    the NON-fatal findings of an accepted certificate: what CT-go files as non-fatal inside an extension body
    (WALK_NF_EXT ⇔ orc_cert.ext_findings) and, with strict_strings on as well, the character sets inside a distribution
    point's nameRelativeToCRLIssuer (WALK_NF_STRING ⇔ string_findings | ext_string_findings).  findings: a list that
    receives (ext finding, string finding) of an accepted certificate."""
    o = orc.parse_cert(c)
    harness.product_set_ext(False)
    p0 = harness.product_walk(c)
    harness.product_set_ext(True)
    p1 = harness.product_walk(c)
    harness.product_set_strings(True)
    p2 = harness.product_walk(c)
    harness.product_set_strings(False)
    harness.product_set_ext(False)
    assert bool(o.ok) == bool(p0.ok), (o.ok, o.err_site, p0.ok)
    strict_ok = bool(o.ok) and not o.ext_fatal
    assert strict_ok == bool(p1.ok) == bool(p2.ok), (o.ok, o.ext_fatal, p1.ok, p2.ok)
    if strict_ok:
        assert bool(p1.nonfatal & NF_EXT) == bool(o.ext_findings) == bool(p2.nonfatal & NF_EXT), (o.ext_findings, p1.nonfatal)
        assert not (p1.nonfatal & NF_STRING)
        assert bool(p2.nonfatal & NF_STRING) == bool(o.string_findings or o.ext_string_findings), \
            (o.string_findings, o.ext_string_findings, p2.nonfatal)
        if findings is not None:
            findings.append((bool(o.ext_findings), bool(o.ext_string_findings)))
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
    for v in (D.seq(D.seq(OCSP)),                                     # Location is not optional: "sequence truncated"
              D.seq(D.seq(loc, OCSP)), D.seq(D.seq(OCSP, b"\x86\x05ab")), D.seq(OCSP, loc),
              D.seq(D.seq(OCSP, loc)) + b"\x00", D.tlv(0x31, D.seq(OCSP, loc)),
              D.seq()):                                               # CT-go (recalled): "x509: empty AuthorityInfoAccess extension"
        bad(aia(v))


SIA_OID = D.tlv(0x06, bytes.fromhex("2b0601050507010b"))
CA_REPO = D.tlv(0x06, bytes.fromhex("2b06010505073005"))


def sia(value):
    return D.seq(SIA_OID, D.tlv(0x04, value))


def test_subject_info_access_is_parsed_like_authority_info_access():
    """CT-go's fork knows 1.3.6.1.5.5.7.1.11 (the standard library does not): []accessDescription filling the value, failure
    and trailing data fatal, an empty list an error (recalled — go.mod:10; round 6)."""
    loc = D.tlv(0x86, b"rsync://repo.example/ca/")
    good(sia(D.seq(D.seq(CA_REPO, loc), D.seq(CA_REPO, D.tlv(0xa4, b"\x31\x00")))))
    good(sia(D.seq(D.seq(CA_REPO, loc, b"\xff"))))
    for v in (D.seq(), D.seq(D.seq(CA_REPO)), D.seq(D.seq(loc, CA_REPO)), D.seq(D.seq(CA_REPO, b"\x86\x05ab")), D.seq(CA_REPO, loc),
              D.seq(D.seq(CA_REPO, loc)) + b"\x00", D.tlv(0x31, D.seq(CA_REPO, loc)), b"", D.seq(D.seq(D.tlv(0x06, b"\x2a\x81"), loc))):
        bad(sia(v))
    # the neighbouring id-pe numbers are not this extension: an unknown extension's value is not looked at
    for last in (0x02, 0x0a, 0x0c):
        good(D.seq(D.tlv(0x06, bytes.fromhex("2b06010505070100")[:-1] + bytes([last])), D.tlv(0x04, b"\xff\xff")))


IPADDR_OID = D.tlv(0x06, bytes.fromhex("2b06010505070107"))
ASNUM_OID = D.tlv(0x06, bytes.fromhex("2b06010505070108"))
NULL = b"\x05\x00"


def ipaddr(value, critical=True):
    return D.seq(IPADDR_OID, *( [D.tlv(0x01, b"\xff")] if critical else []), D.tlv(0x04, value))


def asnum(value):
    return D.seq(ASNUM_OID, D.tlv(0x01, b"\xff"), D.tlv(0x04, value))


def bits(b, pad=0):
    return D.tlv(0x03, bytes([pad]) + b)


def test_rfc3779_address_blocks_findings_are_non_fatal():
    """sbgp-ipAddrBlock as CT-go's x509/rpki.go decodes it (recalled; go.mod:10): strict asn1.Unmarshal calls, every failure
    an nfe.AddError — the certificate parses, a precertificate and a Chain[0] issuer are dropped over the finding."""
    v4, v6 = D.tlv(0x04, b"\x00\x01"), D.tlv(0x04, b"\x00\x02\x01")
    fam = lambda af, choice, *more: D.seq(af, choice, *more)
    ok = [D.seq(), D.seq(fam(v4, NULL)), D.seq(fam(v4, D.seq(bits(b"\x0a"), bits(b"\xc0\xa8\x80", 7)))),
          D.seq(fam(v4, D.seq(D.seq(bits(b"\x0a\x00"), bits(b"\x0a\xff"))), b"\xff\xff"), fam(v6, NULL)),   # behind the last field: ignored
          D.seq(fam(v6, D.seq())), D.seq(fam(v4, D.seq(D.seq(bits(b"\x0a"), bits(b"\x0b"), b"\x00"))))]
    for v in ok:
        f = []
        assert verdicts(D.cert(exts=[ipaddr(v)]), f) == (True, True), v.hex()
        assert f == [(False, False)], (f, v.hex())
    findings = [b"", D.tlv(0x31, fam(v4, NULL)), D.seq(fam(v4, NULL)) + b"\x00",            # not one SEQUENCE filling the value
                D.seq(D.tlv(0x31, v4 + NULL)), D.seq(fam(D.tlv(0x24, v4), NULL)), D.seq(fam(D.tlv(0x03, b"\x00\x01"), NULL)),
                D.seq(D.seq(v4)), D.seq(D.seq(v4, b"\x30\x05\x00")),                     # Choice missing / does not fit
                D.seq(fam(D.tlv(0x04, b"\x01"), NULL)), D.seq(fam(D.tlv(0x04, b"\x00\x01\x01\x01"), NULL)),   # AFI of 1 or 4 octets
                D.seq(fam(v4, b"\x05\x81\x00")), D.seq(fam(v4, D.tlv(0x04, b""))), D.seq(fam(v4, D.tlv(0x31, bits(b"\x0a")))),
                D.seq(fam(v4, D.seq(b"\x03\x05\x00"))),                                   # an element that does not fit
                D.seq(fam(v4, D.seq(D.tlv(0x03, b"")))), D.seq(fam(v4, D.seq(bits(b"\x0a", 8)))), D.seq(fam(v4, D.seq(bits(b"\x0b", 1)))),
                D.seq(fam(v4, D.seq(D.tlv(0x83, b"\x00\x0a")))), D.seq(fam(v4, D.seq(D.tlv(0x23, bits(b"\x0a"))))),   # number 3, not universal primitive
                D.seq(fam(v4, D.seq(D.seq(bits(b"\x0a"))))), D.seq(fam(v4, D.seq(D.seq(bits(b"\x0a"), D.tlv(0x04, b"\x00"))))),
                D.seq(fam(v4, D.seq(D.tlv(0xb0, bits(b"\x0a") + bits(b"\x0b"))))),       # number 16 of another class
                D.seq(fam(v4, D.seq(D.tlv(0x04, b"\x0a")))), D.seq(fam(v4, D.seq(b"\x9f\x21\x00"))),
                D.seq(fam(v4, NULL), fam(v4, D.seq(D.tlv(0x02, b"\x01"))))]
    for v in findings:
        nonfatal(ipaddr(v))
    nonfatal(ipaddr(findings[5], critical=False))
    # a finding next to a fatal body elsewhere stays fatal; the switch off: nothing is looked at
    bad_and = D.cert(exts=[ipaddr(findings[0]), x(15, D.tlv(0x03, b""))])
    assert verdicts(bad_and) == (True, False)


def test_rfc3779_as_identifiers_findings_are_non_fatal():
    """sbgp-autonomousSysNum: struct { ASNum RawValue `optional,tag:0`; RDI RawValue `optional,tag:1` } filling the value; a
    choice's contents are NULL or one []RawValue of INTEGERs (`int`: minimal, at most 8 octets) and ASIDRange pairs."""
    i = lambda b: D.tlv(0x02, b)
    ok = [D.seq(), D.seq(D.tlv(0xa0, NULL)), D.seq(D.tlv(0xa0, D.seq(i(b"\x01"), D.seq(i(b"\x02"), i(b"\x00\xff\xff"))))),
          D.seq(D.tlv(0xa0, NULL), D.tlv(0xa1, D.seq(i(b"\x7f" + b"\xff" * 7)))), D.seq(D.tlv(0xa1, D.seq())),
          D.seq(D.tlv(0x80, NULL)),                                  # `tag:0` on a RawValue: the class and the number, either form
          D.seq(D.tlv(0xa0, D.seq(i(b"\x01"))), b"\xff\xff\xff"[:0] + D.tlv(0xa2, b"\x01")),   # what follows the two fields: ignored
          D.seq(D.tlv(0xa1, NULL), D.tlv(0xa0, b"\x02")),            # [0] behind [1]: never reached
          D.seq(D.tlv(0xa0, D.seq(D.seq(i(b"\x01"), i(b"\x02"), b"\x05\x00"))))]
    for v in ok:
        f = []
        assert verdicts(D.cert(exts=[asnum(v)]), f) == (True, True), v.hex()
        assert f == [(False, False)], (f, v.hex())
    findings = [b"", D.tlv(0x31, D.tlv(0xa0, NULL)), D.seq(D.tlv(0xa0, NULL)) + b"\x00",
                D.seq(b"\xa0\x05\x05\x00"), D.seq(b"\x9f"),                            # a matching field must fit; a header must parse
                D.seq(D.tlv(0xa0, b"")), D.seq(D.tlv(0xa0, b"\x05\x81\x00")), D.seq(D.tlv(0xa0, NULL + b"\x00")),
                D.seq(D.tlv(0xa0, D.tlv(0x31, i(b"\x01")))), D.seq(D.tlv(0xa0, D.seq(i(b"\x01")) + b"\x00")),
                D.seq(D.tlv(0xa0, D.seq(b"\x02\x05\x01"))), D.seq(D.tlv(0xa0, D.seq(i(b"")))), D.seq(D.tlv(0xa0, D.seq(i(b"\x00\x01")))),
                D.seq(D.tlv(0xa0, D.seq(i(b"\x01" * 9)))), D.seq(D.tlv(0xa0, D.seq(D.tlv(0x82, b"\x01")))),
                D.seq(D.tlv(0xa0, D.seq(D.seq(i(b"\x01"))))), D.seq(D.tlv(0xa0, D.seq(D.seq(i(b"\x01"), D.tlv(0x04, b"\x02"))))),
                D.seq(D.tlv(0xa0, D.seq(D.tlv(0x04, b"\x01")))), D.seq(D.tlv(0xa0, NULL), D.tlv(0xa1, D.seq(i(b"\xff\xff"))))]
    for v in findings:
        nonfatal(asnum(v))


def san(*names):
    return x(17, D.seq(*names))


def uri(u):
    return D.tlv(0x86, u if isinstance(u, bytes) else u.encode())


def nonfatal(e, ext=True, strings=False):
    f = []
    assert verdicts(D.cert(exts=[e]), f) == (True, True), e.hex()
    assert f == [(ext, strings)], (f, e.hex())


def test_subject_alt_name_is_a_sequence_of_general_names():
    """forEachSAN: one universal constructed SEQUENCE filling the value, every element a TLV that fits, dispatched on the
    tag NUMBER alone."""
    good(san(D.tlv(0x82, b"a.example"), D.tlv(0x81, b"x@a.example"), D.tlv(0x87, bytes(4)), D.tlv(0x87, bytes(16)),
             uri("https://a.example/x")))
    good(san())                                                       # empty: parses (only "unhandled" when critical)
    good(san(D.tlv(0x82, b"\xff\x00 not IA5")))                       # no character set is checked on the parse side
    good(san(D.tlv(0xa0, b"\xff\xff"), D.tlv(0xa4, b"\x05"), D.tlv(0x88, b""), D.tlv(0x05, b"")))   # other names: not looked into
    good(san(b"\x9f\x21\x01\x00"))                                   # a high-tag-number element: number 33, no case
    for v in (b"", D.tlv(0x31, D.tlv(0x82, b"a")), D.tlv(0x10, D.tlv(0x82, b"a")),                  # not 0x30
              D.seq(D.tlv(0x82, b"a")) + b"\x00", D.seq(b"\x82\x05ab"), D.seq(b"\x82"), D.seq(b"\x82\x81\x01a"),
              D.seq(b"\x9f\x1e\x00")):                                # a high-tag-number form for a number below 31
        bad(x(17, v))
    # iPAddress of another length: CT-go files it as NON-fatal (the stdlib fails); any class with number 7 counts
    for n in (0, 3, 5, 8, 15, 17, 32):
        nonfatal(san(D.tlv(0x87, bytes(n))))
    nonfatal(san(D.tlv(0x07, b"abc")))
    nonfatal(san(D.tlv(0xa7, D.tlv(0x04, b"abc"))))
    nonfatal(san(D.tlv(0x82, b"ok"), D.tlv(0x87, bytes(4)), D.tlv(0x87, bytes(5))))


def test_subject_alt_name_uri_goes_through_url_parse():
    ok = ["https://a.example/x", "http://a.example:8080/p?q#f", "urn:isbn:0451450523", "mailto:u@h", "", "*", "/just/a/path",
          "relative/path", "?q=1", "#frag", "http://", "http:///path", "///three", "//host.example/x", "http://[::1]:80/",
          "http://[fe80::1%25en0]/", "http://u:p@h.example/", "http://u%41@h.example",
          "http://h.example/%41", "http://h.example/a%20b", "http://a.b/c?%zz", "HTTP://A.EXAMPLE", "a+b-c.d:rest",
          "http://h.example:/x", "http://h!$&'()*+,;=.example", "http://h.example/\x80\xff",
          "http://a.example/#%41", "http://a_b.example", "x://h<>\"/", "http://h/a:b", "./a:b", "http://h/?\x01"[:9], "*#f",
          "http://@h/", "http://:@h/", "http://h.example:80", "x:", "x:%zz", "http://a@b@c/",
          # ONE leading dot passes domainToReverseLabels in the release the reference builds with: labels are cut off the END and
          # the empty one in front is never recorded (later releases append it: "domain is prefixed with an empty label")
          "http://.example/", "//.h/", "http://.a.b:80/"]
    bad_ = [":", ":x", "a b"[0:0] + "\x7f", "http://a\x01", "\x00", "http://h/\x1f", "a b:c",   # control characters; a colon in the first segment
            "//h:port/", "http://h:80x/", "http://h:-1", "http://[::1", "http://[::1]x", "http://[::1]:x", "http://a:b:c/",
            "http://h%20x/", "http://h%41/", "http://%zz/", "http://h/%", "http://h/%4", "http://h/%zz", "http://h/#%", "http://h/#%g1",
            "http://h x/", "http://h\\x/", "http://h^x/", "http://h`x/", "http://h{x/", "http://h|x/", "http://h}x/",
            "http://u ser@h/", "http://us\x80er@h/", "http://u%zz@h/", "http://u:p%@h/", "http://a@b@c d/",
            "http://..example/", "http://./", "http://example./", "http://a..b/", "http://h\x80/", "http://h%c3%a9/", "http://[::1%25\x80]/",
            "http://[fe80::1%25e%7fn]/", "//..h/", "*\x01", "http://h/%41%zz",
            "1http://x/y",                                            # no scheme (a digit first): a colon in the first path segment
            "http://[fe80::1%25e%20n]/"]                              # url.Parse takes the space in the zone, domainToReverseLabels does not
    for u in ok:
        good(san(uri(u.encode("latin-1"))))
    for u in bad_:
        if u:
            bad(san(uri(u.encode("latin-1"))))
    # any class with number 6; the URI among other names
    bad(san(D.tlv(0x06, b":x")))
    bad(san(D.tlv(0x82, b"fine"), uri("http://ok.example"), D.tlv(0xa6, b"\x7f")))


def dps(*points):
    return x(31, D.seq(*points))


def fullname(*names):
    return D.tlv(0xa0, D.tlv(0xa0, b"".join(names)))


def test_crl_distribution_points_follow_the_struct_rules():
    u = D.tlv(0x86, b"http://crl.example/a.crl")
    good(dps(D.seq(fullname(u))))
    good(dps())                                                       # an empty SEQUENCE OF
    good(dps(D.seq()))                                                # every field optional
    good(dps(D.seq(fullname(u), D.tlv(0x81, b"\x01\x06"), D.tlv(0xa2, D.tlv(0xa4, b"")))))
    good(dps(D.seq(fullname(u, D.tlv(0x82, b"x"), D.tlv(0x05, b"")), b"\x05\x00\xff")))   # names of any kind; behind the fields only the next header must parse
    good(dps(D.seq(D.tlv(0x81, b"\x00"), fullname(b"\xff"))))                      # fields out of order: [0] behind reasons is never parsed
    good(dps(D.seq(D.tlv(0x82, b"anything"))))                        # cRLIssuer primitive
    good(dps(D.seq(D.tlv(0x80, b"\xff"))))                            # [0] primitive: no field matches, skipped
    good(dps(D.seq(D.tlv(0xa0, b""))))                                # an empty distributionPointName
    good(dps(D.seq(D.tlv(0xa0, D.tlv(0xa1, D.tlv(0x31, D.seq(D.oid(0x55, 4, 3), D.tlv(0x0c, b"rel"))))))))   # nameRelativeToCRLIssuer
    good(dps(D.seq(D.tlv(0xa0, D.tlv(0xa0, b"") + D.tlv(0xa1, b"")))))              # both, empty
    good(dps(D.seq(D.tlv(0xa0, D.tlv(0xa2, b"\xff") + b"\x00\x00"))))              # inside [0]: another tag skipped, the rest ignored
    for v in (b"", D.tlv(0x31, b""), D.seq() + b"\x00", D.seq(D.tlv(0x31, b"")), D.seq(D.tlv(0x04, b"")),      # outer, elements
              D.seq(D.seq(b"\xa0")), D.seq(D.seq(b"\xa0\x05ab")), D.seq(D.seq(b"\x9f")),                      # a field header that does not parse / fit
              D.seq(D.seq(D.tlv(0xa0, b"\xa0"))), D.seq(D.seq(D.tlv(0xa0, b"\xa0\x03ab"))),                   # … inside the name
              D.seq(D.seq(fullname(b"\x86\x05ab"))), D.seq(D.seq(fullname(u, b"\x86"))),                     # FullName elements must fit
              D.seq(D.seq(D.tlv(0x81, b""))), D.seq(D.seq(D.tlv(0x81, b"\x08\x00"))), D.seq(D.seq(D.tlv(0x81, b"\x01\x01"))),   # reasons: parseBitString
              D.seq(D.seq(D.tlv(0x81, b"\x00"), b"\x82\x05ab")), D.seq(D.seq(b"\xa2\x05ab")),               # cRLIssuer must fit
              D.seq(D.seq(fullname(u), b"\x9f")), D.seq(D.seq(D.tlv(0x81, b"\x00"), b"\x81")),               # the header behind a field
              D.seq(D.seq(D.tlv(0xa0, D.tlv(0xa1, D.tlv(0x30, b""))))),                                        # RelativeName: SET elements
              D.seq(D.seq(D.tlv(0xa0, D.tlv(0xa1, D.tlv(0x31, D.seq(D.oid(0x55, 4, 3))))))),                   # … the value is not optional
              D.seq(D.seq(D.tlv(0xa0, D.tlv(0xa1, D.tlv(0x31, D.seq(D.tlv(0x06, b"\x55\x84"), D.tlv(0x0c, b"x"))))))),
              D.seq(D.seq(D.tlv(0xa0, D.tlv(0xa1, D.tlv(0x31, D.seq(D.oid(0x55, 4, 3), D.tlv(0x02, b""))))))),
              D.seq(D.seq(D.tlv(0xa0, D.tlv(0xa0, b"") + D.tlv(0xa1, b"\x31"))))):
        bad(x(31, v))
    # findings inside nameRelativeToCRLIssuer: an INTEGER only the lax parser takes; a PrintableString with '@'
    rel = lambda val: dps(D.seq(D.tlv(0xa0, D.tlv(0xa1, D.tlv(0x31, D.seq(D.oid(0x55, 4, 5), val))))))
    nonfatal(rel(D.tlv(0x02, b"\x00\x01")))
    nonfatal(rel(D.tlv(0x13, b"a@b")), ext=False, strings=True)
    # where the URIs lie (kernels/meta*.h, orc_cert_meta): FullName elements with tag NUMBER 6, positional fields
    def uris(value):
        c = D.cert(exts=[x(31, value)])
        _, ulist, m = orc.cert_meta(c)
        o = orc.parse_cert(c)
        at = c.index(value)
        got = harness.product_crl_uris(c, at, at + len(value))
        want = None if m.bad_crl else ulist[:8]
        assert got == want and o.ok, (got, want)
        return got
    assert uris(D.seq(D.seq(fullname(u, D.tlv(0x06, b"oid-as-uri"), D.tlv(0x82, b"dns"), D.tlv(0xa6, b"cons"))))) == \
        [b"http://crl.example/a.crl", b"oid-as-uri", b"cons"]
    assert uris(D.seq(D.seq(D.tlv(0x81, b"\x00"), fullname(u)))) == []                # out of order: not reached
    assert uris(D.seq(D.seq(fullname(u)), D.seq(fullname(D.tlv(0x86, b"second"))))) == [b"http://crl.example/a.crl", b"second"]
    assert uris(D.seq(D.seq(fullname(u), b"\x9f"))) is None
    assert uris(D.seq(D.seq(D.tlv(0xa0, D.tlv(0xa0, u) + D.tlv(0xa1, D.tlv(0x30, b"")))))) == [b"http://crl.example/a.crl"]   # RelativeName contents: not the metadata's business


def nc(permitted=None, excluded=None, raw=None):
    body = b""
    if permitted is not None:
        body += D.tlv(0xa0, b"".join(D.seq(c) for c in permitted))
    if excluded is not None:
        body += D.tlv(0xa1, b"".join(D.seq(c) for c in excluded))
    return x(30, D.seq(body) if raw is None else raw)


def test_name_constraints_as_cryptobyte_reads_them():
    dns = lambda b: D.tlv(0x82, b)
    good(nc([dns(b"example.com")], [dns(b".example.org")]))
    # one leading dot is trimmed by the caller, a second one passes domainToReverseLabels (see the URI cases); a third does not
    good(nc([dns(b"..example.com"), D.tlv(0x81, b"user@.example"), D.tlv(0x81, b"..example.com"), D.tlv(0x86, b"..example.com"),
             dns(b"."), dns(b"")]))                                                     # (trimmed to nothing: no labels, no complaint)
    good(nc([dns(b"")]))                                              # an empty domain has no labels: fine
    good(nc(None, [D.tlv(0x87, bytes(4) + b"\xff\xff\xf0\x00")]))
    good(nc([D.tlv(0x87, bytes(16) + b"\xff" * 8 + bytes(8))]))
    good(nc([D.tlv(0x81, b"user@example.com"), D.tlv(0x81, b"example.com"), D.tlv(0x81, b".example.com"),
             D.tlv(0x81, b'"quoted string"@example.com'), D.tlv(0x81, b"a\\@b@example.com"), D.tlv(0x81, b"u@")]))
    good(nc([D.tlv(0x86, b"example.com"), D.tlv(0x86, b".example.com"), D.tlv(0x86, b"1.2.3"), D.tlv(0x86, b"1.2.3.256")]))
    good(nc([D.tlv(0xa4, b"\xff"), D.tlv(0x88, b"\x2a"), D.tlv(0x02, b"")]))        # other name forms: unhandled, not an error
    good(nc([dns(b"a") + D.tlv(0x80, b"\x00") + b"\xff\xff"]))                     # minimum / maximum are never read
    for e in (nc(raw=b""), nc(raw=D.seq()), nc([], []), nc(raw=D.tlv(0x31, D.tlv(0xa0, D.seq(dns(b"a"))))),
              nc(raw=D.seq(D.tlv(0xa0, D.seq(dns(b"a")))) + b"\x00"),
              nc(raw=D.seq(D.tlv(0xa1, D.seq(dns(b"a"))) + D.tlv(0xa0, D.seq(dns(b"a"))))),   # [1] before [0]: toplevel not empty
              nc(raw=D.seq(D.tlv(0xa0, D.seq(dns(b"a"))) + b"\x05\x00")),
              nc(raw=D.seq(b"\xa0\x05ab")), nc(raw=D.seq(D.tlv(0xa0, D.tlv(0x31, dns(b"a"))))),
              nc(raw=D.seq(D.tlv(0xa0, D.seq()))), nc(raw=D.seq(D.tlv(0xa0, D.seq(b"\x82")))),
              nc(raw=D.seq(D.tlv(0xa0, D.seq(b"\x9f\x21\x00")))),                     # cryptobyte refuses the high-tag-number form
              nc(raw=D.seq(D.tlv(0x80, b"")))):                                        # [0] primitive: not the optional element
        bad(e)
    for c in (dns(b"exa mple.com"), dns(b"example..com"), dns(b"example.com."), dns(b"...example.com"), dns(b".."), dns(b"\x80"), dns(b"a\x7fb"),
              D.tlv(0x87, bytes(7)), D.tlv(0x87, bytes(9)), D.tlv(0x87, bytes(4) + b"\xff\x00\xff\x00"), D.tlv(0x87, bytes(4) + b"\xfd\x00\x00\x00"),
              D.tlv(0x87, bytes(16) + b"\x00" * 15 + b"\x01"),
              D.tlv(0x81, b"@example.com"), D.tlv(0x81, b".user@example.com"), D.tlv(0x81, b"us..er@example.com"), D.tlv(0x81, b"user.@x"),
              D.tlv(0x81, b'"unterminated@example.com'), D.tlv(0x81, b'"a"b@example.com'), D.tlv(0x81, b"a b@example.com"),
              D.tlv(0x81, b"user@exa mple"), D.tlv(0x81, b"user@..example"), D.tlv(0x81, b"user@example."), D.tlv(0x81, b"u\\@x"), D.tlv(0x81, b'"\\\n"@x'), D.tlv(0x81, b"ex ample.com"),
              D.tlv(0x81, b"\xe9@example.com"),
              D.tlv(0x86, b"1.2.3.4"), D.tlv(0x86, b"01.02.03.004"), D.tlv(0x86, b"::1"), D.tlv(0x86, b"::"), D.tlv(0x86, b"2001:db8::1"),
              D.tlv(0x86, b"1:2:3:4:5:6:7:8"), D.tlv(0x86, b"::ffff:1.2.3.4"), D.tlv(0x86, b"1:2:3:4:5:6:1.2.3.4"),
              D.tlv(0x86, b"exa mple"), D.tlv(0x86, b"a..b"), D.tlv(0x86, b"\xff")):
        bad(nc([c]))
        bad(nc(None, [c]))
    # not addresses → judged as domains (all fine: ':' is a printable label octet)
    for c in (b"1:2:3:4:5:6:7", b"1:2:3:4:5:6:7:8:9", b"::1::", b"1::2::3", b"12345::", b":1", b"1:", b"::1.2.3", b"1.2.3.4.5",
              b"1:2:3:4:5:1.2.3.4", b"g::1", b"1:2:3:4:5:6:7:1.2.3.4", b"ffffff::"):
        good(nc([D.tlv(0x86, c)]))


SCT_OID = D.tlv(0x06, bytes.fromhex("2b06010401d679020402"))


def sct_ext(value):
    return D.seq(SCT_OID, D.tlv(0x04, value))


def test_embedded_sct_list_findings_are_non_fatal():
    one = b"\x00\x03abc"
    lst = lambda *s: D.tlv(0x04, len(b"".join(s)).to_bytes(2, "big") + b"".join(s))
    good(sct_ext(lst(one)))
    good(sct_ext(lst(one, b"\x00\x01x")))
    for v in (b"", D.tlv(0x04, b""), D.tlv(0x04, b"\x00"), D.tlv(0x04, b"\x00\x00"), lst(), lst(b"\x00\x00"), lst(one) + b"\x00",
              D.tlv(0x04, b"\x00\x06" + one), D.tlv(0x04, b"\x00\x04" + one), lst(b"\x00\x05abc"), lst(one, b"\x00"),
              D.tlv(0x03, b"\x00\x05" + one), D.tlv(0x24, lst(one))):
        nonfatal(sct_ext(v))
    # the neighbouring OID …2.4.3 (the precertificate poison) is not an SCT list
    good(D.seq(D.tlv(0x06, bytes.fromhex("2b06010401d679020403")), D.tlv(0x01, b"\xff"), D.tlv(0x04, b"\x05\x00")))


def test_other_extensions_and_the_switch_off():
    # unknown extensions are not looked into
    for e in (x(18, b"\xff\xff"), x(9, b""), D.seq(D.tlv(0x06, b"\x2a\x03\x04"), D.tlv(0x04, b"\xff"))):
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


def rich_seeds():
    """Certificates that carry every extension body strict_extensions looks into (the fuzz campaigns' seeds)."""
    old = [x(15, D.tlv(0x03, b"\x05\xa0")), x(14, D.tlv(0x04, b"\x11" * 20)), x(37, D.seq(EKU_SRV, EKU_CLI)),
           x(35, D.seq(D.tlv(0x80, b"\x22" * 20))), x(32, D.seq(D.seq(POL))),
           aia(D.seq(D.seq(OCSP, D.tlv(0x86, b"http://o.example")))), D.BC_NOT_CA]
    u = D.tlv(0x86, b"http://crl.example/a.crl")
    rel = D.tlv(0xa1, D.tlv(0x31, D.seq(D.oid(0x55, 4, 3), D.tlv(0x13, b"rel name"))))
    lst = lambda *q: D.tlv(0x04, len(b"".join(q)).to_bytes(2, "big") + b"".join(q))
    web = [san(D.tlv(0x82, b"a.example"), D.tlv(0x82, b"*.b.example"), D.tlv(0x87, bytes(4)), D.tlv(0x87, bytes(16)),
               uri("https://u:p@a.example:8443/x%20y?q#f"), D.tlv(0x81, b"m@a.example"), uri("http://[fe80::1%25en0]:80/")),
           dps(D.seq(fullname(u)), D.seq(D.tlv(0xa0, rel), D.tlv(0x81, b"\x01\x06"), D.tlv(0xa2, D.tlv(0xa4, b"")))),
           sct_ext(lst(b"\x00\x03abc", b"\x00\x02xy")), D.BC_NOT_CA]
    ca = [nc([D.tlv(0x82, b".example.com"), D.tlv(0x87, bytes(4) + b"\xff\xff\x00\x00"), D.tlv(0x81, b"user@example.com"),
              D.tlv(0x81, b'"q s"@example.com'), D.tlv(0x86, b".example.com")],
             [D.tlv(0x87, bytes(16) + b"\xff" * 6 + bytes(10)), D.tlv(0x82, b"x.y"), D.tlv(0x86, b"1:2:3:4:5:6:7")]),
          D.BC_CA]
    # round 6: CT-go's own extensions — subjectInfoAccess, RFC 3779 address blocks and AS identifiers
    i = lambda v: D.tlv(0x02, v)
    rpki = [sia(D.seq(D.seq(CA_REPO, D.tlv(0x86, b"rsync://repo.example/ca/")))),
            ipaddr(D.seq(D.seq(D.tlv(0x04, b"\x00\x01"), D.seq(bits(b"\x0a"), D.seq(bits(b"\xc0\xa8"), bits(b"\xc0\xa9\x80", 7)))),
                         D.seq(D.tlv(0x04, b"\x00\x02\x01"), NULL))),
            asnum(D.seq(D.tlv(0xa0, D.seq(i(b"\x00\xfd\xe8"), D.seq(i(b"\x01"), i(b"\x02")))), D.tlv(0xa1, NULL))), D.BC_CA]
    return [D.cert(exts=old), D.cert(exts=web), D.cert(exts=ca), D.cert(exts=web[:2] + old), D.cert(exts=rpki)]


def mutate_exts(rng, der, lo, hi):
    c = bytearray(der)
    for _k in range(rng.choice((1, 1, 2, 3))):
        p = rng.randrange(lo, hi)
        c[p] = rng.choice((c[p] ^ (1 << rng.randrange(8)), rng.randrange(256), 0x00, 0x80, 0x30, 0x06, 0x04, 0x03, 0x25, 0x2e, 0x3a,
                           0x40, 0x5b, 0x5d, 0x2f, 0x23, 0x3f, 0x5c, 0x22, 0xa0, 0xa1, 0x86, 0x87, 0x82, 0x81, 0x31))
    return bytes(c)


def test_product_equals_oracle_on_mutated_extension_bodies():
    rng = random.Random(20261012)
    seeds = rich_seeds()
    rng_ranges = []
    for sd in seeds:
        o = orc.parse_cert(sd)
        assert o.ok and not o.ext_fatal and not o.ext_findings
        rng_ranges.append((o.exts_off, o.exts_end))
    n = rejected = nf = 0
    for _ in range(40000):
        k = rng.randrange(len(seeds))
        f = []
        a, b = verdicts(mutate_exts(rng, seeds[k], *rng_ranges[k]), f)
        n += 1
        rejected += a and not b
        nf += bool(f and (f[0][0] or f[0][1]))
    assert rejected > 1000 and nf > 20                                 # the switch had something to say, both ways


def test_product_equals_oracle_on_random_uris_and_constraints():
    """Strings made for the content parsers (url.Parse, parseRFC2821Mailbox, net.ParseIP, domainToReverseLabels, the IP mask):
    product (host build) ≡ oracle on each, as a subjectAltName URI and as each kind of name constraint."""
    rng = random.Random(20261013)
    alpha_u = "ab1:/@[]%25.?#-+~! \\\"<>_*|\x01\x7f\x80é"
    alpha_m = "ab1.@\\\" -!#\x0b\x7f"
    alpha_i = "0123456789abcdefg:.:."
    seen = set()
    for t in range(12000):
        a = (alpha_u, alpha_m, alpha_i)[t % 3]
        sx = "".join(rng.choice(a) for _ in range(rng.randrange(0, 14)))
        if t % 7 == 0:
            sx = rng.choice(("http://", "//", "http://[", "x:", "http://u@", "")) + sx
        b = sx.encode("latin-1")
        e = [san(uri(b)), nc([D.tlv(0x86, b)]), nc(None, [D.tlv(0x81, b)]), nc([D.tlv(0x82, b)])][rng.randrange(4)]
        seen.add(verdicts(D.cert(exts=[e]))[1])
        if t % 5 == 0:
            m = bytes(rng.choice((0, 0xff, 0x80, 0xfe, 0xf0, 0x01)) for _ in range(rng.choice((4, 4, 16, 3))))
            seen.add(verdicts(D.cert(exts=[nc([D.tlv(0x87, bytes(len(m)) + m)])]))[1])
    assert seen == {True, False}
