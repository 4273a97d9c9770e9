"""strict_strings (opt-in): the character-set rules Go's encoding/asn1 applies to the string values of a Name
(parsePrintableString / parseNumericString / parseIA5String / parseUTF8String) — the oracle's restatement
(orc_cert.string_findings) against the product's host build (der_walk.h name_strings_ok), hand-built cases per rule and
a randomised comparison with Python's own UTF-8 decoder as a third opinion.  CPU only.

Stdlib rules; what certificate-transparency-go's lax fork makes of a violation is unverified — hence
the switch, off by default, and the finding filed as non-fatal."""
import random

from oracle import oracle as orc
from tests import der as D
from tests import harness

PRINTABLE = set(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789 '()+,-./:=?*&")


def verdicts(c):
    o = orc.parse_cert(c)
    p = harness.product_name_strings(c)
    assert o.ok and p in (0, 1), (o.ok, o.err_site, p)
    assert (o.string_findings == 0) == (p == 1), (o.string_findings, p)
    return o.string_findings


def with_value(tag, value, where="subject", attr=10):
    nm = D.name(D.rdn(3, b"Some CA"), D.rdn(attr, value, tag=tag))
    return D.cert(**{where: nm})


def test_printable_string_rule():
    for where in ("issuer", "subject"):
        assert verdicts(with_value(0x13, b"Plain Org (EU) +1, a-b.c/d:e=f?'", where)) == 0
        assert verdicts(with_value(0x13, b"*.example.com & Sons", where)) == 0          # '*' and '&': tolerated by Go
        for bad in (b"under_score", b"at@sign", b"semi;colon", b"caf\xe9", b"nul\x00", b"quote\"", b"hash#", b"ex!"):
            assert verdicts(with_value(0x13, bad, where)) == orc.SF_PRINTABLE, bad
    for b in range(256):                                                                  # the whole alphabet, octet by octet
        assert (verdicts(with_value(0x13, b"ab" + bytes([b]) + b"cd")) == 0) == (b in PRINTABLE), b


def test_numeric_ia5_rules():
    assert verdicts(with_value(0x12, b"0123 456 789")) == 0
    for bad in (b"12a", b"1-2", b"\xb2"):
        assert verdicts(with_value(0x12, bad)) == orc.SF_NUMERIC
    assert verdicts(with_value(0x16, bytes(range(0, 128)))) == 0                          # IA5: any 7-bit octet
    assert verdicts(with_value(0x16, b"mail@ex\xe4mple")) == orc.SF_IA5
    assert verdicts(with_value(0x14, b"T61 takes \xe4\xff\x00 anything")) == 0            # T61String: as it is
    assert verdicts(with_value(0x1e, b"\xd8\x00")) == 0                                   # BMPString: no Go string on this path
    assert verdicts(with_value(0x04, b"\xff\xfe")) == 0                                   # not a string type


def test_utf8_rule_against_pythons_decoder():
    good = ["", "plain", "Zürich", "東京", "\U0001d518\U0001d52b", "߿ࠀ￿\U00010000\U0010ffff", "a b"]
    for s in good:
        assert verdicts(with_value(0x0c, s.encode("utf-8"))) == 0, s
    bad = [b"\x80", b"\xbf", b"\xc0\x80", b"\xc1\xbf", b"\xe0\x80\x80", b"\xe0\x9f\xbf", b"\xed\xa0\x80", b"\xed\xbf\xbf",
           b"\xf0\x80\x80\x80", b"\xf0\x8f\xbf\xbf", b"\xf4\x90\x80\x80", b"\xf5\x80\x80\x80", b"\xff", b"\xfe",
           b"ab\xc3", b"ab\xe2\x82", b"ab\xf0\x9f\x98", b"\xc3\x28", b"\xe2\x28\xa1", b"\xe2\x82\x28", b"\xf0\x28\x8c\xbc",
           b"\xf8\x88\x80\x80\x80"]
    for v in bad:
        assert verdicts(with_value(0x0c, v)) == orc.SF_UTF8, v
    rng = random.Random(5)
    alphabet = [0x00, 0x41, 0x7f, 0x80, 0x8f, 0x90, 0x9f, 0xa0, 0xbf, 0xc0, 0xc1, 0xc2, 0xdf, 0xe0, 0xe1, 0xec, 0xed, 0xee, 0xef,
                0xf0, 0xf1, 0xf3, 0xf4, 0xf5, 0xff]
    n_bad = 0
    for _ in range(4000):
        v = bytes(rng.choice(alphabet) for _ in range(rng.randrange(0, 9)))
        try:
            v.decode("utf-8")
            py_ok = True
        except UnicodeDecodeError:
            py_ok = False
        n_bad += not py_ok
        assert (verdicts(with_value(0x0c, v)) == 0) == py_ok, v
    assert 1000 < n_bad < 4000


def test_values_in_every_position_and_form():
    """multi-valued RDNs, long-form lengths, values that cross the 4-byte steps of the reader, several findings at once"""
    long_ok = b"x" * 300
    long_bad = b"x" * 299 + b"_"
    assert verdicts(with_value(0x13, long_ok)) == 0
    assert verdicts(with_value(0x13, long_bad)) == orc.SF_PRINTABLE
    for k in range(1, 9):
        assert verdicts(with_value(0x13, b"a" * (k - 1) + b"_")) == orc.SF_PRINTABLE
        assert verdicts(with_value(0x0c, b"a" * (k - 1) + "é".encode("utf-8"))) == 0
        assert verdicts(with_value(0x0c, b"a" * (k - 1) + b"\xc3")) == orc.SF_UTF8
    multi = D.seq(D.tlv(0x31, D.seq(D.oid(0x55, 4, 3), D.tlv(0x13, b"fine")) + D.seq(D.oid(0x55, 4, 10), D.tlv(0x16, b"b\xff"))),
                  D.tlv(0x31, D.seq(D.oid(0x55, 4, 11), D.tlv(0x12, b"12x"))))
    assert verdicts(D.cert(subject=multi)) == orc.SF_IA5 | orc.SF_NUMERIC
    assert verdicts(D.cert(issuer=multi, subject=D.name(D.rdn(3, b"bad\xff", tag=0x0c)))) == orc.SF_IA5 | orc.SF_NUMERIC | orc.SF_UTF8
    # bytes behind the value inside an AttributeTypeAndValue are ignored, whatever they are
    atv = D.seq(D.oid(0x55, 4, 3), D.tlv(0x13, b"ok"), D.tlv(0x13, b"_ignored_"))
    assert verdicts(D.cert(subject=D.seq(D.tlv(0x31, atv)))) == 0


def test_oracle_engine_files_the_finding_as_non_fatal_only_when_asked():
    """X509 entry: kept; precertificate: dropped (any err, ct-fetch.go:202-209); Chain[0] issuer: dropped (:221-225)."""
    iname = D.name(D.rdn(3, b"Good CA"))
    issuer = D.cert(serial=b"\x01", subject=iname, issuer=iname, exts=[D.BC_CA])
    leaf_bad = D.cert(serial=b"\x05", issuer=iname, subject=D.name(D.rdn(3, b"under_score", tag=0x13)))
    leaf_ok = D.cert(serial=b"\x06", issuer=iname)
    bad_issuer = D.cert(serial=b"\x07", subject=iname, issuer=D.name(D.rdn(10, b"caf\xe9", tag=0x0c)), exts=[D.BC_CA])
    for strict in (False, True):
        e = orc.Engine(b"", True, 0)
        e.set_strict_strings(strict)
        st = [e.entry(leaf_bad, issuer, 0)[0], e.entry(leaf_bad, issuer, 1)[0], e.entry(leaf_ok, bad_issuer, 0)[0]]
        if strict:
            assert st == [orc.ST_PASS, orc.ST_PARSE_ERROR, orc.ST_ISSUER_PARSE_ERROR]
        else:
            assert st == [orc.ST_PASS, orc.ST_PASS, orc.ST_PASS]
        e.close()
