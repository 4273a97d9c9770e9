"""The host mirror of the reference's storage package, tested the way the reference tests it
(storage/types_test.go, knowncertificates_test.go, issuermetadata_test.go,
filesystemdatabase_test.go, localdiskbackend_test.go) — here with MockRemoteCache/MockBackend;
tests/test_storage_gpu.py repeats the cache-facing ones against the HBM-backed GpuRemoteCache."""
import calendar
import json
import os

import pytest

from tests import storage_mirror as S
from tests import der as D


def utc(*a):
    return calendar.timegm(tuple(a) + (0,) * (6 - len(a)))


# ---- types_test.go ---------------------------------------------------------------------------
def test_IssuerLazyInit():                                     # :41-57
    i = S.Issuer(spki=b"\xff")
    assert i.id is None
    assert i.ID() == "qBAK5qoZQNC2Y7sxzUZhQuu9vVGHExuS2TgYmHgy64k="
    assert i.id is not None


def test_Serial():                                             # :59-79,103-170
    x, y = S.Serial.FromHex("DEADBEEF"), S.Serial(b"\xde\xad\xbe\xef")
    assert x == y and x.Cmp(y) == 0 and y.String() == "deadbeef"
    assert S.Serial.FromIDString(x.ID()) == x
    with pytest.raises(Exception):
        S.Serial.FromIDString("not base64")
    for h in ("ABCDEF", "001100", "ABCDEF0100101010010101010100101010", "00ABCDEF01001010101010101010010101",
              "FFFFFFFFFFFFFF00F00FFFFFFFFFFFFFFF"):
        s = S.Serial.FromHex(h)
        assert S.Serial.FromBinaryString(s.BinaryString()) == s
        assert S.Serial.UnmarshalJSON(s.MarshalJSON()) == s
    assert S.Serial(bytes.fromhex("CAFEDEAD")).AsBigInt() == 0xCAFEDEAD
    assert S.Serial(b"\x00\xaa").ID() == "AKo="               # :81-101


def test_Log():                                                # :172-201
    assert S.CertificateLog(ShortURL="log.example.com/2525").ID() == "bG9nLmV4YW1wbGUuY29tLzI1MjU="
    assert S.CertificateLog(ShortURL="yeti2021.ct.digicert.com/log/").ID() == "eWV0aTIwMjEuY3QuZGlnaWNlcnQuY29tL2xvZy8="


def test_ExpDate():                                            # :203-252
    for d in ("2004-01-19", "2004-01-19-04", "2004-01-19-23"):
        assert S.ExpDate.Parse(d).ID() == d
    hourless = S.ExpDate.Parse("2004-01-19")
    assert hourless.IsExpiredAt(utc(2004, 1, 20) * 1000)
    assert not hourless.IsExpiredAt(utc(2004, 1, 19, 23, 59, 59) * 1000)
    four = S.ExpDate.Parse("2004-01-19-04")
    assert four.IsExpiredAt(utc(2004, 1, 19, 5) * 1000)
    assert not four.IsExpiredAt(utc(2004, 1, 19, 4, 59, 59) * 1000)
    date = utc(2004, 1, 20, 4, 22, 19)
    e = S.ExpDate.FromTime(date)
    assert e.IsExpiredAt(date * 1000)                           # "expired at its own time"
    assert not e.IsExpiredAt(utc(2004, 1, 20) * 1000 - 1)
    assert e.ID() == "2004-01-20-04" and e.ExpireTime() == utc(2004, 1, 20, 4)
    with pytest.raises(ValueError):
        S.ExpDate.Parse("garbage")


def test_ParseUniqueCertIdentifier():                          # :254-269
    with pytest.raises(ValueError):
        S.UniqueCertIdentifier.Parse("a::b")
    expected = "2019-04-28-22::an issuer::AESq_w=="
    assert S.UniqueCertIdentifier.Parse(expected).String() == expected


# ---- knowncertificates_test.go ------------------------------------------------------------------
def known_certificates_suite(cache):
    testIssuer = S.Issuer.FromString("test issuer")
    kc = S.KnownCertificates(S.ExpDate.Parse("2029-01-30"), testIssuer, cache)
    for h in ("01", "02", "03", "04"):
        cache.SetInsert(kc.serialId(), S.Serial.FromHex(h).BinaryString())
    for h in ("01", "02", "03", "04"):
        assert kc.WasUnknown(S.Serial.FromHex(h)) is False
    assert kc.WasUnknown(S.Serial.FromHex("05")) is True
    assert kc.WasUnknown(S.Serial.FromHex("05")) is False
    got = [m.decode("latin1") for m in cache.SetList(kc.serialId())]
    assert json.dumps(got) == '["\\u0001", "\\u0002", "\\u0003", "\\u0004", "\\u0005"]'
    assert sorted(kc.Known()) == [S.Serial.FromHex(h) for h in ("01", "02", "03", "04", "05")]
    assert kc.Count() == 5


def expire_at_suite(cache, expirations):
    date = utc(2004, 1, 20, 4, 22, 19)
    kc = S.KnownCertificates(S.ExpDate.FromTime(date), S.Issuer.FromString("test issuer"), cache)
    assert kc.WasUnknown(S.Serial.FromHex("05")) is True
    assert kc.serialId() == "serials::2004-01-20-04::test issuer"
    assert expirations() == {b"serials::2004-01-20-04::test issuer": utc(2004, 1, 20, 4)}


def test_Unknown_and_Known_mock():
    known_certificates_suite(S.MockRemoteCache())


def test_ExpireAt_mock():
    c = S.MockRemoteCache()
    expire_at_suite(c, lambda: c.Expirations)


# ---- issuermetadata_test.go ------------------------------------------------------------------------
def duplicate_crls_suite(cache):                               # :16-60
    meta = S.IssuerMetadata(S.Issuer.FromString("issuer"), cache)
    meta.addCRL("ldaps://ldap.crl")
    meta.addCRL("schema://192.168.1.1:129/file.crl")
    meta.addCRL("http://::1/file.crl")
    assert len(meta.CRLs()) == 1
    for v in ("http://::1/file.crl", "http://::1/file.crl ", " http://::1/file.crl ", " http://::1/file.crl   "):
        meta.addCRL(v)
        assert len(meta.CRLs()) == 1
    # url.String(): the same vectors as Suite_DuplicateCRLs in tests/host/test_storage.cpp — both mirrors produce the
    # same set members
    meta.addCRL("HTTP://Example.com/a b.crl")                  # scheme lower-cased, path escaped
    assert sorted(meta.CRLs()) == ["http://::1/file.crl", "http://Example.com/a%20b.crl"]
    meta.addCRL("http://host:bad/x.crl")                       # invalid port: url.Parse error, ignored
    meta.addCRL("http://host/%zz")                             # invalid escape, ignored
    meta.addCRL("http://host/a\tb.crl")                        # control character inside: url.Parse error
    meta.addCRL("no-scheme/file.crl")
    assert len(meta.CRLs()) == 2


def accumulate_suite(cache):                                   # :100-136
    issuerCN = b"My First Issuer (tm)"
    mk = lambda serial: S.HostCert(D.cert(serial=serial, issuer=D.name(D.rdn(3, issuerCN)),
                                          not_after=D.utctime("010101000000Z")))
    meta = S.IssuerMetadata(S.Issuer.FromString("issuer"), cache)
    hour = utc(2001, 1, 1) // 3600
    assert meta.Accumulate(mk(b"\x00"), hour) is False          # a new day
    assert meta.Accumulate(mk(b"\x01"), hour) is True
    assert meta.CRLs() == []
    assert meta.Issuers() == ["CN=My First Issuer (tm)"]


def test_DuplicateCRLs_mock():
    duplicate_crls_suite(S.MockRemoteCache())


def test_Accumulate_mock():
    accumulate_suite(S.MockRemoteCache())


def test_issuer_dn_string_and_crl_extraction(golden_certs):
    from ct_mapreduce_amd import synth
    cfg = synth.config(n_issuers=40)
    der, iss, _ = synth.leaf(cfg, 7)
    c = S.HostCert(der)
    assert c.issuer_string() == "CN=Synth Issuer %03d,O=Synth CA Org,C=US" % iss
    assert c.crl_dps == ["http://crl.synth-%03d.example/ca.crl" % iss]
    real = S.HostCert(golden_certs["kRealSPKI"])
    assert real.crl_dps == ["http://public.wisekey.com/crl/wcidsg1ca.crl"]
    assert real.issuer_string() == ("CN=WISeKey CertifyID Standard G1 CA,OU=Copyright (c) 2005 WISeKey SA+"
                                    "OU=International,O=WISeKey,C=CH")
    assert S.HostCert(D.cert(issuer=D.name(D.rdn(3, b'a,b+c "q"'), D.rdn(10, b" x "))))\
        .issuer_string() == 'CN=a\\,b\\+c \\"q\\",O=\\ x\\ '
    # attribute types outside the nine FillFromRDNSequence knows (DC, emailAddress) and values that are no Go string
    # (BMPString) are DROPPED by the pinned CT-go's Name.String(): the issuer:: set member
    # of a reference deployment does not show them
    dc = D.tlv(0x31, D.seq(D.oid(0x09, 0x92, 0x26, 0x89, 0x93, 0xf2, 0x2c, 0x64, 0x01, 0x19), D.tlv(0x16, b"example")))
    mail = D.tlv(0x31, D.seq(D.oid(0x2a, 0x86, 0x48, 0x86, 0xf7, 0x0d, 0x01, 0x09, 0x01), D.tlv(0x16, b"ca@example.org")))
    odd = D.name(dc, D.rdn(6, b"DE", tag=0x13), D.rdn(10, b"\x00O\x00r\x00g", tag=0x1e), D.rdn(10, b"Org"), mail, D.rdn(3, b"The CA"))
    assert S.HostCert(D.cert(issuer=odd)).issuer_string() == "CN=The CA,O=Org,C=DE"


# ---- filesystemdatabase_test.go (cache-facing parts) --------------------------------------------------
def issuer_and_dates_suite(db):                                # :218-279
    assert db.GetIssuerAndDatesFromCache() == []
    issuer = S.Issuer.FromString("Honesty Issuer")
    db.GetKnownCertificates(S.ExpDate.Parse("2040-02-03-19"), issuer).WasUnknown(S.Serial.FromHex("FEEDBEEF"))
    l2 = db.GetIssuerAndDatesFromCache()
    assert len(l2) == 1 and len(l2[0].ExpDates) == 1
    db.GetKnownCertificates(S.ExpDate.Parse("2040-02-03"), issuer).WasUnknown(S.Serial.FromHex("BEEF"))
    l3 = db.GetIssuerAndDatesFromCache()
    assert len(l3) == 1 and len(l3[0].ExpDates) == 2


def log_state_suite(cache, db):                                # :281-340
    log = db.GetLogState("go.pher", "")
    assert log is not None
    log = db.GetLogState("log.ct", "/2019")
    assert log.ShortURL == "log.ct/2019" and log.MaxEntry == 0 and log.LastEntryTime == 0
    log.MaxEntry = 9
    db.SaveLogState(log)
    assert cache.LoadLogState(log.ShortURL) == log
    upd = db.GetLogState("log.ct", "/2019")
    assert upd.MaxEntry == 9 and upd.LastEntryTime == 0


def test_GetIssuerAndDatesFromCache_mock():
    issuer_and_dates_suite(S.FilesystemDatabase(S.MockBackend(), S.MockRemoteCache()))


def test_LogState_mock_and_noop():
    c = S.MockRemoteCache()
    log_state_suite(c, S.FilesystemDatabase(S.MockBackend(), c))
    c = S.MockRemoteCache()
    log_state_suite(c, S.FilesystemDatabase(S.NoopBackend(), c))


def test_ListExpiration():                                     # :132-216
    be = S.MockBackend()
    db = S.FilesystemDatabase(be, S.MockRemoteCache())
    for d in ("2017-11-28", "2018-11-28", "2019-11-28"):
        be.AllocateExpDateAndIssuer(S.ExpDate.Parse(d), S.Issuer.FromString("test issuer"))
    ids = lambda t: sorted(e.ID() for e in db.ListExpirationDates(t))
    assert ids(utc(2016, 11, 29, 15, 4, 5)) == ["2017-11-28", "2018-11-28", "2019-11-28"]
    assert ids(utc(2018, 11, 29, 15, 4, 5)) == ["2019-11-28"]
    assert ids(utc(2019, 11, 28, 1, 4, 5)) == ["2019-11-28"]
    assert ids(utc(2020, 11, 29, 15, 4, 5)) == []
    assert ids(utc(2018, 11, 28, 23, 59, 59)) == ["2018-11-28", "2019-11-28"]


def test_NoopBackend():                                        # :355-377
    db = S.FilesystemDatabase(S.NoopBackend(), S.MockRemoteCache())
    assert db.markDirty("0001-01-01") is None
    with pytest.raises(RuntimeError):
        db.ListExpirationDates(0)
    with pytest.raises(RuntimeError):
        db.ListIssuersForExpirationDate(S.ExpDate.Parse("2040-02-03"))


# ---- localdiskbackend_test.go ------------------------------------------------------------------------
def test_LocalDisk_KnownCertificateList_and_LogState_and_layout(tmp_path):
    root = str(tmp_path / "root")
    db = S.LocalDiskBackend(0o644, root)
    issuer = S.Issuer.FromString("issuerAKI")
    db.StoreKnownCertificateList(issuer, [S.Serial.FromHex(h) for h in ("01", "02", "03")])
    assert open(os.path.join(root, issuer.ID()), "rb").read() == bytes.fromhex("30310A30320A30330A")   # :60-85
    log = db.LoadLogState("log.ct/2019")
    assert log.ShortURL == "log.ct/2019" and log.MaxEntry == 0
    log.MaxEntry, log.LastEntryTime = 0xDEADBEEF, 1567016306
    db.StoreLogState(log)
    got = db.LoadLogState("log.ct/2019")
    assert got.MaxEntry == 0xDEADBEEF and got.LastEntryTime == 1567016306
    assert os.path.exists(os.path.join(root, "state", S.CertificateLogIDFromShortURL("log.ct/2019")))
    # certificate path = root/expDateID/issuerID/serialID — no ".pem" suffix (localdiskbackend.go:194-199)
    e = S.ExpDate.Parse("2019-11-28-04")
    db.StoreCertificatePEM(S.Serial.FromHex("02"), e, issuer, b"\xda\xda")
    assert open(os.path.join(root, "2019-11-28-04", "issuerAKI", "Ag=="), "rb").read() == b"\xda\xda"
    # MarkDirty is relative to the CURRENT directory (localdiskbackend.go:89-91)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        db.MarkDirty("2019-11-28")
        assert open(tmp_path / "2019-11-28" / "dirty", "rb").read() == b"\x00"
    finally:
        os.chdir(cwd)


def test_pem_encode_matches_reference_fixture(golden_certs, golden_dir):
    for n, der in golden_certs.items():
        want = open(os.path.join(golden_dir, n + ".pem"), "rb").read().strip() + b"\n"
        assert S.pem_encode(der) == want


def test_redis_dump_is_exact_resp_and_round_trips():
    """N4: the sets as a Redis protocol stream (SADD + the EXPIREAT of knowncertificates.go:98-104) and back."""
    import io
    c = S.MockRemoteCache()
    c.SetInsert("serials::2020-02-05-00::iss", b"\x00\xaa")            # the raw serial octets of kLeadingZeroes
    c.SetInsert("serials::2020-02-05-00::iss", b"\x01")
    c.SetInsert("crl::iss", b"http://a/b.crl")
    buf = io.BytesIO()
    assert S.redis_dump(c, buf) == {"keys": 2, "members": 3}
    key = b"serials::2020-02-05-00::iss"
    assert buf.getvalue() == (
        b"*4\r\n$4\r\nSADD\r\n$%d\r\n%s\r\n$2\r\n\x00\xaa\r\n$1\r\n\x01\r\n" % (len(key), key) +
        b"*3\r\n$8\r\nEXPIREAT\r\n$%d\r\n%s\r\n$10\r\n%d\r\n" % (len(key), key, utc(2020, 2, 5)) +
        b"*3\r\n$4\r\nSADD\r\n$8\r\ncrl::iss\r\n$14\r\nhttp://a/b.crl\r\n")
    d = S.MockRemoteCache()
    assert S.redis_load(d, io.BytesIO(buf.getvalue())) == {"commands": 3, "inserted": 3}
    assert d.Data == c.Data and d.Expirations == {key: utc(2020, 2, 5)}
    # many members: several SADD commands per key, nothing lost, members with CR/LF survive
    big = S.MockRemoteCache()
    for i in range(1300):
        big.SetInsert("issuer::x", b"\r\n%d\r\n" % i)
    buf = io.BytesIO()
    S.redis_dump(big, buf, members_per_command=512)
    assert buf.getvalue().count(b"SADD") == 3
    back = S.MockRemoteCache()
    S.redis_load(back, io.BytesIO(buf.getvalue()))
    assert back.Data == big.Data
    with pytest.raises(ValueError):
        S.redis_load(back, io.BytesIO(b"*1\r\n$4\r\nPING\r\n"))
