"""TEST SCAFFOLDING (moved out of the product package in round 3): a Python twin of the reference's `storage` package
over the C ABI — same names, argument meaning and error behaviour, so that the end-to-end tests read like the reference's
own (storage/*_test.go).  The host mirror a cgo binding would look like is the C++ one, include/ctmr_storage.hpp (with the
reference's test suites in tests/host/); what the PRODUCT keeps in Python is ct_mapreduce_amd/remote_cache.py
(GpuRemoteCache + the Redis-protocol export/import of N4).

  types        Issuer, SPKI, Serial, ExpDate, UniqueCertIdentifier, CertificateLog   storage/types.go
  RemoteCache  GpuRemoteCache (libctmr: serials sets in HBM) / MockRemoteCache       storage/types.go:83-102, mockcache.go
  KnownCertificates, IssuerMetadata                                                  storage/knowncertificates.go, issuermetadata.go
  StorageBackend: NoopBackend, MockBackend, LocalDiskBackend                         storage/{noop,mock,localdisk}backend.go
  FilesystemDatabase (Store / StoreBatch = batched insertCTWorker + Store)           storage/filesystemdatabase.go

The per-entry hot path (parse, filters, WasUnknown) runs on the GPU through Engine.map_batch; what
stays here is what the reference also does only for *newly unknown* certificates (IssuerMetadata,
PEM write-back, dirty markers) plus bookkeeping strings.  Nothing here calls the oracle.
"""
import base64
import calendar
import datetime as _dt
import fnmatch
import hashlib
import json
import os
import time as _time

import numpy as np

from ct_mapreduce_amd import _native as N
from ct_mapreduce_amd.engine import Batch, Engine
from ct_mapreduce_amd import remote_cache as _rc
from ct_mapreduce_amd.remote_cache import redis_dump, redis_load  # noqa: F401  (re-exported for the tests)

kExpirationFormat = "%Y-%m-%d"
kExpirationFormatWithHour = "%Y-%m-%d-%H"


# --------------------------------------------------------------------------------------- types.go
class SPKI:
    def __init__(self, spki: bytes):
        self.spki = bytes(spki)

    def ID(self):
        return base64.urlsafe_b64encode(self.spki).decode()

    def String(self):
        return self.spki.hex()

    def Sha256DigestURLEncodedBase64(self):       # types.go:155-159
        return base64.urlsafe_b64encode(hashlib.sha256(self.spki).digest()).decode()


class Issuer:
    """types.go:104-143.  id is lazily b64url(SHA-256(RawSubjectPublicKeyInfo))."""

    def __init__(self, spki: bytes = None, id: str = None):
        self.id = id
        self.spki = SPKI(spki) if spki is not None else None

    @staticmethod
    def FromString(s):
        return Issuer(id=s)

    def ID(self):
        if self.id is None:
            self.id = self.spki.Sha256DigestURLEncodedBase64()
        return self.id

    def __eq__(self, o):
        return isinstance(o, Issuer) and self.ID() == o.ID()

    def __hash__(self):
        return hash(self.ID())

    def __repr__(self):
        return f"Issuer({self.ID()})"


class Serial:
    """types.go:161-246 — the raw INTEGER content octets, leading zeroes preserved."""

    def __init__(self, b: bytes):
        self.serial = bytes(b)

    @staticmethod
    def FromHex(s):
        return Serial(bytes.fromhex(s))          # panics (raises) on bad hex like the reference

    @staticmethod
    def FromIDString(s):
        return Serial(base64.urlsafe_b64decode(s.encode()))

    @staticmethod
    def FromBinaryString(s: bytes):
        return Serial(s)

    def ID(self):
        return base64.urlsafe_b64encode(self.serial).decode()

    def String(self):
        return self.HexString()

    def HexString(self):
        return self.serial.hex()

    def BinaryString(self):
        return self.serial

    def Cmp(self, o):
        return (self.serial > o.serial) - (self.serial < o.serial)

    def MarshalJSON(self):
        return json.dumps(self.HexString())

    @staticmethod
    def UnmarshalJSON(data: str):
        if not (data.startswith('"') and data.endswith('"')):
            raise ValueError("Expected surrounding quotes")
        return Serial(bytes.fromhex(data[1:-1]))

    def AsBigInt(self):
        return int.from_bytes(self.serial, "big")

    def __eq__(self, o):
        return isinstance(o, Serial) and self.serial == o.serial

    def __hash__(self):
        return hash(self.serial)

    def __lt__(self, o):
        return self.serial < o.serial

    def __repr__(self):
        return f"Serial({self.serial.hex()})"


def _utc_ms(y, mo, d, h=0, mi=0, s=0, ms=0):
    return calendar.timegm((y, mo, d, h, mi, s)) * 1000 + ms


class ExpDate:
    """types.go:333-405.  Times are integer milliseconds since the epoch (UTC)."""

    def __init__(self, date_ms, last_good_ms, hour_resolution):
        self.date = date_ms
        self.lastGood = last_good_ms
        self.hourResolution = hour_resolution

    @staticmethod
    def FromTime(unix_seconds):                   # NewExpDateFromTime :339-346
        trunc = (int(unix_seconds) // 3600) * 3600 * 1000
        return ExpDate(trunc, trunc - 1, True)

    @staticmethod
    def FromHour(exp_hour):                       # the device encoding of the same thing
        return ExpDate.FromTime(int(exp_hour) * 3600)

    @staticmethod
    def Parse(s):                                 # NewExpDate :348-367
        if len(s) > 10:
            try:
                t = _dt.datetime.strptime(s, kExpirationFormatWithHour)
                if t.strftime(kExpirationFormatWithHour) == s or True:
                    ms = _utc_ms(t.year, t.month, t.day, t.hour)
                    return ExpDate(ms, ms + 3600_000 - 1, True)
            except ValueError:
                pass
        t = _dt.datetime.strptime(s, kExpirationFormat)      # raises ValueError like the Go error
        ms = _utc_ms(t.year, t.month, t.day)
        return ExpDate(ms, ms + 24 * 3600_000 - 1, False)

    def IsExpiredAt(self, t_ms):
        return self.lastGood < t_ms

    def ExpireTime(self):
        return self.date // 1000

    def hour(self):
        return self.date // 3600_000

    def ID(self):
        t = _time.gmtime(self.date // 1000)
        if self.hourResolution:
            return "%04d-%02d-%02d-%02d" % (t.tm_year, t.tm_mon, t.tm_mday, t.tm_hour)
        return "%04d-%02d-%02d" % (t.tm_year, t.tm_mon, t.tm_mday)

    String = ID

    def __eq__(self, o):
        return isinstance(o, ExpDate) and self.ID() == o.ID()

    def __hash__(self):
        return hash(self.ID())

    def __lt__(self, o):
        return self.date < o.date

    def __repr__(self):
        return f"ExpDate({self.ID()})"


class UniqueCertIdentifier:
    def __init__(self, expDate, issuer, serial):
        self.ExpDate, self.Issuer, self.SerialNum = expDate, issuer, serial

    @staticmethod
    def Parse(s):                                 # types.go:286-311
        parts = s.split("::")
        if len(parts) != 3:
            raise ValueError("Expected 3 parts, got %d" % len(parts))
        return UniqueCertIdentifier(ExpDate.Parse(parts[0]), Issuer.FromString(parts[1]),
                                    Serial.FromIDString(parts[2]))

    def String(self):
        return "%s::%s::%s" % (self.ExpDate.ID(), self.Issuer.ID(), self.SerialNum.ID())


def CertificateLogIDFromShortURL(shortURL):
    return base64.urlsafe_b64encode(shortURL.encode()).decode()


class CertificateLog:
    """types.go:25-44.  Times are unix seconds (0 = Go's zero time)."""

    def __init__(self, ShortURL="", MaxEntry=0, LastEntryTime=0, LastUpdateTime=0):
        self.ShortURL, self.MaxEntry = ShortURL, MaxEntry
        self.LastEntryTime, self.LastUpdateTime = LastEntryTime, LastUpdateTime

    def ID(self):
        return CertificateLogIDFromShortURL(self.ShortURL)

    def to_json(self):
        return json.dumps({"ShortURL": self.ShortURL, "MaxEntry": self.MaxEntry,
                           "LastEntryTime": self.LastEntryTime, "LastUpdateTime": self.LastUpdateTime})

    @staticmethod
    def from_json(s):
        d = json.loads(s)
        return CertificateLog(d["ShortURL"], d["MaxEntry"], d["LastEntryTime"], d["LastUpdateTime"])

    def __eq__(self, o):
        return isinstance(o, CertificateLog) and self.__dict__ == o.__dict__


# ---------------------------------------------------------------------------------- RemoteCache
class RemoteCache:
    """storage/types.go:83-102."""

    def Exists(self, key): raise NotImplementedError
    def SetInsert(self, key, entry): raise NotImplementedError
    def SetRemove(self, key, entry): raise NotImplementedError
    def SetContains(self, key, entry): raise NotImplementedError
    def SetList(self, key): raise NotImplementedError
    def SetToChan(self, key): raise NotImplementedError
    def SetCardinality(self, key): raise NotImplementedError
    def ExpireAt(self, key, unix_seconds): raise NotImplementedError
    def KeysToChan(self, pattern): raise NotImplementedError
    def StoreLogState(self, log): raise NotImplementedError
    def LoadLogState(self, shortUrl): raise NotImplementedError


def _b(x):
    return x.encode() if isinstance(x, str) else bytes(x)


class MockRemoteCache(RemoteCache):
    """storage/mockcache.go: sorted-slice sets; used by the CPU tests of the host logic."""

    def __init__(self):
        self.Data = {}
        self.Expirations = {}
        self.Duplicate = 0

    def SetInsert(self, key, entry):
        key, entry = _b(key), _b(entry)
        lst = self.Data.setdefault(key, [])
        import bisect
        i = bisect.bisect_left(lst, entry)
        if i < len(lst) and lst[i] == entry:
            return False
        lst.insert(i, entry)
        return True

    def SetRemove(self, key, entry):
        key, entry = _b(key), _b(entry)
        lst = self.Data.get(key, [])
        if entry in lst:
            lst.remove(entry)
            return True
        return False

    def SetContains(self, key, entry):
        return _b(entry) in self.Data.get(_b(key), [])

    def SetList(self, key):
        return list(self.Data.get(_b(key), []))

    def SetToChan(self, key):
        for _ in range(self.Duplicate + 1):
            for v in self.Data.get(_b(key), []):
                yield v

    def SetCardinality(self, key):
        return len(self.Data.get(_b(key), []))

    def Exists(self, key):
        return _b(key) in self.Data

    def ExpireAt(self, key, unix_seconds):
        self.Expirations[_b(key)] = unix_seconds

    def KeysToChan(self, pattern):
        pat = _b(pattern).decode("latin1")
        for k in list(self.Data):
            if fnmatch.fnmatchcase(k.decode("latin1"), pat):
                yield k

    def StoreLogState(self, log):
        self.Data[_b(log.ShortURL)] = [log.to_json().encode()]

    def LoadLogState(self, shortUrl):
        d = self.Data.get(_b(shortUrl))
        if d is None:
            raise KeyError("Log state not found")
        if len(d) != 1:
            raise ValueError("Unexpected number of log states")
        return CertificateLog.from_json(d[0].decode())


class GpuRemoteCache(_rc.GpuRemoteCache, RemoteCache):
    """The product's cache (ct_mapreduce_amd/remote_cache.py) with the typed log-state methods of the mirror."""

    def StoreLogState(self, log):                 # rediscache.go:180-190: key "log::<shortURL>"
        self.StoreLogStateJSON(log.ShortURL, log.to_json().encode())

    def LoadLogState(self, shortUrl):
        return CertificateLog.from_json(self.LoadLogStateJSON(shortUrl).decode())


# -------------------------------------------------------------------------- KnownCertificates
kSerials = "serials"


class KnownCertificates:
    """storage/knowncertificates.go."""

    def __init__(self, expDate: ExpDate, issuer: Issuer, cache: RemoteCache):
        self.expDate, self.issuer, self.cache = expDate, issuer, cache
        self.expirySet = False

    def id(self, *params):
        return "%s%s::%s" % (self.expDate.ID(), "".join(params), self.issuer.ID())

    def serialId(self, *params):
        return "%s::%s" % (kSerials, self.id(*params))

    def WasUnknown(self, serial: Serial) -> bool:             # :38-55
        result = self.cache.SetInsert(self.serialId(), serial.BinaryString())
        if not self.expirySet:
            self.cache.ExpireAt(self.serialId(), self.expDate.ExpireTime())   # :98-104
            self.expirySet = True
        return result

    def Count(self) -> int:                                    # :57-63
        return self.cache.SetCardinality(self.serialId())

    def Known(self):                                           # :65-96 (dedups what SetToChan yields)
        return [Serial.FromBinaryString(s) for s in set(self.cache.SetToChan(self.serialId()))]


# ----------------------------------------------------------------------------- IssuerMetadata
kIssuers = "issuer"
kCrls = "crl"

_ATTR_NAMES = {3: "CN", 5: "SERIALNUMBER", 6: "C", 7: "L", 8: "ST", 9: "STREET", 10: "O", 11: "OU",
               17: "POSTALCODE"}
_ORDER = [6, 8, 7, 9, 17, 10, 11, 3, 5]     # pkix.Name.ToRDNSequence: C, ST, L, STREET, POSTALCODE, O, OU, CN, SERIALNUMBER


def _tlv(d, p):
    tag = d[p]
    b = d[p + 1]
    if b < 0x80:
        return tag, p + 2, p + 2 + b
    n = b & 0x7f
    ln = int.from_bytes(d[p + 2:p + 2 + n], "big")
    return tag, p + 2 + n, p + 2 + n + ln


def _children(d, s, e):
    out = []
    while s < e:
        tag, cs, ce = _tlv(d, s)
        out.append((tag, cs, ce))
        s = ce
    return out


def _escape_rdn_value(v: str) -> str:
    out = []
    for k, c in enumerate(v):
        esc = c in ',+"\\<>;' or (k == 0 and c in " #") or (k == len(v) - 1 and c == " ")
        out.append("\\" + c if esc else c)
    return "".join(out)


class HostCert:
    """The few fields the host-only branch needs from a *newly unknown* certificate
    (issuermetadata.go:92-138): Issuer.String() and CRLDistributionPoints.  Pure Python: this is
    the rare path; the per-entry parse is the GPU's."""

    def __init__(self, der: bytes):
        d = der
        _, cs, ce = _tlv(d, 0)
        _, ts, te = _tlv(d, cs)
        kids = _children(d, ts, te)
        k = 1 if kids[0][0] == 0xa0 else 0
        self.serial = d[kids[k][1]:kids[k][2]]
        issuer = kids[k + 2]
        validity = kids[k + 3]
        self.issuer_atvs = []
        for (_, ss, se) in _children(d, issuer[1], issuer[2]):
            for (_, a_s, a_e) in _children(d, ss, se):
                (ot, os_, oe), (vt, vs, ve) = _children(d, a_s, a_e)[:2]
                self.issuer_atvs.append((bytes(d[os_:oe]), vt, bytes(d[vs:ve])))
        times = _children(d, validity[1], validity[2])
        self.not_after_raw = bytes(d[times[1][1]:times[1][2]])
        self.crl_dps = []
        for (tag, s, e) in kids[k + 6:]:
            if tag != 0xa3:
                continue
            _, es, ee = _tlv(d, s)
            for (_, xs, xe) in _children(d, es, ee):
                parts = _children(d, xs, xe)
                if bytes(d[parts[0][1]:parts[0][2]]) != b"\x55\x1d\x1f":
                    continue
                # asn1.Unmarshal(value, &[]distributionPoint): positional struct fields, the WHOLE value must unmarshal or
                # there are no distribution points at all (round 6, ADVICE r05: this mirror kept the URIs of a value whose
                # later part was malformed).  The rules are the checker's (this file is test scaffolding): orc_cert_meta.
                from oracle import oracle as _orc
                meta = _orc.cert_meta(bytes(der))
                if meta is not None and not meta[2].bad_crl:
                    self.crl_dps += [u.decode("latin1") for u in meta[1]]

    def issuer_string(self) -> str:
        return name_string(self.issuer_atvs)


def name_atvs(name_der: bytes):
    """AttributeTypeAndValues of a Name TLV (the bytes a CTMR_MK_DN item carries)."""
    d = name_der
    _, cs, ce = _tlv(d, 0)
    out = []
    for (_, ss, se) in _children(d, cs, ce):
        for (_, a_s, a_e) in _children(d, ss, se):
            (ot, os_, oe), (vt, vs, ve) = _children(d, a_s, a_e)[:2]
            out.append((bytes(d[os_:oe]), vt, bytes(d[vs:ve])))
    return out


_GO_STRING_TAGS = (0x0c, 0x12, 0x13, 0x14, 0x16)     # UTF8, Numeric, Printable, T61, IA5: what Go's asn1 decodes to a `string`


def name_string(atvs) -> str:
    """pkix.Name.String() of certificate-transparency-go v1.1.0 (go.mod:10) = ToRDNSequence().String(): the Name is
    rebuilt from the nine typed fields FillFromRDNSequence fills (C, ST, L, STREET, POSTALCODE, O, OU, CN,
    SERIALNUMBER; string-typed values only) plus ExtraNames, which a parsed certificate never has — every other
    attribute type (DC, emailAddress, organizationIdentifier, jurisdiction*, …) is DROPPED.  (Go >= 1.15's
    crypto/x509/pkix appends them as `oid=#hex`; the pinned fork predates that.)  RDNs are printed in reverse, the
    values of one type joined with '+'."""
    named = {}
    for oid, vt, val in atvs:
        if len(oid) == 3 and oid[:2] == b"\x55\x04" and oid[2] in _ATTR_NAMES and vt in _GO_STRING_TAGS:
            named.setdefault(oid[2], []).append(val.decode("utf-8", "replace"))
    rdns = []
    for a in _ORDER:
        if a in named:
            rdns.append("+".join(_ATTR_NAMES[a] + "=" + _escape_rdn_value(x) for x in named[a]))
    return ",".join(reversed(rdns))


def go_url_normalise(raw: str):
    """url.Parse(strings.TrimSpace(s)) + url.String() for the shapes a CRL distribution point takes — the same rules
    as go_url_normalise in include/ctmr_storage.hpp (one set of vectors pins both: tests/test_storage_cpu.py,
    tests/host/test_storage.cpp): scheme lower-cased; control characters, malformed %-escapes and a non-numeric port
    are parse errors; path and fragment bytes outside Go's unescaped set are %XX-escaped by String().
    Returns (ok, scheme, normalised)."""
    s = raw.strip(" \t\n\v\f\r")
    b = s.encode("utf-8", "surrogateescape")
    if any(c < 0x20 or c == 0x7f for c in b):
        return False, "", ""
    i = 0
    while i < len(s):
        c = s[i]
        if c.isascii() and (c.isalpha() or (i > 0 and (c.isdigit() or c in "+-."))):
            i += 1
            continue
        break
    if i == 0 or i >= len(s) or s[i] != ":":
        return True, "", s                                     # no scheme
    scheme = s[:i].lower()
    rest, frag, query = s[i + 1:], None, None
    if "#" in rest:
        rest, frag = rest.split("#", 1)
    if "?" in rest:
        rest, query = rest.split("?", 1)

    def bad_escape(t):
        hexd = "0123456789abcdefABCDEF"
        return any(t[k] == "%" and (k + 2 >= len(t) or t[k + 1] not in hexd or t[k + 2] not in hexd)
                   for k in range(len(t)))
    if bad_escape(rest) or bad_escape(frag or ""):
        return False, "", ""
    authority, path = None, rest
    if rest.startswith("//"):
        sl = rest.find("/", 2)
        authority = rest[2:] if sl < 0 else rest[2:sl]
        path = "" if sl < 0 else rest[sl:]
        host = authority.rsplit("@", 1)[-1]
        if not host.startswith("["):                           # validOptionalPort on what follows the last ':'
            if ":" in host and not all("0" <= ch <= "9" for ch in host.rsplit(":", 1)[1]):
                return False, "", ""

    def esc(t, is_path):
        out = []
        for c in t.encode("utf-8", "surrogateescape"):
            ch = chr(c)
            keep = (c < 0x80 and ch.isalnum()) or (c != 0 and ch in "-_.~$&+,/:;=@%!*'()") or (not is_path and ch == "?")
            out.append(ch if keep else "%%%02X" % c)
        return "".join(out)
    norm = scheme + ":"
    if authority is not None:
        norm += "//" + authority
    norm += esc(path, True)
    if query is not None:
        norm += "?" + query
    if frag is not None:
        norm += "#" + esc(frag, False)
    return True, scheme, norm


class IssuerMetadata:
    """storage/issuermetadata.go."""

    def __init__(self, issuer: Issuer, cache: RemoteCache):
        self.issuer, self.cache = issuer, cache
        self.knownCrlDPs, self.knownIssuerDNs, self.knownExpDates = set(), set(), set()

    def id(self): return self.issuer.ID()
    def crlId(self): return "%s::%s" % (kCrls, self.id())
    def issuersId(self): return "%s::%s" % (kIssuers, self.id())

    def addCRL(self, aCRL: str):                               # :48-73
        ok, scheme, norm = go_url_normalise(aCRL)
        if not ok:
            return                                             # "Not a valid CRL DP URL"
        if scheme in ("ldap", "ldaps"):
            return
        if scheme not in ("http", "https"):
            return                                             # "Ignoring unknown CRL scheme"
        self.cache.SetInsert(self.crlId(), norm)

    def addIssuerDN(self, dn: str):                            # :75-87
        self.cache.SetInsert(self.issuersId(), dn)

    def Accumulate(self, cert: HostCert, exp_hour: int) -> bool:   # :92-138
        expID = ExpDate.FromHour(exp_hour).ID()
        dn = cert.issuer_string()
        seenExpDateBefore = expID in self.knownExpDates
        seenIssuerDn = dn in self.knownIssuerDNs
        self.knownExpDates.add(expID)
        for dp in cert.crl_dps:
            if dp not in self.knownCrlDPs:
                self.knownCrlDPs.add(dp)
                self.addCRL(dp)
        if not seenIssuerDn:
            self.knownIssuerDNs.add(dn)
            self.addIssuerDN(dn)
        return seenExpDateBefore

    def Issuers(self): return [x.decode() for x in self.cache.SetList(self.issuersId())]
    def CRLs(self): return [x.decode() for x in self.cache.SetList(self.crlId())]


# ----------------------------------------------------------------------------- StorageBackend
class NoopBackend:
    """storage/noopbackend.go: stores succeed silently, loads/listings error."""

    def _err(self): return RuntimeError("Unable to load from the NoopBackend.")
    def MarkDirty(self, id): return None
    def AllocateExpDateAndIssuer(self, expDate, issuer): return None
    def StoreCertificatePEM(self, serial, expDate, issuer, b): return None
    def StoreLogState(self, log): return None
    def StoreKnownCertificateList(self, issuer, serials): return None
    def LoadCertificatePEM(self, serial, expDate, issuer): raise self._err()
    def LoadLogState(self, logURL): raise self._err()
    def ListExpirationDates(self, notBefore): raise self._err()
    def ListIssuersForExpirationDate(self, expDate): raise self._err()
    def ListSerialsForExpirationDateAndIssuer(self, expDate, issuer): raise self._err()


class MockBackend:
    """storage/mockbackend.go."""

    def __init__(self):
        self.expDateToIssuer, self.expDateIssuerIDToSerials, self.store = {}, {}, {}
        self.dirty = []

    def MarkDirty(self, id):
        self.dirty.append(id)

    def AllocateExpDateAndIssuer(self, expDate, issuer):
        lst = self.expDateToIssuer.setdefault(expDate.ID(), [])
        if issuer.ID() not in [i.ID() for i in lst]:
            lst.append(issuer)
            lst.sort(key=lambda i: i.ID())

    def StoreCertificatePEM(self, serial, expDate, issuer, b):
        self.store["pem" + expDate.ID() + issuer.ID() + serial.ID()] = b
        self.expDateIssuerIDToSerials.setdefault(expDate.ID() + issuer.ID(), []).append(serial)

    def StoreLogState(self, log):
        self.store["logstate" + log.ShortURL] = log.to_json()

    def StoreKnownCertificateList(self, issuer, serials):
        self.store[issuer.ID()] = json.dumps([s.HexString() for s in serials])

    def LoadCertificatePEM(self, serial, expDate, issuer):
        k = "pem" + expDate.ID() + issuer.ID() + serial.ID()
        if k in self.store:
            return self.store[k]
        raise KeyError("Couldn't find")

    def LoadLogState(self, logURL):
        if "logstate" + logURL in self.store:
            return CertificateLog.from_json(self.store["logstate" + logURL])
        return CertificateLog(ShortURL=logURL)

    def ListExpirationDates(self, notBefore_unix):
        day = (int(notBefore_unix) // 86400) * 86400 * 1000
        out = []
        for key in self.expDateToIssuer:
            ed = ExpDate.Parse(key)
            t = _dt.datetime.strptime(key[:10], kExpirationFormat)
            if _utc_ms(t.year, t.month, t.day) >= day:
                out.append(ed)
        return out

    def ListIssuersForExpirationDate(self, expDate):
        return self.expDateToIssuer.get(expDate.ID(), [])

    def ListSerialsForExpirationDateAndIssuer(self, expDate, issuer):
        return self.expDateIssuerIDToSerials.get(expDate.ID() + issuer.ID(), [])


class LocalDiskBackend:
    """storage/localdiskbackend.go — including its quirks: certificates are written WITHOUT the
    `.pem` suffix the listers look for (:194-199 vs :131,170), and MarkDirty writes `<id>/dirty`
    relative to the current directory, not rootPath (:89-91)."""
    kStateDirName, kDirtyMarker = "state", "dirty"

    def __init__(self, perms, path):
        self.perms, self.rootPath = perms, path

    def _store(self, path, data: bytes):
        d = os.path.dirname(path)
        if d and not os.path.isdir(d):
            os.makedirs(d, exist_ok=True)
        fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, self.perms)
        try:
            os.write(fd, data)
        finally:
            os.close(fd)

    def MarkDirty(self, id):
        self._store(os.path.join(id, self.kDirtyMarker), b"\x00")

    def AllocateExpDateAndIssuer(self, expDate, issuer):
        d = os.path.dirname(os.path.join(self.rootPath, expDate.ID(), issuer.ID()))
        os.makedirs(d, exist_ok=True)               # makeDirectoryIfNotExist splits off the last element

    def StoreCertificatePEM(self, serial, expDate, issuer, b):
        self.AllocateExpDateAndIssuer(expDate, issuer)
        self._store(os.path.join(self.rootPath, expDate.ID(), issuer.ID(), serial.ID()), b)

    def StoreLogState(self, log):
        self._store(os.path.join(self.rootPath, self.kStateDirName, log.ID()), log.to_json().encode())

    def StoreKnownCertificateList(self, issuer, serials):
        self._store(os.path.join(self.rootPath, issuer.ID()),
                    b"".join((s.HexString() + "\n").encode() for s in serials))

    def LoadCertificatePEM(self, serial, expDate, issuer):
        raise NotImplementedError("Unimplemented")  # :239-242

    def LoadLogState(self, logURL):
        path = os.path.join(self.rootPath, self.kStateDirName, CertificateLogIDFromShortURL(logURL))
        try:
            return CertificateLog.from_json(open(path).read())
        except OSError:
            return CertificateLog(ShortURL=logURL)


def pem_encode(der: bytes) -> bytes:
    """encoding/pem.EncodeToMemory of {Type: CERTIFICATE, no headers} (filesystemdatabase.go:167-175)."""
    b64 = base64.b64encode(der)
    lines = [b64[i:i + 64] for i in range(0, len(b64), 64)]
    return b"-----BEGIN CERTIFICATE-----\n" + b"\n".join(lines) + b"\n-----END CERTIFICATE-----\n"


# ------------------------------------------------------------------------- FilesystemDatabase
class IssuerDate:
    def __init__(self, issuer):
        self.Issuer, self.ExpDates = issuer, []


class FilesystemDatabase:
    """storage/filesystemdatabase.go with the per-entry part moved onto the GPU.

    Store(cert_der, issuer_der, …) keeps the reference's signature for one entry; StoreBatch is
    the batched insertCTWorker+Store: one Engine.map_batch, then — exactly as Store does after
    WasUnknown (:183-201) — IssuerMetadata.Accumulate, AllocateExpDateAndIssuer and
    StoreCertificatePEM for the newly unknown certificates, and markDirty for every stored entry."""

    def __init__(self, backend, cache: RemoteCache, engine: Engine = None):
        self.backend, self.extCache = backend, cache
        self.engine = engine if engine is not None else getattr(cache, "engine", None)
        self.meta = {}
        self._issuer_idx = {}            # sha256(chain[0] DER) → index in the GPU issuer table

    # -- reference API --------------------------------------------------------------------
    def GetIssuerMetadata(self, issuer: Issuer) -> IssuerMetadata:
        im = self.meta.get(issuer.ID())
        if im is None:
            im = self.meta[issuer.ID()] = IssuerMetadata(issuer, self.extCache)
        return im

    def GetKnownCertificates(self, expDate: ExpDate, issuer: Issuer) -> KnownCertificates:
        return KnownCertificates(expDate, issuer, self.extCache)

    def GetIssuerAndDatesFromCache(self):                        # :59-100
        issuerMap = {}
        for entry in self.extCache.KeysToChan("serials::*"):
            parts = entry.decode("latin1").split("::")
            if len(parts) != 3:
                raise ValueError("Unexpected key format: %s" % entry)
            try:
                expDate = ExpDate.Parse(parts[1])
            except ValueError:
                continue
            issuerMap.setdefault(parts[2], IssuerDate(Issuer.FromString(parts[2]))).ExpDates.append(expDate)
        return list(issuerMap.values())

    def ListExpirationDates(self, notBefore_unix): return self.backend.ListExpirationDates(notBefore_unix)
    def ListIssuersForExpirationDate(self, expDate): return self.backend.ListIssuersForExpirationDate(expDate)

    def SaveLogState(self, log):                                  # :110-118
        try:
            self.extCache.StoreLogState(log)
        except Exception:
            pass
        return self.backend.StoreLogState(log)

    def GetLogState(self, host, path):                            # :120-139
        shortUrl = "%s%s" % (host, path)
        try:
            return self.extCache.LoadLogState(shortUrl)
        except Exception:
            pass
        try:
            log = self.backend.LoadLogState(shortUrl)
            if log is not None:
                return log
        except Exception:
            pass
        return CertificateLog(ShortURL=shortUrl)

    def markDirty(self, not_after_day: str):                      # :141-144
        return self.backend.MarkDirty(not_after_day)

    # -- the batched path -----------------------------------------------------------------
    def _register_issuers(self, chain0_ders):
        idx, fresh = [], []
        for d in chain0_ders:
            if d is None:
                idx.append(N.NO_ISSUER)
                continue
            h = hashlib.sha256(d).digest()
            if h not in self._issuer_idx:
                self._issuer_idx[h] = self.engine.issuer_count() + len(fresh)
                fresh.append(d)
            idx.append(self._issuer_idx[h])
        if fresh:
            self.engine.add_issuers(fresh)
        return idx

    def StoreBatch(self, leaf_ders, chain0_ders, entry_types=None):
        """leaf_ders[i]: the X509 cert or the precert's Submitted.Data (ct-fetch.go:198-204);
        chain0_ders[i]: Chain[0].Data or None.  Returns the BatchResult (records, new_idx, stats)."""
        iss = self._register_issuers(chain0_ders)
        res = self.engine.map_batch(Batch.from_certs(list(leaf_ders), iss, entry_types))
        self._after_map(res, lambda i: leaf_ders[i])
        return res

    def StoreRawBatch(self, raw):
        """Raw get-entries input (engine.RawEntries): ct.LogEntryFromLeaf, the choice of certificate and Chain[0]
        and the issuer registration all happen on the GPU (Engine.map_entries); the host work is the same."""
        res = self.engine.map_entries(raw)
        cache = {}

        def cert_of(i):                      # only for the rare CTMR_MK_HOST / long-serial certificates
            if i not in cache:
                from . import _entry_host
                cache[i] = _entry_host.certificate_of(raw.leaf_input(i), raw.extra_data(i))
            return cache[i]
        self._after_map(res, cert_of)
        return res

    def _after_map(self, res, cert_of):
        """FilesystemDatabase.Store after WasUnknown (:183-205) for a whole batch."""
        rec = res.records
        pems = self.engine.pem_new()                              # pem.EncodeToMemory on the GPU (N1)
        issuers, exp_dates = {}, {}

        def issuer_of(idx):
            if idx not in issuers:
                issuers[idx] = Issuer.FromString(self.engine.issuer_info(idx).issuer_id.decode())
            return issuers[idx]

        def serial_of(i):
            n = int(rec["serial_len"][i])
            return Serial(bytes(rec["serial"][i][:n])) if n <= 20 else Serial(HostCert(cert_of(i)).serial)

        if getattr(self.engine, "collect_meta", False):
            # N3: the memo maps of IssuerMetadata live on the GPU; only first sightings come back
            for kind, i, idx, exp_hour, b in self.engine.meta_new():
                issuer, md = issuer_of(idx), self.GetIssuerMetadata(issuer_of(idx))
                if kind == N.MK_EXPDATE:                          # seenExpDateBefore == false (:189-195)
                    md.knownExpDates.add(ExpDate.FromHour(exp_hour).ID())
                    self.backend.AllocateExpDateAndIssuer(ExpDate.FromHour(exp_hour), issuer)
                elif kind == N.MK_CRL:
                    dp = b.decode("latin1")
                    if dp not in md.knownCrlDPs:
                        md.knownCrlDPs.add(dp)
                        md.addCRL(dp)
                elif kind == N.MK_DN:
                    dn = name_string(name_atvs(b))
                    if dn not in md.knownIssuerDNs:
                        md.knownIssuerDNs.add(dn)
                        md.addIssuerDN(dn)
                else:                                             # CTMR_MK_HOST: the reference's own per-cert path
                    if not md.Accumulate(HostCert(cert_of(i)), exp_hour):
                        self.backend.AllocateExpDateAndIssuer(ExpDate.FromHour(exp_hour), issuer)
            for k, i in enumerate(res.new_idx):
                i = int(i)
                self.backend.StoreCertificatePEM(serial_of(i), ExpDate.FromHour(int(rec["exp_hour"][i])),
                                                 issuer_of(int(rec["issuer_idx"][i])), pems[k])
        else:
            for k, i in enumerate(res.new_idx):                   # certWasUnknown branch :183-201, per certificate
                i = int(i)
                issuer = issuer_of(int(rec["issuer_idx"][i]))
                expDate = ExpDate.FromHour(int(rec["exp_hour"][i]))
                cert = HostCert(cert_of(i))
                seenBefore = self.GetIssuerMetadata(issuer).Accumulate(cert, int(rec["exp_hour"][i]))
                if not seenBefore:
                    self.backend.AllocateExpDateAndIssuer(expDate, issuer)
                self.backend.StoreCertificatePEM(Serial(cert.serial), expDate, issuer, pems[k])
        # :205 — every stored entry marks its NotAfter day dirty (distinct days only: the marker is idempotent)
        hours = np.unique(rec["exp_hour"][rec["status"] == N.ST_PASS] // 24)
        for day in hours:
            t = _time.gmtime(int(day) * 86400)
            self.markDirty("%04d-%02d-%02d" % (t.tm_year, t.tm_mon, t.tm_mday))

    def Store(self, cert_der, issuer_der, logURL=None, entryId=None):
        return self.StoreBatch([cert_der], [issuer_der])


def storage_statistics(db: FilesystemDatabase):
    """cmd/storage-statistics/storage-statistics.go:28-82 → {issuerID: (hours, serials, crls, dns)}, totals."""
    out, totalSerials, totalCRLs = {}, 0, 0
    for issuerObj in db.GetIssuerAndDatesFromCache():
        md = db.GetIssuerMetadata(issuerObj.Issuer)
        crls, dns = md.CRLs(), md.Issuers()
        count = sum(db.GetKnownCertificates(e, issuerObj.Issuer).Count() for e in issuerObj.ExpDates)
        totalSerials += count
        totalCRLs += len(crls)
        out[issuerObj.Issuer.ID()] = (len(issuerObj.ExpDates), count, sorted(crls), sorted(dns))
    return out, totalSerials, totalCRLs


