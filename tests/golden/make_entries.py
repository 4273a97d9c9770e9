#!/usr/bin/env python3
"""Raw get-entries fixtures built from the reference's own certificates (round 3).

The reference holds no raw `get-entries` fixture, so the N2 decode oracle was pinned on hand-built RFC 6962
vectors only.  This script wraps the three PEM certificates of the reference's unit tests (kLeadingZeroes / kEmptySPKI /
kRealSPKI, tests/golden/*.pem) into RFC 6962 §3.4 MerkleTreeLeaf structures and §4.6 extra_data with an encoder that
shares NO code with the oracle, the product or the other tests (struct + a 20-line DER reader), and writes next to each
entry what the reference's loop must make of it — derived from the goldens the reference's tests pin (serial `00aa`,
types_test.go:81-101; expDate hour, knowncertificates_test.go:85-110; Issuer.ID = base64url(SHA-256(SPKI)),
types_test.go:41-57) and SURVEY.md §8(c)'s OpenSSL cross-check of the same certificates.

  python tests/golden/make_entries.py            # verify entries_from_reference_pems.json
  python tests/golden/make_entries.py --write    # rewrite it
"""
import base64
import hashlib
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "entries_from_reference_pems.json")


def pem(name):
    body = open(os.path.join(HERE, name + ".pem")).read().split("-----")[2]
    return base64.b64decode("".join(body.split()))


def tlv(buf, pos):
    """(tag, content_start, content_end) of the DER element at pos — definite lengths only."""
    tag = buf[pos]
    ln = buf[pos + 1]
    if ln < 0x80:
        return tag, pos + 2, pos + 2 + ln
    k = ln & 0x7f
    return tag, pos + 2 + k, pos + 2 + k + int.from_bytes(buf[pos + 2:pos + 2 + k], "big")


def tbs_and_spki(cert):
    _, c0, _ = tlv(cert, 0)                       # Certificate ::= SEQUENCE
    _, t0, t1 = tlv(cert, c0)                     # tbsCertificate
    tbs = cert[c0:t1]
    p = t0
    if cert[p] == 0xa0:                           # [0] version
        p = tlv(cert, p)[2]
    for _ in range(5):                            # serial, signature, issuer, validity, subject
        p = tlv(cert, p)[2]
    spki = cert[p:tlv(cert, p)[2]]
    return tbs, spki


def u24(n):
    return struct.pack(">I", n)[1:]


def asn1cert(b):
    return u24(len(b)) + b


def chain_of(certs):
    body = b"".join(asn1cert(c) for c in certs)
    return u24(len(body)) + body


def x509_entry(cert, chain, ts, ext=b""):
    leaf = struct.pack(">BBQH", 0, 0, ts, 0) + asn1cert(cert) + struct.pack(">H", len(ext)) + ext
    return leaf, chain_of(chain)


def precert_entry(precert, issuer, chain, ts):
    tbs, _ = tbs_and_spki(precert)               # (a log would strip the poison extension; LogEntryFromLeaf does not compare)
    ikh = hashlib.sha256(tbs_and_spki(issuer)[1]).digest()
    leaf = struct.pack(">BBQH", 0, 0, ts, 1) + ikh + asn1cert(tbs) + struct.pack(">H", 0)
    return leaf, asn1cert(precert) + chain_of(chain)


def build():
    lz, empty, real = pem("kLeadingZeroes"), pem("kEmptySPKI"), pem("kRealSPKI")
    issuer_id = base64.urlsafe_b64encode(hashlib.sha256(tbs_and_spki(empty)[1]).digest()).decode()
    assert issuer_id == "VCIlmPM9NkgFQtrs4Oa5TeFcDu6MWRTKSNdePEhOgD8="      # OpenSSL's view of the same SPKI
    assert base64.urlsafe_b64encode(hashlib.sha256(tbs_and_spki(real)[1]).digest()).decode() == "d_Kor69hknpIfroNumzs6NkLxxCUNhMn46dzck_SZSQ="
    key = "serials::2020-02-05-00::" + issuer_id                              # notAfter 2020-02-05 00:00:00Z, truncated to the hour
    entries = []

    def add(name, pair, **want):
        leaf, extra = pair
        entries.append(dict(name=name, leaf_input=base64.b64encode(leaf).decode(), extra_data=base64.b64encode(extra).decode(), **want))

    ts = 1577923200123                                                        # 2020-01-02T00:00:00.123Z
    add("x509 kLeadingZeroes, chain [kEmptySPKI, kRealSPKI]", x509_entry(lz, [empty, real], ts),
        entry_type=0, timestamp=ts, cert_sha256=hashlib.sha256(lz).hexdigest(), chain0_sha256=hashlib.sha256(empty).hexdigest(),
        status="PASS", was_unknown=True, serial_hex="00aa", exp_date="2020-02-05-00", issuer_id=issuer_id, key=key)
    add("x509 kRealSPKI (a CA certificate), chain [kEmptySPKI]", x509_entry(real, [empty], ts + 1),
        entry_type=0, timestamp=ts + 1, cert_sha256=hashlib.sha256(real).hexdigest(), chain0_sha256=hashlib.sha256(empty).hexdigest(),
        status="FILTERED_CA", was_unknown=False)
    add("precert: leaf TBS of kLeadingZeroes, submitted precertificate kLeadingZeroes, chain [kEmptySPKI]",
        precert_entry(lz, empty, [empty], ts + 2),
        entry_type=1, timestamp=ts + 2, cert_sha256=hashlib.sha256(lz).hexdigest(), chain0_sha256=hashlib.sha256(empty).hexdigest(),
        status="PASS", was_unknown=False, serial_hex="00aa", exp_date="2020-02-05-00", issuer_id=issuer_id, key=key)
    add("x509 kLeadingZeroes, empty chain", x509_entry(lz, [], ts + 3),
        entry_type=0, timestamp=ts + 3, cert_sha256=hashlib.sha256(lz).hexdigest(), chain0_sha256=None,
        status="NO_ISSUER", was_unknown=False)
    add("x509 kLeadingZeroes with CtExtensions, issued by kRealSPKI's key holder in name only", x509_entry(lz, [real], ts + 4, ext=b"\x01\x02\x03"),
        entry_type=0, timestamp=ts + 4, cert_sha256=hashlib.sha256(lz).hexdigest(), chain0_sha256=hashlib.sha256(real).hexdigest(),
        status="PASS", was_unknown=True, serial_hex="00aa", exp_date="2020-02-05-00",
        issuer_id="d_Kor69hknpIfroNumzs6NkLxxCUNhMn46dzck_SZSQ=", key="serials::2020-02-05-00::d_Kor69hknpIfroNumzs6NkLxxCUNhMn46dzck_SZSQ=")
    return {"now": 1577836800, "filter": "", "log_expired": False,
            "note": "now = 2020-01-01T00:00:00Z (before the certificates' notAfter); entries in log order, one stream",
            "entries": entries,
            "final_keys": sorted({e["key"] for e in entries if e.get("key")}), "final_total_count": 2}


def main():
    want = build()
    if "--write" in sys.argv:
        json.dump(want, open(OUT, "w"), indent=1)
        print("written", OUT)
        return 0
    ok = os.path.exists(OUT) and json.load(open(OUT)) == want
    print("entries_from_reference_pems.json", "ok" if ok else "DIFFERS")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
