"""Real-world DER diversity without a network: the CA bundles shipped in the image (certifi's cacert.pem and
/etc/ssl/certs/ca-certificates.crt — RSA-2048/4096 and EC P-256/P-384 keys, UTCTime and GeneralizedTime, Printable /
UTF8 / T61 strings, names without a CommonName, 20-byte and 1-byte serials, v1 certificates).  Oracle, the product's
walk (host build) and OpenSSL 3 must agree on every field the path consumes.  No GPU."""
import base64
import hashlib
import os
import re

import pytest

from oracle import oracle as orc
from tests import harness


def bundle_ders():
    paths = ["/etc/ssl/certs/ca-certificates.crt"]
    try:
        import certifi
        paths.append(certifi.where())
    except ImportError:
        pass
    seen, out = set(), []
    for p in paths:
        if not os.path.exists(p):
            continue
        for b in re.findall(r"-----BEGIN CERTIFICATE-----(.*?)-----END CERTIFICATE-----", open(p).read(), re.S):
            d = base64.b64decode("".join(b.split()))
            if d not in seen:
                seen.add(d)
                out.append(d)
    return out


def test_real_roots_oracle_product_openssl_agree():
    ders = bundle_ders()
    if len(ders) < 50:
        pytest.skip("no CA bundle in this image")
    n_cn = n_gt = n_ec = 0
    for d in ders:
        c = orc.parse_cert(d)
        p = harness.product_walk(d, 0xA5)
        o = harness.ossl_extract(d)
        assert c.ok and p.ok and o is not None                      # Go parses every Mozilla root; so must the profile
        # product walk ≡ oracle
        assert (p.serial_off, p.serial_len, p.not_before, p.not_after, p.cn_off, p.cn_len, p.spki_off, p.spki_len,
                bool(p.bc_valid), bool(p.is_ca)) == \
               (c.serial_off, c.serial_len, c.not_before, c.not_after, c.cn_off, c.cn_len, c.spki_off, c.spki_len,
                bool(c.bc_valid), bool(c.is_ca))
        # oracle ≡ OpenSSL on what OpenSSL exposes
        assert (c.not_before, c.not_after) == (o.not_before, o.not_after)
        assert d[c.cn_off:c.cn_off + c.cn_len] == bytes(o.cn[:o.cn_len])
        assert bool(c.bc_valid and c.is_ca) == bool(o.is_ca) and bool(c.bc_valid) == bool(o.has_bc)
        assert d[c.spki_off:c.spki_off + c.spki_len] == bytes(o.spki[:o.spki_len])
        mag = d[c.serial_off:c.serial_off + c.serial_len].lstrip(b"\x00") or b"\x00"   # OpenSSL normalises, the path does not
        assert mag == (bytes(o.serial[:o.serial_len]).lstrip(b"\x00") or b"\x00")
        assert orc.issuer_id(d[c.spki_off:c.spki_off + c.spki_len]) == \
            base64.urlsafe_b64encode(hashlib.sha256(bytes(o.spki[:o.spki_len])).digest()).decode()
        n_cn += c.cn_len > 0
        n_gt += d[c.tbs_off:c.tbs_off + c.tbs_len].find(b"\x18\x0f") >= 0
        n_ec += d.find(bytes.fromhex("2a8648ce3d0201")) >= 0
    assert n_cn > 50 and n_ec > 10                                  # the bundle really is diverse
