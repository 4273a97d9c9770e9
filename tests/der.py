"""Tiny DER encoder for hand-built edge-case certificates (test helper)."""


def tlv(tag, content=b""):
    n = len(content)
    if n < 0x80:
        l = bytes([n])
    else:
        b = n.to_bytes((n.bit_length() + 7) // 8, "big")
        l = bytes([0x80 | len(b)]) + b
    return bytes([tag]) + l + content


def seq(*items):
    return tlv(0x30, b"".join(items))


def oid(*bs):
    return tlv(0x06, bytes(bs))


def rdn(attr, value, tag=0x0c):
    return tlv(0x31, seq(oid(0x55, 0x04, attr), tlv(tag, value)))


def name(*rdns):
    return seq(*rdns)


def utctime(s):
    return tlv(0x17, s.encode())


def gentime(s):
    return tlv(0x18, s.encode())


SIGALG = bytes.fromhex("300d06092a864886f70d01010b0500")
# id-ecPublicKey / prime256v1 with the curve's base point G as the key: a point ON the curve (CT-go's parsePublicKey —
# elliptic.Unmarshal — and OpenSSL both reject a certificate whose point is not)
P256_G = bytes.fromhex("6b17d1f2e12c4247f8bce6e563a440f277037d812deb33a0f4a13945d898c296"
                       "4fe342e2fe1a7f9b8ee7eb4a7c0f9e162bce33576b315ececbb6406837bf51f5")
EC_SPKI = bytes.fromhex("3059301306072a8648ce3d020106082a8648ce3d030107034200") + b"\x04" + P256_G
# a second key on the curve (2·G), for tests that need two distinct SubjectPublicKeyInfos
P256_2G = bytes.fromhex("7cf27b188d034f7e8a52380304b51ac3c08969e277f21b35a60b48fc47669978"
                        "07775510db8ed040293d9ac69f7430dbba7dade63ce982299e04b79d227873d1")
EC_SPKI_2 = EC_SPKI[:27] + P256_2G


def spki(alg_oid, params, key_bits, pad=0):
    """SubjectPublicKeyInfo from the algorithm OID's content octets, the parameters TLV (b"" = absent) and the BIT
    STRING's octets behind the pad count."""
    return seq(seq(tlv(0x06, alg_oid), params), tlv(0x03, bytes([pad]) + key_bits))


OID_RSA = bytes.fromhex("2a864886f70d010101")
OID_DSA = bytes.fromhex("2a8648ce380401")
OID_EC = bytes.fromhex("2a8648ce3d0201")
NULL = b"\x05\x00"


def rsa_spki(n=b"\x00" + b"\xc3" * 256, e=b"\x01\x00\x01", params=NULL, inner_extra=b"", outer_extra=b"", pad=0, alg=OID_RSA):
    return spki(alg, params, seq(tlv(0x02, n), tlv(0x02, e), inner_extra) + outer_extra, pad)


def ext(oid_last, value, critical=None):
    items = [oid(0x55, 0x1d, oid_last)]
    if critical is not None:
        items.append(tlv(0x01, b"\xff" if critical else b"\x00"))
    items.append(tlv(0x04, value))
    return seq(*items)


def cert(serial=b"\x01", issuer=None, not_before=None, not_after=None, subject=None, spki=EC_SPKI,
         exts=None, version=True, sig=b"\x00" + b"\x5a" * 64, extra_tbs=b"", tbs_sigalg=SIGALG, outer_sigalg=SIGALG):
    """version: True = [0]{INTEGER 2}, False = absent, bytes = those bytes verbatim."""
    issuer = issuer if issuer is not None else name(rdn(3, b"Test CA"))
    subject = subject if subject is not None else name(rdn(3, b"leaf"))
    not_before = not_before or utctime("250101000000Z")
    not_after = not_after or utctime("270101000000Z")
    tbs = b""
    if isinstance(version, bytes):
        tbs += version
    elif version:
        tbs += tlv(0xa0, tlv(0x02, b"\x02"))
    tbs += tlv(0x02, serial) + tbs_sigalg + issuer + seq(not_before, not_after) + subject + spki
    if exts is not None:
        tbs += tlv(0xa3, seq(*exts))
    tbs += extra_tbs
    return seq(tlv(0x30, tbs), outer_sigalg, tlv(0x03, sig))


BC_CA = ext(0x13, seq(tlv(0x01, b"\xff")), critical=True)
BC_NOT_CA = ext(0x13, seq(), critical=True)
