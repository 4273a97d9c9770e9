"""N2: ct.LogEntryFromLeaf restated — oracle (oracle/ctmr_oracle.c) against hand-built RFC 6962
vectors, and the product's decoder (ct_mapreduce_amd/csrc/entry_decode.h, host build) against the oracle on
synthetic and mutated entries.  No GPU.

PARITY UNPINNED for this row: the reference holds no raw get-entries fixture (cmd/ct-getcert and
cmd/ct-fetch/ct-fetch.go:452 are its only users of the decode and neither has a test), and CT-go v1.1.0 is not on
this machine.  The vectors below are built from the RFC 6962 §3.4 / §4.6 text with Python's struct module."""
import random
import struct

import numpy as np
import pytest

from ct_mapreduce_amd import synth
from oracle import oracle as orc
from tests.harness import product_decode_entry


def u24(n):
    return struct.pack(">I", n)[1:]


def asn1cert(b):
    return u24(len(b)) + b


def chain(certs):
    body = b"".join(asn1cert(c) for c in certs)
    return u24(len(body)) + body


def x509_leaf(cert, ts=1234567890123, ext=b"", version=0, leaf_type=0, entry_type=0):
    return struct.pack(">BBQH", version, leaf_type, ts, entry_type) + asn1cert(cert) + struct.pack(">H", len(ext)) + ext


def precert_leaf(tbs, ikh=b"\x11" * 32, ts=99, ext=b""):
    return struct.pack(">BBQH", 0, 0, ts, 1) + ikh + asn1cert(tbs) + struct.pack(">H", len(ext)) + ext


CERT = b"\x30\x03\x02\x01\x05" * 40
ISS = b"\x30\x82\x01\x00" + bytes(range(256))
ROOT = b"\x30\x05ROOT!"


def both(leaf, extra):
    """oracle and product must agree on everything the path consumes; returns the oracle's view."""
    o = orc.decode_entry(leaf, extra)
    for prefix in (b"", b"\x00" * 7):
        p = product_decode_entry(leaf, extra, prefix=prefix)
        assert bool(p.ok) == bool(o.ok)
        if not o.ok:
            continue
        base = len(prefix)
        assert p.entry_type == o.entry_type and p.timestamp == o.timestamp and p.n_chain == o.n_chain
        cert_base = base + (len(leaf) if o.cert_in_extra else 0)
        assert (p.cert_lo, p.cert_hi) == (cert_base + o.cert_off, cert_base + o.cert_off + o.cert_len)
        assert p.chain0_len == o.chain0_len
        if o.chain0_len:
            assert p.chain0_lo == base + len(leaf) + o.chain0_off
        if o.entry_type == 1:
            assert (p.tbs_lo, p.tbs_len) == (base + o.tbs_off, o.tbs_len)
    return o


def test_x509_entry():
    leaf, extra = x509_leaf(CERT), chain([ISS, ROOT])
    o = both(leaf, extra)
    assert o.ok and o.entry_type == 0 and o.timestamp == 1234567890123 and not o.cert_in_extra
    assert leaf[o.cert_off:o.cert_off + o.cert_len] == CERT
    assert extra[o.chain0_off:o.chain0_off + o.chain0_len] == ISS and o.n_chain == 2


def test_precert_entry_uses_submitted_precertificate():
    tbs = b"\x30\x04TBS!"
    leaf, extra = precert_leaf(tbs), asn1cert(CERT) + chain([ISS])
    o = both(leaf, extra)
    assert o.ok and o.entry_type == 1 and o.cert_in_extra and o.timestamp == 99
    assert extra[o.cert_off:o.cert_off + o.cert_len] == CERT          # Precert.Submitted.Data (ct-fetch.go:202)
    assert leaf[o.tbs_off:o.tbs_off + o.tbs_len] == tbs
    assert extra[o.chain0_off:o.chain0_off + o.chain0_len] == ISS and o.n_chain == 1


def test_empty_chain_is_valid_and_means_no_issuer():
    o = both(x509_leaf(CERT), chain([]))
    assert o.ok and o.n_chain == 0 and o.chain0_len == 0              # len(Chain) < 1 (ct-fetch.go:215)
    o = both(precert_leaf(b"\x30\x00"), asn1cert(CERT) + chain([]))
    assert o.ok and o.n_chain == 0


def test_extensions_and_version_byte():
    assert both(x509_leaf(CERT, ext=b"\x01\x02\x03"), chain([ISS])).ok
    assert both(x509_leaf(CERT, version=7), chain([ISS])).ok          # CT-go bounds the enum by maxval only


@pytest.mark.parametrize("leaf,extra", [
    (x509_leaf(CERT) + b"\x00", chain([ISS])),                          # MerkleTreeLeaf: trailing data
    (x509_leaf(CERT), chain([ISS]) + b"\x00"),                          # CertificateChain: trailing data
    (x509_leaf(b""), chain([ISS])),                                     # ASN.1Cert<1..>
    (x509_leaf(CERT), u24(3) + u24(0)),                                 # empty chain element
    (x509_leaf(CERT), u24(10) + asn1cert(ISS)),                         # chain length ≠ contents
    (x509_leaf(CERT), u24(len(asn1cert(ISS)) + 2) + asn1cert(ISS) + b"\x00\x00"),  # truncated element header
    (x509_leaf(CERT, entry_type=2), chain([ISS])),                      # unknown entry type
    (x509_leaf(CERT, entry_type=0x8000), chain([ISS])),                 # JSON entry: not handled by LogEntryFromLeaf
    (x509_leaf(CERT, leaf_type=1), chain([ISS])),                       # no timestamped_entry
    (x509_leaf(CERT)[:-1], chain([ISS])),                               # extensions length cut
    (struct.pack(">BBQH", 0, 0, 5, 0) + u24(500) + CERT + b"\x00\x00", chain([ISS])),  # cert longer than the leaf
    (precert_leaf(b""), asn1cert(CERT) + chain([ISS])),                 # TBSCertificate<1..>
    (precert_leaf(b"\x30\x00"), chain([ISS])),                          # extra_data lacks pre_certificate framing
    (precert_leaf(b"\x30\x00"), asn1cert(b"") + chain([ISS])),          # empty pre_certificate
    (precert_leaf(b"\x30\x00")[:30], asn1cert(CERT) + chain([ISS])),    # issuer_key_hash cut
    (b"", b""),
    (x509_leaf(CERT), b""),
])
def test_rejects(leaf, extra):
    assert not both(leaf, extra).ok


def test_every_truncation_is_rejected():
    for leaf, extra in ((x509_leaf(CERT, ext=b"xy"), chain([ISS, ROOT])),
                        (precert_leaf(b"\x30\x03abc"), asn1cert(CERT) + chain([ISS]))):
        for k in range(len(leaf)):
            assert not both(leaf[:k], extra).ok, k
        for k in range(len(extra)):
            assert not both(leaf, extra[:k]).ok, k


def test_synthetic_entries_wrap_the_synthetic_certificates():
    cfg = synth.config(seed=11, n_issuers=8, dup_permille=100)
    raw, b, iss = synth.host_entries(cfg, 5, 300), synth.host_batch(cfg, 5, 300), synth.issuers(cfg)
    types = set()
    for i in range(raw.n):
        leaf, extra = raw.leaf_input(i), raw.extra_data(i)
        o = both(leaf, extra)
        assert o.ok and o.entry_type == b.entry_type[i] and o.timestamp == synth.BASE_TIME * 1000 + 5 + i
        src = extra if o.cert_in_extra else leaf
        assert src[o.cert_off:o.cert_off + o.cert_len] == b.cert(i)
        assert extra[o.chain0_off:o.chain0_off + o.chain0_len] == iss[b.issuer_idx[i]]
        types.add(o.entry_type)
    assert types == {0, 1}


def mutate_entry(rng, leaf, extra):
    """Damage the TLS framing (not the certificate bodies): length fields, types, cuts, insertions."""
    leaf, extra = bytearray(leaf), bytearray(extra)
    k = rng.randrange(8)
    if k == 0:
        leaf[rng.randrange(min(len(leaf), 18))] ^= 1 << rng.randrange(8)
    elif k == 1:
        extra[rng.randrange(min(len(extra), 9))] ^= 1 << rng.randrange(8)
    elif k == 2:
        del leaf[rng.randrange(len(leaf)):]
    elif k == 3:
        del extra[rng.randrange(len(extra)):]
    elif k == 4:
        leaf += bytes(rng.randrange(1, 4))
    elif k == 5:
        extra += bytes(rng.randrange(1, 4))
    elif k == 6:
        leaf[-2:] = struct.pack(">H", rng.randrange(1, 9))    # extensions length without the bytes
    else:
        p = rng.randrange(len(extra))
        extra[p:p] = bytes([rng.randrange(256)])
    return bytes(leaf), bytes(extra)


def test_product_decoder_equals_oracle_on_mutations():
    rng = random.Random(6962)
    cfg = synth.config(seed=12, n_issuers=4)
    raw = synth.host_entries(cfg, 0, 200)
    n_ok = n_bad = 0
    for i in range(raw.n):
        for _ in range(10):
            leaf, extra = mutate_entry(rng, raw.leaf_input(i), raw.extra_data(i))
            o = both(leaf, extra)
            n_ok += bool(o.ok)
            n_bad += not o.ok
    assert n_ok > 50 and n_bad > 500


def test_raw_batch_equals_packed_batch_through_the_oracle():
    """Feeding raw entries or the certificates they wrap is the same job (statuses, WasUnknown, sets)."""
    cfg = synth.config(seed=13, n_issuers=16, dup_permille=150, ca_permille=20, expired_permille=20)
    raw, b, iss = synth.host_entries(cfg, 0, 2000), synth.host_batch(cfg, 0, 2000), synth.issuers(cfg)
    filt = b"Synth Issuer 00"
    o1, o2 = orc.Engine(filt, False, synth.BASE_TIME), orc.Engine(filt, False, synth.BASE_TIME)
    st1, unk1, eh1, ts = o1.raw_batch(raw.blob, raw.bounds)
    io = np.zeros(len(iss) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in iss])
    st2, unk2, eh2 = o2.batch(b.payload, b.offsets, b.issuer_idx, np.frombuffer(b"".join(iss), np.uint8), io)
    assert (st1 == st2).all() and (unk1 == unk2).all() and (eh1 == eh2).all()
    assert o1.keys() == o2.keys() and o1.total_count() == o2.total_count()
    assert (ts == synth.BASE_TIME * 1000 + np.arange(2000, dtype=np.uint64)).all()


def test_strict_leaf_drops_precert_entries_whose_leaf_tbs_does_not_parse():
    """ct.LogEntryFromLeaf parses a precertificate entry's leaf TBSCertificate too (ct-fetch.go:452); with strict_leaf (the
    default: the reference profile) the oracle fails the entry when that parse fails fatally; X509 entries and the fast
    profile are untouched."""
    from ct_mapreduce_amd import synth
    from ct_mapreduce_amd.engine import RawEntries
    from tests.test_walk_cpu import tbs_of
    cfg = synth.config(seed=12, n_issuers=2, dup_permille=0, ca_permille=0, expired_permille=0)
    iss = synth.issuer(cfg, 0)
    certs = [synth.leaf(cfg, i)[0] for i in range(6)]
    good_tbs = tbs_of(certs[0])
    bad_tbs = bytearray(good_tbs)
    bad_tbs[1] = 0x84                                           # the TBSCertificate's own length field lies
    pairs = [
        (precert_leaf(good_tbs, ts=1), asn1cert(certs[0]) + chain([iss])),            # fine
        (precert_leaf(bytes(bad_tbs), ts=2), asn1cert(certs[1]) + chain([iss])),      # leaf TBS broken, submitted precert fine
        (precert_leaf(good_tbs + b"\x00", ts=3), asn1cert(certs[2]) + chain([iss])),  # trailing data behind the TBS
        (precert_leaf(certs[3], ts=4), asn1cert(certs[3]) + chain([iss])),            # a whole certificate where a TBS belongs
        (x509_leaf(certs[4], ts=5), chain([iss])),                                    # X509 entries have no leaf TBS
        (precert_leaf(tbs_of(certs[5]), ts=6), asn1cert(certs[5]) + chain([iss])),
    ]
    raw = RawEntries.from_pairs(pairs)
    blob = np.concatenate([raw.blob, np.zeros(64, np.uint8)])
    lax = orc.Engine(b"", False, synth.BASE_TIME)
    lax.set_strict_leaf(False)                                  # CTMR_PROFILE_FAST's choice
    st0, unk0, _, _ = lax.raw_batch(blob, raw.bounds)
    assert (st0 == orc.ST_PASS).all() and unk0.all()
    strict = orc.Engine(b"", False, synth.BASE_TIME)            # the default since round 6: the reference profile
    st1, unk1, _, ts1 = strict.raw_batch(blob, raw.bounds)
    E = orc.ST_ENTRY_DECODE_ERROR
    assert list(st1) == [orc.ST_PASS, E, E, E, orc.ST_PASS, orc.ST_PASS]
    assert list(unk1) == [1, 0, 0, 0, 1, 1] and list(ts1) == [1, 0, 0, 0, 5, 6]
    assert strict.total_count() == 3
