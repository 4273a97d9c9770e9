"""strict_extensions on the GPU: packed batches (leaf, precertificate, Chain[0] issuer) and raw entries with strict_leaf
against the oracle, switch on and off."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import ct_mapreduce_amd as ctmr  # noqa: E402
from ct_mapreduce_amd import synth, _native as N  # noqa: E402
from ct_mapreduce_amd.engine import RECORD_DTYPE  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests import der as D  # noqa: E402
from tests.gpu_common import run_oracle  # noqa: E402
from tests.test_ext_cpu import (x, aia, san, uri, dps, fullname, nc, sct_ext, EKU_SRV, EKU_CLI, POL, OCSP,  # noqa: E402
                                sia, ipaddr, asnum, bits, CA_REPO, NULL)

NOW = synth.BASE_TIME


def corpus(rng, n):
    issuer_name = D.name(D.rdn(3, b"Ext Issuer"))
    good = [x(15, D.tlv(0x03, b"\x05\xa0")), x(14, D.tlv(0x04, b"\x11" * 20)), x(37, D.seq(EKU_SRV, EKU_CLI)),
            x(35, D.seq(D.tlv(0x80, b"\x22" * 20))), x(32, D.seq(D.seq(POL))),
            aia(D.seq(D.seq(OCSP, D.tlv(0x86, b"http://o.example")))), D.BC_NOT_CA,
            # round 5: subjectAltName (long enough to leave the walk's LDS window; a URI: the exact reader's business),
            # cRLDistributionPoints with a nameRelativeToCRLIssuer, an embedded SCT list
            san(*[D.tlv(0x82, b"host-%02d.a-rather-long-name.example" % k) for k in range(12)], D.tlv(0x87, bytes(4)),
                uri("https://u@a.example:8443/x%20y?q#f")),
            dps(D.seq(fullname(D.tlv(0x86, b"http://crl.example/a.crl"))),
                D.seq(D.tlv(0xa0, D.tlv(0xa1, D.tlv(0x31, D.seq(D.oid(0x55, 4, 3), D.tlv(0x13, b"rel"))))), D.tlv(0x81, b"\x01\x06"))),
            sct_ext(D.tlv(0x04, b"\x00\x05\x00\x03abc")),
            # round 6: what only CT-go's fork parses — subjectInfoAccess, RFC 3779 address blocks and AS identifiers
            sia(D.seq(D.seq(CA_REPO, D.tlv(0x86, b"rsync://repo.example/ca/")))),
            ipaddr(D.seq(D.seq(D.tlv(0x04, b"\x00\x01"), D.seq(bits(b"\x0a"), D.seq(bits(b"\xc0\xa8"), bits(b"\xc0\xa9\x80", 7)))))),
            asnum(D.seq(D.tlv(0xa0, D.seq(D.tlv(0x02, b"\x00\xfd\xe8"))), D.tlv(0xa1, NULL)))]
    bad = [aia(D.seq()), sia(D.seq()), sia(D.seq(D.seq(CA_REPO))), sia(D.seq(D.seq(CA_REPO, NULL)) + b"\x00"),          # fatal (CT-go: recalled)
           ipaddr(D.seq(D.seq(D.tlv(0x04, b"\x01"), NULL))), ipaddr(D.seq(D.seq(D.tlv(0x04, b"\x00\x01"), D.seq(bits(b"\x0b", 1))))),   # non-fatal
           asnum(D.seq(D.tlv(0xa0, D.seq(D.tlv(0x02, b"\x00\x01"))))), asnum(D.seq(D.tlv(0xa0, NULL)) + b"\x00"), asnum(b""),          # non-fatal
           x(15, D.tlv(0x03, b"\x08\x00")), x(14, D.tlv(0x03, b"\x00\x11")), x(37, D.seq(EKU_SRV, D.tlv(0x0c, b"x"))),
           x(35, D.seq(b"\x80\x7f\x01")), x(32, D.seq(POL)), aia(D.seq(D.seq(OCSP))), x(15, D.tlv(0x03, b"\x05\xa0") + b"\x00"),
           san(D.tlv(0x82, b"a"), uri("http://a b/")), san(uri(":x")), san(D.tlv(0x82, b"a")) [:-1] + b"\x00",   # fatal
           san(D.tlv(0x87, bytes(5))), sct_ext(D.tlv(0x04, b"\x00\x06\x00\x03abc")),                           # non-fatal: precertificates only
           dps(D.seq(D.tlv(0x81, b"\x08\x00"))), dps(D.seq(fullname(b"\x86\x05ab"))),
           dps(D.seq(D.tlv(0xa0, D.tlv(0xa1, D.tlv(0x31, D.seq(D.oid(0x55, 4, 3))))))),
           nc([D.tlv(0x82, b"exa mple")])]
    certs = []
    for i in range(n):
        exts = list(good)
        rng.shuffle(exts)
        if rng.random() < 0.4:
            exts[rng.randrange(len(exts))] = rng.choice(bad)
        c = D.cert(serial=bytes([1 + i % 120, i // 120 % 256, rng.randrange(256)]), issuer=issuer_name, exts=exts,
                   not_after=D.utctime("270101000000Z"))
        if rng.random() < 0.3:                                        # random damage inside the extensions block
            o = orc.parse_cert(c)
            b = bytearray(c)
            p = rng.randrange(o.exts_off, o.exts_end)
            b[p] = rng.choice((b[p] ^ (1 << rng.randrange(8)), 0x00, 0x30, 0x06, 0x80))
            c = bytes(b)
        certs.append(c)
    return certs


@pytest.mark.parametrize("strict", [False, True])
def test_packed_batches_follow_the_oracle(strict):
    rng = random.Random(5)
    issuer_ok = D.cert(subject=D.name(D.rdn(3, b"Ext Issuer")), exts=[D.BC_CA, x(15, D.tlv(0x03, b"\x01\x06"))])
    issuer_bad = D.cert(subject=D.name(D.rdn(3, b"Ext Issuer")), exts=[D.BC_CA, x(15, D.tlv(0x03, b"\x08\x00"))])
    issuers = [issuer_ok, issuer_bad,
               D.cert(subject=D.name(D.rdn(3, b"Ext Issuer")), spki=D.EC_SPKI_2,
                      exts=[D.BC_CA, nc([D.tlv(0x82, b".example.com"), D.tlv(0x81, b"u@example.com")], [D.tlv(0x87, bytes(4) + b"\xff\xff\x00\x00")])]),
               D.cert(subject=D.name(D.rdn(3, b"Ext Issuer")), exts=[D.BC_CA, nc([D.tlv(0x86, b"1.2.3.4")])]),        # a URI constraint that is an IP
               D.cert(subject=D.name(D.rdn(3, b"Ext Issuer")), exts=[D.BC_CA, san(D.tlv(0x87, bytes(3)))]),         # a non-fatal finding drops an issuer
               D.cert(subject=D.name(D.rdn(3, b"Ext Issuer")), spki=D.EC_SPKI_2,
                      exts=[D.BC_CA, asnum(D.seq(D.tlv(0xa0, D.seq(D.tlv(0x04, b"\x01")))))]),                       # … an RFC 3779 finding too (round 6)
               D.cert(subject=D.name(D.rdn(3, b"Ext Issuer")), spki=D.EC_SPKI_2,
                      exts=[D.BC_CA, asnum(D.seq(D.tlv(0xa0, D.seq(D.tlv(0x02, b"\x01"))))), sia(D.seq(D.seq(CA_REPO, NULL)))])]   # a well-formed RPKI CA
    certs = corpus(rng, 3000)
    b = ctmr.Batch.from_certs(certs, [rng.randrange(len(issuers)) for _ in certs], [rng.randrange(2) for _ in certs])
    o = orc.Engine(b"", True, NOW)
    o.set_strict_extensions(strict)
    o, st, unk, eh = run_oracle(b, issuers, b"", True, NOW, engine=o)
    eng = ctmr.Engine(device=0, table_slots=1 << 14, pair_slots=1 << 10)
    eng.set_strict_extensions(strict)
    eng.add_issuers(issuers)
    eng.set_filter(b"", True, NOW)
    res = eng.map_batch(b)
    assert (res.records["status"] == st).all(), np.nonzero(res.records["status"] != st)[0][:10]
    assert (((res.records["flags"] & 2) != 0) == (unk != 0)).all()
    assert eng.total_count() == o.total_count()
    if strict:
        assert (st == orc.ST_PARSE_ERROR).sum() > 1000 and (st == orc.ST_ISSUER_PARSE_ERROR).sum() > 300   # … and the refused issuer's entries
    eng.close()


def test_the_reference_profile_is_the_four_switches():
    """ctmr_set_profile(CTMR_PROFILE_REFERENCE) ≡ strict_spki + strict_leaf + strict_strings + strict_extensions ≡ what
    ctmr_create gives (round 6), and CTMR_PROFILE_FAST ≡ the three opt-outs — on a corpus where each switch has something to say."""
    rng = random.Random(7)
    issuer = D.cert(subject=D.name(D.rdn(3, b"Ext Issuer")), exts=[D.BC_CA])
    certs = corpus(rng, 1500)
    for i in range(0, len(certs), 7):                                 # Names with a character-set finding
        certs[i] = D.cert(serial=bytes([7, i % 251, i // 251]), issuer=D.name(D.rdn(10, b"a@b", 0x13), D.rdn(3, b"Ext Issuer")))
    b = ctmr.Batch.from_certs(certs, [0] * len(certs), [rng.randrange(2) for _ in certs])
    got = {}
    for mode in ("reference", "switches", "fast", "fast_switches", "defaults"):
        eng = ctmr.Engine(device=0, table_slots=1 << 13, pair_slots=1 << 10)
        if mode == "switches":
            eng.set_profile("fast")
            eng.set_strict_leaf(True); eng.set_strict_strings(True); eng.set_strict_extensions(True); eng.set_strict_spki(True)
        elif mode == "reference":
            eng.set_profile("fast"); eng.set_profile("reference")
        elif mode == "fast":
            eng.set_profile("fast")
        elif mode == "fast_switches":
            eng.set_strict_leaf(False); eng.set_strict_strings(False); eng.set_strict_extensions(False)
        eng.add_issuers([issuer])
        eng.set_filter(b"", True, NOW)
        got[mode] = eng.map_batch(b).records["status"].copy()
        eng.close()
    assert (got["reference"] == got["switches"]).all() and (got["reference"] == got["defaults"]).all()
    assert (got["fast"] == got["fast_switches"]).all()
    o = orc.Engine(b"", True, NOW)                                   # the oracle's defaults are the reference profile too
    st = run_oracle(b, [issuer], b"", True, NOW, engine=o)[1]
    assert (got["reference"] == st).all()
    of = orc.Engine(b"", True, NOW)
    of.set_profile("fast")
    assert (got["fast"] == run_oracle(b, [issuer], b"", True, NOW, engine=of)[1]).all()
    assert (got["reference"] != got["fast"]).sum() > 300


def test_raw_entries_with_strict_leaf_and_the_switch():
    rng = random.Random(6)
    issuer = D.cert(subject=D.name(D.rdn(3, b"Ext Issuer")), exts=[D.BC_CA])
    certs = corpus(rng, 400)
    pairs = []
    for c in certs:
        o = orc.parse_cert(c)
        if o.ok and rng.random() < 0.5:
            pairs.append(synth_precert_entry(c, o, issuer))
        else:
            pairs.append(synth_x509_entry(c, issuer))
    raw = ctmr.engine.RawEntries.from_pairs(pairs)
    raw.blob = np.concatenate([raw.blob, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    for strict in (False, True):
        oe = orc.Engine(b"", True, NOW)
        oe.set_strict_leaf(True)
        oe.set_strict_extensions(strict)
        ost = oe.raw_batch(raw.blob, raw.bounds)[0]
        eng = ctmr.Engine(device=0, table_slots=1 << 12, pair_slots=1 << 10)
        eng.set_strict_leaf(True)
        eng.set_strict_extensions(strict)
        eng.set_filter(b"", True, NOW)
        res = eng.map_entries(raw)
        assert (res.records["status"] == ost).all(), (strict, np.nonzero(res.records["status"] != ost)[0][:10])
        eng.close()


def tls_vec(data, n):
    return len(data).to_bytes(n, "big") + data


def synth_x509_entry(cert, issuer):
    leaf = b"\x00\x00" + (1234).to_bytes(8, "big") + b"\x00\x00" + tls_vec(cert, 3) + b"\x00\x00"
    return leaf, tls_vec(tls_vec(issuer, 3), 3)


def synth_precert_entry(cert, o, issuer):
    tbs = cert[o.tbs_off:o.tbs_off + o.tbs_len]
    leaf = b"\x00\x00" + (1234).to_bytes(8, "big") + b"\x00\x01" + b"\x42" * 32 + tls_vec(tbs, 3) + b"\x00\x00"
    return leaf, tls_vec(cert, 3) + tls_vec(tls_vec(issuer, 3), 3)
