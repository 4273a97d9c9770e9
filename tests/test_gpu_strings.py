"""-m gpu: ctmr_set_strict_strings — the Go-stdlib character-set rules for the string values of both Names as an opt-in,
non-fatal finding (k_name_strings pre-pass + the map's non-fatal rule; k_issuer_ids for Chain[0] issuers), against the
oracle in the same mode; the default mode keeps its answers on the same input."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402,F401

import ct_mapreduce_amd as ctmr  # noqa: E402
from ct_mapreduce_amd import synth, _native as N  # noqa: E402
from ct_mapreduce_amd.engine import Batch, RawEntries  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests import der as D  # noqa: E402
from tests.gpu_common import run_oracle, assert_records_equal, assert_state_equal  # noqa: E402
from tests.test_entry_decode_cpu import x509_leaf, precert_leaf, chain, asn1cert  # noqa: E402
from tests.test_gpu_entries import check_against_oracle  # noqa: E402
from tests.test_walk_cpu import tbs_of  # noqa: E402

NOW = synth.BASE_TIME

VALUES = [(0x13, b"Plain Org (EU) *&"), (0x13, b"under_score"), (0x13, b"caf\xe9"), (0x12, b"0123 456"), (0x12, b"12a"),
          (0x16, b"mail@example"), (0x16, b"m\xe4il"), (0x0c, "Zürich 東京".encode("utf-8")), (0x0c, b"ab\xc3"),
          (0x0c, b"\xed\xa0\x80"), (0x0c, b"\xf4\x90\x80\x80"), (0x0c, b"\xc0\x80"), (0x14, b"t61 \xe4\xff"), (0x1e, b"\xd8\x00"),
          (0x0c, b"x" * 300 + b"\xff"), (0x13, b"y" * 200)]


def test_packed_batch_x509_kept_precert_and_issuer_dropped():
    rng = random.Random(3)
    iname = D.name(D.rdn(3, b"Synth Issuer 000"))
    good_issuer = D.cert(serial=b"\x01", subject=iname, issuer=iname, exts=[D.BC_CA])
    bad_issuer = D.cert(serial=b"\x02", subject=D.name(D.rdn(3, b"Synth Issuer 001"), D.rdn(10, b"Org_with_underscore", tag=0x13)),
                        issuer=iname, exts=[D.BC_CA])
    issuers = [good_issuer, bad_issuer]
    certs, iss, ets = [], [], []
    k = 0
    for tag, val in VALUES:
        for where in ("issuer", "subject"):
            for et in (0, 1):
                for which in (0, 1):
                    k += 1
                    nm = D.name(D.rdn(10, val, tag), D.rdn(3, b"Synth Issuer 000"))
                    kw = {"issuer": iname, where: nm}
                    certs.append(D.cert(serial=b"\x11" + k.to_bytes(2, "big"), **kw))
                    iss.append(which)
                    ets.append(et)
    order = list(range(len(certs)))
    rng.shuffle(order)
    certs = [certs[i] for i in order] + [certs[order[0]], certs[order[1]]]          # and two duplicates
    iss = [iss[i] for i in order] + [iss[order[0]], iss[order[1]]]
    ets = [ets[i] for i in order] + [ets[order[0]], ets[order[1]]]
    batch = Batch.from_certs(certs, iss, ets)
    batch.payload = np.concatenate([batch.payload, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    seen = {}
    for strict in (False, True):
        eng = ctmr.Engine(device=0, table_slots=1 << 12, pair_slots=1 << 10)
        eng.set_strict_strings(strict)                         # before the issuers: they are judged when registered
        eng.add_issuers(issuers)
        eng.set_filter(b"Synth", True, NOW)
        res = eng.map_batch(batch)
        o = orc.Engine(b"Synth", True, NOW)
        o.set_strict_strings(strict)
        o, st, unk, eh = run_oracle(batch, issuers, b"Synth", True, NOW, engine=o)
        assert_records_equal(res, batch, st, unk, eh, strict)
        assert_state_equal(eng, o, len(issuers))
        seen[strict] = [int((st == s).sum()) for s in range(8)]
        eng.close()
    assert seen[False][orc.ST_PARSE_ERROR] == 0 and seen[False][orc.ST_ISSUER_PARSE_ERROR] == 0
    assert seen[True][orc.ST_PARSE_ERROR] > 30 and seen[True][orc.ST_ISSUER_PARSE_ERROR] > 30


def test_raw_entries_with_string_findings_in_leaf_precertificate_and_chain0():
    """The raw get-entries path: X509 entries keep a certificate with a finding, precertificate entries lose it, and a
    Chain[0] with a finding registers as an issuer that does not parse (its entries: ISSUER_PARSE_ERROR)."""
    iname = D.name(D.rdn(3, b"Synth Issuer 000"))
    good_issuer = D.cert(serial=b"\x01", subject=iname, issuer=iname, exts=[D.BC_CA])
    bad_issuer = D.cert(serial=b"\x02", subject=D.name(D.rdn(3, b"Synth Issuer 000"), D.rdn(11, b"unit\xff", tag=0x0c)),
                        issuer=iname, exts=[D.BC_CA])
    pairs = []
    k = 0
    for tag, val in VALUES:
        for ch in (good_issuer, bad_issuer):
            k += 1
            c = D.cert(serial=b"\x11" + k.to_bytes(2, "big"), issuer=iname, subject=D.name(D.rdn(3, val, tag)))
            pairs.append((x509_leaf(c, ts=k), chain([ch])))
            k += 1
            c = D.cert(serial=b"\x11" + k.to_bytes(2, "big"), issuer=iname, subject=D.name(D.rdn(3, val, tag)))
            pairs.append((precert_leaf(tbs_of(c), ts=k), asn1cert(c) + chain([ch])))
    random.Random(4).shuffle(pairs)
    raw = RawEntries.from_pairs(pairs)
    raw.blob = np.concatenate([raw.blob, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    seen = {}
    for strict in (False, True):
        eng = ctmr.Engine(device=0, table_slots=1 << 12, pair_slots=1 << 10)
        eng.set_filter(b"", True, NOW)
        eng.set_strict_strings(strict)
        res = eng.map_entries(raw)
        o = orc.Engine(b"", True, NOW)
        o.set_strict_strings(strict)
        st, _ = check_against_oracle(eng, raw, o, res)
        seen[strict] = [int((st == s).sum()) for s in range(8)]
        eng.close()
    assert seen[False][orc.ST_PASS] == len(pairs)
    assert seen[True][orc.ST_PARSE_ERROR] > 8 and seen[True][orc.ST_ISSUER_PARSE_ERROR] > 8 and seen[True][orc.ST_PASS] > 8


def test_mutated_synthetic_names_against_the_oracle():
    """A synthetic batch whose Name bytes are damaged at random (every string type of the generator's Names), both entry
    types, strict mode against the oracle — and the switch costs the default mode nothing: same records as without it."""
    rng = random.Random(8)
    cfg = synth.config(seed=31, n_issuers=8, dup_permille=50)
    issuers = synth.issuers(cfg)
    certs, iss, ets = [], [], []
    for i in range(6000):
        der, k = synth.leaf(cfg, i)[0], synth.leaf(cfg, i)[1]
        c = orc.parse_cert(der)
        b = bytearray(der)
        if rng.random() < 0.7:                                   # one byte inside the issuer or subject Name
            lo = c.issuer_off if rng.random() < 0.5 else c.issuer_off + c.issuer_len + 32
            at = lo + rng.randrange(4, 60)
            b[at] = rng.choice((0x80, 0xff, 0x5f, 0xc3, 0xe0, 0x40, 0x00, b[at] ^ 0x20))
        certs.append(bytes(b)); iss.append(int(k)); ets.append(i & 1)
    batch = Batch.from_certs(certs, iss, ets)
    batch.payload = np.concatenate([batch.payload, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    recs = {}
    for strict in (False, True):
        eng = ctmr.Engine(device=0, table_slots=1 << 15, pair_slots=1 << 12)
        eng.set_strict_strings(strict)
        eng.add_issuers(issuers)
        eng.set_filter(b"", True, NOW)
        res = eng.map_batch(batch)
        o = orc.Engine(b"", True, NOW)
        o.set_strict_strings(strict)
        o, st, unk, eh = run_oracle(batch, issuers, b"", True, NOW, engine=o)
        assert_records_equal(res, batch, st, unk, eh, strict)
        recs[strict] = (res.records.copy(), st.copy())
        eng.close()
    dropped = (recs[True][1] == orc.ST_PARSE_ERROR) & (recs[False][1] == orc.ST_PASS)
    assert 300 < int(dropped.sum()) < 3000                       # precertificates with a finding, and only those
    assert (np.asarray(ets)[dropped] == 1).all()
