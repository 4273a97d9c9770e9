"""-m gpu: ctmr_set_strict_spki (ON by default) — the public key inside subjectPublicKeyInfo as CT-go's parsePublicKey
judges it, through the C ABI on the GPU against the oracle: hand-built keys per rule, key-targeted mutations, the three
roles (X509 entry, precertificate, Chain[0] issuer), the switch off, raw entries with strict_leaf, and the mixed
synthetic corpus (whose EC keys are real curve points since round 4)."""
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
from tests.test_spki_cpu import CURVES, cert, ec_spki, key_seeds, point, spki_mutate  # noqa: E402

NOW = synth.BASE_TIME


def rule_certs():
    """One certificate per rule of tests/test_spki_cpu.py (accepted, findings, fatal), distinct serials."""
    spkis = [D.rsa_spki(), D.rsa_spki(params=b""), D.rsa_spki(params=D.tlv(0x30, b"")), D.rsa_spki(outer_extra=b"\x00"),
             D.rsa_spki(inner_extra=D.NULL), D.rsa_spki(n=b"\xc3" * 256), D.rsa_spki(n=b"\x00"), D.rsa_spki(n=b"\x00\x00\x00"),
             D.rsa_spki(n=b"\x00\x00\x00\x01"), D.rsa_spki(n=b"\x00" * 300 + b"\x01"), D.rsa_spki(n=b""),
             D.rsa_spki(e=b"\x00"), D.rsa_spki(e=b"\xff"), D.rsa_spki(e=b"\x00\x01"), D.rsa_spki(e=b"\x00" * 8), D.rsa_spki(e=b"\x01" * 9),
             D.rsa_spki(e=b"\x7f" + b"\xff" * 7), D.rsa_spki(n=b"\x00" + b"\xa7" * 512), D.rsa_spki(n=b"\x00" + b"\xa7" * 512, e=b"\x00"),
             D.rsa_spki(alg=bytes.fromhex("2a864886f70d010107"), params=D.seq()), D.spki(bytes.fromhex("2b6570"), b"", b"\x01" * 31),
             D.spki(D.OID_RSA, D.NULL, b""), D.spki(D.OID_RSA, D.NULL, D.seq(D.tlv(0x02, b"\x05"))),
             D.spki(D.OID_RSA, D.NULL, D.seq(D.tlv(0x02, b"\x05"), D.tlv(0x02, b"\x03")))]
    for curve in CURVES:
        pt = point(curve, 11)
        bad = bytearray(pt)
        bad[-1] ^= 1
        spkis += [ec_spki(curve, pt), ec_spki(curve, bytes(bad)), ec_spki(curve, pt[:-1]), ec_spki(curve, pt, prefix=b"\x03")]
    good = bytes.fromhex(CURVES["P256"][0])
    spkis += [ec_spki(params=b""), ec_spki(params=D.NULL), ec_spki(params=D.tlv(0x06, bytes.fromhex("2b8104000a"))),
              ec_spki(params=D.tlv(0x06, good) + D.NULL), ec_spki("P256", bytes(64))]
    v = int.from_bytes(b"\x04" + point("P256"), "big") << 1                      # pad bits: the key is shifted right
    spkis.append(D.spki(D.OID_EC, D.tlv(0x06, good), v.to_bytes(66, "big")[1:], pad=1))
    P, Q, G, Y = b"\x00\xe3" + b"\x11" * 126, b"\x00\xc9" + b"\x22" * 19, b"\x5a" * 128, b"\x3c" * 128
    par = D.seq(D.tlv(0x02, P), D.tlv(0x02, Q), D.tlv(0x02, G))
    spkis += [D.spki(D.OID_DSA, par, D.tlv(0x02, Y)), D.spki(D.OID_DSA, par, D.tlv(0x02, b"\x00" + Y)),
              D.spki(D.OID_DSA, b"", D.tlv(0x02, Y)), D.spki(D.OID_DSA, par, D.tlv(0x02, b"\x00")),
              D.spki(D.OID_DSA, D.seq(D.tlv(0x02, P), D.tlv(0x02, b"\xff"), D.tlv(0x02, G)), D.tlv(0x02, Y))]
    long_subject = D.name(*[D.rdn(10, b"organisation %02d of a very long subject" % i) for i in range(8)])
    out = []
    for k, sp in enumerate(spkis):
        out.append(D.cert(serial=b"\x21" + k.to_bytes(2, "big"), spki=sp, exts=[D.BC_NOT_CA]))
        out.append(D.cert(serial=b"\x22" + k.to_bytes(2, "big"), spki=sp, subject=long_subject, exts=[D.BC_NOT_CA]))
    return out


def run_both(certs, iss, ets, issuers, strict_spki=True, table=1 << 14):
    batch = Batch.from_certs(certs, iss, ets)
    batch.payload = np.concatenate([batch.payload, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    eng = ctmr.Engine(device=0, table_slots=table, pair_slots=1 << 10)
    if not strict_spki:
        eng.set_strict_spki(False)                      # before the issuers: they are judged when registered
    eng.add_issuers(issuers)
    eng.set_filter(b"", True, NOW)                      # (set_filter must not reset the switch)
    res = eng.map_batch(batch)
    o = orc.Engine(b"", True, NOW)
    o.set_strict_spki(strict_spki)
    o, st, unk, eh = run_oracle(batch, issuers, b"", True, NOW, engine=o)
    assert_records_equal(res, batch, st, unk, eh, strict_spki=strict_spki)
    assert_state_equal(eng, o, len(issuers))
    eng.close()
    return st


def test_every_rule_in_all_three_roles_and_with_the_switch_off():
    good_issuer = D.cert(serial=b"\x01", exts=[D.BC_CA])
    bad_key_issuer = D.cert(serial=b"\x02", spki=ec_spki(pt=bytes(range(64))), exts=[D.BC_CA])          # fatal
    finding_issuer = D.cert(serial=b"\x03", spki=D.rsa_spki(params=b""), exts=[D.BC_CA])                  # non-fatal
    issuers = [good_issuer, bad_key_issuer, finding_issuer]
    rc = rule_certs()
    certs, iss, ets = [], [], []
    for k, c in enumerate(rc):
        for et in (0, 1):
            certs.append(c); iss.append(0); ets.append(et)
        certs.append(c); iss.append(1 + k % 2); ets.append(0)
    order = list(range(len(certs)))
    random.Random(5).shuffle(order)
    certs, iss, ets = [certs[i] for i in order], [iss[i] for i in order], [ets[i] for i in order]
    on = run_both(certs, iss, ets, issuers, True)
    off = run_both(certs, iss, ets, issuers, False)
    hist = lambda st: [int((st == s).sum()) for s in range(8)]
    assert hist(off)[orc.ST_PARSE_ERROR] == 0 and hist(off)[orc.ST_ISSUER_PARSE_ERROR] == 0
    assert hist(on)[orc.ST_PARSE_ERROR] > 60 and hist(on)[orc.ST_ISSUER_PARSE_ERROR] > 20 and hist(on)[orc.ST_PASS] > 30
    # a finding costs the X509 entry nothing and the precertificate its place
    x509_pass = {(certs[i], ets[i]) for i in range(len(certs)) if on[i] == orc.ST_PASS and iss[i] == 0}
    assert any(et == 0 and (c, 1) not in x509_pass for c, et in x509_pass)


def test_key_targeted_mutations_against_the_oracle():
    rng = random.Random(20261002)
    cfg = synth.config(seed=78, n_issuers=4, profile=1)
    issuers = synth.issuers(cfg)
    seeds = key_seeds() + [synth.leaf(cfg, i)[0] for i in range(24)]
    spans = []
    for s in seeds:
        c = orc.parse_cert(s)
        spans.append((c.spki_off, c.spki_off + c.spki_len))
    certs, iss, ets = [], [], []
    for r in range(40000):
        i = r % len(seeds)
        der = spki_mutate(rng, seeds[i], *spans[i])
        if rng.randrange(4) == 0:
            der = spki_mutate(rng, der, *spans[i])
        certs.append(der); iss.append(r % 4); ets.append(r & 1)
    st = run_both(certs, iss, ets, issuers, True, table=1 << 17)
    n_err = int((st == orc.ST_PARSE_ERROR).sum())
    assert 5000 < n_err < 35000
    # the same mutants as Chain[0] issuers: each registered certificate's verdict is the oracle's
    muts = certs[:600]
    eng = ctmr.Engine(device=0, table_slots=1 << 12, pair_slots=1 << 10)
    eng.add_issuers(muts)
    leaf = cert(D.rsa_spki())
    b = Batch.from_certs([leaf] * len(muts), list(range(len(muts))), [0] * len(muts))
    b.payload = np.concatenate([b.payload, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    eng.set_filter(b"", True, NOW)
    res = eng.map_batch(b)
    for k, m in enumerate(muts):
        c = orc.parse_cert(m)
        want = orc.ST_PASS if (c.ok and c.nonfatal == 0) else orc.ST_ISSUER_PARSE_ERROR
        assert res.records["status"][k] == want, k
    eng.close()


def test_raw_entries_strict_leaf_parses_the_leaf_tbs_key_too():
    """ct.LogEntryFromLeaf → x509.ParseTBSCertificate ends in the same parsePublicKey: with strict_leaf a precertificate
    entry whose LEAF TBSCertificate carries a key that does not parse is dropped by the downloader (fatal errors only)."""
    issuer = D.cert(serial=b"\x01", exts=[D.BC_CA])
    pairs = []
    for k, c in enumerate(rule_certs()[:120]):
        pairs.append((x509_leaf(c, ts=2 * k), chain([issuer])))
        pairs.append((precert_leaf(tbs_of(c), ts=2 * k + 1), asn1cert(c) + chain([issuer])))
    random.Random(6).shuffle(pairs)
    raw = RawEntries.from_pairs(pairs)
    raw.blob = np.concatenate([raw.blob, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    seen = {}
    for strict_leaf in (False, True):
        for spki in (True, False):
            eng = ctmr.Engine(device=0, table_slots=1 << 12, pair_slots=1 << 10)
            eng.set_filter(b"", True, NOW)
            eng.set_strict_leaf(strict_leaf)
            eng.set_strict_spki(spki)
            res = eng.map_entries(raw)
            o = orc.Engine(b"", True, NOW)
            o.set_strict_leaf(strict_leaf)
            o.set_strict_spki(spki)
            st, _ = check_against_oracle(eng, raw, o, res)
            seen[strict_leaf, spki] = [int((st == s).sum()) for s in range(8)]
            eng.close()
    assert seen[True, True][orc.ST_ENTRY_DECODE_ERROR] > 10 and seen[False, True][orc.ST_ENTRY_DECODE_ERROR] == 0
    assert seen[True, False][orc.ST_ENTRY_DECODE_ERROR] == 0 and seen[True, False][orc.ST_PARSE_ERROR] == 0


def test_mixed_corpus_is_accepted_whole_and_the_switch_changes_nothing_on_it():
    """The generator's mixed profile (half the leaf keys EC P-256) lies inside what the reference accepts: no parse errors,
    and the same records with the key parse on and off."""
    cfg = synth.config(seed=20260922, n_issuers=32, profile=1, dup_permille=100, ca_permille=20, expired_permille=20)
    batch = synth.host_batch(cfg, 0, 20000)
    issuers = synth.issuers(cfg)
    recs = []
    for spki in (True, False):
        eng = ctmr.Engine(device=0, table_slots=1 << 16, pair_slots=1 << 12)
        eng.set_strict_spki(spki)
        eng.add_issuers(issuers)
        eng.set_filter(b"", False, NOW)
        res = eng.map_batch(batch)
        assert res.stats.by_status[orc.ST_PARSE_ERROR] == 0 and res.stats.by_status[orc.ST_ISSUER_PARSE_ERROR] == 0
        recs.append(res.records.copy())
        if spki:
            o, st, unk, eh = run_oracle(batch, issuers, b"", False, NOW)
            assert_records_equal(res, batch, st, unk, eh)
        eng.close()
    assert (recs[0] == recs[1]).all()


@pytest.mark.parametrize("mode", ["owner", "bloom"])
@pytest.mark.parametrize("world", [2, 3, 4])
def test_groups_withdraw_the_keys_of_certificates_whose_point_is_off_the_curve(world, mode):
    """Global dedup with EC keys: an entry whose key belongs to another rank leaves the map as a key record BEFORE its curve
    point has been checked (k_ec_resolve runs behind the map); when the point is bad the record is withdrawn — a staged
    32-byte record, or, for a 21..40-octet serial, the entry's place in the 64-byte record list (whose COUNT must follow:
    scripts/fuzz_gpu_groups.py found a stale count sending an uninitialised record that cost entry 0 its WasUnknown)."""
    from ct_mapreduce_amd.distributed import Group
    from tests.test_gpu_exchange import to_dev, dev_shard
    rng = random.Random(100 * world + len(mode))
    iname = D.name(D.rdn(3, b"Synth Issuer 000"))
    issuers = [D.cert(serial=b"\x01", subject=iname, issuer=iname, exts=[D.BC_CA]),
               D.cert(serial=b"\x02", subject=iname, issuer=iname, spki=D.EC_SPKI_2, exts=[D.BC_CA])]
    pool = []
    for k in range(160):
        ln = rng.choice((8, 16, 20, 21, 30, 40))
        serial = bytes([1 + rng.randrange(0x7e)] + [rng.randrange(256) for _ in range(ln - 1)])
        kind = k % 4
        pt = point("P256", 3 + k)
        if kind == 1:
            pt = pt[:-1] + bytes([pt[-1] ^ 1])                         # off the curve: parse error in every role
        sp = D.rsa_spki() if kind == 3 else ec_spki("P384" if kind == 2 else "P256", point("P384", 3 + k) if kind == 2 else pt)
        pool.append((D.cert(serial=serial, issuer=iname, spki=sp, exts=[D.BC_NOT_CA]), k % 2))
    items = [rng.choice(pool) for _ in range(1500)]
    items[0] = pool[3]                                                 # entry 0 of rank 0: an RSA key that must stay NEW
    bounds = [0] + sorted(rng.randrange(1, len(items)) for _ in range(world - 1)) + [len(items)]
    engines = []
    for _ in range(world):
        e = ctmr.Engine(device=0, table_slots=1 << 13, pair_slots=1 << 10)
        e.add_issuers(issuers)
        e.set_filter(b"", True, NOW)
        engines.append(e)
    g = Group.local(engines)
    if mode == "bloom":
        g.bloom_config(1 << 14)
    o = orc.Engine(b"", True, NOW)
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    blob = np.frombuffer(b"".join(issuers), np.uint8)
    keep, shards, want = [], [], []
    for r in range(world):
        lo, hi = bounds[r], bounds[r + 1]
        b = Batch.from_certs([c for c, _ in items[lo:hi]], [k for _, k in items[lo:hi]], [rng.randrange(2) for _ in range(hi - lo)])
        pay = np.concatenate([b.payload, np.zeros(N.PAYLOAD_PAD, np.uint8)])
        want.append(o.batch(pay, b.offsets, b.issuer_idx, blob, io, entry_type=b.entry_type))
        keep.append(to_dev(b))
        shards.append(dev_shard(keep[-1], b.n, order_base=lo))
    stats = g.map_batch(mode, shards)
    n_err = 0
    for r in range(world):
        st, unk, _ = want[r]
        rec = keep[r][4].cpu().numpy().view(ctmr.engine.RECORD_DTYPE)[:len(st)]
        assert (rec["status"] == st).all(), r
        assert (((rec["flags"] & 2) != 0) == (unk != 0)).all(), (r, np.nonzero(((rec["flags"] & 2) != 0) != (unk != 0))[0][:5])
        assert stats[r].n_new == int(unk.sum())
        n_err += int((st == orc.ST_PARSE_ERROR).sum())
    assert n_err > 200
    assert g.total_count() == o.total_count()
    g.close()
    for e in engines:
        e.close()


def test_an_entry_the_downloader_dropped_keeps_its_status_whatever_its_key():
    """Advisor, round 4: entry_type CTMR_ENTRY_INVALID through ctmr_map_batch is ENTRY_DECODE_ERROR — with a well-formed
    certificate whose EC point is off its curve k_ec_resolve used to rewrite it to PARSE_ERROR, while a bad RSA key (judged
    inside the map) left it alone: the status histogram depended on the key type."""
    issuer = D.cert(serial=b"\x01", exts=[D.BC_CA])
    off_curve = D.cert(serial=b"\x21", spki=ec_spki(pt=bytes(range(64))))
    bad_rsa = D.cert(serial=b"\x22", spki=D.rsa_spki(e=b"\x00"))
    fine = D.cert(serial=b"\x23")
    certs = [off_curve, bad_rsa, fine, off_curve, fine]
    ets = [N.ENTRY_INVALID, N.ENTRY_INVALID, N.ENTRY_INVALID, 0, 0]
    batch = Batch.from_certs(certs, [0] * 5, ets)
    batch.payload = np.concatenate([batch.payload, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    eng = ctmr.Engine(device=0, table_slots=1 << 10, pair_slots=1 << 10)
    eng.add_issuers([issuer])
    eng.set_filter(b"", True, NOW)
    res = eng.map_batch(batch)
    assert list(res.records["status"]) == [N.ST_ENTRY_DECODE_ERROR] * 3 + [N.ST_PARSE_ERROR, N.ST_PASS]
    assert res.stats.by_status[N.ST_ENTRY_DECODE_ERROR] == 3 and res.stats.by_status[N.ST_PARSE_ERROR] == 1
    assert eng.total_count() == 1
    eng.close()
