"""-m gpu: BASELINE.json's configs[1] and configs[2] at their full sizes.

configs[1]  1 M synthetic ~1.5 KB certificates, single issuer, known-certificate dedup — bit-exact against the oracle on
            every entry (the batch is generated in HBM and copied back for the CPU run).
configs[2]  10 M certificates, 256 issuers, issuerCN prefix filter + per-issuer unique counts — too large for the
            oracle to run whole inside a test, so checked (i) through size-independent properties of the generator and
            of set insertion: new ⇔ PASS ∧ first carrier of its key, per-issuer counts = histogram of the new entries,
            idempotent replay, the NEW list; and (ii) bit-exact against the oracle on a strided, duplicate-closed
            sample of 1.1 M entries (bench.py's parity leg).
Both configurations run under BOTH profiles (round 6): the engine's default — CTMR_PROFILE_REFERENCE — and CTMR_PROFILE_FAST,
each against an oracle set to the same profile.  configs[2] runs a second time with a KNOWN DAMAGED FRACTION: 1.3 ‰ of the
entries hurt in place (subjectAltName elements, the bodies of five more extensions, Name strings) so that every family of
reference-profile rules fires at BASELINE size, with the oracle sample covering every damaged entry.
(configs[3]/[4] — 100 M-entry batch, 1 B-entry stream — run in bench.py / bench.py --stream with the same checks.)"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth, _native as N
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_is_dup, synth_src, oracle_sample_check, strided_sample, gather_sample  # noqa: E402  (the generator's duplicate structure; bench.py's oracle sample)

NOW = synth.BASE_TIME


def device_batch(eng, cfg, first, n, dev):
    d_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    total = eng.synth_device(cfg, first, n, d_off.data_ptr(), 0, 0, 0, 0)
    d_pay = torch.empty(total + N.PAYLOAD_PAD + 16, dtype=torch.uint8, device=dev)
    d_iss = torch.empty(n, dtype=torch.int32, device=dev)
    d_et = torch.empty(n, dtype=torch.uint8, device=dev)
    eng.synth_device(cfg, first, n, d_off.data_ptr(), d_pay.data_ptr(), d_pay.numel(), d_iss.data_ptr(), d_et.data_ptr())
    return d_off, d_pay, d_iss, d_et, total


@pytest.mark.parametrize("profile", ["reference", "fast"])
def test_config1_one_million_single_issuer_bit_exact(profile):
    n, dev = 1_000_000, torch.device("cuda:0")
    cfg = synth.config(seed=20260921 + 1, n_issuers=1, dup_permille=50)
    issuers = synth.issuers(cfg)
    eng = ctmr.Engine(device=0, table_slots=1 << 21, pair_slots=1 << 16)
    if profile == "fast":
        eng.set_profile("fast")                      # (the default IS the reference profile)
    eng.add_issuers(issuers)
    eng.set_filter(b"", False, NOW)
    d_off, d_pay, d_iss, d_et, total = device_batch(eng, cfg, 0, n, dev)
    d_rec = torch.empty(n * 32, dtype=torch.uint8, device=dev)
    d_new = torch.empty(n, dtype=torch.int64, device=dev)
    st = eng.map_batch_device(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), n,
                              d_rec.data_ptr(), d_new.data_ptr())
    # the reference loop on the same bytes
    o = orc.Engine(b"", False, NOW)
    o.set_profile(profile)
    io = np.array([0, len(issuers[0])], np.uint64)
    ost, ounk, oeh = o.batch(d_pay.cpu().numpy(), d_off.cpu().numpy().astype(np.uint64),
                             d_iss.cpu().numpy().astype(np.uint32), np.frombuffer(issuers[0], np.uint8), io)
    rec = d_rec.cpu().numpy().view(ctmr.engine.RECORD_DTYPE)
    assert (rec["status"] == ost).all()
    assert (((rec["flags"] & 2) != 0) == (ounk != 0)).all()
    assert (rec["exp_hour"][ost != orc.ST_PARSE_ERROR] == oeh[ost != orc.ST_PARSE_ERROR]).all()
    assert (d_new[:st.n_new].cpu().numpy() == np.nonzero(ounk)[0]).all()
    assert st.n_new == int(ounk.sum()) == o.total_count() == eng.total_count()
    assert int(eng.issuer_counts()[0]) == o.issuer_count(eng.issuer_id(0))
    assert sorted(eng.keys(b"serials::*")) == [k for k in o.keys() if k.startswith(b"serials::")]
    assert 0 < st.n_dup < st.n_new
    eng.close()


@pytest.mark.parametrize("profile", ["reference", "fast"])
def test_config2_ten_million_256_issuers_properties(profile):
    n, dev = 10_000_000, torch.device("cuda:0")
    cfg = synth.config(seed=20260921 + 2, n_issuers=256, zipf=1, dup_permille=100, ca_permille=10, expired_permille=10)
    eng = ctmr.Engine(device=0, table_slots=1 << 25, pair_slots=1 << 22)
    eng.set_profile(profile)
    eng.add_issuers(synth.issuers(cfg))
    eng.set_filter(b"Synth Issuer 0,Synth Issuer 1", False, NOW)        # passes issuers 000-199
    d_off, d_pay, d_iss, d_et, total = device_batch(eng, cfg, 0, n, dev)
    d_rec = torch.empty(n * 32, dtype=torch.uint8, device=dev)
    d_new = torch.empty(n, dtype=torch.int64, device=dev)
    st = eng.map_batch_device(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), n,
                              d_rec.data_ptr(), d_new.data_ptr())
    rec = d_rec.view(-1, 32)
    status = rec[:, 0].cpu().numpy()
    flags = rec[:, 1].cpu().numpy()
    iss = d_iss.cpu().numpy()
    # filter semantics: CN filter ⇔ issuer index ≥ 200 (unless an earlier filter fired), nothing fails to parse
    assert (status != N.ST_PARSE_ERROR).all() and st.by_status[N.ST_PARSE_ERROR] == 0
    cn = status == N.ST_FILTERED_CN
    assert (iss[cn] >= 200).all() and (iss[status == N.ST_PASS] < 200).all()
    # dedup: an entry is new ⇔ it reached Store and is the first carrier of its key (the generator re-emits an
    # EARLIER entry's key exactly for the entries synth_is_dup marks)
    dup = synth_is_dup(cfg.seed, 0, n, 100, np)
    new = (flags & N.FL_WAS_UNKNOWN) != 0
    assert (new == ((status == N.ST_PASS) & ~dup)).all()
    assert st.n_new == int(new.sum()) and st.n_dup == int(((status == N.ST_PASS) & dup).sum())
    new_idx = d_new[:st.n_new].cpu().numpy()
    assert (new_idx == np.nonzero(new)[0]).all()                         # ascending, complete: checksum-free equality
    # per-issuer unique counts (storage-statistics.go:44-53) = histogram of the new entries' issuers
    counts = eng.issuer_counts().astype(np.int64)
    assert (counts == np.bincount(iss[new], minlength=256)).all() and counts[200:].sum() == 0
    assert eng.total_count() == st.n_new
    # Σ SCARD over the keys of one issuer = its count (the statistics tool's own arithmetic)
    k0 = [k for k in eng.keys(b"serials::*::" + eng.issuer_id(7).encode())]
    assert sum(eng.set_cardinality(k) for k in k0) == int(counts[7])
    # … and against the ORACLE (round 3): 40 equally spaced slices of 25 000 entries over the whole batch + every entry
    # outside them whose key a sampled duplicate repeats, in log order — the closed set on which the oracle's answer is
    # the whole batch's answer (tests/test_bench_helpers_cpu.py); ≈ 1.1 M entries, status and WasUnknown bit-exact
    filt = b"Synth Issuer 0,Synth Issuer 1"
    _, pinfo, _, _ = oracle_sample_check(np, torch, ctmr, synth, N, cfg, 100, synth.issuers(cfg), filt, NOW, dev, 0, n,
                                         d_off, d_pay, d_iss, d_et, d_rec, 40, 25_000, profile=profile)
    assert pinfo["status_mismatches"] == 0 and pinfo["was_unknown_mismatches"] == 0
    assert pinfo["entries"] > 1_000_000 and pinfo["known_duplicates"] > 50_000 and pinfo["sources_outside_the_slices"] > 50_000
    # idempotence: replaying the batch finds every stored entry known
    st2 = eng.map_batch_device(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), n,
                               d_rec.data_ptr(), d_new.data_ptr())
    assert st2.n_new == 0 and st2.n_dup == int((status == N.ST_PASS).sum())
    assert (eng.issuer_counts().astype(np.int64) == counts).all()
    eng.close()


# ---- configs[2] with a known damaged fraction: every family of reference-profile rules fires at BASELINE size ------------
def _ext_list(der, c):
    """(oid content, value offset, value end) of every extension of a synthetic leaf (short and two-octet lengths only)."""
    def hdr(p):
        ln = der[p + 1]
        if ln < 0x80:
            return p + 2, p + 2 + ln
        k = ln & 0x7f
        return p + 2 + k, p + 2 + k + int.from_bytes(der[p + 2:p + 2 + k], "big")
    out, e = [], c.exts_off
    while e < c.exts_end:
        x, x_end = hdr(e)
        o, o_end = hdr(x)
        v = o_end
        if der[v] == 0x01:
            v = hdr(v)[1]
        out.append((bytes(der[o:o_end]), *hdr(v)))
        e = x_end
    return out


def _hurt(der, kind, c):
    """One in-place, length-preserving injury per rule family; returns the damaged bytes.  What each one must cost is the
    ORACLE's call — the comments say what is expected, the test asserts the counts."""
    b = bytearray(der)
    exts = {oid: (v, ve) for oid, v, ve in _ext_list(der, c)}
    san_v, san_e = exts[b"\x55\x1d\x11"]
    first = san_v + (2 if b[san_v + 1] < 0x80 else 2 + (b[san_v + 1] & 0x7f))     # the first GeneralName
    if kind == "san_uri_ctl":          # a URI with a control character: url.Parse fails — fatal
        b[first] = 0x86; b[first + 3] = 0x01
    elif kind == "san_uri_space_host":  # "//a b…": invalid character in host name — fatal
        b[first] = 0x86; b[first + 2] = 0x2f; b[first + 3] = 0x2f; b[first + 5] = 0x20
    elif kind == "san_ip_length":      # an iPAddress of 22..49 octets: CT-go's NON-fatal finding — precertificates only
        b[first] = 0x87
    elif kind == "san_truncated":      # the last dNSName claims one octet too many: "data truncated" — fatal
        p = first
        while p + 2 + b[p + 1] < san_e:
            p += 2 + b[p + 1]
        b[p + 1] += 1
    elif kind == "san_not_a_sequence":  # "bad SAN sequence" — fatal
        b[san_v] = 0x31
    elif kind == "san_uri_deep":       # a bad URI far into the subjectAltName (beyond the walk's second window) — fatal
        p, k = first, 0
        while p + 2 + b[p + 1] < san_e and k < 9:
            p += 2 + b[p + 1]; k += 1
        b[p] = 0x86; b[p + 4] = 0x7f
    elif kind == "key_usage_pad":      # parseBitString: pad count 8 — fatal
        v, _ = exts[b"\x55\x1d\x0f"]; b[v + 2] = 0x08
    elif kind == "eku_element":        # SEQUENCE OF OBJECT IDENTIFIER with a UTF8String in it — fatal
        v, _ = exts[b"\x55\x1d\x25"]; b[v + 2] = 0x0c
    elif kind == "ski_tag":            # not an OCTET STRING — fatal
        v, _ = exts[b"\x55\x1d\x0e"]; b[v] = 0x03
    elif kind == "aki_fit":            # the keyIdentifier does not fit — fatal
        v, _ = exts[b"\x55\x1d\x23"]; b[v + 3] = 0x7f
    elif kind == "crl_relative_name":  # fullName → nameRelativeToCRLIssuer holding a URI where a SET belongs — fatal
        v, _ = exts[b"\x55\x1d\x1f"]; b[v + 6] = 0xa1
    elif kind == "crl_reasons":        # distributionPoint → reasons [1] with pad count 0x30 — fatal
        v, _ = exts[b"\x55\x1d\x1f"]; b[v + 4] = 0x81
    elif kind == "name_constraints":   # subjectKeyIdentifier relabelled nameConstraints: its value is no SEQUENCE — fatal
        v, _ = exts[b"\x55\x1d\x0e"]; b[v - 3] = 0x1e
    elif kind == "issuer_utf8":        # the issuer's O (UTF8String) is not UTF-8: a string finding — precertificates only
        at = der.index(b"Synth CA Org", c.issuer_off); b[at + 3] = 0xff
    elif kind == "subject_utf8":       # … and the subject's CN
        at = der.index(b"host-", c.issuer_off + c.issuer_len); b[at + 2] = 0xc0
    else:
        raise KeyError(kind)
    return bytes(b)


KINDS_FATAL = ["san_uri_ctl", "san_uri_space_host", "san_truncated", "san_not_a_sequence", "san_uri_deep", "key_usage_pad",
               "eku_element", "ski_tag", "aki_fit", "crl_relative_name", "crl_reasons", "name_constraints"]
KINDS_FINDING = ["san_ip_length", "issuer_utf8", "subject_utf8"]


def test_config2_reference_rules_fire_at_scale():
    """10 M entries, 13 000 of them damaged in place (15 kinds, ≈ 870 each), chosen among the entries whose key no other
    entry shares (so that a dropped certificate changes nobody else's WasUnknown).  The reference profile (the default) must
    agree with the oracle on EVERY damaged entry and on a duplicate-closed 1 M sample around them, and every rule family must
    have fired; the fast profile, on the same bytes, sees only what lies on its path (the truncated / relabelled structures
    it walks anyway) — also bit-exact against an oracle in that profile."""
    n, dev = 10_000_000, torch.device("cuda:0")
    cfg = synth.config(seed=20260921 + 2, n_issuers=256, zipf=1, dup_permille=100, ca_permille=10, expired_permille=10)
    issuers = synth.issuers(cfg)
    filt = b"Synth Issuer 0,Synth Issuer 1"
    eng = ctmr.Engine(device=0, table_slots=1 << 25, pair_slots=1 << 22)
    eng.add_issuers(issuers)
    eng.set_filter(filt, False, NOW)
    d_off, d_pay, d_iss, d_et, total = device_batch(eng, cfg, 0, n, dev)
    # entries that are neither a duplicate nor the source of one
    src, isdup = synth_src(cfg.seed, np.arange(n, dtype=np.uint64), 100, np)
    single = ~isdup
    single[np.unique(src[isdup]).astype(np.int64)] = False
    rng = np.random.default_rng(20261001)
    kinds = KINDS_FATAL + KINDS_FINDING
    pick = np.sort(rng.choice(np.nonzero(single)[0], size=13_000, replace=False))
    offs = d_off.cpu().numpy().astype(np.int64)
    iss_h, et_h = d_iss.cpu().numpy().astype(np.uint32), d_et.cpu().numpy().astype(np.uint8)
    hurt, kind_of = {}, {}
    for j, i in enumerate(pick):
        lo, hi = int(offs[i]), int(offs[i + 1])
        der = d_pay[lo:hi].cpu().numpy().tobytes()
        k = kinds[j % len(kinds)]
        bad = _hurt(der, k, orc.parse_cert(der))
        assert len(bad) == len(der) and bad != der
        d_pay[lo:hi] = torch.from_numpy(np.frombuffer(bad, np.uint8).copy()).to(dev)
        hurt[int(i)], kind_of[int(i)] = bad, k
    torch.cuda.synchronize()
    # the sample: 40 strided slices, the sources of their duplicates, and EVERY damaged entry outside the slices
    ranges = strided_sample(n, 40, 25_000)
    in_sample = np.concatenate([np.arange(lo, hi, dtype=np.uint64) for lo, hi in ranges])
    closure = np.setdiff1d(src[in_sample.astype(np.int64)][isdup[in_sample.astype(np.int64)]], in_sample)
    outside = np.setdiff1d(np.array(sorted(hurt), np.uint64), in_sample)
    extra = np.concatenate([closure, outside])                  # (disjoint: a damaged entry is nobody's source)
    extra_certs = [synth.leaf(cfg, int(i)) for i in closure] + [(hurt[int(i)], int(iss_h[int(i)]), int(et_h[int(i)])) for i in outside]
    pay, off, iss, et, idx = gather_sample(d_off, d_pay, d_iss, d_et, ranges, extra, extra_certs, N.PAYLOAD_PAD, np)
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    blob = np.frombuffer(b"".join(issuers), np.uint8)
    gidx = torch.from_numpy(idx.astype(np.int64)).to(dev)
    d_rec = torch.empty(n * 32, dtype=torch.uint8, device=dev)
    d_new = torch.empty(n, dtype=torch.int64, device=dev)
    seen = {}
    for profile in ("reference", "fast"):
        eng.set_profile(profile)
        eng.reset_known()
        st = eng.map_batch_device(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), n,
                                  d_rec.data_ptr(), d_new.data_ptr())
        o = orc.Engine(filt, False, NOW)
        o.set_profile(profile)
        ost, ounk, _ = o.batch(pay, off, iss, blob, io, entry_type=et)
        rec = d_rec.view(-1, 32)[gidx].cpu().numpy().reshape(-1).view(ctmr.engine.RECORD_DTYPE)
        assert (rec["status"] == ost).all(), (profile, idx[np.nonzero(rec["status"] != ost)[0][:10]])
        assert (((rec["flags"] & 2) != 0) == (ounk != 0)).all(), profile
        # per rule family: how many of its damaged entries the profile refused (every damaged entry is in the sample)
        pos = {int(i): k for k, i in enumerate(idx)}
        refused = {k: 0 for k in kinds}
        count = {k: 0 for k in kinds}
        for i, k in kind_of.items():
            count[k] += 1
            refused[k] += int(ost[pos[i]] == orc.ST_PARSE_ERROR)
        seen[profile] = (refused, count, int(st.by_status[N.ST_PARSE_ERROR]))
        # the whole batch's parse errors are the damaged entries' (the generator emits none): the GPU's count over all
        # 10 M entries equals the oracle's over the damaged ones
        assert int(st.by_status[N.ST_PARSE_ERROR]) == sum(refused.values()), (profile, st.by_status, refused)
    refused, count, _ = seen["reference"]
    for k in KINDS_FATAL:
        assert refused[k] == count[k] > 800, (k, refused[k], count[k])                 # fatal in every role
    for k in KINDS_FINDING:                                                             # findings: precertificates only (≈ half)
        assert 0.35 * count[k] < refused[k] < 0.65 * count[k], (k, refused[k], count[k])
    fr, _, _ = seen["fast"]
    # the fast profile skips these bodies by length: it refuses only what breaks the structure it walks — none of these kinds
    # but the truncated SAN… which it never reads either (the value fits its OCTET STRING)
    assert sum(fr.values()) == 0, fr
    eng.close()


def test_aligned_entry_view_is_the_same_certificates_at_aligned_offsets():
    """ctmr_synth_view_device (bench.py --aligned, secondary.aligned128): the synthetic certificates laid at multiples of
    128 bytes as an entry view — byte-identical certificates, identical records / NEW list / counts through
    ctmr_map_view_device."""
    n, dev = 200_000, torch.device("cuda:0")
    cfg = synth.config(seed=20260921 + 4, n_issuers=64, zipf=1, dup_permille=50, ca_permille=10, expired_permille=10)
    results = []
    for align in (0, 128):
        eng = ctmr.Engine(device=0, table_slots=1 << 19, pair_slots=1 << 16)
        eng.add_issuers(synth.issuers(cfg))
        eng.set_filter(b"Synth Issuer 0", False, NOW)
        d_rec = torch.empty(n * 32, dtype=torch.uint8, device=dev)
        d_new = torch.empty(n, dtype=torch.int64, device=dev)
        if align:
            d_st = torch.empty(n + 1, dtype=torch.int64, device=dev)
            d_en = torch.empty(n, dtype=torch.int64, device=dev)
            total = eng.synth_view_device(cfg, 0, n, align, d_st.data_ptr(), d_en.data_ptr(), 0, 0, 0, 0)
            d_pay = torch.zeros(total + N.PAYLOAD_PAD + 16, dtype=torch.uint8, device=dev)
            d_iss = torch.empty(n, dtype=torch.int32, device=dev)
            d_et = torch.empty(n, dtype=torch.uint8, device=dev)
            eng.synth_view_device(cfg, 0, n, align, d_st.data_ptr(), d_en.data_ptr(), d_pay.data_ptr(), d_pay.numel(),
                                  d_iss.data_ptr(), d_et.data_ptr())
            assert int((d_st[:n] % align).abs().sum().item()) == 0 and int(d_st[n].item()) == total
            view = N.EntryView(cert_start=d_st.data_ptr(), cert_end=d_en.data_ptr(), issuer_idx=d_iss.data_ptr(),
                               entry_type=d_et.data_ptr(), timestamp=None, chain0_start=None, chain0_len=None)
            st = eng.map_view_device(d_pay.data_ptr(), total, view, n, d_rec.data_ptr(), d_new.data_ptr())
            lens = (d_en - d_st[:n]).cpu().numpy()
            first = d_pay[int(d_st[5].item()):int(d_en[5].item())].cpu().numpy().tobytes()
        else:
            d_off, d_pay, d_iss, d_et, total = device_batch(eng, cfg, 0, n, dev)
            st = eng.map_batch_device(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), n,
                                      d_rec.data_ptr(), d_new.data_ptr())
            lens = np.diff(d_off.cpu().numpy())
            first = d_pay[int(d_off[5].item()):int(d_off[6].item())].cpu().numpy().tobytes()
        results.append((d_rec.cpu().numpy().copy(), d_new[:st.n_new].cpu().numpy().copy(), int(st.n_new),
                        [int(x) for x in st.by_status], eng.issuer_counts().copy(), lens, first, d_iss.cpu().numpy().copy()))
        eng.close()
    a, b = results
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and a[2] == b[2] > 0 and a[3] == b[3]
    assert (a[4] == b[4]).all() and (a[5] == b[5]).all() and a[6] == b[6] == synth.leaf(cfg, 5)[0] and (a[7] == b[7]).all()
