"""One-off fuzz campaign on the GPU box: mutated certificates through the default map kernel (strict LDS window +
miss fallback) against the oracle, every observable compared.  Not a test (minutes, not seconds):
    gpurun -- 'python scripts/fuzz_gpu.py 400000'
"""
import glob
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

import ct_mapreduce_amd as ctmr  # noqa: E402
from ct_mapreduce_amd import synth, _native as N  # noqa: E402
from ct_mapreduce_amd.engine import Batch  # noqa: E402
from tests import der as D  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.gpu_common import run_oracle, expected_records  # noqa: E402
from tests.test_walk_cpu import mutate, edge_seeds  # noqa: E402
from tests.test_spki_cpu import key_seeds, spki_mutate  # noqa: E402
from tests.test_gpu_meta import expected_first_sightings, got_first_sightings  # noqa: E402
from tests.test_ext_cpu import rich_seeds, mutate_exts  # noqa: E402


def pem_certs(path):
    import base64
    out, cur = [], None
    for line in open(path, "rb"):
        if line.startswith(b"-----BEGIN CERTIFICATE"):
            cur = []
        elif line.startswith(b"-----END CERTIFICATE") and cur is not None:
            out.append(base64.b64decode(b"".join(cur)))
            cur = None
        elif cur is not None:
            cur.append(line.strip())
    return out


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    chunk = 50_000
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20260923)
    now = synth.BASE_TIME
    cfg = synth.config(seed=11, n_issuers=8, ca_permille=100, expired_permille=100)
    cfgm = synth.config(seed=12, n_issuers=8, ca_permille=100, expired_permille=100, profile=1)
    issuers = synth.issuers(cfg)
    seeds = []
    for f in glob.glob(os.path.join(ROOT, "tests", "golden", "*.pem")):
        seeds += pem_certs(f)
    seeds += [synth.leaf(cfg, i)[0] for i in range(60)] + [synth.leaf(cfgm, i)[0] for i in range(60)]
    for bundle in ("/etc/ssl/certs/ca-certificates.crt",):
        if os.path.exists(bundle):
            seeds += pem_certs(bundle)[:80]
    n1 = D.name(D.rdn(6, b"US", 0x13), D.rdn(10, b"Edge Org"), D.rdn(3, b"Synth Issuer 000"))
    seeds += [D.cert(serial=bytes([k + 1]) * (k + 1), issuer=n1,
                     exts=[D.BC_NOT_CA, D.ext(0x1f, D.seq(D.seq(D.tlv(0xa0, D.tlv(0xa0, D.tlv(0x86, b"http://c.example/%d" % k))))))])
              for k in range(24)]
    if os.environ.get("STRICT_STRINGS"):   # Names whose string values sit at the edges of their character sets
        for tag, val in ((0x13, b"Org (EU) *&+,-./:=?'"), (0x12, b"0123 456"), (0x16, b"a@b.example"), (0x0c, "Zürich 東京".encode()),
                         (0x0c, b"\xf0\x9f\x98\x80\xed\x9f\xbf\xe0\xa0\x80"), (0x14, b"t61 \xe4")):
            seeds += [D.cert(serial=bytes([9, tag, k]), issuer=D.name(D.rdn(10, val, tag), D.rdn(3, b"Synth Issuer 000")),
                             subject=D.name(D.rdn(3, val, tag))) for k in range(4)]
    ext_mode = bool(os.environ.get("STRICT_EXT"))   # round 5: strict_extensions / the reference profile, extension-targeted damage
    if ext_mode:
        seeds += rich_seeds() * 40
    seeds += edge_seeds() * 6      # the Go-specific rules (numeric zones, lax INTEGERs, unique ids, high tags, lying wrappers): weighted up
    seeds += key_seeds() * 6       # round 4: keys of every algorithm parsePublicKey knows (RSA forms, the five curves, DSA)
    spans = []                     # where each seed's SubjectPublicKeyInfo lies: a third of the mutations aim there
    xspans = []                    # … and its extensions block (STRICT_EXT: a third of the mutations)
    for sd in seeds:
        c = orc.parse_cert(sd, strict_spki=False)
        spans.append((c.spki_off, c.spki_off + c.spki_len) if c.ok and c.spki_len else None)
        xspans.append((c.exts_off, c.exts_end) if c.ok and c.exts_end > c.exts_off else None)
    ca_seeds = [sd for sd in rich_seeds() if orc.parse_cert(sd).is_ca] + issuers[:2]
    print("seeds", len(seeds), flush=True)
    bad = 0
    done = 0
    t0 = time.time()
    while done < total:
        certs, iss, ets = [], [], []
        for r in range(chunk):
            k = rng.randrange(len(seeds))
            s = seeds[k]
            if ext_mode and xspans[k] and rng.randrange(3) == 0:
                m = mutate_exts(rng, s, *xspans[k])
            elif spans[k] and rng.randrange(3) == 0:
                m = spki_mutate(rng, s, *spans[k])
            else:
                m = mutate(rng, s) if rng.randrange(8) else s
            if rng.randrange(4) == 0 and len(m) > 1:
                m = mutate(rng, m)
            certs.append(m); iss.append(rng.randrange(len(issuers))); ets.append(rng.randrange(2))
        chunk_issuers = list(issuers)
        if ext_mode:                                 # the Chain[0] role: CA certificates with damaged extension bodies among the issuers
            for _ in range(8):
                ca = rng.choice(ca_seeds)
                co = orc.parse_cert(ca)
                chunk_issuers.append(mutate_exts(rng, ca, co.exts_off, co.exts_end) if rng.randrange(4) else ca)
            iss = [rng.randrange(len(chunk_issuers)) for _ in iss]
        batch = Batch.from_certs(certs, iss, ets)
        batch.payload = np.concatenate([batch.payload, np.zeros(N.PAYLOAD_PAD, np.uint8)])
        filt, log_exp = rng.choice(((b"", False), (b"Synth Issuer 00,Test", False), (b"zz,", True), (b"Synth", True)))
        eng = ctmr.Engine(device=0, table_slots=1 << 18, pair_slots=1 << 18, collect_meta=True)
        strict = (bool(os.environ.get("STRICT_STRINGS")) or ext_mode) and rng.random() < 0.5
        xstrict = ext_mode and rng.random() < 0.8
        spki = rng.random() < 0.85                   # the key parse (on by default) — and now and then off
        if ext_mode and strict and xstrict and spki and rng.random() < 0.5:
            eng.set_profile("reference")             # the four switches through the one call
        else:
            eng.set_strict_strings(strict)
            eng.set_strict_extensions(xstrict)
            eng.set_strict_spki(spki)
        eng.add_issuers(chunk_issuers)               # (after the switches: a Chain[0] is judged when it is registered)
        eng.set_filter(filt, log_exp, now)
        res = eng.map_batch(batch)
        o = orc.Engine(filt, log_exp, now)
        o.set_strict_strings(strict)
        o.set_strict_extensions(xstrict)
        o.set_strict_spki(spki)
        o, st, unk, eh = run_oracle(batch, chunk_issuers, filt, log_exp, now, engine=o)
        flags, serial_len, exp_hour, serial = expected_records(batch, st, unk, eh, strict, spki, xstrict)
        r = res.records
        diff = ((r["status"] != st) | (r["flags"] != flags) | (r["serial_len"] != serial_len) | (r["exp_hour"] != exp_hour) |
                (r["serial"] != serial).any(axis=1))
        nb = int(diff.sum()) + int(not np.array_equal(res.new_idx, np.nonzero(unk)[0]))
        if nb:
            bad += nb
            for i in np.nonzero(diff)[0][:5]:
                print("MISMATCH", i, "gpu", int(r["status"][i]), int(r["flags"][i]), "oracle", int(st[i]), int(flags[i]),
                      batch.cert(int(i)).hex()[:200], flush=True)
        # the IssuerMetadata memo over the hostile NEW certificates: first sightings as the reference's memo defines them
        canon = [eng.issuer_info(k).canonical_idx for k in range(len(chunk_issuers))]
        exp_meta = expected_first_sightings(certs, [canon[k] for k in iss], [int(k) for k in res.new_idx], r["exp_hour"])
        got_meta = got_first_sightings(eng, eng.meta_new())
        if got_meta != exp_meta:
            bad += 1
            print("META MISMATCH", sorted(got_meta ^ exp_meta, key=repr)[:4], flush=True)
        okeys = [k for k in o.keys() if k.startswith(b"serials::")]
        if sorted(eng.keys(b"serials::*")) != okeys or eng.total_count() != o.total_count():
            bad += 1
            print("STATE MISMATCH", flush=True)
        eng.close()
        done += chunk
        print(f"{done} certificates, {int((st == 0).sum())} PASS in the last chunk, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print("FUZZ", "OK" if bad == 0 else "FAILED", done, bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
