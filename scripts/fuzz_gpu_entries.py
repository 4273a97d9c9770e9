"""One-off fuzz campaign on the GPU box for the raw get-entries path (k_decode_match → map/reduce):
damaged TLS framing and damaged certificate bodies against the oracle's LogEntryFromLeaf + insertCTWorker restatement.
    gpurun -- 'python scripts/fuzz_gpu_entries.py 1000000'
    gpurun -- 'STRICT_LEAF=1 python scripts/fuzz_gpu_entries.py 1000000'     # half the engines in strict_leaf mode (round 3)
    gpurun -- 'STRICT_STRINGS=1 python scripts/fuzz_gpu_entries.py 1000000'  # half the engines in strict_strings mode (round 3)
    gpurun -- 'REFERENCE=1 python scripts/fuzz_gpu_entries.py 1000000'       # half the engines under ctmr_set_profile(REFERENCE); seed
                                                                             # entries that carry every extension body it looks into (round 5)
"""
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
from ct_mapreduce_amd.engine import RawEntries  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.test_entry_decode_cpu import mutate_entry  # noqa: E402
from tests.test_walk_cpu import mutate  # noqa: E402
from tests.test_ext_cpu import rich_seeds, mutate_exts  # noqa: E402
from tests.test_gpu_ext import synth_x509_entry, synth_precert_entry  # noqa: E402


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 6962)
    chunk = 40_000
    now = synth.BASE_TIME
    seeds = []
    for seed, prof in ((21, 0), (22, 1)):
        cfg = synth.config(seed=seed, n_issuers=12, dup_permille=100, ca_permille=50, expired_permille=50, profile=prof)
        raw = synth.host_entries(cfg, 0, 1500)
        seeds += [(raw.leaf_input(i), raw.extra_data(i)) for i in range(raw.n)]
    reference = bool(os.environ.get("REFERENCE"))
    if reference:   # certificates with subjectAltName URIs, CRL distribution points, name constraints, SCT lists — as X509 and
        cas = [c for c in rich_seeds() if orc.parse_cert(c).is_ca]          # precertificate entries under issuers with and without them
        for c in rich_seeds():
            o = orc.parse_cert(c)
            for ca in cas + [synth.issuers(synth.config(seed=21, n_issuers=12))[0]]:
                seeds += [synth_x509_entry(c, ca), synth_precert_entry(c, o, ca)] * 40
    print("seed entries", len(seeds), flush=True)
    bad = done = 0
    t0 = time.time()
    while done < total:
        pairs = []
        for _ in range(chunk):
            leaf, extra = seeds[rng.randrange(len(seeds))]
            k = rng.randrange(10)
            if k < 4:
                leaf, extra = mutate_entry(rng, leaf, extra)
            elif reference and k < 8 and len(extra) > 300:   # extension-targeted damage in the (pre)certificate or Chain[0], framing intact
                cut = rng.randrange(60, len(extra) - 60)
                extra = extra[:cut] + mutate_exts(rng, extra[cut:cut + 200], 0, min(200, len(extra) - cut)) + extra[cut + 200:]
            elif k < 6 and len(leaf) > 20:              # damage inside the certificate / TBS bytes, framing intact
                leaf = leaf[:15] + mutate(rng, leaf[15:])[:len(leaf) - 15].ljust(len(leaf) - 15, b"\0")
            elif k < 7 and len(extra) > 12:             # damage inside the chain certificates, framing intact
                extra = extra[:6] + mutate(rng, extra[6:])[:len(extra) - 6].ljust(len(extra) - 6, b"\0")
            pairs.append((leaf, extra))
        raw = RawEntries.from_pairs(pairs)
        raw.blob = np.concatenate([raw.blob, np.zeros(N.PAYLOAD_PAD, np.uint8)])
        filt, log_exp = rng.choice(((b"", True), (b"Synth Issuer 00", False), (b"", False)))
        strict = bool(os.environ.get("STRICT_LEAF")) and rng.random() < 0.5
        eng = ctmr.Engine(device=0, table_slots=1 << 18, pair_slots=1 << 18, collect_meta=True)
        eng.set_filter(filt, log_exp, now)
        eng.set_profile("fast")     # (an engine is created under the reference profile since round 6: every switch is set here)
        eng.set_strict_leaf(strict)
        strings = bool(os.environ.get("STRICT_STRINGS")) and rng.random() < 0.5
        eng.set_strict_strings(strings)
        ref = reference and rng.random() < 0.6
        if ref:
            eng.set_profile("reference")
            strict = strings = True
        res = eng.map_entries(raw)
        o = orc.Engine(filt, log_exp, now)
        o.set_profile("fast")
        o.set_strict_leaf(strict)
        o.set_strict_strings(strings)
        o.set_strict_extensions(ref)
        st, unk, eh, ts = o.raw_batch(raw.blob, raw.bounds)
        r = res.records
        parsed = (st != orc.ST_PARSE_ERROR) & (st != orc.ST_ENTRY_DECODE_ERROR)
        diff = (r["status"] != st) | (((r["flags"] & 2) != 0) != (unk != 0)) | (parsed & (r["exp_hour"] != eh)) | (res.timestamp != ts)
        nb = int(diff.sum()) + int(not np.array_equal(res.new_idx, np.nonzero(unk)[0]))
        okeys = [k for k in o.keys() if k.startswith(b"serials::")]
        nb += int(sorted(eng.keys(b"serials::*")) != okeys) + int(eng.total_count() != o.total_count())
        if nb:
            bad += nb
            for i in np.nonzero(diff)[0][:5]:
                print("MISMATCH", i, "gpu", int(r["status"][i]), "oracle", int(st[i]), flush=True)
        eng.meta_new()
        eng.close()
        done += chunk
        hist = [int((st == k).sum()) for k in range(8)]
        print(f"{done} entries{' (reference profile)' if ref else ' (strict_leaf)' if strict else ''}, status histogram {hist}, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print("FUZZ", "OK" if bad == 0 else "FAILED", done, bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
