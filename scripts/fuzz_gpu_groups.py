"""Randomised stress of the multi-GPU rounds on the one reachable GPU (LOCAL groups: the same per-rank kernels and phase drivers
the RCCL transport runs): random world (1-4), random split points (empty shards included), several rounds through persisting
engines, serials of 1-40 octets, duplicates within shards, across shards and across rounds, mutated certificates — every
shard's records, NEW list and statistics, the per-issuer counts and the union of the ranks' sets against the single-stream oracle.
    gpurun -- 'python scripts/fuzz_gpu_groups.py 400'        # number of trials
"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ct_mapreduce_amd as ctmr  # noqa: E402
from ct_mapreduce_amd import synth, _native as N  # noqa: E402
from ct_mapreduce_amd.distributed import Group, shard  # noqa: E402
from ct_mapreduce_amd.engine import Batch, RECORD_DTYPE  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests import der as D  # noqa: E402
from tests.test_walk_cpu import mutate  # noqa: E402

DEV = torch.device("cuda:0")
NOW = synth.BASE_TIME


def to_dev(b):
    pay = torch.from_numpy(np.concatenate([b.payload, np.zeros(64, np.uint8)])).to(DEV)
    off = torch.from_numpy(b.offsets.astype(np.int64)).to(DEV)
    iss = torch.from_numpy(b.issuer_idx.astype(np.int32)).to(DEV)
    et = torch.from_numpy(b.entry_type.astype(np.uint8)).to(DEV)
    rec = torch.zeros(max(b.n, 1) * 32, dtype=torch.uint8, device=DEV)
    new = torch.zeros(max(b.n, 1), dtype=torch.int64, device=DEV)
    return pay, off, iss, et, rec, new


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20260923)
    cfg = synth.config(seed=31, n_issuers=6, dup_permille=0, ca_permille=30, expired_permille=30)
    issuers = synth.issuers(cfg)
    names = [D.name(D.rdn(3, b"Synth Issuer %03d" % k)) for k in range(len(issuers))]
    pool_synth = [(synth.leaf(cfg, i)[0], int(synth.host_batch(cfg, i, 1).issuer_idx[0])) for i in range(600)]
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    blob = np.frombuffer(b"".join(issuers), np.uint8)
    bad = entries = 0
    t0 = time.time()
    for trial in range(trials):
        world = rng.choice((1, 2, 2, 3, 4))
        mode = rng.choice(("owner", "bloom"))
        filt, log_exp = rng.choice(((b"", True), (b"Synth Issuer 00", False), (b"", False)))
        # a pool of certificates for this trial: synthetic ones, hand-built ones with serials of every length, mutated ones
        pool = list(rng.sample(pool_synth, 120))
        for _ in range(60):
            ln = rng.choice((1, 2, 8, 16, 17, 19, 20, 21, 22, 30, 39, 40, 41, 44))
            k = rng.randrange(len(issuers))
            s = bytes([rng.randrange(1, 0x7f)] + [rng.randrange(256) for _ in range(ln - 1)])
            pool.append((D.cert(serial=s, issuer=names[k], not_after=D.utctime("270101000000Z")), k))
        for _ in range(0 if os.environ.get("NO_MUTANTS") else 20):
            c, k = rng.choice(pool)
            pool.append((mutate(rng, c) if len(c) > 8 else c, k))
        engines = []
        for _ in range(world):
            e = ctmr.Engine(device=0, table_slots=1 << rng.choice((10, 12, 14)), pair_slots=1 << 10)
            if os.environ.get("NO_SPKI"):
                e.set_strict_spki(False)
            e.add_issuers(issuers)
            e.set_filter(filt, log_exp, NOW)
            engines.append(e)
        g = Group.local(engines)
        if mode == "bloom":
            g.bloom_config(1 << rng.choice((12, 14, 16)))
        big = False
        if mode == "owner" and rng.random() < 0.5:      # shards mapped in chunks (ctmr_group_set_chunks): the same answers
            g.set_chunks(rng.choice((2, 3, 5)))
            big = rng.random() < 0.4                     # … and rounds large enough for several non-empty chunks per shard
        o = orc.Engine(filt, log_exp, NOW)
        if os.environ.get("NO_SPKI"):
            o.set_strict_spki(False)
        base = 0
        ok = True
        for rnd in range(rng.choice((1, 2, 3))):
            n = rng.randrange(0, 700) if not big else rng.randrange(2500, 6000)
            items = [rng.choice(pool) for _ in range(n)]                  # with replacement: duplicates anywhere
            cuts = sorted(rng.randrange(0, n + 1) for _ in range(world - 1))
            bounds = [0] + cuts + [n]
            keep, shards, want = [], [], []
            for r in range(world):
                lo, hi = bounds[r], bounds[r + 1]
                b = Batch.from_certs([c for c, _ in items[lo:hi]], [k for _, k in items[lo:hi]],
                                     [rng.randrange(2) for _ in range(hi - lo)])
                pay = np.concatenate([b.payload, np.zeros(N.PAYLOAD_PAD, np.uint8)])
                st, unk, eh = o.batch(pay, b.offsets, b.issuer_idx, blob, io, entry_type=b.entry_type)   # log order, shard by shard
                want.append((st, unk))
                t = to_dev(b)
                keep.append(t)
                shards.append(shard(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), b.n, t[4].data_ptr(),
                                    t[5].data_ptr(), order_base=base + lo))
            stats = g.map_batch(mode, shards)
            for r in range(world):
                st, unk = want[r]
                m = len(st)
                rec = keep[r][4].cpu().numpy().view(RECORD_DTYPE)[:m]
                good = (rec["status"] == st).all() and (((rec["flags"] & 2) != 0) == (unk != 0)).all() and \
                    stats[r].n_new == int(unk.sum()) and (keep[r][5][:stats[r].n_new].cpu().numpy() == np.nonzero(unk)[0]).all()
                if not good:
                    ok = False
                    if os.environ.get("VERBOSE"):
                        lo_, hi_ = bounds[r], bounds[r + 1]
                        for q in np.nonzero(((rec["flags"] & 2) != 0) != (unk != 0))[0][:6]:
                            c = items[lo_ + int(q)][0]
                            pc = orc.parse_cert(c)
                            print("   entry", int(q), "of", m, "gpu flags", int(rec["flags"][q]), "oracle unk", int(unk[q]), "status", int(st[q]),
                                  "serial_len", pc.serial_len, "ec" if c.find(bytes.fromhex("2a8648ce3d0201")) >= 0 else "rsa",
                                  "copies in round", sum(1 for cc, _ in items if cc == c), flush=True)
                            for rr in range(world):
                                st2, unk2 = want[rr]
                                rec2 = keep[rr][4].cpu().numpy().view(RECORD_DTYPE)[:len(st2)]
                                for q2 in range(len(st2)):
                                    if items[bounds[rr] + q2][0] == c:
                                        print("      copy: rank", rr, "idx", q2, "gpu flags", int(rec2["flags"][q2]), "status", int(rec2["status"][q2]),
                                              "oracle unk", int(unk2[q2]), "et", int(rec2["flags"][q2]) & 1, flush=True)
                            nbad = sum(1 for rr in range(world) for q2 in range(len(want[rr][0]))
                                       if want[rr][0][q2] == 1 and items[bounds[rr] + q2][0].find(bytes.fromhex("2a8648ce3d0201")) >= 0)
                            print("      EC certificates with PARSE_ERROR in the round:", nbad, flush=True)
                    print(f"MISMATCH trial {trial} round {rnd} rank {r} world {world} mode {mode}: "
                          f"status {int((rec['status'] != st).sum())} flags {int((((rec['flags'] & 2) != 0) != (unk != 0)).sum())} "
                          f"n_new {stats[r].n_new} vs {int(unk.sum())}", flush=True)
            base += n
            entries += n
        tot = g.issuer_counts(len(issuers))
        for k in range(len(issuers)):
            ok = ok and int(tot[k]) == o.issuer_count(engines[0].issuer_id(k))
        ok = ok and g.total_count() == o.total_count() == sum(e.total_count() for e in engines)
        allkeys = sorted(sum((e.keys(b"serials::*") for e in engines), []))
        ok = ok and sorted(set(allkeys)) == [k for k in o.keys() if k.startswith(b"serials::")]
        if mode == "owner":      # every key lives on exactly one rank: the members of a set, united over the ranks, are the oracle's
            for key in sorted(set(allkeys))[:8]:
                members = sorted(sum((e.set_list(key) for e in engines), []))
                ok = ok and members == sorted(o.members(key))
        if not ok:
            bad += 1
            print(f"FAILED trial {trial}: world {world} mode {mode} filter {filt!r}", flush=True)
        g.close()
        for e in engines:
            e.close()
        if (trial + 1) % 25 == 0:
            print(f"{trial + 1} trials, {entries} entries, {bad} failed, {time.time() - t0:.0f} s", flush=True)
    print("FUZZ GROUPS", "OK" if bad == 0 else "FAILED", trials, entries, bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
