"""What the exact global-dedup modes cost ONE rank in kernels when the world is larger than one — measured on the one
reachable GPU by driving the per-rank steps (ctmr_xchg_* / ctmr_bloom_*) directly with `world = W, rank = 0`:
  owner  map with W−1 of W keys leaving as 32-byte records (staging + partition gather), then the owner-side insert of as
         many RECEIVED records as a rank gets at steady state (its own exported records stand in for the peers': same
         count, same key distribution), the resolve, and the apply of the returned bytes
  bloom  map + filter add, probe of the locally-new keys against W−1 peer filters (empty ones: the probe's cost does not
         depend on what they hold), apply
Wire time is NOT in here (there is no second device): this is the kernel side of "what exactness costs before a byte
crosses xGMI" (VERDICT r02 weak #5).  The sets this leaves behind are meaningless (keys inserted on the wrong owner).

    python scripts/rank_cost_at_world.py ENTRIES WORLD [steps]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

import ct_mapreduce_amd as ctmr  # noqa: E402
from ct_mapreduce_amd import synth, _native as N  # noqa: E402
from ct_mapreduce_amd.distributed import shard  # noqa: E402


def pow2(v):
    p = 1
    while p < v:
        p <<= 1
    return p


def main():
    E, W = int(sys.argv[1]), int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dev = torch.device("cuda:0")
    cfg = synth.config(seed=20260921 + 4, n_issuers=256, zipf=1, dup_permille=100, ca_permille=10, expired_permille=10)
    eng = ctmr.Engine(device=0, table_slots=pow2(2 * E), pair_slots=1 << 22, profile=True)
    eng.add_issuers(synth.issuers(cfg))
    eng.set_filter(b"Synth Issuer 0,Synth Issuer 1", False, synth.BASE_TIME)
    d_off = torch.empty(E + 1, dtype=torch.int64, device=dev)
    total = eng.synth_device(cfg, 0, E, d_off.data_ptr(), 0, 0, 0, 0)
    d_pay = torch.empty(total + N.PAYLOAD_PAD + 16, dtype=torch.uint8, device=dev)
    d_iss = torch.empty(E, dtype=torch.int32, device=dev)
    d_et = torch.empty(E, dtype=torch.uint8, device=dev)
    eng.synth_device(cfg, 0, E, d_off.data_ptr(), d_pay.data_ptr(), d_pay.numel(), d_iss.data_ptr(), d_et.data_ptr())
    d_rec = torch.empty(E * 32, dtype=torch.uint8, device=dev)
    d_new = torch.empty(E, dtype=torch.int64, device=dev)
    d_keys = torch.empty(E * 32, dtype=torch.uint8, device=dev)
    d_fl = torch.empty(E, dtype=torch.uint8, device=dev)
    sh = shard(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), E, d_rec.data_ptr(), d_new.data_ptr())
    out = {"entries": E, "world": W, "steps": steps}

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, (time.perf_counter() - t0) * 1e3

    # ---- plain reduce (the yardstick)
    ms = []
    for _ in range(steps + 1):
        eng.reset_known()
        st, t = timed(lambda: eng.map_batch_device(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), E,
                                                   d_rec.data_ptr(), d_new.data_ptr()))
        ms.append(t)
    out["plain_ms"] = min(ms[1:])
    # ---- owner-computes, this rank's side
    ph = []
    for _ in range(steps + 1):
        eng.reset_known()
        (counts, n_long), t_map = timed(lambda: eng.xchg_map(sh, W, 0, 0))
        n32 = sum(counts)
        _, t_keys = timed(lambda: eng.xchg_keys(W, d_keys.data_ptr()))
        _, t_ins = timed(lambda: eng.xchg_insert(d_keys.data_ptr(), n32, d_fl.data_ptr()))
        st, t_app = timed(lambda: eng.xchg_apply(d_keys.data_ptr(), d_fl.data_ptr(), n32))
        ph.append((t_map, t_keys, t_ins, t_app, n32, int(st.n_new)))
    best = min(ph[1:], key=lambda p: sum(p[:4]))
    out["owner"] = {"map_and_local_insert_and_counts_ms": best[0], "partition_gather_ms": best[1],
                    "owner_insert_of_as_many_records_and_resolve_ms": best[2], "apply_and_compaction_ms": best[3],
                    "sum_ms": sum(best[:4]), "records_exported": best[4], "record_bytes_to_the_wire": best[4] * 32 + best[4],
                    "n_new_after_apply": best[5]}
    # ---- Bloom variant, this rank's side
    bits = pow2(16 * E)
    d_filters = torch.zeros(W * bits // 8, dtype=torch.uint8, device=dev)
    eng.bloom_config(bits, d_filters.data_ptr())          # rank 0's row of the gather buffer; the peers' rows stay empty
    d_k64 = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    ph = []
    for _ in range(steps + 1):
        eng.reset_known()
        st, t_map = timed(lambda: eng.map_batch_device(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), E,
                                                       d_rec.data_ptr(), 0))
        _, t_add = timed(lambda: eng.bloom_add(d_pay.data_ptr(), d_off.data_ptr(), 0, E, d_rec.data_ptr()))
        (cnt, fits), t_probe = timed(lambda: eng.bloom_probe(d_pay.data_ptr(), d_off.data_ptr(), 0, E, d_rec.data_ptr(),
                                                             d_filters.data_ptr(), W, 0, 0, d_k64.data_ptr(), (1 << 20) // 64))
        st2, t_app = timed(lambda: eng.bloom_apply(d_rec.data_ptr(), E, 0, 0, 0, d_new.data_ptr()))
        ph.append((t_map, t_add, t_probe, t_app, sum(cnt)))
    best = min(ph[1:], key=lambda p: sum(p[:4]))
    out["bloom"] = {"map_insert_and_filter_add_ms": best[0], "round_open_ms": best[1], "probe_of_peer_filters_ms": best[2],
                    "apply_and_compaction_ms": best[3], "sum_ms": sum(best[:4]), "filter_bytes_to_gather_per_rank": bits // 8,
                    "candidate_records": best[4]}
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    main()
