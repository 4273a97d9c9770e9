"""A LOCAL group of three engines on the one reachable GPU: do the ranks' kernels run together?  (the
round-2 drivers called the ranks one after the other, each call ending in a stream synchronisation.)

    rocprofv3 --kernel-trace -d OUT -o trace --output-format csv -- python scripts/local_group_overlap.py run
    python scripts/local_group_overlap.py report OUT
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    import ct_mapreduce_amd as ctmr
    from ct_mapreduce_amd import synth, _native as N
    from ct_mapreduce_amd.distributed import Group, shard
    dev = torch.device("cuda:0")
    world, n = 3, 2_000_000
    cfg = synth.config(seed=20260921 + 4, n_issuers=256, zipf=1, dup_permille=100, ca_permille=10, expired_permille=10)
    issuers = synth.issuers(cfg)
    engines, keep, shards = [], [], []
    for r in range(world):
        e = ctmr.Engine(device=0, table_slots=1 << 23, pair_slots=1 << 16)
        e.add_issuers(issuers)
        e.set_filter(b"Synth Issuer 0,Synth Issuer 1", False, synth.BASE_TIME)
        d_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
        total = e.synth_device(cfg, r * n, n, d_off.data_ptr(), 0, 0, 0, 0)
        d_pay = torch.empty(total + N.PAYLOAD_PAD + 16, dtype=torch.uint8, device=dev)
        d_iss = torch.empty(n, dtype=torch.int32, device=dev)
        d_et = torch.empty(n, dtype=torch.uint8, device=dev)
        e.synth_device(cfg, r * n, n, d_off.data_ptr(), d_pay.data_ptr(), d_pay.numel(), d_iss.data_ptr(), d_et.data_ptr())
        d_rec = torch.empty(n * 32, dtype=torch.uint8, device=dev)
        d_new = torch.empty(n, dtype=torch.int64, device=dev)
        engines.append(e)
        keep.append((d_off, d_pay, d_iss, d_et, d_rec, d_new))
        shards.append(shard(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), n, d_rec.data_ptr(),
                            d_new.data_ptr(), order_base=r * n))
    g = Group.local(engines)
    for mode in ("local", "owner"):
        for _ in range(3):
            for e in engines:
                e.reset_known()
            torch.cuda.synchronize()
            st = g.map_batch(mode, shards)
        print(mode, [int(s.n_new) for s in st], flush=True)
    g.close()
    for e in engines:
        e.close()


def report(outdir):
    rows = []
    for f in glob.glob(os.path.join(outdir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_map_fused" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", r.get("Stream_Id", "?"))))
    rows.sort()
    print(f"{len(rows)} k_map_fused launches (3 ranks x 6 rounds expected)")
    # group into rounds of three launches that start within one another's lifetime
    k = 0
    tot_union = tot_sum = 0
    while k + 3 <= len(rows):
        grp = rows[k:k + 3]
        k += 3
        t0 = min(s for s, _, _ in grp)
        t1 = max(e for _, e, _ in grp)
        ssum = sum(e - s for s, e, _ in grp)
        tot_union += t1 - t0
        tot_sum += ssum
        print("round: " + "  ".join(f"queue {q}: +{(s - t0) / 1e3:8.1f} .. +{(e - t0) / 1e3:8.1f} us" for s, e, q in grp)
              + f"   union {(t1 - t0) / 1e3:.1f} us, sum of the three {ssum / 1e3:.1f} us")
    if tot_union:
        print(f"all rounds: the three ranks' map kernels are in flight together — wall span of a round's three launches = "
              f"{tot_union / tot_sum:.2f} x the sum of their durations (1/3 = perfectly concurrent, 1 = one after the other)")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
