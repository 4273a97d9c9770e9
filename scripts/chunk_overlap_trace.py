"""Do the key records of chunk c travel while chunk c + 1 is being mapped?  (VERDICT r03 #5a.)  A LOCAL group of two
engines on the one reachable GPU, owner-computes rounds, shards mapped whole (chunks = 1) and in four chunks; the
rocprofv3 kernel trace says which device-to-device copies of key records ran while a map kernel was in flight.

    rocprofv3 --kernel-trace -d OUT -o trace --output-format csv -- python scripts/chunk_overlap_trace.py run
    python scripts/chunk_overlap_trace.py report OUT
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
    world, n = 2, 4_000_000
    cfg = synth.config(seed=20260921 + 4, n_issuers=256, zipf=1, dup_permille=100, ca_permille=10, expired_permille=10)
    issuers = synth.issuers(cfg)
    for chunks in (1, 4):
        engines, keep, shards = [], [], []
        for r in range(world):
            e = ctmr.Engine(device=0, table_slots=1 << 24, pair_slots=1 << 16)
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
        g.set_chunks(chunks)
        for _ in range(3):
            for e in engines:
                e.reset_known()
            torch.cuda.synchronize()
            st = g.map_batch("owner", shards)
        print("chunks", chunks, "n_new", [int(s.n_new) for s in st], "ms_phase", [round(x, 2) for x in g.info().ms_phase[:7]], flush=True)
        g.close()
        for e in engines:
            e.close()
        del keep, shards
        torch.cuda.empty_cache()


def report(outdir):
    # on one device a device-to-device hipMemcpyAsync runs as a blit KERNEL (__amd_rocclr_copyBuffer), not as an SDMA copy:
    # both the maps and the key-record copies are in the kernel trace
    maps, copies = [], []
    for f in glob.glob(os.path.join(outdir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            iv = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
            if "k_map_fused" in r["Kernel_Name"]:
                maps.append(iv)
            elif "copyBuffer" in r["Kernel_Name"] and iv[1] - iv[0] > 20_000:   # tens of MB of key records, not control words
                copies.append(iv)
    maps.sort()
    copies.sort()
    print(f"{len(maps)} k_map_fused launches, {len(copies)} device-to-device copies longer than 20 us")
    # split the trace at the largest gap between map launches: first part = chunks 1, second = chunks 4
    gaps = sorted(((maps[i + 1][0] - maps[i][1], i) for i in range(len(maps) - 1)), reverse=True)
    cut = maps[gaps[0][1]][1] if gaps else 0
    for name, lo, hi in (("shards mapped whole (chunks = 1)", 0, cut), ("shards mapped in 4 chunks", cut, 1 << 62)):
        cs = [c for c in copies if lo <= c[0] < hi]
        ms = [m for m in maps if lo <= m[0] < hi]
        inside = tot = 0
        for s, e in cs:
            tot += e - s
            cov = 0
            for a, b in ms:
                cov += max(0, min(e, b) - max(s, a))
            inside += min(cov, e - s)
        print(f"{name}: {len(ms)} map launches, {len(cs)} key-record copies, {tot / 1e3:.0f} us of copying, "
              f"{inside / 1e3:.0f} us of it ({100.0 * inside / max(tot, 1):.0f} %) while a map kernel was running")


def timeline(outdir):
    """The last chunked round, event by event: which queue ran what (the engines' streams map, the transfer streams copy)."""
    rows = []
    for f in glob.glob(os.path.join(outdir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n, s0, e0 = r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            kind = "MAP" if "k_map_fused" in n else "gather" if "k_key_gather" in n else \
                "copy" if ("copyBuffer" in n and e0 - s0 > 4000) else "insert" if "k_keys_insert<" in n else None
            if kind:
                rows.append((s0, e0, kind, r["Queue_Id"]))
    rows.sort()
    maps = [r for r in rows if r[2] == "MAP"]
    t0 = maps[-8][0]
    print("the last round (2 ranks x 4 chunks), microseconds from its first map launch; queue = HIP stream:")
    for s0, e0, k, q in rows:
        if s0 >= t0 - 1000:
            print(f"  {(s0 - t0) / 1e3:9.1f} .. {(e0 - t0) / 1e3:9.1f}  {k:6s} queue {q}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
        timeline(sys.argv[2])
