#!/usr/bin/env python3
"""bench.py — certificates/sec + achieved HBM GB/s of the map/reduce hot path on MI355X.

A "step" is one pass of the hot path (packed DER → TBS walk → 3 filters → known-certificate set
insert → WasUnknown resolve → per-issuer counts → NEW-list compaction) over one synthetic CT
batch that is already resident in HBM.  The known-certificate table is cleared inside every
timed step, so every step does the same "first sighting" work.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: the entry stream is sharded by log-index range (rank r owns [r·E, (r+1)·E), weak scaling);
the only data-path collective of the default mode is the RCCL all-reduce of the per-issuer count vector, issued by
the library itself (ctmr_group_issuer_counts); torch.distributed only carries the 128-byte group id at start-up.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
ALG_BYTES_FIXED = 45     # offsets 8 + issuer_idx 4 + entry_type 1 + record 32


MAP_KERNELS = {1: "k_map_tile", 2: "k_map_direct", 13: "k_map_winc<16>", 15: "k_map_fused<16, false>"}   # 1, 2: sweep build only
DEFAULT_VARIANT = 15
FUSED = (15,)          # map kernels that also do pass 1 of the known-certificate insert
ALG_BYTES_PROBE = 64   # per PASS entry: 32 B slot read + 32 B slot write — fused kernels only


def pow2_at_least(v):
    p = 1
    while p < v:
        p <<= 1
    return p


def cpu_baseline(batch_arrays, issuers, filt, now, sample, entry_type=None):
    """The oracle's restatement of the reference loop, timed on one host core (kind "port"), over `sample` entries of
    the SAME batch the GPU processed (payload, offsets, issuer_idx[, entry_type])."""
    import numpy as np
    from oracle import oracle as orc
    payload, offsets, issuer_idx = batch_arrays
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    blob = np.frombuffer(b"".join(issuers), np.uint8)
    o = orc.Engine(filt, False, now)
    t0 = time.perf_counter()
    st, unk, eh = o.batch(payload, offsets, issuer_idx, blob, io, entry_type=entry_type)
    dt = time.perf_counter() - t0
    return {"value": sample / dt, "unit": "certificates/sec", "cores": 1, "kind": "port",
            "sample": f"{sample} entries, oracle/ctmr_oracle.c "
                      f"(in-process hash set stands in for Redis; not the Go binary), {dt:.1f} s",
            "host_cores_available": os.cpu_count()}, (st, unk)


def cpu_quota():
    """CPUs this process may use at once: the cgroup quota when there is one (a 256-thread box may be shared out in
    16-CPU pods), else the affinity mask."""
    aff = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]            # cgroup v2
        if q != "max":
            return min(aff, max(1, -(-int(q) // int(per)))), int(q) / int(per)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())           # cgroup v1
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return min(aff, max(1, -(-q // per))), q / per
    except (OSError, ValueError):
        pass
    return aff, None


def cpu_baseline_threads(batch_arrays, issuers, filt, now, sample, threads):
    """The same restatement on `threads` host threads: contiguous slices of the sample, one oracle engine (its own
    in-process sets) per thread — T reference processes with -offset/-limit, minus the Redis they would share, so
    this flatters the CPU side slightly.  ctypes releases the GIL for the duration of each call."""
    import threading
    import numpy as np
    from oracle import oracle as orc
    payload, offsets, issuer_idx = batch_arrays
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    blob = np.frombuffer(b"".join(issuers), np.uint8)
    engines = [orc.Engine(filt, False, now) for _ in range(threads)]
    bounds = [sample * t // threads for t in range(threads + 1)]
    gate = threading.Barrier(threads + 1)
    n_pass = [0] * threads

    def work(t):
        lo, hi = bounds[t], bounds[t + 1]
        gate.wait()
        if hi > lo:
            st, _, _ = engines[t].batch(payload, offsets[lo:hi + 1], issuer_idx[lo:hi], blob, io)
            n_pass[t] = int((st == 0).sum())
        gate.wait()

    ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for th in ths:
        th.start()
    gate.wait()
    t0 = time.perf_counter()
    gate.wait()
    dt = time.perf_counter() - t0
    for th in ths:
        th.join()
    for e in engines:
        e.close()
    return sample / dt, dt, sum(n_pass)


def _mix64(z, np):
    z = z + np.uint64(0x9e3779b97f4a7c15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
    return z ^ (z >> np.uint64(31))


def synth_is_dup(seed, first, n, dup_permille, np):
    """numpy restatement of csrc/synth.h synth_is_dup for entries [first, first+n)."""
    with np.errstate(over="ignore"):
        i = np.arange(first, first + n, dtype=np.uint64)
        base = _mix64(np.uint64(seed) ^ np.uint64((1 * 0xd6e8feb86659fd93) & 0xffffffffffffffff), np)
        h = _mix64(base + i, np)
        return (i > 0) & ((h % np.uint64(1000)) < np.uint64(dup_permille))


def lib_hash():
    """sha256 of the library the kernels come from: traffic measured on another build is refused."""
    import hashlib
    from ct_mapreduce_amd import _native as N
    return hashlib.sha256(open(N.LIB_PATH, "rb").read()).hexdigest()[:16]


def parse_pmc_csv(outdir, counter, kernel_substr):
    """Average per-launch value of one rocprofv3 --pmc counter over the launches of one kernel."""
    import csv
    import glob
    vals = []
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter and kernel_substr in row.get("Kernel_Name", ""):
                vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def measure_traffic(args, entries, kernel_substr):
    """HBM traffic of the map kernel, measured NOW on this build: re-executes this script on a smaller batch of the same
    corpus under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, no trace flags:
    MI355X_MICROARCH.md §HBM / §rocprofv3 PMC slots) and returns bytes per certificate.  gfx950 correction: FETCH_SIZE
    tallies 128-byte requests at 64 bytes — x2 (calibrated on this access pattern too: scripts/calib_fetch.hip,
    profiles/r01/s2); WRITE_SIZE as is.  Both are reported in KiB by rocprofv3."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    tmp = tempfile.mkdtemp(prefix="ctmr_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    env = dict(os.environ, CTMR_BENCH_CHILD="1", TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    child = [sys.executable, os.path.abspath(__file__), "--entries", str(entries), "--steps", "2", "--warmup", "1",
             "--no-cpu", "--traffic", "off", "--issuers", str(args.issuers), "--variant", str(args.variant),
             "--dup-permille", str(args.dup_permille)] + (["--mixed"] if args.mixed else [])
    out = {"entries": entries}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(tmp, c)
        try:
            r = subprocess.run([exe, "--pmc", c, "-d", d, "-o", "pmc", "--output-format", "csv", "--"] + child,
                               cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=420)
        except subprocess.TimeoutExpired:
            return None, f"rocprofv3 --pmc {c} pass timed out"
        v, n = parse_pmc_csv(d, c, kernel_substr)
        if v is None:
            return None, f"rocprofv3 --pmc {c}: no rows for {kernel_substr} (rc {r.returncode}): " + r.stdout.decode(errors="replace")[-300:]
        out[c + "_KB_per_launch"] = v
        out["launches"] = n
    shutil.rmtree(tmp, ignore_errors=True)
    out["fetch_bytes_per_cert"] = 2.0 * out["FETCH_SIZE_KB_per_launch"] * 1024.0 / entries
    out["write_bytes_per_cert"] = out["WRITE_SIZE_KB_per_launch"] * 1024.0 / entries
    out["traffic_bytes_per_cert"] = out["fetch_bytes_per_cert"] + out["write_bytes_per_cert"]
    out["lib_sha256_16"] = lib_hash()
    return out, None


def needed_bytes_per_cert(certs, starts, filt):
    """What the walk READS of a certificate (the product's walk compiled for the host with a marking reader,
    tests/harness): bytes covered by its reads, and the distinct 128-byte HBM lines they lie in at the certificate's
    real position in the payload — the floor of the map kernel's fetch traffic at line granularity."""
    from tests import harness
    nb = nl = 0
    for der, st in zip(certs, starts):
        _, b, l = harness.walk_touched(der, int(st) & 127, filt)
        nb += b
        nl += l
    return nb / len(certs), nl * 128.0 / len(certs)


def strided_sample(E, slices, per_slice):
    """[lo, hi) ranges of `slices` equally spaced slices of `per_slice` entries over [0, E)."""
    per_slice = min(per_slice, max(1, E // slices))
    return [(k * (E // slices), k * (E // slices) + per_slice) for k in range(slices)]


def gather_sample(d_off, d_pay, d_iss, d_et, ranges, extra_idx, extra_certs, pad, np):
    """One host batch = the sampled slices copied back from HBM + `extra` single entries (index → (der, issuer_idx,
    entry_type)), all in ascending log-index order.  Returns (payload, offsets, issuer_idx, entry_type, global_index)."""
    pieces = []          # (first_index, payload u8, lens u64, iss u32, et u8)
    for lo, hi in ranges:
        offs = d_off[lo:hi + 1].cpu().numpy().astype(np.uint64)
        pay = d_pay[int(offs[0]):int(offs[-1])].cpu().numpy()
        pieces.append((lo, pay, np.diff(offs), d_iss[lo:hi].cpu().numpy().astype(np.uint32),
                       d_et[lo:hi].cpu().numpy().astype(np.uint8), np.arange(lo, hi, dtype=np.uint64)))
    for i, (der, iss, et) in zip(extra_idx, extra_certs):
        pieces.append((int(i), np.frombuffer(der, np.uint8), np.array([len(der)], np.uint64),
                       np.array([iss], np.uint32), np.array([et], np.uint8), np.array([i], np.uint64)))
    pieces.sort(key=lambda t: t[0])
    payload = np.concatenate([t[1] for t in pieces] + [np.zeros(pad, np.uint8)])
    lens = np.concatenate([t[2] for t in pieces])
    offsets = np.zeros(len(lens) + 1, np.uint64)
    offsets[1:] = np.cumsum(lens)
    return (payload, offsets, np.concatenate([t[3] for t in pieces]), np.concatenate([t[4] for t in pieces]),
            np.concatenate([t[5] for t in pieces]))


def synth_src(seed, idx, dup_permille, np):
    """numpy restatement of csrc/synth.h synth_src: the entry whose key entry i repeats (i itself if it is no duplicate)."""
    with np.errstate(over="ignore"):
        idx = np.asarray(idx, dtype=np.uint64)
        src = idx.copy()
        dup = synth_is_dup_at(seed, idx, dup_permille, np)
        base = _mix64(np.uint64(seed) ^ np.uint64((2 * 0xd6e8feb86659fd93) & 0xffffffffffffffff), np)
        j = _mix64(base + idx[dup], np) % idx[dup]
        while True:                                  # walk down to the nearest entry that is no duplicate itself
            d = synth_is_dup_at(seed, j, dup_permille, np)
            if not d.any():
                break
            j = np.where(d, j - np.uint64(1), j)
        src[dup] = j
        return src, dup


def synth_is_dup_at(seed, idx, dup_permille, np):
    with np.errstate(over="ignore"):
        idx = np.asarray(idx, dtype=np.uint64)
        base = _mix64(np.uint64(seed) ^ np.uint64((1 * 0xd6e8feb86659fd93) & 0xffffffffffffffff), np)
        h = _mix64(base + idx, np)
        return (idx > 0) & ((h % np.uint64(1000)) < np.uint64(dup_permille))


def share_group_id(dist, rank, make_id):
    """Control path of `--gpus N`: rank 0 makes the 128-byte group id (ncclGetUniqueId through the library), every rank
    receives it — the one thing the host carries between its processes; the data path never touches torch.distributed."""
    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def max_over_ranks(dist, seconds, dev):
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def run_stream(args, ctmr, synth, N, torch, np, dev, local, rank, world, cfg, filt, now, issuers):
    """BASELINE config 5 on ONE GPU (the cross-GPU form is distributed.run_global_dedup): a long stream with 10 %
    duplicates, the known-certificate table persisting across waves."""
    T = args.stream
    W = args.entries if args.entries != 100_000_000 else 50_000_000
    W = min(W, T)
    cfg = synth.config(seed=20260921 + 5, n_issuers=args.issuers, zipf=1, dup_permille=100, ca_permille=10,
                       expired_permille=10)
    slots = pow2_at_least(int(T * 1.6))
    eng = ctmr.Engine(device=local, table_slots=min(slots, 1 << 31), pair_slots=1 << 22, map_variant=args.variant,
                      profile=True)
    eng.add_issuers(synth.issuers(cfg))
    eng.set_filter(filt, False, now)
    d_off = torch.empty(W + 1, dtype=torch.int64, device=dev)
    d_iss = torch.empty(W, dtype=torch.int32, device=dev)
    d_et = torch.empty(W, dtype=torch.uint8, device=dev)
    d_rec = torch.empty(W * 32, dtype=torch.uint8, device=dev)
    d_new = torch.empty(W, dtype=torch.int64, device=dev)
    d_pay = torch.empty(int(W * 1600) + 4096, dtype=torch.uint8, device=dev)
    t_gpu = t_map = 0.0
    tot_new = tot_dup = tot_pass = tot_bytes = 0
    ok = True
    waves = 0
    first = 0
    while first < T:
        n = min(W, T - first)
        eng.synth_device(cfg, first, n, d_off.data_ptr(), d_pay.data_ptr(), d_pay.numel(), d_iss.data_ptr(),
                         d_et.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = eng.map_batch_device(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), n,
                                  d_rec.data_ptr(), d_new.data_ptr())
        t_gpu += time.perf_counter() - t0
        t_map += st.ms_map
        # the generator's structure: entry i duplicates an EARLIER entry's key iff synth_is_dup(i) — in this wave or
        # any earlier one — so PASS ∧ dup must be known and PASS ∧ ¬dup must be new, wave by wave
        status = d_rec.view(-1, 32)[:n, 0].cpu().numpy()
        dup = synth_is_dup(cfg.seed, first, n, 100, np)
        exp_new = int(((status == 0) & ~dup).sum())
        exp_dup = int(((status == 0) & dup).sum())
        good = exp_new == int(st.n_new) and exp_dup == int(st.n_dup)
        ok = ok and good
        sys.stderr.write(f"stream: wave {waves} [{first}, {first + n}) new {st.n_new} dup {st.n_dup} "
                         f"({'ok' if good else 'MISMATCH: expected %d/%d' % (exp_new, exp_dup)}) "
                         f"map {st.ms_map:.2f} ms total {st.ms_total:.2f} ms\n")
        tot_new += int(st.n_new); tot_dup += int(st.n_dup); tot_pass += int(st.by_status[0])
        tot_bytes += int(st.payload_bytes) + ALG_BYTES_FIXED * n + ALG_BYTES_PROBE * int(st.by_status[0])
        first += n
        waves += 1
    ok = ok and eng.total_count() == tot_new
    achieved = tot_bytes / (t_map * 1e-3) / 1e9
    out = {"metric": "certificates/sec whole-node + achieved HBM GB/s, 100M-entry synthetic CT batch",
           "value": T / t_gpu, "unit": "certificates/sec", "n_gpus": 1, "steps": waves, "warmup": 0,
           "ms_per_step": t_gpu / waves * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u8", "data": "synthetic",
           "config": {"workload": f"STREAM of {T} entries with 10% duplicates in {waves} waves of {W} through one "
                                  "known-certificate table (BASELINE configs[4] on one GPU); generation untimed",
                      "table_slots": int(min(slots, 1 << 31)), "map_variant": args.variant or DEFAULT_VARIANT},
           "roofline": {"bound": "hbm", "kernel": MAP_KERNELS[args.variant or DEFAULT_VARIANT], "achieved": achieved,
                        "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                        "alg_bytes_formula": "sum(L_i) + 45*E + 64*PASS (table probe), summed over the waves"},
           "result": {"n_new": tot_new, "n_dup": tot_dup, "n_pass": tot_pass, "total_count": eng.total_count(),
                      "duplicate_structure_matches_generator_in_every_wave": bool(ok)}}
    print(json.dumps(out))
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--entries", type=int, default=int(os.environ.get("CTMR_BENCH_ENTRIES", 100_000_000)),
                    help="entries per GPU (weak scaling); default = BASELINE's 100M-entry batch "
                         "(≈152 GB of DER resident in HBM); halved automatically if it does not fit")
    ap.add_argument("--issuers", type=int, default=256)
    ap.add_argument("--table-slots-log2", type=int, default=0,
                    help="known-certificate table size (default: the power of two >= 2 x entries: load 0.35-0.47 when full)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--certs-per-tile", type=int, default=0)
    ap.add_argument("--lds-bytes", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=6_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="host threads of the all-core cpu_baseline leg (0 = the CPUs this process may use at once: the "
                         "cgroup quota if there is one, else the affinity mask; 1 = only the one-core leg)")
    ap.add_argument("--cpu-sample-mt", type=int, default=0,
                    help="entries of the all-core leg (default ≈400 k per thread, at most 16 M)")
    ap.add_argument("--meta", action="store_true",
                    help="also run the IssuerMetadata memo kernel (k_meta_new, SURVEY §8(f) N3) over the NEW list of "
                         "every step (engine created with collect_meta) and report its time")
    ap.add_argument("--fingerprint", action="store_true",
                    help="also time the auxiliary whole-certificate SHA-256 kernel (k_fingerprint; VALU-bound, not on "
                         "the reference's path) over the batch")
    ap.add_argument("--stream", type=int, default=0, metavar="TOTAL",
                    help="BASELINE config 5 on one GPU: stream TOTAL entries with 10%% duplicates through one engine in "
                         "waves of --entries (default 50M), the known-certificate table persisting across waves; "
                         "checks n_new / n_dup of every wave against the generator's duplicate structure")
    ap.add_argument("--mixed", action="store_true",
                    help="the mixed synthetic corpus (half EC P-256 keys, 40%% OV-like subjects of 120-260 bytes, longer "
                         "issuer names, one GeneralizedTime in four) instead of the SURVEY §8(d) corpus: how the map "
                         "behaves when the lanes of a wave do not walk identical layouts; not the default workload")
    ap.add_argument("--pem", action="store_true",
                    help="also time the PEM write-back kernels (k_pem_len + scan + k_pem_encode, SURVEY §8(f) N1) over "
                         "the first 16M entries of the NEW list")
    ap.add_argument("--global-dedup", nargs="?", const="owner", default=None, choices=["owner", "bloom"],
                    help="BASELINE config 5's cross-GPU form instead of the shard-local reduce.  owner (default): the "
                         "owner-computes key exchange (distributed.run_global_dedup: export → all-to-all over RCCL → "
                         "owner insert → flags back → apply); at N=1 the exchange is local and this measures the three "
                         "exchange-mode kernels.  bloom: the north_star's all-gather of per-GPU Bloom filters as an exact "
                         "pre-filter (distributed.run_bloom_dedup: local insert → filter all-gather → probe → exact "
                         "lookup at the peers whose filter matched → apply); at N=1 this measures the add/probe/apply "
                         "kernels on top of the ordinary reduce")
    ap.add_argument("--raw", action="store_true",
                    help="feed raw get-entries blobs (leaf_input ‖ extra_data, ≈3.06 KB per entry): adds the "
                         "LogEntryFromLeaf decode and the Chain[0] → issuer match in front of the map (SURVEY §8(f) N2); "
                         "not the default workload")
    ap.add_argument("--trusted-chain", action="store_true",
                    help="with --raw: CTMR_CHAIN0_TRUSTED_LOG — a registered Chain[0] certificate is compared bytewise on its "
                         "first sighting per call and identified by length + first/last 16 bytes afterwards "
                         "(include/ctmr.h); the default compares every byte of every entry's Chain[0]")
    ap.add_argument("--traffic", default="auto", choices=["auto", "off"],
                    help="auto (default, N=1 only): after the timed steps re-execute this script on --traffic-entries "
                         "entries of the same corpus under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate "
                         "passes) and report the map kernel's measured HBM traffic; off: roofline.traffic = null")
    ap.add_argument("--traffic-entries", type=int, default=10_000_000)
    ap.add_argument("--traffic-file", default=None,
                    help="use this earlier measurement (the JSON this script writes to gpurun_out/traffic_map.json) instead "
                         "of measuring; refused unless it was taken on the same build of libctmr.so")
    ap.add_argument("--dup-permille", type=int, default=20,
                    help="entries that repeat an earlier entry's (issuer, serial, notAfter) — anywhere earlier in the "
                         "batch — so that the DEFER / duplicate paths of the insert run at headline scale (default 2 %%)")
    ap.add_argument("--sample-slices", type=int, default=60,
                    help="the oracle-checked sample (= the one-core cpu_baseline leg) is this many equally spaced slices "
                         "of the batch …")
    ap.add_argument("--sample-per-slice", type=int, default=0,
                    help="… of this many entries each (default: --cpu-sample / --sample-slices), plus every entry "
                         "outside the slices whose key a sampled duplicate repeats")
    args = ap.parse_args()

    import numpy as np
    import torch
    import ct_mapreduce_amd as ctmr
    from ct_mapreduce_amd import synth, _native as N

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local %= max(torch.cuda.device_count(), 1)      # (CTMR_DIST_BACKEND=gloo lets two test ranks share one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.gpus > 1 or world > 1:
        # CONTROL path only: torch.distributed (gloo over 127.0.0.1, the launcher's rendezvous) carries the 128-byte
        # group id from rank 0 to the others.  Everything on the data path — shard maps, key exchange, Bloom
        # all-gather, the all-reduce of the per-issuer counts, the barriers and the max-over-ranks of the step time —
        # goes through the library's own RCCL group (ctmr_group_*, csrc/engine/group.inc).
        import torch.distributed as dist
        dist.init_process_group(os.environ.get("CTMR_DIST_BACKEND", "gloo"))
        world, rank = dist.get_world_size(), dist.get_rank()
    else:
        dist = None

    filt = b"Synth Issuer 0,Synth Issuer 1"      # BASELINE config 3: passes issuers 000-199
    # the global-dedup modes run BASELINE config 5's corpus: 10 % of the entries repeat an earlier entry's key —
    # anywhere earlier in the stream, i.e. usually in another rank's shard
    dup_permille = 100 if args.global_dedup else args.dup_permille
    cfg = synth.config(seed=20260921 + 4, n_issuers=args.issuers, zipf=1, dup_permille=dup_permille,
                       ca_permille=10, expired_permille=10, profile=1 if args.mixed else 0)
    now = synth.BASE_TIME
    issuers = synth.issuers(cfg)

    raw_view = {}

    def setup_raw(E):
        eng = ctmr.Engine(device=local, table_slots=pow2_at_least(int(E * 2)), pair_slots=1 << 22,
                          map_variant=args.variant, profile=True, collect_meta=args.meta)
        eng.set_filter(filt, False, now)              # no add_issuers: Chain[0] certificates register themselves
        if world > 1:
            # … in shard order, so issuer index k would name different issuers on different ranks and the count
            # all-reduce would add apples to oranges: with several ranks, register the same list up front
            eng.add_issuers(issuers)
        first = rank * E
        d_bounds = torch.empty(2 * E + 1, dtype=torch.int64, device=dev)
        total = eng.synth_entries_device(cfg, first, E, d_bounds.data_ptr(), 0, 0)
        d_blob = torch.empty(total + N.PAYLOAD_PAD + 16, dtype=torch.uint8, device=dev)
        eng.synth_entries_device(cfg, first, E, d_bounds.data_ptr(), d_blob.data_ptr(), d_blob.numel())
        d_rec = torch.empty(E * 32, dtype=torch.uint8, device=dev)
        d_new = torch.empty(E, dtype=torch.int64, device=dev)
        d_ts = torch.empty(E, dtype=torch.int64, device=dev)
        # caller-owned entry view: the decode fills it, map / meta / PEM read certificates through it
        raw_view["start"] = torch.empty(E, dtype=torch.int64, device=dev)
        raw_view["end"] = torch.empty(E, dtype=torch.int64, device=dev)
        raw_view["iss"] = torch.empty(E, dtype=torch.int32, device=dev)
        raw_view["et"] = torch.empty(E, dtype=torch.uint8, device=dev)
        raw_view["view"] = N.EntryView(cert_start=raw_view["start"].data_ptr(), cert_end=raw_view["end"].data_ptr(),
                                       issuer_idx=raw_view["iss"].data_ptr(), entry_type=raw_view["et"].data_ptr(),
                                       timestamp=d_ts.data_ptr(), chain0_start=None, chain0_len=None)
        raw_view["blob_bytes"] = total
        torch.cuda.synchronize()
        return eng, d_bounds, d_blob, d_ts, None, d_rec, d_new

    def setup(E):
        if args.raw:
            return setup_raw(E)
        eng = ctmr.Engine(device=local, table_slots=(1 << args.table_slots_log2) if args.table_slots_log2 else pow2_at_least(int(E * 2)),
                          pair_slots=1 << 22, map_variant=args.variant, certs_per_tile=args.certs_per_tile,
                          lds_tile_bytes=args.lds_bytes, profile=True, collect_meta=args.meta)
        eng.add_issuers(issuers)
        eng.set_filter(filt, False, now)
        # ---- synthetic shard [rank·E, (rank+1)·E), generated directly in HBM
        first = rank * E
        d_off = torch.empty(E + 1, dtype=torch.int64, device=dev)
        total = eng.synth_device(cfg, first, E, d_off.data_ptr(), 0, 0, 0, 0)
        d_pay = torch.empty(total + N.PAYLOAD_PAD + 16, dtype=torch.uint8, device=dev)
        d_iss = torch.empty(E, dtype=torch.int32, device=dev)
        d_et = torch.empty(E, dtype=torch.uint8, device=dev)
        eng.synth_device(cfg, first, E, d_off.data_ptr(), d_pay.data_ptr(), d_pay.numel(),
                         d_iss.data_ptr(), d_et.data_ptr())
        d_rec = torch.empty(E * 32, dtype=torch.uint8, device=dev)
        d_new = torch.empty(E, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        return eng, d_off, d_pay, d_iss, d_et, d_rec, d_new

    if args.stream:
        return run_stream(args, ctmr, synth, N, torch, np, dev, local, rank, world, cfg, filt, now, issuers)

    E = args.entries
    if args.raw and "CTMR_BENCH_ENTRIES" not in os.environ and E == 100_000_000:
        E = 40_000_000          # ≈122 GB of raw entries + the table
    t_gen = time.perf_counter()
    while True:
        try:
            eng, d_off, d_pay, d_iss, d_et, d_rec, d_new = setup(E)
            break
        except (ctmr.CtmrError, RuntimeError) as ex:   # does not fit in this GPU's HBM: halve
            if E <= 1_000_000:
                raise
            sys.stderr.write(f"bench: {E} entries do not fit ({ex}); trying {E // 2}\n")
            eng = d_off = d_pay = d_iss = d_et = d_rec = d_new = None
            torch.cuda.empty_cache()
            E //= 2
    t_gen = time.perf_counter() - t_gen
    if args.raw and args.trusted_chain:
        eng.set_chain0_match(N.CHAIN0_TRUSTED_LOG)
    from ct_mapreduce_amd.distributed import Group, shard as make_shard
    group = None
    torch_fallback = None
    if dist is not None:
        try:
            group = Group.rccl(eng, share_group_id(dist, rank, Group.unique_id), rank, world)
        except Exception as ex:          # noqa: BLE001 — librccl missing / communicator refused: say so, keep measuring
            # the count all-reduce, barriers and the max of the step time then go over torch.distributed (the control
            # path's process group); the line says so in config.parallelism.  The global-dedup modes need the group.
            if args.global_dedup:
                raise
            torch_fallback = f"{type(ex).__name__}: {ex}"
            sys.stderr.write(f"bench: RCCL group not available ({torch_fallback}); collectives over torch.distributed\n")
    elif args.global_dedup:
        group = Group.local([eng])          # N = 1: the exchange is local, this measures each mode's kernels
    if args.global_dedup == "bloom":
        group.bloom_config(pow2_at_least(16 * E))       # ≈16 filter bits per key held
    global_counts = [None]
    dstats = []
    meta_ms, meta_items = [], []
    d_items = torch.empty(32 * (1 << 22), dtype=torch.uint8, device=dev) if args.meta else None

    def step():
        eng.reset_known()
        if args.raw:   # d_off = bounds, d_pay = blob, d_iss = timestamps
            ds = eng.decode_entries_device(d_pay.data_ptr(), d_off.data_ptr(), E, raw_view["view"])
            st = eng.map_view_device(d_pay.data_ptr(), raw_view["blob_bytes"], raw_view["view"], E, d_rec.data_ptr(),
                                     d_new.data_ptr())
            dstats.append(ds)
        elif group is not None:
            # one native call: this rank's shard map + (owner | Bloom) exchange over RCCL (copies when N = 1)
            st = group.map_batch(args.global_dedup or "local",
                                 [make_shard(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), E,
                                             d_rec.data_ptr(), d_new.data_ptr(), order_base=rank * E)])[0]
        else:
            st = eng.map_batch_device(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(),
                                      E, d_rec.data_ptr(), d_new.data_ptr())
        if args.meta:
            t_m = time.perf_counter()
            offs_p = raw_view["start"].data_ptr() if args.raw else d_off.data_ptr()
            ends_p = raw_view["end"].data_ptr() if args.raw else 0
            meta_items.append(eng.meta_new_device(d_pay.data_ptr(), offs_p, ends_p, d_rec.data_ptr(),
                                                  d_new.data_ptr(), int(st.n_new), d_items.data_ptr(), 1 << 22))
            meta_ms.append((time.perf_counter() - t_m) * 1e3)
        if group is not None and world > 1:
            # per-issuer unique counts merged over xGMI: ncclAllReduce inside ctmr_group_issuer_counts (2 KiB)
            global_counts[0] = group.issuer_counts(len(issuers))
        elif torch_fallback and world > 1:
            c = torch.from_numpy(eng.issuer_counts()[:len(issuers)].astype(np.int64))
            dist.all_reduce(c)
            global_counts[0] = c.numpy().astype(np.uint64)
        return st

    def barrier():
        if group is not None and world > 1:
            group.barrier()
        elif torch_fallback and world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ms_map = []
    stats = None
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = step()
        ms_map.append(stats.ms_map)
    barrier()
    dt = time.perf_counter() - t0
    if group is not None and world > 1:
        dt = float(group.all_reduce_u64([int(dt * 1e9)], op_max=True)[0]) * 1e-9      # the slowest rank's time
    elif torch_fallback and world > 1:
        dt = max_over_ranks(dist, dt, torch.device("cpu"))

    # HBM traffic of the map kernel per launch: PMC counters can only be collected under rocprofv3, in their own
    # passes — this script re-executes itself under the profiler on a smaller batch of the same corpus (bytes per
    # certificate do not depend on the batch size: every launch streams ≫ the 256 MB of on-die cache) and scales.
    traffic = traffic_info = traffic_err = None
    kname = MAP_KERNELS[args.variant or DEFAULT_VARIANT].split("<")[0]
    plain = not (args.raw or args.global_dedup or args.meta)
    if rank == 0 and world == 1 and plain and not os.environ.get("CTMR_BENCH_CHILD"):
        if args.traffic_file:
            try:
                t = json.load(open(args.traffic_file))
                if t.get("lib_sha256_16") == lib_hash() and "traffic_bytes_per_cert" in t:
                    traffic_info = dict(t, source=os.path.relpath(args.traffic_file, ROOT))
                else:
                    traffic_err = "traffic file refused: taken on another build of libctmr.so"
            except (ValueError, OSError) as ex:
                traffic_err = f"traffic file unreadable: {ex}"
        elif args.traffic == "auto":
            t_tr = time.perf_counter()
            traffic_info, traffic_err = measure_traffic(args, min(E, args.traffic_entries), kname)
            if traffic_info:
                traffic_info["source"] = "measured by this run (rocprofv3 --pmc, two passes)"
                traffic_info["seconds"] = round(time.perf_counter() - t_tr, 1)
                try:
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    json.dump(traffic_info, open(os.path.join(ROOT, "gpurun_out", "traffic_map.json"), "w"))
                except OSError:
                    pass
        if traffic_info:
            traffic = traffic_info["traffic_bytes_per_cert"] * E

    n_total = E * world
    if args.global_dedup:   # no per-kernel events here: the map time is not separable
        ms_map = [dt / args.steps * 1e3]
    value = n_total * args.steps / dt
    alg_bytes = stats.payload_bytes + ALG_BYTES_FIXED * E     # raw mode: payload_bytes is the whole blob — see "raw"
    if (args.variant or DEFAULT_VARIANT) in FUSED:
        alg_bytes += ALG_BYTES_PROBE * int(stats.by_status[0])
    avg_ms = sum(ms_map) / len(ms_map)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    out = {
        "metric": "certificates/sec whole-node + achieved HBM GB/s, 100M-entry synthetic CT batch",
        "value": value, "unit": "certificates/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{E} synthetic ~1.5 KB DER CT entries per GPU, {args.issuers} issuers (Zipf), "
                               f"{dup_permille / 10:g} % duplicates of earlier entries, "
                               "issuerCN prefix filter + known-certificate dedup + per-issuer unique counts "
                               "(BASELINE configs[2]/[3] shape)",
                   "entries_per_gpu": E, "mean_der_bytes": stats.payload_bytes / E,
                   "parallelism": f"log-index shards x{world}" + ("" if world == 1 else
                                  " + per-issuer count all-reduce over RCCL inside the library (ctmr_group_issuer_counts)" if group is not None
                                  else f" + count all-reduce over torch.distributed (RCCL group unavailable: {torch_fallback})"),
                   "map_variant": args.variant or DEFAULT_VARIANT,
                   "gen_seconds": round(t_gen, 2)},
        # roofline of the dominant kernel.  `frac` is PHYSICAL when the traffic was measured: HBM bytes the kernel moved
        # (PMC counters) ÷ its average launch time ÷ peak.  The walk skips key, SAN body and signature by length, so the
        # SURVEY §8(d) algorithmic figure (every certificate byte "read once") counts bytes that never move:
        # `frac_algorithmic` is kept beside it, never instead of it.
        "roofline": {"bound": "hbm", "kernel": MAP_KERNELS[args.variant or DEFAULT_VARIANT],
                     "achieved": (traffic if traffic else alg_bytes) / (avg_ms * 1e-3) / 1e9,
                     "achieved_basis": "measured HBM traffic (FETCH_SIZE x2 + WRITE_SIZE) / avg launch time" if traffic
                                       else "ALGORITHMIC bytes / avg launch time (no PMC measurement in this run"
                                            + (": " + traffic_err if traffic_err else "") + ")",
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": (traffic if traffic else alg_bytes) / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "traffic": traffic, "traffic_measurement": traffic_info,
                     "frac_of_streaming_ceiling": (traffic / (avg_ms * 1e-3) / 1e9 / 6290.0) if traffic else None,
                     "achieved_algorithmic": achieved, "frac_algorithmic": achieved / HBM_PEAK_GBPS,
                     "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_ms,
                     "alg_bytes_formula": "sum(L_i) + 45*E" + (" + 64*PASS (table probe)" if (args.variant or DEFAULT_VARIANT) in FUSED else "")},
        "kernel_ms": {"map": stats.ms_map, "insert": stats.ms_insert, "resolve": stats.ms_resolve,
                      "compact": stats.ms_compact, "total": stats.ms_total},
        "result": {"n_new": int(stats.n_new), "n_dup": int(stats.n_dup), "by_status": [int(x) for x in stats.by_status],
                   "issuer_counts_all_ranks_sum": int(global_counts[0].sum()) if global_counts[0] is not None else None},
    }
    if args.pem:
        m = min(int(stats.n_new), 16_000_000)
        d_po = torch.empty(m + 1, dtype=torch.int64, device=dev)

        def pem_call(d_pem_ptr, cap):
            if args.raw:
                return eng.pem_encode_view_device(d_pay.data_ptr(), raw_view["view"], d_new.data_ptr(), m, d_pem_ptr, cap,
                                                  d_po.data_ptr())
            return eng.pem_encode_device(d_pay.data_ptr(), d_off.data_ptr(), d_new.data_ptr(), m, d_pem_ptr, cap,
                                         d_po.data_ptr())
        total = pem_call(0, 0)
        d_pem = torch.empty(total + 64, dtype=torch.uint8, device=dev)
        t_p = []
        for _ in range(3):
            t0p = time.perf_counter()
            pem_call(d_pem.data_ptr(), total + 64)
            t_p.append(time.perf_counter() - t0p)
        import base64
        po = d_po[:3].cpu().numpy()
        first = int(d_new[0].item())
        starts_t = raw_view["start"] if args.raw else d_off[:-1]
        ends_t = raw_view["end"] if args.raw else d_off[1:]
        der = d_pay[int(starts_t[first].item()):int(ends_t[first].item())].cpu().numpy().tobytes()
        b64 = base64.b64encode(der)                  # pem.EncodeToMemory: 64-column base64 between the two marker lines
        want = (b"-----BEGIN CERTIFICATE-----\n" + b"".join(b64[k:k + 64] + b"\n" for k in range(0, len(b64), 64)) +
                b"-----END CERTIFICATE-----\n")
        ok_pem = d_pem[int(po[0]):int(po[1])].cpu().numpy().tobytes() == want
        in_bytes = int((ends_t[d_new[:m]] - starts_t[d_new[:m]]).sum().item())
        out["pem"] = {"certificates": m, "pem_bytes": int(total), "der_bytes": in_bytes, "ms_wall": min(t_p) * 1e3,
                      "certs_per_s": m / min(t_p), "GBps_read_plus_written": (in_bytes + total) / min(t_p) / 1e9,
                      "first_block_matches_stdlib_base64": bool(ok_pem)}
    if args.global_dedup:
        out["config"]["workload"] = out["config"]["workload"].replace(
            "(BASELINE configs[2]/[3] shape)", "— mostly in other ranks' shards "
            "(BASELINE configs[4] corpus, one round)")
        # exactness of the GLOBAL dedup against the generator's structure: entry i repeats an earlier entry's key iff
        # synth_is_dup(i), wherever that earlier entry lives — so this rank's NEW entries are its PASS ∧ ¬dup ones
        status = d_rec.view(-1, 32)[:E, 0].cpu().numpy()
        is_new = (d_rec.view(-1, 32)[:E, 1].cpu().numpy() & 2) != 0
        dup = synth_is_dup(cfg.seed, rank * E, E, dup_permille, np)
        bad = int((is_new != ((status == 0) & ~dup)).sum()) + int(int(stats.n_new) != int(is_new.sum()))
        bad, n_new_all = (int(v) for v in group.all_reduce_u64([bad, int(stats.n_new)]))
        gi = group.info()
        out["result"]["global_dedup"] = {"mode": args.global_dedup, "n_new_all_ranks": n_new_all,
                                         "entries_disagreeing_with_generator": bad,
                                         "transport": "rccl" if gi.transport else "local (one rank: copies)",
                                         "key_records_sent_by_rank0": int(gi.keys_sent),
                                         "filter_bytes_received_by_rank0": int(gi.filter_bytes_received)}
        out["roofline"]["note"] = "avg_launch_ms is the wall time of the whole step, not one kernel"
        out["roofline"]["achieved"] = out["roofline"]["frac"] = None      # no single kernel to price: see kernel_ms of the default mode
        if args.global_dedup == "bloom":
            out["config"]["parallelism"] = f"log-index shards x{world} + Bloom-filter all-gather pre-filter + exact lookup (global dedup)"
            out["roofline"]["kernel"] = "whole Bloom-mode step (map + insert + filter add + all-gather + probe + lookup + apply)"
        else:
            out["config"]["parallelism"] = f"log-index shards x{world} + owner-computes key exchange (global dedup)"
            out["roofline"]["kernel"] = "whole exchange-mode step (export + all-to-all + owner insert + apply)"
    if args.fingerprint and not args.raw:
        d_dg = torch.empty(E * 32, dtype=torch.uint8, device=dev)
        fp_ms = [eng.fingerprint_device(d_pay.data_ptr(), d_off.data_ptr(), 0, E, d_dg.data_ptr()) for _ in range(3)]
        import hashlib
        okfp = True
        offs_h = d_off[:1001].cpu().numpy()
        pay_h = d_pay[: int(offs_h[-1])].cpu().numpy().tobytes()
        dg_h = d_dg[: 1000 * 32].cpu().numpy().tobytes()
        for k in range(1000):
            okfp = okfp and hashlib.sha256(pay_h[int(offs_h[k]):int(offs_h[k + 1])]).digest() == dg_h[32 * k:32 * k + 32]
        ms_fp = min(fp_ms)
        out["fingerprint"] = {"kernel": "k_fingerprint", "ms": ms_fp, "certs_per_s": E / (ms_fp * 1e-3),
                              "hashed_GBps": stats.payload_bytes / (ms_fp * 1e-3) / 1e9,
                              "bound": "valu", "blocks_per_s": (stats.payload_bytes / 64 + 1.5 * E) / (ms_fp * 1e-3),
                              "matches_hashlib_on_first_1000": bool(okfp),
                              "note": "auxiliary op, not on the reference's path (SURVEY D2); VALU roofline in DESIGN.md §5"}
    if args.mixed:
        out["config"]["workload"] = "MIXED corpus (EC/RSA keys, OV-like subjects, GeneralizedTime): " + out["config"]["workload"]
    if args.meta and meta_ms:
        out["kernel_ms"]["meta_new_cold_wall"] = meta_ms[0]
        out["kernel_ms"]["meta_new_warm_wall"] = sum(meta_ms[1:]) / max(len(meta_ms) - 1, 1)
        out["meta"] = {"first_sightings_cold": meta_items[0], "first_sightings_warm": meta_items[-1],
                       "new_certificates": int(stats.n_new),
                       "note": "cold = first call (empty memo: every (issuer, expDate), DN and CRL DP is a first "
                               "sighting); warm = later steps (the known-certificate table is cleared every step, the "
                               "memo is not: every certificate is new again, nothing is a first sighting)"}
    if args.raw:
        ds = dstats[-1]
        out["config"]["workload"] = (f"{E} RAW get-entries (leaf_input+extra_data, {stats.payload_bytes / E:.0f} B/entry) per GPU: "
                                     "LogEntryFromLeaf decode + Chain[0] issuer match + " + out["config"]["workload"])
        out["raw"] = {"blob_bytes": int(ds.blob_bytes), "ms_decode": ds.ms_decode, "ms_match": ds.ms_match,
                      "n_x509": int(ds.n_x509), "n_precert": int(ds.n_precert),
                      "issuers_registered_by_the_engine": eng.issuer_count(),
                      "chain0_match": "trusted-log (bytewise on first sighting per call, then length + first/last 16 B)"
                                      if args.trusted_chain else "exact (every byte of every Chain[0])",
                      "note": "roofline.achieved counts the WHOLE blob as the map kernel's algorithmic bytes although it "
                              "skips extra_data and the precert TBS; decode and the first match round are one kernel (ms_decode = 0, ms_match = both)"}
        out["kernel_ms"]["decode"] = ds.ms_decode
        out["kernel_ms"]["match"] = ds.ms_match
    if rank == 0:
        if world == 1 and not args.no_cpu and not args.raw:
            # ---- the oracle-checked sample = the one-core cpu_baseline leg: equally spaced slices over the WHOLE batch,
            # copied back from HBM, plus — generated on the host, byte-identical to the device generator
            # (tests/test_gpu_parity.py) — every entry outside the slices whose key a sampled duplicate repeats, all in
            # log order.  The generator repeats keys of NON-duplicate entries only (csrc/synth.h synth_src), so the
            # oracle's WasUnknown over this closed set is the whole batch's answer for every entry in it.
            per = args.sample_per_slice or max(1, min(args.cpu_sample, E) // args.sample_slices)
            ranges = strided_sample(E, args.sample_slices, per)
            in_sample = np.concatenate([np.arange(lo, hi, dtype=np.uint64) for lo, hi in ranges])
            src, isdup = synth_src(cfg.seed, in_sample, dup_permille, np)
            extra = np.setdiff1d(src[isdup], in_sample)
            extra_certs = [synth.leaf(cfg, int(i)) for i in extra]
            arrays = gather_sample(d_off, d_pay, d_iss, d_et, ranges, extra, extra_certs, N.PAYLOAD_PAD, np)
            sample = len(arrays[4])
            base, (ost, ounk) = cpu_baseline(arrays[:3], issuers, filt, now, sample, arrays[3])
            base["sample"] = (f"{args.sample_slices} equally spaced slices of {per} entries of the same synthetic batch, copied "
                              f"back from HBM, + the {len(extra)} entries outside them whose keys sampled duplicates repeat; "
                              + base["sample"])
            out["cpu_baseline"] = base
            gidx = torch.from_numpy(arrays[4].astype(np.int64)).to(dev)
            rec = d_rec.view(-1, 32)[gidx].cpu().numpy().reshape(-1).view(ctmr.engine.RECORD_DTYPE)
            gnew = (rec["flags"] & 2) != 0
            out["parity_vs_oracle_on_sample"] = bool((rec["status"] == ost).all() and (gnew == (ounk != 0)).all())
            out["parity_sample"] = {"entries": int(sample), "slices": args.sample_slices, "entries_per_slice": per,
                                    "sources_outside_the_slices": int(len(extra)),
                                    "pass": int((ost == 0).sum()), "was_unknown": int((ounk != 0).sum()),
                                    "known_duplicates": int(((ost == 0) & (ounk == 0)).sum()),
                                    "status_mismatches": int((rec["status"] != ost).sum()),
                                    "was_unknown_mismatches": int((gnew != (ounk != 0)).sum())}
            # what the walk must read of these certificates, against the measured traffic
            k = min(20000, per)
            offs_k = arrays[1][:k + 1]
            certs_k = [arrays[0][int(offs_k[i]):int(offs_k[i + 1])].tobytes() for i in range(k)]
            starts_k = d_off[ranges[0][0]:ranges[0][0] + k].cpu().numpy()
            nb, nl = needed_bytes_per_cert(certs_k, starts_k, filt)
            n_pass = int(stats.by_status[0])
            fixed = ALG_BYTES_FIXED * E + 4 * E + ALG_BYTES_PROBE * n_pass      # arrays + record + ent[] word; slot read + write
            out["roofline"]["needed_bytes"] = nb * E + fixed
            out["roofline"]["needed_line_bytes"] = nl * E + fixed + 64 * n_pass   # … when every touched 128-B line moves whole (slot lines too)
            out["roofline"]["needed_note"] = (f"per certificate the walk's reads cover {nb:.0f} bytes lying in {nl / 128:.2f} lines of 128 B "
                                              f"(product walk on the host with a marking reader, first {k} sampled certificates); "
                                              "+ 45 B arrays/record + 4 B reduce state per entry + the 64-B slot (a 128-B line) per PASS entry")
            if traffic:
                out["roofline"]["over_fetch_vs_needed_bytes"] = traffic / out["roofline"]["needed_bytes"]
                out["roofline"]["over_fetch_vs_needed_lines"] = traffic / out["roofline"]["needed_line_bytes"]
            quota_threads, quota = cpu_quota()
            threads = args.cpu_threads or quota_threads
            if threads > 1:
                # … and on every host CPU this process may use: a contiguous sample of the same batch (≈400 k entries per thread)
                del arrays
                sample_mt = min(E, args.cpu_sample_mt or min(400_000 * threads, 16_000_000))
                offs_mt = d_off[: sample_mt + 1].cpu().numpy().astype(np.uint64)
                nb_mt = int(offs_mt[-1])
                pay_mt = torch.zeros(nb_mt + N.PAYLOAD_PAD, dtype=torch.uint8)
                pay_mt[:nb_mt].copy_(d_pay[:nb_mt])
                arrays_mt = (pay_mt.numpy(), offs_mt, d_iss[:sample_mt].cpu().numpy().astype(np.uint32))
                best = None
                for _ in range(3):
                    v, dt_mt, npass = cpu_baseline_threads(arrays_mt, issuers, filt, now, sample_mt, threads)
                    if best is None or v > best[0]:
                        best = (v, dt_mt, npass)
                ok_mt = best[2] == int((d_rec.view(-1, 32)[:sample_mt, 0] == 0).sum().item())
                out["cpu_baseline"] = {
                    "value": best[0], "unit": "certificates/sec", "cores": threads, "kind": "port",
                    "sample": f"first {sample_mt} entries of the same synthetic batch in {threads} contiguous slices, one "
                              f"oracle/ctmr_oracle.c engine per thread (per-thread in-process sets stand in for the "
                              f"shared Redis; not the Go binary), best of 3, {best[1]:.2f} s",
                    "host_cores_available": os.cpu_count(), "cgroup_cpu_quota": quota,
                    "pass_count_matches_gpu": bool(ok_mt),
                    "one_core": {"value": base["value"], "sample": base["sample"]}}
        print(json.dumps(out))
    if group is not None:
        group.close()
    if dist is not None:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
