"""bench.py's own checkers, pinned without a GPU: the numpy restatement of the generator's duplicate structure (what
`--stream` and `--global-dedup` compare the engine against) and the CPU quota probe of the all-core baseline leg."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from ct_mapreduce_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def test_synth_is_dup_describes_the_generators_duplicates():
    cfg = synth.config(seed=20260921 + 5, n_issuers=16, zipf=1, dup_permille=100, ca_permille=10, expired_permille=10)
    issuers = synth.issuers(cfg)
    n = 20000
    b = synth.host_batch(cfg, 0, n)
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    o = orc.Engine(b"", False, synth.BASE_TIME)
    st, unk, eh = o.batch(b.payload, b.offsets, b.issuer_idx, np.frombuffer(b"".join(issuers), np.uint8), io)
    dup = bench.synth_is_dup(cfg.seed, 0, n, 100, np)
    assert 0.05 * n < dup.sum() < 0.15 * n
    passed = st == 0
    assert (unk[passed & ~dup] == 1).all()          # a non-duplicate that passes the filters is new …
    assert (unk[passed & dup] == 0).all()           # … and a duplicate repeats the key of an earlier entry
    # the same predicate for a later window of the stream (what the waves of --stream use)
    dup2 = bench.synth_is_dup(cfg.seed, 5000, 1000, 100, np)
    assert (dup2 == dup[5000:6000]).all()


def test_cpu_quota_is_sane():
    threads, quota = bench.cpu_quota()
    assert 1 <= threads <= len(os.sched_getaffinity(0))
    assert quota is None or quota > 0


def test_cpu_baseline_threads_counts_every_entry_once():
    cfg = synth.config(seed=9, n_issuers=4, dup_permille=0, ca_permille=100, expired_permille=100)
    issuers = synth.issuers(cfg)
    n = 3001
    b = synth.host_batch(cfg, 0, n)
    pay = np.concatenate([b.payload, np.zeros(64, np.uint8)])
    arrays = (pay, b.offsets.astype(np.uint64), b.issuer_idx.astype(np.uint32))
    one = bench.cpu_baseline_threads(arrays, issuers, b"", synth.BASE_TIME, n, 1)
    four = bench.cpu_baseline_threads(arrays, issuers, b"", synth.BASE_TIME, n, 4)
    assert one[2] == four[2] > 0                    # same PASS count whatever the slicing


def test_strided_sample_closure_gives_the_whole_batch_answer():
    """bench.py's parity leg: the oracle over (sampled slices + the sources of sampled duplicates), in log order, must
    say for every one of those entries what the oracle over the WHOLE batch says — here the whole batch is small enough
    to run, at 100 M entries only the closed sample is."""
    import torch
    cfg = synth.config(seed=20260925, n_issuers=16, zipf=1, dup_permille=150, ca_permille=10, expired_permille=10)
    E = 30000
    b = synth.host_batch(cfg, 0, E)
    issuers = synth.issuers(cfg)
    filt = b"Synth Issuer 00"
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    blob = np.frombuffer(b"".join(issuers), np.uint8)
    whole = orc.Engine(filt, False, synth.BASE_TIME)
    st_w, unk_w, _ = whole.batch(np.concatenate([b.payload, np.zeros(32, np.uint8)]), b.offsets, b.issuer_idx, blob, io,
                                 entry_type=b.entry_type)
    ranges = bench.strided_sample(E, 12, 500)
    assert len(ranges) == 12 and ranges[1][0] == E // 12 and all(hi - lo == 500 for lo, hi in ranges)
    in_sample = np.concatenate([np.arange(lo, hi, dtype=np.uint64) for lo, hi in ranges])
    src, isdup = bench.synth_src(cfg.seed, in_sample, 150, np)
    assert (isdup == bench.synth_is_dup_at(cfg.seed, in_sample, 150, np)).all() and isdup.sum() > 500
    extra = np.setdiff1d(src[isdup], in_sample)
    assert len(extra) > 300                                      # most sources lie outside the slices
    extra_certs = [synth.leaf(cfg, int(i)) for i in extra]
    d_off = torch.from_numpy(b.offsets.astype(np.int64))
    arrays = bench.gather_sample(d_off, torch.from_numpy(b.payload), torch.from_numpy(b.issuer_idx.astype(np.int32)),
                                 torch.from_numpy(b.entry_type), ranges, extra, extra_certs, 32, np)
    gidx = arrays[4].astype(np.int64)
    assert (np.diff(gidx) > 0).all() and len(gidx) == len(in_sample) + len(extra)
    for k in (0, 17, len(gidx) - 1):                             # the gathered bytes are the batch's bytes
        assert arrays[0][int(arrays[1][k]):int(arrays[1][k + 1])].tobytes() == b.cert(int(gidx[k]))
    part = orc.Engine(filt, False, synth.BASE_TIME)
    st_s, unk_s, _ = part.batch(arrays[0], arrays[1], arrays[2], blob, io, entry_type=arrays[3])
    assert (st_s == st_w[gidx]).all()
    assert (unk_s == unk_w[gidx]).all()
    assert ((st_s == 0) & (unk_s == 0)).sum() > 300              # known duplicates are in the sample


def test_the_torch_form_of_the_duplicate_predicate_is_the_numpy_one():
    """bench.py checks every entry's WasUnknown against the generator at every N; at 100 M entries the predicate is
    evaluated on the device with int64 tensors (no unsigned 64-bit type in torch): it must be the numpy restatement."""
    import torch
    for seed, first, permille in ((20260925, 0, 20), (20260925, 123_456_789, 100), (7, 99_999_000, 150)):
        n = 200_000
        want = bench.synth_is_dup(seed, first, n, permille, np)
        got = bench.synth_is_dup_torch(seed, first, n, permille, torch, torch.device("cpu")).numpy()
        assert (got == want).all() and 0 < want.sum() < n
