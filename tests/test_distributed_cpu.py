"""world_size-2 gloo test of the multi-GPU host logic (shard ranges + count all-reduce).
The per-rank map is the ORACLE here (tests may use it as the checker/stand-in); on GPUs the same
driver is fed Engine.map_batch (bench.py --gpus N)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from ct_mapreduce_amd.distributed import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_the_stream():
    for n in (0, 1, 7, 100, 1001):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def _worker(rank, world, port, n_total, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ct_mapreduce_amd import synth
    from ct_mapreduce_amd.distributed import run_sharded
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = synth.config(seed=77, n_issuers=8, dup_permille=0, ca_permille=20, expired_permille=20)
    issuers = synth.issuers(cfg)
    ids = [orc.issuer_id(d[orc.parse_cert(d).spki_off:][:orc.parse_cert(d).spki_len]) for d in issuers]
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    blob = np.frombuffer(b"".join(issuers), np.uint8)
    eng = orc.Engine(b"Synth Issuer 00", False, synth.BASE_TIME)

    def map_fn(b):
        st, unk, eh = eng.batch(b.payload, b.offsets, b.issuer_idx, blob, io)
        return int(unk.sum())

    res = run_sharded(n_total, len(issuers), lambda lo, hi: synth.host_batch(cfg, lo, hi - lo), map_fn,
                      lambda: np.array([eng.issuer_count(i) for i in ids], dtype=np.uint64))
    if rank == 0:
        np.save(out, np.concatenate([[res.n_new_global], res.global_counts]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_counts_match_single_process(tmp_path):
    from ct_mapreduce_amd import synth
    from oracle import oracle as orc
    n_total = 3001
    out = str(tmp_path / "counts.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, n_total, out), nprocs=2, join=True)
    got = np.load(out)
    # single process over the whole stream
    cfg = synth.config(seed=77, n_issuers=8, dup_permille=0, ca_permille=20, expired_permille=20)
    issuers = synth.issuers(cfg)
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    eng = orc.Engine(b"Synth Issuer 00", False, synth.BASE_TIME)
    b = synth.host_batch(cfg, 0, n_total)
    st, unk, eh = eng.batch(b.payload, b.offsets, b.issuer_idx, np.frombuffer(b"".join(issuers), np.uint8), io)
    ids = [orc.issuer_id(d[orc.parse_cert(d).spki_off:][:orc.parse_cert(d).spki_len]) for d in issuers]
    want = [int(unk.sum())] + [eng.issuer_count(i) for i in ids]
    assert list(got) == want
