"""CPU side of the multi-GPU path.  The data path (shard maps, key exchange, Bloom all-gather, count all-reduce) is
native and needs GPUs (tests/test_gpu_exchange.py, test_gpu_bloom.py drive it through a local group); what is left
for the host — and tested here, with world_size-2 `gloo` processes — is the shard arithmetic, carrying the 128-byte
group id from one rank to all (bench.py's control path) and the property the shard-local mode rests on: per-issuer
counts of log-index shards add up to the single-process counts when no key spans two shards (BASELINE config 4).
The per-rank map is the ORACLE here (a stand-in for the engine, as the checker may be)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from ct_mapreduce_amd.distributed import shard_range, union_in_agreed_order, decode_synchronised
from ct_mapreduce_amd.engine import CtmrError
from ct_mapreduce_amd import _native as N

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
    import torch
    import torch.distributed as dist
    import bench
    from ct_mapreduce_amd import synth
    from oracle import oracle as orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    # bench.py's control path under a launcher: rank 0 makes the group id, one broadcast over the launcher's
    # rendezvous, every rank ends up with the same 128 bytes — and the process group is gone again afterwards
    gid = bench.group_id_from_launcher(rank, lambda: bytes([7]) * 100 + os.urandom(28))
    assert len(gid) == 128 and gid[:100] == bytes([7]) * 100 and not dist.is_initialized()
    os.environ["MASTER_PORT"] = str(port + 1)
    dist.init_process_group("gloo", rank=rank, world_size=world)      # (the rest of this test's own plumbing)
    box = [None] * world
    dist.all_gather_object(box, gid)
    assert all(b == box[0] for b in box)
    # the shard-local mode: rank r maps [lo, hi) with its own known-certificate sets; counts are summed
    cfg = synth.config(seed=77, n_issuers=8, dup_permille=0, ca_permille=20, expired_permille=20)
    issuers = synth.issuers(cfg)
    ids = [orc.issuer_id(d[orc.parse_cert(d).spki_off:][:orc.parse_cert(d).spki_len]) for d in issuers]
    io = np.zeros(len(issuers) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in issuers])
    blob = np.frombuffer(b"".join(issuers), np.uint8)
    eng = orc.Engine(b"Synth Issuer 00", False, synth.BASE_TIME)
    lo, hi = shard_range(n_total, rank, world)
    b = synth.host_batch(cfg, lo, hi - lo)
    st, unk, eh = eng.batch(b.payload, b.offsets, b.issuer_idx, blob, io, entry_type=b.entry_type)
    t = torch.tensor([int(unk.sum())] + [eng.issuer_count(i) for i in ids], dtype=torch.int64)
    dist.all_reduce(t)                      # what ctmr_group_issuer_counts does natively (ncclAllReduce)
    if rank == 0:
        np.save(out, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
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
    st, unk, eh = eng.batch(b.payload, b.offsets, b.issuer_idx, np.frombuffer(b"".join(issuers), np.uint8), io,
                            entry_type=b.entry_type)
    ids = [orc.issuer_id(d[orc.parse_cert(d).spki_off:][:orc.parse_cert(d).spki_len]) for d in issuers]
    want = [int(unk.sum())] + [eng.issuer_count(i) for i in ids]
    assert list(got) == want


class _FakeEngine:
    """Issuer registry of an engine with auto-registration off: a decode lists what it has not seen registered."""

    def __init__(self, needs):
        self.needs, self.registered, self.calls = list(needs), [], 0

    def decode(self):
        self.calls += 1
        self._pending = [d for d in self.needs if d not in self.registered]
        if self._pending:
            raise CtmrError(N.E_NOTFOUND, "unregistered Chain[0] certificates")
        return "decoded"

    def pending_issuers(self):
        return list(self._pending)

    def add_issuers(self, ders):
        self.registered += list(ders)


def test_issuer_tables_are_synchronised_in_an_agreed_order():
    assert union_in_agreed_order([[b"b", b"a"], [b"c", b"a"], []]) == [b"a", b"b", b"c"]
    e0, e1 = _FakeEngine([b"iss-2", b"iss-1"]), _FakeEngine([b"iss-3", b"iss-1"])
    res = decode_synchronised([e0, e1], [e0.decode, e1.decode])
    assert res == ["decoded", "decoded"]
    assert e0.registered == e1.registered == [b"iss-1", b"iss-2", b"iss-3"]      # the same certificates, the same order
    # one process per rank: the host supplies the gather
    a, b = _FakeEngine([b"x"]), _FakeEngine([b"y", b"x"])
    pend_b = [b"y", b"x"]
    assert decode_synchronised(a, a.decode, gather=lambda mine: [mine, [d for d in pend_b if d not in a.registered]]) == "decoded"
    assert a.registered == [b"x", b"y"]
    with pytest.raises(ValueError):
        decode_synchronised(a, a.decode)


def test_group_entry_points_reject_bad_arguments_without_a_gpu():
    import ctypes as C
    lib = N.lib()
    h = C.c_void_p()
    assert lib.ctmr_group_create_local(None, 0, C.byref(h)) == N.E_INVAL
    assert lib.ctmr_group_create_rccl(None, b"\0" * 128, 0, 1, C.byref(h)) == N.E_INVAL
    assert lib.ctmr_group_map_batch(None, 0, None, None) == N.E_INVAL
    assert lib.ctmr_group_issuer_counts(None, None, 0) == N.E_INVAL
    lib.ctmr_group_destroy(None)
