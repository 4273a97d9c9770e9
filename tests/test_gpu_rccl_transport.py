"""-m gpu: the RCCL TRANSPORT of the native group layer (csrc/engine/group.inc: grouped ncclSend/ncclRecv all-to-all,
in-place ncclAllGather of the counts matrix and the Bloom filters, ncclAllReduce of the per-issuer counts) driven with
a world of 2–4 ranks although only one GPU is reachable: the ranks are THREADS of one child process, each with its own
engine and its own `ctmr_group_create_rccl` communicator, and `CTMR_RCCL_LIB` points the library at an in-process
stand-in for librccl (tests/harness/fake_rccl.cpp) that moves the bytes with device-to-device copies.  What is under
test is everything group.inc does around the collectives — counts, offsets, peers, in-place rules, phase order — which
the in-process LOCAL transport (tests/test_gpu_exchange.py) does not run.  The real librccl is exercised with a world of
one there; N > 1 over real RCCL stays unmeasured until a multi-GPU node exists."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_SRC = os.path.join(ROOT, "tests", "harness", "fake_rccl.cpp")
FAKE_LIB = os.path.join(ROOT, "tests", "harness", "libfake_rccl.so")

CHILD = r'''
import json, sys, threading
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth
from ct_mapreduce_amd.distributed import Group, shard, shard_range
from ct_mapreduce_amd.engine import RECORD_DTYPE
from tests.gpu_common import run_oracle

world, mode = int(sys.argv[2]), sys.argv[3]
DEV = torch.device("cuda:0")
NOW, FILT = synth.BASE_TIME, b"Synth Issuer 0"
cfg = synth.config(seed=57, n_issuers=16, dup_permille=300, ca_permille=30, expired_permille=30)
n_total = 6000
issuers = synth.issuers(cfg)
whole = synth.host_batch(cfg, 0, n_total)
o, st, unk, eh = run_oracle(whole, issuers, FILT, False, NOW)
gid = Group.unique_id()
out, errs = [None] * world, []

def rank_main(r):
    try:
        lo, hi = shard_range(n_total, r, world)
        b = synth.host_batch(cfg, lo, hi - lo)
        pay = torch.from_numpy(np.concatenate([b.payload, np.zeros(64, np.uint8)])).to(DEV)
        off = torch.from_numpy(b.offsets.astype(np.int64)).to(DEV)
        iss = torch.from_numpy(b.issuer_idx.astype(np.int32)).to(DEV)
        et = torch.from_numpy(b.entry_type).to(DEV)
        rec = torch.zeros(b.n * 32, dtype=torch.uint8, device=DEV)
        new = torch.zeros(max(b.n, 1), dtype=torch.int64, device=DEV)
        e = ctmr.Engine(device=0, table_slots=1 << 16, pair_slots=1 << 12)
        e.add_issuers(issuers)
        e.set_filter(FILT, False, NOW)
        g = Group.rccl(e, gid, r, world)                      # collective: returns when every rank has joined
        if mode == "bloom":
            g.bloom_config(1 << 16)
        stats = g.map_batch(mode, [shard(pay.data_ptr(), off.data_ptr(), iss.data_ptr(), et.data_ptr(), b.n,
                                         rec.data_ptr(), new.data_ptr(), order_base=lo)])[0]
        counts = g.issuer_counts(len(issuers))
        total = g.total_count()
        mx = int(g.all_reduce_u64([r + 1], op_max=True)[0])
        g.barrier()
        info = g.info()
        recs = rec.cpu().numpy().view(RECORD_DTYPE)
        ok = bool((recs["status"] == st[lo:hi]).all() and (((recs["flags"] & 2) != 0) == (unk[lo:hi] != 0)).all())
        newl = new[:stats.n_new].cpu().numpy()
        ok = ok and stats.n_new == int(unk[lo:hi].sum()) and bool((newl == np.nonzero(unk[lo:hi])[0]).all())
        out[r] = {"ok": ok, "n_new": int(stats.n_new), "counts": [int(c) for c in counts], "total": int(total), "max": mx,
                  "keys_sent": int(info.keys_sent), "keys_received": int(info.keys_received)}
        g.close()
        e.close()
    except Exception as ex:   # noqa: BLE001
        errs.append(f"rank {r}: {type(ex).__name__}: {ex}")

ts = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
for t in ts: t.start()
for t in ts: t.join(30)
hung = [r for r, t in enumerate(ts) if t.is_alive()]
print(json.dumps({"out": out, "errs": errs, "hung": hung, "oracle_total": int(o.total_count()),
                  "oracle_new": int(unk.sum())}))
sys.stdout.flush()
import os
os._exit(0 if not hung else 3)
'''


def build_fake():
    from tests.harness import build_fake_rccl
    return build_fake_rccl()


@pytest.mark.parametrize("mode", ["owner", "bloom", "local"])
@pytest.mark.parametrize("world", [2, 3, 4])
def test_rccl_transport_with_several_ranks_on_one_gpu(world, mode):
    env = dict(os.environ, CTMR_RCCL_LIB=build_fake())
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(world), mode], env=env, capture_output=True, text=True,
                       timeout=100)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert line, p.stderr[-2000:]
    res = json.loads(line[-1])
    assert not res["errs"] and not res["hung"], res
    outs = res["out"]
    assert all(o_ is not None for o_ in outs)
    if mode != "local":                                         # shard-local dedup is not the global answer
        assert all(o_["ok"] for o_ in outs), outs
        assert sum(o_["n_new"] for o_ in outs) == res["oracle_new"]
        assert all(o_["total"] == res["oracle_total"] for o_ in outs)
    assert all(o_["counts"] == outs[0]["counts"] for o_ in outs)   # the all-reduce gave every rank the same sums
    assert all(o_["max"] == world for o_ in outs)
    if mode == "owner":
        assert sum(o_["keys_sent"] for o_ in outs) == sum(o_["keys_received"] for o_ in outs) > 0


FAIL_CHILD = r'''
import json, sys, threading, time
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth
from ct_mapreduce_amd.distributed import Group, shard, shard_range

world, mode = int(sys.argv[2]), sys.argv[3]
DEV = torch.device("cuda:0")
cfg = synth.config(seed=58, n_issuers=8, dup_permille=200)
issuers = synth.issuers(cfg)
gid = Group.unique_id()
out = [None] * world

def rank_main(r):
    lo, hi = shard_range(4000, r, world)
    b = synth.host_batch(cfg, lo, hi - lo)
    pay = torch.from_numpy(np.concatenate([b.payload, np.zeros(64, np.uint8)])).to(DEV)
    off = torch.from_numpy(b.offsets.astype(np.int64)).to(DEV)
    iss = torch.from_numpy(b.issuer_idx.astype(np.int32)).to(DEV)
    et = torch.from_numpy(b.entry_type).to(DEV)
    rec = torch.zeros(b.n * 32, dtype=torch.uint8, device=DEV)
    e = ctmr.Engine(device=0, table_slots=1 << 14, pair_slots=1 << 10)
    e.add_issuers(issuers)
    e.set_filter(b"", False, synth.BASE_TIME)
    g = Group.rccl(e, gid, r, world)
    if mode == "bloom":
        g.bloom_config(1 << 14)
    t0 = time.time()
    try:
        # rank 1 forgets its records buffer: ITS call fails — and so must everybody's, instead of waiting for it for ever
        g.map_batch(mode, [shard(pay.data_ptr(), off.data_ptr(), iss.data_ptr(), et.data_ptr(), b.n,
                                 0 if r == 1 else rec.data_ptr(), 0, order_base=lo)])
        out[r] = {"error": None}
    except ctmr.CtmrError as ex:
        out[r] = {"error": ex.code, "msg": str(ex), "seconds": time.time() - t0}
    g.close()
    e.close()

ts = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
for t in ts: t.start()
for t in ts: t.join(40)
print(json.dumps({"out": out, "hung": [r for r, t in enumerate(ts) if t.is_alive()]}))
sys.stdout.flush()
import os
os._exit(0)
'''


@pytest.mark.parametrize("mode", ["owner", "bloom"])
def test_a_rank_that_fails_does_not_leave_its_peers_in_a_collective(mode):
    """Round-2 advisor finding: a rank-local failure before or between collectives returned early on that rank only and
    every peer blocked for ever in the next all-gather / send-recv.  The control rows carry a status word now: all three
    ranks come back with an error, promptly."""
    env = dict(os.environ, CTMR_RCCL_LIB=build_fake())
    p = subprocess.run([sys.executable, "-c", FAIL_CHILD, ROOT, "3", mode], env=env, capture_output=True, text=True, timeout=120)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert line, p.stderr[-2000:]
    res = json.loads(line[-1])
    assert not res["hung"], res
    outs = res["out"]
    assert all(o_ is not None and o_["error"] is not None for o_ in outs), outs
    assert outs[1]["error"] == -1 and "d_records" in outs[1]["msg"]              # the culprit: CTMR_E_INVAL, its own message
    assert all("rank 1 failed" in outs[r]["msg"] for r in (0, 2)), outs          # the others: told who it was
    assert all(o_["seconds"] < 20 for o_ in outs)


LONG_CHILD = r'''
import json, random, sys, threading
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth, _native as N
from ct_mapreduce_amd.distributed import Group, shard, shard_range
from ct_mapreduce_amd.engine import Batch, RECORD_DTYPE
from tests import der as D
from tests.gpu_common import run_oracle

world, mode = int(sys.argv[2]), sys.argv[3]
DEV = torch.device("cuda:0")
rng = random.Random(7)
issuer = synth.issuer(synth.config(n_issuers=1), 0)
name = D.name(D.rdn(3, b"Synth Issuer 000"))
def cert(ln):
    s = bytes([rng.randrange(1, 0x7f)] + [rng.randrange(256) for _ in range(ln - 1)])
    return D.cert(serial=s, issuer=name, not_after=D.utctime("270101000000Z"))
first = [cert(rng.choice((41, 44, 48, 60, 12, 30))) for _ in range(45)]
fresh = [cert(rng.choice((41, 45, 50, 8))) for _ in range(30)]
round2 = first[::-1] + fresh + fresh[::-1] + first[:10]
rng.shuffle(round2)
rounds, o, base = [], None, 0
for certs in (first, round2, round2):
    whole = Batch.from_certs(certs, [0] * len(certs))
    whole.payload = np.concatenate([whole.payload, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    o, st, unk, eh = run_oracle(whole, [issuer], b"", True, 0, engine=o)
    rounds.append((certs, st, unk, base, int(o.total_count())))
    base += len(certs)
gid = Group.unique_id()
out, errs = [None] * world, []

def rank_main(r):
    try:
        e = ctmr.Engine(device=0, table_slots=1 << 12, pair_slots=1 << 10)
        e.add_issuers([issuer])
        e.set_filter(b"", True, 0)
        g = Group.rccl(e, gid, r, world)
        if mode == "bloom":
            g.bloom_config(1 << 14)
        if len(sys.argv) > 4:
            g.set_chunks(int(sys.argv[4]))
        ok, totals = True, []
        for certs, st, unk, base, total in rounds:
            lo, hi = shard_range(len(certs), r, world)
            b = Batch.from_certs(certs[lo:hi], [0] * (hi - lo))
            pay = torch.from_numpy(np.concatenate([b.payload, np.zeros(N.PAYLOAD_PAD, np.uint8)])).to(DEV)
            off = torch.from_numpy(b.offsets.astype(np.int64)).to(DEV)
            iss = torch.from_numpy(b.issuer_idx.astype(np.int32)).to(DEV)
            et = torch.from_numpy(b.entry_type).to(DEV)
            rec = torch.zeros(b.n * 32, dtype=torch.uint8, device=DEV)
            new = torch.zeros(max(b.n, 1), dtype=torch.int64, device=DEV)
            stats = g.map_batch(mode, [shard(pay.data_ptr(), off.data_ptr(), iss.data_ptr(), et.data_ptr(), b.n,
                                             rec.data_ptr(), new.data_ptr(), order_base=base + lo)])[0]
            recs = rec.cpu().numpy().view(RECORD_DTYPE)
            ok = ok and bool((recs["status"] == st[lo:hi]).all() and (((recs["flags"] & 2) != 0) == (unk[lo:hi] != 0)).all())
            newl = new[:stats.n_new].cpu().numpy()
            ok = ok and stats.n_new == int(unk[lo:hi].sum()) and bool((newl == np.nonzero(unk[lo:hi])[0]).all())
            totals.append(int(g.total_count()) == total)
        out[r] = {"ok": ok, "totals": totals}
        g.close()
        e.close()
    except Exception as ex:   # noqa: BLE001
        errs.append(f"rank {r}: {type(ex).__name__}: {ex}")

ts = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
for t in ts: t.start()
for t in ts: t.join(40)
hung = [r for r, t in enumerate(ts) if t.is_alive()]
print(json.dumps({"out": out, "errs": errs, "hung": hung}))
sys.stdout.flush()
import os
os._exit(0 if not hung else 3)
'''


def test_chunked_owner_rounds_over_the_rccl_transport():
    """The same three rounds with every shard mapped in three chunks (ctmr_group_set_chunks): one control row and one
    grouped send/recv per chunk on the transfer stream, the 64-byte records and the long serials behind the last chunk."""
    env = dict(os.environ, CTMR_RCCL_LIB=build_fake())
    p = subprocess.run([sys.executable, "-c", LONG_CHILD, ROOT, "3", "owner", "3"], env=env, capture_output=True, text=True,
                       timeout=100)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert line, p.stderr[-2000:]
    res = json.loads(line[-1])
    assert not res["errs"] and not res["hung"], res
    assert all(o_ is not None and o_["ok"] and all(o_["totals"]) for o_ in res["out"]), res


@pytest.mark.parametrize("mode", ["owner", "bloom"])
def test_long_serials_are_settled_over_the_rccl_transport(mode):
    """Members with serials beyond CTMR_MAX_SERIAL live in host-side sets; a group round settles them between the ranks
    through the closing control rows (all-gather of the round's additions, all-reduce of "held before").  Three ranks over
    the stand-in librccl, three rounds: keys new in round 1, the same keys on other ranks plus fresh ones twice in
    round 2, a replay — records, NEW lists and the summed cardinality equal the single-stream oracle's."""
    env = dict(os.environ, CTMR_RCCL_LIB=build_fake())
    p = subprocess.run([sys.executable, "-c", LONG_CHILD, ROOT, "3", mode], env=env, capture_output=True, text=True,
                       timeout=100)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert line, p.stderr[-2000:]
    res = json.loads(line[-1])
    assert not res["errs"] and not res["hung"], res
    assert all(o_ is not None and o_["ok"] and all(o_["totals"]) for o_ in res["out"]), res
