"""-m gpu: cross-GPU global dedup (owner-computes key exchange) validated on one GPU —
world=1 (self exchange) equals the local path, and a simulated world of 2 and 3 ranks (several
engines on one device, tensors sliced instead of sent) equals the oracle over the WHOLE stream."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth, _native as N
from ct_mapreduce_amd.distributed import GlobalDedupRank, run_global_dedup, run_simulated, shard_range
from ct_mapreduce_amd.engine import RECORD_DTYPE
from tests.gpu_common import run_oracle

DEV = torch.device("cuda:0")
NOW = synth.BASE_TIME
FILT = b"Synth Issuer 0"


def to_dev(b):
    pay = torch.from_numpy(np.concatenate([b.payload, np.zeros(64, np.uint8)])).to(DEV)
    off = torch.from_numpy(b.offsets.astype(np.int64)).to(DEV)
    iss = torch.from_numpy(b.issuer_idx.astype(np.int32)).to(DEV)
    et = torch.from_numpy(b.entry_type).to(DEV)
    rec = torch.zeros(b.n * 32, dtype=torch.uint8, device=DEV)
    new = torch.zeros(max(b.n, 1), dtype=torch.int64, device=DEV)
    return pay, off, iss, et, rec, new


def make_engine(issuers):
    e = ctmr.Engine(device=0, table_slots=1 << 16, pair_slots=1 << 12)
    e.add_issuers(issuers)
    e.set_filter(FILT, False, NOW)
    return e


@pytest.mark.parametrize("world", [1, 2, 3])
def test_global_dedup_matches_oracle_over_the_whole_stream(world):
    cfg = synth.config(seed=55, n_issuers=16, dup_permille=300, ca_permille=30, expired_permille=30)
    n_total = 6000
    issuers = synth.issuers(cfg)
    whole = synth.host_batch(cfg, 0, n_total)
    o, st, unk, eh = run_oracle(whole, issuers, FILT, False, NOW)
    assert 0 < unk.sum() < (st == 0).sum()                       # duplicates exist, also across shards
    engines = [make_engine(issuers) for _ in range(world)]
    ranks = [GlobalDedupRank(engines[r], r, world, DEV) for r in range(world)]
    shards, keep, ranges = [], [], []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        b = synth.host_batch(cfg, lo, hi - lo)
        pay, off, iss, et, rec, new = to_dev(b)
        keep.append((pay, off, iss, et, rec, new))
        shards.append((pay.data_ptr(), off.data_ptr(), iss.data_ptr(), et.data_ptr(), b.n, rec.data_ptr()))
        ranges.append((lo, hi))
    if world == 1:
        stats = [run_global_dedup(ranks[0], *shards[0], keep[0][5].data_ptr())]
    else:
        stats = run_simulated(ranks, shards, [k[5].data_ptr() for k in keep])
    # every shard's records equal the oracle's view of that slice of the global stream
    for r in range(world):
        lo, hi = ranges[r]
        rec = keep[r][4].cpu().numpy().view(RECORD_DTYPE)
        assert (rec["status"] == st[lo:hi]).all()
        assert (((rec["flags"] & 2) != 0) == (unk[lo:hi] != 0)).all()
        assert stats[r].n_new == int(unk[lo:hi].sum())
        new = keep[r][5][:stats[r].n_new].cpu().numpy()
        assert (new == np.nonzero(unk[lo:hi])[0]).all()
        for k in range(7):
            assert stats[r].by_status[k] == int((st[lo:hi] == k).sum())
    # owner-local per-issuer counters sum to the global unique counts; every key has one owner
    total = np.zeros(len(issuers), dtype=np.uint64)
    for e in engines:
        total += e.issuer_counts()
    for k in range(len(issuers)):
        assert int(total[k]) == o.issuer_count(engines[0].issuer_id(k))
    assert sum(e.total_count() for e in engines) == o.total_count()
    allkeys = sorted(sum((e.keys(b"serials::*") for e in engines), []))
    assert sorted(set(allkeys)) == [k for k in o.keys() if k.startswith(b"serials::")]
    # replay: nothing is new anywhere
    if world > 1:
        stats2 = run_simulated(ranks, shards)
        assert all(s.n_new == 0 for s in stats2)
    for e in engines:
        e.close()


def test_global_path_equals_local_path_on_one_gpu():
    cfg = synth.config(seed=56, n_issuers=4, dup_permille=200)
    issuers = synth.issuers(cfg)
    b = synth.host_batch(cfg, 0, 5000)
    pay, off, iss, et, rec, new = to_dev(b)
    e1, e2 = make_engine(issuers), make_engine(issuers)
    st_local = e1.map_batch_device(pay.data_ptr(), off.data_ptr(), iss.data_ptr(), et.data_ptr(), b.n,
                                   rec.data_ptr(), new.data_ptr())
    rec_local = rec.cpu().numpy().copy()
    new_local = new[:st_local.n_new].cpu().numpy().copy()
    rec.zero_()
    st_glob = run_global_dedup(GlobalDedupRank(e2, 0, 1, DEV), pay.data_ptr(), off.data_ptr(), iss.data_ptr(),
                               et.data_ptr(), b.n, rec.data_ptr(), new.data_ptr())
    assert (rec.cpu().numpy() == rec_local).all()
    assert (new[:st_glob.n_new].cpu().numpy() == new_local).all()
    assert st_glob.n_new == st_local.n_new and list(st_glob.by_status) == list(st_local.by_status)
    assert (e1.issuer_counts() == e2.issuer_counts()).all()
    e1.close(); e2.close()


def test_raw_shards_with_synchronised_issuer_registration():
    """Global dedup fed by RAW get-entries shards: two "ranks" (engines) see the issuers in different orders; with
    auto-registration off and the pending lists merged in an agreed order their tables stay index-identical, and the
    key exchange over entry views reproduces the single-stream oracle."""
    from ct_mapreduce_amd.distributed import decode_synchronised
    from oracle import oracle as orc
    world = 2
    cfg = synth.config(seed=57, n_issuers=24, dup_permille=300, ca_permille=30, expired_permille=30)
    n_total = 5000
    whole = synth.host_entries(cfg, 0, n_total)
    o = orc.Engine(FILT, False, NOW)
    st, unk, eh, ts = o.raw_batch(whole.blob, whole.bounds)
    assert 0 < unk.sum() < (st == 0).sum()
    engines = []
    for _ in range(world):
        e = ctmr.Engine(device=0, table_slots=1 << 16, pair_slots=1 << 12)
        e.set_filter(FILT, False, NOW)
        e.set_issuer_autoregister(False)
        engines.append(e)
    ranks = [GlobalDedupRank(engines[r], r, world, DEV) for r in range(world)]
    keep, calls, ranges = [], [], []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        raw = synth.host_entries(cfg, lo, hi - lo)
        n = raw.n
        d_blob = torch.from_numpy(raw.blob.copy()).to(DEV)
        d_bounds = torch.from_numpy(raw.bounds.astype(np.int64)).to(DEV)
        t = {k: torch.zeros(n, dtype=dt, device=DEV) for k, dt in
             (("start", torch.int64), ("end", torch.int64), ("iss", torch.int32), ("et", torch.uint8))}
        view = N.EntryView(cert_start=t["start"].data_ptr(), cert_end=t["end"].data_ptr(), issuer_idx=t["iss"].data_ptr(),
                           entry_type=t["et"].data_ptr(), timestamp=None, chain0_start=None, chain0_len=None)
        rec = torch.zeros(n * 32, dtype=torch.uint8, device=DEV)
        new = torch.zeros(n, dtype=torch.int64, device=DEV)
        keep.append((d_blob, d_bounds, t, view, rec, new, n, int(raw.bounds[-1])))
        calls.append(lambda e=engines[r], b=d_blob, bd=d_bounds, n=n, v=view: e.decode_entries_device(b.data_ptr(), bd.data_ptr(), n, v))
        ranges.append((lo, hi))
    with pytest.raises(ctmr.CtmrError) as ei:                   # nothing is registered silently
        calls[0]()
    assert ei.value.code == N.E_NOTFOUND and len(engines[0].pending_issuers()) > 0
    decode_synchronised(engines, calls)
    assert engines[0].issuer_count() == engines[1].issuer_count() > 0
    for k in range(engines[0].issuer_count()):                  # index-identical issuer tables
        assert engines[0].issuer_id(k) == engines[1].issuer_id(k)
    # key exchange over the views (same data movement as run_simulated, export through the view entry point)
    counts = []
    for r in range(world):
        d_blob, _, _, view, rec, _, n, nbytes = keep[r]
        ranks[r].n, ranks[r].d_records = n, rec.data_ptr()
        ranks[r].keys = torch.empty(max(n, 1) * 64, dtype=torch.uint8, device=DEV)
        ranks[r].send_counts = engines[r].exchange_export_view(d_blob.data_ptr(), nbytes, view, n, rec.data_ptr(), world,
                                                               ranks[r].keys.data_ptr())
        ranks[r].n_keys = sum(ranks[r].send_counts)
        counts.append(ranks[r].send_counts)
    for ow, rk in enumerate(ranks):
        rk.insert(torch.cat([ranks[s].partition(ow) for s in range(world)]), [counts[s][ow] for s in range(world)])
    for r, rk in enumerate(ranks):
        stats = rk.apply(torch.cat([ranks[ow].flags_for(r) for ow in range(world)]), keep[r][5].data_ptr())
        lo, hi = ranges[r]
        rec = keep[r][4].cpu().numpy().view(RECORD_DTYPE)
        assert (rec["status"] == st[lo:hi]).all()
        assert (((rec["flags"] & 2) != 0) == (unk[lo:hi] != 0)).all()
        assert stats.n_new == int(unk[lo:hi].sum())
    assert sum(e.total_count() for e in engines) == o.total_count()
    for e in engines:
        e.close()
