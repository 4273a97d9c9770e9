"""-m gpu: cross-GPU global dedup, Bloom pre-filter variant (BASELINE north_star "all-gather of per-GPU Bloom
fingerprints") validated on one GPU: a simulated world of 1–4 ranks (several engines on one
device, tensor copies instead of collectives) equals the oracle over the WHOLE stream — one round, several rounds with
duplicates across rounds and ranks, a filter so small that false positives dominate — and equals the
owner-computes exchange."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth, _native as N
from ct_mapreduce_amd.distributed import (BloomDedupRank, GlobalDedupRank, run_bloom_dedup, run_simulated,
                                          run_simulated_bloom, shard_range)
from ct_mapreduce_amd.engine import RECORD_DTYPE
from oracle import oracle as orc
from tests.gpu_common import run_oracle
from tests.test_gpu_exchange import to_dev, make_engine, FILT, NOW, DEV


def build_world(world, issuers, bits):
    engines = [make_engine(issuers) for _ in range(world)]
    return engines, [BloomDedupRank(engines[r], r, world, DEV, bits) for r in range(world)]


def load_shards(cfg, lo_all, hi_all, world):
    shards, keep, ranges = [], [], []
    for r in range(world):
        lo, hi = shard_range(hi_all - lo_all, r, world)
        lo, hi = lo + lo_all, hi + lo_all
        b = synth.host_batch(cfg, lo, hi - lo)
        t = to_dev(b)
        keep.append(t)
        shards.append((t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), b.n, t[4].data_ptr()))
        ranges.append((lo, hi))
    return shards, keep, ranges


def check_round(keep, ranges, stats, st, unk, base=0):
    for r, (lo, hi) in enumerate(ranges):
        lo, hi = lo - base, hi - base
        rec = keep[r][4].cpu().numpy().view(RECORD_DTYPE)
        assert (rec["status"] == st[lo:hi]).all()
        got = (rec["flags"] & 2) != 0
        assert (got == (unk[lo:hi] != 0)).all(), (r, np.nonzero(got != (unk[lo:hi] != 0))[0][:10])
        assert stats[r].n_new == int(unk[lo:hi].sum())
        assert stats[r].n_dup == int(((st[lo:hi] == 0) & (unk[lo:hi] == 0)).sum())
        new = keep[r][5][:stats[r].n_new].cpu().numpy()
        assert (new == np.nonzero(unk[lo:hi])[0]).all()
        for k in range(8):
            assert stats[r].by_status[k] == int((st[lo:hi] == k).sum())


def check_state(engines, o, n_issuers):
    total = np.zeros(n_issuers, dtype=np.uint64)
    for e in engines:
        total += e.issuer_counts()
    for k in range(n_issuers):
        assert int(total[k]) == o.issuer_count(engines[0].issuer_id(k)), k
    assert sum(e.total_count() for e in engines) == o.total_count()
    okeys = [k for k in o.keys() if k.startswith(b"serials::")]
    assert sorted(set(sum((e.keys(b"serials::*") for e in engines), []))) == okeys
    # the ranks' sets are disjoint: cardinalities and member lists add up to the oracle's
    step = max(1, len(okeys) // 40)
    for key in okeys[::step]:
        assert sum(e.set_cardinality(key) for e in engines) == o.set_cardinality(key)
        members = sum((e.set_list(key) for e in engines), [])
        assert len(members) == len(set(members)) and sorted(members) == sorted(o.members(key))


@pytest.mark.parametrize("world", [1, 2, 3, 4])
def test_bloom_dedup_matches_oracle_over_the_whole_stream(world):
    cfg = synth.config(seed=61, n_issuers=16, dup_permille=300, ca_permille=30, expired_permille=30)
    n_total = 6000
    issuers = synth.issuers(cfg)
    o, st, unk, eh = run_oracle(synth.host_batch(cfg, 0, n_total), issuers, FILT, False, NOW)
    assert 0 < unk.sum() < (st == 0).sum()
    engines, ranks = build_world(world, issuers, 1 << 17)
    shards, keep, ranges = load_shards(cfg, 0, n_total, world)
    if world == 1:
        stats = [run_bloom_dedup(ranks[0], *shards[0], keep[0][5].data_ptr())]
    else:
        stats = run_simulated_bloom(ranks, shards, [k[5].data_ptr() for k in keep], [lo for lo, _ in ranges])
        assert sum(sum(r.send_counts) for r in ranks) > 0            # cross-rank duplicates were exchanged …
        assert sum(sum(r.send_counts) for r in ranks) < int((st == 0).sum())   # … and most keys were not
    check_round(keep, ranges, stats, st, unk)
    check_state(engines, o, len(issuers))
    if world > 1:   # replay: nothing is new anywhere
        stats2 = run_simulated_bloom(ranks, shards, None, [lo for lo, _ in ranges])
        assert all(s.n_new == 0 for s in stats2)
        check_state(engines, o, len(issuers))
    for e in engines:
        e.close()


@pytest.mark.parametrize("world,bits", [(3, 1 << 18), (4, 1 << 12)])
def test_bloom_dedup_stream_in_rounds(world, bits):
    """Several rounds through persistent tables and cumulative filters: duplicates of earlier rounds held by OTHER
    ranks are found through the filters; with a 4096-bit filter false positives dominate (and the key buffer's
    grow-and-retry path runs) — the result does not change."""
    cfg = synth.config(seed=62, n_issuers=12, dup_permille=300, ca_permille=20, expired_permille=20)
    issuers = synth.issuers(cfg)
    engines, ranks = build_world(world, issuers, bits)
    o = None
    W = 3000
    exchanged = 0
    for wave in range(4):
        lo_all, hi_all = wave * W, (wave + 1) * W
        o, st, unk, eh = run_oracle(synth.host_batch(cfg, lo_all, W), issuers, FILT, False, NOW, engine=o)
        shards, keep, ranges = load_shards(cfg, lo_all, hi_all, world)
        stats = run_simulated_bloom(ranks, shards, [k[5].data_ptr() for k in keep], [lo for lo, _ in ranges])
        check_round(keep, ranges, stats, st, unk, base=lo_all)
        exchanged += sum(sum(r.send_counts) for r in ranks)
    check_state(engines, o, len(issuers))
    if bits == 1 << 12:
        assert exchanged > o.total_count() // 2       # 64-word filter filling up: false positives dominate
    else:
        assert 0 < exchanged < o.total_count()        # only cross-rank duplicates travel
    for e in engines:
        e.close()


def test_bloom_and_owner_exchange_agree():
    world = 3
    cfg = synth.config(seed=63, n_issuers=8, dup_permille=400)
    n_total = 5000
    issuers = synth.issuers(cfg)
    engines_b, ranks_b = build_world(world, issuers, 1 << 16)
    engines_o = [make_engine(issuers) for _ in range(world)]
    ranks_o = [GlobalDedupRank(engines_o[r], r, world, DEV) for r in range(world)]
    shards_b, keep_b, ranges = load_shards(cfg, 0, n_total, world)
    shards_o, keep_o, _ = load_shards(cfg, 0, n_total, world)
    sb = run_simulated_bloom(ranks_b, shards_b, [k[5].data_ptr() for k in keep_b], [lo for lo, _ in ranges])
    so = run_simulated(ranks_o, shards_o, [k[5].data_ptr() for k in keep_o])
    for r in range(world):
        assert (keep_b[r][4].cpu().numpy() == keep_o[r][4].cpu().numpy()).all()
        assert sb[r].n_new == so[r].n_new and list(sb[r].by_status) == list(so[r].by_status)
        assert (keep_b[r][5][:sb[r].n_new].cpu().numpy() == keep_o[r][5][:so[r].n_new].cpu().numpy()).all()
    tb = sum(e.issuer_counts() for e in engines_b)
    to = sum(e.issuer_counts() for e in engines_o)
    assert (tb == to).all()
    for e in engines_b + engines_o:
        e.close()


def test_bloom_dedup_over_entry_views():
    """Raw get-entries shards: the probe reads certificates through the entry view (d_ends)."""
    from ct_mapreduce_amd.distributed import decode_synchronised
    from oracle import oracle as orc
    world = 2
    cfg = synth.config(seed=64, n_issuers=20, dup_permille=300, ca_permille=30, expired_permille=30)
    n_total = 4000
    whole = synth.host_entries(cfg, 0, n_total)
    o = orc.Engine(FILT, False, NOW)
    st, unk, eh, ts = o.raw_batch(whole.blob, whole.bounds)
    engines = []
    for _ in range(world):
        e = ctmr.Engine(device=0, table_slots=1 << 16, pair_slots=1 << 12)
        e.set_filter(FILT, False, NOW)
        e.set_issuer_autoregister(False)
        engines.append(e)
    ranks = [BloomDedupRank(engines[r], r, world, DEV, 1 << 16) for r in range(world)]
    keep, calls, ranges = [], [], []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        raw = synth.host_entries(cfg, lo, hi - lo)
        n = raw.n
        d_blob = torch.from_numpy(raw.blob.copy()).to(DEV)
        d_bounds = torch.from_numpy(raw.bounds.astype(np.int64)).to(DEV)
        t = {k: torch.zeros(n, dtype=dt, device=DEV) for k, dt in
             (("start", torch.int64), ("end", torch.int64), ("iss", torch.int32), ("et", torch.uint8))}
        view = N.EntryView(cert_start=t["start"].data_ptr(), cert_end=t["end"].data_ptr(),
                           issuer_idx=t["iss"].data_ptr(), entry_type=t["et"].data_ptr(), timestamp=None,
                           chain0_start=None, chain0_len=None)
        rec = torch.zeros(n * 32, dtype=torch.uint8, device=DEV)
        new = torch.zeros(n, dtype=torch.int64, device=DEV)
        keep.append((d_blob, d_bounds, t, view, rec, new, n, len(raw.blob)))
        calls.append(lambda e=engines[r], b=d_blob, bd=d_bounds, n=n, v=view: e.decode_entries_device(
            b.data_ptr(), bd.data_ptr(), n, v))
        ranges.append((lo, hi))
    decode_synchronised(engines, calls)
    for r, rk in enumerate(ranks):
        d_blob, d_bounds, t, view, rec, new, n, nbytes = keep[r]
        rk.map(d_blob.data_ptr(), t["start"].data_ptr(), 0, 0, n, rec.data_ptr(), d_ends=t["end"].data_ptr(),
               order_base=ranges[r][0], view=view, blob_bytes=nbytes)
    for rk in ranks:
        for p, q in enumerate(ranks):
            if q is not rk:
                rk.filters[p].copy_(q.own_filter())
    counts = [rk.probe() for rk in ranks]
    for o_, rk in enumerate(ranks):
        rk.lookup(torch.cat([ranks[s].partition(o_) for s in range(world)]), [counts[s][o_] for s in range(world)])
    for k, rk in enumerate(ranks):
        fl = torch.cat([ranks[p].flags_for(k) for p in range(world)])
        if fl.numel() == 0:
            fl = torch.zeros(1, dtype=torch.uint8, device=DEV)
        stt = rk.apply(fl, keep[k][5].data_ptr())
        lo, hi = ranges[k]
        rec = keep[k][4].cpu().numpy().view(RECORD_DTYPE)
        assert (rec["status"] == st[lo:hi]).all()
        assert (((rec["flags"] & 2) != 0) == (unk[lo:hi] != 0)).all()
        assert stt.n_new == int(unk[lo:hi].sum())
    assert sum(e.total_count() for e in engines) == o.total_count()
    for e in engines:
        e.close()


def test_shadow_members_are_not_counted_twice_by_sweep_and_remove():
    world = 2
    cfg = synth.config(seed=65, n_issuers=4, dup_permille=500)
    issuers = synth.issuers(cfg)
    engines, ranks = build_world(world, issuers, 1 << 15)
    shards, keep, ranges = load_shards(cfg, 0, 3000, world)
    run_simulated_bloom(ranks, shards, None, [lo for lo, _ in ranges])
    before = sum(e.total_count() for e in engines)
    removed = sum(e.expire_sweep(NOW + 400 * 86400) for e in engines)      # everything has expired by then
    assert removed == before
    assert sum(e.total_count() for e in engines) == 0
    assert all(e.keys(b"serials::*") == [] for e in engines)
    for e in engines:
        e.close()


def test_bloom_config_errors():
    e = make_engine(synth.issuers(synth.config(seed=66, n_issuers=2)))
    with pytest.raises(ctmr.CtmrError):
        e.bloom_config(1000)                     # not a power of two
    with pytest.raises(ctmr.CtmrError):
        e.bloom_add(0, 0, 0, 0)                  # no filter configured
    e.bloom_config(1 << 12)
    p, nw = e.bloom_device()
    assert p != 0 and nw == 64
    e.bloom_add(0, 0, 0, 0)                      # empty batch: fine
    e.close()


def test_point_inserted_keys_are_in_the_filter():
    """A member added with SetInsert on one rank (ctmr_set_insert) is known to the others: the point insert sets the
    filter bits too, so the batch entry of another rank is sent over, found and loses WasUnknown."""
    world = 2
    cfg = synth.config(seed=67, n_issuers=2)
    issuers = synth.issuers(cfg)
    engines, ranks = build_world(world, issuers, 1 << 14)
    b = synth.host_batch(cfg, 0, 200)
    o, st, unk, eh = run_oracle(b, issuers, FILT, False, NOW)
    i = int(np.nonzero(unk)[0][3])                               # some new, passing entry of rank 0's batch
    c = orc.parse_cert(b.cert(i))
    key = b"serials::%s::%s" % (orc.exp_date_id(int(eh[i])).encode(), engines[1].issuer_id(int(b.issuer_idx[i])).encode())
    assert engines[1].set_insert(key, b.cert(i)[c.serial_off:c.serial_off + c.serial_len]) is True
    t = to_dev(b)
    empty = to_dev(synth.host_batch(cfg, 1000, 0))
    shards = [(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), b.n, t[4].data_ptr()),
              (empty[0].data_ptr(), empty[1].data_ptr(), empty[2].data_ptr(), empty[3].data_ptr(), 0, empty[4].data_ptr())]
    stats = run_simulated_bloom(ranks, shards, [t[5].data_ptr(), empty[5].data_ptr()], [0, 1000])
    rec = t[4].cpu().numpy().view(RECORD_DTYPE)
    got = (rec["flags"] & 2) != 0
    want = unk != 0
    want[i] = False                                              # known on rank 1 since before this round
    assert (got == want).all()
    assert stats[0].n_new == int(want.sum()) and sum(ranks[0].send_counts) >= 1
    assert sum(e.total_count() for e in engines) == int(unk.sum())      # counted once, on rank 1
    for e in engines:
        e.close()
