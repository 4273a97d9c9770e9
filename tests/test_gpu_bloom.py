"""-m gpu: cross-GPU global dedup, Bloom pre-filter variant (BASELINE north_star "all-gather of per-GPU Bloom
fingerprints") through the NATIVE group layer on one GPU: a local group of 1–4 ranks (several
engines on one device: the RCCL transport's phase drivers and kernels, copies where it would send) equals the oracle
over the WHOLE stream — one round, several rounds with duplicates across rounds and ranks, a filter so small that
false positives dominate — and equals the owner-computes exchange.  (tests/test_gpu_exchange.py runs both modes
through the same scenarios; this file holds what is specific to the filters.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth, _native as N
from ct_mapreduce_amd.distributed import Group, shard_range
from ct_mapreduce_amd.engine import RECORD_DTYPE
from oracle import oracle as orc
from tests.gpu_common import run_oracle
from tests.test_gpu_exchange import to_dev, dev_shard, make_engine, FILT, NOW, DEV


def build_world(world, issuers, bits):
    engines = [make_engine(issuers) for _ in range(world)]
    g = Group.local(engines)
    g.bloom_config(bits)
    return engines, g


def load_shards(cfg, lo_all, hi_all, world):
    shards, keep, ranges = [], [], []
    for r in range(world):
        lo, hi = shard_range(hi_all - lo_all, r, world)
        lo, hi = lo + lo_all, hi + lo_all
        b = synth.host_batch(cfg, lo, hi - lo)
        t = to_dev(b)
        keep.append(t)
        shards.append(dev_shard(t, b.n, order_base=lo))
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


def check_state(engines, g, o, n_issuers):
    total = g.issuer_counts(n_issuers)
    for k in range(n_issuers):
        assert int(total[k]) == o.issuer_count(engines[0].issuer_id(k)), k
    assert g.total_count() == o.total_count() == sum(e.total_count() for e in engines)
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
    engines, g = build_world(world, issuers, 1 << 17)
    shards, keep, ranges = load_shards(cfg, 0, n_total, world)
    stats = g.map_batch("bloom", shards)
    info = g.info()
    if world > 1:
        assert 0 < info.keys_sent < int((st == 0).sum())           # cross-rank duplicates were exchanged, most keys were not
        assert info.filter_bytes_received == (world - 1) * (1 << 17) // 8
    else:
        assert info.keys_sent == 0
    check_round(keep, ranges, stats, st, unk)
    check_state(engines, g, o, len(issuers))
    stats2 = g.map_batch("bloom", shards)                            # replay: nothing is new anywhere
    assert all(s.n_new == 0 for s in stats2)
    check_state(engines, g, o, len(issuers))
    g.close()
    for e in engines:
        e.close()


@pytest.mark.parametrize("world,bits", [(3, 1 << 18), (4, 1 << 12)])
def test_bloom_dedup_stream_in_rounds(world, bits):
    """Several rounds through persistent tables and cumulative filters: duplicates of earlier rounds held by OTHER
    ranks are found through the filters; with a 4096-bit filter false positives dominate (and the key buffer's
    grow-and-retry path runs) — the result does not change."""
    cfg = synth.config(seed=62, n_issuers=12, dup_permille=300, ca_permille=20, expired_permille=20)
    issuers = synth.issuers(cfg)
    engines, g = build_world(world, issuers, bits)
    o = None
    W = 3000
    exchanged = 0
    for wave in range(4):
        lo_all, hi_all = wave * W, (wave + 1) * W
        o, st, unk, eh = run_oracle(synth.host_batch(cfg, lo_all, W), issuers, FILT, False, NOW, engine=o)
        shards, keep, ranges = load_shards(cfg, lo_all, hi_all, world)
        stats = g.map_batch("bloom", shards)
        check_round(keep, ranges, stats, st, unk, base=lo_all)
        exchanged += g.info().keys_sent
    check_state(engines, g, o, len(issuers))
    if bits == 1 << 12:
        assert exchanged > o.total_count() // 2       # 64-word filter filling up: false positives dominate
    else:
        assert 0 < exchanged < o.total_count()        # only cross-rank duplicates travel
    g.close()
    for e in engines:
        e.close()


def test_bloom_and_owner_exchange_agree():
    world = 3
    cfg = synth.config(seed=63, n_issuers=8, dup_permille=400)
    n_total = 5000
    issuers = synth.issuers(cfg)
    engines_b, gb = build_world(world, issuers, 1 << 16)
    engines_o = [make_engine(issuers) for _ in range(world)]
    go = Group.local(engines_o)
    shards_b, keep_b, ranges = load_shards(cfg, 0, n_total, world)
    shards_o, keep_o, _ = load_shards(cfg, 0, n_total, world)
    sb = gb.map_batch("bloom", shards_b)
    so = go.map_batch("owner", shards_o)
    for r in range(world):
        assert (keep_b[r][4].cpu().numpy() == keep_o[r][4].cpu().numpy()).all()
        assert sb[r].n_new == so[r].n_new and list(sb[r].by_status) == list(so[r].by_status)
        assert (keep_b[r][5][:sb[r].n_new].cpu().numpy() == keep_o[r][5][:so[r].n_new].cpu().numpy()).all()
    assert (gb.issuer_counts(len(issuers)) == go.issuer_counts(len(issuers))).all()
    assert gb.info().keys_sent < go.info().keys_sent                 # the filters keep most keys at home
    gb.close(); go.close()
    for e in engines_b + engines_o:
        e.close()


def test_shadow_members_are_not_counted_twice_by_sweep_and_remove():
    world = 2
    cfg = synth.config(seed=65, n_issuers=4, dup_permille=500)
    issuers = synth.issuers(cfg)
    engines, g = build_world(world, issuers, 1 << 15)
    shards, keep, ranges = load_shards(cfg, 0, 3000, world)
    g.map_batch("bloom", shards)
    before = g.total_count()
    removed = sum(e.expire_sweep(NOW + 400 * 86400) for e in engines)      # everything has expired by then
    assert removed == before
    assert g.total_count() == 0
    assert all(e.keys(b"serials::*") == [] for e in engines)
    g.close()
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
    g = Group.local([e])
    with pytest.raises(ctmr.CtmrError):
        g.bloom_config(1000)
    with pytest.raises(ctmr.CtmrError):
        Group.local([make_engine(synth.issuers(synth.config(seed=66, n_issuers=2)))]).map_batch("bloom", [dev_shard(to_dev(synth.host_batch(synth.config(seed=66, n_issuers=2), 0, 10)), 10)])
    g.close()
    e.close()


def test_point_inserted_keys_are_in_the_filter():
    """A member added with SetInsert on one rank (ctmr_set_insert) is known to the others: the point insert sets the
    filter bits too, so the batch entry of another rank is sent over, found and loses WasUnknown."""
    world = 2
    cfg = synth.config(seed=67, n_issuers=2)
    issuers = synth.issuers(cfg)
    engines, g = build_world(world, issuers, 1 << 14)
    b = synth.host_batch(cfg, 0, 200)
    o, st, unk, eh = run_oracle(b, issuers, FILT, False, NOW)
    i = int(np.nonzero(unk)[0][3])                               # some new, passing entry of rank 0's batch
    c = orc.parse_cert(b.cert(i))
    key = b"serials::%s::%s" % (orc.exp_date_id(int(eh[i])).encode(), engines[1].issuer_id(int(b.issuer_idx[i])).encode())
    assert engines[1].set_insert(key, b.cert(i)[c.serial_off:c.serial_off + c.serial_len]) is True
    t = to_dev(b)
    empty = to_dev(synth.host_batch(cfg, 1000, 0))
    stats = g.map_batch("bloom", [dev_shard(t, b.n, order_base=0), dev_shard(empty, 0, order_base=1000)])
    rec = t[4].cpu().numpy().view(RECORD_DTYPE)
    got = (rec["flags"] & 2) != 0
    want = unk != 0
    want[i] = False                                              # known on rank 1 since before this round
    assert (got == want).all()
    assert stats[0].n_new == int(want.sum()) and g.info().keys_sent >= 1
    assert g.total_count() == int(unk.sum())                     # counted once, on rank 1
    g.close()
    for e in engines:
        e.close()


def test_a_round_the_group_must_refuse_fails_before_anything_is_inserted():
    """The opening control row of an exact round (group.inc round_header): overlapping or descending order ranges in a
    Bloom round, a Bloom round without a filter, an unknown mode and a second mode on one group are refused on every rank
    with nothing inserted — and the group is still usable afterwards."""
    world = 2
    cfg = synth.config(seed=68, n_issuers=2)
    issuers = synth.issuers(cfg)
    engines, g = build_world(world, issuers, 1 << 14)
    shards, keep, ranges = load_shards(cfg, 0, 400, world)
    same = [dev_shard(keep[0], 200, order_base=0), dev_shard(keep[1], 200, order_base=0)]          # the forgotten order_base
    down = [dev_shard(keep[0], 200, order_base=200), dev_shard(keep[1], 200, order_base=0)]
    lap = [dev_shard(keep[0], 200, order_base=0), dev_shard(keep[1], 200, order_base=199)]
    for bad in (same, down, lap):
        with pytest.raises(ctmr.CtmrError, match="order ranges"):
            g.map_batch("bloom", bad)
        assert g.total_count() == 0
    with pytest.raises(ctmr.CtmrError):
        g.map_batch(7, shards)
    assert g.total_count() == 0
    b = synth.host_batch(cfg, 0, 400)
    o, st, unk, eh = run_oracle(b, issuers, FILT, False, NOW)
    stats = g.map_batch("bloom", shards)                                                          # the proper round still runs
    check_round(keep, ranges, stats, st, unk)
    with pytest.raises(ctmr.CtmrError, match="group of its own"):
        g.map_batch("owner", shards)
    assert g.total_count() == int(unk.sum())
    g.close()
    for e in engines:
        e.close()
    # no filter configured: refused through the same row
    engines = [make_engine(issuers) for _ in range(world)]
    g = Group.local(engines)
    with pytest.raises(ctmr.CtmrError, match="bloom_config"):
        g.map_batch("bloom", shards)
    assert g.total_count() == 0
    stats = g.map_batch("owner", shards)                                                          # the refused call did not fix the mode
    assert sum(s.n_new for s in stats) == int(unk.sum())
    g.close()
    for e in engines:
        e.close()
