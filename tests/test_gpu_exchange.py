"""-m gpu: cross-GPU global dedup through the NATIVE group layer (ctmr_group_*, csrc/engine/group.inc) on one GPU:
several engines on the one reachable device form a local group — the same phase drivers and kernels the RCCL transport
runs, with device-to-device copies where RCCL would send — and every shard's records must equal the oracle's view of
that slice of the WHOLE stream; world = 1 equals the plain local path; an RCCL group of world 1 exercises the
librccl loading, communicator and collectives that exist with one rank."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth, _native as N
from ct_mapreduce_amd.distributed import Group, shard, shard_range, decode_synchronised
from ct_mapreduce_amd.engine import RECORD_DTYPE
from tests.gpu_common import run_oracle
from oracle import oracle as orc

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


def dev_shard(t, n, order_base=0):
    pay, off, iss, et, rec, new = t
    return shard(pay.data_ptr(), off.data_ptr(), iss.data_ptr(), et.data_ptr(), n, rec.data_ptr(), new.data_ptr(),
                 order_base=order_base)


def make_engine(issuers, **kw):
    kw.setdefault("table_slots", 1 << 16)
    kw.setdefault("pair_slots", 1 << 12)
    e = ctmr.Engine(device=0, **kw)
    e.add_issuers(issuers)
    e.set_filter(FILT, False, NOW)
    return e


def check_shards_against_oracle(keep, stats, ranges, st, unk):
    for r, (lo, hi) in enumerate(ranges):
        rec = keep[r][4].cpu().numpy().view(RECORD_DTYPE)
        assert (rec["status"] == st[lo:hi]).all()
        assert (((rec["flags"] & 2) != 0) == (unk[lo:hi] != 0)).all()
        assert stats[r].n_new == int(unk[lo:hi].sum())
        new = keep[r][5][:stats[r].n_new].cpu().numpy()
        assert (new == np.nonzero(unk[lo:hi])[0]).all()
        for k in range(7):
            assert stats[r].by_status[k] == int((st[lo:hi] == k).sum())


@pytest.mark.parametrize("mode", ["owner", "bloom"])
@pytest.mark.parametrize("world", [1, 2, 3, 4])
def test_global_dedup_matches_oracle_over_the_whole_stream(world, mode):
    cfg = synth.config(seed=55, n_issuers=16, dup_permille=300, ca_permille=30, expired_permille=30)
    n_total = 6000
    issuers = synth.issuers(cfg)
    whole = synth.host_batch(cfg, 0, n_total)
    o, st, unk, eh = run_oracle(whole, issuers, FILT, False, NOW)
    assert 0 < unk.sum() < (st == 0).sum()                       # duplicates exist, also across shards
    engines = [make_engine(issuers) for _ in range(world)]
    g = Group.local(engines)
    if mode == "bloom":
        g.bloom_config(1 << 16)
    keep, shards, ranges = [], [], []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        t = to_dev(synth.host_batch(cfg, lo, hi - lo))
        keep.append(t)
        shards.append(dev_shard(t, hi - lo, order_base=lo))
        ranges.append((lo, hi))
    stats = g.map_batch(mode, shards)
    check_shards_against_oracle(keep, stats, ranges, st, unk)
    info = g.info()
    assert info.world == world and info.n_local == world and info.transport == N.TRANSPORT_LOCAL
    if world > 1:
        assert info.keys_sent == info.keys_received > 0
    # the ranks' per-issuer counters sum to the global unique counts (the group's all-reduce); every key counted once
    total = g.issuer_counts(len(issuers))
    for k in range(len(issuers)):
        assert int(total[k]) == o.issuer_count(engines[0].issuer_id(k))
    assert g.total_count() == o.total_count() == sum(e.total_count() for e in engines)
    allkeys = sorted(sum((e.keys(b"serials::*") for e in engines), []))
    assert sorted(set(allkeys)) == [k for k in o.keys() if k.startswith(b"serials::")]
    # replay: nothing is new anywhere
    stats2 = g.map_batch(mode, shards)
    assert all(s.n_new == 0 for s in stats2)
    assert g.total_count() == o.total_count()
    g.close()
    for e in engines:
        e.close()


@pytest.mark.parametrize("mode", ["owner", "bloom"])
def test_four_rounds_with_duplicates_across_rounds_and_ranks(mode):
    world, rounds, per = 3, 4, 1500
    cfg = synth.config(seed=58, n_issuers=8, dup_permille=350, ca_permille=20, expired_permille=20)
    issuers = synth.issuers(cfg)
    engines = [make_engine(issuers) for _ in range(world)]
    g = Group.local(engines)
    if mode == "bloom":
        g.bloom_config(1 << 12)                                  # tiny filter: false positives dominate, results stay exact
    o = None
    for rnd in range(rounds):
        base = rnd * per * world
        keep, shards, want = [], [], []
        for r in range(world):
            lo, hi = shard_range(per * world, r, world)
            if rnd == 2 and r == 1:
                hi = lo                                          # an EMPTY shard in one round
            b = synth.host_batch(cfg, base + lo, hi - lo)
            o, st, unk, _ = run_oracle(b, issuers, FILT, False, NOW, engine=o)   # the stream in log order, shard by shard
            want.append((st, unk))
            t = to_dev(b)
            keep.append(t)
            shards.append(dev_shard(t, hi - lo, order_base=base + lo))
        stats = g.map_batch(mode, shards)
        for r in range(world):
            st, unk = want[r]
            rec = keep[r][4].cpu().numpy().view(RECORD_DTYPE)[:len(st)]
            assert (rec["status"] == st).all(), (rnd, r)
            assert (((rec["flags"] & 2) != 0) == (unk != 0)).all(), (rnd, r)
            assert stats[r].n_new == int(unk.sum())
            assert (keep[r][5][:stats[r].n_new].cpu().numpy() == np.nonzero(unk)[0]).all()
    assert g.total_count() == o.total_count()
    total = g.issuer_counts(len(issuers))
    for k in range(len(issuers)):
        assert int(total[k]) == o.issuer_count(engines[0].issuer_id(k))
    # expiry sweep over every rank: the group total drops to zero (SHADOW members of the Bloom mode are not counted twice)
    removed = sum(e.expire_sweep(NOW + 400 * 86400) for e in engines)
    assert removed == o.total_count() and g.total_count() == 0
    g.close()
    for e in engines:
        e.close()


def test_global_path_equals_local_path_on_one_gpu():
    cfg = synth.config(seed=56, n_issuers=4, dup_permille=200)
    issuers = synth.issuers(cfg)
    b = synth.host_batch(cfg, 0, 5000)
    t = to_dev(b)
    pay, off, iss, et, rec, new = t
    e1, e2, e3 = make_engine(issuers), make_engine(issuers), make_engine(issuers)
    st_local = e1.map_batch_device(pay.data_ptr(), off.data_ptr(), iss.data_ptr(), et.data_ptr(), b.n,
                                   rec.data_ptr(), new.data_ptr())
    rec_local = rec.cpu().numpy().copy()
    new_local = new[:st_local.n_new].cpu().numpy().copy()
    for eng, mode in ((e2, "owner"), (e3, "local")):
        rec.zero_()
        g = Group.local([eng])
        st_glob = g.map_batch(mode, [dev_shard(t, b.n)])[0]
        assert (rec.cpu().numpy() == rec_local).all()
        assert (new[:st_glob.n_new].cpu().numpy() == new_local).all()
        assert st_glob.n_new == st_local.n_new and list(st_glob.by_status) == list(st_local.by_status)
        assert (e1.issuer_counts() == eng.issuer_counts()).all()
        assert (g.issuer_counts(len(issuers)) == e1.issuer_counts()[:len(issuers)]).all()
        g.close()
    e1.close(); e2.close(); e3.close()


@pytest.mark.parametrize("mode", ["local", "owner", "bloom"])
def test_rccl_transport_with_one_rank(mode):
    """The RCCL transport (librccl loaded on demand, ncclCommInitRank, ncclAllGather of the counts / filters,
    ncclAllReduce of the per-issuer counts) with world = 1 — all that can run where one GPU is reachable; N > 1 over
    RCCL stays unmeasured until a multi-GPU node exists."""
    cfg = synth.config(seed=59, n_issuers=6, dup_permille=250)
    issuers = synth.issuers(cfg)
    b = synth.host_batch(cfg, 0, 4000)
    t = to_dev(b)
    e = make_engine(issuers)
    gid = Group.unique_id()
    assert len(gid) == N.GROUP_ID_BYTES and any(gid)
    g = Group.rccl(e, gid, 0, 1)
    if mode == "bloom":
        g.bloom_config(1 << 16)
    stats = g.map_batch(mode, [dev_shard(t, b.n)])[0]
    o, st, unk, eh = run_oracle(b, issuers, FILT, False, NOW)
    rec = t[4].cpu().numpy().view(RECORD_DTYPE)
    assert (rec["status"] == st).all() and (((rec["flags"] & 2) != 0) == (unk != 0)).all()
    assert stats.n_new == int(unk.sum()) == g.total_count()
    info = g.info()
    assert info.transport == N.TRANSPORT_RCCL and info.world == 1
    counts = g.issuer_counts(len(issuers))
    for k in range(len(issuers)):
        assert int(counts[k]) == o.issuer_count(e.issuer_id(k))
    assert (g.all_reduce_u64([5, 7]) == [5, 7]).all() and (g.all_reduce_u64([9], op_max=True) == [9]).all()
    g.barrier()
    g.close()
    e.close()


def test_raw_shards_with_synchronised_issuer_registration():
    """Global dedup fed by RAW get-entries shards: two ranks see the issuers in different orders; with
    auto-registration off and the pending lists merged in an agreed order their tables stay index-identical, and both
    exchanges over entry views reproduce the single-stream oracle."""
    from oracle import oracle as orc
    world = 2
    cfg = synth.config(seed=57, n_issuers=24, dup_permille=300, ca_permille=30, expired_permille=30)
    n_total = 5000
    whole = synth.host_entries(cfg, 0, n_total)
    o = orc.Engine(FILT, False, NOW)
    st, unk, eh, ts = o.raw_batch(whole.blob, whole.bounds)
    assert 0 < unk.sum() < (st == 0).sum()
    for mode in ("owner", "bloom"):
        engines = []
        for _ in range(world):
            e = ctmr.Engine(device=0, table_slots=1 << 16, pair_slots=1 << 12)
            e.set_filter(FILT, False, NOW)
            e.set_issuer_autoregister(False)
            engines.append(e)
        g = Group.local(engines)
        if mode == "bloom":
            g.bloom_config(1 << 16)
        keep, calls, ranges, shards = [], [], [], []
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
            shards.append(shard(d_blob.data_ptr(), t["start"].data_ptr(), t["iss"].data_ptr(), t["et"].data_ptr(), n,
                                rec.data_ptr(), new.data_ptr(), order_base=lo, d_ends=t["end"].data_ptr(),
                                blob_bytes=int(raw.bounds[-1])))
        with pytest.raises(ctmr.CtmrError) as ei:                   # nothing is registered silently
            calls[0]()
        assert ei.value.code == N.E_NOTFOUND and len(engines[0].pending_issuers()) > 0
        decode_synchronised(engines, calls)
        assert engines[0].issuer_count() == engines[1].issuer_count() > 0
        for k in range(engines[0].issuer_count()):                  # index-identical issuer tables
            assert engines[0].issuer_id(k) == engines[1].issuer_id(k)
        stats = g.map_batch(mode, shards)
        for r, (lo, hi) in enumerate(ranges):
            rec = keep[r][4].cpu().numpy().view(RECORD_DTYPE)
            assert (rec["status"] == st[lo:hi]).all(), mode
            assert (((rec["flags"] & 2) != 0) == (unk[lo:hi] != 0)).all(), mode
            assert stats[r].n_new == int(unk[lo:hi].sum())
        assert g.total_count() == o.total_count()
        g.close()
        for e in engines:
            e.close()


@pytest.mark.parametrize("mode", ["owner", "bloom"])
@pytest.mark.parametrize("world", [2, 4])
def test_serials_of_every_length_across_ranks(world, mode):
    """Serial numbers of 1..45 octets, duplicated across shards.  The owner-computes exchange sends keys with serials of
    up to 20 octets as 32-byte records and the rare 21..40-octet ones as 64-byte records on a path of their own; serials
    beyond CTMR_MAX_SERIAL live in host-side sets, which the group settles between the ranks at the end of the round.
    Everything must dedup globally exactly like the single-stream oracle, whichever rank holds the key and whichever
    rank saw it first."""
    import random
    from ct_mapreduce_amd.engine import Batch
    from tests import der as D
    rng = random.Random(4242 + world)
    issuer = synth.issuer(synth.config(n_issuers=1), 0)
    name = D.name(D.rdn(3, b"Synth Issuer 000"))
    uniq = []
    for ln in list(range(1, 46)) * 6:                                # six keys of every length 1..45
        s = bytes([rng.randrange(1, 0x7f)] + [rng.randrange(256) for _ in range(ln - 1)])
        uniq.append(D.cert(serial=s, issuer=name, not_after=D.utctime("270101000000Z")))
    certs = list(uniq)
    certs += [uniq[rng.randrange(len(uniq))] for _ in range(len(uniq))]          # every key about once more, anywhere
    rng.shuffle(certs)
    whole = Batch.from_certs(certs, [0] * len(certs))
    whole.payload = np.concatenate([whole.payload, np.zeros(N.PAYLOAD_PAD, np.uint8)])
    o, st, unk, eh = run_oracle(whole, [issuer], b"", True, 0)
    assert (st == 0).all() and int(unk.sum()) == len(uniq)
    engines = []
    for _ in range(world):
        e = ctmr.Engine(device=0, table_slots=1 << 12, pair_slots=1 << 10)
        e.add_issuers([issuer])
        e.set_filter(b"", True, 0)
        engines.append(e)
    g = Group.local(engines)
    if mode == "bloom":
        g.bloom_config(1 << 14)
    keep, shards, ranges = [], [], []
    for r in range(world):
        lo, hi = shard_range(len(certs), r, world)
        b = Batch.from_certs(certs[lo:hi], [0] * (hi - lo))
        t = to_dev(b)
        keep.append(t)
        shards.append(dev_shard(t, hi - lo, order_base=lo))
        ranges.append((lo, hi))
    stats = g.map_batch(mode, shards)
    check_shards_against_oracle(keep, stats, ranges, st, unk)
    assert g.total_count() == o.total_count() == len(uniq)
    if mode == "owner":
        assert g.info().wire_bytes_sent > 0
    stats2 = g.map_batch(mode, shards)                              # replay: everything is known, wherever it lives
    assert all(s.n_new == 0 for s in stats2)
    g.close()
    for e in engines:
        e.close()


@pytest.mark.parametrize("mode", ["owner", "bloom"])
def test_long_serials_across_ranks_and_rounds(mode):
    """Serials beyond CTMR_MAX_SERIAL (41..60 octets; Go's parser takes any length): round 1 puts each key on some rank,
    round 2 presents the same keys to OTHER ranks together with fresh ones, twice within the round on different ranks —
    a member some rank held before is known everywhere, a fresh one goes to the lowest log index, every member is stored
    once (Σ total_count = the oracle's), and the NEW lists follow."""
    import random
    from ct_mapreduce_amd.engine import Batch
    from tests import der as D
    rng = random.Random(99)
    world = 3
    issuer = synth.issuer(synth.config(n_issuers=1), 0)
    name = D.name(D.rdn(3, b"Synth Issuer 000"))
    def cert(ln):
        s = bytes([rng.randrange(1, 0x7f)] + [rng.randrange(256) for _ in range(ln - 1)])
        return D.cert(serial=s, issuer=name, not_after=D.utctime("270101000000Z"))
    first = [cert(rng.choice((41, 44, 48, 60, 12, 30))) for _ in range(60)]
    fresh = [cert(rng.choice((41, 45, 50, 8))) for _ in range(40)]
    round1 = list(first)
    round2 = first[::-1] + fresh + fresh[::-1] + first[:10]          # old keys on other ranks, fresh ones twice
    rng.shuffle(round2)
    o = None
    engines = []
    for _ in range(world):
        e = ctmr.Engine(device=0, table_slots=1 << 12, pair_slots=1 << 10)
        e.add_issuers([issuer])
        e.set_filter(b"", True, 0)
        engines.append(e)
    g = Group.local(engines)
    if mode == "bloom":
        g.bloom_config(1 << 14)
    base = 0
    for rnd, certs in enumerate((round1, round2, round2)):
        whole = Batch.from_certs(certs, [0] * len(certs))
        whole.payload = np.concatenate([whole.payload, np.zeros(N.PAYLOAD_PAD, np.uint8)])
        o, st, unk, eh = run_oracle(whole, [issuer], b"", True, 0, engine=o)
        assert (st == 0).all() and int(unk.sum()) == (len(first), len(fresh), 0)[rnd]
        keep, shards, ranges = [], [], []
        for r in range(world):
            lo, hi = shard_range(len(certs), r, world)
            t = to_dev(Batch.from_certs(certs[lo:hi], [0] * (hi - lo)))
            keep.append(t)
            shards.append(dev_shard(t, hi - lo, order_base=base + lo))
            ranges.append((lo, hi))
        stats = g.map_batch(mode, shards)
        check_shards_against_oracle(keep, stats, ranges, st, unk)
        assert sum(int(s.n_host_set) for s in stats) > 0
        assert g.total_count() == o.total_count()
        base += len(certs)
    assert g.total_count() == len(first) + len(fresh)
    g.close()
    for e in engines:
        e.close()


def test_local_group_runs_its_ranks_concurrently():
    """A LOCAL group's ranks run their phases on host threads of their own (round 2: one after the other, each ending in
    a stream synchronisation — a single-process group over 8 devices mapped one GPU at a time).  Three engines on the one
    device: the map kernels of the three ranks are in flight together, so the round takes clearly less than three times
    one rank's round (each rank's kernel alone cannot fill the device at this size)."""
    import time
    cfg = synth.config(seed=61, n_issuers=16, dup_permille=100)
    issuers = synth.issuers(cfg)
    n = 4096                                                       # 64 workgroups per rank: a quarter of the CUs
    def one_round(world, reps=30):
        engines = [make_engine(issuers, table_slots=1 << 20) for _ in range(world)]
        g = Group.local(engines)
        keep = [to_dev(synth.host_batch(cfg, r * n, n)) for r in range(world)]
        shards = [dev_shard(keep[r], n, order_base=r * n) for r in range(world)]
        g.map_batch("local", shards)
        best = 1e9
        for _ in range(reps):
            for e in engines:
                e.reset_known()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.map_batch("local", shards)
            best = min(best, time.perf_counter() - t0)
        g.close()
        for e in engines:
            e.close()
        return best
    t1, t3 = one_round(1), one_round(3)
    assert t3 < 2.2 * t1, (t1, t3)


@pytest.mark.parametrize("world,chunks", [(2, 2), (3, 4), (4, 7), (2, 64)])
def test_owner_rounds_mapped_in_chunks_give_the_unchunked_answers(world, chunks):
    """ctmr_group_set_chunks: every shard is mapped in K pieces, chunk c's key records travel on the transfer streams while
    chunk c + 1 is walked, the owner-side insert runs once over everything.  Inside one round the order in which keys
    reach their owner does not matter: records, NEW lists, statistics, counts and sets equal the oracle's over three
    rounds — shards of uneven size (one smaller than a chunk, one empty), duplicates within chunks, across chunks, across
    ranks and across rounds, EC keys whose records are withdrawn after the map had staged them."""
    cfg = synth.config(seed=71, n_issuers=12, dup_permille=300, ca_permille=20, expired_permille=20, profile=1)
    issuers = synth.issuers(cfg)
    engines = [make_engine(issuers) for _ in range(world)]
    g = Group.local(engines)
    g.set_chunks(chunks)
    o = None
    rng = np.random.default_rng(chunks * 10 + world)
    sizes_by_round = [[5000, 300, 0, 2600], [1024, 2048, 3000, 1], [4100, 4100, 10, 700]]
    base = 0
    for rnd, sizes in enumerate(sizes_by_round):
        keep, shards, want = [], [], []
        lo = base
        for r in range(world):
            n = sizes[r]
            b = synth.host_batch(cfg, lo, n)
            if n > 40:                                            # keys damaged at their end: EC points off their curve
                certs = [b.cert(i) for i in range(n)]
                for j in rng.choice(n, size=8, replace=False):
                    c = orc.parse_cert(certs[j], strict_spki=False)
                    if c.ok:
                        end = c.spki_off + c.spki_len
                        certs[j] = certs[j][:end - 3] + b"\x00\x01\x02" + certs[j][end:]
                b = ctmr.Batch.from_certs(certs, b.issuer_idx, b.entry_type)
            o, st, unk, _ = run_oracle(b, issuers, FILT, False, NOW, engine=o)
            want.append((st, unk))
            t = to_dev(b)
            keep.append(t)
            shards.append(dev_shard(t, n, order_base=lo))
            lo += n
        base = lo
        stats = g.map_batch("owner", shards)
        for r in range(world):
            st, unk = want[r]
            rec = keep[r][4].cpu().numpy().view(RECORD_DTYPE)[:len(st)]
            assert (rec["status"] == st).all(), (rnd, r)
            assert (((rec["flags"] & 2) != 0) == (unk != 0)).all(), (rnd, r, np.nonzero(((rec["flags"] & 2) != 0) != (unk != 0))[0][:8])
            assert stats[r].n_new == int(unk.sum())
            assert (keep[r][5][:stats[r].n_new].cpu().numpy() == np.nonzero(unk)[0]).all()
    assert g.total_count() == o.total_count() == sum(e.total_count() for e in engines)
    total = g.issuer_counts(len(issuers))
    for k in range(len(issuers)):
        assert int(total[k]) == o.issuer_count(engines[0].issuer_id(k))
    allkeys = sorted(sum((e.keys(b"serials::*") for e in engines), []))
    assert sorted(set(allkeys)) == [k for k in o.keys() if k.startswith(b"serials::")]
    key = sorted(set(allkeys))[len(set(allkeys)) // 2]            # the ranks hold disjoint parts of a set
    members = sum((e.set_list(key) for e in engines), [])
    assert len(members) == len(set(members)) and sorted(members) == sorted(o.members(key))
    g.close()
    for e in engines:
        e.close()
