"""-m gpu: more issuers than the resolve kernels keep LDS bins for (RES_LDS_ISSUERS = 4096) — public CT logs have
thousands — so that the wave-aggregated global atomics of k_resolve (kernels/reduce.h) and k_keys_resolve
(kernels/exchange.h) run, under a uniform and a Zipf issuer popularity, against the oracle's per-issuer counts
(cmd/storage-statistics/storage-statistics.go:44-53)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth
from ct_mapreduce_amd.distributed import Group, shard_range
from ct_mapreduce_amd.engine import RECORD_DTYPE
from tests.gpu_common import run_oracle
from tests.test_gpu_exchange import to_dev, dev_shard

DEV = torch.device("cuda:0")
NOW = synth.BASE_TIME
N_ISSUERS = 6000
FILT = b"Synth Issuer"


def expected_counts(batch, unk, n_issuers):
    """Σ_expDate SCARD per issuer = the entries that were unknown, by issuer (distinct SPKIs: canonical = index)."""
    return np.bincount(batch.issuer_idx[unk != 0], minlength=n_issuers).astype(np.uint64)


@pytest.mark.parametrize("zipf", [0, 1], ids=["uniform", "zipf"])
@pytest.mark.parametrize("variant", [13, 15], ids=["separate_insert", "fused"])
def test_per_issuer_counts_with_6000_issuers(zipf, variant):
    cfg = synth.config(seed=77 + zipf, n_issuers=N_ISSUERS, zipf=zipf, dup_permille=150, ca_permille=10, expired_permille=10)
    issuers = synth.issuers(cfg)
    batch = synth.host_batch(cfg, 0, 50000)
    assert int((batch.issuer_idx >= 4096).sum()) > (300 if zipf else 10000)      # the wave-aggregated path is exercised
    eng = ctmr.Engine(device=0, table_slots=1 << 18, pair_slots=1 << 18, max_issuers=8192, map_variant=variant)
    assert eng.add_issuers(issuers) == 0
    eng.set_filter(FILT, False, NOW)
    res = eng.map_batch(batch)
    o, st, unk, eh = run_oracle(batch, issuers, FILT, False, NOW)
    assert (res.records["status"] == st).all()
    assert (((res.records["flags"] & 2) != 0) == (unk != 0)).all()
    want = expected_counts(batch, unk, N_ISSUERS)
    got = eng.issuer_counts()[:N_ISSUERS]
    assert (got == want).all(), np.nonzero(got != want)[0][:10]
    assert int(want[4096:].sum()) > 0 and eng.total_count() == o.total_count() == int(want.sum())
    for k in (0, 4095, 4096, 4097, 5000, N_ISSUERS - 1):                            # … and by the oracle's own definition
        assert int(got[k]) == o.issuer_count(eng.issuer_id(k))
    # a second, overlapping batch: counts keep accumulating only for keys that are new
    batch2 = synth.host_batch(cfg, 30000, 40000)
    res2 = eng.map_batch(batch2)
    _, st2, unk2, _ = run_oracle(batch2, issuers, engine=o)
    assert (((res2.records["flags"] & 2) != 0) == (unk2 != 0)).all()
    want2 = want + expected_counts(batch2, unk2, N_ISSUERS)
    assert (eng.issuer_counts()[:N_ISSUERS] == want2).all()
    eng.close()


@pytest.mark.parametrize("mode", ["owner", "bloom"])
def test_global_dedup_counts_with_6000_issuers(mode):
    world = 2
    cfg = synth.config(seed=79, n_issuers=N_ISSUERS, zipf=0, dup_permille=250, ca_permille=10, expired_permille=10)
    issuers = synth.issuers(cfg)
    n_total = 40000
    whole = synth.host_batch(cfg, 0, n_total)
    o, st, unk, eh = run_oracle(whole, issuers, FILT, False, NOW)
    engines = []
    for _ in range(world):
        e = ctmr.Engine(device=0, table_slots=1 << 17, pair_slots=1 << 17, max_issuers=8192)
        e.add_issuers(issuers)
        e.set_filter(FILT, False, NOW)
        engines.append(e)
    g = Group.local(engines)
    if mode == "bloom":
        g.bloom_config(1 << 20)
    shards, keep = [], []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        t = to_dev(synth.host_batch(cfg, lo, hi - lo))
        keep.append(t)
        shards.append(dev_shard(t, hi - lo, order_base=lo))
    stats = g.map_batch(mode, shards)
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        rec = keep[r][4].cpu().numpy().view(RECORD_DTYPE)
        assert (rec["status"] == st[lo:hi]).all()
        assert (((rec["flags"] & 2) != 0) == (unk[lo:hi] != 0)).all()
        assert stats[r].n_new == int(unk[lo:hi].sum())
    total = g.issuer_counts(N_ISSUERS)
    want = expected_counts(whole, unk, N_ISSUERS)
    assert (total == want).all(), np.nonzero(total != want)[0][:10]
    assert int(want[4096:].sum()) > 5000
    g.close()
    for e in engines:
        e.close()
