"""-m gpu: 32-bit TAG collisions in the known-certificate table.  Two different keys with the same tag and the same
home slot make pass 1 remember a candidate slot whose key is NOT theirs; pass 2 (k_insert2) must notice and fall back
to the fully synchronised upsert.  At 2^-32 per probe this never shows up in random data, so the colliding serials are
searched for here with a numpy port of key_meta/key_hash (ct_mapreduce_amd/csrc/ctmr_dev.h) and fed through the
real path; everything must still equal the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402,F401

import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth, _native as N
from ct_mapreduce_amd.engine import Batch
from oracle import oracle as orc
from tests import der as D
from tests.gpu_common import run_oracle, assert_records_equal, assert_state_equal

U = np.uint64


def mixk(z):
    z = (z ^ (z >> U(30))) * U(0xbf58476d1ce4e5b9)
    z = (z ^ (z >> U(27))) * U(0x94d049bb133111eb)
    return z ^ (z >> U(31))


def rotl(x, r):
    return (x << U(r)) | (x >> U(64 - r))


def key_hash(meta, s0):
    """key_hash for serials of at most 8 octets (s1..s4 = 0)."""
    h = mixk(meta + U(0x9e3779b97f4a7c15))
    h = mixk(h ^ s0 ^ U(0x3c6ef372fe94f82b))
    return mixk(h)          # s2 = s3 = s4 = 0


def test_tag_collisions_take_the_synchronised_path():
    issuer = synth.issuer(synth.config(n_issuers=1), 0)
    not_after = D.utctime("270101000000Z")
    exp_hour = orc.exp_hour(orc.parse_cert(D.cert(not_after=not_after)).not_after)
    rng = np.random.default_rng(20260923)
    n = 1 << 22
    with np.errstate(over="ignore"):
        serials = rng.integers(0, 1 << 63, n, dtype=np.int64).astype(np.uint64) | U(1)        # 8 octets, little-endian in s0
        serials &= ~(U(0x80))                                                                # first octet < 0x80: minimal, positive
        serials |= U(0x01)
        meta = U((1 << 63) | (8 << 56) | (0 << 32) | (exp_hour & 0xffffffff))
        h = key_hash(meta, serials)
    slots = 256
    sig = (h >> U(32)) << U(8) | (h & U(slots - 1))             # tag + home slot
    order = np.argsort(sig, kind="stable")
    same = np.nonzero(sig[order][1:] == sig[order][:-1])[0]
    assert len(same) >= 3, "search space too small"
    pairs = [(int(order[k]), int(order[k + 1])) for k in same[:12]]
    certs = []
    for a, b in pairs:
        for k in (a, b):
            certs.append(D.cert(serial=int(serials[k]).to_bytes(8, "little"), not_after=not_after,
                                issuer=D.name(D.rdn(3, b"Synth Issuer 000"))))
    for k in range(100):                                         # filler: a well-loaded 256-slot table
        certs.append(D.cert(serial=b"\x01" + int(k).to_bytes(2, "big"), not_after=not_after,
                            issuer=D.name(D.rdn(3, b"Synth Issuer 000"))))
    certs += certs[:10]                                          # and real duplicates of colliding keys
    # the numpy port really is the device's hash: the owner-computes exchange partitions keys by a second hash of key_hash
    dev = torch.device("cuda:0")
    eng = ctmr.Engine(device=0, table_slots=1 << 12, pair_slots=1 << 10)
    eng.add_issuers([issuer])
    eng.set_filter(b"", True, 0)
    b0 = Batch.from_certs(certs[:len(pairs) * 2], [0] * (len(pairs) * 2))
    d_pay = torch.from_numpy(np.concatenate([b0.payload, np.zeros(64, np.uint8)])).to(dev)
    d_off = torch.from_numpy(b0.offsets.astype(np.int64)).to(dev)
    d_iss = torch.zeros(b0.n, dtype=torch.int32, device=dev)
    d_rec = torch.zeros(b0.n * 32, dtype=torch.uint8, device=dev)
    d_keys = torch.zeros(b0.n * 32, dtype=torch.uint8, device=dev)
    from ct_mapreduce_amd.distributed import shard as make_shard
    counts, n_long = eng.xchg_map(make_shard(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), 0, b0.n, d_rec.data_ptr()),
                                  16, 15, 1000)
    assert n_long == 0 and sum(counts) >= b0.n // 2 and counts[15] == 0     # rank 15's own keys are inserted, not exported
    eng.xchg_keys(16, d_keys.data_ptr())
    recs = np.frombuffer(d_keys.cpu().numpy().tobytes(), dtype=np.dtype([("meta", "<u8"), ("s0", "<u8"), ("s1", "<u8"),
                                                                         ("s2", "<u4"), ("ord", "<u4")]))[:sum(counts)]
    bounds = np.cumsum([0] + counts)
    with np.errstate(over="ignore"):
        for pos, r in enumerate(recs):
            assert r["meta"] == meta and r["s1"] == 0 and r["s2"] == 0
            hh = key_hash(U(r["meta"]), U(r["s0"]))
            owner = int((int(mixk(hh ^ U(0x5bd1e995))) >> 32) * 16 >> 32)       # keyrec.h: key_owner_h
            assert bounds[owner] <= pos < bounds[owner + 1] and owner != 15    # it lies in its owner's partition
            src = int(r["ord"]) - 1000                                         # order in the round = ord_base + index
            k = pairs[src // 2][src % 2]
            assert hh == h[k]                                    # and the collision search used the same values
    eng.close()
    for one_batch in (True, False):
        eng = ctmr.Engine(device=0, table_slots=slots, pair_slots=1 << 10)
        eng.add_issuers([issuer])
        eng.set_filter(b"", True, 0)
        o = orc.Engine(b"", True, 0)
        chunks = [certs] if one_batch else [certs[0::2], certs[1::2]]   # second form: partners arrive in different batches
        for chunk in chunks:
            batch = Batch.from_certs(chunk, [0] * len(chunk))
            batch.payload = np.concatenate([batch.payload, np.zeros(N.PAYLOAD_PAD, np.uint8)])
            res = eng.map_batch(batch)
            _, st, unk, eh = run_oracle(batch, [issuer], engine=o)
            assert (st == 0).all()
            assert_records_equal(res, batch, st, unk, eh)
            assert_state_equal(eng, o, 1)
        # point operations on the colliding keys (the same synchronised upsert)
        key = sorted(eng.keys(b"serials::*"))[0]
        for a, b in pairs[:3]:
            for k in (a, b):
                m = int(serials[k]).to_bytes(8, "little")
                assert eng.set_contains(key, m) and not eng.set_insert(key, m)
        assert eng.total_count() == o.total_count() == len(pairs) * 2 + 100
        eng.close()
