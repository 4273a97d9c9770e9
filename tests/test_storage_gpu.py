"""-m gpu: the reference's cache-facing test suites against the HBM-backed GpuRemoteCache, and
FilesystemDatabase.StoreBatch (batched insertCTWorker + Store) end to end."""
import calendar
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402,F401

import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth
from tests import storage_mirror as S
from oracle import oracle as orc
from tests.test_storage_cpu import (known_certificates_suite, duplicate_crls_suite, accumulate_suite,
                                    issuer_and_dates_suite, log_state_suite, utc)


@pytest.fixture
def engine():
    e = ctmr.Engine(device=0, table_slots=1 << 16, pair_slots=1 << 12)
    yield e
    e.close()


def test_Unknown_and_Known_gpu_cache(engine):
    known_certificates_suite(S.GpuRemoteCache(engine))          # "test issuer": host-side store
    # the same suite on a key that lives in the HBM table
    cfg = synth.config(n_issuers=1)
    engine.add_issuers(synth.issuers(cfg))
    cache = S.GpuRemoteCache(engine)
    kc = S.KnownCertificates(S.ExpDate.Parse("2029-01-30-00"), S.Issuer.FromString(engine.issuer_id(0)), cache)
    for h in ("01", "02", "03", "04"):
        cache.SetInsert(kc.serialId(), S.Serial.FromHex(h).BinaryString())
    for h in ("01", "02", "03", "04"):
        assert kc.WasUnknown(S.Serial.FromHex(h)) is False
    assert kc.WasUnknown(S.Serial.FromHex("05")) is True and kc.WasUnknown(S.Serial.FromHex("05")) is False
    assert sorted(kc.Known()) == [S.Serial.FromHex(h) for h in ("01", "02", "03", "04", "05")]
    assert kc.Count() == 5 and int(engine.issuer_counts()[0]) == 5


def test_DuplicateCRLs_and_Accumulate_gpu_cache(engine):
    duplicate_crls_suite(S.GpuRemoteCache(engine))
    engine2 = ctmr.Engine(device=0, table_slots=1 << 10, pair_slots=1 << 10)
    accumulate_suite(S.GpuRemoteCache(engine2))
    engine2.close()


def test_IssuerAndDates_and_LogState_gpu_cache(engine):
    cache = S.GpuRemoteCache(engine)
    issuer_and_dates_suite(S.FilesystemDatabase(S.MockBackend(), cache))
    log_state_suite(cache, S.FilesystemDatabase(S.MockBackend(), cache))


@pytest.mark.parametrize("mode", ["host_meta", "device_meta", "raw_entries"])
def test_StoreBatch_end_to_end(tmp_path, mode):
    """Batched insertCTWorker + FilesystemDatabase.Store: same sets, counts, metadata and files as
    feeding the entries one at a time through the oracle's restatement of the reference loop.
    host_meta: IssuerMetadata.Accumulate per new certificate on the host; device_meta: its memo on the GPU (N3);
    raw_entries: additionally fed with raw get-entries blobs (N2: decode + Chain[0] registration on the GPU)."""
    cfg = synth.config(seed=9, n_issuers=6, dup_permille=150, ca_permille=50, expired_permille=50)
    n = 1500
    batch = synth.host_batch(cfg, 0, n)
    issuers = synth.issuers(cfg)
    leafs = [batch.cert(i) for i in range(n)]
    chain0 = [issuers[int(k)] for k in batch.issuer_idx]
    chain0[17] = None                                            # len(Chain) < 1
    now = synth.BASE_TIME
    eng = ctmr.Engine(device=0, table_slots=1 << 14, pair_slots=1 << 12, collect_meta=mode != "host_meta")
    eng.set_filter(b"Synth Issuer 00", False, now)
    root = str(tmp_path / "certs")
    backend = S.LocalDiskBackend(0o644, root)
    db = S.FilesystemDatabase(backend, S.GpuRemoteCache(eng))
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        if mode == "raw_entries":
            from tests.test_entry_decode_cpu import x509_leaf, precert_leaf, chain, asn1cert
            from tests.test_walk_cpu import tbs_of
            from ct_mapreduce_amd.engine import RawEntries

            def raw_of(lo, hi):
                pairs = []
                for i in range(lo, hi):
                    ch = chain([chain0[i]] if chain0[i] is not None else [])
                    if batch.entry_type[i] == 0:
                        pairs.append((x509_leaf(leafs[i], ts=i), ch))
                    else:
                        # (the MerkleTreeLeaf of a precertificate entry carries its TBSCertificate: LogEntryFromLeaf parses it —
                        #  strict_leaf, part of the default profile since round 6)
                        pairs.append((precert_leaf(tbs_of(leafs[i]), ts=i), asn1cert(leafs[i]) + ch))
                r = RawEntries.from_pairs(pairs)
                r.blob = np.concatenate([r.blob, np.zeros(32, np.uint8)])
                return r
            r1, r2 = db.StoreRawBatch(raw_of(0, 700)), db.StoreRawBatch(raw_of(700, n))
        else:
            r1 = db.StoreBatch(leafs[:700], chain0[:700], batch.entry_type[:700])
            r2 = db.StoreBatch(leafs[700:], chain0[700:], batch.entry_type[700:])
    finally:
        os.chdir(cwd)
    # oracle, entry by entry (ct-fetch.go:191-235)
    o = orc.Engine(b"Synth Issuer 00", False, now)
    status, unknown = [], []
    for i in range(n):
        st, unk, eh = o.entry(leafs[i], chain0[i])
        status.append(st)
        unknown.append(unk)
    rec = np.concatenate([r1.records, r2.records])
    assert list(rec["status"]) == status
    assert list((rec["flags"] & 2) != 0) == unknown
    assert status[17] == orc.ST_NO_ISSUER
    # known-certificate sets and the statistics tool's numbers
    okeys = [k for k in o.keys() if k.startswith(b"serials::")]
    assert sorted(eng.keys(b"serials::*")) == okeys
    synth_ids = [orc.issuer_id(d[orc.parse_cert(d).spki_off:][:orc.parse_cert(d).spki_len]) for d in issuers]
    stats, total, total_crls = S.storage_statistics(db)
    assert total == o.total_count() == sum(unknown)
    for iid, (hours, serials, crls, dns) in stats.items():
        assert serials == o.issuer_count(iid)
        k = synth_ids.index(iid)
        assert dns == ["CN=Synth Issuer %03d,O=Synth CA Org,C=US" % k]
        assert crls == ["http://crl.synth-%03d.example/ca.crl" % k]
        assert hours == len([x for x in okeys if x.endswith(iid.encode())])
    # PEM write-back: one file per newly unknown certificate at root/expDate/issuer/serialID
    files = [os.path.join(dp, f) for dp, _, fs in os.walk(root) for f in fs]
    assert len(files) == sum(unknown)
    i = unknown.index(True)
    c = orc.parse_cert(leafs[i])
    serial = S.Serial(leafs[i][c.serial_off:c.serial_off + c.serial_len])
    path = os.path.join(root, orc.exp_date_id(orc.exp_hour(c.not_after)),
                        synth_ids[int(batch.issuer_idx[i])], serial.ID())
    assert open(path, "rb").read() == S.pem_encode(leafs[i])
    # dirty markers: one per NotAfter day of every stored entry (relative to the CWD, as the reference)
    days = {orc.day_id(orc.parse_cert(leafs[j]).not_after) for j in range(n) if status[j] == 0}
    assert {d for d in os.listdir(tmp_path) if os.path.exists(tmp_path / d / "dirty")} == days
    eng.close()


def test_redis_dump_of_the_gpu_sets_equals_the_oracle_and_restores(engine):
    """N4: dump the HBM-resident sets as a Redis protocol stream, load it into a mock cache and into a second
    engine: keys, members, cardinalities and expiry times are the oracle's."""
    import io
    cfg = synth.config(seed=91, n_issuers=6, dup_permille=200, ca_permille=20, expired_permille=20)
    issuers = synth.issuers(cfg)
    now = synth.BASE_TIME
    engine.add_issuers(issuers)
    engine.set_filter(b"", False, now)
    b = synth.host_batch(cfg, 0, 3000)
    engine.map_batch(b)
    o = orc.Engine(b"", False, now)
    for i in range(b.n):
        o.entry(b.cert(i), issuers[int(b.issuer_idx[i])])
    okeys = [k for k in o.keys() if k.startswith(b"serials::")]
    buf = io.BytesIO()
    n = S.redis_dump(S.GpuRemoteCache(engine), buf)
    assert n == {"keys": len(okeys), "members": o.total_count()}
    mock = S.MockRemoteCache()
    S.redis_load(mock, io.BytesIO(buf.getvalue()))
    assert sorted(mock.Data) == okeys
    for k in okeys:
        assert mock.Data[k] == sorted(o.members(k))
        assert mock.Expirations[k] == o.key_expiry(k)
    e2 = ctmr.Engine(device=0, table_slots=1 << 14, pair_slots=1 << 12)
    e2.add_issuers(issuers)
    res = S.redis_load(S.GpuRemoteCache(e2), io.BytesIO(buf.getvalue()))
    assert res["inserted"] == o.total_count()
    assert sorted(e2.keys(b"serials::*")) == okeys
    assert (e2.issuer_counts() == engine.issuer_counts()).all()
    for k in okeys[::7]:
        assert e2.set_list(k) == sorted(o.members(k))
    e2.close()


def test_set_insert_of_known_members_does_not_eat_arena_cells():
    """Advisor, round 4: every SetInsert took a fresh 64-byte key cell and a known member never gave it back — a host that
    re-reads a log of known certificates through RemoteCache.SetInsert grew the arena until hipMalloc failed."""
    import ct_mapreduce_amd as ctmr
    from ct_mapreduce_amd import synth
    cfg = synth.config(seed=3, n_issuers=2)
    eng = ctmr.Engine(device=0, table_slots=1 << 10, pair_slots=1 << 10)
    eng.add_issuers(synth.issuers(cfg))
    key = b"serials::2027-01-01-00::" + eng.issuer_id(0).encode()
    assert eng.set_insert(key, b"\x01\x02") and eng.set_insert(key, b"\x01\x03")
    used = eng.table_info().arena_used
    for _ in range(300):
        assert not eng.set_insert(key, b"\x01\x02")
    assert eng.table_info().arena_used == used and eng.set_cardinality(key) == 2
    assert eng.set_insert(key, b"\x01\x04") and eng.table_info().arena_used == used + 1
    eng.close()
