"""The host half of the stream's write-back (ct_mapreduce_amd/host/host_writeback.cpp → libctmr_host.so): records + PEM
bytes in, the reference's LocalDiskBackend layout out (storage/localdiskbackend.go:188-199: root/<expDate>/<issuerID>/
<serialID>, no suffix; MarkDirty relative to the CURRENT directory, :89-91) — or nothing at all through the NoopBackend.
No GPU: the records are made by hand."""
import base64
import os

import numpy as np

from ct_mapreduce_amd.host_writeback import HostWriter
from oracle import oracle as orc

REC = np.dtype([("status", "u1"), ("flags", "u1"), ("serial_len", "<u2"), ("exp_hour", "<i4"), ("issuer_idx", "<u4"),
                ("serial", "u1", 20)])
IDS = ["qBAK5qoZQNC2Y7sxzUZhQuu9vVGHExuS2TgYmHgy64k=", "VCIlmPM9NkgFQtrs4Oa5TeFcDu6MWRTKSNdePEhOgD8="]


def chunk(n, seed=1):
    rng = np.random.default_rng(seed)
    recs = np.zeros(n, REC)
    pems, serials = [], []
    for k in range(n):
        sl = int(rng.integers(1, 21))
        s = bytes(rng.integers(0, 256, sl, dtype=np.uint8))
        serials.append(s)
        recs[k]["serial_len"] = sl
        recs[k]["serial"][:sl] = np.frombuffer(s, np.uint8)
        recs[k]["exp_hour"] = 490000 + int(rng.integers(0, 50))
        recs[k]["issuer_idx"] = k % 2
        recs[k]["flags"] = 2
        pems.append(orc.pem_encode(bytes(rng.integers(0, 256, int(rng.integers(300, 900)), dtype=np.uint8))))
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum([len(p) for p in pems])
    blob = np.frombuffer(b"".join(pems), np.uint8).copy()
    return recs, blob, off, pems, serials


def test_local_disk_layout_and_bytes(tmp_path):
    recs, blob, off, pems, serials = chunk(500)
    root = tmp_path / "certs"
    w = HostWriter(str(root), IDS, threads=4)
    j1 = w.submit(blob.ctypes.data, off.ctypes.data, recs.ctypes.data, 250)          # two chunks in flight
    off2 = off[250:].copy()
    j2 = w.submit(blob.ctypes.data, off2.ctypes.data, recs[250:].ctypes.data, 250)
    f1, b1, s1, _ = w.wait(j1)
    f2, b2, s2, _ = w.wait(j2)
    assert (f1, f2, s1, s2) == (250, 250, 0, 0) and b1 + b2 == len(blob)
    for k in range(500):
        path = root / orc.exp_date_id(int(recs[k]["exp_hour"])) / IDS[k % 2] / base64.urlsafe_b64encode(serials[k]).decode()
        assert path.read_bytes() == pems[k], k
    n_files = sum(len(fs) for _, _, fs in os.walk(root))
    assert n_files == len({(int(recs[k]["exp_hour"]), k % 2, serials[k]) for k in range(500)})
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        w.mark_dirty([20440, 20441])                  # FilesystemDatabase.markDirty → backend.MarkDirty("2025-12-18")
    finally:
        os.chdir(cwd)
    assert (tmp_path / orc.day_id(20440 * 86400) / "dirty").read_bytes() == b"\x00"
    assert (tmp_path / orc.day_id(20441 * 86400) / "dirty").exists()
    w.close()


def test_long_serials_are_left_to_the_host_parse_and_noop_writes_nothing(tmp_path):
    recs, blob, off, pems, serials = chunk(40, seed=2)
    recs[7]["serial_len"] = 33                        # the record carries 20 octets only: not enough for the file name
    w = HostWriter(str(tmp_path / "c"), IDS, threads=2)
    f, b, s, _ = w.wait(w.submit(blob.ctypes.data, off.ctypes.data, recs.ctypes.data, 40))
    assert (f, s) == (39, 1)
    w.close()
    n = HostWriter(None, IDS, threads=2)              # storage.NoopBackend
    f, b, s, _ = n.wait(n.submit(blob.ctypes.data, off.ctypes.data, recs.ctypes.data, 40))
    assert (f, s) == (39, 1) and b == len(blob) - (int(off[8]) - int(off[7]))
    n.mark_dirty([1, 2, 3])
    n.close()
    assert not (tmp_path / "1970-01-02").exists()
