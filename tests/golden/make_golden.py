#!/usr/bin/env python3
"""Regenerates / verifies the fixtures in this directory.

  python tests/golden/make_golden.py            # verify (what the CPU test suite does through test_golden_files)
  python tests/golden/make_golden.py --write    # rewrite them (needs /root/reference for the PEM literals)

1. The three PEM certificates are TEST DATA copied out of the reference's own unit tests
   (storage/types_test.go:21-39, storage/filesystemdatabase_test.go:17-33,35-64): the Go string literals
   kLeadingZeroes / kEmptySPKI / kRealSPKI, byte for byte.
2. synth_small.json freezes a small synthetic batch: the generator's bytes (SHA-256 of the packed payload, of the
   raw get-entries blob and of the issuer certificates) and what the oracle says about every entry — so that neither
   the benchmark workload nor the oracle can drift unnoticed."""
import hashlib
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
PEMS = {"kLeadingZeroes": "storage/types_test.go", "kEmptySPKI": "storage/filesystemdatabase_test.go",
        "kRealSPKI": "storage/filesystemdatabase_test.go"}


def pem_from_reference(name, path):
    src = open(os.path.join(REF, path)).read()
    m = re.search(name + r"\s*=\s*`(-----BEGIN CERTIFICATE-----.*?-----END CERTIFICATE-----)`", src, re.S)
    return m.group(1).strip() + "\n"


def synth_small():
    import numpy as np
    from ct_mapreduce_amd import synth
    from oracle import oracle as orc
    cfg = synth.config(seed=20260921, n_issuers=4, dup_permille=200, ca_permille=100, expired_permille=100)
    n = 96
    b, raw, iss = synth.host_batch(cfg, 0, n), synth.host_entries(cfg, 0, n), synth.issuers(cfg)
    filt = b"Synth Issuer 00"
    o = orc.Engine(filt, False, synth.BASE_TIME)
    io = np.zeros(len(iss) + 1, np.uint64)
    io[1:] = np.cumsum([len(x) for x in iss])
    st, unk, eh = o.batch(b.payload, b.offsets, b.issuer_idx, np.frombuffer(b"".join(iss), np.uint8), io)
    return {"config": {"seed": 20260921, "n_issuers": 4, "dup_permille": 200, "ca_permille": 100,
                       "expired_permille": 100, "n": n, "filter": filt.decode(), "now": synth.BASE_TIME},
            "payload_sha256": hashlib.sha256(b.payload[:int(b.offsets[-1])].tobytes()).hexdigest(),
            "raw_blob_sha256": hashlib.sha256(raw.blob[:int(raw.bounds[-1])].tobytes()).hexdigest(),
            "issuers_sha256": hashlib.sha256(b"".join(iss)).hexdigest(),
            "issuer_idx": [int(x) for x in b.issuer_idx], "entry_type": [int(x) for x in b.entry_type],
            "status": [int(x) for x in st], "was_unknown": [int(x) for x in unk], "exp_hour": [int(x) for x in eh],
            "keys": [k.decode() for k in o.keys()], "total_count": o.total_count()}


def main():
    write = "--write" in sys.argv
    bad = 0
    for name, path in PEMS.items():
        f = os.path.join(HERE, name + ".pem")
        if os.path.isdir(REF):
            want = pem_from_reference(name, path)
            if write:
                open(f, "w").write(want)
            elif open(f).read().strip() != want.strip():
                print("MISMATCH", name)
                bad += 1
    f = os.path.join(HERE, "synth_small.json")
    want = synth_small()
    if write:
        json.dump(want, open(f, "w"), indent=1)
    elif json.load(open(f)) != want:
        print("MISMATCH synth_small.json")
        bad += 1
    print("golden fixtures", "written" if write else ("ok" if not bad else "DIFFER"))
    return bad


if __name__ == "__main__":
    sys.exit(main())
