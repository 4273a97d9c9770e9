"""N1: GPU PEM write-back ≡ the oracle's restatement of pem.EncodeToMemory
(storage/filesystemdatabase.go:167-175,196-200), through the C ABI."""
import numpy as np
import pytest

import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def test_pem_new_matches_oracle_on_synthetic_batch():
    cfg = synth.config(seed=20260921 + 7, n_issuers=16, dup_permille=150, ca_permille=20, expired_permille=20)
    batch = synth.host_batch(cfg, 0, 5000)
    eng = ctmr.Engine(device=0, table_slots=1 << 16, pair_slots=1 << 14)
    eng.add_issuers(synth.issuers(cfg))
    eng.set_filter(b"", False, synth.BASE_TIME)
    res = eng.map_batch(batch)
    pems = eng.pem_new()
    assert len(pems) == len(res.new_idx) == res.stats.n_new > 0
    for pem, i in zip(pems, res.new_idx):
        assert pem == orc.pem_encode(bytes(batch.cert(int(i)))), int(i)
    # a second batch of pure duplicates: nothing new, nothing to encode
    res2 = eng.map_batch(batch)
    assert res2.stats.n_new == 0 and eng.pem_new() == []
    eng.close()


def test_pem_device_every_length_and_padding_case():
    """Arbitrary byte strings (the encoder does not care that they are not certificates): all lengths
    0..200 cover every quarter-line / padding / line-end combination; plus typical DER sizes."""
    import torch
    rng = np.random.default_rng(5)
    lens = list(range(0, 201)) + [717, 1297, 1523, 1536, 2000, 4095]
    blobs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in lens]
    offs = np.zeros(len(blobs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(b) for b in blobs])
    payload = np.frombuffer(b"".join(blobs) + bytes(64), np.uint8)
    idx = np.arange(len(blobs), dtype=np.uint64)[::-1].copy()      # any order
    dev = torch.device("cuda:0")
    d_pay = torch.from_numpy(payload.copy()).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_idx = torch.from_numpy(idx.astype(np.int64)).to(dev)
    d_po = torch.zeros(len(idx) + 1, dtype=torch.int64, device=dev)
    eng = ctmr.Engine(device=0, table_slots=1 << 10, pair_slots=1 << 10)
    total = eng.pem_encode_device(d_pay.data_ptr(), d_off.data_ptr(), d_idx.data_ptr(), len(idx), 0, 0,
                                  d_po.data_ptr())
    exp = [orc.pem_encode(blobs[int(i)]) for i in idx]
    assert total == sum(len(x) for x in exp)
    d_pem = torch.zeros(total, dtype=torch.uint8, device=dev)
    eng.pem_encode_device(d_pay.data_ptr(), d_off.data_ptr(), d_idx.data_ptr(), len(idx), d_pem.data_ptr(), total,
                          d_po.data_ptr())
    po = d_po.cpu().numpy()
    raw = d_pem.cpu().numpy().tobytes()
    for k in range(len(idx)):
        assert raw[po[k]:po[k + 1]] == exp[k], int(idx[k])
    with pytest.raises(ctmr.CtmrError):
        eng.pem_encode_device(d_pay.data_ptr(), d_off.data_ptr(), d_idx.data_ptr(), len(idx), d_pem.data_ptr(),
                              total - 1, d_po.data_ptr())
    eng.close()


def test_pem_device_output_blocks_unaligned_pointer_large_and_tiny_certificates():
    """k_pem_encode cuts the PEM stream into 7 KiB output blocks on 16-byte boundaries of the output ADDRESS: output
    pointers at every offset mod 16, certificates far larger than a block (70 000 bytes: 13 blocks), runs of tiny ones
    (more than 64 certificates inside one block: a second pass), and a stream that ends inside its first block."""
    import base64
    import torch
    rng = np.random.default_rng(11)
    dev = torch.device("cuda:0")
    eng = ctmr.Engine(device=0, table_slots=1 << 10, pair_slots=1 << 10)

    def pem(b):
        e = base64.b64encode(b)
        return b"-----BEGIN CERTIFICATE-----\n" + b"".join(e[i:i + 64] + b"\n" for i in range(0, len(e), 64)) + \
            b"-----END CERTIFICATE-----\n"

    cases = [[70000, 3, 0, 1523, 12, 4095, 4096, 4097, 1],
             [0] * 200 + [1, 2] * 100 + [1523] * 5,
             [5],
             [int(x) for x in rng.normal(1523, 64, 300)],
             [int(x) for x in rng.integers(0, 9000, 120)],
             # every residue of "input chunks in a block" mod 64, with 3-20 certificates per block: a row of chunk loads whose
             # few active lanes do not include their owner's lane (round 5: a gather under that mask read zero)
             [int(x) for x in rng.integers(150, 1300, 6000)],
             [int(x) for x in rng.integers(40, 3000, 3000)]]
    for ci, lens in enumerate(cases):
        blobs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in lens]
        lead = int(rng.integers(0, 16))                                  # the first certificate at any address mod 16
        offs = np.zeros(len(blobs) + 1, np.uint64)
        offs[0] = lead
        offs[1:] = lead + np.cumsum([len(b) for b in blobs])
        payload = np.frombuffer(bytes(lead) + b"".join(blobs) + bytes(64), np.uint8)
        idx = np.arange(len(blobs), dtype=np.uint64)
        d_pay = torch.from_numpy(payload.copy()).to(dev)
        d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
        d_idx = torch.from_numpy(idx.astype(np.int64)).to(dev)
        d_po = torch.zeros(len(idx) + 1, dtype=torch.int64, device=dev)
        want = b"".join(pem(b) for b in blobs)
        for shift in ([0, 1, 7, 15] if ci else list(range(16))):
            d_pem = torch.full((len(want) + 64,), 0xEE, dtype=torch.uint8, device=dev)
            total = eng.pem_encode_device(d_pay.data_ptr(), d_off.data_ptr(), d_idx.data_ptr(), len(idx),
                                          d_pem.data_ptr() + shift, len(want), d_po.data_ptr())
            assert total == len(want)
            raw = d_pem.cpu().numpy().tobytes()
            assert raw[shift:shift + total] == want, (ci, shift)
            assert raw[:shift] == b"\xee" * shift and raw[shift + total:] == b"\xee" * (64 - shift), "wrote outside its buffer"
    eng.close()


def test_fingerprint_matches_hashlib():
    """Auxiliary whole-certificate SHA-256 (not on the reference's path): every length 0..300 covers
    all block/padding cases (55/56/63/64-byte boundaries), plus typical DER sizes; packed and ranged addressing."""
    import hashlib
    import torch
    rng = np.random.default_rng(9)
    lens = list(range(0, 301)) + [717, 1297, 1523, 1536, 2000, 4095, 70000]
    blobs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in lens]
    offs = np.zeros(len(blobs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(b) for b in blobs])
    payload = np.frombuffer(b"".join(blobs) + bytes(64), np.uint8)
    dev = torch.device("cuda:0")
    d_pay = torch.from_numpy(payload.copy()).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_dg = torch.zeros(len(blobs) * 32, dtype=torch.uint8, device=dev)
    eng = ctmr.Engine(device=0, table_slots=1 << 10, pair_slots=1 << 10)
    eng.fingerprint_device(d_pay.data_ptr(), d_off.data_ptr(), 0, len(blobs), d_dg.data_ptr())
    got = d_dg.cpu().numpy().tobytes()
    for k, b in enumerate(blobs):
        assert got[32 * k:32 * k + 32] == hashlib.sha256(b).digest(), lens[k]
    # ranged form: [start, end) pairs, here every certificate without its last byte
    d_start = torch.from_numpy(offs[:-1].astype(np.int64)).to(dev)
    ends = np.maximum(offs[1:].astype(np.int64) - 1, offs[:-1].astype(np.int64))
    d_end = torch.from_numpy(ends).to(dev)
    eng.fingerprint_device(d_pay.data_ptr(), d_start.data_ptr(), d_end.data_ptr(), len(blobs), d_dg.data_ptr())
    got = d_dg.cpu().numpy().tobytes()
    for k, b in enumerate(blobs):
        assert got[32 * k:32 * k + 32] == hashlib.sha256(b[:-1]).digest(), lens[k]
    eng.close()


def test_pem_device_many_blocks_per_wave():
    """The resident grid of k_pem_encode is 12 waves per CU; a NEW list of a few thousand certificates gives every wave one
    block and never enters the software-pipelined loop (next block's bytes and the bounds of the block after in flight).  300 000
    certificates are ≈ 90 000 blocks: ≈ 30 per wave, compared bytewise on the device."""
    import torch
    from scripts.fuzz_gpu_pem import big
    eng = ctmr.Engine(device=0, table_slots=1 << 10, pair_slots=1 << 10)
    big(eng, torch.device("cuda:0"), np.random.default_rng(23), 300_000)      # raises on a difference
    eng.close()
