"""GPU campaign for k_pem_encode (N1): random certificate-length mixtures, payload alignments and output addresses against
Python's base64 — the block cutter, the owner gathers, the last-line tasks and the slow path for blocks of tiny certificates
all depend on how lengths fall into the 7 KiB output blocks.
    python scripts/fuzz_gpu_pem.py <trials> <seed> [<certificates of one big call in front>]
"""
import base64
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ct_mapreduce_amd as ctmr  # noqa: E402

HEAD, TAIL = b"-----BEGIN CERTIFICATE-----\n", b"-----END CERTIFICATE-----\n"


def pem(b):
    e = base64.b64encode(b)
    return HEAD + b"".join(e[i:i + 64] + b"\n" for i in range(0, len(e), 64)) + TAIL


def lengths(rng):
    kind = int(rng.integers(0, 8))
    n = int(rng.integers(1, 400))
    if kind == 0:
        return rng.integers(0, 60, n)                               # tiny: the certificate-by-certificate path
    if kind == 1:
        return rng.integers(100, 260, n)                            # 12-25 certificates per block: either path
    if kind == 2:
        return rng.integers(300, 1400, n)
    if kind == 3:
        return rng.normal(1523, 64, n).clip(0).astype(np.int64)     # the bench's shape
    if kind == 4:
        return rng.integers(0, 9000, n)
    if kind == 5:
        return np.where(rng.random(n) < 0.05, rng.integers(20000, 90000, n), rng.integers(0, 200, n))   # giants among tiny ones
    if kind == 6:
        k = int(rng.integers(1, 6)) * 48 + int(rng.integers(-2, 3))
        return np.full(n, max(k, 0))                                # around whole lines: every last-line length
    return rng.choice([0, 1, 2, 3, 47, 48, 49, 95, 96, 97, 5375, 5376, 5377], n)


def big(eng, dev, rng, n):
    """One call over n certificates: every wave of the resident grid walks MANY blocks (the software-pipelined loop, which a
    few hundred certificates never enter: they give every wave one block)."""
    t0 = time.time()
    lens = rng.normal(1523, 64, n).clip(0).astype(np.int64)
    odd = rng.random(n)
    lens = np.where(odd < 0.01, rng.integers(0, 100, n), np.where(odd > 0.999, rng.integers(10000, 60000, n), lens))
    offs = np.zeros(n + 1, np.int64)
    offs[0] = int(rng.integers(0, 16))
    offs[1:] = offs[0] + np.cumsum(lens)
    payload = rng.integers(0, 256, int(offs[-1]) + 64, dtype=np.uint8)
    pb = payload.tobytes()
    want = b"".join(pem(pb[offs[i]:offs[i + 1]]) for i in range(n))
    d_pay = torch.from_numpy(payload).to(dev)
    d_off = torch.from_numpy(offs).to(dev)
    d_idx = torch.arange(n, dtype=torch.int64, device=dev)
    d_po = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    shift = int(rng.integers(0, 16))
    d_pem = torch.full((len(want) + 64,), 0xEE, dtype=torch.uint8, device=dev)
    total = eng.pem_encode_device(d_pay.data_ptr(), d_off.data_ptr(), d_idx.data_ptr(), n, d_pem.data_ptr() + shift, len(want),
                                  d_po.data_ptr())
    d_want = torch.from_numpy(np.frombuffer(want, np.uint8).copy()).to(dev)
    same = total == len(want) and bool(torch.equal(d_pem[shift:shift + total], d_want)) and \
        bool((d_pem[:shift] == 0xEE).all()) and bool((d_pem[shift + total:] == 0xEE).all())
    if not same:
        bad = torch.nonzero(d_pem[shift:shift + len(want)] != d_want)
        raise AssertionError(f"FUZZ GPU PEM MISMATCH big call of {n}: total {total} want {len(want)}, {bad.numel()} bad bytes, "
                             f"first {int(bad[0]) if bad.numel() else -1}")
    print(f"FUZZ GPU PEM BIG OK one call over {n} certificates: {total} PEM bytes = {(total + 7167) // 7168} blocks of 7 KiB, "
          f"0 differences, {time.time() - t0:.0f} s", flush=True)


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    eng = ctmr.Engine(device=0, table_slots=1 << 10, pair_slots=1 << 10)
    if len(sys.argv) > 3:
        big(eng, dev, rng, int(sys.argv[3]))
    t0 = time.time()
    certs = nbytes = 0
    for t in range(trials):
        lens = [int(x) for x in lengths(rng)]
        blobs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in lens]
        lead = int(rng.integers(0, 16))
        order = rng.permutation(len(blobs)) if rng.random() < 0.3 else np.arange(len(blobs))    # the NEW list in any order
        offs = np.zeros(len(blobs) + 1, np.uint64)
        offs[0] = lead
        offs[1:] = lead + np.cumsum([len(b) for b in blobs])
        payload = np.frombuffer(bytes(lead) + b"".join(blobs) + bytes(64), np.uint8)
        d_pay = torch.from_numpy(payload.copy()).to(dev)
        d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
        d_idx = torch.from_numpy(order.astype(np.int64)).to(dev)
        d_po = torch.zeros(len(blobs) + 1, dtype=torch.int64, device=dev)
        want = b"".join(pem(blobs[int(i)]) for i in order)
        shift = int(rng.integers(0, 16))
        d_pem = torch.full((len(want) + 64,), 0xEE, dtype=torch.uint8, device=dev)
        total = eng.pem_encode_device(d_pay.data_ptr(), d_off.data_ptr(), d_idx.data_ptr(), len(blobs), d_pem.data_ptr() + shift,
                                      len(want), d_po.data_ptr())
        raw = d_pem.cpu().numpy().tobytes()
        if total != len(want) or raw[shift:shift + total] != want or raw[:shift] != b"\xee" * shift or \
                raw[shift + total:] != b"\xee" * (64 - shift):
            got = np.frombuffer(raw[shift:shift + len(want)], np.uint8)
            bad = np.nonzero(got != np.frombuffer(want, np.uint8))[0]
            print(f"FUZZ GPU PEM MISMATCH trial {t} seed {seed}: total {total} want {len(want)}, first bad byte "
                  f"{int(bad[0]) if len(bad) else -1} of {len(bad)}, lead {lead} shift {shift} lens[:12] {lens[:12]}", flush=True)
            sys.exit(1)
        po = d_po.cpu().numpy()
        assert po[0] == 0 and po[-1] == total
        certs += len(blobs)
        nbytes += total
    print(f"FUZZ GPU PEM OK {trials} trials, seed {seed}: {certs} certificates, {nbytes} PEM bytes, 0 differences, "
          f"{time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
