"""Synthetic CT batches on the host; same bytes as the device generator."""
import ctypes as C

import numpy as np

from . import _native as N
from .engine import Batch, RawEntries

BASE_TIME = 1767225600  # 2026-01-01T00:00:00Z


def config(seed=20260921, n_issuers=1, zipf=1, dup_permille=0, ca_permille=10, expired_permille=10,
           mean_len=1536, base_time=BASE_TIME, profile=0):
    """profile 0: the SURVEY §8(d) corpus; 1: mixed keys (EC/RSA), long OV-like subjects, GeneralizedTime."""
    return N.SynthConfig(seed=seed, n_issuers=n_issuers, zipf=zipf, dup_permille=dup_permille,
                         ca_permille=ca_permille, expired_permille=expired_permille,
                         mean_len=mean_len, base_time=base_time, profile=profile, reserved=0)


def leaf(cfg, i):
    L = N.lib()
    buf = (C.c_uint8 * 4096)()
    ii, et = C.c_uint32(), C.c_uint8()
    n = L.ctmr_synth_leaf(C.byref(cfg), i, buf, 4096, C.byref(ii), C.byref(et))
    return bytes(buf)[:n], ii.value, et.value


def issuer(cfg, k):
    L = N.lib()
    buf = (C.c_uint8 * 4096)()
    n = L.ctmr_synth_issuer(C.byref(cfg), k, buf, 4096)
    return bytes(buf)[:n]


def issuers(cfg):
    return [issuer(cfg, k) for k in range(max(cfg.n_issuers, 1))]


def host_batch(cfg, first, n) -> Batch:
    L = N.lib()
    offsets = np.zeros(n + 1, dtype=np.uint64)
    iss = np.zeros(max(n, 1), dtype=np.uint32)
    et = np.zeros(max(n, 1), dtype=np.uint8)
    cap = n * 2048 + 64
    payload = np.zeros(cap, dtype=np.uint8)
    used = L.ctmr_synth_host(C.byref(cfg), first, n, offsets.ctypes.data, payload.ctypes.data, cap,
                             iss.ctypes.data, et.ctypes.data)
    assert used <= cap
    return Batch(payload[:used + N.PAYLOAD_PAD], offsets, iss[:n], et[:n])


def host_entries(cfg, first, n) -> RawEntries:
    """Raw get-entries form (leaf_input ‖ extra_data per entry) of the same synthetic entries."""
    L = N.lib()
    bounds = np.zeros(2 * n + 1, dtype=np.uint64)
    cap = n * 6144 + 64
    blob = np.zeros(cap, dtype=np.uint8)
    used = L.ctmr_synth_entries_host(C.byref(cfg), first, n, bounds.ctypes.data, blob.ctypes.data, cap)
    assert used <= cap
    return RawEntries(blob[:used + N.PAYLOAD_PAD], bounds)
