#!/usr/bin/env python3
"""PCIe-inclusive rate of the HOST-buffer entry point (ctmr_map_batch: stage H2D, map + reduce, records and NEW list
D2H) at the batch sizes a ct-fetch host would use — 1 001 entries per get-entries response (ct-fetch.go:417-424),
the entryChan capacity 16 384 (:132), and larger.  DESIGN.md §7 quotes these; `bench.py`'s value never includes PCIe."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import ct_mapreduce_amd as ctmr  # noqa: E402
from ct_mapreduce_amd import synth  # noqa: E402


def main():
    cfg = synth.config(seed=20260921 + 4, n_issuers=256, zipf=1, ca_permille=10, expired_permille=10)
    issuers = synth.issuers(cfg)
    out = []
    for pinned, n, calls in ((False, 1001, 200), (False, 16384, 50), (False, 262144, 8), (True, 16384, 50), (True, 262144, 8)):
        eng = ctmr.Engine(device=0, table_slots=1 << 24, pair_slots=1 << 16)
        eng.add_issuers(issuers)
        eng.set_filter(b"Synth Issuer 0,Synth Issuer 1", False, synth.BASE_TIME)
        batches = [synth.host_batch(cfg, k * n, n) for k in range(min(calls, 4))]
        if pinned:                               # same bytes in page-locked memory (ctmr_alloc_pinned)
            for b in batches:
                p = eng.pinned_array(b.payload.nbytes)
                p[:] = b.payload
                b.payload = p
        eng.map_batch(batches[0])
        eng.reset_known()
        t0 = time.perf_counter()
        new = 0
        for k in range(calls):
            new += eng.map_batch(batches[k % len(batches)]).stats.n_new
        dt = time.perf_counter() - t0
        nbytes = sum(int(b.offsets[-1]) for b in batches) / len(batches)
        out.append({"payload_memory": "pinned" if pinned else "pageable", "entries_per_call": n, "calls": calls, "ms_per_call": dt / calls * 1e3,
                    "certs_per_s": n * calls / dt, "payload_GBps": nbytes * calls / dt / 1e9})
        eng.close()
    print(json.dumps({"host_buffer_entry_point": "ctmr_map_batch (records + NEW list copied back)",
                      "results": out}))
    # ---- the asynchronous entry points: get-entries-sized batches submitted ahead, collected WINDOW tickets later
    out = []
    for pinned, n, calls, window in ((False, 1001, 2000, 64), (True, 1001, 2000, 64), (True, 1001, 2000, 8), (True, 16384, 200, 8)):
        eng = ctmr.Engine(device=0, table_slots=1 << 24, pair_slots=1 << 16)
        eng.add_issuers(issuers)
        eng.set_filter(b"Synth Issuer 0,Synth Issuer 1", False, synth.BASE_TIME)
        batches = [synth.host_batch(cfg, k * n, n) for k in range(8)]
        arrs = []
        for b in batches:
            pay = b.payload
            if pinned:
                p = eng.pinned_array(pay.nbytes + 32)
                p[:pay.nbytes] = pay
                pay = p
            arrs.append((pay, b.offsets.astype("uint64"), b.issuer_idx.astype("uint32"), b.entry_type.astype("uint8")))
        t = eng.submit_batch(*arrs[0], n)
        eng.wait(t, n)
        eng.reset_known()
        t0 = time.perf_counter()
        inflight, new = [], 0
        for k in range(calls):
            inflight.append(eng.submit_batch(*arrs[k % 8], n))
            if len(inflight) > window:
                new += eng.wait(inflight.pop(0), n, want_new=False).stats.n_new
        for tk in inflight:
            new += eng.wait(tk, n, want_new=False).stats.n_new
        dt = time.perf_counter() - t0
        nbytes = sum(int(b.offsets[-1]) for b in batches) / len(batches)
        out.append({"payload_memory": "pinned" if pinned else "pageable", "entries_per_submit": n, "submits": calls,
                    "tickets_in_flight": window, "us_per_submit": dt / calls * 1e6, "certs_per_s": n * calls / dt,
                    "payload_GBps": nbytes * calls / dt / 1e9})
        eng.close()
    print(json.dumps({"host_buffer_entry_point": "ctmr_submit_batch / ctmr_wait (records copied back per ticket)",
                      "results": out}))


if __name__ == "__main__":
    main()
