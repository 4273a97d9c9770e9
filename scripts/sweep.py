#!/usr/bin/env python3
"""GPU tuning sweep: one resident synthetic batch, many map-kernel configurations.
usage: sweep.py ENTRIES "variant:certs_per_tile:lds_bytes,..." [issuers]
Runs against libctmr_sweep.so (the product + the baseline designs 1 = LDS tile, 2 = direct; built here if stale)."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ct_mapreduce_amd import build as _b
os.environ["CTMR_LIB"] = _b.build(sweep=True)
import torch
import ct_mapreduce_amd as ctmr
from ct_mapreduce_amd import synth, _native as N

E = int(sys.argv[1])
cfgs = [tuple(int(x) for x in c.split(":")) for c in sys.argv[2].split(",")]
ni = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
cfg = synth.config(seed=20260921 + 4, n_issuers=ni, zipf=1, dup_permille=0, ca_permille=10, expired_permille=10)
issuers = synth.issuers(cfg)
gen = ctmr.Engine(device=0, table_slots=1 << 10, pair_slots=1 << 10)
d_off = torch.empty(E + 1, dtype=torch.int64, device=dev)
total = gen.synth_device(cfg, 0, E, d_off.data_ptr(), 0, 0, 0, 0)
d_pay = torch.empty(total + N.PAYLOAD_PAD + 16, dtype=torch.uint8, device=dev)
d_iss = torch.empty(E, dtype=torch.int32, device=dev)
d_et = torch.empty(E, dtype=torch.uint8, device=dev)
gen.synth_device(cfg, 0, E, d_off.data_ptr(), d_pay.data_ptr(), d_pay.numel(), d_iss.data_ptr(), d_et.data_ptr())
gen.close()
d_rec = torch.empty(E * 32, dtype=torch.uint8, device=dev)
d_new = torch.empty(E, dtype=torch.int64, device=dev)
slots = 1
while slots < 2 * E:
    slots <<= 1
ref = None
for (v, c, lds) in cfgs:
    eng = ctmr.Engine(device=0, table_slots=slots, map_variant=v, certs_per_tile=c, lds_tile_bytes=lds, profile=True)
    eng.add_issuers(issuers)
    eng.set_filter(b"Synth Issuer 0,Synth Issuer 1", False, synth.BASE_TIME)
    ms = []
    for it in range(4):
        eng.reset_known()
        st = eng.map_batch_device(d_pay.data_ptr(), d_off.data_ptr(), d_iss.data_ptr(), d_et.data_ptr(), E,
                                  d_rec.data_ptr(), d_new.data_ptr())
        ms.append((st.ms_map, st.ms_insert, st.ms_resolve, st.ms_compact, st.ms_total))
    best = min(ms[1:])
    alg = st.payload_bytes + 45 * E
    sig = (int(st.n_new), [int(x) for x in st.by_status])
    if ref is None:
        ref = sig
    print(json.dumps({"cfg": [v, c, lds], "ms_map": round(best[0], 4), "GBps": round(alg / best[0] / 1e6, 1),
                      "frac": round(alg / best[0] / 1e6 / 8000, 4), "ms_insert": round(best[1], 4),
                      "ms_resolve": round(best[2], 4), "ms_compact": round(best[3], 4),
                      "ms_total": round(best[4], 4), "same_result": sig == ref}), flush=True)
    eng.close()
