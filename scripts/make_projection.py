#!/usr/bin/env python3
"""profiles/rNN/projected_ms_phase_by_N.json from the per-rank costs scripts/rank_cost_at_world.py measured on ONE GPU
(final_rank_cost_w{2,4,8}.json) and the default line (final_bench_default.json): per phase kernel milliseconds of a rank at
world W, the bytes it hands to the wire, and a step projected with xGMI taken at (W - 1) links x 50 GB/s per direction and
0.3 ms of control collectives — the table DESIGN.md §8 prints, to be held against the driver's SCALE run phase by phase.
    python scripts/make_projection.py profiles/r06
"""
import json
import os
import sys


def last(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


def main():
    R = sys.argv[1]
    link, ctl = 50.0, 0.3
    d = last(os.path.join(R, "final_bench_default.json"))
    out = {"what": "per-rank kernel milliseconds of ONE rank at world W, measured on one MI355X by driving the per-rank steps with "
                   "world = W, rank = 0 (scripts/rank_cost_at_world.py E W; the rank's own exported records stand in for what it "
                   "would receive); reference profile (the engine's default since round 6); strong scaling of ONE 100 M-entry "
                   "batch: E = 100 M / W.  To be held against the driver's SCALE run phase by phase (exchange.ms_phase_rank0 of "
                   "bench.py --gpus N).  Real RCCL between devices is UNMEASURED.",
           "N1_measured": {"file": "final_bench_default.json", "ms_per_step": d["ms_per_step"], "kernel_ms": d["kernel_ms"]},
           "by_N": {}}
    for W in (2, 4, 8):
        r = last(os.path.join(R, "final_rank_cost_w%d.json" % W))
        links = W - 1
        row = {"entries_per_rank": r["entries"], "plain_reduce_of_the_shard_ms": r["plain_ms"]}
        for mode, wire in (("owner", r["owner"]["record_bytes_to_the_wire"]),
                           ("bloom", r["bloom"]["filter_bytes_to_gather_per_rank"] * links + r["bloom"]["candidate_records"] * 64)):
            m = r[mode]
            ms_wire = wire / (links * link * 1e9) * 1e3
            step = m["sum_ms"] + ms_wire + ctl
            row[mode] = {"phases_ms": {k: v for k, v in m.items() if k.endswith("_ms") and k != "sum_ms"}, "kernels_sum_ms": m["sum_ms"],
                         "wire_bytes_per_rank": wire, "wire_ms_at_%d_links_x_50_GBps" % links: ms_wire, "control_ms_assumed": ctl,
                         "projected_step_ms": step, "projected_certificates_per_s": 1e8 / (step / 1e3)}
        out["by_N"][str(W)] = row
    syncs = {}
    for mode, f in (("owner", "bench_n2_owner_20m_one_gpu_stand_in_transport.json"), ("bloom", "bench_n2_bloom_20m_one_gpu_stand_in_transport.json")):
        p = os.path.join(R, f)
        if os.path.exists(p):
            x = last(p)["exchange"]
            syncs[mode] = {"host_syncs_per_round": x["host_syncs_per_round"], "ms_control_over_the_stand_in": x["ms_control"], "file": f}
    out["host_syncs_per_round_measured_over_the_stand_in_transport"] = syncs
    open(os.path.join(R, "projected_ms_phase_by_N.json"), "w").write(json.dumps(out, indent=1) + "\n")
    for W, row in out["by_N"].items():
        print(W, {m: (round(row[m]["kernels_sum_ms"], 1), round(row[m]["projected_step_ms"], 1), "%.2e" % row[m]["projected_certificates_per_s"]) for m in ("owner", "bloom")})


if __name__ == "__main__":
    main()
