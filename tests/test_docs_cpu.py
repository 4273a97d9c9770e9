"""The documents stay reviewable: DESIGN.md within 120 columns (round-3 review), every evidence file listed."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_md_fits_120_columns():
    for n, line in enumerate(open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read().split("\n"), 1):
        assert len(line.encode()) <= 120, (n, len(line.encode()))      # bytes: what `awk 'length > 120'` counts


def test_the_latest_profiles_readme_lists_every_file():
    rounds = sorted(d for d in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d\d", d))
    d = os.path.join(ROOT, "profiles", rounds[-1])
    text = open(os.path.join(d, "README.md"), encoding="utf-8").read()
    names = [t.lstrip("…") for t in re.findall(r"`([^`]+)`", text)]
    for f in sorted(os.listdir(d)):
        if f == "README.md":
            continue
        assert any(f == t or (t and f.endswith(t)) for t in names), f


def _last_json(path):
    import json
    return json.loads([l for l in open(os.path.join(ROOT, path)).read().splitlines() if l.startswith("{")][-1])


def _kernel_avg_ms(path, needle):
    import csv
    for row in csv.DictReader(open(os.path.join(ROOT, path))):
        if needle in row["Name"]:
            return float(row["AverageNs"]) / 1e6
    raise AssertionError((path, needle))


def test_numbers_design_md_quotes_for_a_profile_file_are_that_files_numbers():
    """VERDICT r04 #8 (a stale "30.8 ms per 50 M wave" sat in DESIGN §9 next to a file that said otherwise): every headline
    number DESIGN.md quotes for a file under profiles/r06/ (§7) and for the round-5 files §9 still cites is recomputed here
    from that file, formatted as the document formats it, and must occur in the document — so must the file's name."""
    text = open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read()
    flat = " ".join(text.split())                                       # quotes may wrap over lines
    g9 = lambda v: ("%.2f × 10⁹" % (v / 1e9))
    R = "profiles/r06/"
    K = "k_map_fused<14, false, 0, true>"
    d = _last_json(R + "final_bench_default.json")
    p = _last_json(R + "final_bench_default_under_rocprofv3.json")
    st = _last_json(R + "final_bench_stream_noop_1b.json")
    wb = st["write_back"]
    pem = _last_json(R + "final_bench_pem.json")["pem"]
    fast = d["secondary"]["fast_profile"]
    rawref = d["secondary"]["raw_reference"]
    assert d["config"]["profile"] == "reference" and d["roofline"]["kernel"] == K and fast["same_results_as_the_reference_profile"] is True
    assert d["roofline"]["frac"] == min(d["roofline"]["frac_physical"], d["roofline"]["frac_algorithmic"])    # the conservative one
    quotes = {
        R + "final_bench_default.json": [g9(d["value"]), "%.1f ms per step" % d["ms_per_step"], "map kernel %.2f ms" % d["kernel_ms"]["map"],
                                         "`roofline.frac` %.3f" % d["roofline"]["frac"], "`frac_physical` %.3f" % d["roofline"]["frac_physical"],
                                         "`traffic_over_algorithmic` %.2f" % d["roofline"]["traffic_over_algorithmic"],
                                         g9(fast["value"]), "map %.1f ms" % fast["map_ms"], "`frac` %.3f" % fast["frac"],
                                         "%.2f × 10⁹ entries/s" % (rawref["value"] / 1e9)],
        R + "final_bench_default_under_rocprofv3.json": [g9(p["value"]), "map %.2f ms by HIP events" % p["kernel_ms"]["map"]],
        R + "final_default_prof_kernel_stats.csv": ["%.2f ms average over 6 launches" % _kernel_avg_ms(R + "final_default_prof_kernel_stats.csv", K)],
        R + "final_bench_stream_noop_1b.json": ["%.3f × 10⁹ entries/s" % (st["value"] / 1e9), "%.1f ms per wave" % st["ms_per_step"],
                                                "%.1f ms in all" % wb["ms_pem_total"], "%.2f TB/s" % (wb["pem_read_plus_written_GB_per_s"] / 1e3),
                                                "%.2f of peak" % wb["pem_frac_of_hbm_peak"]],
        R + "final_bench_pem.json": ["%.1f ms = %.2f TB/s" % (pem["ms_wall"], pem["GBps_read_plus_written"] / 1e3)],
        R + "final_bench_mixed.json": [g9(_last_json(R + "final_bench_mixed.json")["value"])],
        R + "final_bench_raw.json": [g9(_last_json(R + "final_bench_raw.json")["value"]) + " entries/s"],
        R + "final_bench_raw_fast_profile.json": [g9(_last_json(R + "final_bench_raw_fast_profile.json")["value"]) + " entries/s"],
        R + "final_bench_stream_noop_1b_fast_profile.json": ["%.3f × 10⁹ entries/s" % (_last_json(R + "final_bench_stream_noop_1b_fast_profile.json")["value"] / 1e9)],
    }
    # §9 N1 / N2 still tell the story of round 5's PEM encoder and raw path with round 5's files
    R5 = "profiles/r05/"
    st5 = _last_json(R5 + "final_bench_stream_noop_1b.json")
    wb5 = st5["write_back"]
    pem5 = _last_json(R5 + "final_bench_pem.json")["pem"]
    quotes.update({
        R5 + "final_bench_stream_noop_1b.json": ["%.3f × 10⁹ entries/s" % (st5["value"] / 1e9), "%.1f ms per wave" % st5["ms_per_step"],
                                                 "%.1f ms in all" % wb5["ms_pem_total"], "%.2f TB/s" % (wb5["pem_read_plus_written_GB_per_s"] / 1e3)],
        R5 + "final_bench_pem.json": ["%.1f ms = %.2f TB/s" % (pem5["ms_wall"], pem5["GBps_read_plus_written"] / 1e3)],
        R5 + "final_pem_prof_kernel_stats.csv": ["%.2f ms average" % _kernel_avg_ms(R5 + "final_pem_prof_kernel_stats.csv", "k_pem_encode")],
        R5 + "final_bench_raw.json": [g9(_last_json(R5 + "final_bench_raw.json")["value"]) + " entries/s"],
    })
    for f, qs in quotes.items():
        assert os.path.basename(f) in text, f
        for q in qs:
            assert q in flat, (f, q)
    r4 = _last_json("profiles/r04/stream_writeback_noop_1b.json")
    assert "%.3f × 10⁹" % (r4["value"] / 1e9) in flat and "%.1f ms of PEM" % r4["write_back"]["ms_pem_total"] in flat


def test_the_projection_file_is_what_its_script_makes_of_the_rank_costs():
    """profiles/r06/projected_ms_phase_by_N.json = scripts/make_projection.py over the committed rank costs,
    and DESIGN.md §8's round-6 table prints its projected steps."""
    import json
    import shutil
    import subprocess
    import sys
    import tempfile
    src = os.path.join(ROOT, "profiles", "r06")
    with tempfile.TemporaryDirectory() as tmp:
        for f in os.listdir(src):
            if f.startswith(("final_rank_cost_w", "final_bench_default.json", "bench_n2_")):
                shutil.copy(os.path.join(src, f), tmp)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "make_projection.py"), tmp], stdout=subprocess.DEVNULL)
        made = json.load(open(os.path.join(tmp, "projected_ms_phase_by_N.json")))
    kept = json.load(open(os.path.join(src, "projected_ms_phase_by_N.json")))
    assert made == kept
    flat = " ".join(open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read().split())
    for W in ("2", "4", "8"):
        x = kept["by_N"][W]
        assert "≈ %.1f / %.1f ms" % (x["bloom"]["projected_step_ms"], x["owner"]["projected_step_ms"]) in flat, W
