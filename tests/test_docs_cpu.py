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
    number DESIGN.md quotes for a file under profiles/r05/ is recomputed here from that file, formatted as the document
    formats it, and must occur in the document — so must the file's name."""
    text = open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read()
    flat = " ".join(text.split())                                       # quotes may wrap over lines
    R = "profiles/r05/"
    d = _last_json(R + "final_bench_default.json")
    p = _last_json(R + "final_bench_default_under_rocprofv3.json")
    st = _last_json(R + "final_bench_stream_noop_1b.json")
    wb = st["write_back"]
    pem = _last_json(R + "final_bench_pem.json")["pem"]
    ref = d["secondary"]["reference_profile"]
    g9 = lambda v: ("%.2f × 10⁹" % (v / 1e9))
    quotes = {
        "final_bench_default.json": [g9(d["value"]), "%.1f ms per step" % d["ms_per_step"], "map kernel %.2f ms" % d["kernel_ms"]["map"],
                                     "`roofline.frac` %.3f" % d["roofline"]["frac"], "%.3f on the algorithmic bytes" % d["roofline"]["frac_algorithmic"],
                                     ("%.1f B per certificate" % (d["roofline"]["traffic"] / 1e8)).replace("1090", "1 090"),
                                     g9(ref["value"]), "map %.1f ms" % ref["map_ms"], "`frac` %.3f" % ref["frac"]],
        "final_bench_default_under_rocprofv3.json": [g9(p["value"]), "map %.2f ms by HIP events" % p["kernel_ms"]["map"]],
        "final_default_prof_kernel_stats.csv": ["%.2f ms average over 6 launches" % _kernel_avg_ms(R + "final_default_prof_kernel_stats.csv", "k_map_fused<16, false, 0, false>")],
        "final_bench_stream_noop_1b.json": ["%.3f × 10⁹ entries/s" % (st["value"] / 1e9), "%.1f ms per wave" % st["ms_per_step"],
                                            "%.1f ms in all" % wb["ms_pem_total"], "%.2f TB/s" % (wb["pem_read_plus_written_GB_per_s"] / 1e3),
                                            "%.2f of peak" % wb["pem_frac_of_hbm_peak"]],
        "final_bench_pem.json": ["%.1f ms = %.2f TB/s" % (pem["ms_wall"], pem["GBps_read_plus_written"] / 1e3)],
        "final_pem_prof_kernel_stats.csv": ["%.2f ms average" % _kernel_avg_ms(R + "final_pem_prof_kernel_stats.csv", "k_pem_encode")],
        "final_bench_reference.json": [g9(_last_json(R + "final_bench_reference.json")["value"])],
        "final_bench_strictext.json": [g9(_last_json(R + "final_bench_strictext.json")["value"])],
        "final_bench_mixed_reference.json": [g9(_last_json(R + "final_bench_mixed_reference.json")["value"])],
        "final_bench_raw.json": [g9(_last_json(R + "final_bench_raw.json")["value"]) + " entries/s"],
    }
    for f, qs in quotes.items():
        assert f in text, f
        for q in qs:
            assert q in flat, (f, q)
    r4 = _last_json("profiles/r04/stream_writeback_noop_1b.json")
    assert "%.3f × 10⁹" % (r4["value"] / 1e9) in flat and "%.1f ms of PEM" % r4["write_back"]["ms_pem_total"] in flat
