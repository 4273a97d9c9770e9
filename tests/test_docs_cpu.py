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
