"""The C-ABI library loads and exports every symbol include/ctmr.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from ct_mapreduce_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="ctmr.h"):
    h = open(os.path.join(ROOT, "include", header)).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(ctmr_[a-z_0-9]+)\s*\(", h)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    bench = declared_functions("ctmr_bench.h")      # the corpus generator: exported, but not the drop-in ABI
    assert len(names) >= 30 and not set(names) & set(bench)
    assert all(n.startswith("ctmr_synth_") for n in bench) and not any(n.startswith("ctmr_synth_") for n in names)
    lib = N.lib()
    for n in names + bench:
        assert hasattr(lib, n), n
        assert n in N.SIGNATURES, f"{n} missing from the ctypes binding"
    assert sorted(N.SIGNATURES) == sorted(names + bench)
    assert lib.ctmr_abi_version() == N.ABI_VERSION == 7


def test_struct_sizes_match_the_header():
    assert C.sizeof(N.Config) == 56
    assert C.sizeof(N.IssuerInfo) == 4 + 4 + 32 + 48
    assert C.sizeof(N.SynthConfig) == 8 + 6 * 4 + 8 + 8
    assert C.sizeof(N.BatchStats) == 8 + 8 * 8 + 4 * 8 + 5 * 4 + 4     # by_status[CTMR_ST__COUNT = 8]
    assert C.sizeof(N.EntryView) == 7 * 8
    assert C.sizeof(N.DecodeStats) == 7 * 8 + 2 * 4
    from ct_mapreduce_amd.engine import RECORD_DTYPE
    assert RECORD_DTYPE.itemsize == 32


def test_no_cpu_fallback_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import ct_mapreduce_amd as ctmr
    with pytest.raises(ctmr.CtmrError):
        ctmr.Engine(device=0)


def test_product_never_imports_the_oracle():
    """oracle/ is the checker: nothing under ct_mapreduce_amd/ may import, link or call it."""
    pkg = os.path.join(ROOT, "ct_mapreduce_amd")
    bad = re.compile(r"import\s+oracle|from\s+oracle|liboracle|ctmr_oracle|\borc_[a-z]|oracle/")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert not bad.search(src), f
    so = os.path.join(pkg, "libctmr.so")
    import subprocess
    deps = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "oracle" not in deps


def test_the_documents_quote_the_header_s_symbol_count():
    """README / DESIGN / INTEGRATION say how many entry points the ABI has: keep them honest."""
    n = len(declared_functions())
    for doc in ("README.md", "DESIGN.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, doc)).read()
        quoted = set(int(m) for m in re.findall(r"\b(\d+) symbols", text))
        assert quoted == {n}, (doc, quoted, n)


def test_an_explicit_rccl_library_is_an_error_when_it_cannot_be_loaded():
    """CTMR_RCCL_LIB (group.inc): a named library that does not exist must fail loudly, not fall back to another one."""
    import subprocess
    import sys
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from ct_mapreduce_amd import _native as N; "
            "buf = C.create_string_buffer(N.GROUP_ID_BYTES); print(N.lib().ctmr_group_unique_id(buf))" % ROOT)
    env = dict(os.environ, CTMR_RCCL_LIB="/nonexistent/librccl.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    assert int(out.stdout.strip().splitlines()[-1]) != 0
