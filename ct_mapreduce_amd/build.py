"""Builds libctmr.so (hipcc, gfx950 only) in-tree.  No CPU fallback is ever built."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libctmr.so")
SWEEP_LIB = os.path.join(HERE, "libctmr_sweep.so")   # + the two baseline map designs (scripts/sweep.py); never shipped
SOURCES = ["ctmr_engine.hip"]


def deps():
    """Every source the library is compiled from: csrc/ (recursively) and the public header."""
    out = [os.path.join(HERE, "..", "include", "ctmr.h")]
    for d, _, files in os.walk(CSRC):
        out += [os.path.join(d, f) for f in files if f.endswith((".h", ".hip", ".inc"))]
    return out


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in deps())


def build(force=False, verbose=False, sweep=False, defines=(), tag=""):
    """defines / tag: experiment builds of the sweep library (e.g. defines=["CTMR_NT_LOADS"], tag="nt") →
    libctmr_sweep_<tag>.so; the shipped library never takes any."""
    lib = (SWEEP_LIB.replace(".so", "_%s.so" % tag) if tag else SWEEP_LIB) if sweep else LIB
    if not force and not stale(lib):
        return lib
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical"] + (["-DCTMR_SWEEP"] if sweep else []) + \
          ["-D" + d for d in defines] + \
          [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl", "-o", lib]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True, sweep="--sweep" in sys.argv)
