"""The C++ host mirror of the reference's storage package (include/ctmr_storage.hpp) — the host side a Go
maintainer would write with cgo, written in C++ because this image has no Go toolchain — is tested by
tests/host/test_storage.cpp, which mirrors storage/*_test.go.  This wrapper builds and runs it:
CPU suites (Mock cache/backends, LocalDisk) without a GPU, the GpuRemoteCache / StoreBatch suites with -m gpu."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "test_storage.cpp")
HDRS = [os.path.join(ROOT, "include", "ctmr_storage.hpp"), os.path.join(ROOT, "include", "ctmr.h"),
        os.path.join(ROOT, "tests", "host", "ctmr_storage_mocks.hpp")]
EXE = os.path.join(ROOT, "tests", "host", "test_storage.run")
LIBDIR = os.path.join(ROOT, "ct_mapreduce_amd")


def build_host_tests(force=False):
    from ct_mapreduce_amd import build as b
    b.build(force=False, verbose=False)                     # libctmr.so (the C ABI the mirror binds)
    newest = max(os.path.getmtime(p) for p in [SRC] + HDRS)
    if force or not os.path.exists(EXE) or os.path.getmtime(EXE) < newest:
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Wno-unused-result", "-D__HIP_PLATFORM_AMD__",
                               "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", SRC, "-o", EXE, "-L" + LIBDIR,
                               "-lctmr", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def run(args, tmp_path):
    exe = build_host_tests()
    p = subprocess.run([exe, "--golden", os.path.join(ROOT, "tests", "golden"), "--tmp", str(tmp_path)] + args,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout
    assert " 0 failed" in p.stdout, p.stdout
    return p.stdout


def test_host_mirror_cpu_suites(tmp_path):
    out = run([], tmp_path)
    for name in ("Test_Serial", "Test_ExpDate", "Test_Unknown_Mock", "Test_ExpireAt_Mock", "Test_HostCert",
                 "Test_Mock_cache_suites", "Test_ListExpiration", "Test_NoopBackend", "Test_LocalDisk"):
        assert "ok   " + name in out, out


@pytest.mark.gpu
def test_host_mirror_gpu_suites(tmp_path):
    out = run(["--gpu"], tmp_path)
    for name in ("Test_IssuerLazyInit_Gpu", "Suite_KnownCertificates", "Suite_DuplicateCRLs", "Suite_Accumulate",
                 "Suite_GetIssuerAndDatesFromCache", "Suite_LogState", "Test_ExpireAt_Gpu", "Test_StoreBatch_Gpu",
                 "Test_StoreRawBatch_DeviceMeta_Gpu", "Test_Pipeline_Gpu", "Test_Group_Gpu"):
        assert "ok   " + name in out, out
