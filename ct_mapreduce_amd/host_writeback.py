"""ctypes binding of libctmr_host.so (ct_mapreduce_amd/host/host_writeback.cpp): the HOST half of FilesystemDatabase.Store's
write-back — backend.StoreCertificatePEM for the newly unknown certificates of a batch the GPU has mapped and PEM-encoded
(storage/filesystemdatabase.go:183-208), through the C++ restatement of the reference's NoopBackend / LocalDiskBackend
(include/ctmr_storage.hpp).  Asynchronous, double-buffered by the caller.  Plain host code: no GPU, no libctmr."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libctmr_host.so")
SRC = os.path.join(HERE, "host", "host_writeback.cpp")
HDR = os.path.join(HERE, "..", "include", "ctmr_storage.hpp")
_lib = None


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-pthread", SRC, "-o", LIB])
    return LIB


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.ctmr_host_writer_open.argtypes = [C.c_char_p, C.c_int]
        L.ctmr_host_writer_open.restype = C.c_void_p
        L.ctmr_host_writer_set_issuers.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
        L.ctmr_host_writer_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.ctmr_host_writer_wait.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        L.ctmr_host_writer_mark_dirty.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_uint32]
        L.ctmr_host_writer_error.argtypes = [C.c_void_p]
        L.ctmr_host_writer_error.restype = C.c_char_p
        L.ctmr_host_writer_close.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class HostWriter:
    """root: directory of a LocalDiskBackend, or None / "" for the NoopBackend (certPath unset, engine/engine.go:36-40)."""

    def __init__(self, root, issuer_ids, threads=0):
        self._L = lib()
        self.threads = threads or min(32, os.cpu_count() or 1)
        self._h = self._L.ctmr_host_writer_open((root or "").encode(), self.threads)
        ids = "".join(issuer_ids).encode()
        assert len(ids) == 44 * len(issuer_ids), "issuer IDs are 44 characters (Issuer.ID())"
        if self._L.ctmr_host_writer_set_issuers(self._h, ids, len(issuer_ids)):
            raise RuntimeError("ctmr_host_writer_set_issuers failed")

    def submit(self, pem_ptr, pem_off_ptr, recs_ptr, count) -> int:
        """Host pointers (ints) that stay valid and untouched until wait(job)."""
        if not self._h:
            raise RuntimeError("host writer is closed")
        job = self._L.ctmr_host_writer_submit(self._h, pem_ptr, pem_off_ptr, recs_ptr, count)
        if job < 0:
            raise RuntimeError("ctmr_host_writer_submit failed")
        return job

    def wait(self, job):
        """→ (files handed to the backend, their PEM bytes, skipped long serials, seconds from submit to done)"""
        if not self._h:
            raise RuntimeError("host writer is closed")
        out = (C.c_uint64 * 3)()
        sec = C.c_double(0)
        if self._L.ctmr_host_writer_wait(self._h, job, out, C.byref(sec)):
            raise RuntimeError("host write-back failed: " + self._L.ctmr_host_writer_error(self._h).decode())
        return int(out[0]), int(out[1]), int(out[2]), sec.value

    def mark_dirty(self, days):
        arr = (C.c_int32 * len(days))(*[int(d) for d in days])
        if self._L.ctmr_host_writer_mark_dirty(self._h, arr, len(days)):
            raise RuntimeError("markDirty failed: " + self._L.ctmr_host_writer_error(self._h).decode())

    def close(self):
        if self._h:
            self._L.ctmr_host_writer_close(self._h)
            self._h = None
