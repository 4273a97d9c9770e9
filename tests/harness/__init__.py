"""Builds and binds the test-only helpers (host build of the product walk, OpenSSL extractor)."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def _build(src, out, cmd):
    outp = os.path.join(HERE, out)
    srcp = os.path.join(HERE, src)
    csrc = os.path.join(HERE, "..", "..", "ct_mapreduce_amd", "csrc")
    deps = [os.path.join(csrc, h) for h in ("der_walk.h", "spki_key.h", "ec_curves.h", "entry_decode.h")]
    if (not os.path.exists(outp) or os.path.getmtime(outp) < os.path.getmtime(srcp)
            or any(os.path.getmtime(outp) < os.path.getmtime(d) for d in deps)):
        subprocess.check_call(cmd + [srcp, "-o", outp])
    return outp


class HarnessOut(C.Structure):
    _fields_ = [("ok", C.c_int32), ("serial_off", C.c_uint32), ("serial_len", C.c_uint32),
                ("not_before", C.c_int64), ("not_after", C.c_int64), ("cn_off", C.c_uint32),
                ("cn_len", C.c_uint32), ("bc_valid", C.c_int32), ("is_ca", C.c_int32),
                ("spki_off", C.c_uint32), ("spki_len", C.c_uint32), ("serial_w", C.c_uint32 * 5),
                ("cn_match", C.c_int32), ("nonfatal", C.c_int32)]


class OsslOut(C.Structure):
    _fields_ = [("ok", C.c_int), ("not_before", C.c_longlong), ("not_after", C.c_longlong),
                ("is_ca", C.c_int), ("has_bc", C.c_int), ("serial_len", C.c_int),
                ("serial", C.c_ubyte * 64), ("serial_neg", C.c_int), ("cn_len", C.c_int),
                ("cn", C.c_ubyte * 256), ("spki_len", C.c_int), ("spki", C.c_ubyte * 1024)]


class EntryOut(C.Structure):
    _fields_ = [("ok", C.c_int32), ("entry_type", C.c_int32), ("timestamp", C.c_uint64), ("cert_lo", C.c_uint64),
                ("cert_hi", C.c_uint64), ("chain0_lo", C.c_uint64), ("chain0_len", C.c_uint32),
                ("n_chain", C.c_uint32), ("tbs_lo", C.c_uint64), ("tbs_len", C.c_uint32)]


_walk = None
_ossl = None
_entry = None


def product_decode_entry(leaf_input: bytes, extra_data: bytes, fill=0xA5, prefix=b"") -> EntryOut:
    """The product's decode_entry (host build) on blob = prefix ‖ leaf_input ‖ extra_data."""
    global _entry
    if _entry is None:
        p = _build("entry_harness.cpp", "libentry_harness.so", ["g++", "-O2", "-std=c++17", "-shared", "-fPIC"])
        _entry = C.CDLL(p)
        _entry.harness_decode_entry.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint8,
                                                C.POINTER(EntryOut)]
        _entry.harness_quick_hash.argtypes = [C.c_char_p, C.c_uint32]
        _entry.harness_quick_hash.restype = C.c_uint64
    blob = prefix + leaf_input + extra_data
    o = EntryOut()
    l0 = len(prefix)
    _entry.harness_decode_entry(blob, len(blob), l0, l0 + len(leaf_input), len(blob), fill, C.byref(o))
    return o


def product_walk(der: bytes, fill=0xA5, cn_filter=None) -> HarnessOut:
    """cn_filter=None: no issuerCNFilter (cn_match is True); bytes: the raw filter string."""
    global _walk
    if _walk is None:
        p = _build("walk_harness.cpp", "libwalk_harness.so",
                   ["g++", "-O2", "-std=c++17", "-shared", "-fPIC"])
        _walk = C.CDLL(p)
        _walk.harness_walk_f.argtypes = [C.c_char_p, C.c_uint32, C.c_uint8, C.c_char_p, C.c_uint32,
                                         C.c_int, C.POINTER(HarnessOut)]
    o = HarnessOut()
    f = cn_filter if cn_filter is not None else b""
    _walk.harness_walk_f(der, len(der), fill, f, len(f), int(cn_filter is not None), C.byref(o))
    return o


def product_set_spki(on: bool):
    """The host build's strict_spki switch (default on): parse the public key as CT-go's parsePublicKey does."""
    product_walk(b"\x30\x00")
    _walk.harness_set_spki(int(bool(on)))


def product_set_ext(on: bool):
    """The host build's strict_extensions switch (default off): the bodies of the extensions Go unmarshals."""
    product_walk(b"\x30\x00")
    _walk.harness_set_ext(int(bool(on)))


def product_set_strings(on: bool):
    """The host build's strict_strings switch (default off): findings arrive as WALK_NF_STRING (4) in .nonfatal."""
    product_walk(b"\x30\x00")
    _walk.harness_set_strings(int(bool(on)))


def product_crl_uris(der: bytes, cv: int, ev: int):
    """der_walk.h crl_dps<COLLECT> over the cRLDistributionPoints value der[cv:ev]: None = malformed, else the URIs."""
    product_walk(b"\x30\x00")
    out = (C.c_uint32 * 16)()
    _walk.harness_crl_uris.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32]
    n = _walk.harness_crl_uris(der, len(der), cv, ev, out, 8)
    if n < 0:
        return None
    return [bytes(der[out[2 * k]:out[2 * k] + out[2 * k + 1]]) for k in range(min(n, 8))]


def product_ec_point_bits(buf: bytes, xbit: int, curve: int) -> bool:
    """k_ec_resolve's loader + curve equation (host build): X starts at BIT xbit of buf; curve 1..5 = P-256, P-384, P-521,
    P-224, secp192r1."""
    product_walk(b"\x30\x00")
    _walk.harness_ec_point_bits.argtypes = [C.c_char_p, C.c_uint32, C.c_uint64, C.c_int]
    return bool(_walk.harness_ec_point_bits(buf, len(buf), xbit, curve))


def product_walk_tbs(tbs: bytes, fill=0xA5) -> HarnessOut:
    """The product's walk over a bare TBSCertificate (strict_leaf)."""
    product_walk(b"\x30\x00")          # builds and binds the library
    _walk.harness_walk_tbs.argtypes = [C.c_char_p, C.c_uint32, C.c_uint8, C.POINTER(HarnessOut)]
    o = HarnessOut()
    _walk.harness_walk_tbs(tbs, len(tbs), fill, C.byref(o))
    return o


def product_name_strings(der: bytes, fill=0xA5) -> int:
    """strict_strings: 1 = the Names' string values keep to their types' character sets, 0 = a finding, -1 = no parse."""
    product_walk(b"\x30\x00")
    _walk.harness_name_strings.argtypes = [C.c_char_p, C.c_uint32, C.c_uint8]
    return _walk.harness_name_strings(der, len(der), fill)


def walk_touched(der: bytes, phase: int = 0, cn_filter: bytes = b"", reference_profile: bool = False):
    """(accepted, bytes the walk's reads cover, distinct 128-byte lines they lie in when the certificate starts at
    byte `phase` of a line) — bench.py's needed_bytes accounting.  reference_profile: the walk with strict_strings and
    strict_extensions (the subjectAltName's headers and the extension bodies are read)."""
    product_walk(b"\x30\x00")          # builds / loads the library
    _walk.harness_touched_profile(int(reference_profile))
    fn = _walk.harness_walk_touched
    fn.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32),
                   C.POINTER(C.c_uint32)]
    nb, nl = C.c_uint32(0), C.c_uint32(0)
    ok = fn(der, len(der), phase & 127, cn_filter, len(cn_filter), C.byref(nb), C.byref(nl))
    return bool(ok), nb.value, nl.value


def walk_window(der: bytes, phase: int = 0, wbytes: int = 224, strings: bool = False, ext: bool = False, skip: int = 0):
    """The walk through a SIMULATED per-lane window of `wbytes` bytes (kernels/readers.h geometry) for a certificate that
    starts at byte `phase` of its 128-byte line: (accepted, per-lane refills the hints asked for, reads that missed the
    window, position of the first refill, position of the first miss, cooperative refills of the subjectAltName walk,
    defer_exact calls)."""
    product_walk(b"\x30\x00")
    out = (C.c_uint32 * 6)()
    if skip:    # the first window begins `skip` octets in, the outer headers come from sixteen octets held apart (WinGeo::SKIP)
        fn = _walk.harness_walk_window_skip
        fn.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
        ok = fn(der, len(der), phase, wbytes, skip, int(strings), int(ext), out)
        return bool(ok), out[0], out[1], out[2], out[3], out[4], out[5]
    fn = _walk.harness_walk_window
    fn.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    ok = fn(der, len(der), phase, wbytes, int(strings), int(ext), out)
    return bool(ok), out[0], out[1], out[2], out[3], out[4], out[5]


def ossl_extract(der: bytes):
    global _ossl
    if _ossl is None:
        outp = os.path.join(HERE, "libossl_extract.so")
        srcp = os.path.join(HERE, "ossl_extract.c")
        if not os.path.exists(outp) or os.path.getmtime(outp) < os.path.getmtime(srcp):
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", srcp, "-o", outp, "-lcrypto"])
        _ossl = C.CDLL(outp)
        _ossl.ossl_extract.argtypes = [C.c_char_p, C.c_long, C.POINTER(OsslOut)]
        _ossl.ossl_pubkey_ok.argtypes = [C.c_char_p, C.c_long]
    o = OsslOut()
    if not _ossl.ossl_extract(der, len(der), C.byref(o)):
        return None
    return o


class OsslVerdict(C.Structure):
    _fields_ = [("stage", C.c_int), ("ext_nid", C.c_int), ("n_ext", C.c_int), ("reason", C.c_char * 96)]


def ossl_verdict(der: bytes) -> OsslVerdict:
    """OpenSSL's accept/reject opinion by stage: 0 everything decodes, 1 d2i_X509, 2 trailing bytes, 3 the public key,
    4 a validity time, 5 the body of an extension it has a decoder for (ext_nid)."""
    ossl_extract(b"\x30\x00")
    _ossl.ossl_verdict.argtypes = [C.c_char_p, C.c_long, C.POINTER(OsslVerdict)]
    v = OsslVerdict()
    _ossl.ossl_verdict(der, len(der), C.byref(v))
    return v


def ossl_pubkey_ok(der: bytes) -> int:
    """OpenSSL's opinion on the public key: 1 = certificate and key decode (X509_get_pubkey), 0 = the key does not,
    -1 = the certificate does not."""
    ossl_extract(b"\x30\x00")
    return _ossl.ossl_pubkey_ok(der, len(der))


_hk = None


def host_keys_lib():
    """Host build of csrc/host_keys.h: the wire form and the verdict of a group round's long-serial settlement."""
    global _hk
    if _hk is None:
        src = os.path.join(HERE, "host_keys_harness.cpp")
        dep = os.path.join(HERE, "..", "..", "ct_mapreduce_amd", "csrc", "host_keys.h")
        out = os.path.join(HERE, "libhost_keys_harness.so")
        if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(dep)):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", src, "-o", out])
        _hk = C.CDLL(out)
        _hk.harness_host_keys_verdict.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint64),
                                                  C.POINTER(C.c_uint8), C.c_uint32]
        _hk.harness_host_keys_append.argtypes = [C.c_uint64, C.c_int32, C.c_uint32, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint64),
                                                 C.c_uint32]
        _hk.harness_host_keys_append.restype = C.c_uint32
    return _hk


def build_fake_rccl() -> str:
    """The stand-in for librccl (fake_rccl.cpp: ranks as threads or processes on ONE GPU, bytes through POSIX shared
    memory), built on demand; hand the path to the library through CTMR_RCCL_LIB."""
    import subprocess
    src = os.path.join(HERE, "fake_rccl.cpp")
    out = os.path.join(HERE, "libfake_rccl.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                               src, "-o", out, "-L/opt/rocm/lib", "-lamdhip64", "-pthread", "-lrt"])
    return out
