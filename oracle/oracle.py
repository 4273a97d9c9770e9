"""ctypes binding of the CPU oracle (oracle/ctmr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by the product package ct_mapreduce_amd.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ST_PASS, ST_PARSE_ERROR, ST_FILTERED_CA, ST_FILTERED_EXPIRED, ST_FILTERED_CN, ST_NO_ISSUER, \
    ST_ISSUER_PARSE_ERROR, ST_ENTRY_DECODE_ERROR = range(8)


class Meta(C.Structure):
    _fields_ = [("issuer_off", C.c_uint32), ("issuer_len", C.c_uint32), ("n_crl", C.c_uint32),
                ("crl_off", C.c_uint32 * 16), ("crl_len", C.c_uint32 * 16), ("n_crl_ext", C.c_uint32),
                ("bad_crl", C.c_int32)]


class Entry(C.Structure):
    _fields_ = [("ok", C.c_int32), ("entry_type", C.c_int32), ("timestamp", C.c_uint64),
                ("cert_in_extra", C.c_int32), ("cert_off", C.c_uint32), ("cert_len", C.c_uint32),
                ("chain0_off", C.c_uint32), ("chain0_len", C.c_uint32), ("n_chain", C.c_uint32),
                ("tbs_off", C.c_uint32), ("tbs_len", C.c_uint32)]


class Cert(C.Structure):
    _fields_ = [("ok", C.c_int32), ("err_site", C.c_int32),
                ("serial_off", C.c_uint32), ("serial_len", C.c_uint32),
                ("not_before", C.c_int64), ("not_after", C.c_int64),
                ("cn_off", C.c_uint32), ("cn_len", C.c_uint32),
                ("bc_valid", C.c_int32), ("is_ca", C.c_int32),
                ("spki_off", C.c_uint32), ("spki_len", C.c_uint32),
                ("tbs_off", C.c_uint32), ("tbs_len", C.c_uint32), ("issuer_off", C.c_uint32), ("issuer_len", C.c_uint32),
                ("exts_off", C.c_uint32), ("exts_end", C.c_uint32), ("nonfatal", C.c_int32),
                ("string_findings", C.c_int32), ("spki_fatal", C.c_int32), ("spki_findings", C.c_int32),
                ("ext_fatal", C.c_uint32), ("ext_findings", C.c_int32), ("ext_string_findings", C.c_int32)]


NF_NEGATIVE_SERIAL, NF_LAX_INTEGER = 1, 2
SF_PRINTABLE, SF_NUMERIC, SF_IA5, SF_UTF8 = 1, 2, 4, 8
PK_RSA_PARAMS, PK_LAX_INTEGER, PK_RSA_MODULUS, PK_INSECURE_CURVE = 1, 2, 4, 8


def pem_encode(der: bytes) -> bytes:
    out = C.create_string_buffer(len(der) * 2 + 128)
    n = lib().orc_pem_encode(bytes(der), len(der), out)
    return out.raw[:n]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_parse_cert.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Cert)]
        L.orc_parse_tbs.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Cert)]
        L.orc_engine_set_strict_leaf.argtypes = [C.c_void_p, C.c_int]
        L.orc_engine_set_strict_strings.argtypes = [C.c_void_p, C.c_int]
        L.orc_engine_set_strict_spki.argtypes = [C.c_void_p, C.c_int]
        L.orc_engine_set_strict_extensions.argtypes = [C.c_void_p, C.c_int]
        L.orc_sha256.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.orc_b64url.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.orc_b64url.restype = C.c_size_t
        L.orc_issuer_id.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.orc_pem_encode.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.orc_pem_encode.restype = C.c_size_t
        L.orc_exp_hour.argtypes = [C.c_int64]
        L.orc_exp_hour.restype = C.c_int32
        L.orc_exp_date_id.argtypes = [C.c_int32, C.c_char_p]
        L.orc_day_id.argtypes = [C.c_int64, C.c_char_p]
        L.orc_cert_is_filtered_out.argtypes = [C.c_char_p, C.POINTER(Cert), C.c_char_p, C.c_size_t,
                                               C.c_int, C.c_int64]
        L.orc_engine_new.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int64]
        L.orc_engine_new.restype = C.c_void_p
        L.orc_engine_free.argtypes = [C.c_void_p]
        L.orc_engine_entry.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int32),
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
        L.orc_set_insert.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.orc_set_contains.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.orc_set_cardinality.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.orc_set_cardinality.restype = C.c_int64
        L.orc_key_count.argtypes = [C.c_void_p]
        L.orc_key_count.restype = C.c_int64
        L.orc_key_at.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_size_t]
        L.orc_key_at.restype = C.c_size_t
        L.orc_key_expiry.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int64)]
        L.orc_set_members.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.orc_set_members.restype = C.c_size_t
        L.orc_issuer_count.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_issuer_count.restype = C.c_int64
        L.orc_total_count.argtypes = [C.c_void_p]
        L.orc_total_count.restype = C.c_int64
        L.orc_inserted.argtypes = [C.c_void_p]
        L.orc_inserted.restype = C.c_int64
        L.orc_engine_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                       C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
        L.orc_cert_meta.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Meta)]
        L.orc_decode_entry.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(Entry)]
        L.orc_engine_raw_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def cert_meta(der: bytes):
    """(RawIssuer bytes, [CRL DP URI bytes…], Meta) or None when the certificate does not parse."""
    m = Meta()
    if not lib().orc_cert_meta(der, len(der), C.byref(m)):
        return None
    uris = [der[m.crl_off[k]:m.crl_off[k] + m.crl_len[k]] for k in range(min(m.n_crl, 16))]
    return der[m.issuer_off:m.issuer_off + m.issuer_len], uris, m


def decode_entry(leaf_input: bytes, extra_data: bytes) -> Entry:
    e = Entry()
    lib().orc_decode_entry(leaf_input, len(leaf_input), extra_data, len(extra_data), C.byref(e))
    return e


NF_SPKI = 8   # parsePublicKey filed a non-fatal finding (mirrors the product's WALK_NF_SPKI)


def _fold_spki(c: Cert, strict_spki: bool) -> Cert:
    """What an engine with strict_spki (the default) makes of parsePublicKey's verdict: a fatal error is a parse error,
    a finding one more non-fatal finding.  The C struct keeps the two apart (spki_fatal / spki_findings)."""
    if strict_spki:
        if c.ok and c.spki_fatal:
            c.ok, c.err_site = 0, c.spki_fatal
        if c.spki_findings:
            c.nonfatal |= NF_SPKI
    return c


def parse_cert(der: bytes, strict_spki: bool = True) -> Cert:
    c = Cert()
    lib().orc_parse_cert(der, len(der), C.byref(c))
    return _fold_spki(c, strict_spki)


def parse_tbs(tbs: bytes, strict_spki: bool = True) -> Cert:
    """A bare TBSCertificate (what ct.LogEntryFromLeaf parses of a precertificate entry's leaf)."""
    c = Cert()
    lib().orc_parse_tbs(tbs, len(tbs), C.byref(c))
    return _fold_spki(c, strict_spki)


def sha256(b: bytes) -> bytes:
    out = C.create_string_buffer(32)
    lib().orc_sha256(b, len(b), out)
    return out.raw


def b64url(b: bytes) -> str:
    out = C.create_string_buffer(4 * ((len(b) + 2) // 3) + 1)
    lib().orc_b64url(b, len(b), out)
    return out.value.decode()


def issuer_id(spki: bytes) -> str:
    out = C.create_string_buffer(45)
    lib().orc_issuer_id(spki, len(spki), out)
    return out.value.decode()


def exp_hour(unix: int) -> int:
    return lib().orc_exp_hour(unix)


def exp_date_id(hour: int) -> str:
    out = C.create_string_buffer(16)
    lib().orc_exp_date_id(hour, out)
    return out.value.decode()


def day_id(unix: int) -> str:
    out = C.create_string_buffer(16)
    lib().orc_day_id(unix, out)
    return out.value.decode()


def is_filtered_out(der: bytes, cert: Cert, filt: bytes, log_expired: bool, now: int) -> int:
    return lib().orc_cert_is_filtered_out(der, C.byref(cert), filt, len(filt), int(log_expired), now)


class Engine:
    """insertCTWorker loop + FilesystemDatabase.Store over an in-process set store."""

    def __init__(self, issuer_cn_filter: bytes = b"", log_expired: bool = False, now: int = 0):
        self._h = lib().orc_engine_new(issuer_cn_filter, len(issuer_cn_filter), int(log_expired), now)

    def set_profile(self, profile):
        """The four switches at once, like ctmr_set_profile: "reference" (the default: all on) or "fast" (strict_spki only)."""
        ref = profile in ("reference", 1)
        self.set_strict_spki(True)
        for f in (self.set_strict_leaf, self.set_strict_strings, self.set_strict_extensions):
            f(ref)

    def set_strict_leaf(self, on: bool):
        """Precertificate entries: fail the entry when its leaf TBSCertificate does not parse (LogEntryFromLeaf)."""
        lib().orc_engine_set_strict_leaf(self._h, int(bool(on)))

    def set_strict_strings(self, on: bool):
        """Character sets of the Names' string values (Go stdlib rules) as one more non-fatal finding; default on (the reference profile)."""
        lib().orc_engine_set_strict_strings(self._h, int(bool(on)))

    def set_strict_spki(self, on: bool):
        """parsePublicKey's verdict on the key inside subjectPublicKeyInfo (default ON, as in the reference)."""
        lib().orc_engine_set_strict_spki(self._h, int(bool(on)))

    def set_strict_extensions(self, on: bool):
        """The bodies of the extensions Go unmarshals by struct rules (keyUsage, key identifiers, extKeyUsage, policies, AIA)
        become a fatal parse error; on by default (the reference profile)."""
        lib().orc_engine_set_strict_extensions(self._h, int(bool(on)))

    def close(self):
        if self._h:
            lib().orc_engine_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def entry(self, leaf: bytes, issuer_der, entry_type: int = 0):
        unk = C.c_int(0)
        eh = C.c_int32(0)
        sp = C.c_void_p()
        sl = C.c_uint32(0)
        st = lib().orc_engine_entry(self._h, leaf, len(leaf), entry_type, issuer_der,
                                    len(issuer_der) if issuer_der is not None else 0,
                                    C.byref(unk), C.byref(eh), C.byref(sp), C.byref(sl))
        return st, bool(unk.value), eh.value

    def batch(self, payload, offsets, issuer_idx, issuer_payload, issuer_offsets, entry_type=None):
        """entry_type: u8[n] (0 X509, 1 precert) or None = all X509 entries."""
        import numpy as np
        n = len(offsets) - 1
        status = np.zeros(n, dtype=np.uint8)
        unknown = np.zeros(n, dtype=np.uint8)
        exp = np.zeros(n, dtype=np.int32)
        payload = np.ascontiguousarray(payload, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        issuer_idx = np.ascontiguousarray(issuer_idx, dtype=np.uint32)
        issuer_payload = np.ascontiguousarray(issuer_payload, dtype=np.uint8)
        issuer_offsets = np.ascontiguousarray(issuer_offsets, dtype=np.uint64)
        if entry_type is not None:
            entry_type = np.ascontiguousarray(entry_type, dtype=np.uint8)
        lib().orc_engine_batch(self._h, payload.ctypes.data, offsets.ctypes.data,
                               issuer_idx.ctypes.data, entry_type.ctypes.data if entry_type is not None else None,
                               n, issuer_payload.ctypes.data,
                               issuer_offsets.ctypes.data, len(issuer_offsets) - 1,
                               status.ctypes.data, unknown.ctypes.data, exp.ctypes.data)
        return status, unknown, exp

    def raw_batch(self, blob, bounds):
        import numpy as np
        n = (len(bounds) - 1) // 2
        status = np.zeros(n, dtype=np.uint8)
        unknown = np.zeros(n, dtype=np.uint8)
        exp = np.zeros(n, dtype=np.int32)
        ts = np.zeros(n, dtype=np.uint64)
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        bounds = np.ascontiguousarray(bounds, dtype=np.uint64)
        lib().orc_engine_raw_batch(self._h, blob.ctypes.data, bounds.ctypes.data, n, status.ctypes.data,
                                   unknown.ctypes.data, exp.ctypes.data, ts.ctypes.data)
        return status, unknown, exp, ts

    def set_insert(self, key: bytes, member: bytes) -> bool:
        return bool(lib().orc_set_insert(self._h, key, len(key), member, len(member)))

    def set_contains(self, key: bytes, member: bytes) -> bool:
        return bool(lib().orc_set_contains(self._h, key, len(key), member, len(member)))

    def set_cardinality(self, key: bytes) -> int:
        return lib().orc_set_cardinality(self._h, key, len(key))

    def keys(self):
        n = lib().orc_key_count(self._h)
        out = []
        buf = C.create_string_buffer(256)
        for i in range(n):
            l = lib().orc_key_at(self._h, i, buf, 256)
            out.append(buf.raw[:l])
        return out

    def key_expiry(self, key: bytes):
        t = C.c_int64(0)
        if lib().orc_key_expiry(self._h, key, len(key), C.byref(t)):
            return t.value
        return None

    def members(self, key: bytes):
        need = lib().orc_set_members(self._h, key, len(key), None, 0)
        buf = (C.c_uint8 * max(need, 1))()
        lib().orc_set_members(self._h, key, len(key), buf, need)
        raw = bytes(buf)[:need]
        out, o = [], 0
        while o < need:
            l = int.from_bytes(raw[o:o + 4], "little")
            out.append(raw[o + 4:o + 4 + l])
            o += 4 + l
        return out

    def issuer_count(self, issuer_id_str: str) -> int:
        return lib().orc_issuer_count(self._h, issuer_id_str.encode())

    def total_count(self) -> int:
        return lib().orc_total_count(self._h)

    def inserted(self) -> int:
        return lib().orc_inserted(self._h)
