"""Host mirror of the reference's per-entry path over the C ABI (include/ctmr.h).

Names follow the reference: `Engine.map_batch` is the batched body of insertCTWorker
(cmd/ct-fetch/ct-fetch.go:191-235) through FilesystemDatabase.Store's WasUnknown
(storage/filesystemdatabase.go:158-183); the set_* methods are storage.RemoteCache
(storage/types.go:83-102) on byte strings.
"""
import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import _native as N

RECORD_DTYPE = np.dtype([("status", "u1"), ("flags", "u1"), ("serial_len", "<u2"),
                         ("exp_hour", "<i4"), ("issuer_idx", "<u4"), ("serial", "u1", (20,))])
assert RECORD_DTYPE.itemsize == 32


class CtmrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ctmr error {code}: {msg}")
        self.code = code


@dataclass
class Batch:
    """Packed CT-entry batch (SURVEY.md §8(d) layout), host side."""
    payload: np.ndarray      # u8, leaf DER back to back
    offsets: np.ndarray      # u64[n+1]
    issuer_idx: np.ndarray   # u32[n], NO_ISSUER = chain empty
    entry_type: np.ndarray   # u8[n], 0 = X509, 1 = precert

    @property
    def n(self):
        return len(self.offsets) - 1

    def cert(self, i):
        return self.payload[int(self.offsets[i]):int(self.offsets[i + 1])].tobytes()

    @staticmethod
    def from_certs(certs, issuer_idx, entry_type=None):
        offs = np.zeros(len(certs) + 1, dtype=np.uint64)
        if certs:
            offs[1:] = np.cumsum([len(c) for c in certs], dtype=np.uint64)
        payload = np.frombuffer(b"".join(certs), dtype=np.uint8).copy() if certs else np.zeros(0, np.uint8)
        et = np.zeros(len(certs), np.uint8) if entry_type is None else np.asarray(entry_type, np.uint8)
        return Batch(payload, offs, np.asarray(issuer_idx, dtype=np.uint32), et)


@dataclass
class RawEntries:
    """Raw get-entries batch: blob = leaf_input_0 ‖ extra_data_0 ‖ leaf_input_1 ‖ …, bounds u64[2n+1] (include/ctmr.h)."""
    blob: np.ndarray         # u8
    bounds: np.ndarray       # u64[2n+1]

    @property
    def n(self):
        return (len(self.bounds) - 1) // 2

    def leaf_input(self, i):
        return self.blob[int(self.bounds[2 * i]):int(self.bounds[2 * i + 1])].tobytes()

    def extra_data(self, i):
        return self.blob[int(self.bounds[2 * i + 1]):int(self.bounds[2 * i + 2])].tobytes()

    @staticmethod
    def from_pairs(pairs):
        """pairs: [(leaf_input bytes, extra_data bytes)] — the base64-decoded members of a get-entries response."""
        parts, bounds, at = [], [0], 0
        for leaf, extra in pairs:
            for b in (leaf, extra):
                parts.append(bytes(b))
                at += len(b)
                bounds.append(at)
        blob = np.frombuffer(b"".join(parts), np.uint8).copy() if at else np.zeros(0, np.uint8)
        return RawEntries(blob, np.asarray(bounds, np.uint64))


@dataclass
class EntriesResult:
    records: np.ndarray      # RECORD_DTYPE[n]
    new_idx: np.ndarray      # u64[n_new], ascending
    timestamp: np.ndarray    # u64[n], ms
    stats: N.BatchStats
    decode: N.DecodeStats


@dataclass
class BatchResult:
    records: np.ndarray      # RECORD_DTYPE[n]
    new_idx: np.ndarray      # u64[n_new], ascending
    stats: N.BatchStats


class Engine:
    def __init__(self, device=0, table_slots=0, pair_slots=0, max_issuers=0, certs_per_tile=0,
                 lds_tile_bytes=0, map_variant=0, profile=False, collect_meta=False, max_table_slots=0):
        self._lib = N.lib()
        if not map_variant:   # kernel experiments (scripts/run.sh TAG lib:PATH STEP… with a sweep build of the library): whole test suites on another variant
            map_variant = int(os.environ.get("CTMR_MAP_VARIANT", "0"))
        cfg = N.Config(struct_size=C.sizeof(N.Config), device=device, table_slots=table_slots,
                       pair_slots=pair_slots, max_issuers=max_issuers, certs_per_tile=certs_per_tile,
                       lds_tile_bytes=lds_tile_bytes, map_variant=map_variant, profile=int(profile),
                       collect_meta=int(collect_meta), max_table_slots=max_table_slots)
        h = C.c_void_p()
        rc = self._lib.ctmr_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise CtmrError(rc, "ctmr_create failed (no usable HIP device? there is no CPU fallback)")
        self._h = h
        self.collect_meta = bool(collect_meta)

    # ---- lifecycle
    def close(self):
        if getattr(self, "_h", None):
            self.free_pinned()
            self._lib.ctmr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise CtmrError(rc, self._lib.ctmr_last_error(self._h).decode(errors="replace"))

    @property
    def handle(self):
        return self._h

    def set_stream(self, hip_stream):
        self._ck(self._lib.ctmr_set_stream(self._h, C.c_void_p(hip_stream)))

    def synchronize(self):
        self._ck(self._lib.ctmr_synchronize(self._h))

    # ---- page-locked host buffers for the host-buffer entry points
    def pinned_array(self, nbytes):
        """uint8 numpy array over hipHostMalloc memory (freed when the array's owner object is collected)."""
        p = C.c_void_p()
        self._ck(self._lib.ctmr_alloc_pinned(self._h, nbytes, C.byref(p)))
        buf = (C.c_uint8 * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=np.uint8, count=nbytes)
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p.value)
        return arr

    def free_pinned(self):
        for p in getattr(self, "_pinned", []):
            self._lib.ctmr_free_pinned(self._h, C.c_void_p(p))
        self._pinned = []

    # ---- issuers / filter
    def add_issuers(self, certs):
        blob = b"".join(certs)
        offs = np.zeros(len(certs) + 1, dtype=np.uint64)
        if certs:
            offs[1:] = np.cumsum([len(c) for c in certs], dtype=np.uint64)
        first = C.c_uint32()
        buf = np.frombuffer(blob, dtype=np.uint8) if blob else np.zeros(1, np.uint8)
        self._ck(self._lib.ctmr_add_issuers(self._h, buf.ctypes.data, offs.ctypes.data, len(certs),
                                            C.byref(first)))
        return first.value

    def issuer_count(self):
        n = C.c_uint32()
        self._ck(self._lib.ctmr_issuer_count(self._h, C.byref(n)))
        return n.value

    def issuer_info(self, idx):
        info = N.IssuerInfo()
        self._ck(self._lib.ctmr_issuer_info_get(self._h, idx, C.byref(info)))
        return info

    def issuer_id(self, idx):
        return self.issuer_info(idx).issuer_id.decode()

    def set_filter(self, issuer_cn_filter=b"", log_expired=False, now=0):
        if isinstance(issuer_cn_filter, str):
            issuer_cn_filter = issuer_cn_filter.encode()
        self._ck(self._lib.ctmr_set_filter(self._h, issuer_cn_filter, len(issuer_cn_filter),
                                           int(log_expired), int(now)))

    # ---- the batched map + reduce
    def map_batch(self, batch: Batch, want_records=True, want_new=True) -> BatchResult:
        n = batch.n
        payload = np.ascontiguousarray(batch.payload, dtype=np.uint8)
        if payload.size == 0:
            payload = np.zeros(1, np.uint8)
        offsets = np.ascontiguousarray(batch.offsets, dtype=np.uint64)
        iss = np.ascontiguousarray(batch.issuer_idx, dtype=np.uint32)
        et = np.ascontiguousarray(batch.entry_type, dtype=np.uint8)
        records = np.zeros(n, dtype=RECORD_DTYPE)
        new_idx = np.zeros(max(n, 1), dtype=np.uint64)
        st = N.BatchStats()
        self._ck(self._lib.ctmr_map_batch(
            self._h, payload.ctypes.data, offsets.ctypes.data, iss.ctypes.data if n else None,
            et.ctypes.data if n else None, n, records.ctypes.data if (want_records and n) else None,
            new_idx.ctypes.data if (want_new and n) else None, C.byref(st)))
        return BatchResult(records, new_idx[:st.n_new] if want_new else new_idx[:0], st)

    # ---- asynchronous host ingestion (ctmr_submit_batch / ctmr_wait / ctmr_flush)
    def submit_batch(self, payload, offsets, issuer_idx, entry_type, n) -> int:
        """Arrays (numpy, contiguous) or raw addresses (ints: e.g. pinned buffers of alloc_pinned).  Returns the ticket.
        The caller keeps the arrays alive — and pinned payloads untouched — until wait(ticket)."""
        def addr(a):
            return a.ctypes.data if hasattr(a, "ctypes") else a
        t = C.c_uint64(0)
        self._ck(self._lib.ctmr_submit_batch(self._h, addr(payload), addr(offsets), addr(issuer_idx),
                                             addr(entry_type) if entry_type is not None else None, n, C.byref(t)))
        return t.value

    def flush(self):
        self._ck(self._lib.ctmr_flush(self._h))

    def wait(self, ticket: int, n: int, want_records=True, want_new=True) -> BatchResult:
        records = np.zeros(n, dtype=RECORD_DTYPE)
        new_idx = np.zeros(max(n, 1), dtype=np.uint64)
        st = N.BatchStats()
        self._ck(self._lib.ctmr_wait(self._h, ticket, records.ctypes.data if (want_records and n) else None,
                                     new_idx.ctypes.data if (want_new and n) else None, C.byref(st)))
        return BatchResult(records, new_idx[:st.n_new] if want_new else new_idx[:0], st)

    def submit_entries(self, blob, bounds, n) -> int:
        """Raw get-entries form of submit_batch (ctmr_submit_entries): blob u8, bounds u64[2n+1] — arrays or addresses."""
        def addr(a):
            return a.ctypes.data if hasattr(a, "ctypes") else a
        t = C.c_uint64(0)
        self._ck(self._lib.ctmr_submit_entries(self._h, addr(blob), addr(bounds), n, C.byref(t)))
        return t.value

    def wait_entries(self, ticket: int, n: int) -> EntriesResult:
        records = np.zeros(n, dtype=RECORD_DTYPE)
        new_idx = np.zeros(max(n, 1), dtype=np.uint64)
        ts = np.zeros(max(n, 1), dtype=np.uint64)
        st, ds = N.BatchStats(), N.DecodeStats()
        self._ck(self._lib.ctmr_wait_entries(self._h, ticket, records.ctypes.data if n else None,
                                             new_idx.ctypes.data if n else None, ts.ctypes.data if n else None,
                                             C.byref(ds), C.byref(st)))
        return EntriesResult(records, new_idx[:st.n_new], ts[:n], st, ds)

    def map_batch_device(self, d_payload, d_offsets, d_issuer_idx, d_entry_type, n, d_records=0,
                         d_new_idx=0) -> N.BatchStats:
        """All pointers are device addresses (ints), e.g. torch tensors' data_ptr()."""
        st = N.BatchStats()
        self._ck(self._lib.ctmr_map_batch_device(
            self._h, C.c_void_p(d_payload), C.c_void_p(d_offsets), C.c_void_p(d_issuer_idx),
            C.c_void_p(d_entry_type) if d_entry_type else None, n,
            C.c_void_p(d_records) if d_records else None,
            C.c_void_p(d_new_idx) if d_new_idx else None, C.byref(st)))
        return st

    # ---- raw get-entries input (N2): ct.LogEntryFromLeaf + the choice of certificate and Chain[0] on the GPU
    def map_entries(self, raw: RawEntries) -> EntriesResult:
        """Downloader decode (ct-fetch.go:452) + insertCTWorker (:191-235) over raw entries.  Chain[0] certificates
        are registered as issuers by the call; entries LogEntryFromLeaf rejects get ST_ENTRY_DECODE_ERROR."""
        n = raw.n
        blob = np.ascontiguousarray(raw.blob, dtype=np.uint8)
        if blob.size == 0:
            blob = np.zeros(1, np.uint8)
        bounds = np.ascontiguousarray(raw.bounds, dtype=np.uint64)
        records = np.zeros(n, dtype=RECORD_DTYPE)
        new_idx = np.zeros(max(n, 1), dtype=np.uint64)
        ts = np.zeros(max(n, 1), dtype=np.uint64)
        st, ds = N.BatchStats(), N.DecodeStats()
        self._ck(self._lib.ctmr_map_entries(self._h, blob.ctypes.data, bounds.ctypes.data, n,
                                            records.ctypes.data if n else None, new_idx.ctypes.data if n else None,
                                            ts.ctypes.data if n else None, C.byref(ds), C.byref(st)))
        return EntriesResult(records, new_idx[:st.n_new], ts[:n], st, ds)

    def decode_entries_device(self, d_blob, d_bounds, n, view: N.EntryView) -> N.DecodeStats:
        ds = N.DecodeStats()
        self._ck(self._lib.ctmr_decode_entries_device(self._h, C.c_void_p(d_blob), C.c_void_p(d_bounds), n,
                                                      C.byref(view), C.byref(ds)))
        return ds

    def map_view_device(self, d_blob, blob_bytes, view: N.EntryView, n, d_records=0, d_new_idx=0) -> N.BatchStats:
        st = N.BatchStats()
        self._ck(self._lib.ctmr_map_view_device(self._h, C.c_void_p(d_blob), blob_bytes, C.byref(view), n,
                                                C.c_void_p(d_records) if d_records else None,
                                                C.c_void_p(d_new_idx) if d_new_idx else None, C.byref(st)))
        return st

    def map_entries_device(self, d_blob, d_bounds, n, d_records=0, d_new_idx=0, d_timestamp=0):
        st, ds = N.BatchStats(), N.DecodeStats()
        self._ck(self._lib.ctmr_map_entries_device(
            self._h, C.c_void_p(d_blob), C.c_void_p(d_bounds), n, C.c_void_p(d_records) if d_records else None,
            C.c_void_p(d_new_idx) if d_new_idx else None, C.c_void_p(d_timestamp) if d_timestamp else None,
            C.byref(ds), C.byref(st)))
        return st, ds

    def synth_entries_device(self, cfg: N.SynthConfig, first, n, d_bounds, d_blob, blob_cap) -> int:
        out = C.c_uint64()
        self._ck(self._lib.ctmr_synth_entries_device(self._h, C.byref(cfg), first, n, C.c_void_p(d_bounds),
                                                     C.c_void_p(d_blob) if d_blob else None, blob_cap, C.byref(out)))
        return out.value

    # ---- IssuerMetadata on device (N3): first sightings among the new certificates of the last host batch
    def meta_new(self):
        """[(kind, entry, issuer_idx, exp_hour, bytes)] — kind N.MK_EXPDATE / MK_CRL (URI bytes) / MK_DN (issuer Name
        TLV) / MK_HOST (parse certificate `entry` on the host).  Needs collect_meta=True."""
        ni, need = C.c_uint64(0), C.c_size_t(0)
        rc = self._lib.ctmr_meta_new(self._h, None, 0, None, 0, C.byref(ni), C.byref(need))
        if rc not in (0, N.E_RANGE):
            self._ck(rc)
        if ni.value == 0:
            return []
        items = (N.MetaItem * ni.value)()
        buf = np.zeros(max(need.value, 1), np.uint8)
        self._ck(self._lib.ctmr_meta_new(self._h, items, ni.value, buf.ctypes.data, need.value, C.byref(ni),
                                         C.byref(need)))
        raw, at, out = buf.tobytes(), 0, []
        for it in items:
            out.append((it.kind, int(it.entry), it.issuer_idx, it.exp_hour, raw[at:at + it.len]))
            at += it.len
        return out

    def meta_new_device(self, d_payload, d_offsets, d_ends, d_records, d_new_idx, n_new, d_items, items_cap) -> int:
        n = C.c_uint64(0)
        self._ck(self._lib.ctmr_meta_new_device(
            self._h, C.c_void_p(d_payload), C.c_void_p(d_offsets), C.c_void_p(d_ends) if d_ends else None,
            C.c_void_p(d_records), C.c_void_p(d_new_idx), n_new, C.c_void_p(d_items), items_cap, C.byref(n)))
        return n.value

    def meta_reset(self):
        self._ck(self._lib.ctmr_meta_reset(self._h))

    # ---- whole-certificate SHA-256 (auxiliary: not on the reference's path)
    def fingerprint_device(self, d_payload, d_offsets, d_ends, n, d_digests) -> float:
        """n × 32-byte SHA-256 digests of the certificates into d_digests; returns the kernel time in ms."""
        ms = C.c_float(0)
        self._ck(self._lib.ctmr_fingerprint_device(self._h, C.c_void_p(d_payload), C.c_void_p(d_offsets),
                                                   C.c_void_p(d_ends) if d_ends else None, n, C.c_void_p(d_digests),
                                                   C.byref(ms)))
        return ms.value

    # ---- PEM write-back (N1): pem.EncodeToMemory of the newly unknown certificates, on the GPU
    def pem_new(self):
        """PEM blocks (bytes) of every WAS_UNKNOWN entry of the last map_batch(), ascending entry order."""
        need, count = C.c_size_t(0), C.c_uint64(0)
        rc = self._lib.ctmr_pem_new(self._h, None, 0, None, C.byref(need), C.byref(count))
        if rc not in (0, N.E_RANGE):
            self._ck(rc)
        if count.value == 0:
            return []
        out = np.zeros(need.value, np.uint8)
        offs = np.zeros(count.value + 1, np.uint64)
        self._ck(self._lib.ctmr_pem_new(self._h, out.ctypes.data, out.size, offs.ctypes.data, C.byref(need),
                                        C.byref(count)))
        raw = out.tobytes()
        return [raw[int(offs[k]):int(offs[k + 1])] for k in range(count.value)]

    def pem_encode_device(self, d_payload, d_offsets, d_idx, n_idx, d_pem, pem_cap, d_pem_offsets) -> int:
        total = C.c_uint64(0)
        self._ck(self._lib.ctmr_pem_encode_device(
            self._h, C.c_void_p(d_payload), C.c_void_p(d_offsets), C.c_void_p(d_idx), n_idx,
            C.c_void_p(d_pem) if d_pem else None, pem_cap, C.c_void_p(d_pem_offsets), C.byref(total)))
        return int(total.value)

    def pem_encode_view_device(self, d_blob, view: N.EntryView, d_idx, n_idx, d_pem, pem_cap, d_pem_offsets) -> int:
        total = C.c_uint64(0)
        self._ck(self._lib.ctmr_pem_encode_view_device(
            self._h, C.c_void_p(d_blob), C.byref(view), C.c_void_p(d_idx), n_idx,
            C.c_void_p(d_pem) if d_pem else None, pem_cap, C.c_void_p(d_pem_offsets), C.byref(total)))
        return int(total.value)

    # ---- cross-GPU key exchange, owner-computes (ctmr.h: ctmr_xchg_*): one round on this rank, device pointers as ints.
    # ctmr_mapreduce_amd.distributed.Group drives whole rounds natively; these are the per-rank steps for a host with a
    # transport of its own, and for tests.
    KEY_BYTES = 32          # a key record (serial of up to 20 octets)
    KEY_BYTES_LONG = 64     # … of 21..40 octets

    def xchg_map(self, shard: N.Shard, world: int, rank: int, ord_base: int):
        """→ (records per owner, number of 64-byte records)."""
        counts = (C.c_uint64 * world)()
        n_long = C.c_uint64(0)
        self._ck(self._lib.ctmr_xchg_map_device(self._h, C.byref(shard), world, rank, ord_base, counts, C.byref(n_long)))
        return [int(c) for c in counts], int(n_long.value)

    def xchg_keys(self, world: int, d_keys32_out=0, d_keys64_out=0):
        """Partitioned key records into the caller's buffers → 64-byte records per owner."""
        counts64 = (C.c_uint64 * world)()
        self._ck(self._lib.ctmr_xchg_keys_device(self._h, C.c_void_p(d_keys32_out) if d_keys32_out else None,
                                                 C.c_void_p(d_keys64_out) if d_keys64_out else None, counts64))
        return [int(c) for c in counts64]

    def xchg_insert(self, d_keys32=0, n32=0, d_flags32=0, d_keys64=0, n64=0, d_flags64=0):
        self._ck(self._lib.ctmr_xchg_insert_device(
            self._h, C.c_void_p(d_keys32) if n32 else None, n32, C.c_void_p(d_keys64) if n64 else None, n64,
            C.c_void_p(d_flags32) if n32 else None, C.c_void_p(d_flags64) if n64 else None))

    def xchg_apply(self, d_sent32=0, d_flags32=0, n32=0, d_sent64=0, d_flags64=0, n64=0) -> N.BatchStats:
        st = N.BatchStats()
        self._ck(self._lib.ctmr_xchg_apply_device(
            self._h, C.c_void_p(d_sent32) if n32 else None, C.c_void_p(d_flags32) if n32 else None, n32,
            C.c_void_p(d_sent64) if n64 else None, C.c_void_p(d_flags64) if n64 else None, n64, C.byref(st)))
        return st

    def set_issuer_autoregister(self, on: bool):
        self._ck(self._lib.ctmr_set_issuer_autoregister(self._h, int(bool(on))))

    def set_chain0_match(self, mode: int):
        """N.CHAIN0_EXACT (default: every byte of every Chain[0] compared) or N.CHAIN0_TRUSTED_LOG (compared on the
        first sighting per call, then identified by length + first/last 16 bytes) — include/ctmr.h."""
        self._ck(self._lib.ctmr_set_chain0_match(self._h, int(mode)))

    def set_strict_extensions(self, on: bool):
        """The bodies of the extensions Go unmarshals by plain struct rules become a fatal parse error (include/ctmr.h); off
        by default."""
        self._ck(self._lib.ctmr_set_strict_extensions(self._h, int(bool(on))))

    def set_profile(self, profile):
        """ctmr_set_profile: "fast" (the defaults) or "reference" (strict_spki + strict_leaf + strict_strings +
        strict_extensions: what the reference does, as far as it can be known here).  Before the issuers are registered."""
        p = {"fast": N.PROFILE_FAST, "reference": N.PROFILE_REFERENCE}.get(profile, profile)
        self._ck(self._lib.ctmr_set_profile(self._h, int(p)))

    def set_strict_leaf(self, on: bool):
        """Walk the leaf TBSCertificate of precertificate entries as ct.LogEntryFromLeaf does (include/ctmr.h); default off."""
        self._ck(self._lib.ctmr_set_strict_leaf(self._h, int(bool(on))))

    def set_strict_spki(self, on: bool):
        """parsePublicKey's verdict on the key inside subjectPublicKeyInfo (ctmr_set_strict_spki): ON by default."""
        self._ck(self._lib.ctmr_set_strict_spki(self._h, int(bool(on))))

    def set_strict_strings(self, on: bool):
        """Go-stdlib character-set rules for the string values of both Names, filed as a non-fatal finding (include/ctmr.h);
        default off.  Set it before registering issuers."""
        self._ck(self._lib.ctmr_set_strict_strings(self._h, int(bool(on))))

    def pending_issuers(self):
        """Distinct Chain[0] certificates the last raw-entry call found unregistered (auto-registration off)."""
        need, cnt = C.c_size_t(), C.c_uint64()
        rc = self._lib.ctmr_pending_issuers(self._h, None, 0, C.byref(need), C.byref(cnt))
        if rc not in (0, N.E_RANGE):
            self._ck(rc)
        if cnt.value == 0:
            return []
        buf = (C.c_uint8 * need.value)()
        self._ck(self._lib.ctmr_pending_issuers(self._h, buf, need.value, C.byref(need), C.byref(cnt)))
        raw, out, o = bytes(buf), [], 0
        while o < len(raw):
            l = int.from_bytes(raw[o:o + 4], "little")
            out.append(raw[o + 4:o + 4 + l])
            o += 4 + l
        return out

    # ---- cross-GPU global dedup, Bloom pre-filter variant (ctmr.h: ctmr_bloom_*), device pointers as ints
    def bloom_config(self, bits: int, d_words=0):
        """d_words: caller-owned device buffer of bits/8 bytes (0 = the library allocates the filter)."""
        self._ck(self._lib.ctmr_bloom_config(self._h, bits, C.c_void_p(d_words) if d_words else None))

    def bloom_device(self):
        """(device pointer, n_words) of this rank's cumulative filter (u64 words)."""
        p, nw = C.c_void_p(), C.c_uint64()
        self._ck(self._lib.ctmr_bloom_device(self._h, C.byref(p), C.byref(nw)))
        return int(p.value), int(nw.value)

    def bloom_add(self, d_payload, d_offsets, d_ends, n, d_records=0):
        self._ck(self._lib.ctmr_bloom_add_device(
            self._h, C.c_void_p(d_payload) if n else None, C.c_void_p(d_offsets) if n else None,
            C.c_void_p(d_ends) if d_ends else None, n, C.c_void_p(d_records) if d_records else None))

    def bloom_probe(self, d_payload, d_offsets, d_ends, n, d_records, d_filters, world, rank, order_base,
                    d_keys_out, keys_cap):
        """→ (counts per peer, fits): fits False = keys_cap too small, nothing written, call again."""
        counts = (C.c_uint64 * world)()
        rc = self._lib.ctmr_bloom_probe_device(
            self._h, C.c_void_p(d_payload) if n else None, C.c_void_p(d_offsets) if n else None,
            C.c_void_p(d_ends) if d_ends else None, n, C.c_void_p(d_records) if d_records else None,
            C.c_void_p(d_filters), world, rank, order_base, C.c_void_p(d_keys_out) if keys_cap else None,
            keys_cap, counts)
        if rc == N.E_RANGE:
            return [int(c) for c in counts], False
        self._ck(rc)
        return [int(c) for c in counts], True

    def bloom_lookup(self, d_keys, n_keys, order_base, d_flags):
        self._ck(self._lib.ctmr_bloom_lookup_device(self._h, C.c_void_p(d_keys) if n_keys else None, n_keys,
                                                    order_base, C.c_void_p(d_flags) if n_keys else None))

    def bloom_apply(self, d_records, n, d_keys_sent, d_flags, n_keys, d_new_idx=0) -> N.BatchStats:
        st = N.BatchStats()
        self._ck(self._lib.ctmr_bloom_apply_device(
            self._h, C.c_void_p(d_records) if d_records else None, n,
            C.c_void_p(d_keys_sent) if n_keys else None, C.c_void_p(d_flags) if n_keys else None, n_keys,
            C.c_void_p(d_new_idx) if d_new_idx else None, C.byref(st)))
        return st

    # ---- storage.RemoteCache set methods (storage/types.go:83-102)
    @staticmethod
    def _b(x):
        return x.encode() if isinstance(x, str) else bytes(x)

    def set_insert(self, key, member) -> bool:
        key, member = self._b(key), self._b(member)
        out = C.c_int()
        self._ck(self._lib.ctmr_set_insert(self._h, key, len(key), member, len(member), C.byref(out)))
        return bool(out.value)

    def set_contains(self, key, member) -> bool:
        key, member = self._b(key), self._b(member)
        out = C.c_int()
        self._ck(self._lib.ctmr_set_contains(self._h, key, len(key), member, len(member), C.byref(out)))
        return bool(out.value)

    def set_remove(self, key, member) -> bool:
        key, member = self._b(key), self._b(member)
        out = C.c_int()
        self._ck(self._lib.ctmr_set_remove(self._h, key, len(key), member, len(member), C.byref(out)))
        return bool(out.value)

    def set_cardinality(self, key) -> int:
        key = self._b(key)
        out = C.c_int64()
        self._ck(self._lib.ctmr_set_cardinality(self._h, key, len(key), C.byref(out)))
        return out.value

    def exists(self, key) -> bool:
        key = self._b(key)
        out = C.c_int()
        self._ck(self._lib.ctmr_exists(self._h, key, len(key), C.byref(out)))
        return bool(out.value)

    def _listing(self, fn, arg):
        need = C.c_size_t()
        cnt = C.c_uint64()
        rc = fn(self._h, arg, len(arg), None, 0, C.byref(need), C.byref(cnt))
        if rc not in (0, N.E_RANGE):
            self._ck(rc)
        buf = (C.c_uint8 * max(need.value, 1))()
        self._ck(fn(self._h, arg, len(arg), buf, need.value, C.byref(need), C.byref(cnt)))
        raw = bytes(buf)[:need.value]
        out, o = [], 0
        while o < len(raw):
            l = int.from_bytes(raw[o:o + 4], "little")
            out.append(raw[o + 4:o + 4 + l])
            o += 4 + l
        return out

    def set_list(self, key):
        """SetList / SetToChan: members, sorted bytewise."""
        return self._listing(self._lib.ctmr_set_members, self._b(key))

    def keys(self, pattern=b"*"):
        """KeysToChan(pattern)."""
        return self._listing(self._lib.ctmr_keys, self._b(pattern))

    def expire_at(self, key, unix_seconds):
        key = self._b(key)
        self._ck(self._lib.ctmr_expire_at(self._h, key, len(key), int(unix_seconds)))

    def expire_sweep(self, now) -> int:
        out = C.c_uint64()
        self._ck(self._lib.ctmr_expire_sweep(self._h, int(now), C.byref(out)))
        return out.value

    # ---- counts (cmd/storage-statistics/storage-statistics.go:44-53)
    def issuer_counts(self):
        n = self.issuer_count()
        out = np.zeros(max(n, 1), dtype=np.uint64)
        self._ck(self._lib.ctmr_issuer_counts(self._h, out.ctypes.data, n))
        return out[:n]

    def total_count(self) -> int:
        out = C.c_uint64()
        self._ck(self._lib.ctmr_total_count(self._h, C.byref(out)))
        return out.value

    def issuer_counts_device(self):
        p = C.c_void_p()
        n = C.c_uint32()
        self._ck(self._lib.ctmr_issuer_counts_device(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def reset_known(self):
        self._ck(self._lib.ctmr_reset_known(self._h))

    def table_info(self) -> N.TableInfo:
        """How full the known-certificate table is: index slots / occupied, arena cells / used, rebuilds, compactions."""
        out = N.TableInfo()
        self._ck(self._lib.ctmr_table_info_get(self._h, C.byref(out)))
        return out

    # ---- synthetic input (bench / tests)
    def synth_view_device(self, cfg: N.SynthConfig, first, n, align, d_starts, d_ends, d_payload, payload_cap,
                          d_issuer_idx, d_entry_type) -> int:
        """The synthetic certificates as an entry view, every certificate at a multiple of `align` bytes."""
        out = C.c_uint64()
        self._ck(self._lib.ctmr_synth_view_device(
            self._h, C.byref(cfg), first, n, align, C.c_void_p(d_starts), C.c_void_p(d_ends),
            C.c_void_p(d_payload) if d_payload else None, payload_cap,
            C.c_void_p(d_issuer_idx) if d_issuer_idx else None,
            C.c_void_p(d_entry_type) if d_entry_type else None, C.byref(out)))
        return out.value

    def synth_device(self, cfg: N.SynthConfig, first, n, d_offsets, d_payload, payload_cap,
                     d_issuer_idx, d_entry_type) -> int:
        out = C.c_uint64()
        self._ck(self._lib.ctmr_synth_device(
            self._h, C.byref(cfg), first, n, C.c_void_p(d_offsets),
            C.c_void_p(d_payload) if d_payload else None, payload_cap,
            C.c_void_p(d_issuer_idx) if d_issuer_idx else None,
            C.c_void_p(d_entry_type) if d_entry_type else None, C.byref(out)))
        return out.value
