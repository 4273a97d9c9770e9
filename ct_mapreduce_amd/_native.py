"""ctypes binding of libctmr.so — exactly the symbols include/ctmr.h declares.

The library is the product; this module only loads it.  There is no Python/CPU fallback: if the
shared object is missing, or no HIP device is usable, the failure is raised to the caller.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# CTMR_LIB: scripts/sweep.py points this at libctmr_sweep.so (the product + the two baseline map designs)
LIB_PATH = os.environ.get("CTMR_LIB") or os.path.join(HERE, "libctmr.so")

ST_PASS, ST_PARSE_ERROR, ST_FILTERED_CA, ST_FILTERED_EXPIRED, ST_FILTERED_CN, ST_NO_ISSUER, \
    ST_ISSUER_PARSE_ERROR, ST_ENTRY_DECODE_ERROR = range(8)
ST_COUNT = 8
ABI_VERSION = 7
CHAIN0_EXACT, CHAIN0_TRUSTED_LOG = 0, 1
PROFILE_FAST, PROFILE_REFERENCE = 0, 1
ENTRY_INVALID = 0xFF
FL_PRECERT, FL_WAS_UNKNOWN, FL_LONG_SERIAL = 1, 2, 4
NO_ISSUER = 0xFFFFFFFF
PAYLOAD_PAD = 32
MAX_SERIAL = 40
E_INVAL, E_HIP, E_NOMEM, E_FULL, E_NOTFOUND, E_RANGE = -1, -2, -3, -4, -5, -6


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("table_slots", C.c_uint64),
                ("pair_slots", C.c_uint64), ("max_issuers", C.c_uint32), ("certs_per_tile", C.c_uint32),
                ("lds_tile_bytes", C.c_uint32), ("map_variant", C.c_uint32), ("profile", C.c_uint32),
                ("collect_meta", C.c_uint32), ("max_table_slots", C.c_uint64)]


class BatchStats(C.Structure):
    _fields_ = [("n", C.c_uint64), ("by_status", C.c_uint64 * ST_COUNT), ("n_new", C.c_uint64),
                ("n_dup", C.c_uint64), ("n_host_set", C.c_uint64), ("payload_bytes", C.c_uint64),
                ("ms_map", C.c_float), ("ms_insert", C.c_float), ("ms_resolve", C.c_float),
                ("ms_compact", C.c_float), ("ms_total", C.c_float), ("map_launches", C.c_uint32)]


class EntryView(C.Structure):
    _fields_ = [("cert_start", C.c_void_p), ("cert_end", C.c_void_p), ("issuer_idx", C.c_void_p),
                ("entry_type", C.c_void_p), ("timestamp", C.c_void_p), ("chain0_start", C.c_void_p),
                ("chain0_len", C.c_void_p)]


class DecodeStats(C.Structure):
    _fields_ = [("n", C.c_uint64), ("n_x509", C.c_uint64), ("n_precert", C.c_uint64),
                ("n_decode_error", C.c_uint64), ("n_no_chain", C.c_uint64), ("n_issuers_added", C.c_uint64),
                ("blob_bytes", C.c_uint64), ("ms_decode", C.c_float), ("ms_match", C.c_float)]


MK_EXPDATE, MK_CRL, MK_DN, MK_HOST = range(4)
DEDUP_LOCAL, DEDUP_OWNER, DEDUP_BLOOM = range(3)
TRANSPORT_LOCAL, TRANSPORT_RCCL = range(2)
GROUP_ID_BYTES = 128


class Shard(C.Structure):
    _fields_ = [("d_payload", C.c_void_p), ("d_offsets", C.c_void_p), ("d_ends", C.c_void_p),
                ("d_issuer_idx", C.c_void_p), ("d_entry_type", C.c_void_p), ("n", C.c_uint64),
                ("blob_bytes", C.c_uint64), ("order_base", C.c_uint64), ("d_records", C.c_void_p),
                ("d_new_idx", C.c_void_p)]


class GroupStats(C.Structure):
    _fields_ = [("world", C.c_uint32), ("n_local", C.c_uint32), ("transport", C.c_uint32),
                ("first_local_rank", C.c_uint32), ("keys_sent", C.c_uint64), ("keys_received", C.c_uint64),
                ("filter_bytes_received", C.c_uint64), ("wire_bytes_sent", C.c_uint64), ("ms_phase", C.c_float * 8)]


class MetaItem(C.Structure):
    _fields_ = [("entry", C.c_uint64), ("kind", C.c_uint32), ("issuer_idx", C.c_uint32), ("exp_hour", C.c_int32),
                ("off", C.c_uint32), ("len", C.c_uint32), ("pad", C.c_uint32)]


class IssuerInfo(C.Structure):
    _fields_ = [("valid", C.c_int32), ("canonical_idx", C.c_uint32), ("spki_sha256", C.c_uint8 * 32),
                ("issuer_id", C.c_char * 48)]


class TableInfo(C.Structure):
    _fields_ = [("slots", C.c_uint64), ("occupied", C.c_uint64), ("arena_cells", C.c_uint64), ("arena_used", C.c_uint64),
                ("rebuilds", C.c_uint64), ("arena_compactions", C.c_uint64), ("arena_growths", C.c_uint64),
                ("reserved", C.c_uint64)]


class SynthConfig(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_issuers", C.c_uint32), ("zipf", C.c_uint32),
                ("dup_permille", C.c_uint32), ("ca_permille", C.c_uint32),
                ("expired_permille", C.c_uint32), ("mean_len", C.c_uint32), ("base_time", C.c_int64),
                ("profile", C.c_uint32), ("reserved", C.c_uint32)]


# name → (restype, argtypes); must list every function include/ctmr.h declares
_P = C.c_void_p
SIGNATURES = {
    "ctmr_abi_version": (C.c_int, []),
    "ctmr_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "ctmr_destroy": (None, [_P]),
    "ctmr_last_error": (C.c_char_p, [_P]),
    "ctmr_set_stream": (C.c_int, [_P, _P]),
    "ctmr_synchronize": (C.c_int, [_P]),
    "ctmr_alloc_pinned": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "ctmr_free_pinned": (C.c_int, [_P, _P]),
    "ctmr_add_issuers": (C.c_int, [_P, _P, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    "ctmr_sha256": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.c_char_p]),
    "ctmr_issuer_count": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "ctmr_issuer_info_get": (C.c_int, [_P, C.c_uint32, C.POINTER(IssuerInfo)]),
    "ctmr_set_filter": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.c_int, C.c_int64]),
    "ctmr_map_batch": (C.c_int, [_P, _P, _P, _P, _P, C.c_uint64, _P, _P, C.POINTER(BatchStats)]),
    "ctmr_map_batch_device": (C.c_int, [_P, _P, _P, _P, _P, C.c_uint64, _P, _P, C.POINTER(BatchStats)]),
    "ctmr_submit_batch": (C.c_int, [_P, _P, _P, _P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "ctmr_flush": (C.c_int, [_P]),
    "ctmr_wait": (C.c_int, [_P, C.c_uint64, _P, _P, C.POINTER(BatchStats)]),
    "ctmr_submit_entries": (C.c_int, [_P, _P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "ctmr_wait_entries": (C.c_int, [_P, C.c_uint64, _P, _P, _P, C.POINTER(DecodeStats), C.POINTER(BatchStats)]),
    "ctmr_set_insert": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "ctmr_set_contains": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "ctmr_set_remove": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "ctmr_set_cardinality": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_int64)]),
    "ctmr_exists": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "ctmr_set_members": (C.c_int, [_P, C.c_char_p, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_size_t),
                                   C.POINTER(C.c_uint64)]),
    "ctmr_keys": (C.c_int, [_P, C.c_char_p, C.c_size_t, _P, C.c_size_t, C.POINTER(C.c_size_t),
                            C.POINTER(C.c_uint64)]),
    "ctmr_expire_at": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.c_int64]),
    "ctmr_expire_sweep": (C.c_int, [_P, C.c_int64, C.POINTER(C.c_uint64)]),
    "ctmr_issuer_counts": (C.c_int, [_P, _P, C.c_uint32]),
    "ctmr_total_count": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "ctmr_issuer_counts_device": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint32)]),
    "ctmr_reset_known": (C.c_int, [_P]),
    "ctmr_table_info_get": (C.c_int, [_P, C.POINTER(TableInfo)]),
    "ctmr_xchg_map_device": (C.c_int, [_P, C.POINTER(Shard), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_uint64)]),
    "ctmr_xchg_map_chunk_device": (C.c_int, [_P, C.POINTER(Shard), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, _P,
                                             C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ctmr_xchg_keys_device": (C.c_int, [_P, _P, _P, C.POINTER(C.c_uint64)]),
    "ctmr_xchg_insert_device": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint64, _P, _P]),
    "ctmr_xchg_apply_device": (C.c_int, [_P, _P, _P, C.c_uint64, _P, _P, C.c_uint64, C.POINTER(BatchStats)]),
    "ctmr_bloom_config": (C.c_int, [_P, C.c_uint64, _P]),
    "ctmr_bloom_device": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "ctmr_bloom_add_device": (C.c_int, [_P, _P, _P, _P, C.c_uint64, _P]),
    "ctmr_bloom_probe_device": (C.c_int, [_P, _P, _P, _P, C.c_uint64, _P, _P, C.c_uint32, C.c_uint32, C.c_uint64,
                                          _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "ctmr_bloom_lookup_device": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, _P]),
    "ctmr_bloom_apply_device": (C.c_int, [_P, _P, C.c_uint64, _P, _P, C.c_uint64, _P, C.POINTER(BatchStats)]),
    "ctmr_group_create_local": (C.c_int, [C.POINTER(_P), C.c_uint32, C.POINTER(_P)]),
    "ctmr_group_unique_id": (C.c_int, [C.c_char_p]),
    "ctmr_group_create_rccl": (C.c_int, [_P, C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "ctmr_group_destroy": (None, [_P]),
    "ctmr_group_last_error": (C.c_char_p, [_P]),
    "ctmr_group_info": (C.c_int, [_P, C.POINTER(GroupStats)]),
    "ctmr_group_set_chunks": (C.c_int, [_P, C.c_uint32]),
    "ctmr_group_bloom_config": (C.c_int, [_P, C.c_uint64]),
    "ctmr_group_map_batch": (C.c_int, [_P, C.c_int, C.POINTER(Shard), C.POINTER(BatchStats)]),
    "ctmr_group_issuer_counts": (C.c_int, [_P, _P, C.c_uint32]),
    "ctmr_group_total_count": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "ctmr_group_all_reduce_u64": (C.c_int, [_P, _P, C.c_uint32, C.c_int]),
    "ctmr_group_barrier": (C.c_int, [_P]),
    "ctmr_pem_encode_device": (C.c_int, [_P, _P, _P, _P, C.c_uint64, _P, C.c_uint64, _P, C.POINTER(C.c_uint64)]),
    "ctmr_set_issuer_autoregister": (C.c_int, [_P, C.c_int]),
    "ctmr_set_chain0_match": (C.c_int, [_P, C.c_int]),
    "ctmr_set_strict_leaf": (C.c_int, [_P, C.c_int]),
    "ctmr_set_strict_extensions": (C.c_int, [_P, C.c_int]),
    "ctmr_set_strict_strings": (C.c_int, [_P, C.c_int]),
    "ctmr_set_strict_spki": (C.c_int, [_P, C.c_int]),
    "ctmr_set_profile": (C.c_int, [_P, C.c_int]),
    "ctmr_pending_issuers": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]),
    "ctmr_pem_encode_view_device": (C.c_int, [_P, _P, C.POINTER(EntryView), _P, C.c_uint64, _P, C.c_uint64, _P,
                                              C.POINTER(C.c_uint64)]),
    "ctmr_pem_new": (C.c_int, [_P, _P, C.c_size_t, _P, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]),
    "ctmr_decode_entries_device": (C.c_int, [_P, _P, _P, C.c_uint64, C.POINTER(EntryView), C.POINTER(DecodeStats)]),
    "ctmr_map_view_device": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(EntryView), C.c_uint64, _P, _P,
                                       C.POINTER(BatchStats)]),
    "ctmr_map_entries_device": (C.c_int, [_P, _P, _P, C.c_uint64, _P, _P, _P, C.POINTER(DecodeStats),
                                          C.POINTER(BatchStats)]),
    "ctmr_map_entries": (C.c_int, [_P, _P, _P, C.c_uint64, _P, _P, _P, C.POINTER(DecodeStats),
                                   C.POINTER(BatchStats)]),
    "ctmr_meta_new_device": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_uint64, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "ctmr_meta_new": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)]),
    "ctmr_meta_reset": (C.c_int, [_P]),
    "ctmr_fingerprint_device": (C.c_int, [_P, _P, _P, _P, C.c_uint64, _P, C.POINTER(C.c_float)]),
    # include/ctmr_bench.h — the synthetic corpus generator: exported by the same library, NOT part of the drop-in ABI
    "ctmr_synth_entries_host": (C.c_uint64, [C.POINTER(SynthConfig), C.c_uint64, C.c_uint64, _P, _P, C.c_uint64]),
    "ctmr_synth_entries_device": (C.c_int, [_P, C.POINTER(SynthConfig), C.c_uint64, C.c_uint64, _P, _P, C.c_uint64,
                                            C.POINTER(C.c_uint64)]),
    "ctmr_synth_leaf_len": (C.c_uint32, [C.POINTER(SynthConfig), C.c_uint64]),
    "ctmr_synth_leaf": (C.c_uint32, [C.POINTER(SynthConfig), C.c_uint64, _P, C.c_uint32,
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]),
    "ctmr_synth_issuer": (C.c_uint32, [C.POINTER(SynthConfig), C.c_uint32, _P, C.c_uint32]),
    "ctmr_synth_host": (C.c_uint64, [C.POINTER(SynthConfig), C.c_uint64, C.c_uint64, _P, _P, C.c_uint64,
                                     _P, _P]),
    "ctmr_synth_device": (C.c_int, [_P, C.POINTER(SynthConfig), C.c_uint64, C.c_uint64, _P, _P,
                                    C.c_uint64, _P, _P, C.POINTER(C.c_uint64)]),
    "ctmr_synth_view_device": (C.c_int, [_P, C.POINTER(SynthConfig), C.c_uint64, C.c_uint64, C.c_uint32, _P, _P, _P,
                                         C.c_uint64, _P, _P, C.POINTER(C.c_uint64)]),
}

_LIB = None


def lib():
    """Load libctmr.so (built in-tree by ct_mapreduce_amd.build).  Raises if it is missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -m ct_mapreduce_amd.build, or __graft_entry__.build()). "
                "There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError ⇒ ABI mismatch, loudly
            fn.restype = res
            fn.argtypes = args
        if L.ctmr_abi_version() != ABI_VERSION:
            raise ImportError("libctmr ABI version mismatch")
        _LIB = L
    return _LIB
