/*
 * ctmr.h — C ABI of libctmr (MI355X-native ct-mapreduce map/reduce hot path).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  A Go host
 * binds it through cgo (INTEGRATION.md shows the stub); the C++ host mirror in
 * ct_mapreduce_amd/csrc/storage.hpp and the Python ctypes binding in
 * ct_mapreduce_amd/_native.py bind exactly the same symbols.
 *
 * Every entry point cites the reference interface it replaces (paths relative to
 * jcjones/ct-mapreduce).  The reference has no FFI of its own (100 % Go): the seam is the
 * `entryChan` consumer (cmd/ct-fetch/ct-fetch.go:191) for the batched map, and the
 * storage.RemoteCache set methods (storage/types.go:83-102) for the reduce state.
 *
 * Conventions
 *   - every function returns CTMR_OK (0) or a negative CTMR_E_* code; ctmr_last_error()
 *     gives the message of the last failure on that engine.
 *   - strings/members are (ptr,len) byte ranges, never NUL-terminated (serials contain \0).
 *   - no pointer passed in is retained after the call returns (cgo rule).
 *   - all entry points are thread-safe (one mutex per engine; GPU work is stream-ordered).
 *   - there is NO CPU fallback: without a usable HIP device ctmr_create fails loudly.
 */
#ifndef CTMR_H
#define CTMR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTMR_ABI_VERSION 7

enum {
  CTMR_OK = 0,
  CTMR_E_INVAL = -1,     /* bad argument */
  CTMR_E_HIP = -2,       /* HIP runtime error (message has the hipError string) */
  CTMR_E_NOMEM = -3,     /* device or host allocation failed */
  CTMR_E_FULL = -4,      /* known-certificate table (at max_table_slots) / issuer table / pair table is full; a map /
                            insert call that returns it has not been applied (no slot claimed, no counter changed) */
  CTMR_E_NOTFOUND = -5,
  CTMR_E_RANGE = -6      /* caller buffer too small; *need tells the size */
};

/* Per-entry status, in the order insertCTWorker tests things
 * (cmd/ct-fetch/ct-fetch.go:191-235). */
enum {
  CTMR_ST_PASS = 0,               /* reached database.Store (:229) */
  CTMR_ST_PARSE_ERROR = 1,        /* x509.ParseCertificate(leaf) failed (:202-209) */
  CTMR_ST_FILTERED_CA = 2,        /* certIsFilteredOut :47-50 */
  CTMR_ST_FILTERED_EXPIRED = 3,   /* :52-55 */
  CTMR_ST_FILTERED_CN = 4,        /* :57-69 */
  CTMR_ST_NO_ISSUER = 5,          /* len(Chain) < 1 (:215-219) */
  CTMR_ST_ISSUER_PARSE_ERROR = 6, /* x509.ParseCertificate(Chain[0]) failed (:221-225) */
  CTMR_ST_ENTRY_DECODE_ERROR = 7, /* ct.LogEntryFromLeaf failed: the downloader drops the entry before entryChan
                                     (:452-459); only produced by the raw-entry calls (ctmr_decode_entries_*) */
  CTMR_ST__COUNT = 8
};

/* record.flags */
#define CTMR_FL_PRECERT 0x01      /* entry_type == 1 (ct.PrecertLogEntryType, :201) */
#define CTMR_FL_WAS_UNKNOWN 0x02  /* KnownCertificates.WasUnknown == true (knowncertificates.go:38) */
#define CTMR_FL_LONG_SERIAL 0x04  /* serial_len > 20: serial[] holds only the first 20 octets */

#define CTMR_NO_ISSUER 0xFFFFFFFFu /* issuer_idx value meaning "len(Chain) < 1" */
#define CTMR_ENTRY_INVALID 0xFFu   /* entry_type value meaning "LogEntryFromLeaf failed" → CTMR_ST_ENTRY_DECODE_ERROR */
#define CTMR_PAYLOAD_PAD 32        /* readable bytes required after offsets[n] (device inputs) */
#define CTMR_MAX_SERIAL 40         /* longest serial the in-HBM set stores; longer → host set */

/* One 32-byte output record per entry (the "+32 B" of B_alg). */
typedef struct {
  uint8_t status;       /* CTMR_ST_* */
  uint8_t flags;        /* CTMR_FL_* */
  uint16_t serial_len;  /* raw INTEGER content length (storage/types.go:171-178) */
  int32_t exp_hour;     /* floor(NotAfter/3600): NewExpDateFromTime (types.go:339-346) */
  uint32_t issuer_idx;  /* as passed in */
  uint8_t serial[20];   /* first min(20,serial_len) raw serial octets, zero padded */
} ctmr_record;

typedef struct {
  uint32_t struct_size;       /* sizeof(ctmr_config) */
  int32_t device;             /* HIP device ordinal */
  uint64_t table_slots;       /* known-certificate table: index slots (8-byte words; rounded up to 2^k); 0 = 2^24.
                                 The 64-byte key cells live in an arena of slots/2 cells that grows on demand */
  uint64_t pair_slots;        /* (expDate,issuer) cardinality table capacity; 0 = 2^22 */
  uint32_t max_issuers;       /* 0 = 65536 */
  uint32_t certs_per_tile;    /* sweep build only (variant 1); 0 = default */
  uint32_t lds_tile_bytes;    /* sweep build only (variant 1); 0 = default */
  uint32_t map_variant;       /* 0 = default (15).  15 = k_map_fused: the map fused with pass 1 of the known-certificate
                                 insert; 13 = k_map_winc + k_insert: the same map and insert as separate kernels.  Any other
                                 value: CTMR_E_INVAL (the baseline designs 1 = whole-certificate LDS tile and 2 = direct
                                 global loads exist only in the sweep build, scripts/sweep.py) */
  uint32_t profile;           /* 1 = bracket every kernel with hipEvents (ctmr_batch_stats.ms_*) */
  uint32_t collect_meta;      /* 1 = the map also records where each certificate's issuer Name and
                                 cRLDistributionPoints lie (8 B per entry) so that ctmr_meta_new* can run */
  uint64_t max_table_slots;   /* growth limit of the known-certificate table; 0 = 2^31.  Redis grows until OOM
                                 (storage/rediscache.go:57-65); here, before a call that may claim more slots than keep
                                 the load under 3/4, the table is rebuilt on the GPU into the power of two that holds
                                 (live members + incoming) at load <= 1/2 — tombstones (expiry sweeps, SetRemove) are
                                 dropped by the rebuild.  When even this limit cannot hold the call, it fails with
                                 CTMR_E_FULL BEFORE anything is inserted: a batch is applied completely or not at all. */
} ctmr_config;

typedef struct {
  uint64_t n;                          /* entries in the batch */
  uint64_t by_status[CTMR_ST__COUNT];  /* histogram of record.status */
  uint64_t n_new;                      /* PASS entries whose key was unknown (first by log index) */
  uint64_t n_dup;                      /* PASS entries already known */
  uint64_t n_host_set;                 /* PASS entries with serial_len > CTMR_MAX_SERIAL (host-side set) */
  uint64_t payload_bytes;              /* offsets[n]-offsets[0] */
  /* filled when config.profile: per-kernel GPU time of this batch, milliseconds */
  float ms_map, ms_insert, ms_resolve, ms_compact, ms_total;
  uint32_t map_launches;
} ctmr_batch_stats;

typedef struct {
  int32_t valid;          /* the issuer certificate parsed (else entries get ISSUER_PARSE_ERROR) */
  uint32_t canonical_idx; /* first registered issuer with the same SPKI digest */
  uint8_t spki_sha256[32];/* SHA-256(RawSubjectPublicKeyInfo), computed on the GPU */
  char issuer_id[48];     /* Issuer.ID(): padded base64url of the digest, NUL terminated (types.go:124-130) */
} ctmr_issuer_info;

typedef struct ctmr_engine ctmr_engine;

/* ---- lifecycle (no reference equivalent; engine.GetConfiguredStorage engine/engine.go:19-48
 *      is where a Go host would construct it) ---- */
int ctmr_abi_version(void);
int ctmr_create(const ctmr_config* cfg, ctmr_engine** out);
void ctmr_destroy(ctmr_engine* e);
/* The message of the last failure on that engine, copied for the calling thread: valid until this thread's next
 * ctmr_last_error call (other threads may fail concurrently; nobody is handed a pointer into a string in flux). */
const char* ctmr_last_error(const ctmr_engine* e);
/* Launch all GPU work on this hipStream_t (default: a stream the engine owns). */
int ctmr_set_stream(ctmr_engine* e, void* hip_stream);
int ctmr_synchronize(ctmr_engine* e);

/* Page-locked host memory for the host-buffer entry points (ctmr_map_batch, ctmr_map_entries, ctmr_add_issuers):
 * buffers obtained here are copied to HBM by DMA at full PCIe rate without the driver's bounce copy.  A cgo host
 * fills them through unsafe.Slice; they are C memory, so the "no Go pointer is retained" rule is not involved. */
int ctmr_alloc_pinned(ctmr_engine* e, size_t bytes, void** out);
int ctmr_free_pinned(ctmr_engine* e, void* p);

/* ---- issuer table: replaces x509.ParseCertificate(Chain[0]) + NewIssuer + Issuer.ID()
 *      (ct-fetch.go:221; storage/types.go:109-130,155-159).  Appends n issuer certificates
 *      (DER blob + n+1 offsets); *first_idx receives the index of the first one. ---- */
int ctmr_add_issuers(ctmr_engine* e, const uint8_t* der, const uint64_t* offsets, uint32_t n,
                     uint32_t* first_idx);
int ctmr_issuer_count(ctmr_engine* e, uint32_t* n);
/* SHA-256 of one byte string, on the GPU: SPKI.Sha256DigestURLEncodedBase64 for an Issuer constructed from raw
 * SPKI bytes instead of a chain certificate (storage/types.go:155-159).  Not a hot-path call. */
int ctmr_sha256(ctmr_engine* e, const uint8_t* data, size_t len, uint8_t out[32]);
int ctmr_issuer_info_get(ctmr_engine* e, uint32_t idx, ctmr_issuer_info* out);

/* ---- filter configuration: *ctconfig.IssuerCNFilter, *ctconfig.LogExpiredEntries and the
 *      time.Now() of certIsFilteredOut (ct-fetch.go:44-70; config/config.go:194-196) ---- */
int ctmr_set_filter(ctmr_engine* e, const char* issuer_cn_filter, size_t len, int log_expired,
                    int64_t now_unix);

/* ---- the batched map + reduce: replaces the body of insertCTWorker's loop
 *      (ct-fetch.go:191-235) through FilesystemDatabase.Store's WasUnknown
 *      (storage/filesystemdatabase.go:158-183; storage/knowncertificates.go:38-55).
 *   payload      packed leaf DER (the X509 cert or Precert.Submitted.Data, :198-204)
 *   offsets      n+1 byte offsets into payload
 *   issuer_idx   per entry index into the issuer table, or CTMR_NO_ISSUER
 *   entry_type   per entry 0 = X509LogEntryType, 1 = PrecertLogEntryType (may be NULL = all 0)
 *   records      n records out (may be NULL)
 *   new_idx      indices (ascending) of entries with CTMR_FL_WAS_UNKNOWN; capacity n (may be NULL)
 * Host variant: all pointers are host memory; the library stages through pinned buffers. */
int ctmr_map_batch(ctmr_engine* e, const uint8_t* payload, const uint64_t* offsets,
                   const uint32_t* issuer_idx, const uint8_t* entry_type, uint64_t n,
                   ctmr_record* records, uint64_t* new_idx, ctmr_batch_stats* stats);
/* Device variant: payload/offsets/issuer_idx/entry_type/records/new_idx are DEVICE pointers
 * (payload 16-byte aligned, CTMR_PAYLOAD_PAD readable bytes after offsets[n]); stats is host. */
int ctmr_map_batch_device(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets,
                          const uint32_t* d_issuer_idx, const uint8_t* d_entry_type, uint64_t n,
                          ctmr_record* d_records, uint64_t* d_new_idx, ctmr_batch_stats* stats);

/* ---- asynchronous host ingestion: the counterpart, in front of the GPU, of the channel between the reference's
 *      downloader and its insertCTWorker goroutines (cmd/ct-fetch/ct-fetch.go:132,140-145,191): the downloader hands
 *      over one get-entries response at a time, at most 1 001 entries (:417-424), and a synchronous ctmr_map_batch per
 *      response is bound by the host↔device round trip, not by the GPU.
 *   submit: same arguments as ctmr_map_batch; returns a ticket at once.  The DER goes to HBM with an asynchronous copy
 *           behind the previous submit's bytes (straight DMA from buffers of ctmr_alloc_pinned — keep them untouched
 *           until ctmr_wait; pageable memory is staged before the call returns); consecutive submits coalesce into a
 *           super-batch that is mapped as ONE batch when it reaches 65 536 entries / 96 MB, on ctmr_flush, or when one
 *           of its tickets is waited for — while the next one is already filling.
 *   wait:   blocks until the ticket's super-batch is done; records / new_idx (indices relative to THIS batch, ascending)
 *           / stats as ctmr_map_batch returns them (any may be NULL).  A ticket is collected once.
 *      Super-batches run in submission order and entries keep their order inside one: the lowest log index of a key
 *      wins WasUnknown across everything in flight, as in a sequence of synchronous calls.  Up to 4 super-batches may
 *      be unfinished or uncollected; a submit that would need a fifth blocks while one is still running and fails
 *      with CTMR_E_RANGE when all four only wait to be collected.  Thread-safe (several downloader threads may submit
 *      and wait).  ctmr_pem_new / ctmr_meta_new refer to synchronous calls only. ---- */
typedef uint64_t ctmr_ticket;
int ctmr_submit_batch(ctmr_engine* e, const uint8_t* payload, const uint64_t* offsets, const uint32_t* issuer_idx,
                      const uint8_t* entry_type, uint64_t n, ctmr_ticket* ticket);
int ctmr_flush(ctmr_engine* e);
int ctmr_wait(ctmr_engine* e, ctmr_ticket ticket, ctmr_record* records, uint64_t* new_idx, ctmr_batch_stats* stats);
/* (raw get-entries responses: ctmr_submit_entries / ctmr_wait_entries, with the raw-entry calls below) */

/* ---- storage.RemoteCache set methods on byte strings (storage/types.go:83-102;
 *      Redis impl storage/rediscache.go:57-120,153-169; mock storage/mockcache.go:38-166).
 *      Keys of the form "serials::<expDateID-with-hour>::<issuerID of a registered issuer>"
 *      live in the HBM table; every other key lives in a host-side string-set store. ---- */
int ctmr_set_insert(ctmr_engine* e, const char* key, size_t key_len, const uint8_t* member,
                    size_t member_len, int* was_new);                      /* SetInsert  */
int ctmr_set_contains(ctmr_engine* e, const char* key, size_t key_len, const uint8_t* member,
                      size_t member_len, int* present);                    /* SetContains */
int ctmr_set_remove(ctmr_engine* e, const char* key, size_t key_len, const uint8_t* member,
                    size_t member_len, int* removed);                      /* SetRemove  */
int ctmr_set_cardinality(ctmr_engine* e, const char* key, size_t key_len, int64_t* n); /* SetCardinality */
int ctmr_exists(ctmr_engine* e, const char* key, size_t key_len, int* exists);         /* Exists */
/* SetList / SetToChan: members serialised [u32 len][bytes]…, sorted bytewise.  *need = bytes
 * required; returns CTMR_E_RANGE (and copies nothing) when cap < *need. */
int ctmr_set_members(ctmr_engine* e, const char* key, size_t key_len, uint8_t* out, size_t cap,
                     size_t* need, uint64_t* count);
/* KeysToChan(pattern): glob as path.Match ('*', '?', '[..]', '\\'); same serialisation, sorted. */
int ctmr_keys(ctmr_engine* e, const char* pattern, size_t pattern_len, uint8_t* out, size_t cap,
              size_t* need, uint64_t* count);
/* ExpireAt: records the key's expiry; ctmr_expire_sweep(now) drops every key whose expiry
 * is <= now (what Redis does lazily).  map_batch records ExpireAt(key, expDate hour) for
 * every serials key it creates (knowncertificates.go:44-47,98-104). */
int ctmr_expire_at(ctmr_engine* e, const char* key, size_t key_len, int64_t unix_seconds);
int ctmr_expire_sweep(ctmr_engine* e, int64_t now_unix, uint64_t* members_removed);

/* ---- per-issuer unique counts: Σ_expDate SCARD(serials::expDate::issuer), the quantity
 *      storage-statistics reports (cmd/storage-statistics/storage-statistics.go:44-53).
 *      out[i] is the count for registered issuer i (equal for issuers sharing an SPKI). ---- */
int ctmr_issuer_counts(ctmr_engine* e, uint64_t* out, uint32_t n);
int ctmr_total_count(ctmr_engine* e, uint64_t* out);
/* Device pointer to the live u64[max_issuers] canonical-issuer counters (for an RCCL
 * all-reduce by the multi-GPU driver). */
int ctmr_issuer_counts_device(ctmr_engine* e, void** d_counts, uint32_t* n);
/* Drop every known certificate (table, pair counts, counters, host-side sets keep keys). */
int ctmr_reset_known(ctmr_engine* e);

/* How full the known-certificate table is (what `INFO memory` / `DBSIZE` tell the operator of the reference's Redis,
 * storage/rediscache.go:21-45).  The table is an index of 8-byte words (slots) over an arena of 64-byte key cells; a batch
 * takes one cell per entry and cells of entries that were not new are squeezed out again when the arena runs short
 * (arena_compactions), before it is grown (arena_growths); the index is rebuilt at load 3/4 (rebuilds). */
typedef struct {
  uint64_t slots;              /* index words */
  uint64_t occupied;           /* index words claimed since the last rebuild: live members + tombstones */
  uint64_t arena_cells;        /* capacity of the key-cell arena */
  uint64_t arena_used;         /* cells handed out (live + not yet squeezed out) */
  uint64_t rebuilds, arena_compactions, arena_growths;
  uint64_t reserved;
} ctmr_table_info;
int ctmr_table_info_get(ctmr_engine* e, ctmr_table_info* out);

/* One rank's input of a multi-GPU round (ctmr_group_map_batch, ctmr_xchg_map_device): device pointers on that rank's
 * GPU, as ctmr_map_batch_device takes them; d_ends != NULL: an entry view (d_offsets = cert_start, d_ends = cert_end,
 * blob_bytes set).  order_base = log index of the shard's entry 0 (Bloom mode: the lowest order keeps WasUnknown; owner
 * mode derives the orders itself from the shards' sizes).  The shards of one Bloom round must cover order ranges
 * [order_base, order_base + n) that ASCEND WITH THE RANK AND DO NOT OVERLAP — log-index shards do —: two ranks presenting
 * the same new key under the same order would both keep WasUnknown.  ctmr_group_map_batch checks this over the whole
 * group in the round's opening control row and fails on every rank with CTMR_E_INVAL otherwise (a host that leaves every
 * order_base at 0 is told so, before anything is inserted). */
typedef struct {
  const uint8_t* d_payload;
  const uint64_t* d_offsets;
  const uint64_t* d_ends;       /* NULL = packed batch */
  const uint32_t* d_issuer_idx;
  const uint8_t* d_entry_type;  /* may be NULL */
  uint64_t n;
  uint64_t blob_bytes;          /* entry view only */
  uint64_t order_base;
  ctmr_record* d_records;
  uint64_t* d_new_idx;          /* may be NULL */
} ctmr_shard;

/* ---- cross-GPU global dedup, owner-computes: replaces the shared Redis set service
 *      (storage/rediscache.go:57-65: one SADD answers "was new" for every ct-fetch process) between shards.
 *      owner(key) = a hash of the key → [0, world).  One round on one rank is four calls; a multi-process host with its
 *      own transport puts its two all-to-alls between them (ctmr_group_map_batch does exactly that, natively):
 *   map:    maps the shard.  A key THIS rank owns is inserted on the spot by the map kernel (the fused path of
 *           ctmr_map_batch_device); a key another rank owns leaves as a 32-byte record (serials of 21..40 octets: a
 *           64-byte record).  ord_base = Σ n of the lower ranks in this round: the round's global log order, < 2^32 —
 *           among entries that bring the same new key in one round the lowest order keeps WasUnknown, as in the reference
 *           loop over one log.  counts32[w] (host) = 32-byte records for owner w; *n_long = 64-byte records (all owners).
 *   keys:   writes the records, partitioned by owner, ascending order inside a partition, into the caller's send
 *           buffers (Σ counts32 × 32 bytes; n_long × 64 bytes, counts64[w] of them for owner w).
 *   insert: the owner inserts what it received (any order of senders) into the round of its own shard and writes one byte
 *           per record: 1 = was unknown.  Bumps the owner's per-issuer counters (a key is counted where it is stored).
 *           Must be called once per round on every rank, received records or not: it also settles the rank's own shard.
 *   apply:  the sender's records left the map with CTMR_FL_WAS_UNKNOWN set; the returned bytes (same order as the
 *           records were sent) take it away from the entries whose key was known; compacts d_new_idx, fills stats.
 *   Every rank must have registered the same issuers in the same order.  Serials longer than CTMR_MAX_SERIAL are not
 *   exchanged by these per-rank steps (each rank's host-side set decides them for its shard); ctmr_group_map_batch
 *   settles them between the ranks at the end of the round.  shard.d_records is required. ---- */
int ctmr_xchg_map_device(ctmr_engine* e, const ctmr_shard* shard, uint32_t world, uint32_t rank, uint32_t ord_base,
                         uint64_t* counts32, uint64_t* n_long);
/* ctmr_xchg_map_device for entries [lo, hi) of the shard (lo a multiple of 1024; chunks in ascending order from 0, the
 * last ends at sh->n): map + this chunk's exported records gathered into d_keys32_out at once (per-owner partitions of the
 * chunk; counts32[w] of them for owner w).  After the last chunk the round stands as after ctmr_xchg_map_device;
 * ctmr_xchg_keys_device then delivers only the 64-byte records (*n_long, set by the last chunk). */
int ctmr_xchg_map_chunk_device(ctmr_engine* e, const ctmr_shard* sh, uint32_t world, uint32_t rank, uint32_t ord_base,
                               uint64_t lo, uint64_t hi, void* d_keys32_out, uint64_t* counts32, uint64_t* n_long);
int ctmr_xchg_keys_device(ctmr_engine* e, void* d_keys32_out, void* d_keys64_out, uint64_t* counts64);
int ctmr_xchg_insert_device(ctmr_engine* e, const void* d_keys32, uint64_t n32, const void* d_keys64, uint64_t n64,
                            uint8_t* d_flags32, uint8_t* d_flags64);
int ctmr_xchg_apply_device(ctmr_engine* e, const void* d_sent32, const uint8_t* d_flags32, uint64_t n32,
                           const void* d_sent64, const uint8_t* d_flags64, uint64_t n64, ctmr_batch_stats* stats);

/* ---- cross-GPU global dedup, Bloom pre-filter variant (the "all-gather of per-GPU Bloom fingerprints" of
 *      BASELINE.json's north_star) — exact, same results as the owner-computes exchange above.
 *      Every rank keeps its own known-certificate table (plain ctmr_map_*_device calls) plus a cumulative Bloom
 *      filter of the keys it found locally new.  One round = one map call per rank, then:
 *   add:    makes sure the filter holds the batch's CTMR_FL_WAS_UNKNOWN keys and opens the round.  Since ABI v4 an engine
 *           with a filter adds to it INSIDE the map kernel, so for the batch the engine mapped last this costs nothing
 *           (ctmr_set_insert sets the bits of its member too: every key a rank holds is in its filter).  The host all-gathers the filters
 *           (ctmr_bloom_device gives the pointer; n_words × 8 bytes per rank, rank-major in the gathered buffer).
 *   probe:  tests the batch's locally-new keys against the OTHER ranks' filters and writes one 64-byte key record per
 *           (key, peer whose filter holds it) into d_keys_out, partitioned by peer, ascending log index inside a
 *           partition; counts[p] (host) = records for peer p.  A key that hits no peer filter is on no other rank and
 *           is not exchanged at all.  order_base = global order of entry 0 of this rank's batch (its log index):
 *           between ranks that meet the same key in the same round the LOWEST order keeps WasUnknown.  When more
 *           than keys_cap records are needed: CTMR_E_RANGE, counts[] filled, nothing written — call again.
 *   lookup: the peer looks the received records up in its table (exact, read-only) and writes one byte per record:
 *           1 = known here before the asker's entry (since an earlier round, or this round under a lower order).
 *           order_base as given to this rank's own probe.
 *   apply:  the asker clears CTMR_FL_WAS_UNKNOWN of every flagged entry (once per entry), takes it out of its
 *           per-issuer count, marks the key's slot as counted elsewhere (it stays known for dedup, but is left out of
 *           SetCardinality / SetList / the per-issuer counts, so that sums over ranks are the global values),
 *           compacts new_idx and fills stats.  d_records must be the records of this engine's LAST map call (the
 *           round continues it: only the entries that lose the flag are touched).
 *   d_ends NULL = packed batch (d_offsets has n+1 entries); otherwise entry-view ranges (ctmr_entry_view).
 *   d_records NULL = the engine's own records of the last map call.  Same issuers in the same order on every rank.
 *   Serials longer than CTMR_MAX_SERIAL: as above (settled between the ranks by ctmr_group_map_batch). ---- */
/* bits: power of two, 2^12..2^40, ≈16 per key this rank will ever hold.  d_words NULL: the library allocates the
 * filter; otherwise a caller-owned device buffer of bits/8 bytes (e.g. this rank's row of the all-gather buffer),
 * zeroed by the call, which must outlive the engine's use of it.  Same size on every rank. */
int ctmr_bloom_config(ctmr_engine* e, uint64_t bits, void* d_words);
int ctmr_bloom_device(ctmr_engine* e, void** d_words, uint64_t* n_words);
int ctmr_bloom_add_device(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets, const uint64_t* d_ends,
                          uint64_t n, const ctmr_record* d_records);
int ctmr_bloom_probe_device(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets,
                            const uint64_t* d_ends, uint64_t n, const ctmr_record* d_records, const void* d_filters,
                            uint32_t world, uint32_t rank, uint64_t order_base, void* d_keys_out, uint64_t keys_cap,
                            uint64_t* counts);
int ctmr_bloom_lookup_device(ctmr_engine* e, const void* d_keys, uint64_t n_keys, uint64_t order_base,
                             uint8_t* d_flags);
int ctmr_bloom_apply_device(ctmr_engine* e, ctmr_record* d_records, uint64_t n, const void* d_keys_sent,
                            const uint8_t* d_flags, uint64_t n_keys, uint64_t* d_new_idx, ctmr_batch_stats* stats);

/* ---- multi-GPU groups: the shared set service of a sharded deployment, natively behind this ABI.  Replaces what one
 *      Redis server gives N ct-fetch processes that split a log by -offset/-limit (cmd/ct-fetch/ct-fetch.go:288-305;
 *      storage/rediscache.go:21-65: SADD answers "was new" for all of them).  A group has `world` ranks, each an
 *      engine on one GPU that maps the entries of its log-index shard; one ctmr_group_map_batch call per round runs the
 *      shard maps AND the exchange that makes the dedup global; ctmr_group_issuer_counts is the all-reduce of the
 *      per-issuer counts.  Two transports, the same per-rank kernels in the same order:
 *        ctmr_group_create_local  every rank lives in this process (one engine per device — a single host process
 *                                 driving the node's GPUs — or several engines on one device): device-to-device copies
 *        ctmr_group_create_rccl   one process per GPU: RCCL over xGMI (ncclSend/ncclRecv all-to-all, ncclAllGather of
 *                                 the Bloom filters, ncclAllReduce of the counts); librccl is loaded on demand.
 *                                 id = the bytes ctmr_group_unique_id produced on one rank, handed to the others by the
 *                                 host over any channel it has.
 *      Engines must have registered the same issuers in the same order (key records carry canonical issuer indices);
 *      they stay usable on their own; destroy the group before its engines.
 *   modes   CTMR_DEDUP_LOCAL  shard-local dedup only (exact when no key spans two shards: BASELINE config 4)
 *           CTMR_DEDUP_OWNER  owner-computes key exchange (ctmr_xchg_* above): map (+ insert of the keys the rank owns)
 *                             → key records all-to-all → owner insert → flags back → apply.  Every key is stored once,
 *                             on its owner.
 *           CTMR_DEDUP_BLOOM  all-gather of per-GPU Bloom filters as an exact pre-filter (ctmr_bloom_* above; needs
 *                             ctmr_group_bloom_config): local insert → filter all-gather → probe → key records only to
 *                             the peers whose filter matched → exact lookup → flags back → apply
 *           Both exact modes also settle the members with serials longer than CTMR_MAX_SERIAL (host-side sets) between
 *           the ranks at the end of a round: the lists of such members added in the round are all-gathered with the
 *           closing status row, a member some rank held before is known to every rank, otherwise the lowest log index
 *           keeps it — stored once, like every other key.  A round without such members pays nothing for it.
 *   shards  one ctmr_shard (above) per LOCAL rank, rank order.  Ranks hold CONTIGUOUS log-index ranges in rank order
 *           (rank r's entries all precede rank r+1's: ct-fetch's -offset/-limit split).  d_records is required in the
 *           OWNER and BLOOM modes.  stats: one per local rank (may be NULL).
 *   errors  an exact round opens with a control row (status, mode, filter size, order range of every rank): a call a rank
 *           must refuse — unknown mode, another mode than the group's, no filter configured, a bad shard, ranks that
 *           disagree about any of these — fails THERE, on every rank together and before anything is inserted.  A rank
 *           that fails later keeps taking part in the round's collectives (its peers would block for ever otherwise)
 *           and every rank's call returns the error at the end; the sets of a round that failed after its opening row
 *           are unspecified — destroy the group. ---- */
typedef struct ctmr_group ctmr_group;
#define CTMR_GROUP_ID_BYTES 128
enum { CTMR_DEDUP_LOCAL = 0, CTMR_DEDUP_OWNER = 1, CTMR_DEDUP_BLOOM = 2 };
enum { CTMR_TRANSPORT_LOCAL = 0, CTMR_TRANSPORT_RCCL = 1 };
typedef struct {
  uint32_t world, n_local, transport, first_local_rank;
  /* what the last ctmr_group_map_batch moved between ranks, summed over the local ranks */
  uint64_t keys_sent, keys_received;   /* key records to / from OTHER ranks */
  uint64_t filter_bytes_received;      /* Bloom mode: the peers' filters, per local rank */
  uint64_t wire_bytes_sent;            /* every byte handed to the transport for another rank: key records (32 B; 64 B for
                                          Bloom candidates and 21..40-octet serials), flag bytes, filters */
  /* wall time of the last round's phases on this process, ms (each phase ends with the ranks' streams drained):
   * [0] map (+ the insert of locally owned keys, + filter add)  [1] key export / filter all-gather + probe
   * [2] key records all-to-all  [3] owner insert / exact lookup (+ this shard's resolve)  [4] flags all-to-all
   * [5] apply + NEW-list compaction  [6] the control collectives (counts, status)
   * [7] NOT a time: how often the round drained a stream on the host (ABI v7) — every control row, every all-to-all and
   *     every phase boundary is one; with RCCL each is a latency no kernel hides */
  float ms_phase[8];
} ctmr_group_stats;
int ctmr_group_create_local(ctmr_engine* const* engines, uint32_t n, ctmr_group** out);
int ctmr_group_unique_id(uint8_t id[CTMR_GROUP_ID_BYTES]);
int ctmr_group_create_rccl(ctmr_engine* engine, const uint8_t id[CTMR_GROUP_ID_BYTES], uint32_t rank, uint32_t world,
                           ctmr_group** out);
void ctmr_group_destroy(ctmr_group* g);
const char* ctmr_group_last_error(const ctmr_group* g);
int ctmr_group_info(ctmr_group* g, ctmr_group_stats* out);
/* Owner-computes rounds map every shard in `chunks` pieces (1..64; default 1 = as one): the key records of chunk c travel on
 * a second stream per rank while chunk c + 1 is being walked, and the owner-side insert, the bytes back and the apply run
 * once at the end over everything.  Results are those of the unchunked round, entry for entry (inside one round the order
 * in which keys reach their owner does not matter: storage/rediscache.go:57-65 answers per key, not per batch).  Costs one
 * small control collective per chunk and exchange buffers sized for the worst case.  Every rank of the group sets the same
 * value (checked in the round's opening control row); Bloom and LOCAL rounds ignore it.  With chunks > 1 on an RCCL group
 * the call is COLLECTIVE (ABI v7): it makes the transfer streams and the second communicator the chunk transfers run on —
 * every rank calls it, with the same count; a rank-local failure fails every rank's call together. */
int ctmr_group_set_chunks(ctmr_group* g, uint32_t chunks);
/* bits per rank: power of two, ≈16 per key a rank will ever hold; same on every rank.  (A group of ONE rank has no peer
 * to ask: it keeps no filter and its Bloom rounds are the plain reduce.)  * One mode per group: the first round's mode is the only one the group accepts afterwards (CTMR_E_INVAL otherwise) — the
 * modes keep a key in different places and cannot see each other's.
 */
int ctmr_group_bloom_config(ctmr_group* g, uint64_t bits);
int ctmr_group_map_batch(ctmr_group* g, int mode, const ctmr_shard* shards, ctmr_batch_stats* stats);
/* Σ over ALL ranks of ctmr_issuer_counts / ctmr_total_count (storage-statistics.go:44-53 over the whole deployment):
 * the ncclAllReduce of the per-issuer count vector in the RCCL transport. */
int ctmr_group_issuer_counts(ctmr_group* g, uint64_t* out, uint32_t n);
int ctmr_group_total_count(ctmr_group* g, uint64_t* out);
/* Sum (op_max = 0) or maximum (1) over the ranks of n host-side u64 values, in place, and a barrier: what a
 * multi-process host needs besides the data path (the round's global NEW count, the slowest rank's time) without a
 * second communication library.  No-ops for a local group (the caller holds every rank's values). */
int ctmr_group_all_reduce_u64(ctmr_group* g, uint64_t* values, uint32_t n, int op_max);
int ctmr_group_barrier(ctmr_group* g);

/* ---- PEM write-back: replaces pem.EncodeToMemory(&pem.Block{Type: "CERTIFICATE",
 *      Bytes: aCert.Raw}) of FilesystemDatabase.Store (storage/filesystemdatabase.go:167-175,196-200); the host
 *      hands each PEM to StorageBackend.StoreCertificatePEM (storage/localdiskbackend.go:194-199).
 *   device variant: d_idx = n_idx entry indices (e.g. the new_idx list of ctmr_map_batch_device); PEM r is written
 *           at d_pem + d_pem_offsets[r]; d_pem_offsets has n_idx+1 slots; *pem_bytes = total.  With d_pem == NULL
 *           only the offsets and the total are produced (size query).
 *   ctmr_pem_new: PEM of every CTMR_FL_WAS_UNKNOWN entry of the LAST ctmr_map_batch (host variant, called with
 *           new_idx != NULL), ascending entry order; *need = bytes required (CTMR_E_RANGE when cap is smaller),
 *           *count = number of PEMs, pem_offsets = count+1 offsets into out (may be NULL). ---- */
int ctmr_pem_encode_device(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets,
                           const uint64_t* d_idx, uint64_t n_idx, uint8_t* d_pem, uint64_t pem_cap,
                           uint64_t* d_pem_offsets, uint64_t* pem_bytes);
int ctmr_pem_new(ctmr_engine* e, uint8_t* out, size_t cap, uint64_t* pem_offsets, size_t* need, uint64_t* count);
/* The device variant over an entry view (raw get-entries batches): certificate i is [cert_start[i], cert_end[i]) of
 * d_blob.  Declared behind ctmr_entry_view below. */

/* ---- CT get-entries leaf decode: replaces ct.LogEntryFromLeaf
 *      (cmd/ct-fetch/ct-fetch.go:452; RFC 6962 §3.4 MerkleTreeLeaf, §4.6 extra_data) and insertCTWorker's choice of
 *      certificate and issuer (:198-204 X509Cert | Precert.Submitted.Data; :215 len(Chain) < 1; :221 Chain[0]).
 *   raw entries: ONE blob holding, back to back, leaf_input_0 ‖ extra_data_0 ‖ leaf_input_1 ‖ extra_data_1 ‖ … (the
 *           base64-decoded "leaf_input"/"extra_data" members of get-entries); bounds u64[2n+1]: leaf_input_i =
 *           [bounds[2i], bounds[2i+1]), extra_data_i = [bounds[2i+1], bounds[2i+2]).  The blob is 16-byte aligned
 *           with CTMR_PAYLOAD_PAD readable bytes behind bounds[2n] (device inputs).
 *   decode: per entry the byte range of the certificate the map parses (inside the blob — nothing is copied), the
 *           entry type, the timestamp and the index of Chain[0] in the issuer table.  Chain[0] is matched bytewise
 *           against the certificates registered so far; one that is not registered yet is registered by the call
 *           (exactly what ctmr_add_issuers does), so a host that feeds raw entries never manages the issuer table.
 *           An entry ct.LogEntryFromLeaf would reject gets entry_type CTMR_ENTRY_INVALID.
 *   map_view: the batched map + reduce of ctmr_map_batch_device over such a view (certificates addressed by
 *           [cert_start, cert_end) instead of n+1 packed offsets).
 *   map_entries: decode + map_view in one call; host variant stages the blob through the engine's buffers. ---- */
typedef struct {
  uint64_t* cert_start;   /* n: first byte of the certificate inside the blob */
  uint64_t* cert_end;     /* n: one past its last byte */
  uint32_t* issuer_idx;   /* n: issuer table index of Chain[0]; CTMR_NO_ISSUER when len(Chain) < 1 */
  uint8_t* entry_type;    /* n: 0 X509LogEntryType, 1 PrecertLogEntryType, CTMR_ENTRY_INVALID */
  uint64_t* timestamp;    /* n or NULL: TimestampedEntry.Timestamp, ms since the epoch (ct-fetch.go:476) */
  uint64_t* chain0_start; /* n or NULL: Chain[0].Data inside the blob */
  uint32_t* chain0_len;   /* n or NULL: 0 when len(Chain) < 1 */
} ctmr_entry_view;

typedef struct {
  uint64_t n;
  uint64_t n_x509, n_precert, n_decode_error;
  uint64_t n_no_chain;        /* decoded entries with len(Chain) < 1 */
  uint64_t n_issuers_added;   /* distinct Chain[0] certificates this call registered */
  uint64_t blob_bytes;        /* bounds[2n] - bounds[0] */
  float ms_decode, ms_match;  /* filled when config.profile; decode and the first match round run as one kernel: ms_match = both;
                                 ms_decode = what runs in front of it: strict_leaf's TBSCertificate walk (k_leaf_tbs_check), else ≈ 0 */
} ctmr_decode_stats;

int ctmr_decode_entries_device(ctmr_engine* e, const uint8_t* d_blob, const uint64_t* d_bounds, uint64_t n,
                               const ctmr_entry_view* d_view, ctmr_decode_stats* stats);
int ctmr_map_view_device(ctmr_engine* e, const uint8_t* d_blob, uint64_t blob_bytes, const ctmr_entry_view* d_view,
                         uint64_t n, ctmr_record* d_records, uint64_t* d_new_idx, ctmr_batch_stats* stats);
int ctmr_map_entries_device(ctmr_engine* e, const uint8_t* d_blob, const uint64_t* d_bounds, uint64_t n,
                            ctmr_record* d_records, uint64_t* d_new_idx, uint64_t* d_timestamp,
                            ctmr_decode_stats* dstats, ctmr_batch_stats* stats);
int ctmr_map_entries(ctmr_engine* e, const uint8_t* blob, const uint64_t* bounds, uint64_t n, ctmr_record* records,
                     uint64_t* new_idx, uint64_t* timestamp, ctmr_decode_stats* dstats, ctmr_batch_stats* stats);
/* Asynchronous ingestion (ctmr_submit_batch above) for RAW get-entries responses — what the reference's downloader actually holds (ct-fetch.go:446-462):
 * blob = leaf_input_0 ‖ extra_data_0 ‖ leaf_input_1 ‖ …, bounds u64[2n+1] (ctmr_map_entries' input).  Submits of
 * either form share the pipeline and its ordering; a super-batch holds one form (a change of form closes the open one).
 * The super-batch is decoded, matched (Chain[0] certificates met for the first time are registered on the way, unless
 * ctmr_set_issuer_autoregister(e, 0): then it fails with CTMR_E_NOTFOUND like ctmr_map_entries) and mapped as ONE batch.
 * ctmr_wait_entries additionally hands out the entries' timestamps (ms, :476) and this batch's decode statistics
 * (n_issuers_added, ms_*: of the whole super-batch); it accepts tickets of ctmr_submit_batch too (timestamps 0,
 * dstats zeroed), and ctmr_wait accepts raw tickets. */
int ctmr_submit_entries(ctmr_engine* e, const uint8_t* blob, const uint64_t* bounds, uint64_t n, ctmr_ticket* ticket);
int ctmr_wait_entries(ctmr_engine* e, ctmr_ticket ticket, ctmr_record* records, uint64_t* new_idx, uint64_t* timestamp,
                      ctmr_decode_stats* dstats, ctmr_batch_stats* stats);
/* Issuer registration policy of the raw-entry calls.  on = 1 (default): a Chain[0] certificate met for the first time
 * is registered by the call.  on = 0: nothing is registered; when a batch contains unregistered Chain[0]
 * certificates the decode fails with CTMR_E_NOTFOUND and ctmr_pending_issuers lists them (distinct, [u32 len][DER]…,
 * ascending log index of first appearance) — for hosts that must keep the issuer tables of several engines
 * identical (multi-GPU key exchange: DESIGN.md §8): gather the pending lists of all ranks, register the union in
 * one agreed order with ctmr_add_issuers on every rank, call again. */
/* strict_extensions (ON by default since ABI v7: part of CTMR_PROFILE_REFERENCE, the engine's default).  Go 1.13's crypto/x509 parseCertificate parses
 * the VALUE of the extensions it knows and fails the certificate when that fails; with on != 0 the walk does the same:
 *   by plain encoding/asn1 struct rules, each filling its OCTET STRING ("trailing data") — keyUsage (one BIT STRING),
 *     subjectKeyIdentifier (one OCTET STRING), extKeyUsage (SEQUENCE OF OID), authorityKeyIdentifier (SEQUENCE with an
 *     optional [0]), certificatePolicies (SEQUENCE OF SEQUENCE { OID, … }), authorityInfoAccess (SEQUENCE OF SEQUENCE
 *     { OID, any element }), cRLDistributionPoints (SEQUENCE OF distributionPoint: three optional fields in order, a
 *     nameRelativeToCRLIssuer parsed like a Name) [the last one: ABI v6];
 *   subjectAltName [v6] — one universal SEQUENCE of elements that fit, dispatched on the tag NUMBER: a URI (6) goes
 *     through net/url's Parse and, when it has a host, x509's domainToReverseLabels;
 *   nameConstraints [v6] — as golang.org/x/crypto/cryptobyte reads it: SEQUENCE { [0] permitted, [1] excluded }, not both
 *     absent or empty; dNSName / rfc822Name / URI constraints IA5 and well-formed (parseRFC2821Mailbox, not an IP literal,
 *     no empty label), iPAddress 8 or 32 octets with a contiguous mask.
 * A violation is a FATAL parse error in every role (leaf, precertificate, Chain[0], the strict_leaf TBSCertificate).
 * NON-FATAL findings, as CT-go files them (an X509 entry keeps its certificate; a precertificate and a Chain[0] issuer are
 * dropped) [v6]: a subjectAltName iPAddress that is not 4 or 16 octets long; an embedded SCT list
 * (1.3.6.1.4.1.11129.2.4.2) that does not decode; an INTEGER inside a nameRelativeToCRLIssuer that is not minimally
 * encoded (with strict_strings: also its string values' character sets).
 * These are the standard library's rules plus what is recalled of certificate-transparency-go v1.1.0's changes to them;
 * neither can be verified without CT-go's source — hence a switch: on = 0 is for a host that has parsed the
 * certificates already.  Set it before the issuers are
 * registered (a Chain[0] is judged when it is registered). */
int ctmr_set_strict_extensions(ctmr_engine* e, int on);
int ctmr_set_issuer_autoregister(ctmr_engine* e, int on);
int ctmr_pending_issuers(ctmr_engine* e, uint8_t* out, size_t cap, size_t* need, uint64_t* count);
/* How Chain[0] is identified (cmd/ct-fetch/ct-fetch.go:221 parses it in full for every entry).
 *   CTMR_CHAIN0_EXACT (default): every byte of every entry's Chain[0] is compared with the registered certificate —
 *     equal means identical, whatever the log serves; the identification costs 871 B of HBM reads per entry on the
 *     synthetic corpus and bounds the raw path.
 *   CTMR_CHAIN0_TRUSTED_LOG: a certificate is parsed in full when it is REGISTERED (its first sighting, or
 *     ctmr_add_issuers); afterwards an entry is attributed to it when length, first 16 and last 16 bytes (the end of
 *     the signature) agree — no byte of Chain[0] beyond the lines the framing decode touches anyway is read.  Identical results on
 *     everything a log that serves the chains it validated can produce (two certificates with the same last 16
 *     signature bytes do not occur); a Chain[0] that agrees with a registered certificate in those 36 bytes but not
 *     in between — a damaged copy — is attributed to that certificate, where the reference would parse the bytes
 *     it was given (a parse error, or another issuer).  Which certificate a (length, head, tail) triple stands for
 *     is decided by the lowest log index that carries it in the call that registers it — not by scheduling.  extra_data is not covered
 *     by the log's Merkle tree and the reference verifies neither STH nor inclusion, so its own trust in the log is no
 *     narrower; the choice is the host's.  Registered certificates that agree in those 36 bytes with another
 *     registered certificate are always compared bytewise. */
#define CTMR_CHAIN0_EXACT 0
#define CTMR_CHAIN0_TRUSTED_LOG 1
int ctmr_set_chain0_match(ctmr_engine* e, int mode);
/* Precertificate entries: ct.LogEntryFromLeaf (cmd/ct-fetch/ct-fetch.go:452) also parses the TBSCertificate the
 * MerkleTreeLeaf carries (CT-go x509.ParseTBSCertificate) and the downloader drops the entry when that fails fatally
 * (:453-459).  on = 0 (CTMR_PROFILE_FAST): the leaf TBSCertificate is length-checked only — identical results on every
 * entry a log that validated its submissions can serve, and one pass less over ≈ 400 bytes of every precertificate entry.
 * on = 1 (the default since ABI v7): the raw-entry calls walk it (the certificate walk without the outer wrapper and the signature) before anything
 * else looks at the entry; an entry whose leaf TBSCertificate does not parse gets CTMR_ENTRY_INVALID /
 * CTMR_ST_ENTRY_DECODE_ERROR and its Chain[0] is never registered, as in the reference. */
int ctmr_set_strict_leaf(ctmr_engine* e, int on);
/* The public key inside subjectPublicKeyInfo.  x509.ParseCertificate (cmd/ct-fetch/ct-fetch.go:202, :221, :452) ends in
 * CT-go's parsePublicKey: an RSA key that is not SEQUENCE { INTEGER n, INTEGER e > 0 } with nothing behind it, a DSA key or
 * parameter set that is not positive INTEGERs, an EC key whose parameters do not name P-224/256/384/521 (or secp192r1)
 * or whose point is not an uncompressed point ON that curve is a FATAL parse error — the entry never reaches Store, in
 * any role; RSA parameters other than NULL, an INTEGER of the key that is not minimally encoded, a modulus <= 0 and
 * secp192r1 are non-fatal findings (an X509 entry keeps its certificate, a precertificate and a Chain[0] issuer are
 * dropped, like every other finding).  Other algorithms' keys are not looked at.  ON by default since ABI v5 (rounds
 * 1-3 skipped the key bits by length and accepted such certificates); on = 0 restores that, for a host that has
 * already parsed the keys.  Applies to the map, to issuers registered AFTER the call and, with strict_leaf, to the leaf
 * TBSCertificate.  Rules and their provenance (recalled from CT-go v1.1.0, cross-checked against OpenSSL): spki_key.h,
 * DESIGN.md §3.1. */
int ctmr_set_strict_spki(ctmr_engine* e, int on);
/* Character sets of the string values in the issuer and subject Names.  Go's encoding/asn1 rejects a PrintableString
 * with an octet outside A-Z a-z 0-9 space ' ( ) + , - . / : = ? (and '*', '&', which it tolerates), a NumericString
 * with anything but digits and space, an IA5String with an octet >= 0x80 and a UTF8String that is not valid UTF-8.
 * certificate-transparency-go's fork of that package is more lenient towards some of these — which, cannot be verified
 * without its source — so the rules are a SWITCH (on by default since ABI v7, as part of the reference profile; 0: not
 * checked, as in ABI v1-v6's defaults) and a violation is filed as a NON-FATAL finding: an X509 entry keeps its certificate, a precertificate and a Chain[0] issuer are
 * dropped (cmd/ct-fetch/ct-fetch.go:202-209, 221-225, 452-459).  on = 1: the map checks the two Names of every
 * precertificate while its walk holds them (since round 4; a pre-pass over every certificate before); issuers are
 * judged when they are registered (set the switch before registering them). */
int ctmr_set_strict_strings(ctmr_engine* e, int on);
/* The accept/reject profile as ONE choice (ABI v6) — what x509.ParseCertificate / ct.LogEntryFromLeaf decide at
 * cmd/ct-fetch/ct-fetch.go:202-209, :221-225, :452-459:
 *   CTMR_PROFILE_REFERENCE  THE DEFAULT of ctmr_create (since ABI v7): all four switches on — what the reference does, as
 *                           far as it can be known without CT-go's source (DESIGN.md §3.1 says which rules are recalled
 *                           from where).  The map reads the subjectAltName and checks the Names' character sets; it is
 *                           what bench.py's headline measures.
 *   CTMR_PROFILE_FAST       the opt-in of a host that has ALREADY parsed its certificates (or trusts its log to have):
 *                           strict_spki on; strict_leaf, strict_strings, strict_extensions off — every rule whose bytes
 *                           the path reads anyway.  Looser than the reference on malformed extension bodies, Name
 *                           character sets and the leaf TBSCertificate of precertificate entries; identical on everything
 *                           a CA's encoder and a log that validated its submissions produce (bench.py reports it as
 *                           secondary.fast_profile).
 * Equivalent to the four ctmr_set_strict_* calls; like them, set it BEFORE the issuers are registered. */
#define CTMR_PROFILE_FAST 0
#define CTMR_PROFILE_REFERENCE 1
int ctmr_set_profile(ctmr_engine* e, int profile);
/* ctmr_pem_encode_device for an entry view: PEM of the certificates d_idx[0..n_idx) names, straight out of the blob. */
int ctmr_pem_encode_view_device(ctmr_engine* e, const uint8_t* d_blob, const ctmr_entry_view* d_view,
                                const uint64_t* d_idx, uint64_t n_idx, uint8_t* d_pem, uint64_t pem_cap,
                                uint64_t* d_pem_offsets, uint64_t* pem_bytes);

/* ---- IssuerMetadata on device: replaces the per-new-certificate part of
 *      IssuerMetadata.Accumulate (storage/issuermetadata.go:92-138) — its three per-issuer memo maps knownExpDates,
 *      knownCrlDPs, knownIssuerDNs live in HBM — so that the host only handles FIRST sightings:
 *        CTMR_MK_EXPDATE  (issuer, expDate hour) not seen before → seenExpDateBefore == false →
 *                         backend.AllocateExpDateAndIssuer (storage/filesystemdatabase.go:189-195)
 *        CTMR_MK_CRL      a CRLDistributionPoints URI not seen for this issuer → addCRL (issuermetadata.go:48-73);
 *                         off/len = the URI bytes inside the certificate
 *        CTMR_MK_DN       an issuer Name not seen for this issuer → addIssuerDN(aCert.Issuer.String()) (:75-87);
 *                         off/len = the Name TLV inside the certificate (the host formats pkix.Name.String())
 *        CTMR_MK_HOST     this certificate must be parsed by the host (an element does not fit the device fast
 *                         path: > 64 KiB offsets, > 4 KiB strings, a repeated extension)
 *      Needs config.collect_meta; call right after the map call of the same batch.  Items are in no particular
 *      order.  Device variant: d_offsets/d_ends/d_records/d_new_idx as given to / produced by that map call
 *      (d_ends NULL for a packed batch); when more than items_cap items exist the call fails with CTMR_E_RANGE,
 *      *n_items = the number needed, and the memo is cleared (a retry re-reports earlier sightings, which the
 *      host's sets tolerate — "Must tolerate duplicate information", issuermetadata.go:89).
 *      Host variant (after ctmr_map_batch / ctmr_map_entries called with new_idx != NULL): items plus their bytes
 *      (item k's bytes follow item k-1's; CTMR_MK_HOST and CTMR_MK_EXPDATE carry none); buffers too small →
 *      CTMR_E_RANGE with *n_items / *bytes_need set, nothing lost — call again. ---- */
enum { CTMR_MK_EXPDATE = 0, CTMR_MK_CRL = 1, CTMR_MK_DN = 2, CTMR_MK_HOST = 3 };
typedef struct {
  uint64_t entry;       /* index into the batch */
  uint32_t kind;        /* CTMR_MK_* */
  uint32_t issuer_idx;  /* as in the record */
  int32_t exp_hour;
  uint32_t off, len;    /* byte range inside the certificate */
  uint32_t pad;
} ctmr_meta_item;
int ctmr_meta_new_device(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets, const uint64_t* d_ends,
                         const ctmr_record* d_records, const uint64_t* d_new_idx, uint64_t n_new,
                         ctmr_meta_item* d_items, uint64_t items_cap, uint64_t* n_items);
int ctmr_meta_new(ctmr_engine* e, ctmr_meta_item* items, uint64_t items_cap, uint8_t* bytes, size_t bytes_cap,
                  uint64_t* n_items, size_t* bytes_need);
int ctmr_meta_reset(ctmr_engine* e);

/* ---- whole-certificate SHA-256 fingerprints (auxiliary; NOT on the reference's path, which never hashes a leaf —
 *      SURVEY.md D2; this is the one-certificate-per-lane SHA-256 kernel BASELINE.json's north_star names).
 *      d_digests receives n × 32 bytes (the digest as crypto/sha256.Sum256 prints it); certificate i is
 *      [d_offsets[i], d_offsets[i+1]) of d_payload, or [d_offsets[i], d_ends[i]) when d_ends != NULL (entry view).
 *      VALU-bound, not HBM-bound: priced against its own roofline in DESIGN.md §5. ---- */
int ctmr_fingerprint_device(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets,
                            const uint64_t* d_ends, uint64_t n, uint8_t* d_digests, float* ms);

/* The synthetic CT corpus generator bench.py and the tests use (ctmr_synth_*) is declared in ctmr_bench.h: the same
 * library exports it, but it is NOT part of the drop-in ABI — a host of the reference never binds it. */

#ifdef __cplusplus
}
#endif
#endif /* CTMR_H */
