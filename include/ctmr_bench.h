/* ctmr_bench.h — the synthetic CT corpus generator of bench.py, the tests and the C++ host benchmarks.
 *
 * NOT part of the drop-in ABI.  include/ctmr.h is what a host of the reference binds; nothing in it
 * depends on these eight entry points and no integration calls them.  They are exported by the same libctmr.so only
 * because the device-side generators run on the engine's stream and write straight into HBM — a 100 M-entry batch is
 * 152 GB, which no host could generate and copy in the time a benchmark has.  The corpus has no counterpart in the
 * reference (its tests hold three certificates); it follows SURVEY.md §8(d) / BASELINE.json's `configs`.
 */
#ifndef CTMR_BENCH_H
#define CTMR_BENCH_H
#include "ctmr.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- the SURVEY.md §8(d) synthetic CT batch.  Deterministic in (seed, index); host and device emit identical bytes. */
typedef struct {
  uint64_t seed;
  uint32_t n_issuers;      /* 1 or 256 … */
  uint32_t zipf;           /* 1 = Zipf(s=1) issuer popularity, 0 = uniform */
  uint32_t dup_permille;   /* entries re-emitting an earlier (issuer, serial, notAfter) */
  uint32_t ca_permille;    /* basicConstraints CA:TRUE leaves (filter 1) */
  uint32_t expired_permille; /* notAfter < base time (filter 2 when now == base) */
  uint32_t mean_len;       /* 0 = 1536 */
  int64_t base_time;       /* 0 = 2026-01-01T00:00:00Z */
  uint32_t profile;        /* 0 = the SURVEY §8(d) corpus (RSA-2048 keys, 38-byte subjects, UTCTime);
                              1 = mixed: half the keys EC P-256, 40 % OV-like subjects of 120…260 bytes, longer issuer
                                  names for two issuers in three, one GeneralizedTime notAfter in four */
  uint32_t reserved;
} ctmr_synth_config;

/* Length of synthetic leaf i / issuer certificate k, and their bytes (host side). */
uint32_t ctmr_synth_leaf_len(const ctmr_synth_config* c, uint64_t i);
uint32_t ctmr_synth_leaf(const ctmr_synth_config* c, uint64_t i, uint8_t* out, uint32_t cap,
                         uint32_t* issuer_idx, uint8_t* entry_type);
uint32_t ctmr_synth_issuer(const ctmr_synth_config* c, uint32_t k, uint8_t* out, uint32_t cap);
/* Host batch [first, first+n): offsets u64[n+1] (relative), payload (capacity cap), issuer_idx,
 * entry_type.  Returns the payload bytes needed (nothing is written past cap). */
uint64_t ctmr_synth_host(const ctmr_synth_config* c, uint64_t first, uint64_t n, uint64_t* offsets,
                         uint8_t* payload, uint64_t cap, uint32_t* issuer_idx, uint8_t* entry_type);
/* Generate entries [first, first+n) directly in HBM: d_offsets u64[n+1] (relative to the
 * batch start), d_payload (capacity payload_cap), d_issuer_idx u32[n], d_entry_type u8[n].
 * *payload_bytes = bytes written.  With d_payload == NULL only offsets are produced. */
int ctmr_synth_device(ctmr_engine* e, const ctmr_synth_config* c, uint64_t first, uint64_t n,
                      uint64_t* d_offsets, uint8_t* d_payload, uint64_t payload_cap,
                      uint32_t* d_issuer_idx, uint8_t* d_entry_type, uint64_t* payload_bytes);

/* The same certificates as an ENTRY VIEW with every certificate starting at a multiple of `align` bytes (a power of two
 * up to 4096): d_starts u64[n+1] (d_starts[n] = bytes used), d_ends u64[n]; feed it to ctmr_map_view_device.  What the map
 * moves per certificate depends on where certificates start inside 128-byte lines: a host that writes
 * its decoded entries at aligned offsets gets the difference for nothing.  With d_payload == NULL only the positions. */
int ctmr_synth_view_device(ctmr_engine* e, const ctmr_synth_config* c, uint64_t first, uint64_t n, uint32_t align,
                           uint64_t* d_starts, uint64_t* d_ends, uint8_t* d_payload, uint64_t payload_cap,
                           uint32_t* d_issuer_idx, uint8_t* d_entry_type, uint64_t* payload_bytes);

/* Raw get-entries form of the same synthetic entries (input of ctmr_decode_entries_*): entry i is
 * leaf_input ‖ extra_data with the certificate of ctmr_synth_leaf(i) as X509Entry (entry_type 0; extra_data = chain
 * [issuer]) or as PrecertChainEntry.pre_certificate (entry_type 1; the leaf carries issuer_key_hash + the TBS; chain
 * [issuer]).  bounds u64[2n+1] relative to the batch start.  Same conventions as ctmr_synth_host / _device. */
uint64_t ctmr_synth_entries_host(const ctmr_synth_config* c, uint64_t first, uint64_t n, uint64_t* bounds,
                                 uint8_t* blob, uint64_t cap);
int ctmr_synth_entries_device(ctmr_engine* e, const ctmr_synth_config* c, uint64_t first, uint64_t n,
                              uint64_t* d_bounds, uint8_t* d_blob, uint64_t blob_cap, uint64_t* blob_bytes);

#ifdef __cplusplus
}
#endif
#endif /* CTMR_BENCH_H */
