// ctmr_dev.h — device-side data layout shared by the kernels and the engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ctmr.h"
#include "der_walk.h"

// Launch bounds of the kernels whose walk evaluates the curve equation on the spot (issuer registration, the strict_leaf
// TBS check — not the map kernels, which defer it: spki_key.h): one wave per workgroup, at most 168 VGPRs.  The bound is
// also what the compiler gives the out-of-line curve checks, whose attributes it derives from their callers.
#define CTMR_WALK_BOUNDS __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3)))

namespace ctmr {

// ---------------------------------------------------------------- known-certificate table
// Open addressing, linear probing, one 64-byte slot per known certificate
// (= one member of a Redis set "serials::<expDate>::<issuerID>", knowncertificates.go:28-55):
//   w[0]  tag32 << 32 | min_idx32   claimed with atomicCAS(0 → …); same-key entries of one
//                                   batch atomicMin their batch index into the low half so the
//                                   lowest log index is the one that "was unknown"
//   w[1]  VALID(63) | serial_len(62..56) | canonical issuer (55..32) | exp_hour (31..0)
//                                   published last (write-through) — readers poll it
//   w[2]  epoch of the batch that created the slot
//   w[3..7] serial octets, zero padded (CTMR_MAX_SERIAL = 40)
struct __attribute__((aligned(64))) Slot {
  unsigned long long w[8];
};
static_assert(sizeof(Slot) == 64, "slot");

constexpr unsigned long long SLOT_VALID = 1ull << 63;
constexpr unsigned long long SLOT_TOMB = 0xffffffff00000000ull;  // removed member
constexpr uint32_t SID_NONE = 0xffffffffu;       // entry did not reach the set
constexpr uint32_t SID_HOST = 0xfffffffeu;       // serial longer than CTMR_MAX_SERIAL
constexpr uint32_t SID_FULL = 0xfffffffdu;       // table full
constexpr uint32_t SID_DUP_OLD = 0xfffffffcu;    // key known since an earlier batch

__host__ __device__ inline unsigned long long key_meta(int32_t exp_hour, uint32_t canon,
                                                        uint32_t serial_len) {
  return SLOT_VALID | ((unsigned long long)(serial_len & 0x7fu) << 56) |
         ((unsigned long long)(canon & 0xffffffu) << 32) | (uint32_t)exp_hour;
}

__host__ __device__ inline unsigned long long mixk(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
// 64-bit hash of (meta, serial words): three multiply-mix rounds (a 64-bit multiply is four quarter-rate 32-bit
// multiplies on CDNA: the six-round version was ≈ 8 % of the map kernel's VALU time).  Equal keys always compare in
// full, so the hash only has to spread; serial words are folded in rotated so that swapped words do not cancel.
__host__ __device__ inline unsigned long long rotl64(unsigned long long x, int r) { return (x << r) | (x >> (64 - r)); }
__host__ __device__ inline unsigned long long key_hash(unsigned long long meta,
                                                        const unsigned long long s[5]) {
  unsigned long long h = mixk(meta + 0x9e3779b97f4a7c15ull);
  h = mixk(h ^ s[0] ^ rotl64(s[1], 29) ^ 0x3c6ef372fe94f82bull);
  h = mixk(h ^ s[2] ^ rotl64(s[3], 29) ^ rotl64(s[4], 47));
  return h;
}
// Probe sequence of the known-certificate table: linear.  (Tried in round 2: the home slot, then the OTHER slot of the
// same 128-byte line — already on die after the first probe —, then the next line.  The map kernel does get faster with a
// sparser table — 23.2 ms at 2^28 slots, 21.9 at 2^29, 20.9 at 2^30 for 94 M keys — but not with this sequence: 23.14–23.18
// ms A/B/A/B against 23.14–23.17 linear, profiles/r02/sweep_probe_sequence.txt.  What a collision costs is the next
// dependent ATOMIC round trip, whether or not its line is already in L2.)  `k` = 0-based index of the probe that failed at j.
__host__ __device__ inline uint64_t probe_next(uint64_t j, uint64_t k, uint64_t mask) {
  (void)k;
  return (j + 1ull) & mask;
}

__host__ __device__ inline uint32_t key_tag(unsigned long long h) {
  uint32_t t = (uint32_t)(h >> 32);
  if (t == 0u) t = 1u;
  if (t == 0xffffffffu) t = 0xfffffffeu;
  return t;
}

// ---------------------------------------------------------------- (expDate, issuer) → cardinality
// 16-byte slots: key = (canon+1) << 32 | (uint32)exp_hour (0 = empty), count.
struct PairSlot {
  unsigned long long key, count;
};

// device-resident per-batch statistics
struct DevStats {
  unsigned long long by_status[CTMR_ST__COUNT];
  unsigned long long n_new, n_dup, n_host, n_full, pair_full;
  unsigned long long n_remote;  // owner-computes rounds: PASS entries whose key left for its owner (counted in n_new, optimistically)
  unsigned long long n_xl;      // … of them with a 21..40-octet serial (they travel as 64-byte records)
  unsigned long long n_pending; // strict_spki: entries whose EC key owes the curve equation (k_ec_resolve exits at once on 0)
};

// map-kernel filter constants (device memory; uniform reads)
struct FilterDev {
  uint32_t active;       // len(*ctconfig.IssuerCNFilter) != 0
  uint32_t n_pieces;     // strings.Split(filter, ",")
  uint32_t log_expired;
  uint32_t strict_spki;  // ctmr_set_strict_spki: the walk also parses the public key (spki_key.h); on by default
  uint32_t strict_strings;  // ctmr_set_strict_strings: character sets of the Names' string values, inside the walk
  uint32_t pad2;
  long long now;
  uint32_t piece_len[64];
  uint32_t piece_word[64];  // index of the piece's first word in words[]
  uint32_t words[1024];     // pieces, zero padded to 4-byte multiples
};

}  // namespace ctmr
