// ctmr_dev.h — device-side data layout shared by the kernels and the engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ctmr.h"
#include "../../include/ctmr_bench.h"
#include "der_walk.h"

// Launch bounds of the kernels whose walk evaluates the curve equation on the spot (issuer registration, the strict_leaf
// TBS check — not the map kernels, which defer it: spki_key.h): one wave per workgroup, at most 168 VGPRs.  The bound is
// also what the compiler gives the out-of-line curve checks, whose attributes it derives from their callers.
#define CTMR_WALK_BOUNDS __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3)))

namespace ctmr {

// ---------------------------------------------------------------- known-certificate table (round 4: index + arena)
// One member of a Redis set "serials::<expDate>::<issuerID>" (knowncertificates.go:28-55) = one claimed word of the
// INDEX — open addressing, linear probing, 8 bytes per slot:
//     tag24 << 40 | ref40        0 = empty, all ones = removed member (tombstone)
// — and one 64-byte CELL of the ARENA, arena[ref], that holds the key.  A round's entries get consecutive cells
// (ref = the round's base + the entry's index; keys received from other ranks are appended behind), so the cells of a
// wave are written as ONE contiguous 4 KiB store instead of 64 random 64-byte slot images, a claim is ONE random 8-byte
// atomic, "known since an earlier round" is ref < the round's base, and a reset clears 8 bytes per slot instead of 64.
// (Rounds 1–3: 64-byte slots holding claim word, key and epoch together — the image store behind the CAS was a second
// random DRAM transaction per new key, ≈ 1.2 ms of the map kernel per 94 M keys, and the bench cleared 17 GB per step.)
//   meta  VALID(63) | SHADOW(62) | serial_len(61..56) | canonical issuer (55..32) | exp_hour (31..0)
//   s     serial octets, zero padded (CTMR_MAX_SERIAL = 40)
//   ord   order of the entry in its round (kernels/keyrec.h): among the presenters of one new key the lowest keeps
//         WasUnknown — pass 2 moves the index word to the lowest presenter's cell
// A cell is written BEFORE a word that points to it can be read for comparison: pass 1 never reads cells of its own
// round (a same-tag word of this round = DEFER), and whoever publishes a word later wrote its cell in pass 1.
struct __attribute__((aligned(64))) KeyCell {
  unsigned long long meta;
  unsigned long long s[5];
  uint32_t ord;
  uint32_t pad0;
  unsigned long long pad1;
};
static_assert(sizeof(KeyCell) == 64, "cell");
struct Table {
  unsigned long long* index;
  uint64_t mask;   // slots − 1
  KeyCell* arena;
};

constexpr unsigned long long SLOT_VALID = 1ull << 63;
constexpr unsigned long long CELL_SHADOW = 1ull << 62;  // Bloom-variant global dedup: known for dedup, counted by another rank
constexpr unsigned long long REF_MASK = (1ull << 40) - 1ull;
constexpr unsigned long long IDX_TOMB = ~0ull;           // removed member
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

__host__ __device__ inline uint32_t key_tag(unsigned long long h) {  // 24 bits, never 0 (empty) nor all ones (tombstone)
  uint32_t t = (uint32_t)(h >> 40);
  if (t == 0u) t = 1u;
  if (t == 0xffffffu) t = 0xfffffeu;
  return t;
}
__host__ __device__ inline unsigned long long idx_word(unsigned long long h, unsigned long long ref) {
  return ((unsigned long long)key_tag(h) << 40) | ref;
}
__host__ __device__ inline bool idx_same_tag(unsigned long long w, unsigned long long h) {
  return (w >> 40) == (unsigned long long)key_tag(h);
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
  unsigned long long n_pending; // strict_spki: non-zero = some entry's EC key owes the curve equation (k_ec_resolve exits at once on 0)
  unsigned long long n_pending_other;  // … on a curve other than P-256 (k_ec_resolve<false>: the kernel with the big register file)
};

// map-kernel filter constants (device memory; uniform reads)
struct FilterDev {
  uint32_t active;       // len(*ctconfig.IssuerCNFilter) != 0
  uint32_t n_pieces;     // strings.Split(filter, ",")
  uint32_t log_expired;
  uint32_t strict_spki;  // ctmr_set_strict_spki: the walk also parses the public key (spki_key.h); on by default
  uint32_t strict_strings;  // ctmr_set_strict_strings: character sets of the Names' string values, inside the walk
  uint32_t strict_ext;      // ctmr_set_strict_extensions: the bodies of the extensions Go unmarshals (der_walk.h ext_body_check)
                            // (in what was a padding word: every other field keeps its place)
  long long now;
  uint32_t piece_len[64];
  uint32_t piece_word[64];  // index of the piece's first word in words[]
  uint32_t words[1024];     // pieces, zero padded to 4-byte multiples
};

}  // namespace ctmr
