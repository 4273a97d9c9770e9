// kernels/keyrec.h — what crosses the wire in the global-dedup modes: key records, the owner of a key, the Bloom word of a key.
// gfx950 (CDNA4, wave64) only; part of kernels.h, which includes the pieces in dependency order.
#pragma once
#include "../ctmr_dev.h"

namespace ctmr {

constexpr uint32_t MAX_WORLD = 16;
constexpr uint32_t KEY_NO_OWNER = 0xffu;

// A key = one member of one serials::<expDate>::<issuerID> set: (exp_hour, canonical issuer, serial octets).
// ORDER = position of the entry in the ROUND's global log order (Σ n of the lower ranks + index in the shard, < 2^32):
// among entries that bring the same new key in one round the lowest order keeps WasUnknown, as in the reference loop
// over one log (cmd/ct-fetch/ct-fetch.go:191-235 with the default numThreads = 1).
//
// KeyRec32 — the owner-computes exchange's record (SURVEY.md §8(e)(ii) sized it at 16–32 bytes): serials up to 20
// octets (RFC 5280's limit; everything a conforming CA issues).
struct __attribute__((aligned(16))) KeyRec32 {
  unsigned long long meta;  // key_meta(exp_hour, canonical issuer, serial_len)
  unsigned long long s0, s1;
  uint32_t s2;              // serial octets 16..19
  uint32_t ord;             // order of the entry in the round
};
static_assert(sizeof(KeyRec32) == 32, "KeyRec32");

// KeyRec — 64 bytes: serials up to CTMR_MAX_SERIAL (40) octets.  The Bloom variant's candidate records (few), and the
// owner-computes exchange's records for the rare 21..40-octet serials.
struct KeyRec {
  unsigned long long meta;
  unsigned long long s[5];  // serial octets
  uint32_t src;             // index of the entry in the sender's batch
  uint32_t owner;           // destination rank
  unsigned long long pad;   // owner exchange: order of the entry in the round (low 32 bits); Bloom: global order (log index)
};
static_assert(sizeof(KeyRec) == 64, "KeyRec");

// owner(key) = a second hash of the key → [0, world) by multiply-shift (no 64-bit division in the map kernel)
__host__ __device__ inline uint32_t key_owner_h(unsigned long long h, uint32_t world) {
  const uint32_t r = (uint32_t)(mixk(h ^ 0x5bd1e995ull) >> 32);
  return (uint32_t)(((unsigned long long)r * world) >> 32);
}

// Bloom filter of the keys a rank holds: blocked Bloom, one 64-bit word per key, 4 bits inside it — one 8-byte atomicOr
// to add, one 8-byte load per peer to probe.  At 16 filter bits per key the false-positive rate is ≈ 0.5 % (only extra
// key traffic, never a wrong answer).
__host__ __device__ inline void bloom_pos(unsigned long long h, uint64_t wmask, uint64_t& word,
                                          unsigned long long& bits) {
  const unsigned long long g = mixk(h ^ 0xa0761d6478bd642full);
  word = g & wmask;
  bits = (1ull << ((g >> 40) & 63)) | (1ull << ((g >> 46) & 63)) | (1ull << ((g >> 52) & 63)) |
         (1ull << ((g >> 58) & 63));
}

// What the fused map kernel needs besides the table in the global-dedup modes (k_map_fused<…, MODE>).
enum : int { XM_LOCAL = 0, XM_OWNER = 1, XM_BLOOM = 2 };
struct XchgArgs {
  // XM_OWNER: keys owned by another rank are not inserted here; they leave as 32-byte records
  uint32_t world, rank;
  KeyRec32* stage;          // n records of room; the records of wave w (entries [64w, 64w+64)) start at stage + 64w,
                            // grouped by owner (ascending), log order inside a group
  uint8_t* wave_cnt;        // MAX_WORLD bytes per wave: records per owner
  // XM_BLOOM: the rank's cumulative filter
  unsigned long long* bloom;
  uint64_t bloom_wmask;
};

}  // namespace ctmr
