// kernels/misc.h — auxiliary kernels: whole-certificate SHA-256, RemoteCache point ops and scans, the synthetic generator.
// gfx950 (CDNA4, wave64) only; part of kernels.h, which includes the pieces in dependency order.
#pragma once
#include "meta.h"

namespace ctmr {

// ------------------------------------------------------------------ whole-certificate SHA-256 (auxiliary)
// NOT on the reference's path — it never hashes a leaf certificate (the only SHA-256 is Issuer.ID's,
// storage/types.go:155-159).  This is the kernel BASELINE.json's north_star names literally ("one-cert-per-lane
// SHA-256 with round constants in LDS"): the fingerprint CT tooling identifies certificates by.  VALU-bound
// (≈2 000 instructions per 64-byte block), not HBM-bound: reported against its own roofline.
// Full blocks are fetched as four unaligned 16-byte loads per lane; the padded tail goes through the byte path.
__global__ void __launch_bounds__(256) k_fingerprint(const uint8_t* payload, const uint64_t* offsets,
                                                     const uint64_t* ends, uint64_t n, uint32_t* digests) {
  __shared__ uint32_t kc[64];
  if (threadIdx.x < 64) kc[threadIdx.x] = K256[threadIdx.x];
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t lo, hi;
  cert_range(offsets, ends, i, lo, hi);
  const uint64_t len64 = hi - lo;
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  const uint8_t* p = payload + lo;
  const uint64_t full = len64 >> 6;
  for (uint64_t b = 0; b < full; b++) {
    uint32_t w[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const U16 v = *(const U16*)(p + b * 64 + q * 16);
      w[4 * q] = __builtin_bswap32(v.a); w[4 * q + 1] = __builtin_bswap32(v.b);
      w[4 * q + 2] = __builtin_bswap32(v.c); w[4 * q + 3] = __builtin_bswap32(v.d);
    }
    sha256_compress(h, w, kc);
  }
  // tail: 0..63 message bytes, 0x80, zeros, 64-bit bit length — one or two blocks
  const uint32_t rem = (uint32_t)(len64 & 63u);
  const uint32_t nt = rem + 9 > 64 ? 2u : 1u;
  const uint8_t* t = p + full * 64;
  for (uint32_t b = 0; b < nt; b++) {
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const uint32_t pos = b * 64 + k * 4;
      uint32_t v = 0;
      if (pos + 4 <= rem) {
        v = __builtin_bswap32(((const U4*)(t + pos))->a);
      } else if (pos <= rem) {
        const uint32_t r = rem - pos;  // 0..3 message bytes in this word
        const uint32_t raw = r ? ((const U4*)(t + pos))->a : 0u;  // ≤ 3 bytes past the certificate: CTMR_PAYLOAD_PAD
        const uint32_t m = r ? (raw & (0xffffffffu >> (8 * (4 - r)))) : 0u;
        v = __builtin_bswap32(m | (0x80u << (8 * r)));
      }
      w[k] = v;
    }
    if (b == nt - 1) {
      w[14] = (uint32_t)((len64 * 8ull) >> 32);
      w[15] = (uint32_t)(len64 * 8ull);
    }
    sha256_compress(h, w, kc);
  }
  uint4* out = (uint4*)(digests + i * 8);  // big-endian digest bytes
  out[0] = make_uint4(__builtin_bswap32(h[0]), __builtin_bswap32(h[1]), __builtin_bswap32(h[2]), __builtin_bswap32(h[3]));
  out[1] = make_uint4(__builtin_bswap32(h[4]), __builtin_bswap32(h[5]), __builtin_bswap32(h[6]), __builtin_bswap32(h[7]));
}

// ------------------------------------------------------------------ RemoteCache point ops
// op: 0 = SetInsert, 1 = SetContains, 2 = SetRemove.  result[0] = 1 when inserted / present /
// removed; result[1] = SID_FULL marker on a full table.  SetInsert: `my_ref` is a fresh arena cell the host reserved
// for this call (it stays unused when the member was known).
__global__ void k_set_op(Table t, unsigned long long meta, unsigned long long s0,
                         unsigned long long s1, unsigned long long s2, unsigned long long s3,
                         unsigned long long s4, int op, unsigned long long my_ref,
                         unsigned long long* issuer_counts, PairSlot* pairs, uint64_t pmask,
                         unsigned long long* bloom, uint64_t bloom_wmask, uint32_t* result) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const unsigned long long s[5] = {s0, s1, s2, s3, s4};
  bool created = false;
  unsigned long long holder = 0ull;
  if (op == 0) {  // the cell first, visible, then the word that points to it
    KeyCell* c = t.arena + my_ref;
    st_agent(&c->meta, meta);
#pragma unroll
    for (int k = 0; k < 5; k++) st_agent(&c->s[k], s[k]);
    __hip_atomic_store(&c->ord, 0xffffffffu, __ATOMIC_RELAXED, AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const uint32_t sid = index_upsert(t, meta, s, my_ref, op == 0, &created, &holder);
  result[0] = 0;
  result[1] = sid == SID_FULL;
  if (sid == SID_FULL) return;
  const uint32_t canon = (uint32_t)(meta >> 32) & 0xffffffu;
  if (op == 0) {
    result[0] = created;
    if (created) atomicAdd(&issuer_counts[canon], 1ull);
    if (bloom) {  // Bloom-variant global dedup: every key this rank holds is in its filter, point inserts included
      uint64_t word;
      unsigned long long bits;
      bloom_pos(key_hash(meta, s), bloom_wmask, word, bits);
      atomicOr(&bloom[word], bits);
    }
  } else if (op == 1) {
    result[0] = sid != SID_NONE;
  } else if (sid != SID_NONE) {
    const bool shadow = (t.arena[holder & REF_MASK].meta & CELL_SHADOW) != 0;  // counted by another rank: nothing to take off here
    t.index[sid] = IDX_TOMB;
    if (!shadow) atomicAdd(&issuer_counts[canon], (unsigned long long)-1ll);
    result[0] = 1;
  }
}

// Drop every member whose (exp_hour, canonical issuer) matches, or — with any_key — every
// member with exp_hour*3600 <= now (Redis EXPIREAT set by knowncertificates.go:98-104).  One index word per thread; the
// key of a live word is read from its cell.
__global__ void __launch_bounds__(256) k_sweep(Table t, int any_key,
                                               long long now, uint32_t exp_hour_key, uint32_t canon_key,
                                               unsigned long long* issuer_counts, PairSlot* pairs,
                                               uint64_t pmask, unsigned long long* removed) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j > t.mask) return;
  const unsigned long long w = t.index[j];
  if (w == 0ull || w == IDX_TOMB) return;
  const unsigned long long w1 = t.arena[w & REF_MASK].meta;
  const int32_t eh = (int32_t)(uint32_t)w1;
  const uint32_t canon = (uint32_t)(w1 >> 32) & 0xffffffu;
  const bool hit = any_key ? ((long long)eh * 3600 <= now) : ((uint32_t)eh == exp_hour_key && canon == canon_key);
  if (!hit) return;
  t.index[j] = IDX_TOMB;
  if (w1 & CELL_SHADOW) return;  // counted by another rank (Bloom-variant global dedup)
  atomicAdd(&issuer_counts[canon], (unsigned long long)-1ll);
  atomicAdd(removed, 1ull);
}

// Table growth: every live word of the old index is re-inserted into the new (zeroed) one — the keys are distinct, so
// one CAS claims the slot.  The word moves verbatim (tag and ref do not depend on the table size; the arena — copied
// as it is when it grows too — keeps every cell where it was).  Tombstones stay behind, which is how their slots are
// recovered.  The slot in the new index comes from the key's hash: read from the cell.
__global__ void __launch_bounds__(256) k_rehash(const unsigned long long* old_index, uint64_t old_slots, Table t,
                                                unsigned long long* moved) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= old_slots) return;
  const unsigned long long w = old_index[j];
  if (w == 0ull || w == IDX_TOMB) return;
  const KeyCell* c = t.arena + (w & REF_MASK);
  unsigned long long s[5];
#pragma unroll
  for (int k = 0; k < 5; k++) s[k] = c->s[k];
  uint64_t q = key_hash(c->meta & ~CELL_SHADOW, s) & t.mask;
  for (uint64_t probes = 0;; probes++) {  // the new index holds at most half as many members as it has slots: this terminates
    if (atomicCAS(&t.index[q], 0ull, w) == 0ull) break;
    q = probe_next(q, probes, t.mask);
  }
  atomicAdd(moved, 1ull);
}

// Arena compaction: a batch reserves one cell per ENTRY, and the cell of an entry that was not new (a duplicate, a filtered
// certificate) is never referenced — in a deployment that keeps meeting known certificates (a restart that re-reads a log,
// overlapping fetch ranges) those cells would fill the arena with nothing.  Before a round that does not fit any more, and
// when at least a quarter of the used cells is garbage, the live cells are moved into a fresh arena: every live index
// word (the words stay where they are: a word's slot depends on the key's hash only) gets the next free cell of the new
// arena, wave-aggregated, and its ref is rewritten in place.  Cells lose their entry order, which nothing relies on
// between rounds ("earlier round" = ref < ref0 still holds: every moved cell lies below the next round's ref0).
__global__ void __launch_bounds__(256) k_arena_compact(unsigned long long* index, uint64_t nslots, const KeyCell* old_arena,
                                                       KeyCell* new_arena, unsigned long long* used) {
  __shared__ uint32_t wcnt[4];
  __shared__ unsigned long long blk_base;
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long w = j < nslots ? index[j] : 0ull;
  const bool live = (w != 0ull) & (w != IDX_TOMB);
  const unsigned long long m = __ballot(live);
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wcnt[wv] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {  // one atomic on the shared counter per workgroup
    const uint32_t tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    blk_base = tot ? atomicAdd(used, (unsigned long long)tot) : 0ull;
  }
  __syncthreads();
  if (!live) return;
  unsigned long long base = blk_base;
  for (uint32_t k = 0; k < wv; k++) base += wcnt[k];
  const unsigned long long nref = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
  const uint4* src = (const uint4*)(old_arena + (w & REF_MASK));
  uint4* dst = (uint4*)(new_arena + nref);
  const uint4 c0 = src[0], c1 = src[1], c2 = src[2], c3 = src[3];
  dst[0] = c0; dst[1] = c1; dst[2] = c2; dst[3] = c3;
  index[j] = (w & ~REF_MASK) | nref;
}

// Rebuild the (expDate, issuer) → SCARD table from the known-certificate table (lazy: only the
// statistics-style queries SetCardinality / Exists / KeysToChan need it).
__global__ void __launch_bounds__(256) k_build_pairs(Table t, PairSlot* pairs, uint64_t pmask, unsigned long long* full) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j > t.mask) return;
  const unsigned long long w = t.index[j];
  if (w == 0ull || w == IDX_TOMB) return;
  const unsigned long long w1 = t.arena[w & REF_MASK].meta;
  if (w1 & CELL_SHADOW) return;
  const uint32_t canon = (uint32_t)(w1 >> 32) & 0xffffffu;
  if (!pair_add(pairs, pmask, ((unsigned long long)(canon + 1) << 32) | (uint32_t)w1, 1)) atomicAdd(full, 1ull);
}

// SetList / SetToChan: gather the serials of one set.  out entries are 48 bytes:
// [u32 len][40 bytes serial][u32 pad].
__global__ void __launch_bounds__(256) k_list(Table t, uint32_t exp_hour_key,
                                              uint32_t canon_key, uint8_t* out, uint64_t cap,
                                              unsigned long long* count) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j > t.mask) return;
  const unsigned long long w = t.index[j];
  if (w == 0ull || w == IDX_TOMB) return;
  const KeyCell* c = t.arena + (w & REF_MASK);
  const unsigned long long w1 = c->meta;
  if (w1 & CELL_SHADOW) return;
  if ((uint32_t)w1 != exp_hour_key || ((uint32_t)(w1 >> 32) & 0xffffffu) != canon_key) return;
  const unsigned long long k = atomicAdd(count, 1ull);
  if (k >= cap) return;
  unsigned long long* o = (unsigned long long*)(out + k * 48);
  o[0] = (w1 >> 56) & 0x3full;
#pragma unroll
  for (int q = 0; q < 5; q++) o[1 + q] = c->s[q];
}

// KeysToChan: dump the non-empty (expDate, issuer) pairs.
__global__ void __launch_bounds__(256) k_pairs(const PairSlot* pairs, uint64_t npairs,
                                               unsigned long long* out, uint64_t cap,
                                               unsigned long long* count) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= npairs) return;
  const unsigned long long k = pairs[j].key, c = pairs[j].count;
  if (k == 0ull || c == 0ull) return;
  const unsigned long long at = atomicAdd(count, 1ull);
  if (at >= cap) return;
  out[2 * at] = k;
  out[2 * at + 1] = c;
}

// ------------------------------------------------------------------ synthetic generator
__global__ void __launch_bounds__(256) k_synth_len(SynthCfg c, uint64_t first, uint64_t n,
                                                   uint64_t* offsets) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  BackWriter w{nullptr, SYNTH_MAX_LEN};
  uint32_t iss;
  uint8_t et;
  synth_leaf_emit(c, first + i, w, iss, et);
  offsets[i + 1] = SYNTH_MAX_LEN - w.pos;
  if (i == 0) offsets[0] = 0;
}

// Entry-view form of the same certificates: certificate i at starts[i] (a multiple of `align` once scanned), ends[i] =
// starts[i] + its length.  k_synth_len_view leaves the PADDED lengths in starts[1..n] (scanned by the host) and the true
// lengths in ends[]; k_synth_emit_view writes the bytes and turns ends[] into end positions.
__global__ void __launch_bounds__(256) k_synth_len_view(SynthCfg c, uint64_t first, uint64_t n, uint32_t align,
                                                        uint64_t* starts, uint64_t* ends) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  BackWriter w{nullptr, SYNTH_MAX_LEN};
  uint32_t iss;
  uint8_t et;
  synth_leaf_emit(c, first + i, w, iss, et);
  const uint64_t len = SYNTH_MAX_LEN - w.pos;
  starts[i + 1] = (len + align - 1u) & ~(uint64_t)(align - 1u);  // align is a power of two
  ends[i] = len;
  if (i == 0) starts[0] = 0;
}

__global__ void __launch_bounds__(256) k_synth_emit_view(SynthCfg c, uint64_t first, uint64_t n, const uint64_t* starts,
                                                         uint64_t* ends, uint8_t* payload, uint32_t* issuer_idx,
                                                         uint8_t* entry_type) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t len = (uint32_t)ends[i];
  BackWriter w{payload + starts[i], len};
  uint32_t iss;
  uint8_t et;
  synth_leaf_emit(c, first + i, w, iss, et);
  issuer_idx[i] = iss;
  entry_type[i] = et;
  ends[i] = starts[i] + len;
}

__global__ void __launch_bounds__(256) k_synth_emit(SynthCfg c, uint64_t first, uint64_t n,
                                                    const uint64_t* offsets, uint8_t* payload,
                                                    uint32_t* issuer_idx, uint8_t* entry_type) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t len = (uint32_t)(offsets[i + 1] - offsets[i]);
  BackWriter w{payload + offsets[i], len};
  uint32_t iss;
  uint8_t et;
  synth_leaf_emit(c, first + i, w, iss, et);
  issuer_idx[i] = iss;
  entry_type[i] = et;
}


// raw get-entries form: lens[2i] = leaf_input bytes, lens[2i+1] = extra_data bytes (scanned into bounds by the host)
__global__ void __launch_bounds__(256) k_synth_entries_len(SynthCfg c, uint64_t first, uint64_t n, uint64_t* bounds) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  BackWriter w{nullptr, SYNTH_ENTRY_MAX};
  const uint32_t leaf = synth_entry_emit(c, first + i, w);
  bounds[2 * i + 1] = leaf;
  bounds[2 * i + 2] = SYNTH_ENTRY_MAX - w.pos - leaf;
  if (i == 0) bounds[0] = 0;
}

__global__ void __launch_bounds__(256) k_synth_entries_emit(SynthCfg c, uint64_t first, uint64_t n,
                                                            const uint64_t* bounds, uint8_t* blob) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t len = (uint32_t)(bounds[2 * i + 2] - bounds[2 * i]);
  BackWriter w{blob + bounds[2 * i], len};
  synth_entry_emit(c, first + i, w);
}

}  // namespace ctmr
