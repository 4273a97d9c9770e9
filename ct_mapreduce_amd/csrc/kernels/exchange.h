// kernels/exchange.h — cross-GPU global dedup: the owner-computes key exchange and the Bloom pre-filter variant.
// gfx950 (CDNA4, wave64) only; part of kernels.h, which includes the pieces in dependency order.
#pragma once
#include "reduce.h"

namespace ctmr {

// ------------------------------------------------------------------ cross-GPU key exchange (owner-computes)
// Global dedup over G GPUs: every key has one OWNER = key_owner_h(hash) ∈ [0, G).  Round 3 shape:
//   sender   k_map_fused<…, XM_OWNER> inserts the keys this rank owns itself (the fused path, unchanged) and leaves the
//            others as 32-byte records in a wave-compacted staging array; k_key_blockcount + k_scan_blocks +
//            k_key_gather turn that into per-owner partitions, ascending log order inside a partition (what the
//            all-to-all sends).  Serials of 21..40 octets (rare) leave as 64-byte records (k_xl_export).
//   owner    k_keys_insert / k_keys_insert2 / k_keys_resolve: the two-pass insert of reduce.h over received records, in
//            the SAME round as the owner's own shard — every key cell carries the entry's ORDER in the round (keyrec.h), so
//            the lowest log index wins whoever held the entry; a received record that beats an entry of the owner's own
//            shard marks it (mark_dup_ord); one "was unknown" byte per record goes back.
//   sender   k_apply_lost: records leave the map optimistically NEW (ES_REMOTE); only the losers are touched.
constexpr uint32_t SID_DEFER = 0x80000000u;  // slot_id bit (owner-side kernels): candidate slot, full compare in pass 2

struct KeyView {
  unsigned long long meta;
  unsigned long long s[5];
  uint32_t ord;
};
__device__ __forceinline__ KeyView load_key(const KeyRec32* keys, uint64_t i) {
  const uint4* p = (const uint4*)(keys + i);
  const uint4 a = p[0], b = p[1];
  KeyView k;
  k.meta = (unsigned long long)a.x | ((unsigned long long)a.y << 32);
  k.s[0] = (unsigned long long)a.z | ((unsigned long long)a.w << 32);
  k.s[1] = (unsigned long long)b.x | ((unsigned long long)b.y << 32);
  k.s[2] = (unsigned long long)b.z;
  k.s[3] = 0;
  k.s[4] = 0;
  k.ord = b.w;
  return k;
}
__device__ __forceinline__ KeyView load_key(const KeyRec* keys, uint64_t i) {
  const KeyRec r = keys[i];
  KeyView k;
  k.meta = r.meta;
#pragma unroll
  for (int q = 0; q < 5; q++) k.s[q] = r.s[q];
  k.ord = (uint32_t)r.pad;
  return k;
}

// Sender, after the fused map: records per (owner, 1024-entry block) from the per-wave counts (owner-major, as
// k_scan_blocks and k_key_gather want them)
__global__ void __launch_bounds__(256) k_key_blockcount(const uint8_t* wave_cnt, uint64_t n_waves, uint64_t nb,
                                                        uint32_t world, uint32_t* cnt) {
  const uint64_t blk = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (blk >= nb) return;
  uint32_t c[MAX_WORLD];
#pragma unroll
  for (uint32_t o = 0; o < MAX_WORLD; o++) c[o] = 0;
  for (uint32_t w = 0; w < 16; w++) {
    const uint64_t wave = blk * 16 + w;
    if (wave >= n_waves) break;
    const uint4 v = *(const uint4*)(wave_cnt + wave * MAX_WORLD);
    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (uint32_t o = 0; o < MAX_WORLD; o++) c[o] += (x[o >> 2] >> (8 * (o & 3))) & 0xffu;
  }
  for (uint32_t o = 0; o < world; o++) cnt[(uint64_t)o * nb + blk] = c[o];
}

// Sender: staging array → per-owner partitions.  One 1024-thread block per 1024 entries (16 waves of the map kernel);
// thread (w, p) moves record p of wave w's compacted group list.  Stable: ascending log order inside a partition.
__global__ void __launch_bounds__(1024) k_key_gather(const KeyRec32* stage, const uint8_t* wave_cnt, uint64_t n_waves,
                                                     uint64_t nb, uint32_t world, const uint64_t* base, KeyRec32* out) {
  __shared__ uint8_t wc[16][MAX_WORLD];     // records of wave w for owner o
  __shared__ uint16_t pre[16][MAX_WORLD];   // … of the earlier waves of this block
  const uint32_t w = threadIdx.x >> 6, p = threadIdx.x & 63u;
  const uint64_t wave = (uint64_t)blockIdx.x * 16 + w;
  if (p < MAX_WORLD) wc[w][p] = wave < n_waves ? wave_cnt[wave * MAX_WORLD + p] : (uint8_t)0;
  __syncthreads();
  if (threadIdx.x < 16 * MAX_WORLD) {
    const uint32_t ww = threadIdx.x / MAX_WORLD, o = threadIdx.x % MAX_WORLD;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < ww; k++) sum += wc[k][o];
    pre[ww][o] = (uint16_t)sum;
  }
  __syncthreads();
  uint32_t cum = 0, owner = KEY_NO_OWNER, rank_in = 0;
  for (uint32_t o = 0; o < world; o++) {
    const uint32_t c = wc[w][o];
    if (owner == KEY_NO_OWNER && p < cum + c) {
      owner = o;
      rank_in = p - cum;
    }
    cum += c;
  }
  if (owner == KEY_NO_OWNER) return;
  const uint4* src = (const uint4*)(stage + wave * 64 + p);
  uint4* dst = (uint4*)(out + base[(uint64_t)owner * nb + blockIdx.x] + pre[w][owner] + rank_in);
  const uint4 a = src[0], b = src[1];
  dst[0] = a;
  dst[1] = b;
}

// Sender, rare: the ES_REMOTE entries whose serial has 21..40 octets, as 64-byte records (unordered append; the host
// sorts the few of them by (owner, order))
__global__ void __launch_bounds__(256) k_xl_export(InsertArgs a, uint32_t world, KeyRec* out, uint64_t cap,
                                                   unsigned long long* count) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  if (ent_state(a.ent[i]) != ES_REMOTE) return;
  const uint4* rp = (const uint4*)(a.records + i);
  const uint4 r0 = rp[0];
  const uint32_t slen = r0.x >> 16;
  if (slen <= 20u) return;
  const uint4 r1 = rp[1];
  unsigned long long s[5];
  record_key(a, i, r0, r1, s);
  const unsigned long long meta = key_meta((int32_t)r0.y, a.canon[r0.z], slen);
  const unsigned long long at = atomicAdd(count, 1ull);
  if (at >= cap) return;
  KeyRec k;
  k.meta = meta;
#pragma unroll
  for (int q = 0; q < 5; q++) k.s[q] = s[q];
  k.src = (uint32_t)i;
  k.owner = key_owner_h(key_hash(meta, s), world);
  k.pad = a.ord_base + (uint32_t)i;
  out[at] = k;
}

// Owner side, pass 1 / pass 2 / resolve on received key records (same protocol as the fused map's pass 1 and k_insert2).
// Received record k of the round gets the arena cell recv_ref0 + k (the cells of a wave as one contiguous store, like
// the map's); `loc` describes the owner's OWN shard of the round (ent, ord_base, n, ref0; records_local): received
// records that beat one of its entries mark it.  A word of THIS round = ref >= loc.ref0 (the received cells lie behind
// the shard's own).
template <class Rec>
__global__ void __launch_bounds__(256) k_keys_insert(const Rec* keys, uint64_t n, Table t, unsigned long long round_ref0,
                                                     unsigned long long recv_ref0, uint32_t* slot_id) {
  __shared__ __attribute__((aligned(16))) uint4 img[4][64 * 4];  // per wave: 64 key cells (store_cells_wave)
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  KeyView k{};
  bool keyed = false;
  if (i < n) {
    k = load_key(keys, i);
    const unsigned long long h = key_hash(k.meta, k.s);
    const unsigned long long mine = idx_word(h, recv_ref0 + i);
    uint64_t j = h & t.mask;
    uint32_t sid = SID_FULL;
    // meta == 0 (no VALID bit): a record its sender withdrew — the EC point of the certificate turned out to be off its
    // curve after the key had been staged (k_ec_resolve).  Never inserted, answered "not new"; the sender's entry is a
    // parse error by then and ignores the answer.
    const uint64_t limit = k.meta == 0ull ? 0ull : t.mask + 1ull;
    if (k.meta == 0ull) sid = SID_NONE;
    keyed = k.meta != 0ull;
    for (uint64_t probes = 0; probes < limit; probes++) {
      const unsigned long long old = atomicCAS(&t.index[j], 0ull, mine);
      if (old == 0ull) {
        sid = (uint32_t)j;
        break;
      }
      if (old != IDX_TOMB && idx_same_tag(old, h)) {
        const unsigned long long oref = old & REF_MASK;
        if (oref < round_ref0) {  // an earlier round's key
          const KeyCell* c = t.arena + oref;
          bool eq = (c->meta & ~CELL_SHADOW) == k.meta;
#pragma unroll
          for (int q = 0; q < 5; q++) eq = eq && c->s[q] == k.s[q];
          if (eq) {
            sid = SID_DUP_OLD;
            break;
          }
        } else {
          sid = (uint32_t)j | SID_DEFER;
          break;
        }
      }
      j = probe_next(j, probes, t.mask);
    }
    slot_id[i] = sid;
  }
  const uint64_t first = (uint64_t)blockIdx.x * 256 + 64u * wv;
  if (first < n) store_cells_wave(t.arena + recv_ref0 + first, n - first, img[wv], lane, keyed, k.meta, k.s, k.ord);
}

template <class Rec>
__global__ void __launch_bounds__(256) k_keys_insert2(const Rec* keys, uint64_t n, InsertArgs loc, unsigned long long recv_ref0,
                                                      ctmr_record* records_local, uint32_t* slot_id) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t sid = slot_id[i];
  if (sid >= SID_DUP_OLD || !(sid & SID_DEFER)) return;
  sid &= ~SID_DEFER;
  const KeyView k = load_key(keys, i);
  const unsigned long long h = key_hash(k.meta, k.s);
  const unsigned long long my_ref = recv_ref0 + i;
  const unsigned long long w = ld_agent(&loc.t.index[sid]);
  if (!cell_equals(loc.t.arena + (w & REF_MASK), k.meta, k.s)) {  // another key under the same tag
    bool created;
    unsigned long long holder = 0ull;
    sid = index_upsert(loc.t, k.meta, k.s, my_ref, true, &created, &holder);
    slot_id[i] = sid;
    if (sid >= SID_DUP_OLD || created) return;  // full (SID_FULL), or holds its slot now
    if ((holder & REF_MASK) < loc.ref0) {       // found in an earlier round after all
      slot_id[i] = SID_DUP_OLD;
      return;
    }
  }
  slot_id[i] = sid;
  (void)settle_order(loc, records_local, sid, h, my_ref, k.ord);  // whether it kept the word is read back in k_keys_resolve
}

template <class Rec>
__global__ void __launch_bounds__(1024) k_keys_resolve(const Rec* keys, uint64_t n, uint64_t nb, Table t,
                                                       unsigned long long recv_ref0, const uint32_t* slot_id, uint8_t* flags,
                                                       unsigned long long* issuer_counts, DevStats* stats) {
  __shared__ uint32_t ih[RES_LDS_ISSUERS];
  __shared__ uint32_t cnt[2];
  for (uint32_t k = threadIdx.x; k < RES_LDS_ISSUERS; k += 1024) ih[k] = 0;
  if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
  __syncthreads();
  for (uint64_t blk = blockIdx.x; blk < nb; blk += gridDim.x) {
    const uint64_t i = blk * 1024 + threadIdx.x;
    bool is_new = false, is_full = false;
    uint32_t canon = 0;
    if (i < n) {
      const uint32_t sid = slot_id[i];
      if (sid == SID_FULL) {
        is_full = true;
      } else if (sid < SID_DUP_OLD) {
        // new here = the slot's word points to THIS record's cell (nobody with a lower order of the round took it)
        is_new = (t.index[sid] & REF_MASK) == recv_ref0 + i;
        canon = (uint32_t)(t.arena[recv_ref0 + i].meta >> 32) & 0xffffffu;
      }
      flags[i] = is_new ? 1 : 0;
    }
    if (is_new && canon < RES_LDS_ISSUERS) atomicAdd(&ih[canon], 1u);
    wave_agg_add(is_new && canon >= RES_LDS_ISSUERS, canon, issuer_counts);
    const unsigned long long m_new = __ballot(is_new), m_full = __ballot(is_full);
    if ((threadIdx.x & 63) == 0) {
      if (m_new) atomicAdd(&cnt[0], (uint32_t)__popcll(m_new));
      if (m_full) atomicAdd(&cnt[1], (uint32_t)__popcll(m_full));
    }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < RES_LDS_ISSUERS; k += 1024)
    if (ih[k]) atomicAdd(&issuer_counts[k], (unsigned long long)ih[k]);
  if (threadIdx.x == 0 && cnt[0]) atomicAdd(&stats->n_new, (unsigned long long)cnt[0]);
  if (threadIdx.x == 1 && cnt[1]) atomicAdd(&stats->n_full, (unsigned long long)cnt[1]);
}

// Sender side: the returned bytes.  Records left the map optimistically NEW: only a record whose byte is 0 (the key
// was known to its owner, or a lower order of this round holds it) is touched — its ent[] word and record flag, the NEW
// count of its 1024-entry block, and the `lost` counter.
template <class Rec>
__global__ void __launch_bounds__(256) k_apply_lost(const Rec* sent, const uint8_t* flags, uint64_t n_keys, InsertArgs loc,
                                                    ctmr_record* records, uint32_t* blk_new, unsigned long long* lost) {
  // Four records per thread and ONE atomic on the global counter per workgroup: with one per wave (160 000 of them on one
  // address for 10 M records, nearly every wave holding a loser) the counter's serialisation was the kernel — 1.85 ms per
  // 10.3 M records, more than the owner-side insert of the same records (round 4, rocprofv3 of rank_cost_at_world.py).
  __shared__ uint32_t blk_lost;
  if (threadIdx.x == 0) blk_lost = 0;
  __syncthreads();
  const uint64_t k0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4u;
  uint32_t mine = 0;
#pragma unroll
  for (uint32_t q = 0; q < 4; q++) {
    const uint64_t k = k0 + q;
    if (k < n_keys && flags[k] == 0) {
      const uint32_t ord = ((const uint32_t*)(sent + k))[sizeof(Rec) == 32 ? 7 : 14];
      const uint32_t li = ord - loc.ord_base;
      if ((uint64_t)li < loc.n && ent_state(loc.ent[li]) == ES_REMOTE) {
        mark_dup(loc.ent, records, li);
        atomicSub(&blk_new[li >> 10], 1u);
        mine++;
      }
    }
  }
  // (count per wave: three ballots over the bits of 0..4)
  const unsigned long long b0 = __ballot(mine & 1u), b1 = __ballot(mine & 2u), b2 = __ballot(mine & 4u);
  const uint32_t wsum = (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
  if (wsum && (threadIdx.x & 63) == 0) atomicAdd(&blk_lost, wsum);
  __syncthreads();
  if (threadIdx.x == 0 && blk_lost) atomicAdd(lost, (unsigned long long)blk_lost);
}

// ------------------------------------------------------------------ cross-GPU dedup, Bloom pre-filter variant
// The all-gather of per-GPU Bloom fingerprints `north_star` names, made exact.  Every rank keeps
// its OWN known-certificate table (the ordinary fused map + insert runs unchanged) and a cumulative Bloom filter of
// every key it ever found locally new.  Per round: the filters are all-gathered; a rank probes its locally-new keys
// against the other ranks' filters — a Bloom filter has no false negatives, so a key that hits no peer filter exists
// on no other rank and needs no exchange at all; a key that hits peer p's filter (a real cross-rank duplicate or a
// false positive) is sent to p, which looks it up EXACTLY in its table and answers "known here before you": found
// in a cell of an earlier round, or found in this round under a lower global order (= lower log index).  Exactly one rank — the
// lowest log index — keeps WasUnknown for each key.  The asker then clears the flag, takes the key out of its
// per-issuer count and marks its slot SHADOW (known for dedup, not counted or listed: the sets of the ranks stay
// disjoint, so Σ over ranks of SCARD / per-issuer counts is the global value, as in the owner-computes variant).
//
// Filter: blocked Bloom, one 64-bit word per key, 4 bits inside it — one 8-byte atomicOr to add, one 8-byte load per
// peer to probe.  At 16 filter bits per key the false-positive rate is ≈ 0.5 % (only extra key traffic, never a wrong
// answer).

__device__ __forceinline__ bool entry_key(const InsertArgs& a, uint64_t i, unsigned long long& meta,
                                          unsigned long long s[5]) {
  const uint4* rp = (const uint4*)(a.records + i);
  const uint4 r0 = rp[0];
  if ((r0.x & 0xffu) != CTMR_ST_PASS) return false;
  const uint32_t slen = r0.x >> 16;
  if (slen > CTMR_MAX_SERIAL) return false;  // host-side set, shard-local
  const uint4 r1 = rp[1];
  record_key(a, i, r0, r1, s);
  meta = key_meta((int32_t)r0.y, a.canon[r0.z], slen);
  return true;
}

// key of entry i when it is a locally-new member of the device set (long serials stay shard-local on the host)
__device__ __forceinline__ bool entry_new_key(const InsertArgs& a, uint64_t i, unsigned long long& meta,
                                              unsigned long long s[5]) {
  const uint32_t head = *(const uint32_t*)(a.records + i);
  if ((head & 0xffu) != CTMR_ST_PASS || !((head >> 8) & CTMR_FL_WAS_UNKNOWN)) return false;
  return entry_key(a, i, meta, s);
}

__global__ void __launch_bounds__(256) k_bloom_add(InsertArgs a, unsigned long long* words, uint64_t wmask) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  unsigned long long meta, s[5];
  if (!entry_new_key(a, i, meta, s)) return;
  uint64_t word;
  unsigned long long bits;
  bloom_pos(key_hash(meta, s), wmask, word, bits);
  if ((ld_agent(&words[word]) & bits) != bits) atomicOr(&words[word], bits);
}

// The all-gather leaves the filters rank-major ([peer][word]): probing one key would touch one random 64-byte sector PER
// PEER (measured, round 3: 14.4 ms per 94 M keys against 7 peer filters — a third of the Bloom round on one rank).
// Interleaved ([word][peer]) the peers' words of one key are neighbours: one sector (two at 16 ranks) per key.  One
// streaming pass over the gathered buffer buys that.
__global__ void __launch_bounds__(256) k_filter_interleave(const unsigned long long* in, uint64_t n_words, uint32_t world,
                                                           unsigned long long* out) {
  const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= n_words) return;
  for (uint32_t p = 0; p < world; p++) out[w * world + p] = in[(uint64_t)p * n_words + w];
}

// pass A: peers whose filter holds the key (bit p of hit_out[i]) + per-(peer, 1024-entry block) counts.
// `filters` is the INTERLEAVED form ([word][peer], k_filter_interleave).
__global__ void __launch_bounds__(1024) k_bloom_probe(InsertArgs a, const unsigned long long* filters,
                                                      uint64_t n_words, uint32_t world, uint32_t rank, uint64_t nb,
                                                      uint16_t* hit_out, uint32_t* cnt) {
  __shared__ uint32_t c[MAX_WORLD];
  if (threadIdx.x < MAX_WORLD) c[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  uint32_t hit = 0;
  if (i < a.n) {
    unsigned long long meta, s[5];
    if (entry_new_key(a, i, meta, s)) {
      uint64_t word;
      unsigned long long bits;
      bloom_pos(key_hash(meta, s), n_words - 1, word, bits);
      const unsigned long long* f = filters + word * world;
      for (uint32_t p = 0; p < world; p++)
        if (p != rank && (f[p] & bits) == bits) hit |= 1u << p;
    }
    hit_out[i] = (uint16_t)hit;
  }
  for (uint32_t w = 0; w < world; w++) {
    const unsigned long long m = __ballot((hit >> w) & 1u);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&c[w], (uint32_t)__popcll(m));
  }
  __syncthreads();
  if (threadIdx.x < world) cnt[(uint64_t)threadIdx.x * nb + blockIdx.x] = c[threadIdx.x];
}

// pass B: stable scatter of the key records into the per-peer partitions (a key goes to every peer it hit);
// KeyRec.owner = destination, KeyRec.pad = global order of the entry (order_base + batch index)
__global__ void __launch_bounds__(1024) k_bloom_scatter(InsertArgs a, uint32_t world, uint64_t nb,
                                                        const uint16_t* hit_in, const uint64_t* base,
                                                        unsigned long long order_base, KeyRec* out) {
  __shared__ uint32_t wc[16][MAX_WORLD];
  const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t hit = i < a.n ? hit_in[i] : 0u;
  uint32_t my_rank[MAX_WORLD];
  for (uint32_t w = 0; w < world; w++) {
    const unsigned long long m = __ballot((hit >> w) & 1u);
    if (lane == 0) wc[wv][w] = (uint32_t)__popcll(m);
    my_rank[w] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  if (hit) {
    unsigned long long meta, s[5];
    entry_key(a, i, meta, s);
    const unsigned long long order = order_base + i;
    for (uint32_t w = 0; w < world; w++) {
      if (!((hit >> w) & 1u)) continue;
      uint32_t before = 0;
      for (uint32_t k = 0; k < wv; k++) before += wc[k][w];
      uint4* q = (uint4*)(out + base[(uint64_t)w * nb + blockIdx.x] + before + my_rank[w]);
      q[0] = make_uint4((uint32_t)meta, (uint32_t)(meta >> 32), (uint32_t)s[0], (uint32_t)(s[0] >> 32));
      q[1] = make_uint4((uint32_t)s[1], (uint32_t)(s[1] >> 32), (uint32_t)s[2], (uint32_t)(s[2] >> 32));
      q[2] = make_uint4((uint32_t)s[3], (uint32_t)(s[3] >> 32), (uint32_t)s[4], (uint32_t)(s[4] >> 32));
      q[3] = make_uint4((uint32_t)i, w, (uint32_t)order, (uint32_t)(order >> 32));
    }
  }
}

// read-only find (the batch that filled the table has completed: plain loads)
__device__ __forceinline__ uint32_t table_find(const Table& t, unsigned long long meta, const unsigned long long s[5]) {
  const unsigned long long h = key_hash(meta, s);
  uint64_t j = h & t.mask;
  for (uint64_t probes = 0; probes <= t.mask; probes++) {
    const unsigned long long w = t.index[j];
    if (w == 0ull) return SID_NONE;
    if (w != IDX_TOMB && idx_same_tag(w, h)) {
      const KeyCell* c = t.arena + (w & REF_MASK);
      bool eq = (c->meta & ~CELL_SHADOW) == meta;
#pragma unroll
      for (int k = 0; k < 5; k++) eq = eq && c->s[k] == s[k];
      if (eq) return (uint32_t)j;
    }
    j = probe_next(j, probes, t.mask);
  }
  return SID_NONE;
}

// Peer side: flags[k] = 1 when the key is known here before the asker's entry — since an earlier round (its cell lies
// below the round's first one), or since this round under a lower global order.
__global__ void __launch_bounds__(256) k_keys_lookup(const KeyRec* keys, uint64_t n, Table t, unsigned long long round_ref0,
                                                     unsigned long long order_base, uint8_t* flags) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const KeyRec k = keys[i];
  const uint32_t sid = table_find(t, k.meta, k.s);
  uint8_t f = 0;
  if (sid != SID_NONE) {
    const unsigned long long ref = t.index[sid] & REF_MASK;
    f = ref < round_ref0 || order_base + t.arena[ref].ord < k.pad;
  }
  flags[i] = f;
}

// Asker side: a flagged key loses WasUnknown (once, however many peers flagged it), leaves the per-issuer count and
// its cell becomes SHADOW; its ent[] word, the NEW count of its 1024-entry block and the `lost` counter follow, so that
// the NEW list is compacted from ent[] as after a plain batch.
__global__ void __launch_bounds__(256) k_bloom_apply(const KeyRec* sent, const uint8_t* flags, uint64_t n_keys,
                                                     ctmr_record* records, uint32_t* ent, uint32_t* blk_new, Table t,
                                                     unsigned long long* issuer_counts, unsigned long long* n_lost) {
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  bool lost = false;
  uint32_t canon = 0;
  if (k < n_keys && flags[k] != 0) {
    const KeyRec kr = sent[k];
    const uint32_t old = atomicAnd((uint32_t*)(records + kr.src), ~((uint32_t)CTMR_FL_WAS_UNKNOWN << 8));
    if ((old >> 8) & CTMR_FL_WAS_UNKNOWN) {
      lost = true;
      canon = (uint32_t)(kr.meta >> 32) & 0xffffffu;
      const uint32_t sid = table_find(t, kr.meta, kr.s);
      if (sid != SID_NONE) atomicOr(&t.arena[t.index[sid] & REF_MASK].meta, CELL_SHADOW);
      ((uint8_t*)(ent + kr.src))[0] = (uint8_t)(CTMR_ST_PASS | (ES_DUP << 3));
      atomicSub(&blk_new[kr.src >> 10], 1u);
    }
  }
  const unsigned long long ml = __ballot(lost);
  if (ml && (threadIdx.x & 63) == 0) atomicAdd(n_lost, (unsigned long long)__popcll(ml));
  unsigned long long todo = ml;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t c = __shfl(canon, leader);
    const unsigned long long same = __ballot(lost && canon == c) & todo;
    if ((int)(threadIdx.x & 63) == leader)
      atomicAdd(&issuer_counts[c], (unsigned long long)(-(long long)__popcll(same)));
    todo &= ~same;
  }
}

}  // namespace ctmr
