// kernels/exchange.h — cross-GPU global dedup: the owner-computes key exchange and the Bloom pre-filter variant.
// gfx950 (CDNA4, wave64) only; part of kernels.h, which includes the pieces in dependency order.
#pragma once
#include "reduce.h"

namespace ctmr {

// ------------------------------------------------------------------ cross-GPU key exchange
// Global dedup over G GPUs: every key has one OWNER = hash(key) mod G.  A
// rank exports the keys of its PASS entries partitioned by owner (ascending log index inside each
// partition), the partitions are exchanged (RCCL send/recv), the owner inserts what it received
// — concatenated in sender-rank order, which IS global log order because shards are contiguous
// log-index ranges — and returns one "was unknown" byte per key.
struct KeyRec {  // 64 bytes
  unsigned long long meta;  // key_meta(exp_hour, canonical issuer, serial_len)
  unsigned long long s[5];  // serial octets
  uint32_t src;             // index of the entry in the sender's batch
  uint32_t owner;
  unsigned long long pad;
};
static_assert(sizeof(KeyRec) == 64, "KeyRec");

constexpr uint32_t KEY_NO_OWNER = 0xffu;
constexpr uint32_t SID_DEFER = 0x80000000u;  // slot_id bit (owner-side kernels): candidate slot, full compare in pass 2
constexpr uint32_t MAX_WORLD = 16;

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

__device__ __forceinline__ uint32_t key_owner(unsigned long long meta, const unsigned long long s[5],
                                              uint32_t world) {
  return (uint32_t)(mixk(key_hash(meta, s) ^ 0x5bd1e995u) % world);
}

// pass A: owner of every entry + per-(owner, 1024-entry block) counts (owner-major layout)
__global__ void __launch_bounds__(1024) k_key_count(InsertArgs a, uint32_t world, uint64_t nb,
                                                    uint8_t* owner_out, uint32_t* cnt) {
  __shared__ uint32_t c[MAX_WORLD];
  if (threadIdx.x < MAX_WORLD) c[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  uint32_t owner = KEY_NO_OWNER;
  if (i < a.n) {
    unsigned long long meta, s[5];
    if (entry_key(a, i, meta, s)) owner = key_owner(meta, s, world);
    owner_out[i] = (uint8_t)owner;
  }
  for (uint32_t w = 0; w < world; w++) {
    const unsigned long long m = __ballot(owner == w);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&c[w], (uint32_t)__popcll(m));
  }
  __syncthreads();
  if (threadIdx.x < world) cnt[(uint64_t)threadIdx.x * nb + blockIdx.x] = c[threadIdx.x];
}

// pass B: stable scatter into the owner partitions
__global__ void __launch_bounds__(1024) k_key_scatter(InsertArgs a, uint32_t world, uint64_t nb,
                                                      const uint8_t* owner_in, const uint64_t* base,
                                                      KeyRec* out) {
  __shared__ uint32_t wc[16][MAX_WORLD];
  const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t owner = i < a.n ? owner_in[i] : KEY_NO_OWNER;
  uint32_t my_rank = 0;
  for (uint32_t w = 0; w < world; w++) {
    const unsigned long long m = __ballot(owner == w);
    if (lane == 0) wc[wv][w] = (uint32_t)__popcll(m);
    if (owner == w) my_rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  if (owner != KEY_NO_OWNER) {
    uint32_t before = 0;
    for (uint32_t k = 0; k < wv; k++) before += wc[k][owner];
    unsigned long long meta, s[5];
    entry_key(a, i, meta, s);
    KeyRec* o = out + base[(uint64_t)owner * nb + blockIdx.x] + before + my_rank;
    uint4* q = (uint4*)o;
    q[0] = make_uint4((uint32_t)meta, (uint32_t)(meta >> 32), (uint32_t)s[0], (uint32_t)(s[0] >> 32));
    q[1] = make_uint4((uint32_t)s[1], (uint32_t)(s[1] >> 32), (uint32_t)s[2], (uint32_t)(s[2] >> 32));
    q[2] = make_uint4((uint32_t)s[3], (uint32_t)(s[3] >> 32), (uint32_t)s[4], (uint32_t)(s[4] >> 32));
    q[3] = make_uint4((uint32_t)i, owner, 0u, 0u);
  }
}

// Owner side, pass 1 / pass 2 / resolve on received key records (same protocol as k_insert…)
__global__ void __launch_bounds__(256) k_keys_insert(const KeyRec* keys, uint64_t n, Slot* table,
                                                     uint64_t mask, uint32_t epoch, uint32_t* slot_id) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const KeyRec k = keys[i];
  const unsigned long long h = key_hash(k.meta, k.s);
  const unsigned long long tagw = (unsigned long long)key_tag(h) << 32;
  uint64_t j = h & mask;
  uint32_t sid = SID_FULL;
  for (uint64_t probes = 0; probes <= mask; probes++) {
    Slot* sl = table + j;
    const unsigned long long old = atomicCAS(&sl->w[0], 0ull, tagw | (uint32_t)i);
    if (old == 0ull) {
      sl->w[1] = k.meta;
      uint4* q = (uint4*)&sl->w[2];
      q[0] = make_uint4(epoch, 0u, (uint32_t)k.s[0], (uint32_t)(k.s[0] >> 32));
      q[1] = make_uint4((uint32_t)k.s[1], (uint32_t)(k.s[1] >> 32), (uint32_t)k.s[2], (uint32_t)(k.s[2] >> 32));
      q[2] = make_uint4((uint32_t)k.s[3], (uint32_t)(k.s[3] >> 32), (uint32_t)k.s[4], (uint32_t)(k.s[4] >> 32));
      sid = (uint32_t)j;
      break;
    }
    if ((old & 0xffffffff00000000ull) == tagw) {
      const uint32_t ep = (uint32_t)ld_agent(&sl->w[2]);
      if (ep != 0u && ep != epoch) {
        bool eq = sl->w[1] == k.meta;
#pragma unroll
        for (int q = 0; q < 5; q++) eq = eq && sl->w[3 + q] == k.s[q];
        if (eq) {
          sid = SID_DUP_OLD;
          break;
        }
      } else {
        sid = (uint32_t)j | SID_DEFER;
        break;
      }
    }
    j = probe_next(j, probes, mask);
  }
  slot_id[i] = sid;
}

__global__ void __launch_bounds__(256) k_keys_insert2(const KeyRec* keys, uint64_t n, Slot* table,
                                                      uint64_t mask, uint32_t epoch, uint32_t* slot_id) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t sid = slot_id[i];
  if (sid >= SID_DUP_OLD || !(sid & SID_DEFER)) return;
  sid &= ~SID_DEFER;
  const KeyRec k = keys[i];
  Slot* sl = table + sid;
  bool eq = sl->w[1] == k.meta;
#pragma unroll
  for (int q = 0; q < 5; q++) eq = eq && sl->w[3 + q] == k.s[q];
  if (eq) {
    atomicMin(&sl->w[0], ((unsigned long long)key_tag(key_hash(k.meta, k.s)) << 32) | (uint32_t)i);
  } else {
    bool created;
    sid = table_upsert(table, mask, k.meta, k.s, (uint32_t)i, epoch, true, &created);
  }
  slot_id[i] = sid;
}

__global__ void __launch_bounds__(1024) k_keys_resolve(const KeyRec* keys, uint64_t n, uint64_t nb,
                                                       const Slot* table, uint32_t epoch,
                                                       const uint32_t* slot_id, uint8_t* flags,
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
        const Slot* sl = table + sid;
        const unsigned long long w0 = sl->w[0], w1 = sl->w[1], w2 = sl->w[2];
        is_new = (uint32_t)w2 == epoch && (uint32_t)w0 == (uint32_t)i;
        canon = (uint32_t)(w1 >> 32) & 0xffffffu;
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

// Sender side: apply the returned flags to the local records, count NEW per 1024-entry block
__global__ void __launch_bounds__(256) k_apply_flags(const KeyRec* sent, const uint8_t* flags, uint64_t n_keys,
                                                     ctmr_record* records, uint32_t* blk_new) {
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const bool is_new = k < n_keys && flags[k] != 0;
  uint32_t src = 0;
  if (is_new) {
    src = sent[k].src;
    uint8_t* fl = (uint8_t*)(records + src) + 1;
    *fl = (uint8_t)(*fl | CTMR_FL_WAS_UNKNOWN);
  }
  // per-1024-entry NEW counts for the compaction: keys of one partition are in ascending log order, so the lanes
  // of a wave nearly always share one counter — one atomic per distinct counter per wave (the per-lane form spent
  // 7.6 ms per 47 M keys serialising on single words)
  unsigned long long todo = __ballot(is_new);
  const uint32_t blk = src >> 10;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t b = __shfl(blk, leader);
    const unsigned long long same = __ballot(is_new && blk == b) & todo;
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&blk_new[b], (uint32_t)__popcll(same));
    todo &= ~same;
  }
}

__global__ void __launch_bounds__(1024) k_status_hist(const ctmr_record* records, uint64_t n, uint64_t nb,
                                                      DevStats* stats) {
  __shared__ uint32_t hist[CTMR_ST__COUNT + 1];
  if (threadIdx.x <= CTMR_ST__COUNT) hist[threadIdx.x] = 0;
  __syncthreads();
  for (uint64_t blk = blockIdx.x; blk < nb; blk += gridDim.x) {
    const uint64_t i = blk * 1024 + threadIdx.x;
    uint32_t status = CTMR_ST__COUNT, longs = 0;
    if (i < n) {
      const uint32_t head = *(const uint32_t*)(records + i);
      status = head & 0xffu;
      longs = status == CTMR_ST_PASS && (head >> 16) > CTMR_MAX_SERIAL;
    }
#pragma unroll
    for (uint32_t st = 0; st < CTMR_ST__COUNT; st++) {
      const unsigned long long m = __ballot(status == st);
      if ((threadIdx.x & 63) == 0 && m) atomicAdd(&hist[st], (uint32_t)__popcll(m));
    }
    const unsigned long long ml = __ballot(longs != 0);
    if ((threadIdx.x & 63) == 0 && ml) atomicAdd(&hist[CTMR_ST__COUNT], (uint32_t)__popcll(ml));
  }
  __syncthreads();
  if (threadIdx.x < CTMR_ST__COUNT && hist[threadIdx.x])
    atomicAdd(&stats->by_status[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
  if (threadIdx.x == CTMR_ST__COUNT && hist[CTMR_ST__COUNT])
    atomicAdd(&stats->n_host, (unsigned long long)hist[CTMR_ST__COUNT]);
}

// ------------------------------------------------------------------ cross-GPU dedup, Bloom pre-filter variant
// The all-gather of per-GPU Bloom fingerprints `north_star` names, made exact.  Every rank keeps
// its OWN known-certificate table (the ordinary fused map + insert runs unchanged) and a cumulative Bloom filter of
// every key it ever found locally new.  Per round: the filters are all-gathered; a rank probes its locally-new keys
// against the other ranks' filters — a Bloom filter has no false negatives, so a key that hits no peer filter exists
// on no other rank and needs no exchange at all; a key that hits peer p's filter (a real cross-rank duplicate or a
// false positive) is sent to p, which looks it up EXACTLY in its table and answers "known here before you": found
// with an older epoch, or found in this round under a lower global order (= lower log index).  Exactly one rank — the
// lowest log index — keeps WasUnknown for each key.  The asker then clears the flag, takes the key out of its
// per-issuer count and marks its slot SHADOW (known for dedup, not counted or listed: the sets of the ranks stay
// disjoint, so Σ over ranks of SCARD / per-issuer counts is the global value, as in the owner-computes variant).
//
// Filter: blocked Bloom, one 64-bit word per key, 4 bits inside it — one 8-byte atomicOr to add, one 8-byte load per
// peer to probe.  At 16 filter bits per key the false-positive rate is ≈ 0.5 % (only extra key traffic, never a wrong
// answer).
constexpr unsigned long long SLOT_SHADOW = 1ull << 63;  // Slot.w[2]: key is counted by another rank

__host__ __device__ inline void bloom_pos(unsigned long long h, uint64_t wmask, uint64_t& word,
                                          unsigned long long& bits) {
  const unsigned long long g = mixk(h ^ 0xa0761d6478bd642full);
  word = g & wmask;
  bits = (1ull << ((g >> 40) & 63)) | (1ull << ((g >> 46) & 63)) | (1ull << ((g >> 52) & 63)) |
         (1ull << ((g >> 58) & 63));
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

// pass A: peers whose filter holds the key (bit p of hit_out[i]) + per-(peer, 1024-entry block) counts
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
      for (uint32_t p = 0; p < world; p++)
        if (p != rank && (filters[(uint64_t)p * n_words + word] & bits) == bits) hit |= 1u << p;
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
__device__ __forceinline__ uint32_t table_find(const Slot* table, uint64_t mask, unsigned long long meta,
                                               const unsigned long long s[5]) {
  const unsigned long long h = key_hash(meta, s);
  const unsigned long long tagw = (unsigned long long)key_tag(h) << 32;
  uint64_t j = h & mask;
  for (uint64_t probes = 0; probes <= mask; probes++) {
    const Slot* sl = table + j;
    const unsigned long long w0 = sl->w[0];
    if (w0 == 0ull) return SID_NONE;
    if ((w0 & 0xffffffff00000000ull) == tagw && w0 != SLOT_TOMB) {
      bool eq = sl->w[1] == meta;
#pragma unroll
      for (int k = 0; k < 5; k++) eq = eq && sl->w[3 + k] == s[k];
      if (eq) return (uint32_t)j;
    }
    j = probe_next(j, probes, mask);
  }
  return SID_NONE;
}

// Peer side: flags[k] = 1 when the key is known here before the asker's entry — since an earlier round, or since
// this round under a lower global order.
__global__ void __launch_bounds__(256) k_keys_lookup(const KeyRec* keys, uint64_t n, const Slot* table, uint64_t mask,
                                                     uint32_t round_epoch, unsigned long long order_base,
                                                     uint8_t* flags) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const KeyRec k = keys[i];
  const uint32_t sid = table_find(table, mask, k.meta, k.s);
  uint8_t f = 0;
  if (sid != SID_NONE) {
    const unsigned long long w0 = table[sid].w[0], w2 = table[sid].w[2];
    f = (uint32_t)w2 != round_epoch || order_base + (uint32_t)w0 < k.pad;
  }
  flags[i] = f;
}

// Asker side: a flagged key loses WasUnknown (once, however many peers flagged it), leaves the per-issuer count and
// its slot becomes SHADOW.
__global__ void __launch_bounds__(256) k_bloom_apply(const KeyRec* sent, const uint8_t* flags, uint64_t n_keys,
                                                     ctmr_record* records, Slot* table, uint64_t mask,
                                                     unsigned long long* issuer_counts) {
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  bool lost = false;
  uint32_t canon = 0;
  if (k < n_keys && flags[k] != 0) {
    const KeyRec kr = sent[k];
    const uint32_t old = atomicAnd((uint32_t*)(records + kr.src), ~((uint32_t)CTMR_FL_WAS_UNKNOWN << 8));
    if ((old >> 8) & CTMR_FL_WAS_UNKNOWN) {
      lost = true;
      canon = (uint32_t)(kr.meta >> 32) & 0xffffffu;
      const uint32_t sid = table_find(table, mask, kr.meta, kr.s);
      if (sid != SID_NONE) atomicOr(&table[sid].w[2], SLOT_SHADOW);
    }
  }
  unsigned long long todo = __ballot(lost);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t c = __shfl(canon, leader);
    const unsigned long long same = __ballot(lost && canon == c) & todo;
    if ((int)(threadIdx.x & 63) == leader)
      atomicAdd(&issuer_counts[c], (unsigned long long)(-(long long)__popcll(same)));
    todo &= ~same;
  }
}

// NEW count per 1024-entry block from the record flags (compaction after k_bloom_apply)
__global__ void __launch_bounds__(1024) k_count_new_flags(const ctmr_record* records, uint64_t n, uint32_t* blk_new) {
  __shared__ uint32_t c;
  if (threadIdx.x == 0) c = 0;
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  const bool is_new = i < n && (((const uint8_t*)(records + i))[1] & CTMR_FL_WAS_UNKNOWN) != 0;
  const unsigned long long m = __ballot(is_new);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&c, (uint32_t)__popcll(m));
  __syncthreads();
  if (threadIdx.x == 0) blk_new[blockIdx.x] = c;
}

}  // namespace ctmr
