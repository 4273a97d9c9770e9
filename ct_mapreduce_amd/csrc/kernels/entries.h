// kernels/entries.h — CT get-entries decode and Chain[0] → issuer match.
// gfx950 (CDNA4, wave64) only; part of kernels.h, which includes the pieces in dependency order.
#pragma once
#include "pem.h"

namespace ctmr {

// ------------------------------------------------------------------ CT get-entries decode
struct __attribute__((packed, aligned(1))) U4 { uint32_t a; };
struct DevBytes {  // arbitrary byte positions of the blob (gfx950 runs with unaligned access mode)
  const uint8_t* p;
  __device__ __forceinline__ uint32_t le32(uint64_t pos) const { return ((const U4*)(p + pos))->a; }
  __device__ __forceinline__ void le128(uint64_t pos, uint32_t out[4]) const {
    const U16 v = *(const U16*)(p + pos);
    out[0] = v.a; out[1] = v.b; out[2] = v.c; out[3] = v.d;
  }
  __device__ __forceinline__ uint32_t u8(uint64_t pos) const { return p[pos]; }
  __device__ __forceinline__ uint32_t be(uint64_t pos, int k) const {  // reads ≤ 3 bytes past pos+k: CTMR_PAYLOAD_PAD
    return __builtin_bswap32(le32(pos)) >> (32 - 8 * k);
  }
};

struct DecodeArgs {
  const uint8_t* blob;
  const uint64_t* bounds;  // 2n+1
  uint64_t n;
  uint64_t* cert_start;
  uint64_t* cert_end;
  uint8_t* entry_type;
  uint64_t* timestamp;     // may be null
  uint64_t* chain0_start;
  uint32_t* chain0_len;
  unsigned long long* counters;  // [0] x509 [1] precert [2] decode error [3] len(Chain) < 1
  const uint8_t* leaf_bad;       // null, or per entry: the precertificate entry's leaf TBSCertificate does not parse
                                 // (k_leaf_tbs_check, CTMR strict_leaf) — LogEntryFromLeaf fails, the entry is dropped
};

// strict_leaf (round 3): ct.LogEntryFromLeaf parses a precertificate entry's leaf TBSCertificate
// (x509.ParseTBSCertificate) and fails — the downloader drops the entry (cmd/ct-fetch/ct-fetch.go:452-459) — when that
// parse fails fatally; the fast profile's decode only length-checks it.  One precertificate entry per lane, the TBS pulled through
// the same per-lane LDS windows as the map (coop_fill), walk_tbs = the certificate walk without the outer wrapper and the
// signature.  Runs BEFORE the decode + match kernel, which treats a flagged entry as undecodable: its Chain[0] is never
// looked at, let alone registered.  Costs one more pass over ≈ 3 windows of every precertificate entry: opt-in.
// (CTMR_WALK_BOUNDS holds the kernel to three waves per SIMD at the price of 20 spilled registers — the inlined curve
//  arithmetic wants more.  Without the attribute: 11.9 instead of 5.3 ms per 20 M raw entries, round 6.)
// (Round 6, late) A wave takes a TILE of 256 entries, finds the precertificate entries among them (four header reads per
// lane), compacts their numbers within the tile into 256 bytes of LDS behind the windows — in log order — and walks them 64
// at a time.  With one entry per lane, as before, the X509 entries' lanes idled AND no wave was whole: every refill of the
// walk fell back to its lane-by-lane form (the cooperative fills and the subjectAltName rounds want all 64 lanes).
constexpr uint32_t LEAF_TILE = 256u;
constexpr uint32_t LEAF_LDS_BYTES = WinGeo<WIN_CH_STRICT>::LDS_BYTES + LEAF_TILE;
// a precertificate entry's TBSCertificate: [lo, lo + len) inside the blob (pre = false: not one, or the decoder's business)
__device__ __forceinline__ bool leaf_tbs_of(const DevBytes& b, const uint64_t* bounds, uint64_t i, uint64_t n, uint64_t& l0, uint64_t& lo,
                                            uint32_t& len) {
  l0 = lo = 0;
  len = 0;
  if (i >= n) return false;
  l0 = bounds[2 * i];
  const uint64_t l1 = bounds[2 * i + 1];
  // MerkleTreeLeaf: version(1) leaf_type(1) timestamp(8) entry_type(2) | issuer_key_hash(32) | TBSCertificate<1..2^24-1>
  if (l1 >= l0 + 47u && b.u8(l0 + 1) == 0u && b.be(l0 + 10, 2) == 1u) {
    len = b.be(l0 + 44, 3);
    lo = l0 + 47u;
    return len != 0u && lo + len <= l1;  // anything else is the decoder's business
  }
  return false;
}
__global__ void CTMR_WALK_BOUNDS k_leaf_tbs_check(const uint8_t* blob, const uint64_t* bounds, uint64_t n, uint64_t limit,
                                                       uint8_t* leaf_bad, uint32_t strict_spki, uint32_t strict_ext) {
  const uint64_t first = (uint64_t)blockIdx.x * LEAF_TILE;
  const uint32_t lane = threadIdx.x;
  DevBytes b{blob};
  uint8_t* const list = smem + WinGeo<WIN_CH_STRICT>::LDS_BYTES;  // tile-relative numbers of the precertificate entries, ascending
  uint32_t cnt = 0;
#pragma unroll
  for (uint32_t k = 0; k < LEAF_TILE / 64u; k++) {
    const uint64_t i = first + 64u * k + lane;
    uint64_t l0, lo;
    uint32_t len;
    const bool pre = leaf_tbs_of(b, bounds, i, n, l0, lo, len);
    if ((i < n) & !pre) leaf_bad[i] = 0;  // (a precertificate entry's byte is written by the pass that walks it)
    const unsigned long long m = __ballot(pre);
    if (pre) list[cnt + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)(64u * k + lane);
    cnt += (uint32_t)__popcll(m);
  }
  __builtin_amdgcn_wave_barrier();
  for (uint32_t p0 = 0; p0 < cnt; p0 += 64u) {  // (wave-uniform)
    const bool pre = p0 + lane < cnt;
    const uint64_t i = first + (pre ? (uint32_t)list[p0 + lane] : 0u);
    uint64_t l0, lo;
    uint32_t len;
    (void)leaf_tbs_of(b, bounds, pre ? i : n, n, l0, lo, len);  // (the bounds again: an L2 hit, instead of 16 bytes of LDS per entry)
    const WaveBuf wb = wave_buf(blob, limit, l0);  // lane 0 holds the pass's first — lowest — entry: the blob's entries ascend
    const uint32_t lrel = wave_rel(wb, lo, pre);
    // (dword aligned: the window loses at most 3 bytes in front; WinGeo::SKIP: it begins behind the TBSCertificate's header — half
    //  of what a Certificate's two headers take — which the walk reads from the sixteen octets below)
    const uint32_t w_me = lrel == REL_NONE ? REL_NONE : ((lrel + WinGeo<WIN_CH_STRICT>::SKIP / 2u) & ~3u);
    U16t hd16{0u, 0u, 0u, 0u};
    bool hd_ok = false;
    if constexpr (WinGeo<WIN_CH_STRICT>::SKIP != 0u) {
      hd_ok = pre & (lo + 16ull <= limit);
      hd16 = *(const U16t*)(blob + (hd_ok ? lo : 0ull));
    }
    coop_fill<WIN_CH_STRICT, true>(wb, w_me, lane);
    bool ok = true;
    if (pre) {
      // Round 6: the map kernel's reader — every read served by the LDS window alone, a walk that leaves it repeated with the
      // exact global-memory reader (rounds 3-5: WinReaderC, whose every read carries a global-memory path of its own — 168
      // VGPRs with 36 spills under this kernel's occupancy attribute; strict_leaf is part of the default profile now)
      WinReaderS<WIN_CH_STRICT> r{{{(const uint32_t*)blob, lo, limit, (uint32_t*)(smem + win_off<WIN_CH_STRICT>(lane)),
                         lrel == REL_NONE ? (int32_t)0x80000000 : (int32_t)(w_me - lrel), wb, lrel}}};
      r.miss = lrel == REL_NONE ? 0xffffffffu : 0u;
      r.tl_pos = 0x80000000u;
      if constexpr (WinGeo<WIN_CH_STRICT>::SKIP != 0u) {
        r.hd[0] = hd16.a; r.hd[1] = hd16.b; r.hd[2] = hd16.c; r.hd[3] = hd16.d;
        r.hd_ok = hd_ok;
      }
      Walk w;
      ok = walk_tbs(r, len, w, strict_spki != 0u, strict_ext != 0u);
      if (r.missed()) {
        GlobalReader g{(const uint32_t*)blob, lo};
        ok = walk_tbs(g, len, w, strict_spki != 0u, strict_ext != 0u);
      }
      leaf_bad[i] = (uint8_t)!ok;
    }
    __builtin_amdgcn_wave_barrier();  // (the next pass fills the windows again)
  }
}


// ct.LogEntryFromLeaf, one raw entry per lane (entry_decode.h).  Reads ≈ 5 scattered header words per entry
// (leaf header, extensions length behind the certificate, the chain headers); the certificates themselves
// are skipped by length.
constexpr uint32_t DECODE_PER_BLOCK = 2048;  // entries per workgroup: counters reach global memory once per 2048 entries
__global__ void __launch_bounds__(256) k_entry_decode(DecodeArgs a) {
  __shared__ uint32_t cnt[4];
  if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
  __syncthreads();
  uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  const uint64_t base = (uint64_t)blockIdx.x * DECODE_PER_BLOCK;
  DevBytes b{a.blob};
#pragma unroll 2
  for (uint32_t k = 0; k < DECODE_PER_BLOCK / 256; k++) {
    const uint64_t i = base + k * 256u + threadIdx.x;
    if (i >= a.n) break;
    EntryDec d;
    decode_entry(b, a.bounds[2 * i], a.bounds[2 * i + 1], a.bounds[2 * i + 2], d);
    if (a.leaf_bad && a.leaf_bad[i]) d.ok = false;
    a.cert_start[i] = d.ok ? d.cert_lo : 0ull;
    a.cert_end[i] = d.ok ? d.cert_hi : 0ull;
    a.entry_type[i] = d.ok ? (uint8_t)d.entry_type : (uint8_t)CTMR_ENTRY_INVALID;
    if (a.timestamp) a.timestamp[i] = d.ok ? d.timestamp : 0ull;
    a.chain0_start[i] = d.ok ? d.chain0_lo : 0ull;
    a.chain0_len[i] = d.ok ? d.chain0_len : 0u;
    c0 += d.ok && d.entry_type == 0;
    c1 += d.ok && d.entry_type == 1;
    c2 += !d.ok;
    c3 += d.ok && d.n_chain == 0;
  }
  // hundreds of thousands of device atomics on one cache line serialise at the memory side (measured: 12 of the
  // 15 ms of the first version of this kernel at 40 M entries): LDS first, then four atomics per workgroup
  if (c0) atomicAdd(&cnt[0], c0);
  if (c1) atomicAdd(&cnt[1], c1);
  if (c2) atomicAdd(&cnt[2], c2);
  if (c3) atomicAdd(&cnt[3], c3);
  __syncthreads();
  if (threadIdx.x < 4 && cnt[threadIdx.x]) atomicAdd(&a.counters[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
}

// Chain[0] → issuer table index: replaces, per entry, x509.ParseCertificate(Chain[0]) + NewIssuer
// (ct-fetch.go:221; storage/types.go:109-115) by a bytewise match against the issuer certificates registered so
// far (each of which went through exactly that parse once, k_issuer_ids).  Phase 1, per lane: candidate from a
// small hash table keyed by cert_quick_hash.  Phase 2, wave-cooperative: the 64 lanes stream the candidate's
// bytes (16 B per lane per step, 1 KiB per instruction) against the registered copy — every byte of Chain[0]
// is compared, so equal means identical.  Unregistered certificates are reported once per distinct hash
// (pend[] claims) for the host to register; `retry` re-examines only entries still marked unregistered.
constexpr uint32_t ISS_UNREGISTERED = 0xfffffffeu;
constexpr uint32_t HT_TWIN = 0x80000000u;  // in the low word of an issuer hash-table entry (issuer indices are < 2^24)
constexpr uint32_t PEND_SLOTS = 8192;  // distinct unknown Chain[0] hashes remembered per launch
#ifndef CTMR_MATCH_PER_STEP
#define CTMR_MATCH_PER_STEP 4
#endif
constexpr uint32_t MATCH_PER_STEP = CTMR_MATCH_PER_STEP;  // candidates whose loads are in flight together (4 and 8 both measure 9.2–9.3 ms per 40 M entries: the kernel is near the HBM rate once partial lines are counted)

struct MatchArgs {
  const uint8_t* blob;
  const uint64_t* chain0_start;
  const uint32_t* chain0_len;
  const uint8_t* entry_type;
  uint32_t* issuer_idx;
  uint64_t n;
  // issuer certificate store
  const uint8_t* idb_der;       // registered certificates, each at a 16-byte aligned offset, zero padded
  const uint64_t* idb_off;      // per issuer: offset into idb_der
  const uint32_t* idb_len;
  const unsigned long long* ht; // open addressing: (candidate hash & ~0xffffffff) | HT_TWIN? | (issuer index + 1), 0 = empty
  uint32_t ht_mask;
  uint32_t retry;
  // CTMR_CHAIN0_TRUSTED_LOG: a candidate whose table word does not carry HT_TWIN is taken on its hash and length
  // (it was parsed in full when it was registered); HT_TWIN = another registered certificate has the same candidate
  // hash, so the bytes decide
  uint32_t trusted;
  // unregistered report
  unsigned long long* pend;     // PEND_SLOTS claim words (zeroed by the host)
  uint32_t* pend_min;           // per claim word: the lowest entry index with that hash (0xffffffff-filled by the host)
  uint32_t* unreg_list;         // entry indices, one per distinct hash
  uint32_t unreg_cap;
  uint32_t report_all;          // 1 = list EVERY unregistered entry (no per-hash claim): see the note at the report
  unsigned long long* counters; // [0] entries left unregistered, [1] list entries, [2] pend overflow
};

// Chain[0] is read once, by the comparison and by nothing else: non-temporal, like the map's window fills (unaligned
// form).  A/B on one box, 40 M raw entries: decode + match 11.74 / 12.21 → 10.96 / 11.00 ms; the map kernel behind it
// 9.90 / 10.31 → 10.44 / 10.49 ms; the step 22.5 / 23.3 → 22.5 / 22.4 ms (profiles/r04/ab_raw_chain0_nontemporal.txt).
typedef uint32_t ctmr_u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
__device__ __forceinline__ U16 ld_chain16(const uint8_t* p) {
  const ctmr_u32x4_u t = __builtin_nontemporal_load((const ctmr_u32x4_u*)p);
  U16 r;
  r.a = t.x; r.b = t.y; r.c = t.z; r.d = t.w;
  return r;
}

// Round 5 asked whether the comparison's instructions or the alignment of its loads hold k_decode_match back
// (profiles/r05/ab_raw_chain0_compare.txt, raw_decode_match_pmc_*.txt; one box, alternating): neither.  Without masks — the last
// chunk anchored at the certificate's END, every lane comparing 16 whole bytes; 1 624 → 846 vector instructions in the
// listing, no SGPR spills without the occupancy attribute — decode + match take 11.19 / 11.30 ms per 40 M entries against
// 10.89 / 11.11 ms for this form; with 16-byte ALIGNED loads of the covering chunks, the neighbour's chunk by
// v_mov_b32_dpp wave_shl:1 and a v_alignbyte funnel: 12.76 ms.  The kernel is bound by what a CU keeps in flight times
// the latency of a pattern with three isolated lines per entry.
__device__ __forceinline__ bool eq16_prefix(const U16& x, const uint4& y, uint32_t rem) {  // first min(rem,16) bytes equal
  const uint32_t d[4] = {x.a ^ y.x, x.b ^ y.y, x.c ^ y.z, x.d ^ y.w};
  bool eq = true;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t have = rem > 4u * k ? rem - 4u * k : 0u;
    const uint32_t mask = have >= 4u ? 0xffffffffu : (have ? (0xffffffffu >> (8 * (4 - have))) : 0u);
    eq = eq && (d[k] & mask) == 0u;
  }
  return eq;
}

// The match of one wave's 64 entries (lane = entry): `need` lanes carry Chain[0] = blob[lo, lo+len); `result` comes in
// as what a lane that needs no match keeps (CTMR_NO_ISSUER, or the previous round's index in a retry).  Writes
// issuer_idx[i] and the unregistered report.  Must be called by all 64 lanes.
__device__ __forceinline__ void match_wave(const MatchArgs& a, bool need, uint64_t lo, uint32_t len, uint32_t result,
                                           uint64_t i, bool live, uint32_t lane) {
  DevBytes b{a.blob};
  unsigned long long qh = 0;
  uint32_t j = 0;
  if (need) {
    qh = cert_quick_hash(b, lo, len);
    j = (uint32_t)qh & a.ht_mask;
    result = ISS_UNREGISTERED;
  }
  bool searching = need;
  while (__ballot(searching)) {
    // next candidate of every searching lane: one table word carries the issuer index and the upper half of its
    // hash; length and store offset come with one more (parallel) pair of loads — no dependent load is left for
    // the cooperative phase
    uint32_t cand = 0xffffffffu;
    uint64_t db_off = 0;
    if (searching) {
      for (;;) {
        const unsigned long long v = a.ht[j];
        if (v == 0ull) {
          searching = false;
          break;
        }
        j = (j + 1u) & a.ht_mask;
        if ((v ^ qh) >> 32 == 0ull) {
          const uint32_t c = ((uint32_t)v & ~HT_TWIN) - 1u;
          const uint32_t clen = a.idb_len[c];
          db_off = a.idb_off[c];
          if (clen == len) {
            if (a.trusted && !((uint32_t)v & HT_TWIN)) {
              result = c;  // length, first and last 16 bytes identify a certificate that was parsed when it was registered
              searching = false;
              break;
            }
            cand = c;
            break;
          }
        }
      }
    }
    // cooperative bytewise verification, MATCH_PER_STEP candidates per step so that their loads are in flight together
    // (a step costs one memory latency; certificates up to 2 KiB need no inner loop)
    unsigned long long todo = __ballot(cand != 0xffffffffu);
    while (todo) {
      int src[MATCH_PER_STEP];
      bool eq[MATCH_PER_STEP];
#pragma unroll
      for (int u = 0; u < (int)MATCH_PER_STEP; u++) {
        src[u] = todo ? __ffsll((long long)todo) - 1 : -1;
        todo &= todo - 1ull;  // 0 stays 0
        eq[u] = true;
      }
#pragma unroll
      for (int u = 0; u < (int)MATCH_PER_STEP; u++) {
        if (src[u] < 0) continue;  // wave-uniform
        const uint64_t s_lo = __shfl(lo, src[u]);
        const uint32_t s_len = __shfl(len, src[u]);
        const uint8_t* db = a.idb_der + __shfl(db_off, src[u]);
        const uint32_t off0 = lane * 16u, off1 = off0 + 1024u;
        if (off0 < s_len) {
          const U16 x = ld_chain16(a.blob + s_lo + off0);  // ≤ 15 bytes past Chain[0]: CTMR_PAYLOAD_PAD
          const uint4 y = *(const uint4*)(db + off0);
          eq[u] = eq16_prefix(x, y, s_len - off0);
        }
        if (off1 < s_len) {
          const U16 x = ld_chain16(a.blob + s_lo + off1);
          const uint4 y = *(const uint4*)(db + off1);
          eq[u] = eq[u] && eq16_prefix(x, y, s_len - off1);
        }
        for (uint32_t off = off0 + 2048u; off < s_len; off += 1024u) {  // > 2 KiB: rare
          const U16 x = ld_chain16(a.blob + s_lo + off);
          const uint4 y = *(const uint4*)(db + off);
          eq[u] = eq[u] && eq16_prefix(x, y, s_len - off);
        }
      }
#pragma unroll
      for (int u = 0; u < (int)MATCH_PER_STEP; u++) {
        if (src[u] < 0) continue;
        const bool all = __ballot(!eq[u]) == 0ull;
        if ((int)lane == src[u] && all) {
          result = cand;
          searching = false;
        }
      }
    }
  }
  if (live) a.issuer_idx[i] = result;
  // report unregistered certificates, once per distinct hash
  const bool unreg = live && result == ISS_UNREGISTERED;
  const unsigned long long mu = __ballot(unreg);
  if (lane == 0 && mu) atomicAdd(&a.counters[0], (unsigned long long)__popcll(mu));
  // The per-hash claim below reports ONE certificate per candidate hash (length, first and last 16 bytes) per round
  // (the host reads pend[] / pend_min[]; unreg_list carries only what overflowed the claim table):
  // right for real chains, where distinct certificates differ there, but a batch with many distinct certificates that
  // agree in those 36 bytes (a hostile or corrupted log: the same issuer certificate damaged in hundreds of places)
  // would register one of them per round.  When registration has not converged after two rounds the host switches
  // to report_all: every unregistered entry goes on the list (up to its capacity) and the host dedups by content.
  if (a.report_all) {
    if (mu) {
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(&a.counters[1], (unsigned long long)__popcll(mu));
      base = __shfl(base, 0);
      const unsigned long long at = base + (unsigned long long)__popcll(mu & ((1ull << lane) - 1ull));
      if (unreg && at < a.unreg_cap) a.unreg_list[at] = (uint32_t)i;
    }
    return;  // of match_wave: wave-uniform (report_all is a kernel argument)
  }
  // one claim per distinct hash per wave (a cold start has every lane here)
  unsigned long long todo_u = mu;
  while (todo_u) {
    const int leader = __ffsll((long long)todo_u) - 1;
    const unsigned long long lq = __shfl(qh, leader);
    const unsigned long long same = __ballot(unreg && qh == lq) & todo_u;
    todo_u &= ~same;
    if ((int)lane != leader) continue;
    // the certificate registered for a hash is the one with the LOWEST log index that carries it — whatever order the
    // waves run in (the leader is its wave's lowest such lane): registration order, and in the trusted-log mode the
    // certificate a hash stands for, are a function of the input alone
    uint32_t k = (uint32_t)(qh >> 32) & (PEND_SLOTS - 1u);
    bool first = false, placed = false;
    for (uint32_t probes = 0; probes < 64u && !placed; probes++) {
      const unsigned long long old = atomicCAS(&a.pend[k], 0ull, qh);
      if (old == 0ull || old == qh) {
        placed = true;
        atomicMin(&a.pend_min[k], (uint32_t)i);
      }
      k = (k + 1u) & (PEND_SLOTS - 1u);
    }
    if (!placed) {
      atomicAdd(&a.counters[2], 1ull);
      first = true;  // claim table overflow: list the entry itself (the host dedups by bytes)
    }
    if (first) {
      const unsigned long long at = atomicAdd(&a.counters[1], 1ull);
      if (at < a.unreg_cap) a.unreg_list[at] = (uint32_t)i;
    }
  }
}

__global__ void __launch_bounds__(256) k_chain0_match(MatchArgs a) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < a.n;
  uint64_t lo = 0;
  uint32_t len = 0;
  bool need = false;
  uint32_t result = CTMR_NO_ISSUER;
  if (live) {
    if (a.retry) {
      result = a.issuer_idx[i];
      need = result == ISS_UNREGISTERED;
    } else {
      need = a.entry_type[i] != CTMR_ENTRY_INVALID;
    }
    if (need) {
      lo = a.chain0_start[i];
      len = a.chain0_len[i];
      need = len != 0u;
      if (!need) result = CTMR_NO_ISSUER;
    }
  }
  match_wave(a, need, lo, len, result, i, live, lane);
}

// Decode and first match round in one kernel (the default for raw input): the lane that decoded an entry's framing goes
// straight on to the match with Chain[0]'s position in registers.  The framing words lie in the lines the match (Chain[0]
// begins right behind the chain header) and the map (the leaf header sits in front of the certificate) fetch anyway; as a
// kernel of its own the decode fetched ≈ 4 lines per entry for ≈ 40 bytes of use and was bound by exactly that
// (3.7 ms per 40 M entries).  Retry rounds (a Chain[0] that had to be registered first) run k_chain0_match on the arrays
// written here.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) k_decode_match(DecodeArgs a, MatchArgs m) {
  __shared__ uint32_t cnt[4];
  if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
  __syncthreads();
  uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  const uint64_t base = (uint64_t)blockIdx.x * DECODE_PER_BLOCK;
  const uint32_t lane = threadIdx.x & 63u;
  DevBytes b{a.blob};
#pragma unroll 1
  for (uint32_t k = 0; k < DECODE_PER_BLOCK / 256; k++) {
    if (base + k * 256u >= a.n) break;  // block-uniform
    const uint64_t i = base + k * 256u + threadIdx.x;
    const bool live = i < a.n;
    EntryDec d;
    d.ok = false; d.entry_type = 0; d.chain0_lo = 0; d.chain0_len = 0; d.n_chain = 0;
    if (live) {
      decode_entry(b, a.bounds[2 * i], a.bounds[2 * i + 1], a.bounds[2 * i + 2], d);
      if (a.leaf_bad && a.leaf_bad[i]) d.ok = false;  // strict_leaf: LogEntryFromLeaf failed on the leaf's TBSCertificate
      a.cert_start[i] = d.ok ? d.cert_lo : 0ull;
      a.cert_end[i] = d.ok ? d.cert_hi : 0ull;
      a.entry_type[i] = d.ok ? (uint8_t)d.entry_type : (uint8_t)CTMR_ENTRY_INVALID;
      if (a.timestamp) a.timestamp[i] = d.ok ? d.timestamp : 0ull;
      a.chain0_start[i] = d.ok ? d.chain0_lo : 0ull;
      a.chain0_len[i] = d.ok ? d.chain0_len : 0u;
      c0 += d.ok && d.entry_type == 0;
      c1 += d.ok && d.entry_type == 1;
      c2 += !d.ok;
      c3 += d.ok && d.n_chain == 0;
    }
    const bool need = live && d.ok && d.chain0_len != 0u;
    match_wave(m, need, d.chain0_lo, d.chain0_len, CTMR_NO_ISSUER, i, live, lane);
  }
  if (c0) atomicAdd(&cnt[0], c0);
  if (c1) atomicAdd(&cnt[1], c1);
  if (c2) atomicAdd(&cnt[2], c2);
  if (c3) atomicAdd(&cnt[3], c3);
  __syncthreads();
  if (threadIdx.x < 4 && cnt[threadIdx.x]) atomicAdd(&a.counters[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
}

}  // namespace ctmr
