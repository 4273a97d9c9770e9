// kernels/meta.h — IssuerMetadata memo on the device.
// gfx950 (CDNA4, wave64) only; part of kernels.h, which includes the pieces in dependency order.
#pragma once
#include "entries.h"

namespace ctmr {

// ------------------------------------------------------------------ IssuerMetadata on device
// IssuerMetadata.Accumulate (storage/issuermetadata.go:92-138) runs for every newly unknown certificate but changes
// state only the first time an issuer meets an (expDate), a CRL distribution point or an issuer DN: its three
// per-issuer memo maps (knownExpDates :96, knownCrlDPs :113, knownIssuerDNs :97) live here as ONE device hash set of
// (kind, issuer, bytes).  k_meta_new walks the NEW list of a batch, and appends an item only for first sightings —
// the host then formats/inserts those few (addCRL :48-73, addIssuerDN :75-87, AllocateExpDateAndIssuer
// filesystemdatabase.go:189-195) instead of parsing every new certificate.
struct ByteReader {  // one unaligned dword per access (k_meta_new's TLV reads)
  const uint8_t* p;
  __device__ __forceinline__ uint32_t ld4(uint32_t pos) const { return ((const U4*)(p + pos))->a; }
};

struct MetaItem {  // 32 bytes, = ctmr_meta_item
  uint64_t entry;
  uint32_t kind, issuer_idx;
  int32_t exp_hour;
  uint32_t off, len, pad;
};
static_assert(sizeof(MetaItem) == 32, "MetaItem");

struct MetaArgs {
  const uint8_t* payload;
  const uint64_t* offsets;
  const uint64_t* ends;
  const ctmr_record* records;
  const uint32_t* canon;
  const uint2* meta_loc;
  const uint64_t* new_idx;
  uint64_t n_new;
  MetaSlot* slots;
  uint64_t mask;
  uint8_t* arena;
  uint64_t arena_cap;
  unsigned long long* counters;  // [0] arena bytes used [1] items appended [2] set/arena overflow events
  MetaItem* items;
  uint64_t items_cap;
  uint32_t epoch;  // launch number (≥ 1): slots of earlier launches are immutable and read through the caches
  uint32_t* const* hour_pages;  // knownExpDates as bitmaps: page c (META_HOUR_PAGE issuers) → one bit per (issuer, hour)
  uint32_t n_hour_pages;
  unsigned long long* refs;  // per issuer: [canon] first recorded DN, [n_refs + canon] first recorded CRL DP (as slot word w1)
  uint32_t n_refs;
  const uint32_t* ent;  // null, or the map kernel's pre-check (k_map_fused<…, META>): entries without ENT_META_UNSEEN are skipped
};

struct GlobalSrc {  // straight from the certificate in HBM: one unaligned dwordx4 load per chunk (≤ 15 bytes past the
  const uint8_t* p; //  item, which lies inside a certificate inside the payload + CTMR_PAYLOAD_PAD)
  uint32_t len;
  __device__ __forceinline__ uint4 chunk(uint32_t k) const {
    const U16 v = *(const U16*)(p + 16u * k);
    return mask_chunk(v.a, v.b, v.c, v.d, len - 16u * k);
  }
};
template <class S>
__device__ __forceinline__ bool meta_upsert(const MetaArgs& a, uint32_t kind, uint32_t issuer, uint32_t key2,
                                            const S& src, uint32_t len) {
  const uint32_t nc = (len + 15u) >> 4;
  const unsigned long long h = meta_hash(kind, issuer, key2, src, len);
  const unsigned long long w2 = ((unsigned long long)issuer << 32) | key2;
  uint64_t j = h & a.mask;
  uint64_t probes = 0;
  for (;;) {
    MetaSlot* sl = a.slots + j;
    unsigned long long w0 = ld_wave(&sl->w[0]);  // a stale 0 only sends us to the CAS, which tells the truth
    if (w0 == 0ull) {
      const unsigned long long old = atomicCAS(&sl->w[0], 0ull, h);
      if (old == 0ull) {  // claimed: copy the bytes, publish
        const unsigned long long need = (unsigned long long)nc * 16ull;
        unsigned long long at = need ? atomicAdd(&a.counters[0], need) : 0ull;
        uint32_t pk = kind;
        if (at + need > a.arena_cap) {  // arena exhausted: a dead slot (never equal to anything); always "new"
          atomicAdd(&a.counters[2], 1ull);
          pk = MK_HOST;
          at = 0;
        } else {
          unsigned long long* dst = (unsigned long long*)(a.arena + at);
          for (uint32_t k = 0; k < nc; k++) {
            const uint4 c = src.chunk(k);
            st_agent(dst + 2 * k, (unsigned long long)c.y << 32 | c.x);
            st_agent(dst + 2 * k + 1, (unsigned long long)c.w << 32 | c.z);
          }
        }
        st_agent(&sl->w[2], w2);
        st_agent(&sl->w[3], (unsigned long long)a.epoch);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long w1 = META_VALID | ((unsigned long long)pk << 60) | ((unsigned long long)len << 40) | (at >> 3);
        st_agent(&sl->w[1], w1);
        // the issuer's first Name / first CRL distribution point, for the map kernel's pre-check (meta_core.h): it reads
        // the item's bytes straight from the arena, without a hash or a slot in between.  Any published item will do.
        if ((pk == MK_DN || pk == MK_CRL) && key2 == 0u && issuer < a.n_refs) {
          unsigned long long* ref = a.refs + (pk == MK_CRL ? a.n_refs : 0u) + issuer;
          if (ld_agent(ref) == 0ull) st_agent(ref, w1);
        }
        return true;
      }
      w0 = old;
    }
    if (w0 == h) {
      unsigned long long m = ld_wave(&sl->w[1]);
      const unsigned long long ep = ld_wave(&sl->w[3]);
      const bool settled = (m & META_VALID) && ep != 0ull && ep < a.epoch;  // published by an earlier launch: immutable
      if (!settled) {
        m = ld_agent(&sl->w[1]);
        if (!(m & META_VALID)) continue;  // the claimer has not published yet: poll again (as table_upsert does)
      }
      bool eq = ((m >> 60) & 3ull) == kind && ((m >> 40) & 0xfffffull) == len &&
                (settled ? ld_wave(&sl->w[2]) : ld_agent(&sl->w[2])) == w2;
      if (eq) {
        const unsigned long long* asrc = (const unsigned long long*)(a.arena + ((m & 0xffffffffffull) << 3));
        for (uint32_t k = 0; (k < nc) & eq; k++) {
          const uint4 c = src.chunk(k);
          unsigned long long s0, s1;
          if (settled) {
            const uint4 v = *(const uint4*)(asrc + 2 * k);  // immutable: plain 16-byte load
            s0 = (unsigned long long)v.y << 32 | v.x;
            s1 = (unsigned long long)v.w << 32 | v.z;
          } else {
            s0 = ld_agent(asrc + 2 * k);
            s1 = ld_agent(asrc + 2 * k + 1);
          }
          eq = s0 == ((unsigned long long)c.y << 32 | c.x) && s1 == ((unsigned long long)c.w << 32 | c.z);
        }
      }
      if (eq) return false;
    }
    j = (j + 1) & a.mask;
    if (++probes > a.mask) break;
  }
  atomicAdd(&a.counters[2], 1ull);  // set full: report every time (the host's sets dedup)
  return true;
}

__device__ __forceinline__ void meta_emit(const MetaArgs& a, uint64_t entry, uint32_t kind, uint32_t issuer_idx,
                                          int32_t exp_hour, uint32_t off, uint32_t len) {
  const unsigned long long at = atomicAdd(&a.counters[1], 1ull);
  if (at < a.items_cap) a.items[at] = MetaItem{entry, kind, issuer_idx, exp_hour, off, len, 0u};
}

// Per-lane LDS staging: the issuer Name (≤ META_LDS_DN bytes) and the cRLDistributionPoints value (≤ META_LDS_CRL) of
// the lane's certificate are fetched with up to 12 independent 16-byte loads issued together — ONE memory latency —
// and everything after that (the DistributionPoint walk, hashing, comparing) reads LDS.  The dependent chain per
// certificate drops from ≈35 global round trips to the three memo probes.  Longer items take the global path.
// The steady state of the memo — the item was published by an EARLIER launch and sits at the home position of its
// hash — decided without the probe loop.  k_meta_new issues the home-slot loads of all of a certificate's items
// together and then the arena loads of all of them together: two memory latencies for the three lookups instead of
// the five or six of three sequential meta_upsert calls (slot, then arena, item after item).  Plain cacheable loads
// are enough: a slot an earlier launch published never changes again, and anything else (empty, claimed in this
// launch, another key) is left to meta_upsert, whose atomics tell the truth.
struct MetaHome {
  uint4 lo, hi;  // the 32-byte slot: w0 | w1 , w2 | w3
  __device__ __forceinline__ unsigned long long w(int k) const {
    const uint4& v = k < 2 ? lo : hi;
    return (k & 1) ? ((unsigned long long)v.w << 32 | v.z) : ((unsigned long long)v.y << 32 | v.x);
  }
};
__device__ __forceinline__ MetaHome meta_home_load(const MetaArgs& a, unsigned long long h, bool want) {
  MetaHome m{make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
  if (want) {
    const uint4* sl = (const uint4*)(a.slots + (h & a.mask));
    m.lo = sl[0];
    m.hi = sl[1];
  }
  return m;
}
// the home slot holds exactly (kind, issuer, key2, len) under hash h, published by an earlier launch → its arena
// offset; ~0 otherwise
__device__ __forceinline__ uint64_t meta_home_match(const MetaArgs& a, const MetaHome& m, unsigned long long h,
                                                    uint32_t kind, unsigned long long w2, uint32_t len) {
  const unsigned long long w1 = m.w(1), ep = m.w(3);
  const bool hit = m.w(0) == h && (w1 & META_VALID) && ep != 0ull && ep < a.epoch && ((w1 >> 60) & 3ull) == kind &&
                   ((w1 >> 40) & 0xfffffull) == len && m.w(2) == w2;
  return hit ? (w1 & 0xffffffffffull) << 3 : ~0ull;
}

__global__ void __launch_bounds__(256) k_meta_new(MetaArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t stage[256 * META_LDS_STRIDE];
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= a.n_new) return;
  const uint64_t i = a.new_idx[r];
  // the map kernel found everything this certificate contributes in the memo of earlier calls: nothing to add, nothing
  // to read (the memo has not changed since: only this kernel writes it)
  if (a.ent && !(a.ent[i] & ENT_META_UNSEEN)) return;
  const uint4 r0 = *(const uint4*)(a.records + i);
  const int32_t exp_hour = (int32_t)r0.y;
  const uint32_t iss = r0.z, canon = a.canon[iss];
  uint64_t lo, hi;
  cert_range(a.offsets, a.ends, i, lo, hi);
  const uint32_t L = (uint32_t)(hi - lo);
  const uint8_t* cert = a.payload + lo;
  const uint2 ml = a.meta_loc[i];
  bool host = ml.x == META_HOST || ml.y == META_HOST || ml.x == META_NONE;
  uint32_t dn_off = 0, dn_len = 0, cr_s = 0, cr_len = 0;
  if (!host) {
    dn_off = ml.x & 0xffffu;
    dn_len = ml.x >> 16;
    host = dn_len > META_MAX_BYTES || dn_off + dn_len > L;
    if (ml.y != META_NONE) {
      cr_s = ml.y & 0xffffu;
      cr_len = ml.y >> 16;
      host = host || cr_s + cr_len > L;
    }
  }
  // ---- stage: every load of this lane is in flight before the first one is needed
  uint8_t* my = stage + threadIdx.x * META_LDS_STRIDE;
  const bool dn_lds = !host && dn_len <= META_LDS_DN, cr_lds = !host && cr_len != 0u && cr_len <= META_LDS_CRL;
  {
    U16 d[META_LDS_DN / 16], c[META_LDS_CRL / 16];
#pragma unroll
    for (uint32_t k = 0; k < META_LDS_DN / 16; k++)
      if (dn_lds && 16u * k < dn_len) d[k] = *(const U16*)(cert + dn_off + 16u * k);
#pragma unroll
    for (uint32_t k = 0; k < META_LDS_CRL / 16; k++)
      if (cr_lds && 16u * k < cr_len) c[k] = *(const U16*)(cert + cr_s + 16u * k);
#pragma unroll
    for (uint32_t k = 0; k < META_LDS_DN / 16; k++)
      if (dn_lds && 16u * k < dn_len) *(uint4*)(my + 16u * k) = make_uint4(d[k].a, d[k].b, d[k].c, d[k].d);
#pragma unroll
    for (uint32_t k = 0; k < META_LDS_CRL / 16; k++)
      if (cr_lds && 16u * k < cr_len) *(uint4*)(my + META_LDS_DN + 16u * k) = make_uint4(c[k].a, c[k].b, c[k].c, c[k].d);
  }
  // ---- what this certificate contributes: (issuer, expDate), its CRL distribution point URIs, its issuer Name.
  // knownCrlDPs (:111-127): CRLDistributionPoints ::= SEQUENCE OF DistributionPoint { [0] { [0] GeneralNames { [6] URI }}}
  // One validating pass collects the URI ranges (a malformed value yields NO URIs, as the oracle defines; more than
  // META_MAX_URIS → host).
  uint32_t uo[META_MAX_URIS], ul[META_MAX_URIS], nu = 0;
#pragma unroll
  for (uint32_t k = 0; k < META_MAX_URIS; k++) uo[k] = ul[k] = 0;
  const uint32_t* cw = (const uint32_t*)(my + META_LDS_DN);
  bool crl_ok = false;
  if (!host && cr_len != 0u) {
    const uint32_t e = cr_s + cr_len;
    bool ok;
    if (cr_lds) {
      LdsTlvReader g{cw, cr_s};
      ok = walk_crl_dps(g, e, cr_s, e, uo, ul, nu, host);  // L = e: reads clamp to the staged value, not to the certificate
    } else {
      ByteReader g{cert};
      ok = walk_crl_dps(g, L, cr_s, e, uo, ul, nu, host);
    }
    crl_ok = ok && !host;
  }
  // ---- fast path: home slots of (expDate | first URI | Name) loaded together, then their arena bytes together
  const bool f_crl = crl_ok && cr_lds && nu >= 1u && ul[0] <= META_LDS_CRL;
  const bool f_dn = dn_lds;
  const LdsSrc s_crl{cw, f_crl ? uo[0] - cr_s : 0u, f_crl ? ul[0] : 0u};
  const LdsSrc s_dn{(const uint32_t*)my, 0u, f_dn ? dn_len : 0u};
  const unsigned long long w2_it = (unsigned long long)canon << 32;
  uint4 cc[META_LDS_CRL / 16], cd[META_LDS_DN / 16];  // the items' chunks, extracted from LDS once
  const uint32_t nc_crl = f_crl ? (ul[0] + 15u) >> 4 : 0u, nc_dn = f_dn ? (dn_len + 15u) >> 4 : 0u;
  MetaHashState st_crl = meta_hash_begin(MK_CRL, canon, 0, s_crl.len), st_dn = meta_hash_begin(MK_DN, canon, 0, s_dn.len);
#pragma unroll
  for (uint32_t k = 0; k < META_LDS_CRL / 16; k++) {
    cc[k] = make_uint4(0, 0, 0, 0);
    if (k < nc_crl) {
      cc[k] = s_crl.chunk(k);
      meta_hash_chunk(st_crl, cc[k]);
    }
  }
#pragma unroll
  for (uint32_t k = 0; k < META_LDS_DN / 16; k++) {
    cd[k] = make_uint4(0, 0, 0, 0);
    if (k < nc_dn) {
      cd[k] = s_dn.chunk(k);
      meta_hash_chunk(st_dn, cd[k]);
    }
  }
  const unsigned long long h_crl = meta_hash_end(st_crl), h_dn = meta_hash_end(st_dn);
  // knownExpDates: the (issuer, hour) bit
  const bool exp_bit = (uint32_t)exp_hour < META_HOUR_BITS && canon / META_HOUR_PAGE < a.n_hour_pages;
  uint32_t* exp_word = nullptr;
  uint32_t exp_have = 0;
  if (exp_bit) {
    exp_word = a.hour_pages[canon / META_HOUR_PAGE] + (uint64_t)(canon % META_HOUR_PAGE) * (META_HOUR_BITS / 32) +
               ((uint32_t)exp_hour >> 5);
    exp_have = __hip_atomic_load(exp_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);  // cacheable; a stale 0 → atomicOr
  }
  const MetaHome m_crl = meta_home_load(a, h_crl, f_crl), m_dn = meta_home_load(a, h_dn, f_dn);
  bool seen_exp = exp_bit && ((exp_have >> ((uint32_t)exp_hour & 31u)) & 1u);
  const uint64_t at_crl = f_crl ? meta_home_match(a, m_crl, h_crl, MK_CRL, w2_it, ul[0]) : ~0ull;
  const uint64_t at_dn = f_dn ? meta_home_match(a, m_dn, h_dn, MK_DN, w2_it, dn_len) : ~0ull;
  bool seen_crl = at_crl != ~0ull, seen_dn = at_dn != ~0ull;
  {
    uint4 ac[META_LDS_CRL / 16], ad[META_LDS_DN / 16];
    const uint32_t lc = seen_crl ? nc_crl : 0u, ld = seen_dn ? nc_dn : 0u;
#pragma unroll
    for (uint32_t k = 0; k < META_LDS_CRL / 16; k++)
      ac[k] = k < lc ? *(const uint4*)(a.arena + at_crl + 16u * k) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (uint32_t k = 0; k < META_LDS_DN / 16; k++)
      ad[k] = k < ld ? *(const uint4*)(a.arena + at_dn + 16u * k) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (uint32_t k = 0; k < META_LDS_CRL / 16; k++)
      if (k < lc) seen_crl = seen_crl && cc[k].x == ac[k].x && cc[k].y == ac[k].y && cc[k].z == ac[k].z && cc[k].w == ac[k].w;
#pragma unroll
    for (uint32_t k = 0; k < META_LDS_DN / 16; k++)
      if (k < ld) seen_dn = seen_dn && cd[k].x == ad[k].x && cd[k].y == ad[k].y && cd[k].z == ad[k].z && cd[k].w == ad[k].w;
  }
  // ---- everything the fast path did not settle goes through the exact upsert
  // knownExpDates → seenExpDateBefore (issuermetadata.go:96-108): no bytes
  if (!seen_exp) {
    bool first;
    if (exp_bit) {
      const uint32_t bit = 1u << ((uint32_t)exp_hour & 31u);
      first = !(atomicOr(exp_word, bit) & bit);
    } else {
      first = meta_upsert(a, MK_EXPDATE, canon, (uint32_t)exp_hour, GlobalSrc{cert, 0}, 0);
    }
    if (first) meta_emit(a, i, MK_EXPDATE, iss, exp_hour, 0, 0);
  }
  if (crl_ok) {
#pragma unroll
    for (uint32_t k = 0; k < META_MAX_URIS; k++) {
      if (k >= nu || (k == 0 && seen_crl)) continue;
      const bool first = cr_lds ? meta_upsert(a, MK_CRL, canon, 0, LdsSrc{cw, uo[k] - cr_s, ul[k]}, ul[k])
                                : meta_upsert(a, MK_CRL, canon, 0, GlobalSrc{cert + uo[k], ul[k]}, ul[k]);
      if (first) meta_emit(a, i, MK_CRL, iss, exp_hour, uo[k], ul[k]);
    }
  }
  // knownIssuerDNs (:97,:130-135): keyed by the Name's DER bytes (Issuer.String() is a function of them)
  if (!host && !seen_dn) {
    const bool first = dn_lds ? meta_upsert(a, MK_DN, canon, 0, LdsSrc{(const uint32_t*)my, 0u, dn_len}, dn_len)
                              : meta_upsert(a, MK_DN, canon, 0, GlobalSrc{cert + dn_off, dn_len}, dn_len);
    if (first) meta_emit(a, i, MK_DN, iss, exp_hour, dn_off, dn_len);
  }
  if (host) meta_emit(a, i, MK_HOST, iss, exp_hour, 0, L);
}

}  // namespace ctmr
