// kernels/meta_core.h — the read-only side of the IssuerMetadata memo: slot layout, item hash, chunk
// sources, the cRLDistributionPoints walk.  Shared by k_meta_new (meta.h) and by the map kernel's pre-check
// (reduce.h, k_map_fused<…, META>): a certificate whose (issuer, expDate), issuer Name and CRL distribution point the
// memo of EARLIER calls already holds contributes nothing, and k_meta_new need not read it again.
// gfx950 (CDNA4, wave64) only; part of kernels.h, which includes the pieces in dependency order.
#pragma once
#include "sha256.h"

namespace ctmr {

// Set semantics are exact: a slot is claimed by CAS on the 64-bit hash, its bytes are copied into an arena and
// published (write-through payload, drained, then the VALID word — the table_upsert recipe); equal hash is
// followed by a full comparison, so a hash collision only costs a probe.
struct MetaSlot {
  unsigned long long w[4];  // w0 hash (claim, never 0) | w1 VALID(63) kind(61..60) len(59..40) arena_off/8(39..0)
};                          // w2 issuer << 32 | key2 | w3 launch number that created the slot
constexpr unsigned long long META_VALID = 1ull << 63;
constexpr uint32_t MK_EXPDATE = 0, MK_CRL = 1, MK_DN = 2, MK_HOST = 3;  // item kinds; MK_HOST = parse this one on the host
constexpr uint32_t META_MAX_BYTES = 4096;
constexpr uint32_t META_MAX_URIS = 4;  // CRL distribution point URIs per certificate on the device path; more → host


// knownExpDates (issuermetadata.go:96-108) is a set of (issuer, expDate hour): small dense integers.  One bit per pair —
// META_HOUR_BITS hours (1970 … 2089) per canonical issuer, pages of META_HOUR_PAGE issuers allocated as issuers are
// registered — decides "seen before" with one cached load, and a first sighting with one atomicOr whose return value
// names the single lane that reports it.  (As entries of the hash set below, the ≈ 2 000 hours of every issuer were 99 %
// of its population: they pushed DN/CRL items off their home slots and, a few lanes per wave at a time, kept nearly
// every wave in the probe loop.)  Hours outside the bitmap's range take the hash-set path.
constexpr uint32_t META_HOUR_BITS = 1u << 20, META_HOUR_PAGE = 64;

// 16-byte chunks of an item, bytes past its end zeroed.  Items are hashed and compared in these chunks: the first
// version of this kernel used dwords and was bound by L2 REQUESTS (1.46 G for 18.8 M new certificates, 76 % of its
// L1 accesses missing — profiles/r01/s4/pmc_meta_20m_dword_version.txt).
__device__ __forceinline__ uint4 mask_chunk(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t rem) {
  if (rem >= 16u) return make_uint4(w0, w1, w2, w3);  // only an item's last chunk has bytes to clear
  uint32_t w[4] = {w0, w1, w2, w3};
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint32_t have = rem > 4u * q ? rem - 4u * q : 0u;
    w[q] = have >= 4u ? w[q] : (have ? (w[q] & (0xffffffffu >> (8u * (4u - have)))) : 0u);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

struct LdsSrc {  // from this lane's staging area in LDS, at any byte offset (5 dwords, 4 alignbytes)
  const uint32_t* w;  // dword-aligned lane area
  uint32_t off;       // byte offset of the item inside it
  uint32_t len;
  __device__ __forceinline__ uint4 chunk(uint32_t k) const {
    const uint32_t at = off + 16u * k, i = at >> 2, sh = at & 3u;
    const uint32_t d0 = w[i], d1 = w[i + 1], d2 = w[i + 2], d3 = w[i + 3], d4 = w[i + 4];
    return mask_chunk(__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
                      __builtin_amdgcn_alignbyte(d3, d2, sh), __builtin_amdgcn_alignbyte(d4, d3, sh), len - 16u * k);
  }
};


// Plain (cacheable) load the compiler may not merge or hoist: wavefront-scope atomic.  Used for memo slots of EARLIER
// launches, which are immutable — the steady state, where the same few hundred DN/CRL slots are read by every new
// certificate and should come out of L1/L2 instead of device-coherent loads.
__device__ __forceinline__ unsigned long long ld_wave(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// true = first sighting of (kind, issuer, key2, bytes); the bytes come through a chunk source
// Item hash: add-rotate-xor over the 16-byte chunks (12 full-rate VALU operations per chunk), one multiply-mix at each
// end.  The first version ran two mixk() — four 64-bit multiplies, quarter-rate on CDNA — per chunk: ≈ 2 700 VALU
// instructions per wave of certificates, half of the kernel's time (pmc_meta, session 5).  Equal hashes are always
// followed by a full comparison, so the hash only has to spread.
struct MetaHashState {
  uint32_t a, b;
};
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return __builtin_amdgcn_alignbit(x, x, 32 - r); }
__device__ __forceinline__ MetaHashState meta_hash_begin(uint32_t kind, uint32_t issuer, uint32_t key2, uint32_t len) {
  const unsigned long long h =
      mixk((((unsigned long long)issuer << 32 | key2) + 0x9e3779b97f4a7c15ull * (kind + 1u)) ^ ((unsigned long long)len << 24));
  return MetaHashState{(uint32_t)h, (uint32_t)(h >> 32)};
}
__device__ __forceinline__ void meta_hash_chunk(MetaHashState& s, const uint4& c) {
  s.a = rotl32(s.a, 5) ^ c.x;  s.a += s.b;
  s.b = rotl32(s.b, 11) ^ c.y; s.b += s.a;
  s.a = rotl32(s.a, 7) ^ c.z;  s.a += s.b;
  s.b = rotl32(s.b, 13) ^ c.w; s.b += s.a;
}
__device__ __forceinline__ unsigned long long meta_hash_end(const MetaHashState& s) {
  const unsigned long long h = mixk((unsigned long long)s.b << 32 | s.a);
  return h ? h : 1ull;
}
template <class S>
__device__ __forceinline__ unsigned long long meta_hash(uint32_t kind, uint32_t issuer, uint32_t key2, const S& src,
                                                        uint32_t len) {
  const uint32_t nc = (len + 15u) >> 4;
  MetaHashState st = meta_hash_begin(kind, issuer, key2, len);
  for (uint32_t k = 0; k < nc; k++) meta_hash_chunk(st, src.chunk(k));
  return meta_hash_end(st);
}


constexpr uint32_t META_LDS_DN = 128, META_LDS_CRL = 64, META_LDS_STRIDE = META_LDS_DN + META_LDS_CRL + 16;

struct LdsTlvReader {  // rd_hdr over the staged cRLDistributionPoints value: positions are certificate offsets
  const uint32_t* w;   // lane area (dwords) of the value
  uint32_t s;          // certificate offset of its first byte
  __device__ __forceinline__ uint32_t ld4(uint32_t pos) const {
    const uint32_t rel = pos - s;  // callers stay within [s, e + 3]; the area has 16 bytes of slack
    const uint32_t i = rel >> 2;
    return __builtin_amdgcn_alignbyte(w[i + 1], w[i], rel & 3u);
  }
};

// Where the URIs of a cRLDistributionPoints value [cv, e) lie: der_walk.h crl_dps<COLLECT> — Go's positional struct
// rules, the FullName elements with tag NUMBER 6 (round 5; rounds 1–4 took every [0] { [0] { [6] } } in any order).
// Returns false when the value is malformed (it yields NO URIs then, as the oracle defines); `host` is set when a URI is
// too long or there are too many.
template <class R>
__device__ __forceinline__ bool walk_crl_dps(const R& g, uint32_t L, uint32_t cv, uint32_t e, uint32_t (&uo)[META_MAX_URIS],
                                             uint32_t (&ul)[META_MAX_URIS], uint32_t& nu, bool& host) {
  bool ok = true;
  uint32_t nf = 0u;
  crl_dps<true, false, false, META_MAX_URIS>(g, L, cv, e, ok, uo, ul, nu, nf, false);
  host = host | (nu > META_MAX_URIS);
#pragma unroll
  for (uint32_t k = 0; k < META_MAX_URIS; k++) host = host | ((k < nu) & (ul[k] > META_MAX_BYTES));
  return ok;
}


// ------------------------------------------------------------------ the map kernel's pre-check
// Read-only view of the memo as EARLIER ctmr_meta_new calls left it (nothing writes it while a map kernel runs).
struct MetaCheck {
  const MetaSlot* slots;
  uint64_t mask;
  const uint8_t* arena;
  uint32_t* const* hour_pages;
  uint32_t n_hour_pages;
  const unsigned long long* refs;  // [canon] the issuer's first recorded Name, [n_refs + canon] its first CRL DP: slot word w1
  uint32_t n_refs;
  uint32_t enabled;
};

// The item is the one `ref` names (an issuer nearly always shows one Name and one CRL distribution point): its bytes
// come straight from the arena — one round trip, MAXC independent 16-byte loads — and are compared with the window's.
template <uint32_t MAXC, class S>
__device__ __forceinline__ bool meta_check_ref(const MetaCheck& m, unsigned long long ref, uint32_t kind, const S& src, uint32_t len) {
  if (!(ref & META_VALID) || ((ref >> 60) & 3ull) != kind || ((ref >> 40) & 0xfffffull) != len || len > 16u * MAXC) return false;
  const uint8_t* at = m.arena + ((ref & 0xffffffffffull) << 3);
  const uint32_t nc = (len + 15u) >> 4;
  uint4 v[MAXC];
#pragma unroll
  for (uint32_t k = 0; k < MAXC; k++) v[k] = k < nc ? *(const uint4*)(at + 16u * k) : make_uint4(0, 0, 0, 0);
  bool eq = true;
#pragma unroll
  for (uint32_t k = 0; k < MAXC; k++)
    if (k < nc) {
      const uint4 c = src.chunk(k);
      eq = eq && v[k].x == c.x && v[k].y == c.y && v[k].z == c.z && v[k].w == c.w;
    }
  return eq;
}

// true = the memo holds exactly (kind, issuer, bytes) at the home position of its hash.  Anything else — another
// position, an empty or unpublished slot, a longer item — answers false: "not known to be seen", and k_meta_new decides.
template <class S>
__device__ __forceinline__ bool meta_check_item(const MetaCheck& m, uint32_t kind, uint32_t issuer, const S& src, uint32_t len) {
  const uint32_t nc = (len + 15u) >> 4;
  const unsigned long long h = meta_hash(kind, issuer, 0u, src, len);
  const uint4* sl = (const uint4*)(m.slots + (h & m.mask));
  const uint4 lo = sl[0], hi = sl[1];
  const unsigned long long w0 = (unsigned long long)lo.y << 32 | lo.x, w1 = (unsigned long long)lo.w << 32 | lo.z,
                           w2 = (unsigned long long)hi.y << 32 | hi.x, w3 = (unsigned long long)hi.w << 32 | hi.z;
  bool eq = w0 == h && (w1 & META_VALID) && w3 != 0ull && ((w1 >> 60) & 3ull) == kind && ((w1 >> 40) & 0xfffffull) == len &&
            w2 == ((unsigned long long)issuer << 32);
  if (eq) {
    const uint8_t* at = m.arena + ((w1 & 0xffffffffffull) << 3);
    for (uint32_t k = 0; (k < nc) & eq; k++) {
      const uint4 v = *(const uint4*)(at + 16u * k);
      const uint4 c = src.chunk(k);
      eq = v.x == c.x && v.y == c.y && v.z == c.z && v.w == c.w;
    }
  }
  return eq;
}

// Hook of the map kernel's window reader (readers.h): the issuer Name is looked up while the FRONT window still holds
// it (der_walk.h calls note_issuer right behind the Name) — by the end of the walk the window holds the extensions.
struct MetaHook {
  MetaCheck mc;
  uint32_t canon;
  unsigned long long dn_ref;  // refs[canon], loaded before the window fill
  bool dn_seen;
  __device__ __forceinline__ void issue() {}
  __device__ __forceinline__ void resolve() {}
  // win = this lane's window words, rel = the Name's offset inside the window (may be outside: then unseen)
  __device__ __forceinline__ void note_issuer(const uint32_t* win, uint32_t rel, uint32_t len, uint32_t wbytes) {
    dn_seen = false;
    if (mc.enabled && len != 0u && len <= META_MAX_BYTES && rel <= wbytes && len <= wbytes - rel) {
      const LdsSrc src{win, rel, len};
      // (the recorded Name's bytes fetched before the window fill and held in registers instead: 25.4 against 25.3–25.8 ms,
      //  not kept)
      dn_seen = meta_check_ref<8>(mc, dn_ref, MK_DN, src, len);
      if (!dn_seen) dn_seen = meta_check_item(mc, MK_DN, canon, src, len);  // another Name of this issuer, or a long one
    }
  }
};

constexpr uint32_t ENT_META_UNSEEN = 0x40u;  // ent[] bit 6 (reduce.h): the certificate may bring a first sighting — k_meta_new looks

// End of the walk, extension window resident: the CRL distribution point (parsed from LDS) and the (issuer, expDate hour)
// bit, in two steps so that their one round trip — the hour word and the arena bytes, fetched together — overlaps the
// known-certificate table's probe: meta_tail_issue() parses and issues the loads, meta_tail_finish() compares.
// meta_crl = Walk.meta_crl (offset | len << 16 of the extension value, certificate offsets); grel = the window's start;
// crl_ref = refs[n_refs + canon] and hour_row = this issuer's bitmap row, both loaded before the window fill.
struct MetaTail {
  uint4 v[4];        // the referenced CRL distribution point's bytes (arena)
  uint32_t hword;    // the bitmap word of the hour
  uint32_t uri_rel, uri_len;  // the certificate's one URI inside the window (uri_len 0 = none to compare)
  uint32_t state;    // 0 = unseen, 1 = CRL side seen without a comparison, 2 = compare v[] with the URI, 3 = look the URI up by hash
  bool hour_ok;
};
__device__ __forceinline__ MetaTail meta_tail_issue(const MetaCheck& m, unsigned long long crl_ref, const uint32_t* hour_row,
                                                    int32_t exp_hour, uint32_t meta_crl, const uint32_t* win, int32_t grel,
                                                    uint32_t wbytes) {
  MetaTail t;
#pragma unroll
  for (int k = 0; k < 4; k++) t.v[k] = make_uint4(0, 0, 0, 0);
  t.hword = 0; t.uri_rel = 0; t.uri_len = 0; t.state = 0;
  t.hour_ok = hour_row != nullptr && (uint32_t)exp_hour < META_HOUR_BITS;
  if (!t.hour_ok) return t;
  t.hword = hour_row[(uint32_t)exp_hour >> 5];
  if (meta_crl == 0u) {
    t.state = 1;                 // META_NONE: no cRLDistributionPoints extension, nothing to contribute
  } else if (meta_crl != 0xffffffffu) {  // (META_HOST stays unseen)
    const uint32_t cr_s = meta_crl & 0xffffu, cr_len = meta_crl >> 16;
    const uint32_t rel = cr_s - (uint32_t)grel;
    if (rel <= wbytes && cr_len + 8u <= wbytes - rel) {  // else not (wholly) in the window: k_meta_new reads it
      LdsTlvReader g{win, (uint32_t)grel};  // positions are certificate offsets, as walk_crl_dps expects
      uint32_t uo[META_MAX_URIS], ul[META_MAX_URIS], nu = 0;
#pragma unroll
      for (uint32_t k = 0; k < META_MAX_URIS; k++) uo[k] = ul[k] = 0;
      bool ok = true, host = false;
      const uint32_t e = cr_s + cr_len;
      ok = walk_crl_dps(g, e, cr_s, e, uo, ul, nu, host);  // L = e: reads clamp to the value, not to the certificate
      if (host) t.state = 0;                 // k_meta_new hands this certificate to the host
      else if (!ok || nu == 0u) t.state = 1; // a malformed value yields NO URIs (k_meta_new, oracle): nothing to contribute
      else if (nu == 1u) {
        t.uri_rel = uo[0] - (uint32_t)grel;
        t.uri_len = ul[0];
        const bool by_ref = (crl_ref & META_VALID) && ((crl_ref >> 60) & 3ull) == MK_CRL &&
                            ((crl_ref >> 40) & 0xfffffull) == ul[0] && ul[0] <= 64u;
        t.state = by_ref ? 2u : 3u;
        if (by_ref) {
          const uint8_t* at = m.arena + ((crl_ref & 0xffffffffffull) << 3);
          const uint32_t nc = (ul[0] + 15u) >> 4;
#pragma unroll
          for (uint32_t k = 0; k < 4; k++)
            if (k < nc) t.v[k] = *(const uint4*)(at + 16u * k);
        }
      }
    }
  }
  return t;
}
__device__ __forceinline__ bool meta_tail_finish(const MetaCheck& m, uint32_t canon, const MetaTail& t, int32_t exp_hour,
                                                 const uint32_t* win) {
  if (!t.hour_ok || t.state == 0u || !((t.hword >> ((uint32_t)exp_hour & 31u)) & 1u)) return false;
  if (t.state == 1u) return true;
  const LdsSrc src{win, t.uri_rel, t.uri_len};
  bool eq = t.state == 2u;
  if (eq) {
    const uint32_t nc = (t.uri_len + 15u) >> 4;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
      if (k < nc) {
        const uint4 c = src.chunk(k);
        eq = eq && t.v[k].x == c.x && t.v[k].y == c.y && t.v[k].z == c.z && t.v[k].w == c.w;
      }
  }
  if (!eq) eq = meta_check_item(m, MK_CRL, canon, src, t.uri_len);  // another distribution point of this issuer
  return eq;
}

}  // namespace ctmr
