// kernels/map.h — the map: one certificate per lane — TBSCertificate walk, the three filters, the 32-byte record.
// gfx950 (CDNA4, wave64) only; part of kernels.h, which includes the pieces in dependency order.
#pragma once
#include "sha256.h"

namespace ctmr {

// ------------------------------------------------------------------ the map
struct MapArgs {
  const uint8_t* payload;
  const uint64_t* offsets;    // packed batch: n+1 offsets; entry view (ends != null): n range starts
  const uint64_t* ends;       // null = packed batch; else certificate i is [offsets[i], ends[i]) (ctmr_entry_view)
  uint64_t limit;             // entry view: readable bytes of payload (blob bytes + CTMR_PAYLOAD_PAD)
  const uint32_t* issuer_idx;
  const uint8_t* entry_type;  // may be null
  ctmr_record* records;
  const uint8_t* issuer_valid;
  const FilterDev* filt;
  uint64_t n;
  uint32_t n_issuers;
  uint32_t certs_per_tile;  // CTMR_SWEEP builds only (k_map_tile)
  uint32_t lds_bytes;       // dynamic LDS size of the launch (k_map_tile)
  uint2* meta_loc;          // null, or per entry (Walk.meta_issuer, Walk.meta_crl) for k_meta_new (config.collect_meta)
  unsigned long long* keypos;  // strict_spki: per entry, written ONLY for an entry whose EC key still owes the curve equation
                               // (keypos_pack; spki_key.h "where the curve equation runs") — k_ec_resolve reads it
  uint32_t optimistic_new;  // 1: PASS records leave the map with CTMR_FL_WAS_UNKNOWN already set — the
                            // reduce only CLEARS it for the (rare) duplicates, so the common case costs
                            // no second scattered write into the record array
};

// Byte range of certificate i and the readable size of the payload, for both input forms.
__device__ __forceinline__ void cert_range(const uint64_t* offsets, const uint64_t* ends, uint64_t i, uint64_t& lo,
                                           uint64_t& hi) {
  lo = offsets[i];
  hi = ends ? ends[i] : offsets[i + 1];
  if (hi < lo) hi = lo;
}
__device__ __forceinline__ uint64_t map_limit(const MapArgs& a) {
  return a.ends ? a.limit : a.offsets[a.n] + CTMR_PAYLOAD_PAD;
}

// An entry whose EC key owes the curve equation ("key pending"): where the point lies, for k_ec_resolve.
//   bits 0..31 certificate offset of X, 32..34 curve (1..5), 35..37 the BIT STRING's pad count,
//   38 the entry's key left for its owner (XM_OWNER) as record 39..44 of its wave's staging group
// Inside a kernel the record's flag byte carries bit 7 for such an entry; it never leaves the library.
constexpr uint32_t FL_KEY_PENDING = 0x80u;
__device__ __forceinline__ unsigned long long keypos_pack(uint32_t pos, uint32_t curve, uint32_t shift) {
  return (unsigned long long)pos | ((unsigned long long)(curve & 7u) << 32) | ((unsigned long long)(shift & 7u) << 35);
}

// Everything after the bytes are addressable: walk, filters, record.
// The per-entry inputs besides the certificate: issuer index, entry type, and whether that issuer's certificate parsed.
// The fused kernel loads them BEFORE it waits for the certificate's bytes (k_map_fused), so that the two dependent
// loads (issuer_idx → issuer_valid) are not two exposed round trips behind the walk.
struct EntryIn {
  uint32_t iss, et;
  bool iss_in_range, iss_valid;
};
__device__ __forceinline__ EntryIn load_entry_in(const MapArgs& a, uint64_t idx) {
  EntryIn e;
  e.iss = a.issuer_idx[idx];
  e.et = a.entry_type ? a.entry_type[idx] : 0u;
  e.iss_in_range = e.iss != CTMR_NO_ISSUER && e.iss < a.n_issuers;
  e.iss_valid = e.iss_in_range && a.issuer_valid[e.iss] != 0;
  return e;
}

// STRICT: the instantiation for engines with ctmr_set_strict_strings (the Names' character sets, inside the walk).
template <bool STRICT = false, class R>
__device__ __forceinline__ void map_one(R& r, uint64_t len64, uint64_t idx, const MapArgs& a, const EntryIn& in, uint4& o0,
                                        uint4& o1, uint2* meta_words = nullptr, unsigned long long* keypos_out = nullptr) {
  Walk w;
  const uint32_t L = len64 <= 0x7fffffffull ? (uint32_t)len64 : 0x80000000u;
  const FilterDev* f = a.filt;
  const FilterView fv{f->n_pieces, f->piece_len, f->piece_word, f->words};
  // L > 2^31-1 is rejected inside, without divergence.  EC_DEFER: an EC key's curve equation is k_ec_resolve's business
  // strict_strings: only a precertificate can lose its place over a string finding (an X509 entry is kept, below)
  const uint32_t iss = in.iss, et = in.et;
  // (STRICT is instantiated for engines with strict_strings OR strict_extensions: the filter says which)
  bool ok = walk_cert<R, false, false, true, STRICT>(r, L, w, f->active != 0u, fv, (f->strict_spki != 0u) & (a.keypos != nullptr),
                                                     STRICT & (f->strict_strings != 0u) & (et == 1u),
                                                     (STRICT & (f->strict_ext != 0u)) ? (et == 1u ? WALK_EXT_ON | WALK_EXT_NF : WALK_EXT_ON) : 0u);
  // An X509 entry's certificate was parsed by ct.LogEntryFromLeaf, which keeps it unless the error is fatal
  // (ct-fetch.go:452-459); a precertificate is parsed in insertCTWorker and dropped on ANY error, CT-go's
  // x509.NonFatalErrors included (:202-209).  Fields of a dropped certificate are not reported.
  ok = ok & !((et == 1u) & (w.nonfatal != 0u));
  uint32_t status;
  if (et == CTMR_ENTRY_INVALID) {
    status = CTMR_ST_ENTRY_DECODE_ERROR;  // never reached entryChan (ct-fetch.go:452-459)
  } else if (!ok) {
    status = CTMR_ST_PARSE_ERROR;
  } else if (w.bc_valid && w.is_ca) {
    status = CTMR_ST_FILTERED_CA;
  } else if (w.not_after < f->now && !f->log_expired) {
    status = CTMR_ST_FILTERED_EXPIRED;
  } else if (!w.cn_match) {
    status = CTMR_ST_FILTERED_CN;
  } else if (!in.iss_in_range) {
    status = CTMR_ST_NO_ISSUER;
  } else if (!in.iss_valid) {
    status = CTMR_ST_ISSUER_PARSE_ERROR;
  } else {
    status = CTMR_ST_PASS;
  }
  uint32_t flags = et == 1u ? CTMR_FL_PRECERT : 0u;
  if (a.optimistic_new && status == CTMR_ST_PASS) flags |= CTMR_FL_WAS_UNKNOWN;
  // key pending — whatever the status: a certificate whose point is off its curve is a PARSE error in the reference,
  // before any filter looks at it; k_ec_resolve rewrites the record when the equation fails
  const bool pending = ok & (w.ec_curve != 0u);
  const unsigned long long kp = pending ? keypos_pack(w.ec_pos, w.ec_curve, w.ec_shift) : 0ull;
  if (pending) flags |= FL_KEY_PENDING;
  if (keypos_out) *keypos_out = kp;
  else if (pending) a.keypos[idx] = kp;
  uint32_t slen = 0, s[5] = {0, 0, 0, 0, 0};
  int32_t exp_hour = 0;
  if (ok) {
    // NewExpDateFromTime: Truncate(time.Hour) = floor (storage/types.go:339-346)
    long long q = w.not_after / 3600;
    if (w.not_after % 3600 < 0) q -= 1;
    exp_hour = (int32_t)q;
    slen = w.serial_len > 0xffffu ? 0xffffu : w.serial_len;
    if (w.serial_len > 20) flags |= CTMR_FL_LONG_SERIAL;
#pragma unroll
    for (int k = 0; k < 5; k++) s[k] = w.serial_w[k];
  }
  o0 = make_uint4(status | (flags << 8) | (slen << 16), (uint32_t)exp_hour, iss, s[0]);
  o1 = make_uint4(s[1], s[2], s[3], s[4]);
  const uint2 ml = make_uint2(ok ? w.meta_issuer : META_NONE, ok ? w.meta_crl : META_NONE);
  if (a.meta_loc) a.meta_loc[idx] = ml;
  if (meta_words) *meta_words = ml;
}

template <bool STRICT = false, class R>
__device__ __forceinline__ void map_one(R& r, uint64_t len64, uint64_t idx, const MapArgs& a, uint4& o0, uint4& o1) {
  const EntryIn in = load_entry_in(a, idx);
  map_one<STRICT>(r, len64, idx, a, in, o0, o1);
}

template <bool STRICT = false, class R>
__device__ __forceinline__ void map_one(R& r, uint64_t len64, uint64_t idx, const MapArgs& a) {
  uint4 o0, o1;
  map_one<STRICT>(r, len64, idx, a, o0, o1);
  uint4* out = (uint4*)(a.records + idx);
  out[0] = o0;
  out[1] = o1;
}

// Record store for the one-wave-per-workgroup window kernels: the 64 records of the wave (2 KiB,
// contiguous) are transposed through LDS so that each of the two store instructions writes 1 KiB of
// consecutive bytes (whole 64-B sectors) instead of 64 half-sectors 32 B apart.  The window area is
// free by now: every lane of the wave has finished its walk.
__device__ __forceinline__ void store_records_wave(const MapArgs& a, uint64_t first, bool live, const uint4& o0,
                                                   const uint4& o1) {
  uint4* t = (uint4*)smem;
  const uint32_t lane = threadIdx.x;
  __builtin_amdgcn_wave_barrier();
  if (live) {
    t[2 * lane] = o0;
    t[2 * lane + 1] = o1;
  }
  __builtin_amdgcn_wave_barrier();
  const uint64_t rem = a.n - first;  // records of this wave
  const uint32_t nvec = rem >= 64 ? 128u : (uint32_t)rem * 2u;
  uint4* out = (uint4*)(a.records + first);
  if (lane < nvec) st_stream16(out + lane, t[lane]);
  if (64u + lane < nvec) st_stream16(out + 64 + lane, t[64 + lane]);
}

// Window map (variant 13; the exchange modes and `map_variant = 13` use it, the default is k_map_fused in reduce.h):
// one certificate per lane, all 64 lanes busy, DER stays in global memory and is pulled through a per-lane LDS
// window.  One wave per workgroup, so LDS (not the 256-thread granule) sets the occupancy: WinGeo<WCH>::LDS_BYTES per wave.
// The first fill is wave-cooperative (coop_fill, readers.h).
template <int WCH, bool STRICT = false>
__global__ void __launch_bounds__(64) k_map_winc(MapArgs a) {
  const uint64_t first = (uint64_t)blockIdx.x * 64;
  const uint32_t lane = threadIdx.x;
  const uint64_t i = first + lane;
  const bool live = i < a.n;
  const uint64_t limit = map_limit(a);
  uint64_t lo = 0, hi = 0;
  if (live) cert_range(a.offsets, a.ends, i, lo, hi);
  const WaveBuf wb = wave_buf(a.payload, limit, lo);  // (lane 0 is live whenever the workgroup exists: k_map_fused)
  const uint32_t lrel = wave_rel(wb, lo, live);
  const uint32_t w_me = lrel == REL_NONE ? REL_NONE : (lrel & ~3u);  // (dword aligned: the window loses at most 3 bytes in front)
  coop_fill<WCH, false>(wb, w_me, lane);
  uint4 o0 = make_uint4(0, 0, 0, 0), o1 = o0;
  if (live) {
    // (out of the descriptor's reach: a window that holds nothing — every read takes this reader's global-memory path)
    WinReaderC<WCH> r{{(const uint32_t*)a.payload, lo, limit, (uint32_t*)(smem + win_off<WCH>(lane)),
                       lrel == REL_NONE ? (int32_t)0x80000000 : (int32_t)(w_me - lrel), wb, lrel}};
    map_one<STRICT>(r, hi - lo, i, a, o0, o1);
  }
  store_records_wave(a, first, live, o0, o1);
}

}  // namespace ctmr
