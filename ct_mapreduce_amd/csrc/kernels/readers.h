// kernels/readers.h — byte readers: how one lane gets at the bytes of its certificate (global memory, or a per-lane LDS window with wave-cooperative fills).
// gfx950 (CDNA4, wave64) only; part of kernels.h, which includes the pieces in dependency order.
#pragma once
#include "../ctmr_dev.h"
#include "../entry_decode.h"
#include "../synth.h"

namespace ctmr {

struct __attribute__((packed, aligned(1))) U16t { uint32_t a, b, c, d; };  // unaligned 16-byte access

// The certificate bytes are read once and the records are written once: window fills and record stores are
// NON-TEMPORAL (`nt` on gfx950), so that they do not displace the known-certificate table's lines from L2 / the
// Infinity Cache on their way through.  Measured A/B/A/B on one box, 100 M entries: 23.63 → 23.11 ms
// (profiles/r02/sweep_nontemporal.txt); loads alone 23.24, stores alone 22.83–23.31; the same hint on the per-entry
// input words and the reduce-state word measured no further change.
typedef uint32_t ctmr_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_payload16(const uint4* p) {
  const ctmr_u32x4 t = __builtin_nontemporal_load((const ctmr_u32x4*)p);
  return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void st_stream16(uint4* p, const uint4& v) {
  ctmr_u32x4 t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, (ctmr_u32x4*)p);
}
// ------------------------------------------------------------------ the per-lane LDS windows of one wave
// One wave per workgroup; lane c owns a 256-byte window (16 chunks of 16 bytes) at smem + win_off(c): lane stride 272
// bytes (16-B aligned for ds_write_b128, ≤ 4-way bank conflicts on equal in-window offsets), filled through registers:
// 16 buffer_load_dwordx4 → 16 ds_write_b128 per lane.
// (Round 3's LDS-DMA variant of the fill — CTMR_WIN_GLDS, global_load_lds_dwordx4 — measured no gain and left the tree in
//  round 6 together with the 64-bit address arithmetic it shared with the fill below: EXPERIMENTS.md.)
extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
// Round 6: the window geometry decides how many waves a CU holds — the map kernels are bound by the 160 KB of LDS, not
// by registers (133 / 157 VGPRs: three waves per SIMD, twelve per CU) — and once the fills and the reads were cheap their
// time followed the number of resident waves (a build that cost one wave per CU lost exactly 1/9: EXPERIMENTS.md).
//   16 chunks, stride 272  17 408 B per wave → 9 waves per CU; equal in-window offsets collide 4-way (rounds 1-5)
//   15 chunks, stride 248  15 872 B per wave: the occupancy API promises 10 waves per CU (scripts/probe_lds_occupancy.hip:
//                           ≤ 16 447 B), the counters show 9 as before — LDS is handed out in pieces of 1 280 B
//                           (160 KB / 128), 15 872 B take thirteen of them, 16 640 B (profiles/r06/win_15_chunks_*)
//   14 chunks, stride 236  15 104 B per wave = twelve pieces → 10 waves per CU; 59 dwords a lane: an ODD stride, equal
//                           in-window offsets never collide; the windows are 4-byte aligned (a chunk is stored as two
//                           ds_write2_b32).  224 bytes hold what the walk reads of the front of a synthetic certificate
//                           — through the first two octets of the modulus, 220 bytes in — only because a window now
//                           begins AT its position give or take 3 bytes (a chunk load needs dword alignment; rounds 1-5
//                           began windows on 16-byte boundaries and lost up to 15) and because the walk's hints
//                           (touch) say what is read, not a round figure.
// Each map kernel takes its geometry as a template parameter (WCH, round 6):
//   fast profile        13 chunks + 2 dwords, stride 220: eleven waves per CU, the same two fills per certificate — 21.2
//                       against 22.6 ms on one box (profiles/r06/win_13_chunks_*); ten waves at 14 chunks were worth 6 % over
//                       nine, the eleventh 6 % again
//   reference profile   14 chunks, stride 236, ten waves.  The eleventh wave bought 1.7 % here (the subjectAltName rounds
//                       make a wave's life longer, and what eleven waves ask of the memory system lengthens every round
//                       trip) and cost the mixed corpus 2 % (long subjects refill once more).  16 chunks (nine waves, the
//                       subjectAltName refilled in whole 128-byte lines: 1 742 instead of 1 961 bytes of traffic per
//                       certificate) ran within a per cent on two boxes and 9 % slower on a third
//                       (profiles/r06/win_geometry_*): -DCTMR_WIN_CH_STRICT=16 / =13 build the others.
#ifndef CTMR_WIN_CH_FAST
#define CTMR_WIN_CH_FAST 13
#endif
#ifndef CTMR_WIN_CH_STRICT
#define CTMR_WIN_CH_STRICT 14
#endif
constexpr int WIN_CH_FAST = CTMR_WIN_CH_FAST, WIN_CH_STRICT = CTMR_WIN_CH_STRICT;
#ifndef CTMR_SAN_ALIGN
#define CTMR_SAN_ALIGN 32u  // where a subjectAltName refill of a window that is no whole number of lines begins (coop_refill_lines)
#endif
//   13 chunks + 2 dwords, stride 220  (late in round 6) 14 080 B per wave = eleven pieces → 11 waves per CU.  216 bytes hold
//                           the front of a synthetic certificate only when the window begins BEHIND the two outer headers
//                           (Certificate, TBSCertificate: 8 octets) — those come from sixteen octets a lane loads for itself
//                           next to the first fill (WinReaderS::hd, der_walk.h HeadView).  The two dwords behind the thirteenth
//                           chunk are what the lane that would load a fourteenth stores of it.
#ifndef CTMR_WIN_XDW14
#define CTMR_WIN_XDW14 0u  // (2u: 14 chunks + 2 dwords = 232 bytes at the same stride — fewer subjectAltName rounds per wave, 3.05 against 3.42, and 3 % SLOWER: a window that ends inside a sector; profiles/r06/win_232_bytes_*)
#endif
template <int WCH>
struct WinGeo {
  static constexpr uint32_t XDW = WCH == 13 ? 2u : WCH == 14 ? CTMR_WIN_XDW14 : 0u;  // dwords behind the last whole chunk
  static constexpr uint32_t WBYTES = (uint32_t)WCH * 16u + XDW * 4u;
  static constexpr uint32_t STRIDE = WCH == 16 ? 272u : WCH == 15 ? 248u : WCH == 13 ? 220u : (uint32_t)WCH * 16u + 12u;
#ifdef CTMR_WIN_NO_SKIP
  static constexpr uint32_t SKIP = 0u;
#else
  static constexpr uint32_t SKIP = WCH == 13 ? 8u : 0u;
#endif  // octets of a Certificate in front of its first window (a TBSCertificate: half)
  static constexpr uint32_t LDS_BYTES = 64u * STRIDE;
  static constexpr uint32_t ALIGN = STRIDE % 16u == 0u ? 16u : STRIDE % 8u == 0u ? 8u : 4u;  // of a chunk in a window
};
constexpr uint32_t WIN_LDS_MAX = WinGeo<WIN_CH_STRICT>::LDS_BYTES > WinGeo<WIN_CH_FAST>::LDS_BYTES ? WinGeo<WIN_CH_STRICT>::LDS_BYTES
                                                                                               : WinGeo<WIN_CH_FAST>::LDS_BYTES;
template <int WCH>
__device__ __forceinline__ uint32_t win_off(uint32_t c) { return c * WinGeo<WCH>::STRIDE; }
typedef uint32_t ctmr_u32x4_a16 __attribute__((ext_vector_type(4), aligned(16)));
typedef uint32_t ctmr_u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef uint32_t ctmr_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t ctmr_u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
template <int WCH>
__device__ __forceinline__ void st_chunk(uint8_t* at, const uint4& v) {
  using vec_t = typename std::conditional<WinGeo<WCH>::ALIGN == 16u, ctmr_u32x4_a16,
                                          typename std::conditional<WinGeo<WCH>::ALIGN == 8u, ctmr_u32x4_a8, ctmr_u32x4_a4>::type>::type;
  vec_t t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  *(vec_t*)at = t;
}
// the XDW dwords behind the last whole chunk (stored by the lane that loaded the chunk they begin)
template <int WCH>
__device__ __forceinline__ void st_part(uint8_t* at, const uint4& v) {
  if constexpr (WinGeo<WCH>::XDW == 2u) {
    ctmr_u32x2_a4 t;
    t.x = v.x; t.y = v.y;
    *(ctmr_u32x2_a4*)at = t;
  } else if constexpr (WinGeo<WCH>::XDW == 1u) {
    *(uint32_t*)at = v.x;
  } else if constexpr (WinGeo<WCH>::XDW == 3u) {
    ctmr_u32x2_a4 t;
    t.x = v.x; t.y = v.y;
    *(ctmr_u32x2_a4*)at = t;
    *(uint32_t*)(at + 8) = v.z;
  }
}
__device__ __forceinline__ void st_half(uint8_t* at, uint32_t x, uint32_t y) {
  ctmr_u32x2_a4 t;
  t.x = x; t.y = y;
  *(ctmr_u32x2_a4*)at = t;
}
// lanes of a group of sixteen that load a chunk: the WCH whole ones and, with XDW, one more
template <int WCH>
constexpr uint32_t win_loaders() { return (uint32_t)WCH + (WinGeo<WCH>::XDW ? 1u : 0u); }

// Round 6: the wave's view of the payload as ONE buffer descriptor (SRSRC in scalar registers) whose base is the 128-byte
// line of the wave's first certificate, so that every window position of every lane is a 32-BIT offset from it.  Rounds
// 1-5 carried 64-bit payload offsets through the fills: per 16-byte chunk two ds_bpermute (the 64-bit window start of the
// certificate the chunk belongs to), two 64-bit adds, two 64-bit compares against the payload's end, an exec-mask branch
// around the load and four v_mov to zero the chunk when it lay beyond — 13 vector + 5 scalar instructions per chunk, and
// an s_waitcnt on every shuffle before its load could issue (≈ 200 + 100 instructions and ≈ 1 000 cycles per fill; two
// fills per certificate in the fast profile, five in the reference profile).  With the descriptor the hardware does the
// range check (a load at or beyond num_records returns zeros): one ds_bpermute, one add and one buffer_load per chunk, all
// sixteen shuffles in flight before the first load.
// REL_NONE: "no window" — beyond every num_records (≤ 2^31 − 1), so such a chunk loads as zeros.
constexpr uint32_t REL_NONE = 0x80000000u;
constexpr uint64_t REL_SPAN = 0x7e000000ull;  // certificates farther than this from the wave's base take the exact reader
struct WaveBuf {
  __amdgpu_buffer_rsrc_t rs;  // base = payload + b128; num_records = min(limit − b128, 2^31 − 1)
  uint64_t b128;              // payload offset of the base: wave-uniform, a multiple of 128
};
// lo0: a payload offset at or below every certificate of the wave IN THE FIRST ACTIVE LANE (a packed batch, a decoded
// get-entries blob: the first lane's own certificate — offsets ascend); other lanes' values are not looked at.  A lane
// whose certificate lies below it or ≥ REL_SPAN beyond (a caller-made entry view in no order) gets REL_NONE from
// wave_rel() and is handed to the exact reader by its kernel.
__device__ __forceinline__ WaveBuf wave_buf(const uint8_t* payload, uint64_t limit, uint64_t lo0) {
  const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)lo0), h = __builtin_amdgcn_readfirstlane((uint32_t)(lo0 >> 32));
  const uint64_t b128 = (((uint64_t)h << 32) | l) & ~127ull;
  uint64_t span = limit > b128 ? limit - b128 : 0ull;
  span = span > 0x7fffffffull ? 0x7fffffffull : span;
  return WaveBuf{__builtin_amdgcn_make_buffer_rsrc((void*)(payload + b128), 0, (int)span, 0x00020000), b128};
}
// the certificate at payload offset lo as an offset from the wave's base (REL_NONE: out of the descriptor's reach)
__device__ __forceinline__ uint32_t wave_rel(const WaveBuf& wb, uint64_t lo, bool has) {
  const uint64_t d = lo - wb.b128;
  return (has & (lo >= wb.b128) & (d < REL_SPAN)) ? (uint32_t)d : REL_NONE;
}
template <bool NT = true>
__device__ __forceinline__ uint4 ld_chunk(const WaveBuf& wb, uint32_t off) {  // non-temporal, like ld_payload16
  const ctmr_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(wb.rs, (int)off, 0, NT ? 2 /* nt */ : 0);
  return make_uint4(t.x, t.y, t.z, t.w);
}

// Wave-cooperative fill of all 64 windows: instead of every lane issuing 16 loads of ITS certificate (64 uncoalesced
// 16-byte requests per instruction), 16 adjacent lanes fetch the 16 chunks of one certificate's window, 4 certificates
// per instruction — the texture addresser sees 8 lanes per 128-byte line.  w_me = this lane's 16-byte aligned window
// start as an offset from the wave's base (REL_NONE: no certificate — its window fills with zeros); bytes at or beyond
// the payload's readable end read as zero.
// BARRIER_BEFORE_STORES: the windows are being re-filled (every lane must be done reading the old contents).
template <int WCH, bool BARRIER_BEFORE_STORES>
__device__ __forceinline__ void coop_fill(const WaveBuf& wb, uint32_t w_me, uint32_t lane) {
  const uint32_t sub16 = (lane & 15u) * 16u;
  const bool mine = (lane & 15u) < win_loaders<WCH>();  // (a window of 14 chunks: two lanes of a group idle)
  const bool whole = (lane & 15u) < (uint32_t)WCH;
  uint32_t o[16];
  uint4 v[16];
#pragma unroll
  for (int it = 0; it < 16; it++) o[it] = __shfl(w_me, 4 * it + (int)(lane >> 4));
#pragma unroll
  for (int it = 0; it < 16; it++) v[it] = ld_chunk(wb, mine ? o[it] + sub16 : REL_NONE);
  if (BARRIER_BEFORE_STORES) __builtin_amdgcn_wave_barrier();
  if constexpr (WinGeo<WCH>::XDW == 2u) {
    // one code path for the loaders: the first half of the chunk from every one of them, the second half from the whole ones
    // (a 4-byte aligned chunk is two ds_write2_b32 anyway; two differently shaped store blocks cost 17 registers)
    if (mine) {
#pragma unroll
      for (int it = 0; it < 16; it++) st_half(smem + win_off<WCH>(4 * it + (lane >> 4)) + sub16, v[it].x, v[it].y);
    }
    if (whole) {
#pragma unroll
      for (int it = 0; it < 16; it++) st_half(smem + win_off<WCH>(4 * it + (lane >> 4)) + sub16 + 8u, v[it].z, v[it].w);
    }
  } else {
    static_assert(WinGeo<WCH>::XDW == 0u, "extra dwords: two or none");
    if (whole) {
#pragma unroll
      for (int it = 0; it < 16; it++) st_chunk<WCH>(smem + win_off<WCH>(4 * it + (lane >> 4)) + sub16, v[it]);
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// The same for SOME windows (der_walk.h ext_san_coop: the lanes still walking a subjectAltName): a lane whose w_me is
// REL_NONE keeps its window as it is (its chunks load as zeros — no memory access — and are not stored).
// (NT = false — plain instead of non-temporal loads for the subjectAltName's refills, so that the half sector at a 224-byte
//  window's end might still be in L2 when the next round asks for it again — measured no less traffic: 1 976 against 1 961
//  bytes per certificate, profiles/r06/san_refills_plain_loads_*.)
template <int WCH, bool NT = true>
__device__ __forceinline__ void coop_refill_some(const WaveBuf& wb, uint32_t w_me, uint32_t lane) {
  const uint32_t sub16 = (lane & 15u) * 16u;
  const bool mine = (lane & 15u) < win_loaders<WCH>();
  const bool whole = (lane & 15u) < (uint32_t)WCH;
  uint32_t o[16];
  uint4 v[16];
#pragma unroll
  for (int it = 0; it < 16; it++) {
    o[it] = __shfl(w_me, 4 * it + (int)(lane >> 4));
    o[it] = mine ? o[it] : REL_NONE;
  }
#pragma unroll
  for (int it = 0; it < 16; it++) v[it] = ld_chunk<NT>(wb, o[it] + sub16);
  __builtin_amdgcn_wave_barrier();  // every lane is done reading the old contents
  if constexpr (WinGeo<WCH>::XDW == 2u) {
#pragma unroll
    for (int it = 0; it < 16; it++) {
      uint8_t* const at = smem + win_off<WCH>(4 * it + (lane >> 4)) + sub16;
      if (o[it] != REL_NONE) st_half(at, v[it].x, v[it].y);
      if ((o[it] != REL_NONE) & whole) st_half(at + 8u, v[it].z, v[it].w);
    }
  } else {
#pragma unroll
    for (int it = 0; it < 16; it++)
      if (o[it] != REL_NONE) st_chunk<WCH>(smem + win_off<WCH>(4 * it + (lane >> 4)) + sub16, v[it]);
  }
  __builtin_amdgcn_wave_barrier();
}

// … and of SOME CHUNKS of them (round 6, late): w_me carries, in its four low bits, the number of chunks wanted minus one —
// the subjectAltName's last round needs what is left of the value, not a whole window: the chunks behind it (the
// signature) are not loaded and keep whatever the window held (der_walk.h ext_san_coop never looks past the value's end;
// WinReaderC::part tells everybody else that the window is no longer whole).  w_me a multiple of 16 otherwise.
template <int WCH, bool NT = true>
__device__ __forceinline__ void coop_refill_some_n(const WaveBuf& wb, uint32_t w_me, uint32_t lane) {
  const uint32_t li = lane & 15u, sub16 = li * 16u;
  const bool whole = li < (uint32_t)WCH;
  uint32_t o[16];
  uint4 v[16];
#pragma unroll
  for (int it = 0; it < 16; it++) {
    const uint32_t x = __shfl(w_me, 4 * it + (int)(lane >> 4));
    o[it] = ((x != REL_NONE) & (li <= (x & 15u))) ? (x & ~15u) : REL_NONE;  // (li <= count - 1 < loaders)
  }
#pragma unroll
  for (int it = 0; it < 16; it++) v[it] = ld_chunk<NT>(wb, o[it] + sub16);
  __builtin_amdgcn_wave_barrier();  // every lane is done reading the old contents
  if constexpr (WinGeo<WCH>::XDW == 2u) {
#pragma unroll
    for (int it = 0; it < 16; it++) {
      uint8_t* const at = smem + win_off<WCH>(4 * it + (lane >> 4)) + sub16;
      if (o[it] != REL_NONE) st_half(at, v[it].x, v[it].y);
      if ((o[it] != REL_NONE) & whole) st_half(at + 8u, v[it].z, v[it].w);
    }
  } else {
#pragma unroll
    for (int it = 0; it < 16; it++)
      if (o[it] != REL_NONE) st_chunk<WCH>(smem + win_off<WCH>(4 * it + (lane >> 4)) + sub16, v[it]);
  }
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------ byte readers
// ld4(pos): 4-byte little-endian window at an arbitrary byte position, from two aligned dwords (ds_read2_b32 + v_alignbyte).
// Round 6 tried ONE ds_read_b32 at the byte address itself — gfx950 runs with unaligned LDS access enabled (hipcc emits
// ds_read_b32 / ds_read_u16 for align-1 pointers; scripts/probe_lds_unaligned.hip: the hardware returns the right bytes) —
// three vector instructions less on each of the walk's ≈ 90 reads.  The LDS pays for them: SQ_LDS_IDX_ACTIVE 4 017 →
// 10 430 cycles per wave (an unaligned dword is not one access), ten waves of a CU kept its one LDS pipeline two-thirds
// busy, and the map kernel ran 4.6 % SLOWER under the reference profile (38.7 against 36.9 ms per 100 M, A/B/A/B on one
// box, profiles/r06/lds_reads_*).  -DCTMR_LDS_UNALIGNED builds that form.
typedef uint32_t __attribute__((aligned(1))) ctmr_u32_u;
typedef uint16_t __attribute__((aligned(1))) ctmr_u16_u;
// the last window offset a 4-byte read may start at: WBYTES − 4 in both forms (the two-dword form's second dword may lie
// in the pad behind the window — every stride leaves at least 4 bytes — and none of its bytes is used then)
constexpr uint32_t LD4_SPAN = 4u;
__device__ __forceinline__ uint32_t lds_ld4(const uint32_t* win, uint32_t rel) {
#ifdef CTMR_LDS_UNALIGNED
  return *(const ctmr_u32_u*)((const uint8_t*)win + rel);
#else
  const uint32_t i = rel >> 2;
  return __builtin_amdgcn_alignbyte(win[i + 1], win[i], rel & 3u);
#endif
}
struct GlobalReader {
  const uint32_t* words;  // 4-byte aligned base of the buffer (kernel argument: global address space)
  uint64_t base;          // byte offset of this certificate inside the buffer
  __device__ __forceinline__ uint32_t ld4(uint32_t pos) const {
    const uint64_t a = base + pos;
    const uint64_t i = a >> 2;
    return __builtin_amdgcn_alignbyte(words[i + 1], words[i], (uint32_t)a & 3u);
  }
  __device__ __forceinline__ uint32_t ldg(uint32_t pos) const { return ld4(pos); }
  __device__ __forceinline__ RawCert raw() const { return RawCert{words, base}; }  // spki_key.h: the out-of-line key checks
  __device__ __forceinline__ void touch(uint32_t, uint32_t) const {}
  __device__ __forceinline__ void touch_tail(uint32_t, uint32_t) const {}
};

// Per-lane LDS window ("software cache") over a certificate that stays in global memory.
// Each lane owns WCH 16-byte chunks of LDS (lane stride WCH*16+16 bytes: 16-B aligned for
// ds_write_b128, and ≤4-way bank conflicts on the dword reads).  touch(pos, need) refills the
// window with WCH independent global_load_dwordx4 (one burst, one memory latency) when the next
// `need` bytes are not resident; ld4 hits LDS inside the window and falls back to a plain global
// load outside it — so correctness never depends on where the window is.  The walk touches the
// front of the certificate and the extension block; SPKI body, SAN body and signature are
// skipped by length and therefore never fetched from HBM.
template <int WCH>
struct WinReader {
  const uint32_t* g32;  // payload, dword view (global)
  uint64_t base;        // certificate start (byte offset into payload)
  uint64_t limit;       // readable bytes of payload (offsets[n] + CTMR_PAYLOAD_PAD)
  uint32_t* win;        // this lane's window words in LDS
  int32_t grel;         // window start relative to the certificate start; (base+grel) % 4 == 0 (% 16 after a per-lane refill)
  // round 6 (set by the kernel behind the constructor): the wave's buffer descriptor and this certificate's start as an
  // offset from its base (REL_NONE: out of reach — cooperative fills leave such a lane's window alone / zero)
  WaveBuf wb;
  uint32_t lrel;
  static constexpr uint32_t WBYTES = WinGeo<WCH>::WBYTES;
  __device__ __forceinline__ uint32_t wrel(uint32_t pos, uint32_t align) const {  // window start for [pos, …), from the wave's base
    return lrel == REL_NONE ? REL_NONE : ((lrel + pos) & ~(align - 1u));
  }

  __device__ __forceinline__ uint32_t ld4(uint32_t pos) const {
    const uint32_t rel = pos - (uint32_t)grel;
    if (rel <= WBYTES - LD4_SPAN) return lds_ld4(win, rel);
    const uint64_t a = base + pos;
    const uint64_t i = a >> 2;
    return __builtin_amdgcn_alignbyte(g32[i + 1], g32[i], (uint32_t)a & 3u);
  }
  __device__ __forceinline__ RawCert raw() const { return RawCert{g32, base}; }  // spki_key.h: the out-of-line key checks
  // spki_key.h key_reader_of: a snapshot of this base reader, by value (window hit, else a plain global load; none of the
  // derived readers' miss bookkeeping)
  __device__ __forceinline__ WinReader<WCH> key_bytes() const { return WinReader<WCH>{g32, base, limit, win, grel, wb, lrel}; }
  __device__ __forceinline__ uint32_t ldg(uint32_t pos) const {  // straight from global memory
    const uint64_t a = base + pos;
    const uint64_t i = a >> 2;
    return __builtin_amdgcn_alignbyte(g32[i + 1], g32[i], (uint32_t)a & 3u);
  }
  __device__ __forceinline__ void refill(uint32_t pos) {
    const uint64_t g = (base + pos) & ~15ull;
    grel = (int32_t)(int64_t)(g - base);
    const uint4* src = (const uint4*)g32 + (g >> 4);
    uint4 v[WCH];
#pragma unroll
    for (int k = 0; k < WCH; k++)
      v[k] = (g + 16u * k + 16u <= limit) ? src[k] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < WCH; k++) st_chunk<WCH>((uint8_t*)win + 16 * k, v[k]);
    if constexpr (WinGeo<WCH>::XDW != 0u) {
      const uint4 x = (g + 16u * WCH + 16u <= limit) ? src[WCH] : make_uint4(0, 0, 0, 0);
      st_part<WCH>((uint8_t*)win + 16 * WCH, x);
    }
  }
  __device__ __forceinline__ void touch(uint32_t pos, uint32_t need) {
    if (need > WBYTES - 16u) need = WBYTES - 16u;  // the window start is 16-B aligned in HBM
    const uint32_t rel = pos - (uint32_t)grel;
    if (rel > WBYTES - need) refill(pos);
  }
  __device__ __forceinline__ void touch_tail(uint32_t pos, uint32_t) { touch(pos, 256); }
};

// WinReader whose touch_tail() (the one refill every lane of the wave reaches at the same program point,
// right behind the SubjectPublicKeyInfo header) is wave-cooperative like the first fill of k_map_winc:
// 16 adjacent lanes fetch the 16 chunks of one certificate's window, 4 certificates per load
// instruction.  Falls back to the per-lane refill when some lane of the wave is not at that point.
template <int WCH>
struct WinReaderC : WinReader<WCH> {
  static_assert(WCH >= 12 && WCH <= 16, "a window is filled by groups of sixteen lanes");
  bool part;  // the last cooperative refill brought only the chunks it was asked for (coop_refill_lines_to): what lies behind
              // them in the window is stale — readers of the window other than the walk (reduce.h: the memo pre-check) ask
  __device__ __forceinline__ void touch_tail(uint32_t pos, uint32_t) {
    if (__ballot(1) != ~0ull) {
      this->refill(pos);
      return;
    }
    const uint32_t w = this->wrel(pos, 4u);  // (a chunk load needs dword alignment only: the window begins AT pos, give or take 3)
    if (w != REL_NONE) this->grel = (int32_t)(w - this->lrel);
    coop_fill<WCH, true>(this->wb, w, threadIdx.x & 63u);
  }
  // der_walk.h touch_all (round 6) — a hint at a point of the walk EVERY lane passes (the walk never returns early): the lanes
  // whose window lacks [pos, pos + need) are refilled TOGETHER, sixteen lanes per certificate, the others keep theirs.
  // touch() refills lane by lane — sixteen uncoalesced loads and a round trip the whole wave waits for: what a long subject
  // (the mixed corpus: 40 % of its certificates) or a window a few bytes short costs there.
  __device__ __forceinline__ void touch_coop(uint32_t pos, uint32_t need) {
    if (need > WinReader<WCH>::WBYTES - 16u) need = WinReader<WCH>::WBYTES - 16u;
    const bool lack = pos - (uint32_t)this->grel > WinReader<WCH>::WBYTES - need;
    if (__ballot(1) != ~0ull) {  // the batch's last wave
      if (lack) this->refill(pos);
      return;
    }
    if (__ballot(lack) == 0ull) return;
    const uint32_t w = lack ? this->wrel(pos, 4u) : REL_NONE;
    if (w != REL_NONE) this->grel = (int32_t)(w - this->lrel);
    else if (lack) this->refill(pos);  // (out of the descriptor's reach: the lane's own loads)
    coop_refill_some<WCH>(this->wb, w, threadIdx.x & 63u);
  }
  // der_walk.h ext_san_coop — wave-collective (every lane of a WHOLE wave calls it from converged code): the lanes that
  // `want` get their window refilled at pos, 16 lanes per certificate as above; the others keep theirs.
  __device__ __forceinline__ void coop_refill(uint32_t pos, bool want) {
    const uint32_t w = want ? this->wrel(pos, 16u) : REL_NONE;
    if (w != REL_NONE) this->grel = (int32_t)(w - this->lrel);
    coop_refill_some<WCH>(this->wb, w, threadIdx.x & 63u);
  }
  // … the same with the window starting on a 128-byte LINE of the payload: two whole lines, the next round's window begins
  // where this one ends (der_walk.h ext_san_coop, round 6)
  __device__ __forceinline__ void coop_refill_lines(uint32_t pos, bool want) {
    // whole lines when the window is a whole number of them (256 bytes); a window of 224 bytes is seven 32-byte sectors and
    // begins on one: from a 64-byte boundary its last half sector was fetched again by the next round — 1 955 against 1 899
    // bytes of traffic per certificate and 40.1 against 38.7 / 38.9 ms per step on one box (16-byte boundaries: 1 912 bytes,
    // 39.1 / 37.8 ms; profiles/r06/san_refill_alignment_*)
    constexpr uint32_t AL = (WinReader<WCH>::WBYTES % 128u == 0u) ? 128u : CTMR_SAN_ALIGN;
    const uint32_t w = want ? this->wrel(pos, AL) : REL_NONE;     // (the wave's base is a multiple of 128)
    if (w != REL_NONE) this->grel = (int32_t)(w - this->lrel);
    coop_refill_some<WCH>(this->wb, w, threadIdx.x & 63u);
  }
  // … and only as far as `end` (+ 4 octets: a read that begins in front of the end may reach across it), a certificate
  // offset: the rest of the window is not loaded
  __device__ __forceinline__ void coop_refill_lines_to(uint32_t pos, bool want, uint32_t end) {
    constexpr uint32_t AL = (WinReader<WCH>::WBYTES % 128u == 0u) ? 128u : CTMR_SAN_ALIGN;
    if constexpr (AL < 16u) {  // (the chunk count travels in the window start's four low bits: whole windows otherwise)
      coop_refill_lines(pos, want);
      return;
    }
    uint32_t w = want ? this->wrel(pos, AL) : REL_NONE;
    if (w != REL_NONE) {
      this->grel = (int32_t)(w - this->lrel);
      const uint32_t need = (this->lrel + end + 4u) - w;  // (pos < end: at least 5)
      uint32_t n = (need + 15u) >> 4;
      n = n > win_loaders<WCH>() ? win_loaders<WCH>() : n;
      part = part | (n < win_loaders<WCH>());
      w |= n - 1u;
    }
    coop_refill_some_n<WCH>(this->wb, w, threadIdx.x & 63u);
  }
  // the two octets at pos, which the caller knows to lie in the window (holds): two byte reads, no alignment arithmetic
  __device__ __forceinline__ uint32_t ld2(uint32_t pos) const {
    const uint8_t* b = (const uint8_t*)this->win + (pos - (uint32_t)this->grel);
#ifdef CTMR_LDS_UNALIGNED
    return *(const ctmr_u16_u*)b;
#else
    return (uint32_t)b[0] | ((uint32_t)b[1] << 8);
#endif
  }
  __device__ __forceinline__ uint32_t wend() const { return (uint32_t)this->grel + WinReader<WCH>::WBYTES; }  // certificate offset of the window's end
  __device__ __forceinline__ bool holds(uint32_t pos, uint32_t need) const {  // [pos, pos + need) lies in the window
    return pos - (uint32_t)this->grel <= WinReader<WCH>::WBYTES - need;
  }
  static __device__ __forceinline__ bool whole_wave() { return __ballot(1) == ~0ull; }
  static __device__ __forceinline__ bool any_lane(bool x) { return __ballot(x) != 0ull; }
};

// WinReaderC whose ld4() is served by the LDS window ALONE: no per-access "outside the window → global load"
// branch (88 ld4 per certificate, each of which used to carry its own exec-mask dance).  An access that does fall
// outside is clamped and remembered in `miss`; the kernel then repeats that certificate with the exact GlobalReader.
// On well-formed certificates the walk's touch() hints keep every ld4 inside (measured on the synthetic corpus: 0
// misses in 200 000 certificates once the three reads behind the TBS go through ldg()).
// Something to do while a refill is in flight (k_map_pipe: one step of the PREVIOUS group's table probe — its atomic
// is issued between the tail loads and the refill, and looked at once the refill's LDS stores have waited for their
// loads, so its round trip hides behind theirs).
struct NoRefillHook {
  __device__ __forceinline__ void issue() {}
  __device__ __forceinline__ void resolve() {}
  __device__ __forceinline__ void note_issuer(const uint32_t*, uint32_t, uint32_t, uint32_t) {}
};

template <int WCH, class Hook = NoRefillHook>
struct WinReaderS : WinReaderC<WCH> {
  static constexpr bool kNoClamp = true;  // ld4 clamps into the window itself
  // `miss` holds the LARGEST window-relative offset any read asked for (round 6: one v_max per read instead of compare +
  // select + or); missed() = some read lay outside the window.  Constructed with 0 (or ~0: no window at all).
  mutable uint32_t miss;
  __device__ __forceinline__ bool missed() const { return miss > WinReader<WCH>::WBYTES - LD4_SPAN; }
  // The 32 bytes behind the TBSCertificate (signatureAlgorithm, the signatureValue header, its pad octet), fetched
  // by touch_tail() TOGETHER with the extension-block refill: the three ldg() reads at the end of the walk were
  // three dependent, uncoalesced global round trips per wave; now they are register selects.
  uint32_t tl[8];
  uint32_t tl_pos;  // certificate offset of tl[0]'s first byte; 0x80000000 = not fetched (positions are < 2^31)
  // … and the 16 octets [pos − 12, pos + 4) around the end of the SubjectPublicKeyInfo (an RSA key's publicExponent sits
  // there, spki_key.h): one more unaligned 16-byte load in the same burst instead of dependent reads ~0.3 KB off the window
  uint32_t kt[4];
  bool kt_ok;
  // … and, for geometries whose first window begins behind the outer headers (WinGeo::SKIP), the certificate's first
  // sixteen octets, loaded by the lane itself next to the first fill (the kernel sets them): der_walk.h reads the
  // Certificate and TBSCertificate headers out of these
  static constexpr bool kHead = WinGeo<WCH>::SKIP != 0u;
  uint32_t hd[4];
  bool hd_ok;
  Hook hook{};  // by value: a pointer to state that lives across loop iterations keeps that state out of registers
  __device__ __forceinline__ uint32_t ld4(uint32_t pos) const {
    uint32_t rel = pos - (uint32_t)this->grel;
    constexpr uint32_t LAST = WinReader<WCH>::WBYTES - LD4_SPAN;
    miss = rel > miss ? rel : miss;
    rel = rel > LAST ? LAST : rel;
    return lds_ld4(this->win, rel);
  }
  // two octets, clamped and remembered like ld4 (the last read of the front window: spki_key.h)
  __device__ __forceinline__ uint32_t ld2c(uint32_t pos) const {
    uint32_t rel = pos - (uint32_t)this->grel;
    constexpr uint32_t LAST2 = WinReader<WCH>::WBYTES - 2u;
    miss = rel > LAST2 ? 0xffffffffu : miss;
    rel = rel > LAST2 ? LAST2 : rel;
    const uint8_t* b = (const uint8_t*)this->win + rel;
    return (uint32_t)b[0] | ((uint32_t)b[1] << 8);
  }
  // der_walk.h (strict_extensions): the contents of this element are read octet by octet — not through the window
  __device__ __forceinline__ void defer_exact() { miss = 0xffffffffu; }
  __device__ __forceinline__ void touch_tail(uint32_t pos, uint32_t tail) {
    const uint64_t ta = this->base + tail;
    const bool have = ta + 32u <= this->limit;
    const uint8_t* tp = (const uint8_t*)this->g32 + (have ? ta : 0ull);
    const U16t a = *(const U16t*)tp, b = *(const U16t*)(tp + 16);  // in flight with the refill below
    const bool kh = (pos >= 12u) & (this->base + pos + 4u <= this->limit);
    const U16t kk = *(const U16t*)((const uint8_t*)this->g32 + (kh ? this->base + pos - 12u : 0ull));
    kt[0] = kk.a; kt[1] = kk.b; kt[2] = kk.c; kt[3] = kk.d;
    kt_ok = kh;
    hook.issue();
    WinReaderC<WCH>::touch_tail(pos, tail);
    hook.resolve();
    tl[0] = a.a; tl[1] = a.b; tl[2] = a.c; tl[3] = a.d;
    tl[4] = b.a; tl[5] = b.b; tl[6] = b.c; tl[7] = b.d;
    tl_pos = have ? tail : 0x80000000u;
  }
  struct KeyTail { bool valid; uint32_t at; uint32_t w[4]; };
  __device__ __forceinline__ KeyTail key_tail(uint32_t pos) const {  // spki_key.h key_tail_of: pos = what touch_tail got
    return KeyTail{kt_ok, pos - 12u, {kt[0], kt[1], kt[2], kt[3]}};
  }
  // der_walk.h: the issuer Name [pos, pos+len) has just been walked — the front window still holds it
  __device__ __forceinline__ void note_issuer(uint32_t pos, uint32_t len) {
    hook.note_issuer(this->win, pos - (uint32_t)this->grel, len, WinReader<WCH>::WBYTES - 8u);
  }
  __device__ __forceinline__ uint32_t ldg(uint32_t pos) const {
    const uint32_t off = pos - tl_pos;
    const uint32_t wi = off >> 2;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (uint32_t k = 0; k < 7; k++) {
      lo = wi == k ? tl[k] : lo;
      hi = wi == k ? tl[k + 1] : hi;
    }
    uint32_t v = __builtin_amdgcn_alignbyte(hi, lo, off & 3u);
    if (off > 27u) v = WinReader<WCH>::ldg(pos);  // a long AlgorithmIdentifier, or no prefetch: the real load
    return v;
  }
};

}  // namespace ctmr
