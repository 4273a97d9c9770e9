// kernels/pem.h — PEM write-back.
// gfx950 (CDNA4, wave64) only; part of kernels.h, which includes the pieces in dependency order.
#pragma once
#include "exchange.h"

namespace ctmr {

// ------------------------------------------------------------------ PEM write-back
// pem.EncodeToMemory(&pem.Block{Type: "CERTIFICATE", Bytes: aCert.Raw}) of every newly unknown
// certificate (storage/filesystemdatabase.go:167-175,196-200): "-----BEGIN CERTIFICATE-----\n",
// base64.StdEncoding in 64-column lines each ended by "\n", "-----END CERTIFICATE-----\n".
__host__ __device__ inline uint64_t pem_len(uint64_t L) {
  const uint64_t b64 = 4 * ((L + 2) / 3);
  return 28 + b64 + (b64 + 63) / 64 + 26;
}

// One entry of the NEW list: where its DER lies and how long it is (k_pem_encode reads these 16 bytes instead of following
// idx → offsets again: one dependent load less in front of every block).
struct PemCertInfo { uint64_t lo, len; };

__global__ void __launch_bounds__(256) k_pem_len(const uint64_t* offsets, const uint64_t* ends, const uint64_t* idx,
                                                 uint64_t n_idx, uint64_t* pem_off, PemCertInfo* info) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r > n_idx) return;
  if (r == n_idx) {
    pem_off[r] = 0;  // the exclusive scan turns this slot into the total
    return;
  }
  uint64_t lo, hi;
  cert_range(offsets, ends, idx[r], lo, hi);
  pem_off[r] = pem_len(hi - lo);
  info[r] = PemCertInfo{lo, hi - lo};
}

struct __attribute__((packed, aligned(1))) U16 { uint32_t a, b, c, d; };  // unaligned 16-byte access (entries.h, meta.h)

// ------------------------------------------------------------------ k_pem_encode (round 5: output-block design)
// The PEM blocks of the NEW list are ONE contiguous byte stream (pem_off is an exclusive scan), so the work is cut by
// OUTPUT bytes, not by certificate: one wave produces one 7 KiB block of the stream at a time — whatever certificates
// and parts of certificates lie in it — and every global access is a naturally aligned 16-byte vector covering whole
// 128-byte lines:
//   1. lane j fetches the bounds of the j-th certificate that overlaps the block (k_pem_blocks left the first one's
//      index per block, k_pem_len its place in the payload) and works out which of its base64 LINES fall into the block;
//   2. the input bytes of those lines (≤ 5.5 KiB, contiguous per certificate) come in as aligned, non-temporal 16-byte
//      loads, 1 KiB per instruction, and are parked in LDS;
//   3. ONE LANE ENCODES ONE LINE (two of them per block: lines lane and 64 + lane): 48 bytes from LDS (aligned dword reads + v_alignbyte), sixteen v_perm_b32 make the
//      big-endian 24-bit groups, the 64 characters come from the 64-byte alphabet in LDS (one v_bfe + one ds_read_u8 per
//      character; sixteen dwords in sixteen banks: conflict-free) and go to the block's image in LDS at their final
//      stream position — a line is 65 bytes, so every line sits at another alignment: sixteen v_alignbyte shift it onto
//      aligned dwords, the two ends and the line end are byte stores.  A certificate's last, shorter line is a line like
//      any other (PEM_FOLD; until late in round 5 up to four lanes encoded it apart in 12-byte → 16-character tasks — 120
//      vector instructions per block for a handful of lanes): the lane encodes 48 octets whatever the line holds (the two
//      octets behind the certificate are zeroed in LDS first), sets the '=' and the line end where they belong, and what it
//      writes behind them is the place of the two framing lines, which are written afterwards;
//   4. the image leaves as 448 aligned, non-temporal 16-byte stores.
// The loop over a wave's blocks is software-pipelined: while block i is encoded, the BYTES of block i + 1 and the BOUNDS of
// block i + 2 are in flight.
// History (profiles/r05/pem_*): round 4's kernel — one wave per certificate, unaligned dwordx3 loads, 16-byte stores at
// a 65-byte pitch that never meet a sector boundary, a byte store per line end, ≈ 9 VALU instructions per character for
// the alphabet — 3.4 TB/s read + written.  The block design with one lane per 12-byte task: 2.9 TB/s unpipelined, 3.1
// pipelined, 3.4 with a conflict-free alphabet table — and the counters said why: 935 vector instructions and 202 LDS
// instructions per 4 KiB block, the vector ALU 80 % busy.  One lane per line cuts both by more than half (580 vector
// instructions per 3.5 KiB block, of which the lines themselves are 190: the rest is per-block bookkeeping — bounds, plan,
// assignment, last lines, framing lines); two lines per lane and 7 KiB blocks pay that bookkeeping once for twice the
// bytes: 13.9 → 12.7 ms on 16 M certificates, at 12 instead of 16 waves per CU (163 VGPRs, 13 KB of LDS per wave).  The last
// lines folded into the line pass: 468 → 427 vector instructions per 3.5 KiB, 151 VGPRs, 12.8 → 12.4 ms.
#ifndef CTMR_PEM_LPL
#define CTMR_PEM_LPL 2
#endif
constexpr uint32_t PEM_LPL = CTMR_PEM_LPL;    // base64 lines per lane and block: the per-block bookkeeping is paid once for them
constexpr uint32_t PEM_S = 3584 * PEM_LPL;    // output bytes per block: ≈ 55 lines per 3584 bytes, so that one lane per line fits
constexpr uint32_t PEM_MARGIN = 80;           // a line that straddles a block edge is encoded whole by both blocks
constexpr uint32_t PEM_NQ = PEM_LPL == 1 ? 3 : PEM_LPL == 2 ? 6 : 8;                // 16-byte input loads per lane and block
constexpr uint32_t PEM_IN_CHUNKS = PEM_LPL == 1 ? 192 : PEM_LPL == 2 ? 352 : 512;     // 16-byte input chunks parked per pass
constexpr uint32_t PEM_LUT_BYTES = 64 + 64;   // the alphabet + the two framing lines
constexpr uint32_t PEM_OBUF = PEM_S + 2 * PEM_MARGIN;
constexpr uint32_t PEM_IBUF = PEM_IN_CHUNKS * 16 + 64;
constexpr uint32_t PEM_TRASH = 256;           // one dword per lane: where a store that must not happen goes
constexpr uint32_t PEM_WAVE_LDS = PEM_OBUF + PEM_IBUF + PEM_TRASH;
constexpr uint32_t PEM_WAVES = 4;             // waves per workgroup (they share the tables, nothing else)
constexpr uint32_t PEM_LDS_BYTES = PEM_WAVES * PEM_WAVE_LDS;  // dynamic; + PEM_LUT_BYTES static
#ifndef CTMR_PEM_FOLD
#define CTMR_PEM_FOLD 1
#endif
constexpr bool PEM_FOLD = CTMR_PEM_FOLD != 0; // a certificate's shorter last line is encoded by a line lane like any other line
constexpr uint32_t PEM_FAST_CERTS = PEM_FOLD ? 32 : 16;  // certificates per block the one-pass path takes (without PEM_FOLD: 4 task lanes each)

__device__ __forceinline__ uint32_t b64_char(uint32_t v) {  // base64.StdEncoding alphabet
  int32_t off = 65;                 // 'A'
  off = v >= 26u ? 71 : off;        // 'a' - 26
  off = v >= 52u ? -4 : off;        // '0' - 52
  off = v == 62u ? -19 : off;       // '+'
  off = v == 63u ? -16 : off;       // '/'
  return (uint32_t)((int32_t)v + off);
}

// block_first[b] = index (into the NEW list) of the certificate whose PEM block holds stream byte max(0, b·S − a),
// a = the output buffer's address mod 16 (block b covers stream bytes [b·S − a, (b+1)·S − a)).
__global__ void __launch_bounds__(256) k_pem_blocks(const uint64_t* pem_off, uint64_t n_idx, uint32_t a, uint32_t* block_first) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_idx) return;
  const uint64_t p0 = pem_off[r], p1 = pem_off[r + 1];
  if (r == 0) block_first[0] = 0u;
  for (uint64_t b = (p0 + a + PEM_S - 1) / PEM_S; b * PEM_S < p1 + a; b++)
    if (b) block_first[b] = (uint32_t)r;
}

// lane j: the j-th certificate of a pass over a block
struct PemCert {
  uint64_t p0, p1, lo, len;  // its PEM block [p0, p1) in the stream, its DER [lo, lo + len) in the payload
  bool have;                 // such a certificate exists (r < n_idx)
};
// what lane j's certificate contributes to the block [B0, B1): nfl whole lines from line ln_lo on (48 bytes → 64 characters
// + line end each), the ntk tasks (12 bytes → 16 characters) of its last, shorter line from task tk_lo on, and the nch
// 16-byte input chunks from payload offset a0 they read
struct PemPlan {
  uint32_t ln_lo, nfl, tk_lo, ntk, nch, nq, ncert;
  uint64_t a0;
  bool inblk;
};
// what a lane does in a block: its line (lines), its task of some certificate's last line (tasks); ib / ob: where the
// certificate's byte 0 lies in ibuf, its first base64 character in obuf
struct PemWork {
  uint32_t ln[PEM_LPL], l_ib[PEM_LPL], l_len[PEM_LPL];  // (l_len: the certificate's length — a line knows from it whether it is the last, shorter one)
  int32_t l_ob[PEM_LPL];
  bool l_act[PEM_LPL];
  uint32_t z_at;   // PEM_FOLD, lane j: where its certificate ends in ibuf — the two octets there are zeroed for the last group
  bool z_act;
  uint32_t tk, t_ib, t_len, t_nq;
  int32_t t_ob;
  bool t_act, any_task;
};

// (unconditional loads from a clamped index: a load under a branch reaches the loop's carried registers through a copy, and
//  the copy makes the compiler wait for EVERYTHING in flight right where the load was issued — measured: the first pipelined
//  build waited for the next block's bytes before it encoded the current one)
__device__ __forceinline__ PemCert pem_cert_load(const uint64_t* __restrict__ pem_off, const PemCertInfo* __restrict__ info,
                                                 uint64_t n_idx, uint64_t r) {
  const uint64_t rc = r < n_idx ? r : n_idx - 1u;  // n_idx >= 1 whenever the kernel runs
  PemCert c;
  c.have = r < n_idx;
  c.p0 = pem_off[rc];
  c.p1 = pem_off[rc + 1];
  const uint4 t = *(const uint4*)(info + rc);
  c.lo = (uint64_t)t.x | ((uint64_t)t.y << 32);
  c.len = (uint64_t)t.z | ((uint64_t)t.w << 32);
  return c;
}

__device__ __forceinline__ PemPlan pem_plan(const PemCert& c, uint64_t B0, uint64_t B1) {
  PemPlan p{0u, 0u, 0u, 0u, 0u, (uint32_t)((c.len + 11u) / 12u), 0u, 0ull, c.have && c.p0 < B1};
  p.ncert = (uint32_t)__popcll(__ballot(p.inblk));  // p0 ascends: lanes 0 … ncert − 1
  if (p.inblk) {
    const uint64_t body = c.p0 + 28u, body_end = c.p1 - 26u;  // the base64 lines with their line ends
    const uint64_t s = B0 > body ? B0 : body, e = B1 < body_end ? B1 : body_end;
    if (e > s) {
      const uint32_t L = (uint32_t)c.len, nfull = L / 48u;        // (a certificate is shorter than 2^31 bytes)
      const uint32_t ln_lo = (uint32_t)(s - body) / 65u, ln_hi = (uint32_t)(e - 1u - body) / 65u;
      const bool last = (L != 48u * nfull) & (ln_hi == nfull);     // the shorter last line lies in the block
      const uint32_t nlines = PEM_FOLD ? nfull + (L != 48u * nfull ? 1u : 0u) : nfull;  // PEM_FOLD: the shorter last line is a line
      const uint32_t fl_end = ln_hi + 1u < nlines ? ln_hi + 1u : nlines;  // lines: [ln_lo, fl_end)
      p.ln_lo = ln_lo;
      p.nfl = fl_end > ln_lo ? fl_end - ln_lo : 0u;
      p.tk_lo = 4u * nfull;
      p.ntk = last ? (PEM_FOLD ? 1u : p.nq - 4u * nfull) : 0u;  // (PEM_FOLD: just "has a last line here")
      // (PEM_FOLD: two octets behind the certificate belong to its chunks too — they are zeroed in ibuf for the last group)
      const uint64_t in_lo = c.lo + 48ull * ln_lo, in_end = last ? c.lo + L + (PEM_FOLD ? 2u : 0u) : c.lo + 48ull * fl_end;
      p.a0 = in_lo & ~15ull;
      p.nch = (uint32_t)((((in_end + 15ull) & ~15ull) - p.a0) >> 4);
    }
  }
  return p;
}

__device__ __forceinline__ uint32_t rl32(uint32_t v, uint32_t j) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)j); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, uint32_t j) {
  return (uint64_t)rl32((uint32_t)v, j) | ((uint64_t)rl32((uint32_t)(v >> 32), j) << 32);
}
__device__ __forceinline__ uint32_t gather32(uint32_t v, uint32_t from) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from << 2), (int)v); }

// Certificates [j0, j1) of a plan: hands every lane its input chunks (src[q]: payload offset of flat chunk 64·q + lane, ~0 =
// none), its lines and its task, and says whether one pass can take them (≤ 352 chunks, ≤ 128 lines, ≤ 16 certificates).  A
// scalar loop over the certificates (three counts each, v_readlane) tells every lane WHICH certificate owns its chunk, its
// line, its task; what it needs of that certificate it then fetches from the certificate's lane (ds_bpermute) — selecting
// every value inside the loop cost ≈ 40 vector instructions per certificate.
__device__ __forceinline__ bool pem_assign(const PemCert& c, const PemPlan& p, uint32_t j0, uint32_t j1, long long B0s, uint32_t lane,
                                           uint64_t (&src)[PEM_NQ], uint32_t& nflat, PemWork& w) {
  uint32_t cown[PEM_NQ], lown[PEM_LPL];  // owner lanes; 64 = none (a bpermute from lane 64 reads lane 0: unused)
#pragma unroll
  for (uint32_t q = 0; q < PEM_NQ; q++) cown[q] = 64u;
#pragma unroll
  for (uint32_t l = 0; l < PEM_LPL; l++) lown[l] = 64u;
  uint32_t my_cbase = 0u, my_lbase = 0u, cbase = 0u, lbase = 0u;
  bool any_task = false;
  for (uint32_t j = j0; j < j1; j++) {
    const uint32_t nch = rl32(p.nch, j), nfl = rl32(p.nfl, j);
    any_task = any_task | (rl32(p.ntk, j) != 0u);
#pragma unroll
    for (uint32_t q = 0; q < PEM_NQ; q++) cown[q] = (64u * q + lane - cbase) < nch ? j : cown[q];
#pragma unroll
    for (uint32_t l = 0; l < PEM_LPL; l++) lown[l] = (64u * l + lane - lbase) < nfl ? j : lown[l];
    my_cbase = lane == j ? cbase : my_cbase;
    my_lbase = lane == j ? lbase : my_lbase;
    cbase += nch;
    lbase += nfl;
  }
  // lane j's certificate: where its byte 0 lies in ibuf, its first base64 character in obuf
  const uint32_t my_ib = 16u * my_cbase + (uint32_t)(c.lo - p.a0);  // (wraps below zero when ln_lo > 0: 48·ln brings it back)
  const int32_t my_ob = (int32_t)((long long)(c.p0 + 28u) - B0s) + (int32_t)PEM_MARGIN;
#pragma unroll
  for (uint32_t q = 0; q < PEM_NQ; q++) {
    const uint32_t o = cown[q];
    const uint64_t a0 = (uint64_t)gather32((uint32_t)p.a0, o) | ((uint64_t)gather32((uint32_t)(p.a0 >> 32), o) << 32);
    // (every gather in a statement of its own: inside the conditional expression it runs under the mask "has an owner", and
    //  ds_bpermute returns 0 for a source lane that is masked off — the owner's lane, when the row's last chunks are few)
    const uint32_t ocb = gather32(my_cbase, o);
    src[q] = o < 64u ? a0 + 16ull * (64u * q + lane - ocb) : ~0ull;
  }
#pragma unroll
  for (uint32_t l = 0; l < PEM_LPL; l++) {
    const uint32_t o = lown[l];
    w.l_act[l] = o < 64u;
    w.ln[l] = gather32(p.ln_lo, o) + (64u * l + lane - gather32(my_lbase, o));
    w.l_ib[l] = gather32(my_ib, o);
    w.l_ob[l] = (int32_t)gather32((uint32_t)my_ob, o);
    w.l_len[l] = PEM_FOLD ? gather32((uint32_t)c.len, o) : 0u;
  }
  if constexpr (PEM_FOLD) {
    w.z_act = (lane >= j0) & (lane < j1) & (p.ntk != 0u);
    w.z_at = my_ib + (uint32_t)c.len;
    w.t_act = false;
    w.tk = w.t_ib = w.t_len = w.t_nq = 0u;
    w.t_ob = 0;
  } else {
    const uint32_t town = j0 + (lane >> 2);  // four task lanes per certificate
    const uint32_t ntk = gather32(p.ntk, town);
    w.t_act = (town < j1) & ((lane & 3u) < ntk);
    w.tk = gather32(p.tk_lo, town) + (lane & 3u);
    w.t_ib = gather32(my_ib, town);
    w.t_ob = (int32_t)gather32((uint32_t)my_ob, town);
    w.t_len = gather32((uint32_t)c.len, town);
    w.t_nq = gather32(p.nq, town);
    w.z_act = false;
    w.z_at = 0u;
  }
  w.any_task = any_task;
  nflat = cbase < PEM_IN_CHUNKS ? cbase : PEM_IN_CHUNKS;
  return cbase <= PEM_IN_CHUNKS && lbase <= 64u * PEM_LPL && j1 - j0 <= PEM_FAST_CERTS;
}

// aligned, non-temporal 16-byte loads, 1 KiB per instruction; v[q] = flat chunk 64·q + lane
__device__ __forceinline__ void pem_issue(const uint8_t* __restrict__ payload, const uint64_t (&src)[PEM_NQ], uint4 (&v)[PEM_NQ]) {
#pragma unroll
  for (uint32_t q = 0; q < PEM_NQ; q++)
    v[q] = src[q] != ~0ull ? ld_payload16((const uint4*)(payload + src[q])) : make_uint4(0, 0, 0, 0);
}

__device__ __forceinline__ void pem_park(uint8_t* ibuf, const uint4 (&v)[PEM_NQ], uint32_t nflat, uint32_t lane) {
#pragma unroll
  for (uint32_t q = 0; q < PEM_NQ; q++)
    if (64u * q < nflat && 64u * q + lane < PEM_IN_CHUNKS) ((uint4*)ibuf)[64u * q + lane] = v[q];
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t pem_enc4(const uint8_t* abc, uint32_t g) {  // a big-endian 24-bit group → four characters
  const uint32_t c0 = abc[__builtin_amdgcn_ubfe(g, 18, 6)], c1 = abc[__builtin_amdgcn_ubfe(g, 12, 6)],
                 c2 = abc[__builtin_amdgcn_ubfe(g, 6, 6)], c3 = abc[g & 63u];
  // two v_perm_b32 and an or (written as shifts and ors the compiler makes four to five instructions of it)
  return __builtin_amdgcn_perm(c1, c0, 0x0c0c0400u) | __builtin_amdgcn_perm(c3, c2, 0x04000c0cu);
}

// the whole lines: one lane, one line
__device__ __forceinline__ void pem_encode_line(bool act, uint32_t ln, uint32_t l_ib, int32_t l_ob, uint32_t l_len, const uint8_t* abc,
                                                uint8_t* obuf, const uint8_t* ibuf, uint32_t* trash) {
  if (!act) return;
  // PEM_FOLD: a certificate's last line may be shorter — rem octets, nch characters of which the last npad are '=', then the
  // line end.  The lane encodes 48 octets all the same (ibuf holds zeros where the last group needs them; what lies behind is
  // arbitrary) and writes 65 octets: behind the line end lie "-----END CERTIFICATE-----\n" and the next certificate's
  // "-----BEGIN CERTIFICATE-----\n", 54 octets that pem_frames writes AFTER the lines — only a line of 4 or 8 characters
  // would reach past them (into the next certificate's first line): its stores above byte 56 go nowhere.
  const uint32_t rem = PEM_FOLD ? l_len - 48u * ln : 48u;
  const bool shortl = rem < 48u;
  const uint32_t nch = shortl ? 4u * ((rem + 2u) / 3u) : 64u;
  const uint32_t npad = shortl ? (nch >> 2) * 3u - rem : 0u;
  const bool tiny = nch < 12u;
  const uint32_t at = l_ib + 48u * ln, m = at & 3u;
  const uint32_t* x = (const uint32_t*)(ibuf + (at & ~3u));
  uint32_t r[13];
#pragma unroll
  for (int k = 0; k < 13; k++) r[k] = x[k];
  uint32_t ch[17];
#pragma unroll
  for (int g = 0; g < 16; g += 4) {  // 12 input bytes = three dwords → four groups
    const int j = 3 * (g >> 2);
    const uint32_t d0 = __builtin_amdgcn_alignbyte(r[j + 1], r[j], m), d1 = __builtin_amdgcn_alignbyte(r[j + 2], r[j + 1], m),
                   d2 = __builtin_amdgcn_alignbyte(r[j + 3], r[j + 2], m);
    ch[g] = pem_enc4(abc, __builtin_amdgcn_perm(d0, d0, 0x0c000102u));
    ch[g + 1] = pem_enc4(abc, __builtin_amdgcn_perm(d1, d0, 0x0c030405u));
    ch[g + 2] = pem_enc4(abc, __builtin_amdgcn_perm(d2, d1, 0x0c020304u));
    ch[g + 3] = pem_enc4(abc, __builtin_amdgcn_perm(d2, d2, 0x0c010203u));
  }
  ch[16] = 0x0au;
  // the 64 characters and the line end at stream position P (body byte 65·ln): with s = P mod 4, aligned dword k of the
  // line is alignbyte(ch[k], ch[k − 1], 4 − s) — for s = 0 that is ch[k − 1], which belongs one dword earlier: the base moves
  const int32_t P = l_ob + (int32_t)(65u * ln);
  uint8_t* const o = obuf + P;
  const uint32_t s = (uint32_t)P & 3u, sh = (4u - s) & 3u;
  uint32_t* const A = (uint32_t*)(o - s - (s ? 0u : 4u));
#pragma unroll
  for (int k = 1; k < 15; k++) A[k] = __builtin_amdgcn_alignbyte(ch[k], ch[k - 1], sh);
  *((PEM_FOLD && tiny) ? trash : A + 15) = __builtin_amdgcn_alignbyte(ch[15], ch[14], sh);
  // dword 16 is a whole one for s = 0 (the last four characters) and s = 3 (three characters and the line end); for
  // s = 1, 2 its other bytes are the next line's — it goes nowhere, and the bytes below cover it
  *((((s == 0u) | (s == 3u)) & !(PEM_FOLD && tiny)) ? A + 16 : trash) = __builtin_amdgcn_alignbyte(ch[16], ch[15], sh);
  uint8_t* const trash_b = (uint8_t*)trash;
#pragma unroll
  for (uint32_t t4 = 0; t4 < 4u; t4++) {  // the first 4 − s characters in front, the last s behind (s = 0: the first four, again)
    const bool front = t4 < 4u - s;
    *(front ? o + t4 : ((PEM_FOLD && tiny) ? trash_b : o + 60u + t4)) = (uint8_t)((front ? ch[0] : ch[15]) >> (8u * t4));
  }
  if constexpr (PEM_FOLD) {  // (behind every other store of this lane: LDS operations of a wave execute in order)
    o[nch] = (uint8_t)'\n';  // nch = 64 for a whole line
    *(npad >= 1u ? o + nch - 1u : trash_b) = (uint8_t)'=';
    *(npad == 2u ? o + nch - 2u : trash_b) = (uint8_t)'=';
  } else {
    o[64] = (uint8_t)'\n';
  }
}
__device__ __forceinline__ void pem_encode_lines(const PemWork& w, const uint8_t* abc, uint8_t* obuf, const uint8_t* ibuf,
                                                 uint32_t* trash) {
#pragma unroll
  for (uint32_t l = 0; l < PEM_LPL; l++) pem_encode_line(w.l_act[l], w.ln[l], w.l_ib[l], w.l_ob[l], w.l_len[l], abc, obuf, ibuf, trash);
}
// PEM_FOLD: the two octets behind every certificate that ends in this pass become zero in ibuf (a partial last group is
// encoded with zero bits for what it lacks); behind pem_park, in front of the lines
__device__ __forceinline__ void pem_zero_tails(const PemWork& w, uint8_t* ibuf) {
  if (w.z_act) {
    ibuf[w.z_at] = 0;
    ibuf[w.z_at + 1u] = 0;
  }
}

// the shorter last lines: one lane, one 12-byte → 16-character task
__device__ __forceinline__ void pem_encode_tasks(const PemWork& w, const uint8_t* abc, uint8_t* obuf, const uint8_t* ibuf) {
  if (!w.t_act) return;
  const uint32_t k = w.tk;
  const uint32_t at = w.t_ib + 12u * k, m = at & 3u;
  const uint32_t* x = (const uint32_t*)(ibuf + (at & ~3u));
  const uint32_t w0 = x[0], w1 = x[1], w2 = x[2], w3 = x[3];
  uint32_t d0 = __builtin_amdgcn_alignbyte(w1, w0, m), d1 = __builtin_amdgcn_alignbyte(w2, w1, m), d2 = __builtin_amdgcn_alignbyte(w3, w2, m);
  const uint32_t left = w.t_len - 12u * k, nin = left < 12u ? left : 12u;
  if (nin < 12u) {  // the certificate's last, partial task: bytes behind the certificate count as zero
    const uint32_t keep = nin & 3u ? (1u << (8u * (nin & 3u))) - 1u : 0u;
    d0 = nin >= 4u ? d0 : (d0 & keep);
    d1 = nin >= 8u ? d1 : (nin > 4u ? (d1 & keep) : 0u);
    d2 = nin > 8u ? (d2 & keep) : 0u;
  }
  uint32_t ch[4] = {pem_enc4(abc, __builtin_amdgcn_perm(d0, d0, 0x0c000102u)), pem_enc4(abc, __builtin_amdgcn_perm(d1, d0, 0x0c030405u)),
                    pem_enc4(abc, __builtin_amdgcn_perm(d2, d1, 0x0c020304u)), pem_enc4(abc, __builtin_amdgcn_perm(d2, d2, 0x0c010203u))};
  bool nl = k == w.t_nq - 1u;  // the 17th byte: the line end (a last line has at most four tasks)
  if (nin < 12u) {  // '=' padding, and the line end right behind the last character
    const uint32_t gl = (nin - 1u) / 3u, rem = nin - 3u * gl;
    const uint32_t msk = rem == 1u ? 0x0000ffffu : rem == 2u ? 0x00ffffffu : 0xffffffffu;
    const uint32_t pad = rem == 1u ? 0x3d3d0000u : rem == 2u ? 0x3d000000u : 0u;
#pragma unroll
    for (uint32_t q = 0; q < 4u; q++) {
      ch[q] = q == gl ? ((ch[q] & msk) | pad) : ch[q];
      ch[q] = q == gl + 1u ? 0x0au : ch[q];
    }
    nl = gl == 3u;
  }
  const int32_t P = w.t_ob + (int32_t)((k >> 2) * 65u + (k & 3u) * 16u);
  uint8_t* const o = obuf + P;
  const uint32_t s = (uint32_t)P & 3u, sh = 4u - s;
  uint32_t* const q = (uint32_t*)(o - s);  // aligned; dwords 1..3 of the five the characters touch are whole
  q[1] = s ? __builtin_amdgcn_alignbyte(ch[1], ch[0], sh) : ch[1];
  q[2] = s ? __builtin_amdgcn_alignbyte(ch[2], ch[1], sh) : ch[2];
  q[3] = s ? __builtin_amdgcn_alignbyte(ch[3], ch[2], sh) : ch[3];
#pragma unroll
  for (uint32_t t4 = 0; t4 < 4u; t4++) {
    const bool front = t4 < sh;
    o[front ? t4 : 12u + t4] = (uint8_t)((front ? ch[0] : ch[3]) >> (8u * t4));
  }
  if (nl) o[16] = (uint8_t)'\n';
}

// the framing lines of certificates [j0, j1), behind the lines and tasks in program order (LDS operations of a wave
// execute in order): the tail of a partial last task overshoots into the END line's place, which is written here
__device__ __forceinline__ void pem_frames(const PemCert& c, uint32_t j0, uint32_t j1, long long B0s, const uint8_t* frame,
                                           uint8_t* obuf, uint32_t lane) {
  const bool hd = lane < 28u, tr = (lane >= 32u) & (lane < 58u);
  const uint8_t fb = frame[lane];
  for (uint32_t j = j0; j < j1; j++) {
    const long long pos = hd ? (long long)(rl64(c.p0, j) + lane) - B0s : (long long)(rl64(c.p1, j) - 26u + (lane - 32u)) - B0s;
    if ((hd | tr) && pos >= -(long long)PEM_MARGIN && pos < (long long)(PEM_S + PEM_MARGIN)) obuf[(int32_t)pos + (int32_t)PEM_MARGIN] = fb;
  }
}

__global__ void __launch_bounds__(64 * PEM_WAVES) k_pem_encode(const uint8_t* __restrict__ payload, const PemCertInfo* __restrict__ info,
                                                              uint64_t n_idx, const uint64_t* __restrict__ pem_off,
                                                              const uint32_t* __restrict__ block_first, uint64_t n_blocks,
                                                              uint64_t total, uint32_t a, uint8_t* __restrict__ out) {
  const uint32_t lane = threadIdx.x & 63u, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // static LDS (its address is a compile-time constant: a table behind the dynamic region's symbol cost one v_add per
  // character): the base64 alphabet, then 28 bytes "-----BEGIN CERTIFICATE-----\n" at +64 and 26 bytes
  // "-----END CERTIFICATE-----\n" at +96.  128 bytes: the dynamic region behind it stays 16-byte aligned.
  __shared__ __attribute__((aligned(16))) uint8_t pem_tab[128];
  uint8_t* const abc = pem_tab;
  uint8_t* const frame = pem_tab + 64;
  if (threadIdx.x < 64u) {
    abc[threadIdx.x] = (uint8_t)b64_char(threadIdx.x);
  } else if (threadIdx.x < 128u) {
    const char* const H = "-----BEGIN CERTIFICATE-----\n";
    const char* const T = "-----END CERTIFICATE-----\n";
    const uint32_t t = threadIdx.x - 64u;
    frame[t] = t < 28u ? (uint8_t)H[t] : (t >= 32u && t < 58u) ? (uint8_t)T[t - 32u] : (uint8_t)0;
  }
  __syncthreads();
  uint8_t* const obuf = smem + wv * PEM_WAVE_LDS;  // image of stream bytes [B0s − MARGIN, B0s + S + MARGIN)
  uint8_t* const ibuf = obuf + PEM_OBUF;
  uint32_t* const trash = (uint32_t*)(ibuf + PEM_IBUF) + lane;
  const uint64_t stride = (uint64_t)gridDim.x * PEM_WAVES;
  const auto bounds = [&](uint64_t b, long long& B0s, uint64_t& B0, uint64_t& B1) {
    B0s = (long long)(b * PEM_S) - (long long)a;  // out + B0s is 16-byte aligned
    B0 = B0s < 0 ? 0ull : (uint64_t)B0s;
    B1 = (uint64_t)(B0s + (long long)PEM_S) < total ? (uint64_t)(B0s + (long long)PEM_S) : total;
  };
  const auto first_of = [&](uint64_t blk) { return (uint64_t)block_first[blk < n_blocks ? blk : n_blocks - 1u]; };
  // A block's work hangs off a chain of dependent loads: block → first certificate → the certificates' bounds → their
  // bytes.  While block i is encoded, the BYTES of block i + 1 (its plan made from bounds that arrived an iteration ago) and
  // the BOUNDS of block i + 2 (from a first-certificate index fetched an iteration ago) are on their way.  A block that one
  // pass cannot serve (more than 16 certificates in it: tiny ones) is done certificate by certificate, outside the pipeline.
  uint64_t b = (uint64_t)blockIdx.x * PEM_WAVES + wv;
  if (b >= n_blocks) return;
  long long B0s;
  uint64_t B0, B1;
  bounds(b, B0s, B0, B1);
  uint64_t c_cur = first_of(b), c_nxt = first_of(b + stride), c_nx2 = first_of(b + 2u * stride);
  PemCert cc = pem_cert_load(pem_off, info, n_idx, c_cur + lane);
  PemPlan pc = pem_plan(cc, B0, B1);
  uint64_t src[PEM_NQ];
  uint4 v[PEM_NQ];
  uint32_t nflat = 0u;
  PemWork wk;
  bool simple = pc.ncert <= PEM_FAST_CERTS && pem_assign(cc, pc, 0u, pc.ncert, B0s, lane, src, nflat, wk);
  if (simple) pem_issue(payload, src, v);
  PemCert cn = pem_cert_load(pem_off, info, n_idx, c_nxt + lane);
  for (;;) {
    if (!simple) {  // certificate by certificate, in passes of 64 certificates
      uint64_t c_first = c_cur;
      for (;;) {
        for (uint32_t j = 0; j < pc.ncert; j++) {
          uint64_t s1[PEM_NQ];
          uint4 w1[PEM_NQ];
          uint32_t nf;
          PemWork w;
          (void)pem_assign(cc, pc, j, j + 1u, B0s, lane, s1, nf, w);  // (one certificate's share of a block always fits a pass)
          pem_issue(payload, s1, w1);
          pem_park(ibuf, w1, nf, lane);
          if (PEM_FOLD && w.any_task) pem_zero_tails(w, ibuf);
          pem_encode_lines(w, abc, obuf, ibuf, trash);
          if (!PEM_FOLD && w.any_task) pem_encode_tasks(w, abc, obuf, ibuf);
          pem_frames(cc, j, j + 1u, B0s, frame, obuf, lane);
          __builtin_amdgcn_wave_barrier();
        }
        if (pc.ncert < 64u) break;
        c_first += 64u;
        cc = pem_cert_load(pem_off, info, n_idx, c_first + lane);
        pc = pem_plan(cc, B0, B1);
      }
    } else {
      pem_park(ibuf, v, nflat, lane);
    }
    // ---- the bounds of the block after the next, then the next block's bytes: in flight while this block is encoded.  (In
    // this order: the bounds are copied into the loop's registers before the stores below are issued, and a wait for them
    // then leaves the byte loads — issued later — in flight.)
    const PemCert cn2 = pem_cert_load(pem_off, info, n_idx, c_nx2 + lane);  // (past the last block: the last block's again, unused)
    const uint64_t c_nx3 = first_of(b + 3u * stride);
    const bool has_next = b + stride < n_blocks;
    long long N0s = 0;
    uint64_t N0 = 0, N1 = 0;
    PemPlan pn = pc;
    PemWork wn = wk;
    bool simple_n = false;
    uint32_t nflat_n = 0u;
    if (has_next) {
      bounds(b + stride, N0s, N0, N1);
      pn = pem_plan(cn, N0, N1);
      simple_n = pn.ncert <= PEM_FAST_CERTS && pem_assign(cn, pn, 0u, pn.ncert, N0s, lane, src, nflat_n, wn);
      if (simple_n) pem_issue(payload, src, v);
    }
    // ---- this block: lines, last lines, framing lines; then the image leaves as aligned 16-byte vectors (only the first
    // and the last block of the stream have edges)
    if (simple) {
      if (PEM_FOLD && wk.any_task) pem_zero_tails(wk, ibuf);
      pem_encode_lines(wk, abc, obuf, ibuf, trash);
      if (!PEM_FOLD && wk.any_task) pem_encode_tasks(wk, abc, obuf, ibuf);
      pem_frames(cc, 0u, pc.ncert, B0s, frame, obuf, lane);
    }
    __builtin_amdgcn_wave_barrier();
    const long long S0s = B0s;  // (this block's place: the loop's registers move on to the next block before the stores)
    const uint64_t S0 = B0, S1 = B1;
    cc = cn; pc = pn; wk = wn; simple = simple_n; nflat = nflat_n;
    cn = cn2;
    c_cur = c_nxt; c_nxt = c_nx2; c_nx2 = c_nx3;
    B0s = N0s; B0 = N0; B1 = N1;
    if (S0s >= 0 && (uint64_t)S0s + PEM_S <= total) {  // an interior block: whole-line stores, nothing to decide
#pragma unroll
      for (uint32_t q = 0; q < (PEM_S + 1023u) / 1024u; q++)
        if (64u * (q + 1u) <= PEM_S / 16u || 64u * q + lane < PEM_S / 16u)
          st_stream16((uint4*)(out + S0s) + 64 * q + lane, *(const uint4*)(obuf + PEM_MARGIN + 16u * (64u * q + lane)));
    } else {
#pragma unroll 1
      for (int q = 0; q < (int)((PEM_S + 1023u) / 1024u); q++) {
        const long long ps = S0s + 16ll * (64 * q + (int)lane);
        if (64u * q + lane >= PEM_S / 16u) continue;
        const uint4 vv = *(const uint4*)(obuf + PEM_MARGIN + 16u * (64u * q + lane));
        if (ps >= (long long)S0 && (uint64_t)ps + 16u <= S1) {
          st_stream16((uint4*)(out + ps), vv);
        } else if (ps + 16 > (long long)S0 && ps < (long long)S1) {
          const uint32_t wds[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll 1
          for (int t = 0; t < 16; t++)
            if (ps + t >= (long long)S0 && ps + t < (long long)S1) out[ps + t] = (uint8_t)(wds[t >> 2] >> (8 * (t & 3)));
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (!has_next) break;
    b += stride;
  }
}

}  // namespace ctmr
