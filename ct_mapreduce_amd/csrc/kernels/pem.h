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
// OUTPUT bytes, not by certificate: one wave produces one 4 KiB block of the stream at a time — whatever certificates
// and parts of certificates lie in it — and every global access is a naturally aligned 16-byte vector covering whole
// 128-byte lines:
//   1. lane j fetches the bounds of the j-th certificate that overlaps the block (k_pem_blocks left the first one's
//      index per block, k_pem_len its place in the payload) and works out which of its 12-byte → 16-character tasks fall
//      into the block;
//   2. the input bytes those tasks need (≤ 3 KiB, contiguous per certificate) come in as aligned, non-temporal
//      16-byte loads, 1 KiB per instruction, and are parked in LDS;
//   3. each lane encodes tasks: 12 bytes from LDS (aligned dword reads + v_alignbyte: the misalignment is the
//      certificate's start address mod 4), four v_perm_b32 make the big-endian 24-bit groups, the 16 characters come
//      from the 64-byte alphabet in LDS (one ds_read_u8 per character, conflict-free: the compare/select alphabet cost
//      ≈ 9 VALU instructions per character), and go to the block's image in LDS at
//      their final stream position — 65-byte lines put every line at another alignment, so the 16 characters are
//      stored as three aligned dwords + four single bytes (an unaligned ds_write_b128 is replayed at 64 cycles);
//      the line ends and the two framing lines are byte stores into the same image;
//   4. the image leaves as 4 × 64 aligned, non-temporal 16-byte stores.
// Round 4's kernel (one wave per certificate, unaligned dwordx3 loads, 16-byte stores at a 65-byte pitch that never
// meet a sector boundary, one byte store per line end) reached 3.4 TB/s read + written; see DESIGN.md §9 N1.
constexpr uint32_t PEM_S = 4096;              // output bytes per block
constexpr uint32_t PEM_MARGIN = 32;           // a task that straddles a block edge is encoded whole by both blocks
constexpr uint32_t PEM_IN_CHUNKS = 256;       // 16-byte input chunks parked per pass
constexpr uint32_t PEM_LUT_BYTES = 64 + 64;   // the alphabet + the two framing lines
constexpr uint32_t PEM_OBUF = PEM_S + 2 * PEM_MARGIN;
constexpr uint32_t PEM_IBUF = PEM_IN_CHUNKS * 16 + 16;
constexpr uint32_t PEM_WAVE_LDS = PEM_OBUF + PEM_IBUF;
constexpr uint32_t PEM_WAVES = 4;             // waves per workgroup (they share the tables, nothing else)
constexpr uint32_t PEM_LDS_BYTES = PEM_LUT_BYTES + PEM_WAVES * PEM_WAVE_LDS;

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
// what lane j's certificate contributes to the block [B0, B1): its 12-byte → 16-character tasks k_lo … k_lo + ntask − 1, the
// nch 16-byte input chunks from payload offset a0 they read; pre = inclusive prefix sum of nch over the lanes
struct PemPlan {
  uint32_t k_lo, ntask, nch, pre, nq, ncert;
  uint64_t a0;
  bool inblk;
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

__device__ __forceinline__ PemPlan pem_plan(const PemCert& c, uint64_t B0, uint64_t B1, uint32_t lane) {
  PemPlan p{0u, 0u, 0u, 0u, (uint32_t)((c.len + 11u) / 12u), 0u, 0ull, c.have && c.p0 < B1};
  p.ncert = (uint32_t)__popcll(__ballot(p.inblk));  // p0 ascends: lanes 0 … ncert − 1
  if (p.inblk) {
    const uint64_t body = c.p0 + 28u, body_end = c.p1 - 26u;  // the base64 lines with their line ends
    const uint64_t s = B0 > body ? B0 : body, e = B1 < body_end ? B1 : body_end;
    if (e > s) {
      const uint32_t rl = (uint32_t)(s - body), rh = (uint32_t)(e - 1u - body);
      const uint32_t l0 = rl / 65u, c0 = rl - 65u * l0, l1 = rh / 65u, c1 = rh - 65u * l1;
      uint32_t k_lo = 4u * l0 + (c0 >> 4 > 3u ? 3u : c0 >> 4), k_hi = 4u * l1 + (c1 >> 4 > 3u ? 3u : c1 >> 4);
      k_lo = k_lo < p.nq ? k_lo : p.nq - 1u;
      k_hi = k_hi < p.nq ? k_hi : p.nq - 1u;
      p.k_lo = k_lo;
      p.ntask = k_hi - k_lo + 1u;
      const uint64_t in_lo = c.lo + 12ull * k_lo, in_end = 12ull * (k_hi + 1u) < c.len ? c.lo + 12ull * (k_hi + 1u) : c.lo + c.len;
      p.a0 = in_lo & ~15ull;
      p.nch = (uint32_t)((((in_end + 15ull) & ~15ull) - p.a0) >> 4);
    }
  }
  uint32_t pre = p.nch;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(pre, d);
    pre += lane >= (uint32_t)d ? t : 0u;
  }
  p.pre = pre;
  return p;
}

// the input bytes of the certificates [done, upto) of a plan: flat chunk f of them belongs to the certificate whose prefix
// range holds it.  Aligned, non-temporal 16-byte loads, 1 KiB per instruction; v[q] = chunk 64·q + lane.
__device__ __forceinline__ uint32_t pem_issue(const uint8_t* __restrict__ payload, const PemPlan& p, uint32_t done, uint32_t upto, uint32_t lane,
                                              uint4 (&v)[4]) {
  const uint32_t base = done ? __shfl(p.pre, (int)done - 1) : 0u;
  uint32_t nflat = __shfl(p.pre, (int)upto - 1) - base;
  nflat = nflat < PEM_IN_CHUNKS ? nflat : PEM_IN_CHUNKS;
  uint64_t src[4] = {~0ull, ~0ull, ~0ull, ~0ull};
  for (uint32_t j = done; j < upto; j++) {
    const uint32_t pj = __shfl(p.pre, (int)j) - base, nj = __shfl(p.nch, (int)j);
    const uint64_t aj = __shfl(p.a0, (int)j);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t f = 64u * q + lane;
      if (f < pj && f >= pj - nj) src[q] = aj + 16ull * (f - (pj - nj));
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++)
    v[q] = (64u * q + lane < nflat && src[q] != ~0ull) ? ld_payload16((const uint4*)(payload + src[q])) : make_uint4(0, 0, 0, 0);
  return nflat;
}

__device__ __forceinline__ void pem_park(uint8_t* ibuf, const uint4 (&v)[4], uint32_t nflat, uint32_t lane) {
#pragma unroll
  for (int q = 0; q < 4; q++)
    if (64u * q < nflat) ((uint4*)ibuf)[64u * q + lane] = v[q];
  __builtin_amdgcn_wave_barrier();
}

// the tasks and the framing lines of the certificates [done, upto) into the block's image
__device__ __forceinline__ void pem_encode_certs(const PemCert& c, const PemPlan& p, uint32_t done, uint32_t upto, long long B0s,
                                                 const uint8_t* abc, const uint8_t* frame, uint8_t* obuf, const uint8_t* ibuf,
                                                 uint32_t lane) {
  const uint32_t base = done ? __shfl(p.pre, (int)done - 1) : 0u;
  for (uint32_t j = done; j < upto; j++) {
    const uint32_t kj = __shfl(p.k_lo, (int)j), nt = __shfl(p.ntask, (int)j), nqj = __shfl(p.nq, (int)j);
    const uint64_t p0j = __shfl(c.p0, (int)j), p1j = __shfl(c.p1, (int)j), loj = __shfl(c.lo, (int)j), aj = __shfl(p.a0, (int)j);
    const uint64_t Lj = __shfl(c.len, (int)j);
    // ibuf offset of the certificate's byte 0 (wraps below zero when k_lo > 0: 12·k brings it back)
    const uint32_t ib = 16u * (__shfl(p.pre, (int)j) - __shfl(p.nch, (int)j) - base) + (uint32_t)(loj - aj);
    const int32_t ob = (int32_t)((long long)(p0j + 28u) - B0s) + (int32_t)PEM_MARGIN;  // obuf offset of body byte 0
    for (uint32_t t = lane; t < nt; t += 64u) {
      const uint32_t k = kj + t;
      const uint32_t at = ib + 12u * k;  // input byte 12·k of the certificate, in ibuf
      const uint32_t m = at & 3u;
      const uint32_t* w = (const uint32_t*)(ibuf + (at & ~3u));
      const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
      uint32_t d0 = __builtin_amdgcn_alignbyte(w1, w0, m), d1 = __builtin_amdgcn_alignbyte(w2, w1, m),
               d2 = __builtin_amdgcn_alignbyte(w3, w2, m);
      const uint64_t ip = 12ull * k;
      const uint32_t nin = (uint32_t)(Lj - ip < 12ull ? Lj - ip : 12ull);
      if (nin < 12u) {  // the certificate's last, partial task: bytes behind the certificate count as zero
        const uint32_t keep = nin & 3u ? (1u << (8u * (nin & 3u))) - 1u : 0u;
        d0 = nin >= 4u ? d0 : (d0 & keep);
        d1 = nin >= 8u ? d1 : (nin > 4u ? (d1 & keep) : 0u);
        d2 = nin > 8u ? (d2 & keep) : 0u;
      }
      // big-endian 24-bit groups, then 4 × 6 bits → the pre-shifted tables
      const uint32_t g0 = __builtin_amdgcn_perm(d0, d0, 0x0c000102u), g1 = __builtin_amdgcn_perm(d1, d0, 0x0c030405u),
                     g2 = __builtin_amdgcn_perm(d2, d1, 0x0c020304u), g3 = __builtin_amdgcn_perm(d2, d2, 0x0c010203u);
      // the 64-byte alphabet: sixteen dwords in sixteen banks, so any two lanes either read the same dword (a broadcast) or
      // different banks — four pre-shifted 64-DWORD tables put entries e and e + 32 on one bank, and 32 random lanes made
      // every read a three- to four-way conflict
      const auto enc = [&](uint32_t g) {
        return (uint32_t)abc[g >> 18] | ((uint32_t)abc[(g >> 12) & 63u] << 8) | ((uint32_t)abc[(g >> 6) & 63u] << 16) |
               ((uint32_t)abc[g & 63u] << 24);
      };
      uint32_t ch[4] = {enc(g0), enc(g1), enc(g2), enc(g3)};
      bool nl = ((k & 3u) == 3u) | (k == nqj - 1u);  // the 17th byte: the line end
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
      // the 16 characters at their stream position: body byte (k >> 2)·65 + (k & 3)·16
      const int32_t P = ob + (int32_t)((k >> 2) * 65u + (k & 3u) * 16u);
      uint8_t* const o = obuf + P;
      const uint32_t s = (uint32_t)P & 3u;
      uint32_t* const q = (uint32_t*)(o - s);  // aligned; dwords 1..3 of the five the characters touch are whole
      const uint32_t sh = 4u - s;               // (s = 0: the selects below take the unshifted dwords)
      q[1] = s ? __builtin_amdgcn_alignbyte(ch[1], ch[0], sh) : ch[1];
      q[2] = s ? __builtin_amdgcn_alignbyte(ch[2], ch[1], sh) : ch[2];
      q[3] = s ? __builtin_amdgcn_alignbyte(ch[3], ch[2], sh) : ch[3];
      // the four bytes left over: ch[0]'s first 4 − s in front, ch[3]'s last s behind (s = 0: ch[0] whole)
#pragma unroll
      for (uint32_t t4 = 0; t4 < 4u; t4++) {
        const bool front = t4 < sh;
        o[front ? t4 : 12u + t4] = (uint8_t)((front ? ch[0] : ch[3]) >> (8u * t4));
      }
      if (nl) o[16] = (uint8_t)'\n';
    }
    // the framing lines, behind the tasks in program order (LDS operations of a wave execute in order): the tail of a
    // partial last task overshoots into the END line's place, which is written here
    {
      const bool hd = lane < 28u, tr = (lane >= 32u) & (lane < 58u);
      const long long pos = hd ? (long long)(p0j + lane) - B0s : (long long)(p1j - 26u + (lane - 32u)) - B0s;
      if ((hd | tr) && pos >= -(long long)PEM_MARGIN && pos < (long long)(PEM_S + PEM_MARGIN))
        obuf[(int32_t)pos + (int32_t)PEM_MARGIN] = frame[lane];
    }
  }
}

__global__ void __launch_bounds__(64 * PEM_WAVES) k_pem_encode(const uint8_t* __restrict__ payload, const PemCertInfo* __restrict__ info,
                                                              uint64_t n_idx, const uint64_t* __restrict__ pem_off,
                                                              const uint32_t* __restrict__ block_first, uint64_t n_blocks,
                                                              uint64_t total, uint32_t a, uint8_t* __restrict__ out) {
  const uint32_t lane = threadIdx.x & 63u, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* const abc = smem;           // the base64 alphabet, 64 bytes
  uint8_t* const frame = smem + 64;    // 28 bytes "-----BEGIN CERTIFICATE-----\n", then 26 bytes "-----END CERTIFICATE-----\n" at +32
  if (threadIdx.x < 64u) {
    abc[threadIdx.x] = (uint8_t)b64_char(threadIdx.x);
  } else if (threadIdx.x < 128u) {
    const char* const H = "-----BEGIN CERTIFICATE-----\n";
    const char* const T = "-----END CERTIFICATE-----\n";
    const uint32_t t = threadIdx.x - 64u;
    if (t < 28u) frame[t] = (uint8_t)H[t];
    else if (t >= 32u && t < 58u) frame[t] = (uint8_t)T[t - 32u];
  }
  __syncthreads();
  uint8_t* const obuf = smem + PEM_LUT_BYTES + wv * PEM_WAVE_LDS;  // image of stream bytes [B0s − MARGIN, B0s + S + MARGIN)
  uint8_t* const ibuf = obuf + PEM_OBUF;
  const uint64_t stride = (uint64_t)gridDim.x * PEM_WAVES;
  const auto bounds = [&](uint64_t b, long long& B0s, uint64_t& B0, uint64_t& B1) {
    B0s = (long long)(b * PEM_S) - (long long)a;  // out + B0s is 16-byte aligned
    B0 = B0s < 0 ? 0ull : (uint64_t)B0s;
    B1 = (uint64_t)(B0s + (long long)PEM_S) < total ? (uint64_t)(B0s + (long long)PEM_S) : total;
  };
  // A block's work hangs off a chain of dependent loads: block → first certificate → the certificates' bounds → their
  // bytes.  With one block at a time per wave the kernel waited on that chain (2.9 TB/s read + written, round 5's first
  // build): the loop is software-pipelined — while block i is encoded, the BYTES of block i + 1 (its plan made from bounds
  // that arrived an iteration ago) and the BOUNDS of block i + 2 (from a first-certificate index fetched an iteration ago)
  // are on their way.  A block that one pass cannot serve (64 or more certificates in it, or more input chunks than the
  // buffer holds: tiny certificates) is done on the spot, outside the pipeline.
  uint64_t b = (uint64_t)blockIdx.x * PEM_WAVES + wv;
  if (b >= n_blocks) return;
  long long B0s;
  uint64_t B0, B1;
  bounds(b, B0s, B0, B1);
  const auto first_of = [&](uint64_t blk) { return (uint64_t)block_first[blk < n_blocks ? blk : n_blocks - 1u]; };
  uint64_t c_cur = first_of(b), c_nxt = first_of(b + stride), c_nx2 = first_of(b + 2u * stride);
  PemCert cc = pem_cert_load(pem_off, info, n_idx, c_cur + lane);
  PemPlan pc = pem_plan(cc, B0, B1, lane);
  bool simple = pc.ncert < 64u && __shfl(pc.pre, 63) <= PEM_IN_CHUNKS;
  uint4 v[4];
  uint32_t nflat = 0u;
  if (simple && pc.ncert) nflat = pem_issue(payload, pc, 0u, pc.ncert, lane, v);
  PemCert cn = pem_cert_load(pem_off, info, n_idx, c_nxt + lane);
  for (;;) {
    if (!simple) {  // passes of up to 64 certificates, sub-passes of up to PEM_IN_CHUNKS input chunks
      uint64_t c_first = c_cur;
      for (;;) {
        if (pc.ncert == 0u) break;
        uint32_t done = 0u;
        while (done < pc.ncert) {
          const uint32_t base = done ? __shfl(pc.pre, (int)done - 1) : 0u;
          uint32_t upto = done + (uint32_t)__popcll(__ballot(pc.inblk && lane >= done && pc.pre - base <= PEM_IN_CHUNKS));
          upto = upto > done ? upto : done + 1u;  // (a single certificate never needs more than the buffer holds)
          uint4 w[4];
          const uint32_t nf = pem_issue(payload, pc, done, upto, lane, w);
          pem_park(ibuf, w, nf, lane);
          pem_encode_certs(cc, pc, done, upto, B0s, abc, frame, obuf, ibuf, lane);
          __builtin_amdgcn_wave_barrier();
          done = upto;
        }
        if (pc.ncert < 64u) break;
        c_first += 64u;
        cc = pem_cert_load(pem_off, info, n_idx, c_first + lane);
        pc = pem_plan(cc, B0, B1, lane);
      }
    } else if (pc.ncert) {
      pem_park(ibuf, v, nflat, lane);
    }
    // ---- the bounds of the block after the next, then the next block's bytes: in flight while this block is encoded.  (In
    // this order: the bounds are copied into the loop's registers before the stores below are issued, and a wait for them
    // then leaves the four byte loads — issued later — in flight.)
    const PemCert cn2 = pem_cert_load(pem_off, info, n_idx, c_nx2 + lane);  // (past the last block: the last block's again, unused)
    const uint64_t c_nx3 = first_of(b + 3u * stride);
    const bool has_next = b + stride < n_blocks;
    long long N0s = 0;
    uint64_t N0 = 0, N1 = 0;
    PemPlan pn = pc;
    bool simple_n = false;
    uint32_t nflat_n = 0u;
    if (has_next) {
      bounds(b + stride, N0s, N0, N1);
      pn = pem_plan(cn, N0, N1, lane);
      simple_n = pn.ncert < 64u && __shfl(pn.pre, 63) <= PEM_IN_CHUNKS;
      if (simple_n && pn.ncert) nflat_n = pem_issue(payload, pn, 0u, pn.ncert, lane, v);
    }
    // ---- this block: the tasks, then the image leaves as aligned 16-byte vectors (only the first and the last block of the
    // stream have edges)
    if (simple && pc.ncert) pem_encode_certs(cc, pc, 0u, pc.ncert, B0s, abc, frame, obuf, ibuf, lane);
    __builtin_amdgcn_wave_barrier();
    const long long S0s = B0s;  // (this block's place: the loop's registers move on to the next block before the stores)
    const uint64_t S0 = B0, S1 = B1;
    cc = cn; pc = pn; simple = simple_n; nflat = nflat_n;
    cn = cn2;
    c_cur = c_nxt; c_nxt = c_nx2; c_nx2 = c_nx3;
    B0s = N0s; B0 = N0; B1 = N1;
    if (S0s >= 0 && (uint64_t)S0s + PEM_S <= total) {  // an interior block: four whole-line stores, nothing to decide
#pragma unroll
      for (int q = 0; q < 4; q++)
        st_stream16((uint4*)(out + S0s) + 64 * q + lane, *(const uint4*)(obuf + PEM_MARGIN + 16u * (64u * q + lane)));
    } else {
#pragma unroll 1
      for (int q = 0; q < 4; q++) {
        const long long ps = S0s + 16ll * (64 * q + (int)lane);
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
