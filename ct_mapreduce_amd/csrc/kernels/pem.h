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

__global__ void __launch_bounds__(256) k_pem_len(const uint64_t* offsets, const uint64_t* ends, const uint64_t* idx,
                                                 uint64_t n_idx, uint64_t* pem_off) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r > n_idx) return;
  if (r == n_idx) {
    pem_off[r] = 0;  // the exclusive scan turns this slot into the total
    return;
  }
  uint64_t lo, hi;
  cert_range(offsets, ends, idx[r], lo, hi);
  pem_off[r] = pem_len(hi - lo);
}

struct __attribute__((packed, aligned(1))) U16 { uint32_t a, b, c, d; };  // unaligned 16-byte access (entries.h, meta.h)

// ------------------------------------------------------------------ k_pem_encode (round 5: output-block design)
// The PEM blocks of the NEW list are ONE contiguous byte stream (pem_off is an exclusive scan), so the work is cut by
// OUTPUT bytes, not by certificate: one wave produces one 4 KiB block of the stream at a time — whatever certificates
// and parts of certificates lie in it — and every global access is a naturally aligned 16-byte vector covering whole
// 128-byte lines:
//   1. lane j fetches the bounds of the j-th certificate that overlaps the block (k_pem_blocks left the first one's
//      index per block) and works out which of its 12-byte → 16-character tasks fall into the block;
//   2. the input bytes those tasks need (≤ 3 KiB, contiguous per certificate) come in as aligned, non-temporal
//      16-byte loads, 1 KiB per instruction, and are parked in LDS;
//   3. each lane encodes tasks: 12 bytes from LDS (aligned dword reads + v_alignbyte: the misalignment is the
//      certificate's start address mod 4), four v_perm_b32 make the big-endian 24-bit groups, the 16 characters come
//      from four pre-shifted 64-entry dword tables in LDS (one ds_read_b32 + a third of a v_or3 per character: the
//      compare/select alphabet cost ≈ 9 VALU instructions per character), and go to the block's image in LDS at
//      their final stream position — 65-byte lines put every line at another alignment, so the 16 characters are
//      stored as three aligned dwords + four single bytes (an unaligned ds_write_b128 is replayed at 64 cycles);
//      the line ends and the two framing lines are byte stores into the same image;
//   4. the image leaves as 4 × 64 aligned, non-temporal 16-byte stores.
// Round 4's kernel (one wave per certificate, unaligned dwordx3 loads, 16-byte stores at a 65-byte pitch that never
// meet a sector boundary, one byte store per line end) reached 3.4 TB/s read + written; see DESIGN.md §9 N1.
constexpr uint32_t PEM_S = 4096;              // output bytes per block
constexpr uint32_t PEM_MARGIN = 32;           // a task that straddles a block edge is encoded whole by both blocks
constexpr uint32_t PEM_IN_CHUNKS = 256;       // 16-byte input chunks parked per pass
constexpr uint32_t PEM_LUT_BYTES = 1024 + 64; // 4 × 64 dwords + the two framing lines
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

__global__ void __launch_bounds__(64 * PEM_WAVES) k_pem_encode(const uint8_t* payload, const uint64_t* offsets,
                                                              const uint64_t* ends, const uint64_t* idx, uint64_t n_idx,
                                                              const uint64_t* pem_off, const uint32_t* block_first,
                                                              uint64_t n_blocks, uint64_t total, uint32_t a, uint8_t* out) {
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  uint32_t* const lut = (uint32_t*)smem;
  uint8_t* const frame = smem + 1024;  // 28 bytes "-----BEGIN CERTIFICATE-----\n", then 26 bytes "-----END CERTIFICATE-----\n" at +32
  if (threadIdx.x < 64u) {
    const uint32_t c = b64_char(threadIdx.x);
    lut[threadIdx.x] = c;
    lut[64u + threadIdx.x] = c << 8;
    lut[128u + threadIdx.x] = c << 16;
    lut[192u + threadIdx.x] = c << 24;
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
  const uint64_t n_waves = (uint64_t)gridDim.x * PEM_WAVES;
  for (uint64_t b = (uint64_t)blockIdx.x * PEM_WAVES + wv; b < n_blocks; b += n_waves) {
    const long long B0s = (long long)(b * PEM_S) - (long long)a;           // out + B0s is 16-byte aligned
    const uint64_t B0 = B0s < 0 ? 0ull : (uint64_t)B0s;
    const uint64_t B1 = (uint64_t)(B0s + (long long)PEM_S) < total ? (uint64_t)(B0s + (long long)PEM_S) : total;
    uint64_t c_first = block_first[b];
    for (;;) {  // passes of up to 64 certificates (one pass unless the certificates are tiny)
      // ---- 1. lane j: the j-th certificate of the pass
      const uint64_t r = c_first + lane;
      uint64_t p0 = 0, p1 = 0, lo = 0, hi = 0;
      bool inblk = false;
      if (r < n_idx) {
        p0 = pem_off[r];
        inblk = p0 < B1;
        if (inblk) {
          p1 = pem_off[r + 1];
          cert_range(offsets, ends, idx[r], lo, hi);
        }
      }
      const uint32_t ncert = (uint32_t)__popcll(__ballot(inblk));  // p0 ascends: lanes 0 … ncert − 1
      if (ncert == 0u) break;
      const uint64_t L = hi - lo;
      const uint32_t nq = (uint32_t)((L + 11u) / 12u);
      const uint64_t body = p0 + 28u, body_end = p1 - 26u;  // base64 lines with their line ends
      uint32_t k_lo = 0u, ntask = 0u, nch = 0u;
      uint64_t a0 = 0;
      if (inblk) {
        const uint64_t s = B0 > body ? B0 : body, e = B1 < body_end ? B1 : body_end;
        if (e > s) {
          const uint32_t rl = (uint32_t)(s - body), rh = (uint32_t)(e - 1u - body);
          const uint32_t l0 = rl / 65u, c0 = rl - 65u * l0, l1 = rh / 65u, c1 = rh - 65u * l1;
          k_lo = 4u * l0 + (c0 >> 4 > 3u ? 3u : c0 >> 4);
          uint32_t k_hi = 4u * l1 + (c1 >> 4 > 3u ? 3u : c1 >> 4);
          k_lo = k_lo < nq ? k_lo : nq - 1u;
          k_hi = k_hi < nq ? k_hi : nq - 1u;
          ntask = k_hi - k_lo + 1u;
          const uint64_t in_lo = lo + 12ull * k_lo, in_end = 12ull * (k_hi + 1u) < L ? lo + 12ull * (k_hi + 1u) : lo + L;
          a0 = in_lo & ~15ull;
          nch = (uint32_t)((((in_end + 15ull) & ~15ull) - a0) >> 4);
        }
      }
      uint32_t pre = nch;  // inclusive prefix sum of the chunk counts
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(pre, d);
        pre += lane >= (uint32_t)d ? t : 0u;
      }
      uint32_t done = 0u;  // certificates of this pass already encoded
      while (done < ncert) {
        // ---- sub-pass: certificates [done, upto) whose chunks fit the input buffer together
        const uint32_t base = done ? __shfl(pre, (int)done - 1) : 0u;
        uint32_t upto = done + (uint32_t)__popcll(__ballot(inblk && lane >= done && pre - base <= PEM_IN_CHUNKS));
        upto = upto > done ? upto : done + 1u;  // (a single certificate never needs more than the buffer holds)
        // ---- 2. the input bytes: flat chunk f of the sub-pass belongs to the certificate whose prefix range holds it
        uint4 v[4];
        uint32_t nflat = __shfl(pre, (int)upto - 1) - base;
        nflat = nflat < PEM_IN_CHUNKS ? nflat : PEM_IN_CHUNKS;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t f = 64u * q + lane;
          uint64_t src = ~0ull;
          for (uint32_t j = done; j < upto; j++) {
            const uint32_t pj = __shfl(pre, (int)j) - base, nj = __shfl(nch, (int)j);
            const uint64_t aj = __shfl(a0, (int)j);
            if (f < pj && f >= pj - nj) src = aj + 16ull * (f - (pj - nj));
          }
          v[q] = (f < nflat && src != ~0ull) ? ld_payload16((const uint4*)(payload + src)) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (64u * q < nflat) ((uint4*)ibuf)[64u * q + lane] = v[q];
        __builtin_amdgcn_wave_barrier();
        // ---- 3. the tasks, certificate by certificate
        for (uint32_t j = done; j < upto; j++) {
          const uint32_t kj = __shfl(k_lo, (int)j), nt = __shfl(ntask, (int)j), nqj = __shfl(nq, (int)j);
          const uint64_t p0j = __shfl(p0, (int)j), p1j = __shfl(p1, (int)j), loj = __shfl(lo, (int)j), aj = __shfl(a0, (int)j);
          const uint64_t Lj = __shfl(L, (int)j);
          const uint32_t ib = 16u * (__shfl(pre, (int)j) - __shfl(nch, (int)j) - base) + (uint32_t)(loj - aj);  // ibuf offset of the certificate's byte 0 (wraps below zero when k_lo > 0: 12·k brings it back)
          const int32_t ob = (int32_t)((long long)(p0j + 28u) - B0s) + (int32_t)PEM_MARGIN;                 // obuf offset of body byte 0
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
            const auto enc = [&](uint32_t g) {
              return lut[g >> 18] | lut[64u + ((g >> 12) & 63u)] | lut[128u + ((g >> 6) & 63u)] | lut[192u + (g & 63u)];
            };
            uint32_t c[4] = {enc(g0), enc(g1), enc(g2), enc(g3)};
            bool nl = ((k & 3u) == 3u) | (k == nqj - 1u);  // the 17th byte: the line end
            if (nin < 12u) {  // '=' padding, and the line end right behind the last character
              const uint32_t gl = (nin - 1u) / 3u, rem = nin - 3u * gl;
              const uint32_t msk = rem == 1u ? 0x0000ffffu : rem == 2u ? 0x00ffffffu : 0xffffffffu;
              const uint32_t pad = rem == 1u ? 0x3d3d0000u : rem == 2u ? 0x3d000000u : 0u;
#pragma unroll
              for (uint32_t q = 0; q < 4u; q++) {
                c[q] = q == gl ? ((c[q] & msk) | pad) : c[q];
                c[q] = q == gl + 1u ? 0x0au : c[q];
              }
              nl = gl == 3u;
            }
            // the 16 characters at their stream position: body byte (k >> 2)·65 + (k & 3)·16
            const int32_t P = ob + (int32_t)((k >> 2) * 65u + (k & 3u) * 16u);
            uint8_t* const o = obuf + P;
            const uint32_t s = (uint32_t)P & 3u;
            uint32_t* const q = (uint32_t*)(o - s);  // aligned; dwords 1..3 of the five the characters touch are whole
            const uint32_t sh = 4u - s;               // (s = 0: the selects below take the unshifted dwords)
            q[1] = s ? __builtin_amdgcn_alignbyte(c[1], c[0], sh) : c[1];
            q[2] = s ? __builtin_amdgcn_alignbyte(c[2], c[1], sh) : c[2];
            q[3] = s ? __builtin_amdgcn_alignbyte(c[3], c[2], sh) : c[3];
            // the four bytes left over: c[0]'s first 4 − s in front, c[3]'s last s behind (s = 0: c[0] whole)
#pragma unroll
            for (uint32_t t4 = 0; t4 < 4u; t4++) {
              const bool front = t4 < sh;
              o[front ? t4 : 12u + t4] = (uint8_t)((front ? c[0] : c[3]) >> (8u * t4));
            }
            if (nl) o[16] = (uint8_t)'\n';
          }
          // the framing lines, behind the tasks in program order (LDS operations of a wave execute in order): the
          // tail of a partial last task overshoots into the END line's place, which is written here
          {
            const bool hd = lane < 28u, tr = (lane >= 32u) & (lane < 58u);
            const long long pos = hd ? (long long)(p0j + lane) - B0s : (long long)(p1j - 26u + (lane - 32u)) - B0s;
            if ((hd | tr) && pos >= -(long long)PEM_MARGIN && pos < (long long)(PEM_S + PEM_MARGIN))
              obuf[(int32_t)pos + (int32_t)PEM_MARGIN] = frame[lane];
          }
        }
        __builtin_amdgcn_wave_barrier();
        done = upto;
      }
      if (ncert < 64u) break;
      c_first += 64u;
    }
    // ---- 4. the block leaves: aligned 16-byte vectors; only the first and the last block of the stream have edges
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const long long ps = B0s + 16ll * (64 * q + (int)lane);
      const uint4 vv = *(const uint4*)(obuf + PEM_MARGIN + 16u * (64u * q + lane));
      if (ps >= (long long)B0 && (uint64_t)ps + 16u <= B1) {
        st_stream16((uint4*)(out + ps), vv);
      } else if (ps + 16 > (long long)B0 && ps < (long long)B1) {
        const uint32_t wds[4] = {vv.x, vv.y, vv.z, vv.w};
        for (int t = 0; t < 16; t++)
          if (ps + t >= (long long)B0 && ps + t < (long long)B1) out[ps + t] = (uint8_t)(wds[t >> 2] >> (8 * (t & 3)));
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace ctmr
