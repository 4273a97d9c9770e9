// kernels/reduce.h — the reduce: known-certificate table insert (fused into the map kernel by default), WasUnknown, per-issuer counts, compaction of the NEW list.
// gfx950 (CDNA4, wave64) only; part of kernels.h, which includes the pieces in dependency order.
#pragma once
#include "keyrec.h"
#include "map.h"

namespace ctmr {

// ------------------------------------------------------------------ the reduce
#define AGENT __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, AGENT);
}

// The key of a cell against (meta, s): the SHADOW bit is not part of the key.
__device__ __forceinline__ bool cell_equals(const KeyCell* c, unsigned long long meta, const unsigned long long s[5]) {
  bool eq = (ld_agent(&c->meta) & ~CELL_SHADOW) == meta;
#pragma unroll
  for (int k = 0; k < 5; k++) eq = eq && ld_agent(&c->s[k]) == s[k];
  return eq;
}
__device__ __forceinline__ void cell_write(KeyCell* c, unsigned long long meta, const unsigned long long s[5], uint32_t ord) {
  uint4* q = (uint4*)c;
  q[0] = make_uint4((uint32_t)meta, (uint32_t)(meta >> 32), (uint32_t)s[0], (uint32_t)(s[0] >> 32));
  q[1] = make_uint4((uint32_t)s[1], (uint32_t)(s[1] >> 32), (uint32_t)s[2], (uint32_t)(s[2] >> 32));
  q[2] = make_uint4((uint32_t)s[3], (uint32_t)(s[3] >> 32), (uint32_t)s[4], (uint32_t)(s[4] >> 32));
  q[3] = make_uint4(ord, 0u, 0u, 0u);
}

// Find-or-insert one key with the fully synchronised protocol (point operations, pass 2's rare fallback): returns the
// index slot (or SID_NONE / SID_FULL) and, through *holder, the word found there.  insert: the caller has WRITTEN
// arena[my_ref] already and made it visible (the word is what publishes the cell) — *created tells whether this call
// claimed the slot.  Every word met on the way points to a complete cell: pass-1 cells are complete behind the kernel
// boundary, and a word published here was preceded by its cell.
__device__ __forceinline__ uint32_t index_upsert(const Table& t, unsigned long long meta, const unsigned long long s[5],
                                                 unsigned long long my_ref, bool insert, bool* created,
                                                 unsigned long long* holder = nullptr) {
  const unsigned long long h = key_hash(meta, s);
  const unsigned long long mine = idx_word(h, my_ref);
  uint64_t j = h & t.mask;
  *created = false;
  for (uint64_t probes = 0; probes <= t.mask; probes++) {
    unsigned long long w = ld_agent(&t.index[j]);
    if (w == 0ull) {
      if (!insert) return SID_NONE;
      const unsigned long long old = atomicCAS(&t.index[j], 0ull, mine);
      if (old == 0ull) {
        *created = true;
        if (holder) *holder = mine;
        return (uint32_t)j;
      }
      w = old;
    }
    if (w != IDX_TOMB && idx_same_tag(w, h) && cell_equals(t.arena + (w & REF_MASK), meta, s)) {
      if (holder) *holder = w;
      return (uint32_t)j;
    }
    j = probe_next(j, probes, t.mask);
  }
  return SID_FULL;
}

struct InsertArgs {
  const ctmr_record* records;
  const uint8_t* payload;  // for serials longer than the 20 octets a record carries
  const uint64_t* offsets;
  const uint64_t* ends;    // null = packed batch (MapArgs)
  const uint32_t* canon;   // issuer_idx → canonical issuer
  Table t;                 // the known-certificate table: index words + key cells (ctmr_dev.h)
  unsigned long long ref0; // arena cell of entry 0 of this batch: entry i's key lives in arena[ref0 + i]; a word whose ref
                           // is below ref0 belongs to an earlier round (complete, comparable at once)
  uint32_t* slot_id;       // candidate slot of DEFER entries (written for those only)
  uint32_t* ent;           // per entry: status(0..2) | state(3..5) | canonical issuer << 8
  uint64_t n;
  DevStats* stats;         // n_xl (owner-computes rounds), n_pending
  uint32_t ord_base;       // order of entry 0 in the round (keyrec.h; 0 outside a group round): w[0] carries ord_base + i
  // k_ec_resolve (strict_spki): where the pending EC points lie, the owner-computes staging array (a key that left for its
  // owner before its point was checked is withdrawn there) and the rank's Bloom filter (null: none)
  const unsigned long long* keypos;
  KeyRec32* stage;
  unsigned long long* bloom;
  uint64_t bloom_wmask;
};

// Per-entry state of the reduce (bits 3..5 of ent[i]); the low 3 bits carry record.status.
enum : uint32_t {
  ES_NONE = 0,     // did not reach the set (filtered / parse error / no issuer)
  ES_CLAIMED = 1,  // claimed an empty slot: WasUnknown unless a lower log index of the same key marks it
  ES_DEFER = 2,    // met a same-tag slot of this batch: decided in pass 2; WasUnknown unless marked
  ES_DUP = 3,      // known: since an earlier batch, or a lower log index of this batch holds the key
  ES_HOST = 4,     // serial longer than CTMR_MAX_SERIAL: exact host-side set
  ES_FULL = 5,     // table full
  ES_REMOTE = 6,   // owner-computes round: the key belongs to another rank and left as a key record; WasUnknown unless
                   // the owner says otherwise (apply clears it) — in the NEW list, NOT in this rank's per-issuer counts
  ES_PENDING = 7   // strict_spki: reached the set, the key is this rank's, but its EC point owes the curve equation: NOT
                   // inserted yet — k_ec_resolve checks the point and inserts (no entry leaves round_begin in this state)
};
// ent[] bit 7: the entry's EC key owes the curve equation (any status, any state); k_ec_resolve clears it
constexpr uint32_t ENT_KEY_PENDING = 0x80u;
__device__ __forceinline__ uint32_t ent_pack(uint32_t status, uint32_t state, uint32_t canon) {
  return (status & 7u) | (state << 3) | (canon << 8);
}
__device__ __forceinline__ uint32_t ent_state(uint32_t e) { return (e >> 3) & 7u; }
__device__ __forceinline__ bool ent_is_new(uint32_t e) {
  const uint32_t st = ent_state(e);
  return (st == ES_CLAIMED) | (st == ES_DEFER) | (st == ES_REMOTE);
}
// a PASS entry lost to a lower log index of the same key: its ent byte 0 and its record flag
__device__ __forceinline__ void mark_dup(uint32_t* ent, ctmr_record* records, uint32_t loser) {
  ((uint8_t*)(ent + loser))[0] = (uint8_t)(CTMR_ST_PASS | (ES_DUP << 3));
  uint8_t* fl = (uint8_t*)(records + loser) + 1;
  *fl = (uint8_t)(*fl & ~CTMR_FL_WAS_UNKNOWN);
}

// Offset of the serialNumber content octets (certificate already accepted by the map): behind the INNER element of
// the optional [0] version wrapper, as walk_cert resumes there.
__device__ __forceinline__ uint32_t serial_content_off(const GlobalReader& r, uint32_t L) {
  bool ok = true;
  uint32_t tag, cs, ce;
  rd_hdr(r, L, 0, L, ok, tag, cs, ce);
  rd_hdr(r, L, cs, L, ok, tag, cs, ce);
  uint32_t q = cs;
  if ((r.ld4(q) & 0xffu) == 0xa0u) {
    rd_hdr<false>(r, L, q, L, ok, tag, cs, ce);
    rd_hdr(r, L, cs, L, ok, tag, cs, ce);
    q = ce;
  }
  rd_hdr(r, L, q, L, ok, tag, cs, ce);
  return cs;
}

__device__ __forceinline__ void record_key(const InsertArgs& a, uint64_t i, const uint4& r0,
                                           const uint4& r1, unsigned long long s[5]) {
  const uint32_t slen = r0.x >> 16;
  s[0] = (unsigned long long)r0.w | ((unsigned long long)r1.x << 32);
  s[1] = (unsigned long long)r1.y | ((unsigned long long)r1.z << 32);
  s[2] = (unsigned long long)r1.w;
  s[3] = 0;
  s[4] = 0;
  if (slen > 20) {
    // octets 20..slen-1 come from the certificate itself
    uint64_t lo, hi;
    cert_range(a.offsets, a.ends, i, lo, hi);
    GlobalReader g{(const uint32_t*)a.payload, lo};
    const uint32_t so = serial_content_off(g, (uint32_t)(hi - lo));
    uint32_t x[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const uint32_t pos = 20u + 4u * k;
      if (pos < slen) {
        const uint32_t rem = slen - pos;
        const uint32_t v = g.ld4(so + pos);
        x[k] = rem >= 4 ? v : (v & (0xffffffffu >> (8 * (4 - rem))));
      }
    }
    s[2] |= (unsigned long long)x[0] << 32;
    s[3] = (unsigned long long)x[1] | ((unsigned long long)x[2] << 32);
    s[4] = (unsigned long long)x[3] | ((unsigned long long)x[4] << 32);
  }
}

// KnownCertificates.WasUnknown → RemoteCache.SetInsert (knowncertificates.go:38-55) for every PASS entry, against the
// in-HBM table.  PASS 1 never reads anything another lane of the same launch wrote except the index word itself:
//   empty slot      → one atomicCAS claims it for arena[ref0 + i] (state CLAIMED)
//   same tag, word of an EARLIER round (ref < ref0) → its cell is complete: compare now (state DUP, or probe on)
//   same tag, word of THIS round → remember the slot (state DEFER), decide in pass 2 behind the kernel boundary
// and every entry's key goes into its own cell, arena[ref0 + i] — whether it claimed or not (a DEFER entry may have to
// insert itself in pass 2, and a word may be moved to any presenter's cell) — the 64 cells of a wave as one contiguous
// 4 KiB store (store_cells_wave).  Records arrive with WAS_UNKNOWN set optimistically by the map; it is cleared here /
// in pass 2 for duplicates only.  The reduce's later passes read the 4-byte ent[] word, never the table.
__device__ __forceinline__ uint32_t insert_probe_h(const InsertArgs& a, uint64_t i, unsigned long long meta,
                                                   const unsigned long long s[5], unsigned long long h) {
  const unsigned long long mine = idx_word(h, a.ref0 + i);
  uint64_t j = h & a.t.mask;
  for (uint64_t probes = 0; probes <= a.t.mask; probes++) {
    const unsigned long long old = atomicCAS(&a.t.index[j], 0ull, mine);
    if (old == 0ull) return ES_CLAIMED;
    if (old != IDX_TOMB && idx_same_tag(old, h)) {
      const unsigned long long oref = old & REF_MASK;
      if (oref < a.ref0) {  // an earlier round's key: complete and visible
        const KeyCell* c = a.t.arena + oref;
        bool eq = (c->meta & ~CELL_SHADOW) == meta;
#pragma unroll
        for (int k = 0; k < 5; k++) eq = eq && c->s[k] == s[k];
        if (eq) return ES_DUP;
      } else {
        a.slot_id[i] = (uint32_t)j;
        return ES_DEFER;
      }
    }
    j = probe_next(j, probes, a.t.mask);
  }
  return ES_FULL;
}

__device__ __forceinline__ uint32_t insert_probe(const InsertArgs& a, uint64_t i, const uint4& r0, const uint4& r1,
                                                 uint32_t canon, unsigned long long& meta, unsigned long long s[5]) {
  const uint32_t slen = r0.x >> 16;
  meta = 0ull;
  if (slen > CTMR_MAX_SERIAL) return ES_HOST;
  record_key(a, i, r0, r1, s);
  meta = key_meta((int32_t)r0.y, canon, slen);
  return insert_probe_h(a, i, meta, s, key_hash(meta, s));
}

// The key cells of one wave's 64 consecutive entries, arena[ref0 + first … + 64): every lane parks its 64-byte cell in
// LDS, then each of four store instructions writes 1 KiB of consecutive bytes.  `keyed` = the lane has a key (a PASS
// entry with a serial of at most 40 octets); the others write a cell without VALID.
__device__ __forceinline__ void store_cells_wave(KeyCell* cells, uint64_t n_left, uint4* img, uint32_t lane, bool keyed,
                                                 unsigned long long meta, const unsigned long long s[5], uint32_t ord) {
  uint4* my = img + lane * 4;
  const unsigned long long m = keyed ? meta : 0ull;
  my[0] = make_uint4((uint32_t)m, (uint32_t)(m >> 32), (uint32_t)s[0], (uint32_t)(s[0] >> 32));
  my[1] = make_uint4((uint32_t)s[1], (uint32_t)(s[1] >> 32), (uint32_t)s[2], (uint32_t)(s[2] >> 32));
  my[2] = make_uint4((uint32_t)s[3], (uint32_t)(s[3] >> 32), (uint32_t)s[4], (uint32_t)(s[4] >> 32));
  my[3] = make_uint4(ord, 0u, 0u, 0u);
  __builtin_amdgcn_wave_barrier();
  const uint32_t nvec = n_left >= 64 ? 256u : (uint32_t)n_left * 4u;
  uint4* out = (uint4*)cells;
#pragma unroll
  for (int r = 0; r < 4; r++)
    if (64u * r + lane < nvec) out[64 * r + lane] = img[64 * r + lane];
}

__global__ void __launch_bounds__(256) k_insert(InsertArgs a) {
  __shared__ __attribute__((aligned(16))) uint4 img[4][64 * 4];  // per wave: 64 key cells
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t state = ES_NONE, status = CTMR_ST__COUNT, canon = 0;
  unsigned long long meta = 0ull, s[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
  bool keyed = false;
  if (i < a.n) {
    const uint4* rp = (const uint4*)(a.records + i);
    const uint4 r0 = rp[0];
    status = r0.x & 0xffu;
    // key pending (the map's internal record flag): the flag moves into ent[], and an entry that would be inserted waits
    const bool pending = ((r0.x >> 8) & FL_KEY_PENDING) != 0u;
    uint32_t fl_out = (r0.x >> 8) & 0xffu & ~FL_KEY_PENDING;
    if (status == CTMR_ST_PASS) {
      canon = a.canon[r0.z];
      const uint4 r1 = rp[1];
      const uint32_t slen = r0.x >> 16;
      if (pending && slen <= CTMR_MAX_SERIAL) {
        state = ES_PENDING;
        record_key(a, i, r0, r1, s);
        meta = key_meta((int32_t)r0.y, canon, slen);
      } else {
        state = insert_probe(a, i, r0, r1, canon, meta, s);
      }
      keyed = slen <= CTMR_MAX_SERIAL;
      if (state != ES_CLAIMED && state != ES_DEFER && state != ES_PENDING) fl_out &= ~(uint32_t)CTMR_FL_WAS_UNKNOWN;  // not (yet) unknown
    }
    if (fl_out != ((r0.x >> 8) & 0xffu)) ((uint8_t*)(a.records + i))[1] = (uint8_t)fl_out;
    a.ent[i] = ent_pack(status, state, canon) | (pending ? ENT_KEY_PENDING : 0u);
    const unsigned long long mp = __ballot(pending);
    if (mp && lane == (uint32_t)(__ffsll((long long)mp) - 1) && ld_agent(&a.stats->n_pending) == 0ull) {
      st_agent(&a.stats->n_pending, 1ull);  // a flag, not a count (see k_map_fused)
      st_agent(&a.stats->n_pending_other, 1ull);  // (this path does not look at the curve: both resolve kernels take a look)
    }
  }
  const uint64_t first = (uint64_t)blockIdx.x * 256 + 64u * wv;
  if (first < a.n)
    store_cells_wave(a.t.arena + a.ref0 + first, a.n - first, img[wv], lane, keyed, meta, s, a.ord_base + (uint32_t)i);
}

// PASS 2: DEFER entries — their candidate word was published by this round, and every cell of the round is complete now.
// The word's cell holds the slot's key (whichever presenter's cell it points to: the word only ever moves between
// presenters of ONE key).  Same key → the lowest ORDER must end up holding the word: if the holder's order is lower I
// lose (DUP); else I move the word to my cell with a compare-and-swap and the former holder loses — when it is an entry
// of THIS shard its ent[] word and its record say so (mark_dup_ord); an entry of another rank (owner-computes round) is
// told by its owner's answer, which asks "does the word still point to MY cell?" (k_keys_resolve).  Another key under
// the same 24-bit tag (2^-24 per probe): the entry inserts itself with the synchronised probe — its cell is there already.
__device__ __forceinline__ void mark_dup_ord(const InsertArgs& a, ctmr_record* records, uint32_t loser_ord) {
  const uint32_t li = loser_ord - a.ord_base;  // wraps for lower ranks' orders: then >= n as well
  if ((uint64_t)li < a.n) mark_dup(a.ent, records, li);
}

// Settles one presenter (key, order `ord`, cell `my_ref`) against the word at slot sid.  Returns true when the presenter
// keeps (so far) WasUnknown; losers of this shard are marked on the way.
__device__ __forceinline__ bool settle_order(const InsertArgs& a, ctmr_record* records, uint32_t sid, unsigned long long h,
                                             unsigned long long my_ref, uint32_t ord) {
  unsigned long long w = ld_agent(&a.t.index[sid]);
  for (;;) {
    const unsigned long long href = w & REF_MASK;
    if (href == my_ref) return true;
    if (href < a.ref0) return false;  // (cannot happen for a DEFER slot; a point insert of an earlier round: known)
    const uint32_t hord = __hip_atomic_load(&a.t.arena[href].ord, __ATOMIC_RELAXED, AGENT);
    if (hord < ord) return false;     // a lower order of this round holds the key
    const unsigned long long old = atomicCAS(&a.t.index[sid], w, idx_word(h, my_ref));
    if (old == w) {
      mark_dup_ord(a, records, hord);
      return true;
    }
    w = old;                          // somebody else moved it meanwhile: look again
  }
}

__device__ __forceinline__ void insert2_one(const InsertArgs& a, ctmr_record* records, uint64_t i, uint32_t e) {
  uint32_t sid = a.slot_id[i];
  const uint4* rp = (const uint4*)(a.records + i);
  const uint4 r0 = rp[0], r1 = rp[1];
  unsigned long long s[5];
  record_key(a, i, r0, r1, s);
  const unsigned long long meta = key_meta((int32_t)r0.y, e >> 8, r0.x >> 16);
  const unsigned long long h = key_hash(meta, s);
  const uint32_t ord = a.ord_base + (uint32_t)i;
  const unsigned long long my_ref = a.ref0 + i;
  const unsigned long long w = ld_agent(&a.t.index[sid]);
  if (!cell_equals(a.t.arena + (w & REF_MASK), meta, s)) {  // another key under the same tag
    bool created;
    unsigned long long holder = 0ull;
    sid = index_upsert(a.t, meta, s, my_ref, true, &created, &holder);
    if (sid == SID_FULL) {
      ((uint8_t*)(a.ent + i))[0] = (uint8_t)(CTMR_ST_PASS | (ES_FULL << 3));
      return;
    }
    if (created) return;  // stays DEFER = unknown unless a lower order joins and marks it
    if ((holder & REF_MASK) < a.ref0) {  // found in an earlier round after all
      mark_dup(a.ent, records, (uint32_t)i);
      return;
    }
  }
  if (!settle_order(a, records, sid, h, my_ref, ord)) mark_dup(a.ent, records, (uint32_t)i);
}

// Four entries per thread, one 16-byte load of ent[]: nearly every entry is not DEFER, so the kernel is a scan of
// ent[] — with one entry per thread it was bound by launching 1.5 M near-empty waves per 100 M entries (1.4 ms).
constexpr uint32_t INSERT2_PER_BLOCK = 1024;
__global__ void __launch_bounds__(256) k_insert2(InsertArgs a, ctmr_record* records) {
  const uint64_t i0 = (uint64_t)blockIdx.x * INSERT2_PER_BLOCK + threadIdx.x * 4u;
  if (i0 >= a.n) return;
  uint32_t e[4] = {0u, 0u, 0u, 0u};
  if (i0 + 4 <= a.n) {
    const uint4 v = *(const uint4*)(a.ent + i0);
    e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w;
  } else {
    for (uint32_t k = 0; i0 + k < a.n; k++) e[k] = a.ent[i0 + k];
  }
  // A DEFER entry costs a chain of five dependent random accesses (slot id, record, index word, cell, the settling CAS),
  // and there are a handful per wave: each thread takes ITS first DEFER entry, whichever of its four that is, so that the
  // wave runs the chain once for all of them — with one pass per k it ran the chain for every k some lane had one at
  // (0.93 → … ms per 100 M entries with 2 % duplicates, round 4).
  uint32_t todo = (ent_state(e[0]) == ES_DEFER ? 1u : 0u) | (ent_state(e[1]) == ES_DEFER ? 2u : 0u) |
                  (ent_state(e[2]) == ES_DEFER ? 4u : 0u) | (ent_state(e[3]) == ES_DEFER ? 8u : 0u);
  while (todo) {
    const uint32_t k = (uint32_t)__builtin_ctz(todo);
    todo &= todo - 1u;
    const uint32_t ek = k == 0u ? e[0] : k == 1u ? e[1] : k == 2u ? e[2] : e[3];
    insert2_one(a, records, i0 + k, ek);
  }
}

// Fused map + pass-1 insert (the default, variant 15): the lane that just finished walking a certificate probes the
// known-certificate table straight from its registers — the 32-byte record is not re-read (−3.2 GB per
// 100 M entries), the WAS_UNKNOWN flag is final before the record is stored (no second scattered write for
// old-batch duplicates) and the random-access latency of the CAS hides behind the walks of the other
// waves of the CU instead of being a kernel of its own.  Pass 2 (k_insert2) is unchanged.
// Every ld4 of the walk is served by the LDS window alone (WinReaderS); a certificate whose walk leaves the window
// is repeated with the exact global reader.
// (Tried and dropped: loading the slot's claim word early, when the key is known but the extension block
// is still in flight, so that the CAS finds the line on-die — +0.6 ms at 100 M entries: the kernel is bound
// by memory transactions, not by the latency of the probe.  An XCD-contiguous workgroup → block mapping: 23.87 ms
// against 23.44 ms, profiles/r01/s5/sweep_xcd_contiguous_blocks_not_default.txt.  The same kernel with the per-access
// global fallback inside ld4 instead of the window-only reader: 0.9–2.5 ms slower, profiles/r01/s4.)
// META (engines created with collect_meta, round 2): the memo of EARLIER ctmr_meta_new calls is looked up while the
// bytes are in the LDS window anyway — the issuer Name from the front window (reader hook, right behind the Name), the
// CRL distribution point from the extension window and the (issuer, expDate hour) bit at the end of the walk — and
// ent[] bit 6 is set only for certificates that may bring a first sighting.  k_meta_new then skips everything else
// instead of re-reading ≈ 4 lines of every new certificate (9.7 ms per 94 M new certificates in round 1).
// MODE (round 3: the global-dedup modes keep the fused kernel — keyrec.h):
//   XM_OWNER  owner-computes exchange: a key owned by THIS rank is inserted here as always; a key owned by another rank is
//             not probed at all — the walking lane has the key in registers and writes it out as a 32-byte record
//             (state ES_REMOTE).  The records of a wave are compacted by owner with W ballots into the wave's own 2 KiB of
//             the staging array (no atomics, no counters shared between waves); k_key_blockcount / k_key_gather turn the
//             staging array into per-owner partitions in log order.
//   XM_BLOOM  Bloom variant: the key of every entry that claimed or may claim a slot (CLAIMED / DEFER) is added to the
//             rank's cumulative filter with one fire-and-forget 8-byte atomicOr — no separate pass over the records.
template <int WCH, bool META, int MODE, bool STRICT = false>
__global__ void __launch_bounds__(64) k_map_fused(MapArgs a, InsertArgs ia, MetaCheck mc, XchgArgs xa) {
  const uint64_t first = (uint64_t)blockIdx.x * 64;
  const uint32_t lane = threadIdx.x;
  const uint64_t i = first + lane;
  const bool live = i < a.n;
  const uint64_t limit = map_limit(a);
  uint64_t lo = 0, hi = 0;
  if (live) cert_range(a.offsets, a.ends, i, lo, hi);
  // the entry's other inputs and what hangs off them (issuer_idx → issuer_valid, canonical index): issued now, they
  // return while the certificate's bytes are on their way, instead of costing dependent round trips behind the walk
  EntryIn in{CTMR_NO_ISSUER, 0u, false, false};
  uint32_t canon = 0;
  if (live) {
    in = load_entry_in(a, i);
    canon = in.iss_in_range ? ia.canon[in.iss] : 0u;
  }
  // META: what the pre-check needs of this issuer's memo entries — like the entry inputs above, on their way while the
  // window fills: the issuer's first recorded Name and CRL distribution point (arena references) and its bitmap row
  unsigned long long dn_ref = 0, crl_ref = 0;
  const uint32_t* hour_row = nullptr;
  if constexpr (META) {
    if (live && mc.enabled && in.iss_in_range) {
      if (canon < mc.n_refs) {
        dn_ref = mc.refs[canon];
        crl_ref = mc.refs[mc.n_refs + canon];
      }
      if (canon / META_HOUR_PAGE < mc.n_hour_pages)
        hour_row = mc.hour_pages[canon / META_HOUR_PAGE] + (uint64_t)(canon % META_HOUR_PAGE) * (META_HOUR_BITS / 32);
    }
  }
  // the wave's buffer descriptor (readers.h): its base is lane 0's certificate — the lowest of a packed batch or a decoded
  // blob; lane 0 is live whenever the workgroup exists — and every window position is a 32-bit offset from it
  const WaveBuf wb = wave_buf(a.payload, limit, lo);
  const uint32_t lrel = wave_rel(wb, lo, live);
  // (dword aligned: the window loses at most 3 bytes in front; WinGeo::SKIP: it begins behind the two outer headers, which the
  //  walk reads from the sixteen octets below — on their way together with the fill)
  const uint32_t w_me = lrel == REL_NONE ? REL_NONE : ((lrel + WinGeo<WCH>::SKIP) & ~3u);
  U16t hd16{0u, 0u, 0u, 0u};
  bool hd_ok = false;
  if constexpr (WinGeo<WCH>::SKIP != 0u) {
    hd_ok = live & (lo + 16ull <= limit);
    hd16 = *(const U16t*)(a.payload + (hd_ok ? lo : 0ull));
  }
  coop_fill<WCH, false>(wb, w_me, lane);
  uint4 o0 = make_uint4(0, 0, 0, 0), o1 = o0;
  unsigned long long kmeta = 0ull, ks[5] = {0ull, 0ull, 0ull, 0ull, 0ull};  // the entry's key, for its arena cell
  bool keyed = false;
  uint32_t rem_owner = KEY_NO_OWNER;  // XM_OWNER: the rank this entry's key record goes to
  bool rem_long = false;
  uint4 k0 = make_uint4(0, 0, 0, 0), k1 = k0;
  unsigned long long keypos = 0ull;
  bool pending = false;
  if (live) {
    using Hook = typename std::conditional<META, MetaHook, NoRefillHook>::type;
    // (a certificate out of the descriptor's reach — an entry view in no order — has no window: every read misses, and the
    //  exact reader below decides)
    WinReaderS<WCH, Hook> r{{{(const uint32_t*)a.payload, lo, limit, (uint32_t*)(smem + win_off<WCH>(lane)),
                              lrel == REL_NONE ? (int32_t)0x80000000 : (int32_t)(w_me - lrel), wb, lrel}},
                            lrel == REL_NONE ? 0xffffffffu : 0u, {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, 0x80000000u};
    if constexpr (WinGeo<WCH>::SKIP != 0u) {
      r.hd[0] = hd16.a; r.hd[1] = hd16.b; r.hd[2] = hd16.c; r.hd[3] = hd16.d;
      r.hd_ok = hd_ok;
    }
    if constexpr (META) {
      r.hook.mc = mc;
      r.hook.canon = canon;
      r.hook.dn_ref = dn_ref;
      r.hook.dn_seen = false;
    }
    uint2 ml = make_uint2(META_NONE, META_NONE);
    map_one<STRICT>(r, hi - lo, i, a, in, o0, o1, &ml, &keypos);
    if (r.missed()) {  // some access left the window: the exact reader decides (rare: hostile or odd layouts)
      GlobalReader g{(const uint32_t*)a.payload, lo};
      map_one<STRICT>(g, hi - lo, i, a, in, o0, o1, nullptr, &keypos);
    }
    pending = keypos != 0ull;  // strict_spki: an EC key that owes the curve equation — ent[] carries that, not the record
    o0.x &= ~(FL_KEY_PENDING << 8);
    const uint32_t status = o0.x & 0xffu;
    uint32_t state = ES_NONE;
    // META: only a certificate that WAS unknown reaches IssuerMetadata.Accumulate; of those, only one that brings
    // something the memo does not hold yet needs k_meta_new.  Anything not positively known to be seen counts as unseen.
    // The lookup's loads go out before the table probe and are looked at behind it.
    MetaTail mt;
    bool meta_try = false;
    if constexpr (META) {
      meta_try = mc.enabled && status == CTMR_ST_PASS && !r.missed() && ml.x != META_NONE && ml.x != META_HOST && r.hook.dn_seen;
      if (meta_try)
        mt = meta_tail_issue(mc, crl_ref, hour_row, (int32_t)o0.y, ml.y, r.win, r.grel, r.part ? 0u : WinReader<WCH>::WBYTES - 8u);
    }
    if (status == CTMR_ST_PASS) {
      const uint32_t slen = o0.x >> 16;
      if (slen > CTMR_MAX_SERIAL) {
        state = ES_HOST;
      } else {
        unsigned long long* const s = ks;
        record_key(ia, i, o0, o1, s);
        const unsigned long long meta = kmeta = key_meta((int32_t)o0.y, canon, slen);
        keyed = true;
        const unsigned long long h = key_hash(meta, s);
        bool mine = true;
        if constexpr (MODE == XM_OWNER) {
          rem_owner = key_owner_h(h, xa.world);
          mine = rem_owner == xa.rank;
        }
        if (mine) {
#ifdef CTMR_EXP_NO_PROBE  // sweep builds, MEASUREMENT ONLY (wrong results): the kernel without any table access
          state = ES_CLAIMED;
#else
          // (a key whose EC point is not checked yet is not inserted: k_ec_resolve does both)
          state = pending ? (uint32_t)ES_PENDING : insert_probe_h(ia, i, meta, s, h);
#endif
        } else {  // (XM_OWNER) the record that leaves; serials of 21..40 octets take the 64-byte path (k_xl_export)
          state = ES_REMOTE;
          rem_long = slen > 20u;
          k0 = make_uint4((uint32_t)meta, (uint32_t)(meta >> 32), (uint32_t)s[0], (uint32_t)(s[0] >> 32));
          k1 = make_uint4((uint32_t)s[1], (uint32_t)(s[1] >> 32), (uint32_t)s[2], ia.ord_base + (uint32_t)i);
        }
        if constexpr (MODE == XM_BLOOM) {
          if (state == ES_CLAIMED || state == ES_DEFER) {
            uint64_t word;
            unsigned long long bits;
            bloom_pos(h, xa.bloom_wmask, word, bits);
            atomicOr(&xa.bloom[word], bits);  // result unused: a fire-and-forget atomic
          }
        }
      }
      if (state != ES_CLAIMED && state != ES_DEFER && state != ES_REMOTE && state != ES_PENDING)
        o0.x &= ~((uint32_t)CTMR_FL_WAS_UNKNOWN << 8);
    }
    uint32_t e = ent_pack(status, state, canon) | (pending ? ENT_KEY_PENDING : 0u);
    if constexpr (META) {
      const bool seen = meta_try && meta_tail_finish(mc, canon, mt, (int32_t)o0.y, r.win);
      e |= seen ? 0u : ENT_META_UNSEEN;
    }
    ia.ent[i] = e;
  }
  store_records_wave(a, first, live, o0, o1);
  __builtin_amdgcn_wave_barrier();
  // the wave's 64 key cells, arena[ref0 + first …): one contiguous 4 KiB store (the window area is free by now)
  if (first < a.n)
    store_cells_wave(ia.t.arena + ia.ref0 + first, a.n - first, (uint4*)smem, lane, keyed, kmeta, ks, ia.ord_base + (uint32_t)i);
  {  // strict_spki: does any entry owe k_ec_resolve a curve equation?  (No, in a batch of RSA keys: it exits at once.)  A flag
     // that the first waves set, not a count: on the mixed corpus nearly every wave has such entries, and 1.5 M atomics on
     // one address serialise at the memory side behind the kernel's back.
    const unsigned long long mp = __ballot(pending);
    if (mp && lane == 0 && ld_agent(&ia.stats->n_pending) == 0ull) st_agent(&ia.stats->n_pending, 1ull);
    const unsigned long long mo = __ballot(pending && ((keypos >> 32) & 7ull) != 1ull);  // a curve other than P-256
    if (mo && lane == 0 && ld_agent(&ia.stats->n_pending_other) == 0ull) st_agent(&ia.stats->n_pending_other, 1ull);
    if constexpr (MODE != XM_OWNER) {
      if (pending) a.keypos[i] = keypos;
    }
  }
  if constexpr (MODE == XM_OWNER) {
    // the wave's key records, grouped by owner: one ballot per rank gives every record its place and the per-owner counts
    const bool rem = rem_owner != KEY_NO_OWNER && rem_owner != xa.rank && !rem_long;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t pos = 0, before = 0, mine_cnt = 0;
    for (uint32_t o = 0; o < xa.world; o++) {
      const unsigned long long m = __ballot(rem && rem_owner == o);
      const uint32_t c = (uint32_t)__popcll(m);
      if (rem && rem_owner == o) pos = before + (uint32_t)__popcll(m & lt);
      if (lane == o) mine_cnt = c;
      before += c;
    }
    if (rem) {
      uint4* out = (uint4*)(xa.stage + first + pos);
      out[0] = k0;
      out[1] = k1;
    }
    // a pending key that left for its owner: k_ec_resolve withdraws record `pos` of this wave's group when the point is bad
    if (pending) a.keypos[i] = keypos | (rem ? (1ull << 38) | ((unsigned long long)pos << 39) : 0ull);
    if (lane < MAX_WORLD) xa.wave_cnt[(uint64_t)blockIdx.x * MAX_WORLD + lane] = (uint8_t)mine_cnt;
    const unsigned long long ml = __ballot(rem_owner != KEY_NO_OWNER && rem_owner != xa.rank && rem_long);
    if (ml && lane == 0) atomicAdd(&ia.stats->n_xl, (unsigned long long)__popcll(ml));
  }
}

// (Round-2 experiment, removed from the tree in round 3 — profiles/r02/sweep_pipelined_probe.txt: k_map_pipe let one wave
// map 2, 4 or 8 consecutive groups and ran the table probe of group g INSIDE group g+1's fills.  26.8–27.1 ms against
// 22.9 ms whatever share of the probe chains was hidden: the chain's latency is already covered by the CU's other
// waves; what a sparser table saves is random-access transactions, and the pending chain's registers cost more than
// there was to hide.)

// (Tried and dropped, round 2: two groups of 64 certificates per wave with nothing carried across but the next group's
// INPUTS — byte range, issuer index, entry type — requested before the current group's fill, so that the offsets →
// addresses round trip is paid once per wave: 25.9 ms against 23.5 ms in alternating runs on one box.  The loop costs
// registers (180 VGPRs; 168 without spills when held to three waves per SIMD) and, like k_map_pipe, runs slower than one
// group per workgroup whatever it hides — profiles/r02/ab_two_groups_per_wave_not_kept.txt.)
// (Tried and dropped, session 4: a software-pipelined form — one wave walks 2 or 4 batches of 64 certificates and
// fetches the next batch's front windows into registers while walking the current one.  256 VGPRs → 8 waves per CU,
// and the first vector load inside the walk (the issuerCN filter words) waits on vmcnt for the prefetch issued just
// before it, so the overlap never materialises: 26.0 ms against 23.1 ms, profiles/r01/s4/sweep_pipe.txt.)
// ------------------------------------------------------------------ strict_spki: the curve equation, then the insert
// spki_key.h "where the curve equation runs": the map leaves an entry whose EC key passed everything but
// y² = x³ − 3x + b "key pending" (ent[] bit 7, the point's place in keypos[]); this kernel — its own register file: the
// four modular products need 84..170 VGPRs — evaluates the equation and
//   point bad   the entry becomes what the map makes of a certificate that does not parse (status PARSE_ERROR, nothing
//               else reported); a key record that already left for its owner (XM_OWNER) is withdrawn (meta = 0);
//   point good  the pending bit goes; an entry in state ES_PENDING (would reach the set, key is this rank's) gets PASS 1
//               of the insert exactly as the map's other entries got it inside the fused kernel (insert_probe_h; its key
//               cell was written by the map): claim, or DEFER on a same-tag word of this round — k_insert2, which runs
//               next, settles those.  (First version, measured: the fully synchronised upsert — agent-scope loads, CAS, six
//               written-through stores and a drained publish per key — 80 ms per 47 M keys; every one of them a DRAM
//               read + write at the memory side, profiles/r03/calib_atomics_by_scope.jsonl.)
// A block takes 1 024 consecutive entries, compacts the pending ones into LDS and walks that list 256 at a time: the
// lanes of a wave are busy whatever share of the batch has EC keys.  A batch without any costs one launch of blocks that
// read one word and leave.
template <class C>
__device__ __forceinline__ bool ec_point_bits(const uint32_t* words, unsigned long long xbit) {
  uint32_t x[C::NL], y[C::NL];
  fe_load_bits<C>(words, xbit, x);
  fe_load_bits<C>(words, xbit + 8ull * C::BYTES, y);
  return ec_equation<C>(x, y);
}

// Two instantiations, launched one after the other: P256 = true takes the P-256 keys — every EC key of the public CT logs
// but a few — with 8-limb field elements (≈ 110 VGPRs: four waves per SIMD); P256 = false takes the other curves, whose
// 17-limb worst case (P-521) costs 206 VGPRs and two waves per SIMD — as ONE kernel the rare curves set the register file,
// and the occupancy, of the common one (7.1 ms per 50 M P-256 keys, round 4).  Each exits at once when the map saw no key
// of its kind.
constexpr uint32_t EC_PER_BLOCK = 1024;
template <bool P256>
__global__ void __launch_bounds__(256) k_ec_resolve(InsertArgs a, ctmr_record* records) {
  __shared__ uint16_t list[EC_PER_BLOCK];
  __shared__ uint32_t n_list;
  if ((P256 ? a.stats->n_pending : a.stats->n_pending_other) == 0ull) return;
  const uint64_t blk0 = (uint64_t)blockIdx.x * EC_PER_BLOCK;
  if (threadIdx.x == 0) n_list = 0u;
  __syncthreads();
  {  // the pending entries of this block's 1 024, four ent[] words per thread
    const uint64_t i0 = blk0 + threadIdx.x * 4u;
    uint32_t e[4] = {0u, 0u, 0u, 0u};
    if (i0 + 4 <= a.n) {
      const uint4 v = *(const uint4*)(a.ent + i0);
      e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w;
    } else {
      for (uint32_t k = 0; i0 + k < a.n; k++) e[k] = a.ent[i0 + k];
    }
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
      if ((e[k] & ENT_KEY_PENDING) && ((((uint32_t)(a.keypos[i0 + k] >> 32) & 7u) == 1u) == P256))
        list[atomicAdd(&n_list, 1u)] = (uint16_t)(threadIdx.x * 4u + k);
  }
  __syncthreads();
  const uint32_t cnt = n_list;
  for (uint32_t base = 0; base < cnt; base += 256u) {
    const uint32_t t = base + threadIdx.x;
    if (t < cnt) {
      const uint64_t i = blk0 + list[t];
      const uint32_t e = a.ent[i];
      const unsigned long long kp = a.keypos[i];
      uint64_t lo, hi;
      cert_range(a.offsets, a.ends, i, lo, hi);
      const uint32_t pos = (uint32_t)kp, curve = (uint32_t)(kp >> 32) & 7u, shift = (uint32_t)(kp >> 35) & 7u;
      const unsigned long long xbit = 8ull * (lo + pos) - shift;  // where X starts, as a bit position in the payload
      // the entry's record, asked for NOW: it arrives while the equation is evaluated instead of costing a round trip of its
      // own behind it (round 5)
      uint4* rp = (uint4*)(records + i);
      const uint4 r0 = rp[0], r1 = rp[1];
      bool good;
      if constexpr (P256) {
        good = ec_point_bits<CurveP256>((const uint32_t*)a.payload, xbit);
      } else {
        switch (curve) {
          case 2u: good = ec_point_bits<CurveP384>((const uint32_t*)a.payload, xbit); break;
          case 3u: good = ec_point_bits<CurveP521>((const uint32_t*)a.payload, xbit); break;
          case 4u: good = ec_point_bits<CurveP224>((const uint32_t*)a.payload, xbit); break;
          default: good = ec_point_bits<CurveP192>((const uint32_t*)a.payload, xbit); break;
        }
      }
      if (!good) {  // x509.ParseCertificate fails: map_one's record of such a certificate
        // (an entry the downloader had dropped already — entry_type CTMR_ENTRY_INVALID through ctmr_map_batch — stays that)
        const uint32_t stn = (r0.x & 0xffu) == CTMR_ST_ENTRY_DECODE_ERROR ? (uint32_t)CTMR_ST_ENTRY_DECODE_ERROR : (uint32_t)CTMR_ST_PARSE_ERROR;
        rp[0] = make_uint4(stn | (((r0.x >> 8) & CTMR_FL_PRECERT) << 8), 0u, r0.z, 0u);
        rp[1] = make_uint4(0u, 0u, 0u, 0u);
        if ((kp >> 38) & 1ull) {
          a.stage[(i & ~63ull) + ((kp >> 39) & 63ull)].meta = 0ull;  // the owner ignores it
        } else if (ent_state(e) == ES_REMOTE) {
          // left for its owner but not staged: a 21..40-octet serial, which k_xl_export collects from ent[] afterwards —
          // it will not find this entry any more, so the count the map took (DevStats.n_xl) must not include it.  (Found
          // by scripts/fuzz_gpu_groups.py: the host sized the 64-byte record list by the stale count and sent its
          // uninitialised tail — owner 0, order 0 — which cost entry 0 of rank 0 its WasUnknown.)
          atomicAdd(&a.stats->n_xl, ~0ull);
        }
        a.ent[i] = ent_pack(stn, ES_NONE, e >> 8);
      } else if (ent_state(e) != ES_PENDING) {  // filtered, without issuer, left for its owner, host-side serial: as the map said
        a.ent[i] = e & ~ENT_KEY_PENDING;
      } else {
        unsigned long long s[5];
        record_key(a, i, r0, r1, s);
        const unsigned long long meta = key_meta((int32_t)r0.y, e >> 8, r0.x >> 16);
        const unsigned long long h = key_hash(meta, s);
        const uint32_t state = insert_probe_h(a, i, meta, s, h);  // (its cell was written by the map, like every entry's)
        if (state != ES_CLAIMED && state != ES_DEFER) {
          ((uint8_t*)(records + i))[1] = (uint8_t)((r0.x >> 8) & ~CTMR_FL_WAS_UNKNOWN);
        } else if (a.bloom) {
          uint64_t word;
          unsigned long long bits;
          bloom_pos(h, a.bloom_wmask, word, bits);
          atomicOr(&a.bloom[word], bits);
        }
        a.ent[i] = (e & ~(ENT_KEY_PENDING | (7u << 3))) | (state << 3);
      }
    }
  }
}

// Wave-aggregated add: one atomic per distinct key per wave (the "match-any" loop).
__device__ __forceinline__ void wave_agg_add(bool active, uint32_t key, unsigned long long* arr) {
  unsigned long long todo = __ballot(active);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t k = __shfl(key, leader);
    const unsigned long long same = __ballot(active && key == k) & todo;
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&arr[k], (unsigned long long)__popcll(same));
    todo &= ~same;
  }
}

__device__ __forceinline__ bool pair_add(PairSlot* pairs, uint64_t pmask, unsigned long long key,
                                         long long delta) {
  uint64_t j = mixk(key) & pmask;
  for (uint64_t probes = 0; probes <= pmask; probes++) {
    unsigned long long k = ld_agent(&pairs[j].key);
    if (k == 0ull) {
      const unsigned long long old = atomicCAS(&pairs[j].key, 0ull, key);
      k = old == 0ull ? key : old;
    }
    if (k == key) {
      atomicAdd(&pairs[j].count, (unsigned long long)delta);
      return true;
    }
    j = (j + 1) & pmask;
  }
  return false;
}

struct ResolveArgs {
  const uint32_t* ent;
  unsigned long long* issuer_counts;  // per canonical issuer
  DevStats* stats;
  uint32_t* blk_new;  // NEW count per 1024-entry block
  uint64_t n;
};

// One streaming pass over ent[] (4 bytes per entry; no table or record access): per-issuer unique
// counts of the entries that WERE unknown (Σ_expDate SCARD, storage-statistics.go:44-53), status
// histogram, NEW count per 1024-entry block for the compaction.
// Persistent blocks: per-issuer counts are first accumulated in an LDS histogram (issuers
// below RES_LDS_ISSUERS) and flushed with ONE global atomic per non-empty bin per block —
// hundreds of thousands of device atomics on the few cache lines of the hot issuers serialise
// at the memory side otherwise.  Issuers beyond the LDS bins use wave-aggregated global atomics.
constexpr uint32_t RES_LDS_ISSUERS = 4096;

__global__ void __launch_bounds__(1024) k_resolve(ResolveArgs a, uint64_t nb) {
  __shared__ uint32_t hist[CTMR_ST__COUNT + 5];
  __shared__ uint32_t ih[RES_LDS_ISSUERS];
  __shared__ uint32_t blk_cnt[4];
  if (threadIdx.x < CTMR_ST__COUNT + 5) hist[threadIdx.x] = 0;
  for (uint32_t k = threadIdx.x; k < RES_LDS_ISSUERS; k += 1024) ih[k] = 0;
  __syncthreads();
  // Four entries per thread, one 16-byte load: a workgroup takes FOUR 1024-entry blocks per step (threads 256q..256q+255
  // hold block q's entries), so that it has four times the bytes in flight between its barriers — with one entry per
  // thread the kernel ran at the ≈ 1 TB/s that 2 048 four-byte loads per CU and round trip give (0.40 ms per 100 M).
  const uint64_t nsb = (nb + 3) / 4;
  for (uint64_t sb = blockIdx.x; sb < nsb; sb += gridDim.x) {
    if (threadIdx.x < 4) blk_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t i0 = sb * 4096 + (uint64_t)threadIdx.x * 4u;
    uint32_t e[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};  // status 7 with state 7: counted nowhere
    bool live[4] = {false, false, false, false};
    if (i0 + 4 <= a.n) {
      const uint4 v = *(const uint4*)(a.ent + i0);
      e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w;
      live[0] = live[1] = live[2] = live[3] = true;
    } else {
      for (uint32_t k = 0; i0 + k < a.n; k++) {
        e[k] = a.ent[i0 + k];
        live[k] = true;
      }
    }
    uint32_t w_new = 0, w_dup = 0, w_host = 0, w_full = 0, w_rem = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      const uint32_t status = live[k] ? (e[k] & 7u) : (uint32_t)CTMR_ST__COUNT;
      const uint32_t st = ent_state(e[k]), canon = e[k] >> 8;
      bool is_new = live[k] & ((st == ES_CLAIMED) | (st == ES_DEFER));
      const bool is_remote = live[k] & (st == ES_REMOTE);  // counted where it is stored: by the key's owner (k_keys_resolve)
      if (is_new && canon < RES_LDS_ISSUERS) atomicAdd(&ih[canon], 1u);
      wave_agg_add(is_new && canon >= RES_LDS_ISSUERS, canon, a.issuer_counts);
      is_new = is_new | is_remote;  // from here on: "in the NEW list" (optimistically, for a key that left)
      // (the per-(expDate, issuer) cardinalities are rebuilt lazily by k_build_pairs on the first
      //  SetCardinality/KeysToChan after a mutation — they are statistics, not hot-path state)
      w_new += (uint32_t)__popcll(__ballot(is_new));
      w_dup += (uint32_t)__popcll(__ballot(live[k] & (st == ES_DUP)));
      w_host += (uint32_t)__popcll(__ballot(live[k] & (st == ES_HOST)));
      w_full += (uint32_t)__popcll(__ballot(live[k] & (st == ES_FULL)));
      w_rem += (uint32_t)__popcll(__ballot(is_remote));
#pragma unroll
      for (uint32_t s = 0; s < CTMR_ST__COUNT; s++) {
        const unsigned long long m = __ballot(status == s);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(&hist[s], (uint32_t)__popcll(m));
      }
    }
    if ((threadIdx.x & 63) == 0) {
      if (w_rem) atomicAdd(&hist[CTMR_ST__COUNT + 4], w_rem);
      if (w_new) {
        atomicAdd(&blk_cnt[threadIdx.x >> 8], w_new);
        atomicAdd(&hist[CTMR_ST__COUNT], w_new);
      }
      if (w_dup) atomicAdd(&hist[CTMR_ST__COUNT + 1], w_dup);
      if (w_host) atomicAdd(&hist[CTMR_ST__COUNT + 2], w_host);
      if (w_full) atomicAdd(&hist[CTMR_ST__COUNT + 3], w_full);
    }
    __syncthreads();
    if (threadIdx.x < 4 && sb * 4 + threadIdx.x < nb) a.blk_new[sb * 4 + threadIdx.x] = blk_cnt[threadIdx.x];
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < RES_LDS_ISSUERS; k += 1024)
    if (ih[k]) atomicAdd(&a.issuer_counts[k], (unsigned long long)ih[k]);
  if (threadIdx.x < CTMR_ST__COUNT) {
    if (hist[threadIdx.x]) atomicAdd(&a.stats->by_status[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
  } else if (threadIdx.x == CTMR_ST__COUNT) {
    if (hist[CTMR_ST__COUNT]) atomicAdd(&a.stats->n_new, (unsigned long long)hist[CTMR_ST__COUNT]);
  } else if (threadIdx.x == CTMR_ST__COUNT + 1) {
    if (hist[CTMR_ST__COUNT + 1]) atomicAdd(&a.stats->n_dup, (unsigned long long)hist[CTMR_ST__COUNT + 1]);
  } else if (threadIdx.x == CTMR_ST__COUNT + 2) {
    if (hist[CTMR_ST__COUNT + 2]) atomicAdd(&a.stats->n_host, (unsigned long long)hist[CTMR_ST__COUNT + 2]);
  } else if (threadIdx.x == CTMR_ST__COUNT + 3) {
    if (hist[CTMR_ST__COUNT + 3]) atomicAdd(&a.stats->n_full, (unsigned long long)hist[CTMR_ST__COUNT + 3]);
  } else if (threadIdx.x == CTMR_ST__COUNT + 4) {
    if (hist[CTMR_ST__COUNT + 4]) atomicAdd(&a.stats->n_remote, (unsigned long long)hist[CTMR_ST__COUNT + 4]);
  }
}

// Stream compaction of the NEW entries, ascending: wave ballot + popcount prefix inside a
// 1024-entry block, block bases from the exclusive scan of blk_new.  The NEW predicate comes
// from ent[] (local reduce) or from the record flag (exchange mode, ent == nullptr).
// Four entries per thread (256 threads per 1024-entry block): the per-thread counts 0..4 are prefix-summed over the
// wave with three ballots (one per bit of the count).
__global__ void __launch_bounds__(256) k_compact(const ctmr_record* records, const uint32_t* ent, uint64_t n,
                                                 const uint64_t* blk_base, uint64_t* new_idx) {
  __shared__ uint32_t wave_cnt[4];
  const uint64_t i0 = (uint64_t)blockIdx.x * 1024 + threadIdx.x * 4u;
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  bool f[4] = {false, false, false, false};
  if (ent) {
    if (i0 + 4 <= n) {
      const uint4 v = *(const uint4*)(ent + i0);
      f[0] = ent_is_new(v.x); f[1] = ent_is_new(v.y); f[2] = ent_is_new(v.z); f[3] = ent_is_new(v.w);
    } else {
      for (uint32_t k = 0; k < 4 && i0 + k < n; k++) f[k] = ent_is_new(ent[i0 + k]);
    }
  } else {
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
      if (i0 + k < n) f[k] = (((const uint8_t*)(records + i0 + k))[1] & CTMR_FL_WAS_UNKNOWN) != 0;
  }
  const uint32_t c = (uint32_t)f[0] + f[1] + f[2] + f[3];
  const unsigned long long m0 = __ballot(c & 1u), m1 = __ballot(c & 2u), m2 = __ballot(c & 4u);
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t pre = (uint32_t)__popcll(m0 & lt) + 2u * (uint32_t)__popcll(m1 & lt) + 4u * (uint32_t)__popcll(m2 & lt);
  if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(m0) + 2u * (uint32_t)__popcll(m1) + 4u * (uint32_t)__popcll(m2);
  __syncthreads();
  if (c) {
    for (uint32_t k = 0; k < wv; k++) pre += wave_cnt[k];
    uint64_t at = blk_base[blockIdx.x] + pre;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
      if (f[k]) new_idx[at++] = i0 + k;
  }
}

// exclusive scan of blk_new (u32) into blk_base (u64): single workgroup
__global__ void __launch_bounds__(1024) k_scan_blocks(const uint32_t* blk_new, uint64_t nb,
                                                      uint64_t* blk_base) {
  // Tiles of 4 096 counts: thread t takes four consecutive ones (coalesced), the threads' sums are scanned over the wave
  // with shuffles and over the 16 waves through LDS, a running carry links the tiles.  (Before: every thread summed a
  // contiguous CHUNK of nb/1024 counts — lanes 96 elements apart, one dependent load after the other: ≈ 0.17 ms per 100 M
  // entries, all of it on the critical path between k_resolve and k_compact.)
  __shared__ unsigned long long wsum[16];
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  unsigned long long carry = 0;
  for (uint64_t base = 0; base < nb; base += 4096u) {
    const uint64_t i0 = base + (uint64_t)threadIdx.x * 4u;
    uint32_t v[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) v[k] = i0 + k < nb ? blk_new[i0 + k] : 0u;
    const unsigned long long tsum = (unsigned long long)v[0] + v[1] + v[2] + v[3];
    unsigned long long x = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long y = __shfl_up(x, d);
      if ((int)lane >= d) x += y;
    }
    if (lane == 63u) wsum[wv] = x;
    __syncthreads();
    unsigned long long before = 0, all = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
      if (k < wv) before += wsum[k];
      all += wsum[k];
    }
    unsigned long long run = carry + before + x - tsum;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      if (i0 + k < nb) blk_base[i0 + k] = run;
      run += v[k];
    }
    carry += all;
    __syncthreads();  // wsum is rewritten by the next tile
  }
}

// ------------------------------------------------------------------ device-wide prefix sum of u64, in place
// (PEM offsets of the NEW list, offsets of the synthetic generators: up to 2·10^8 elements.)  Three launches, the data
// read twice and written once: per-tile sums (4 096 elements per 256-thread block, 16 consecutive elements per thread,
// loaded as eight 16-byte vectors) → k_scan64_sums, one workgroup, exclusive scan of the tile sums → every tile scans
// itself on top of its base.  Hand-written instead of hipcub::DeviceScan (round 2 review): the same three passes,
// no temporary-storage query, no library in the product's link line.
constexpr uint32_t SCAN_TILE = 4096;

__device__ __forceinline__ unsigned long long block_exclusive_256(unsigned long long v, unsigned long long* total) {
  __shared__ unsigned long long wsum[4];
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned long long x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long y = __shfl_up(x, d);
    if ((int)lane >= d) x += y;
  }
  __syncthreads();  // (wsum may still be read by the previous call)
  if (lane == 63) wsum[wv] = x;
  __syncthreads();
  unsigned long long before = 0, all = 0;
#pragma unroll
  for (uint32_t k = 0; k < 4; k++) {
    if (k < wv) before += wsum[k];
    all += wsum[k];
  }
  if (total) *total = all;
  return before + x - v;
}

__global__ void __launch_bounds__(256) k_scan64_tiles(const unsigned long long* data, uint64_t n, unsigned long long* tile_sum) {
  const uint64_t i0 = (uint64_t)blockIdx.x * SCAN_TILE + threadIdx.x * 16u;
  unsigned long long s = 0;
  if (i0 + 16 <= n) {  // (8-byte loads: every caller passes an array + 1 — 8-byte, not 16-byte aligned)
#pragma unroll
    for (int k = 0; k < 16; k++) s += data[i0 + k];
  } else {
    for (uint64_t i = i0; i < n && i < i0 + 16; i++) s += data[i];
  }
  unsigned long long all;
  (void)block_exclusive_256(s, &all);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = all;
}

__global__ void __launch_bounds__(1024) k_scan64_sums(unsigned long long* sums, uint64_t nt) {  // exclusive, in place
  __shared__ unsigned long long part[1024];
  const uint64_t per = (nt + 1023) / 1024;
  const uint64_t lo = (uint64_t)threadIdx.x * per < nt ? (uint64_t)threadIdx.x * per : nt;
  const uint64_t hi = lo + per < nt ? lo + per : nt;
  unsigned long long sum = 0;
  for (uint64_t i = lo; i < hi; i++) sum += sums[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    const unsigned long long t = threadIdx.x >= d ? part[threadIdx.x - d] : 0ull;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  unsigned long long run = part[threadIdx.x] - sum;
  for (uint64_t i = lo; i < hi; i++) {
    const unsigned long long v = sums[i];
    sums[i] = run;
    run += v;
  }
}

template <bool INCLUSIVE>
__global__ void __launch_bounds__(256) k_scan64_apply(unsigned long long* data, uint64_t n, const unsigned long long* tile_base) {
  const uint64_t i0 = (uint64_t)blockIdx.x * SCAN_TILE + threadIdx.x * 16u;
  unsigned long long v[16];
  unsigned long long s = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    v[k] = i0 + k < n ? data[i0 + k] : 0ull;
    s += v[k];
  }
  unsigned long long run = tile_base[blockIdx.x] + block_exclusive_256(s, nullptr);
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const unsigned long long x = v[k];
    if (i0 + k < n) data[i0 + k] = INCLUSIVE ? run + x : run;
    run += x;
  }
}

}  // namespace ctmr
