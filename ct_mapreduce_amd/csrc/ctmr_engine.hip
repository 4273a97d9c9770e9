// ctmr_engine.hip — libctmr: the C ABI of include/ctmr.h over the gfx950 kernels.
//
// Plain HIP runtime, no torch, no CPU fallback: if the HIP device is unusable every entry
// point fails with CTMR_E_HIP.  Host-side state kept here is only what the reference keeps in
// Redis for keys that are not known-certificate sets (crl::, issuer::, log:: …) plus the
// issuer registry.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <system_error>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "host_keys.h"
#include "kernels.h"

using namespace ctmr;

namespace {

struct HostReader {  // host instantiation of the walk for the long-serial slow path only
  const uint8_t* p;
  uint32_t ld4(uint32_t pos) const {
    uint32_t v;
    memcpy(&v, p + pos, 4);
    return v;
  }
  uint32_t ldg(uint32_t pos) const { return ld4(pos); }
  void touch(uint32_t, uint32_t) const {}
  void touch_tail(uint32_t, uint32_t) const {}
};

struct IssuerRec {
  bool valid;
  uint32_t canon;
  uint8_t digest[32];
  std::string id;
};

std::string b64url(const uint8_t* in, size_t n) {
  static const char A[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789-_";
  std::string o;
  size_t i = 0;
  for (; i + 3 <= n; i += 3) {
    uint32_t v = (in[i] << 16) | (in[i + 1] << 8) | in[i + 2];
    o += A[(v >> 18) & 63]; o += A[(v >> 12) & 63]; o += A[(v >> 6) & 63]; o += A[v & 63];
  }
  if (n - i == 1) {
    uint32_t v = in[i] << 16;
    o += A[(v >> 18) & 63]; o += A[(v >> 12) & 63]; o += "==";
  } else if (n - i == 2) {
    uint32_t v = (in[i] << 16) | (in[i + 1] << 8);
    o += A[(v >> 18) & 63]; o += A[(v >> 12) & 63]; o += A[(v >> 6) & 63]; o += '=';
  }
  return o;
}

// ExpDate.ID() "2006-01-02-15" ↔ hours since the epoch (storage/types.go:339-384)
std::string exp_date_id(int32_t exp_hour) {
  int64_t days = exp_hour / 24;
  int hh = exp_hour % 24;
  if (hh < 0) { hh += 24; days -= 1; }
  int32_t y; uint32_t m, d;
  civil_from_days(days, y, m, d);
  char buf[32];
  snprintf(buf, sizeof buf, "%04d-%02u-%02u-%02d", y, m, d, hh);
  return buf;
}

bool parse_exp_date_id(const char* s, size_t n, int32_t* out) {
  if (n != 13) return false;
  for (int i = 0; i < 13; i++) {
    if (i == 4 || i == 7 || i == 10) { if (s[i] != '-') return false; }
    else if (s[i] < '0' || s[i] > '9') return false;
  }
  int y = (s[0]-'0')*1000 + (s[1]-'0')*100 + (s[2]-'0')*10 + (s[3]-'0');
  uint32_t m = (s[5]-'0')*10 + (s[6]-'0'), d = (s[8]-'0')*10 + (s[9]-'0'), h = (s[11]-'0')*10 + (s[12]-'0');
  if (m < 1 || m > 12 || d < 1 || d > 31 || h > 23) return false;
  int64_t hours = days_from_civil(y, m, d) * 24 + h;
  if (hours < INT32_MIN || hours > INT32_MAX) return false;
  if (exp_date_id((int32_t)hours) != std::string(s, n)) return false;  // e.g. Feb 30
  *out = (int32_t)hours;
  return true;
}

// Redis-style glob (KEYS pattern, rediscache.go:153-169): * ? [set] backslash-escape
bool glob_match(const char* p, size_t pn, const char* s, size_t sn) {
  size_t pi = 0, si = 0, star_p = (size_t)-1, star_s = 0;
  while (si < sn) {
    bool adv = false;
    if (pi < pn) {
      char c = p[pi];
      if (c == '*') { star_p = pi++; star_s = si; continue; }
      if (c == '?') { pi++; si++; adv = true; }
      else if (c == '[') {
        size_t q = pi + 1; bool neg = false, hit = false;
        if (q < pn && p[q] == '^') { neg = true; q++; }
        while (q < pn && p[q] != ']') {
          char lo = p[q];
          if (lo == '\\' && q + 1 < pn) lo = p[++q];
          char hi = lo;
          if (q + 2 < pn && p[q + 1] == '-' && p[q + 2] != ']') { hi = p[q + 2]; q += 2; }
          if (lo > hi) std::swap(lo, hi);
          if (s[si] >= lo && s[si] <= hi) hit = true;
          q++;
        }
        if (hit != neg) { pi = q < pn ? q + 1 : q; si++; adv = true; }
      } else {
        if (c == '\\' && pi + 1 < pn) c = p[++pi];
        if (c == s[si]) { pi++; si++; adv = true; }
      }
    }
    if (adv) continue;
    if (star_p == (size_t)-1) return false;
    pi = star_p + 1;
    si = ++star_s;
  }
  while (pi < pn && p[pi] == '*') pi++;
  return pi == pn;
}

}  // namespace

struct ctmr_pipeline;

// One batch of the reduce in flight or just finished (engine/map.inc: round_*; engine/exchange.inc continues it)
struct RoundCtx {
  bool valid = false;  // the scratch state (ent[], blk_new[]) still describes this batch
  int mode = XM_LOCAL;
  uint32_t world = 1, rank = 0, ord_base = 0;
  const uint8_t* d_payload = nullptr;
  const uint64_t* d_offsets = nullptr;
  const uint64_t* d_ends = nullptr;
  const uint32_t* d_issuer_idx = nullptr;
  const uint8_t* d_entry_type = nullptr;
  uint64_t n = 0, blob_bytes = 0, payload_bytes_known = ~0ull;
  ctmr_record* d_records = nullptr;
  uint64_t* d_new_idx = nullptr;
  bool compacted = false;   // the NEW list has been written
  DevStats hs{};            // after round_collect
  uint64_t payload_bytes = 0, host_new = 0;
  uint64_t ref0 = 0;        // arena cell of this shard's entry 0
  uint64_t recv_ref0 = 0;   // owner-computes round: arena cell of the first received key
  uint64_t lost = 0;        // entries that lost WasUnknown to another rank after the resolve (exchange / Bloom apply)
  uint64_t remote_new = 0;  // owner-computes round: received keys that were new here
  uint64_t n_xl = 0;        // owner-computes round: keys with 21..40-octet serials that left as 64-byte records
  std::vector<uint64_t> counts32;
  std::vector<KeyRec> xl;   // … those records, sorted by (owner, order)
  bool resolved = false;    // owner-computes round: ctmr_xchg_insert_device has run
  bool chunked = false;     // … mapped in chunks (ctmr_xchg_map_chunk_device): the 32-byte records were gathered chunk by chunk
  uint64_t chunk_next = 0;  // … first entry not mapped yet (the round is open for the next chunk while < n)
  bool own_counted = false; // the table was rebuilt between round_begin and round_collect: `occupied` (= the rebuild's live
                            // count) already holds this shard's own claims — round_collect must not add them again
  // members with serials longer than CTMR_MAX_SERIAL that this batch added to the host-side set, in log order: a group
  // round settles them between the ranks afterwards (engine/group.inc: round_finish)
  struct HostNew { uint64_t i; int32_t exp_hour; uint32_t canon; std::string member; };
  std::vector<HostNew> host_list;
};

struct ctmr_engine {
  std::mutex mu;
  std::atomic<ctmr_pipeline*> pipe{nullptr};  // asynchronous host ingestion (engine/pipeline.inc), created on first use
  std::mutex pipe_mu;
  mutable std::string err;
  mutable std::mutex err_mu;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  ctmr_config cfg{};
  bool tile_attr_set = false;  // CTMR_SWEEP builds: k_map_tile's dynamic-LDS attribute is set on this engine's device
  // known-certificate table: index words + key cells (ctmr_dev.h)
  unsigned long long* index = nullptr;
  uint64_t nslots = 0;
  KeyCell* arena = nullptr;
  uint64_t arena_cap = 0;   // cells
  uint64_t arena_used = 0;  // cells handed out (a round takes one per entry + one per received key; point inserts one each)
  Table tbl() const { return Table{index, nslots - 1, arena}; }
  uint64_t max_slots = 0;   // growth limit (config.max_table_slots; at most 2^31: slot ids are 32-bit)
  uint64_t occupied = 0;    // slots claimed since the table was last (re)built: live members + tombstones
  uint32_t rebuilds = 0;
  uint32_t arena_compactions = 0, arena_growths = 0;
  PairSlot* pairs = nullptr;
  uint64_t npairs = 0;
  unsigned long long* issuer_counts = nullptr;
  uint32_t epoch = 0;
  bool pairs_dirty = false;  // pair table must be rebuilt before the next cardinality / keys query
  // issuer table
  uint32_t max_issuers = 0;
  uint8_t* d_issuer_valid = nullptr;
  uint32_t* d_canon = nullptr;
  std::vector<IssuerRec> issuers;
  std::unordered_map<std::string, uint32_t> id_to_canon;
  // issuer certificate store for the Chain[0] match of the raw-entry path (k_chain0_match): every registered
  // certificate's bytes at a 16-byte aligned offset, its length and candidate hash, and a hash table index+1
  uint8_t* d_idb_der = nullptr;
  size_t idb_cap = 0, idb_used = 0;
  uint64_t* d_idb_off = nullptr;
  uint32_t* d_idb_len = nullptr;
  unsigned long long* d_idb_ht = nullptr;  // (candidate hash & ~0xffffffff) | (issuer index + 1)
  uint32_t idb_ht_size = 0;
  std::vector<unsigned long long> h_idb_ht;
  std::unordered_map<std::string, uint32_t> der_to_idx;  // first registration of each distinct certificate
  bool auto_register = true;               // raw-entry calls register unseen Chain[0] certificates themselves
  int chain0_mode = CTMR_CHAIN0_EXACT;     // ctmr_set_chain0_match
  // ctmr_create = CTMR_PROFILE_REFERENCE (round 6): all four switches on — what the reference does; ctmr_set_profile(e,
  // CTMR_PROFILE_FAST) is the opt-in of a host that has parsed already
  bool strict_ext = true;                  // ctmr_set_strict_extensions: extension bodies Go unmarshals (fatal)
  bool strict_strings = true;              // ctmr_set_strict_strings: character sets of the Names' string values (non-fatal finding)
  bool strict_spki = true;                 // ctmr_set_strict_spki: parsePublicKey's verdict on the key inside subjectPublicKeyInfo (spki_key.h)
  bool strict_leaf = true;                 // ctmr_set_strict_leaf: parse the leaf TBSCertificate of precertificate entries
  uint64_t meta_precheck_n = 0;            // entries of the last map call whose ent[] carries the memo pre-check (0 = none)
  const uint32_t* meta_precheck_ent = nullptr;
  std::unordered_map<unsigned long long, uint32_t> qh_first;  // (upper half of the candidate hash, length) → table slot of the first registered certificate with it
  std::vector<std::string> pending_issuers;  // auto_register off: what the last decode found unregistered
  unsigned long long* d_pend = nullptr;    // PEND_SLOTS claim words
  uint32_t* d_unreg = nullptr;             // UNREG_CAP entry indices
  unsigned long long* d_dcount = nullptr;  // 8 counters of the decode / match kernels
  // IssuerMetadata memo on device (k_meta_new)
  MetaSlot* d_meta_slots = nullptr;
  uint64_t n_meta_slots = 0;
  uint8_t* d_meta_arena = nullptr;
  unsigned long long* d_meta_refs = nullptr;  // [canon] the issuer's first recorded DN, [max_issuers + canon] its first CRL DP (MetaCheck)
  uint64_t meta_arena_cap = 0;
  std::vector<uint32_t*> meta_hour_pages;  // knownExpDates bitmaps, META_HOUR_PAGE issuers each
  uint32_t** d_meta_hour_pages = nullptr;  // device copy of the page pointers
  unsigned long long* d_mcount = nullptr;  // [0] arena used [1] items [2] overflow events
  uint64_t meta_n = 0;                     // entries the SC_META scratch describes (the last map call)
  uint32_t meta_epoch = 0;                 // k_meta_new launches so far
  // Bloom-variant global dedup: cumulative filter of the keys this rank found locally new
  unsigned long long* d_bloom = nullptr;
  uint64_t bloom_words = 0;
  bool bloom_owned = false;                // false: the filter lives in a caller-owned buffer
  uint64_t bloom_round_ref0 = ~0ull;       // arena cell of entry 0 of the current Bloom round's batch (~0 = empty batch: nothing of this round)
  bool last_meta_valid = false;            // SC_ITEMS holds the items of the last host batch
  uint64_t last_meta_items = 0;
  // the last ctmr_map_entries (host variant): ctmr_pem_new encodes from its view
  bool last_is_view = false;
  size_t last_o_start = 0, last_o_end = 0;
  // filter
  FilterDev* d_filter = nullptr;
  FilterDev h_filter{};
  // scratch
  DevStats* d_stats = nullptr;
  uint32_t* d_result = nullptr;        // 2 words for point ops
  unsigned long long* d_count = nullptr;
  void* d_scratch[28] = {};            // growable buffers
  size_t scratch_cap[28] = {};
  RoundCtx rd;                         // the last batch (engine/map.inc)
  // the last ctmr_map_batch (host variant): what ctmr_pem_new encodes
  uint64_t last_n = 0, last_n_new = 0;
  size_t last_o_off = 0, last_o_new = 0;
  hipEvent_t ev[8] = {};
  std::vector<uint64_t> h_rel;  // host staging of the rebased offsets of ctmr_map_batch
  // host-side store: non-table keys + members with serials longer than CTMR_MAX_SERIAL
  std::map<std::string, std::set<std::string>> hstore;
  std::map<std::string, int64_t> expiry;          // explicit ExpireAt overrides / host keys
  std::unordered_map<uint32_t, uint64_t> host_issuer_counts;  // canon → long-serial members
};

namespace {

enum { SC_RECORDS = 0, SC_SLOTID, SC_BLKNEW, SC_BLKBASE, SC_ENT, SC_STAGE_A, SC_STAGE_B, SC_MISC, SC_PEM, SC_PEMOFF, SC_TMP, SC_VIEW, SC_ISS_A, SC_ISS_B, SC_ISS_C, SC_META, SC_ITEMS,
       SC_XSTAGE, SC_XWCNT, SC_XCNT, SC_XBASE, SC_XSLOT, SC_XL,
       SC_NFX,    // (round 3: strict_strings' pre-pass; unused since the check moved into the walk)
       SC_KEYPOS, // strict_spki: where the EC points lie that owe the curve equation (map → k_ec_resolve)
       SC_PEMBLK, // k_pem_blocks: per 7 KiB output block (PEM_S) of the PEM stream, the first certificate in it
       SC_PEMINFO // k_pem_len: (offset, length) of every certificate of the NEW list
       };  // strict_strings: the pre-pass's finding per entry  // owner-computes exchange (engine/exchange.inc)
constexpr uint32_t UNREG_CAP = 16384;

int fail(const ctmr_engine* e, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) {
    std::lock_guard<std::mutex> g(e->err_mu);  // a leaf lock: the ingestion pipeline reports errors outside the engine mutex
    e->err = buf;
  }
  return code;
}

#define HIPCHK(e, call)                                                                   \
  do {                                                                                    \
    hipError_t _r = (call);                                                               \
    if (_r != hipSuccess)                                                                 \
      return fail(e, _r == hipErrorOutOfMemory ? CTMR_E_NOMEM : CTMR_E_HIP, "%s: %s", #call, \
                  hipGetErrorString(_r));                                                 \
  } while (0)

int ensure(ctmr_engine* e, int which, size_t bytes) {
  if (e->scratch_cap[which] >= bytes) return CTMR_OK;
  if (e->d_scratch[which]) {
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipFree(e->d_scratch[which]));
    e->d_scratch[which] = nullptr;
    e->scratch_cap[which] = 0;
  }
  size_t cap = bytes + bytes / 8 + 256;
  HIPCHK(e, hipMalloc(&e->d_scratch[which], cap));
  e->scratch_cap[which] = cap;
  return CTMR_OK;
}

uint64_t pow2_at_least(uint64_t v) {
  uint64_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

// "serials::<expDateID>::<issuerID>" of a registered issuer → (exp_hour, canon)
bool table_key(ctmr_engine* e, const char* key, size_t n, int32_t* exp_hour, uint32_t* canon) {
  if (n < 9 + 13 + 2 + 1 || memcmp(key, "serials::", 9) != 0) return false;
  if (key[22] != ':' || key[23] != ':') return false;
  if (!parse_exp_date_id(key + 9, 13, exp_hour)) return false;
  auto it = e->id_to_canon.find(std::string(key + 24, n - 24));
  if (it == e->id_to_canon.end()) return false;
  *canon = it->second;
  return true;
}

std::string make_key(ctmr_engine* e, int32_t exp_hour, uint32_t canon) {
  return "serials::" + exp_date_id(exp_hour) + "::" + e->issuers[canon].id;
}

void pack_serial(const uint8_t* m, size_t n, unsigned long long s[5]) {
  uint8_t buf[40] = {0};
  memcpy(buf, m, n);
  memcpy(s, buf, 40);
}

int upload_filter(ctmr_engine* e) {
  HIPCHK(e, hipMemcpyAsync(e->d_filter, &e->h_filter, sizeof(FilterDev), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return CTMR_OK;
}

int ensure_capacity(ctmr_engine* e, uint64_t incoming, bool round_start = false, uint64_t incoming_cells = ~0ull);

// in-place prefix sum of n u64 on the engine's stream (reduce.h: k_scan64_*); tile sums live in scratch buffer `which`
int scan_u64(ctmr_engine* e, uint64_t* d_data, uint64_t n, bool inclusive, int which) {
  if (n == 0) return CTMR_OK;
  const uint64_t nt = (n + SCAN_TILE - 1) / SCAN_TILE;
  int r;
  if ((r = ensure(e, which, nt * 8))) return r;
  unsigned long long* d_sums = (unsigned long long*)e->d_scratch[which];
  unsigned long long* d = (unsigned long long*)d_data;
  hipLaunchKernelGGL(k_scan64_tiles, dim3((unsigned)nt), dim3(256), 0, e->stream, (const unsigned long long*)d, n, d_sums);
  hipLaunchKernelGGL(k_scan64_sums, dim3(1), dim3(1024), 0, e->stream, d_sums, nt);
  if (inclusive) hipLaunchKernelGGL((k_scan64_apply<true>), dim3((unsigned)nt), dim3(256), 0, e->stream, d, n, (const unsigned long long*)d_sums);
  else hipLaunchKernelGGL((k_scan64_apply<false>), dim3((unsigned)nt), dim3(256), 0, e->stream, d, n, (const unsigned long long*)d_sums);
  return CTMR_OK;
}

int point_op(ctmr_engine* e, int op, int32_t exp_hour, uint32_t canon, const uint8_t* m, size_t n,
             int* out) {
  unsigned long long s[5];
  pack_serial(m, n, s);
  const unsigned long long meta = key_meta(exp_hour, canon, (uint32_t)n);
  unsigned long long my_ref = 0;
  if (op == 0) {
    int rc = ensure_capacity(e, 1);
    if (rc) return rc;
    e->epoch++;
    my_ref = e->arena_used++;  // the member's cell (left unused when the member turns out to be known)
  }
  if (op != 1) e->pairs_dirty = true;
  hipLaunchKernelGGL(k_set_op, dim3(1), dim3(64), 0, e->stream, e->tbl(), meta, s[0],
                     s[1], s[2], s[3], s[4], op, my_ref, e->issuer_counts, e->pairs,
                     e->npairs - 1, e->d_bloom, e->bloom_words ? e->bloom_words - 1 : 0, e->d_result);
  uint32_t res[2];
  HIPCHK(e, hipMemcpyAsync(res, e->d_result, 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (op == 0 && !res[0]) e->arena_used--;  // nothing was inserted (known member, or no room): the cell goes back — the call
                                            // holds e->mu and nothing allocated in between (a host that re-reads a log of known
                                            // certificates through SetInsert leaked 64 bytes per call: advisor, round 4)
  if (res[1]) return fail(e, CTMR_E_FULL, "known-certificate or pair table is full");
  if (op == 0 && res[0]) e->occupied++;
  *out = (int)res[0];
  return CTMR_OK;
}

// Known-certificate table capacity (what Redis does by growing until OOM, storage/rediscache.go:57-65): before a call
// that may claim up to `incoming` slots, make sure the load stays under 3/4 afterwards.  If it would not, the table is
// rebuilt on the GPU (k_rehash) into the smallest power of two that holds (live + incoming) at load <= 1/2 — larger
// when members were added, the SAME size when the slots were eaten by tombstones (expiry sweeps, SetRemove), which a
// rebuild leaves behind.  When even max_slots cannot hold the batch the call fails with CTMR_E_FULL BEFORE anything
// was inserted: a batch is applied completely or not at all.  CONSERVATIVE on purpose: `incoming` counts every entry of
// the batch as a new key (which ones are duplicates is what the call is about to find out), so a capped table
// (max_table_slots) may refuse a batch of mostly-known entries that would have fitted.
// The arena: `incoming` more cells behind arena_used.  Grows by doubling; the cells keep their places (every index word
// stays valid), the old block is copied device to device and freed.  At the START of a round (never inside one: a round's
// own cells are told from older ones by their position) unreferenced cells are squeezed out instead, when that alone makes
// room and at least a quarter of the used cells is garbage (k_arena_compact; `occupied` — the index slots claimed since
// the last rebuild, tombstones included — bounds the live cells from above).
static int ensure_arena(ctmr_engine* e, uint64_t incoming, bool round_start) {
  if (e->arena_used + incoming <= e->arena_cap) return CTMR_OK;
  if (round_start && e->occupied + incoming <= e->arena_cap && e->occupied * 4 <= e->arena_used * 3) {
    KeyCell* na = nullptr;
    if (hipMalloc(&na, e->arena_cap * sizeof(KeyCell)) == hipSuccess) {
      HIPCHK(e, hipMemsetAsync(e->d_count, 0, 8, e->stream));
      hipLaunchKernelGGL(k_arena_compact, dim3((unsigned)((e->nslots + 255) / 256)), dim3(256), 0, e->stream, e->index,
                         e->nslots, (const KeyCell*)e->arena, na, e->d_count);
      unsigned long long live = 0;
      HIPCHK(e, hipMemcpyAsync(&live, e->d_count, 8, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(e, hipStreamSynchronize(e->stream));
      HIPCHK(e, hipGetLastError());
      (void)hipFree(e->arena);
      e->arena = na;
      e->arena_used = live;
      e->arena_compactions++;
      if (e->arena_used + incoming <= e->arena_cap) return CTMR_OK;
    } else {
      (void)hipGetLastError();  // no room for a second arena of this size: fall through to the exact-size growth below
    }
  }
  uint64_t want = e->arena_cap ? e->arena_cap : 1024;
  while (want < e->arena_used + incoming) want *= 2;
  if (want > (1ull << 40)) return fail(e, CTMR_E_FULL, "known-certificate arena: more than 2^40 key cells");
  KeyCell* na = nullptr;
  if (hipMalloc(&na, want * sizeof(KeyCell)) != hipSuccess) {
    (void)hipGetLastError();
    want = e->arena_used + incoming;  // the exact size, when doubling does not fit the device
    if (hipMalloc(&na, want * sizeof(KeyCell)) != hipSuccess) {
      (void)hipGetLastError();
      return fail(e, CTMR_E_NOMEM, "known-certificate arena: cannot allocate %llu key cells", (unsigned long long)want);
    }
  }
  if (e->arena_used) HIPCHK(e, hipMemcpyAsync(na, e->arena, e->arena_used * sizeof(KeyCell), hipMemcpyDeviceToDevice, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  (void)hipFree(e->arena);
  e->arena = na;
  e->arena_cap = want;
  e->arena_growths++;
  return CTMR_OK;
}

// incoming: index slots the call may claim; incoming_cells: arena cells it takes (default: as many) — an owner-computes
// round reserves slots for its own shard AND the received keys behind the map, but its own cells are taken already.
int ensure_capacity(ctmr_engine* e, uint64_t incoming, bool round_start, uint64_t incoming_cells) {
  int ar;
  if ((ar = ensure_arena(e, incoming_cells == ~0ull ? incoming : incoming_cells, round_start))) return ar;  // one cell per entry / received key / point insert
  if ((e->occupied + incoming) * 4 <= e->nslots * 3) return CTMR_OK;
  // how many members are alive decides the new size: count them with the rebuild itself when tombstones may exist
  uint64_t want = pow2_at_least((e->occupied + incoming) * 2);
  if (want > e->max_slots) want = e->max_slots;
  if (want < e->nslots) want = e->nslots;
  // the smallest index that still keeps the 3/4 bound: what to fall back to when the comfortable size (load 1/2, old and
  // new index resident at once) does not fit the device
  uint64_t least = pow2_at_least(((e->occupied + incoming) * 4 + 2) / 3);
  if (least < e->nslots) least = e->nslots;
  for (int attempt = 0; attempt < 3; attempt++) {
    unsigned long long* nt = nullptr;
    if (hipMalloc(&nt, want * 8) != hipSuccess) {
      (void)hipGetLastError();
      if (want > least && least <= e->max_slots) {  // retry at the smallest size that holds the call
        want = least;
        continue;
      }
      return fail(e, CTMR_E_NOMEM, "known-certificate table: cannot allocate %llu slots for the rebuild",
                  (unsigned long long)want);
    }
    HIPCHK(e, hipMemsetAsync(nt, 0, want * 8, e->stream));
    HIPCHK(e, hipMemsetAsync(e->d_count, 0, 8, e->stream));
    hipLaunchKernelGGL(k_rehash, dim3((unsigned)((e->nslots + 255) / 256)), dim3(256), 0, e->stream,
                       (const unsigned long long*)e->index, e->nslots, Table{nt, want - 1, e->arena}, e->d_count);
    unsigned long long live = 0;
    HIPCHK(e, hipMemcpyAsync(&live, e->d_count, 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipGetLastError());
    (void)hipFree(e->index);
    e->index = nt;
    e->nslots = want;
    e->occupied = live;
    e->rebuilds++;
    if ((e->occupied + incoming) * 4 <= e->nslots * 3) return CTMR_OK;
    // the rebuild found fewer tombstones than hoped for and the size was capped: one more try at the cap, else full
    if (want >= e->max_slots) break;
    want = pow2_at_least((e->occupied + incoming) * 2);
    if (want > e->max_slots) want = e->max_slots;
  }
  return fail(e, CTMR_E_FULL, "known-certificate table full: %llu members + %llu incoming do not fit %llu slots "
              "(max_table_slots); nothing of this call was applied", (unsigned long long)e->occupied,
              (unsigned long long)incoming, (unsigned long long)e->nslots);
}

int ensure_pairs(ctmr_engine* e) {
  if (!e->pairs_dirty) return CTMR_OK;
  for (;;) {
    HIPCHK(e, hipMemsetAsync(e->pairs, 0, e->npairs * sizeof(PairSlot), e->stream));
    HIPCHK(e, hipMemsetAsync(e->d_count, 0, 8, e->stream));
    hipLaunchKernelGGL(k_build_pairs, dim3((unsigned)((e->nslots + 255) / 256)), dim3(256), 0, e->stream,
                       e->tbl(), e->pairs, e->npairs - 1, e->d_count);
    unsigned long long full;
    HIPCHK(e, hipMemcpyAsync(&full, e->d_count, 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (!full) break;
    // more distinct (expDate, issuer) sets than slots: the statistics table grows like the member table does
    if (e->npairs >= (1ull << 30))
      return fail(e, CTMR_E_FULL, "(expDate,issuer) table full (%llu slots)", (unsigned long long)e->npairs);
    PairSlot* np = nullptr;
    if (hipMalloc(&np, e->npairs * 4 * sizeof(PairSlot)) != hipSuccess) {
      (void)hipGetLastError();
      return fail(e, CTMR_E_NOMEM, "(expDate,issuer) table: cannot grow past %llu slots", (unsigned long long)e->npairs);
    }
    (void)hipFree(e->pairs);
    e->pairs = np;
    e->npairs *= 4;
  }
  e->pairs_dirty = false;
  return CTMR_OK;
}

// device pairs → list of (key, count)
int dump_pairs(ctmr_engine* e, std::vector<std::pair<unsigned long long, unsigned long long>>* out) {
  { int r0 = ensure_pairs(e); if (r0) return r0; }
  HIPCHK(e, hipMemsetAsync(e->d_count, 0, 8, e->stream));
  size_t cap = 1 << 16;
  for (;;) {
    int r = ensure(e, SC_MISC, cap * 16);
    if (r) return r;
    HIPCHK(e, hipMemsetAsync(e->d_count, 0, 8, e->stream));
    hipLaunchKernelGGL(k_pairs, dim3((unsigned)((e->npairs + 255) / 256)), dim3(256), 0, e->stream,
                       e->pairs, e->npairs, (unsigned long long*)e->d_scratch[SC_MISC],
                       (uint64_t)cap, e->d_count);
    unsigned long long cnt;
    HIPCHK(e, hipMemcpyAsync(&cnt, e->d_count, 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (cnt <= cap) {
      std::vector<unsigned long long> buf(cnt * 2);
      if (cnt) HIPCHK(e, hipMemcpy(buf.data(), e->d_scratch[SC_MISC], cnt * 16, hipMemcpyDeviceToHost));
      out->clear();
      for (size_t i = 0; i < cnt; i++) out->push_back({buf[2 * i], buf[2 * i + 1]});
      return CTMR_OK;
    }
    cap = cnt + 1024;
  }
}

int pair_count(ctmr_engine* e, int32_t exp_hour, uint32_t canon, uint64_t* n) {
  // probe the pair table on the host side through a full dump would be wasteful: read slots
  const unsigned long long key = ((unsigned long long)(canon + 1) << 32) | (uint32_t)exp_hour;
  { int r0 = ensure_pairs(e); if (r0) return r0; }
  uint64_t j = mixk(key) & (e->npairs - 1);
  *n = 0;
  for (uint64_t probes = 0; probes < e->npairs; probes++) {
    PairSlot ps;
    HIPCHK(e, hipMemcpy(&ps, e->pairs + j, sizeof ps, hipMemcpyDeviceToHost));
    if (ps.key == 0) return CTMR_OK;
    if (ps.key == key) { *n = ps.count; return CTMR_OK; }
    j = (j + 1) & (e->npairs - 1);
  }
  return CTMR_OK;
}

void serialise(const std::vector<std::string>& v, uint8_t* out, size_t cap, size_t* need,
               uint64_t* count, int* rc) {
  size_t tot = 0;
  for (auto& s : v) tot += 4 + s.size();
  if (need) *need = tot;
  if (count) *count = v.size();
  if (tot > cap || (!out && tot)) { *rc = CTMR_E_RANGE; return; }
  size_t w = 0;
  for (auto& s : v) {
    uint32_t l = (uint32_t)s.size();
    memcpy(out + w, &l, 4);
    memcpy(out + w + 4, s.data(), l);
    w += 4 + l;
  }
  *rc = CTMR_OK;
}

std::vector<uint32_t> zipf_cdf(uint32_t n) {
  std::vector<uint32_t> cdf(n);
  double hn = 0;
  for (uint32_t k = 1; k <= n; k++) hn += 1.0 / k;
  double acc = 0;
  for (uint32_t k = 0; k < n; k++) {
    acc += 1.0 / (k + 1);
    double t = std::floor(acc / hn * 4294967296.0);
    cdf[k] = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
  }
  return cdf;
}

SynthCfg to_synth(const ctmr_synth_config* c, const uint32_t* cdf) {
  SynthCfg s;
  s.seed = c->seed;
  s.n_issuers = c->n_issuers ? c->n_issuers : 1;
  s.zipf = c->zipf;
  s.dup_permille = c->dup_permille;
  s.ca_permille = c->ca_permille;
  s.expired_permille = c->expired_permille;
  s.mean_len = c->mean_len ? c->mean_len : 1536;
  s.base_time = c->base_time ? c->base_time : 1767225600ll;  // 2026-01-01T00:00:00Z
  s.zipf_cdf = cdf;
  s.profile = c->profile;
  return s;
}

const std::vector<uint32_t>& host_cdf(uint32_t n) {
  static std::mutex mu;
  static std::map<uint32_t, std::vector<uint32_t>> cache;
  std::lock_guard<std::mutex> g(mu);
  auto it = cache.find(n);
  if (it == cache.end()) it = cache.emplace(n, zipf_cdf(n)).first;
  return it->second;
}

}  // namespace

extern "C++" {
namespace {
void pipe_shutdown(ctmr_engine* e);  // engine/pipeline.inc
}
}

extern "C" {

#include "engine/lifecycle.inc"
#include "engine/issuers.inc"
#include "engine/map.inc"
#include "engine/entries.inc"
#include "engine/pipeline.inc"
#include "engine/meta.inc"
#include "engine/pem.inc"
#include "engine/exchange.inc"
#include "engine/sets.inc"
#include "engine/group.inc"
#include "engine/synth.inc"

}  // extern "C"
