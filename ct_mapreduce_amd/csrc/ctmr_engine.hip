// ctmr_engine.hip — libctmr: the C ABI of include/ctmr.h over the gfx950 kernels.
//
// Plain HIP runtime, no torch, no CPU fallback: if the HIP device is unusable every entry
// point fails with CTMR_E_HIP.  Host-side state kept here is only what the reference keeps in
// Redis for keys that are not known-certificate sets (crl::, issuer::, log:: …) plus the
// issuer registry.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
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

struct ctmr_engine {
  std::mutex mu;
  mutable std::string err;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  ctmr_config cfg{};
  // known-certificate table
  Slot* table = nullptr;
  uint64_t nslots = 0;
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
  std::vector<std::string> pending_issuers;  // auto_register off: what the last decode found unregistered
  unsigned long long* d_pend = nullptr;    // PEND_SLOTS claim words
  uint32_t* d_unreg = nullptr;             // UNREG_CAP entry indices
  unsigned long long* d_dcount = nullptr;  // 8 counters of the decode / match kernels
  // IssuerMetadata memo on device (k_meta_new)
  MetaSlot* d_meta_slots = nullptr;
  uint64_t n_meta_slots = 0;
  uint8_t* d_meta_arena = nullptr;
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
  uint32_t bloom_round_epoch = 0;          // epoch of the batch of the current round (0 = empty batch)
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
  void* d_scratch[20] = {};            // growable buffers
  size_t scratch_cap[20] = {};
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

enum { SC_RECORDS = 0, SC_SLOTID, SC_BLKNEW, SC_BLKBASE, SC_ENT, SC_STAGE_A, SC_STAGE_B, SC_MISC, SC_PEM, SC_PEMOFF, SC_TMP, SC_VIEW, SC_ISS_A, SC_ISS_B, SC_ISS_C, SC_META, SC_ITEMS };
constexpr uint32_t UNREG_CAP = 16384;

int fail(const ctmr_engine* e, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) e->err = buf;
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

int point_op(ctmr_engine* e, int op, int32_t exp_hour, uint32_t canon, const uint8_t* m, size_t n,
             int* out) {
  unsigned long long s[5];
  pack_serial(m, n, s);
  const unsigned long long meta = key_meta(exp_hour, canon, (uint32_t)n);
  if (op == 0) e->epoch++;
  if (op != 1) e->pairs_dirty = true;
  hipLaunchKernelGGL(k_set_op, dim3(1), dim3(64), 0, e->stream, e->table, e->nslots - 1, meta, s[0],
                     s[1], s[2], s[3], s[4], op, e->epoch, e->issuer_counts, e->pairs,
                     e->npairs - 1, e->d_result);
  uint32_t res[2];
  HIPCHK(e, hipMemcpyAsync(res, e->d_result, 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (res[1]) return fail(e, CTMR_E_FULL, "known-certificate or pair table is full");
  *out = (int)res[0];
  return CTMR_OK;
}

int ensure_pairs(ctmr_engine* e) {
  if (!e->pairs_dirty) return CTMR_OK;
  HIPCHK(e, hipMemsetAsync(e->pairs, 0, e->npairs * sizeof(PairSlot), e->stream));
  HIPCHK(e, hipMemsetAsync(e->d_count, 0, 8, e->stream));
  hipLaunchKernelGGL(k_build_pairs, dim3((unsigned)((e->nslots + 255) / 256)), dim3(256), 0, e->stream,
                     e->table, e->nslots, e->pairs, e->npairs - 1, e->d_count);
  unsigned long long full;
  HIPCHK(e, hipMemcpyAsync(&full, e->d_count, 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (full) return fail(e, CTMR_E_FULL, "(expDate,issuer) table full (%llu slots)", (unsigned long long)e->npairs);
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

extern "C" {

int ctmr_abi_version(void) { return CTMR_ABI_VERSION; }

const char* ctmr_last_error(const ctmr_engine* e) { return e ? e->err.c_str() : "null engine"; }

int ctmr_create(const ctmr_config* cfg, ctmr_engine** out) {
  if (!cfg || !out || cfg->struct_size != sizeof(ctmr_config)) return CTMR_E_INVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
    return CTMR_E_HIP;  // no GPU → no engine: there is no CPU fallback
  ctmr_engine* e = new ctmr_engine();
  e->cfg = *cfg;
  e->device = cfg->device;
  auto bail = [&](int rc) { std::string m = e->err; ctmr_destroy(e); fprintf(stderr, "ctmr_create: %s\n", m.c_str()); return rc; };
#define CK(call) do { hipError_t _r = (call); if (_r != hipSuccess) { e->err = std::string(#call) + ": " + hipGetErrorString(_r); return bail(_r == hipErrorOutOfMemory ? CTMR_E_NOMEM : CTMR_E_HIP); } } while (0)
  CK(hipSetDevice(e->device));
  CK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  e->own_stream = true;
  e->nslots = pow2_at_least(cfg->table_slots ? cfg->table_slots : (1ull << 24));
  e->npairs = pow2_at_least(cfg->pair_slots ? cfg->pair_slots : (1ull << 22));
  e->max_issuers = cfg->max_issuers ? cfg->max_issuers : 65536;
  if (e->max_issuers > (1u << 24) - 2) { e->err = "max_issuers > 2^24-2"; return bail(CTMR_E_INVAL); }
  if (e->nslots > (1ull << 31)) { e->err = "table_slots > 2^31"; return bail(CTMR_E_INVAL); }
  CK(hipMalloc(&e->table, e->nslots * sizeof(Slot)));
  CK(hipMalloc(&e->pairs, e->npairs * sizeof(PairSlot)));
  CK(hipMalloc(&e->issuer_counts, (size_t)e->max_issuers * 8));
  CK(hipMalloc(&e->d_issuer_valid, e->max_issuers));
  CK(hipMalloc(&e->d_canon, (size_t)e->max_issuers * 4));
  CK(hipMalloc(&e->d_filter, sizeof(FilterDev)));
  CK(hipMalloc(&e->d_stats, sizeof(DevStats)));
  CK(hipMalloc(&e->d_result, 16));
  CK(hipMalloc(&e->d_count, 16));
  e->idb_ht_size = (uint32_t)pow2_at_least((uint64_t)e->max_issuers * 4 < 1024 ? 1024 : (uint64_t)e->max_issuers * 4);
  CK(hipMalloc(&e->d_idb_off, (size_t)e->max_issuers * 8));
  CK(hipMalloc(&e->d_idb_len, (size_t)e->max_issuers * 4));
  CK(hipMalloc(&e->d_idb_ht, (size_t)e->idb_ht_size * 8));
  CK(hipMalloc(&e->d_pend, (size_t)PEND_SLOTS * 8));
  CK(hipMalloc(&e->d_unreg, (size_t)UNREG_CAP * 4));
  CK(hipMalloc(&e->d_dcount, 64));
  if (cfg->collect_meta) {
    e->n_meta_slots = 1ull << 22;  // 128 MB: a load factor near 0.1 keeps nearly every item at the home position of its hash (k_meta_new's fast path)
    e->meta_arena_cap = 64ull << 20;
    CK(hipMalloc(&e->d_meta_slots, e->n_meta_slots * sizeof(MetaSlot)));
    CK(hipMalloc(&e->d_meta_arena, e->meta_arena_cap));
    CK(hipMalloc(&e->d_mcount, 64));
    CK(hipMemsetAsync(e->d_meta_slots, 0, e->n_meta_slots * sizeof(MetaSlot), e->stream));
    CK(hipMemsetAsync(e->d_mcount, 0, 64, e->stream));
  }
  CK(hipMemsetAsync(e->d_idb_ht, 0, (size_t)e->idb_ht_size * 8, e->stream));
  e->h_idb_ht.assign(e->idb_ht_size, 0ull);
  CK(hipMemsetAsync(e->table, 0, e->nslots * sizeof(Slot), e->stream));
  CK(hipMemsetAsync(e->pairs, 0, e->npairs * sizeof(PairSlot), e->stream));
  CK(hipMemsetAsync(e->issuer_counts, 0, (size_t)e->max_issuers * 8, e->stream));
  CK(hipMemsetAsync(e->d_issuer_valid, 0, e->max_issuers, e->stream));
  CK(hipMemsetAsync(e->d_canon, 0, (size_t)e->max_issuers * 4, e->stream));
  for (auto& ev : e->ev) CK(hipEventCreate(&ev));
  memset(&e->h_filter, 0, sizeof e->h_filter);
  e->h_filter.n_pieces = 1;  // strings.Split("", ",") = [""]
  CK(hipMemcpyAsync(e->d_filter, &e->h_filter, sizeof(FilterDev), hipMemcpyHostToDevice, e->stream));
  CK(hipStreamSynchronize(e->stream));
#undef CK
  *out = e;
  return CTMR_OK;
}

void ctmr_destroy(ctmr_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  (void)hipFree(e->table); (void)hipFree(e->pairs); (void)hipFree(e->issuer_counts);
  (void)hipFree(e->d_issuer_valid); (void)hipFree(e->d_canon); (void)hipFree(e->d_filter);
  (void)hipFree(e->d_stats); (void)hipFree(e->d_result); (void)hipFree(e->d_count);
  (void)hipFree(e->d_idb_der); (void)hipFree(e->d_idb_off); (void)hipFree(e->d_idb_len);
  (void)hipFree(e->d_idb_ht); (void)hipFree(e->d_pend); (void)hipFree(e->d_unreg); (void)hipFree(e->d_dcount);
  (void)hipFree(e->d_meta_slots); (void)hipFree(e->d_meta_arena); (void)hipFree(e->d_mcount);
  for (auto p : e->meta_hour_pages) (void)hipFree(p);
  (void)hipFree(e->d_meta_hour_pages);
  if (e->bloom_owned) (void)hipFree(e->d_bloom);
  for (auto p : e->d_scratch) if (p) (void)hipFree(p);
  for (auto ev : e->ev) if (ev) (void)hipEventDestroy(ev);
  if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int ctmr_set_stream(ctmr_engine* e, void* s) {
  if (!e) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (e->own_stream) { (void)hipStreamDestroy(e->stream); e->own_stream = false; }
  e->stream = (hipStream_t)s;
  return CTMR_OK;
}

int ctmr_synchronize(ctmr_engine* e) {
  if (!e) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return CTMR_OK;
}

static int add_issuers_locked(ctmr_engine* e, const uint8_t* der, const uint64_t* offsets, uint32_t n,
                              uint32_t* first_idx) {
  HIPCHK(e, hipSetDevice(e->device));
  const uint32_t first = (uint32_t)e->issuers.size();
  if (first_idx) *first_idx = first;
  if (n == 0) return CTMR_OK;
  if ((uint64_t)first + n > e->max_issuers) return fail(e, CTMR_E_FULL, "issuer table full (%u)", e->max_issuers);
  for (uint32_t i = 0; i < n; i++)
    if (offsets[i + 1] < offsets[i]) return fail(e, CTMR_E_INVAL, "issuer offsets not monotone");
  const uint64_t base = offsets[0], bytes = offsets[n] - base;
  int r;
  if ((r = ensure(e, SC_ISS_A, bytes + CTMR_PAYLOAD_PAD))) return r;
  if ((r = ensure(e, SC_ISS_B, (size_t)(n + 1) * 8))) return r;
  if ((r = ensure(e, SC_ISS_C, (size_t)n * 32 + n))) return r;
  std::vector<uint64_t> rel(n + 1);
  for (uint32_t i = 0; i <= n; i++) rel[i] = offsets[i] - base;
  HIPCHK(e, hipMemcpyAsync(e->d_scratch[SC_ISS_A], der + base, bytes, hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->d_scratch[SC_ISS_B], rel.data(), (n + 1) * 8, hipMemcpyHostToDevice, e->stream));
  uint32_t* d_digest = (uint32_t*)e->d_scratch[SC_ISS_C];
  uint8_t* d_valid = (uint8_t*)e->d_scratch[SC_ISS_C] + (size_t)n * 32;
  hipLaunchKernelGGL(k_issuer_ids, dim3((n + 63) / 64), dim3(64), 0, e->stream,
                     (const uint8_t*)e->d_scratch[SC_ISS_A], (const uint64_t*)e->d_scratch[SC_ISS_B],
                     n, d_valid, d_digest);
  std::vector<uint32_t> dg(n * 8);
  std::vector<uint8_t> valid(n);
  HIPCHK(e, hipMemcpyAsync(dg.data(), d_digest, (size_t)n * 32, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipMemcpyAsync(valid.data(), d_valid, n, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  std::vector<uint32_t> canon(n);
  for (uint32_t i = 0; i < n; i++) {
    IssuerRec rec;
    rec.valid = valid[i] != 0;
    for (int k = 0; k < 8; k++) {
      uint32_t w = dg[i * 8 + k];
      rec.digest[4 * k] = w >> 24; rec.digest[4 * k + 1] = w >> 16;
      rec.digest[4 * k + 2] = w >> 8; rec.digest[4 * k + 3] = w;
    }
    rec.canon = first + i;
    if (rec.valid) {
      rec.id = b64url(rec.digest, 32);
      auto it = e->id_to_canon.find(rec.id);
      if (it != e->id_to_canon.end()) rec.canon = it->second;
      else e->id_to_canon.emplace(rec.id, rec.canon);
    }
    canon[i] = rec.canon;
    e->issuers.push_back(rec);
  }
  HIPCHK(e, hipMemcpyAsync(e->d_issuer_valid + first, valid.data(), n, hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->d_canon + first, canon.data(), (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  // ---- certificate store for the Chain[0] match (raw-entry path)
  {
    std::vector<uint64_t> off(n);
    std::vector<uint32_t> len(n);
    size_t need = e->idb_used;
    for (uint32_t i = 0; i < n; i++) {
      off[i] = need;
      len[i] = (uint32_t)(offsets[i + 1] - offsets[i]);
      need += ((size_t)len[i] + 15) / 16 * 16 + 16;
    }
    if (need > e->idb_cap) {
      const size_t cap = std::max(need * 2, (size_t)1 << 20);
      uint8_t* nb = nullptr;
      HIPCHK(e, hipMalloc(&nb, cap));
      HIPCHK(e, hipMemsetAsync(nb, 0, cap, e->stream));
      if (e->idb_used) HIPCHK(e, hipMemcpyAsync(nb, e->d_idb_der, e->idb_used, hipMemcpyDeviceToDevice, e->stream));
      HIPCHK(e, hipStreamSynchronize(e->stream));
      (void)hipFree(e->d_idb_der);
      e->d_idb_der = nb;
      e->idb_cap = cap;
    }
    bool ht_dirty = false;
    for (uint32_t i = 0; i < n; i++) {
      const uint8_t* c = der + offsets[i];
      if (len[i]) HIPCHK(e, hipMemcpyAsync(e->d_idb_der + off[i], c, len[i], hipMemcpyHostToDevice, e->stream));
      const unsigned long long qh = cert_quick_hash(HostBytes{c}, 0, len[i]);
      if (len[i] && e->der_to_idx.emplace(std::string((const char*)c, len[i]), first + i).second) {
        uint32_t j = (uint32_t)qh & (e->idb_ht_size - 1);
        while (e->h_idb_ht[j]) j = (j + 1) & (e->idb_ht_size - 1);
        e->h_idb_ht[j] = (qh & 0xffffffff00000000ull) | (unsigned long long)(first + i + 1);
        ht_dirty = true;
      }
    }
    e->idb_used = need;
    HIPCHK(e, hipMemcpyAsync(e->d_idb_off + first, off.data(), (size_t)n * 8, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipMemcpyAsync(e->d_idb_len + first, len.data(), (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
    if (ht_dirty)
      HIPCHK(e, hipMemcpyAsync(e->d_idb_ht, e->h_idb_ht.data(), (size_t)e->idb_ht_size * 8, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
  }
  return CTMR_OK;
}

int ctmr_alloc_pinned(ctmr_engine* e, size_t bytes, void** out) {
  if (!e || !out) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  *out = nullptr;
  HIPCHK(e, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return CTMR_OK;
}

int ctmr_free_pinned(ctmr_engine* e, void* p) {
  if (!e) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  if (p) HIPCHK(e, hipHostFree(p));
  return CTMR_OK;
}

int ctmr_add_issuers(ctmr_engine* e, const uint8_t* der, const uint64_t* offsets, uint32_t n,
                     uint32_t* first_idx) {
  if (!e || (n && (!der || !offsets))) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  return add_issuers_locked(e, der, offsets, n, first_idx);
}

int ctmr_sha256(ctmr_engine* e, const uint8_t* data, size_t len, uint8_t out[32]) {
  if (!e || !out || (len && !data) || len > 0x7fffff00u) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  int r;
  if ((r = ensure(e, SC_TMP, len + 128))) return r;
  uint8_t* d = (uint8_t*)e->d_scratch[SC_TMP];
  uint32_t* d_dg = (uint32_t*)(d + ((len + 3) & ~(size_t)3) + 32);
  HIPCHK(e, hipMemsetAsync(d + (len & ~(size_t)3), 0, 8, e->stream));  // the reader loads whole words
  if (len) HIPCHK(e, hipMemcpyAsync(d, data, len, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_sha256_one, dim3(1), dim3(64), 0, e->stream, (const uint8_t*)d, (uint32_t)len, d_dg);
  uint32_t dg[8];
  HIPCHK(e, hipMemcpyAsync(dg, d_dg, 32, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  for (int k = 0; k < 8; k++) {
    out[4 * k] = (uint8_t)(dg[k] >> 24); out[4 * k + 1] = (uint8_t)(dg[k] >> 16);
    out[4 * k + 2] = (uint8_t)(dg[k] >> 8); out[4 * k + 3] = (uint8_t)dg[k];
  }
  return CTMR_OK;
}

int ctmr_issuer_count(ctmr_engine* e, uint32_t* n) {
  if (!e || !n) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  *n = (uint32_t)e->issuers.size();
  return CTMR_OK;
}

int ctmr_issuer_info_get(ctmr_engine* e, uint32_t idx, ctmr_issuer_info* out) {
  if (!e || !out) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  if (idx >= e->issuers.size()) return fail(e, CTMR_E_NOTFOUND, "issuer %u not registered", idx);
  const IssuerRec& r = e->issuers[idx];
  memset(out, 0, sizeof *out);
  out->valid = r.valid;
  out->canonical_idx = r.canon;
  memcpy(out->spki_sha256, r.digest, 32);
  snprintf(out->issuer_id, sizeof out->issuer_id, "%s", r.id.c_str());
  return CTMR_OK;
}

int ctmr_set_filter(ctmr_engine* e, const char* filter, size_t len, int log_expired, int64_t now) {
  if (!e || (len && !filter)) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  FilterDev f;
  memset(&f, 0, sizeof f);
  f.active = len != 0;
  f.log_expired = log_expired != 0;
  f.now = now;
  // strings.Split(filter, ","): pieces are NOT trimmed (ct-fetch.go:58)
  size_t s = 0;
  uint32_t nw = 0;
  for (;;) {
    size_t t = s;
    while (t < len && filter[t] != ',') t++;
    const size_t pl = t - s;
    if (f.n_pieces >= 64 || nw + (pl + 3) / 4 > 1024)
      return fail(e, CTMR_E_INVAL, "issuerCNFilter too large (max 64 pieces / 4096 bytes)");
    f.piece_len[f.n_pieces] = (uint32_t)pl;
    f.piece_word[f.n_pieces] = nw;
    memcpy((uint8_t*)(f.words + nw), filter + s, pl);
    nw += (uint32_t)((pl + 3) / 4);
    f.n_pieces++;
    if (t >= len) break;
    s = t + 1;
  }
  e->h_filter = f;
  return upload_filter(e);
}

static int launch_map(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets,
                      const uint32_t* d_issuer_idx, const uint8_t* d_entry_type, uint64_t n,
                      ctmr_record* d_records, bool optimistic_new, const InsertArgs* fuse = nullptr,
                      const uint64_t* d_ends = nullptr, uint64_t limit = 0) {
  MapArgs ma;
  ma.optimistic_new = optimistic_new ? 1u : 0u;
  ma.ends = d_ends; ma.limit = limit;
  ma.meta_loc = nullptr;
  if (e->cfg.collect_meta) {
    int mr;
    if ((mr = ensure(e, SC_META, n * 8))) return mr;
    ma.meta_loc = (uint2*)e->d_scratch[SC_META];
    e->meta_n = n;
    e->last_meta_valid = false;
  }
  ma.payload = d_payload; ma.offsets = d_offsets; ma.issuer_idx = d_issuer_idx;
  ma.entry_type = d_entry_type; ma.records = d_records; ma.issuer_valid = e->d_issuer_valid;
  ma.filt = e->d_filter; ma.n = n; ma.n_issuers = (uint32_t)e->issuers.size();
  uint32_t variant = e->cfg.map_variant ? e->cfg.map_variant : 15;
  if (variant == 14 && !fuse) variant = 13;  // the fused kernels only exist with the local reduce behind them
  if (variant == 1 && d_ends) variant = 13;  // the whole-certificate tile copy needs the packed layout
  uint32_t C = e->cfg.certs_per_tile ? e->cfg.certs_per_tile : 32;
  if (C > 64) C = 64;
  uint32_t lds = e->cfg.lds_tile_bytes ? e->cfg.lds_tile_bytes : 65536;
  if (lds > 160 * 1024) lds = 160 * 1024;
  ma.certs_per_tile = C; ma.lds_bytes = lds;
  if (variant == 15 && !fuse) variant = 13;
  if (variant == 14) {
    hipLaunchKernelGGL((k_map_fused<16, false>), dim3((unsigned)((n + 63) / 64)), dim3(64), 64 * (16 * 16 + 16), e->stream, ma, *fuse);
  } else if (variant == 15) {
    hipLaunchKernelGGL((k_map_fused<16, true>), dim3((unsigned)((n + 63) / 64)), dim3(64), 64 * (16 * 16 + 16), e->stream, ma, *fuse);
  } else if (variant == 2) {
    hipLaunchKernelGGL(k_map_direct, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, ma);
  } else if (variant == 3) {
    hipLaunchKernelGGL(k_map_win<16>, dim3((unsigned)((n + 63) / 64)), dim3(64), 64 * (16 * 16 + 16), e->stream, ma);
  } else if (variant == 4) {
    hipLaunchKernelGGL(k_map_win<8>, dim3((unsigned)((n + 63) / 64)), dim3(64), 64 * (8 * 16 + 16), e->stream, ma);
  } else if (variant == 5) {
    hipLaunchKernelGGL(k_map_win<12>, dim3((unsigned)((n + 63) / 64)), dim3(64), 64 * (12 * 16 + 16), e->stream, ma);
  } else if (variant == 6) {
    hipLaunchKernelGGL(k_map_win<14>, dim3((unsigned)((n + 63) / 64)), dim3(64), 64 * (14 * 16 + 16), e->stream, ma);
  } else if (variant == 13) {
    hipLaunchKernelGGL(k_map_winc<16>, dim3((unsigned)((n + 63) / 64)), dim3(64), 64 * (16 * 16 + 16), e->stream, ma);
  } else if (variant == 10) {
    hipLaunchKernelGGL((k_map_wint<16, 192, 208>), dim3((unsigned)((n + 63) / 64)), dim3(64), 64 * (16 * 16 + 16), e->stream, ma);
  } else if (variant == 11) {
    hipLaunchKernelGGL((k_map_wint<16, 208, 224>), dim3((unsigned)((n + 63) / 64)), dim3(64), 64 * (16 * 16 + 16), e->stream, ma);
  } else if (variant == 12) {
    hipLaunchKernelGGL((k_map_wint<16, 176, 192>), dim3((unsigned)((n + 63) / 64)), dim3(64), 64 * (16 * 16 + 16), e->stream, ma);
  } else if (variant == 7) {
    hipLaunchKernelGGL(k_map_win2<16>, dim3((unsigned)((n + 63) / 64)), dim3(64), 64 * (19 * 16), e->stream, ma);
  } else if (variant == 8) {
    hipLaunchKernelGGL(k_map_win2<14>, dim3((unsigned)((n + 63) / 64)), dim3(64), 64 * (17 * 16), e->stream, ma);
  } else if (variant == 9) {
    hipLaunchKernelGGL(k_map_win2<12>, dim3((unsigned)((n + 63) / 64)), dim3(64), 64 * (15 * 16), e->stream, ma);
  } else {
    static bool attr_set = false;
    if (!attr_set) {
      HIPCHK(e, hipFuncSetAttribute((const void*)k_map_tile, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set = true;
    }
    const uint64_t tiles = (n + C - 1) / C;
    hipLaunchKernelGGL(k_map_tile, dim3((unsigned)tiles), dim3(64), lds, e->stream, ma);
  }
  return CTMR_OK;
}

static int map_device_locked(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets,
                             const uint32_t* d_issuer_idx, const uint8_t* d_entry_type, uint64_t n,
                             ctmr_record* d_records, uint64_t* d_new_idx, ctmr_batch_stats* stats,
                             const uint64_t* d_ends = nullptr, uint64_t blob_bytes = 0,
                             uint64_t payload_bytes_known = ~0ull) {
  HIPCHK(e, hipSetDevice(e->device));
  if (stats) memset(stats, 0, sizeof *stats);
  if (n == 0) return CTMR_OK;
  if (n >= 0xfffffff0ull) return fail(e, CTMR_E_INVAL, "batch too large (n < 2^32-16)");
  if (((uintptr_t)d_payload & 15) != 0) return fail(e, CTMR_E_INVAL, "payload must be 16-byte aligned");
  int r;
  if (!d_records) {
    if ((r = ensure(e, SC_RECORDS, n * sizeof(ctmr_record)))) return r;
    d_records = (ctmr_record*)e->d_scratch[SC_RECORDS];
  }
  const uint64_t nb = (n + 1023) / 1024;
  if ((r = ensure(e, SC_SLOTID, n * 4))) return r;
  if ((r = ensure(e, SC_ENT, n * 4))) return r;
  if ((r = ensure(e, SC_BLKNEW, nb * 4))) return r;
  if ((r = ensure(e, SC_BLKBASE, nb * 8))) return r;
  uint32_t* d_slot = (uint32_t*)e->d_scratch[SC_SLOTID];
  uint32_t* d_ent = (uint32_t*)e->d_scratch[SC_ENT];
  uint32_t* d_blk_new = (uint32_t*)e->d_scratch[SC_BLKNEW];
  uint64_t* d_blk_base = (uint64_t*)e->d_scratch[SC_BLKBASE];
  const bool prof = e->cfg.profile != 0;
  e->epoch++;
  e->pairs_dirty = true;
  HIPCHK(e, hipMemsetAsync(e->d_stats, 0, sizeof(DevStats), e->stream));

  // ---- map (PASS records leave it with WAS_UNKNOWN set; the reduce clears it for duplicates)
  InsertArgs ia;
  ia.records = d_records; ia.payload = d_payload; ia.offsets = d_offsets; ia.ends = d_ends; ia.canon = e->d_canon;
  ia.table = e->table; ia.mask = e->nslots - 1; ia.slot_id = d_slot; ia.ent = d_ent; ia.n = n; ia.epoch = e->epoch;
  const bool fused = e->cfg.map_variant == 0 || e->cfg.map_variant == 14 || e->cfg.map_variant == 15;
  if (prof) HIPCHK(e, hipEventRecord(e->ev[0], e->stream));
  if ((r = launch_map(e, d_payload, d_offsets, d_issuer_idx, d_entry_type, n, d_records, true, fused ? &ia : nullptr,
                      d_ends, blob_bytes + CTMR_PAYLOAD_PAD))) return r;
  if (prof) HIPCHK(e, hipEventRecord(e->ev[1], e->stream));
  // ---- insert (pass 1 ran inside the map kernel when fused)
  if (!fused) hipLaunchKernelGGL(k_insert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, ia);
  hipLaunchKernelGGL(k_insert2, dim3((unsigned)((n + INSERT2_PER_BLOCK - 1) / INSERT2_PER_BLOCK)), dim3(256), 0, e->stream, ia, d_records);
  if (prof) HIPCHK(e, hipEventRecord(e->ev[2], e->stream));
  // ---- resolve
  ResolveArgs ra;
  ra.ent = d_ent; ra.issuer_counts = e->issuer_counts; ra.stats = e->d_stats; ra.blk_new = d_blk_new; ra.n = n;
  hipLaunchKernelGGL(k_resolve, dim3((unsigned)(nb < 512 ? nb : 512)), dim3(1024), 0, e->stream, ra, nb);
  if (prof) HIPCHK(e, hipEventRecord(e->ev[3], e->stream));
  // ---- compaction of the NEW list, queued right behind the resolve: the common case needs ONE synchronisation per
  //      batch (a host that feeds 1 001-entry batches — one get-entries response — pays for every extra round trip)
  auto compact = [&]() {
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, e->stream, d_blk_new, nb, d_blk_base);
    hipLaunchKernelGGL(k_compact, dim3((unsigned)nb), dim3(256), 0, e->stream, (const ctmr_record*)d_records, (const uint32_t*)d_ent, n, (const uint64_t*)d_blk_base, d_new_idx);
  };
  if (d_new_idx) compact();
  if (prof) HIPCHK(e, hipEventRecord(e->ev[4], e->stream));
  DevStats hs;
  uint64_t ends[2] = {0, blob_bytes};
  HIPCHK(e, hipMemcpyAsync(&hs, e->d_stats, sizeof hs, hipMemcpyDeviceToHost, e->stream));
  if (!d_ends && payload_bytes_known == ~0ull) {
    HIPCHK(e, hipMemcpyAsync(&ends[0], d_offsets, 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipMemcpyAsync(&ends[1], d_offsets + n, 8, hipMemcpyDeviceToHost, e->stream));
  } else if (!d_ends) {
    ends[1] = payload_bytes_known;
  }
  HIPCHK(e, hipStreamSynchronize(e->stream));
  HIPCHK(e, hipGetLastError());
  if (hs.n_full) return fail(e, CTMR_E_FULL, "known-certificate table full (%llu slots): %llu entries dropped",
                             (unsigned long long)e->nslots, hs.n_full);

  // ---- serials longer than CTMR_MAX_SERIAL: exact host-side set, in log order (rare: the NEW list is rebuilt)
  uint64_t host_new = 0;
  if (hs.n_host) {
    std::vector<uint32_t> ent(n);
    HIPCHK(e, hipMemcpy(ent.data(), d_ent, n * 4, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n; i++) {
      if (((ent[i] >> 3) & 7u) != ES_HOST) continue;
      uint64_t off[2];
      ctmr_record rec;
      HIPCHK(e, hipMemcpy(off, d_offsets + i, d_ends ? 8 : 16, hipMemcpyDeviceToHost));
      if (d_ends) HIPCHK(e, hipMemcpy(off + 1, d_ends + i, 8, hipMemcpyDeviceToHost));
      HIPCHK(e, hipMemcpy(&rec, d_records + i, sizeof rec, hipMemcpyDeviceToHost));
      std::vector<uint8_t> der(off[1] - off[0] + 32);
      HIPCHK(e, hipMemcpy(der.data(), d_payload + off[0], off[1] - off[0], hipMemcpyDeviceToHost));
      HostReader hr{der.data()};
      Walk w;
      if (!walk_cert(hr, (uint32_t)(off[1] - off[0]), w)) continue;  // cannot happen: map accepted it
      const uint32_t canon = e->issuers[rec.issuer_idx].canon;
      std::string key = make_key(e, rec.exp_hour, canon);
      std::string member((const char*)der.data() + w.serial_off, w.serial_len);
      if (e->hstore[key].insert(member).second) {
        host_new++;
        e->host_issuer_counts[canon]++;
        uint8_t fl = rec.flags | CTMR_FL_WAS_UNKNOWN;
        HIPCHK(e, hipMemcpy((uint8_t*)(d_records + i) + 1, &fl, 1, hipMemcpyHostToDevice));
        const uint32_t en = (ent[i] & ~(7u << 3)) | (ES_CLAIMED << 3);  // compaction reads ent[]
        HIPCHK(e, hipMemcpy(d_ent + i, &en, 4, hipMemcpyHostToDevice));
        uint32_t bn;
        HIPCHK(e, hipMemcpy(&bn, d_blk_new + i / 1024, 4, hipMemcpyDeviceToHost));
        bn++;
        HIPCHK(e, hipMemcpy(d_blk_new + i / 1024, &bn, 4, hipMemcpyHostToDevice));
      }
    }
    if (host_new && d_new_idx) {
      compact();
      HIPCHK(e, hipStreamSynchronize(e->stream));
      HIPCHK(e, hipGetLastError());
    }
  }
  if (stats) {
    stats->n = n;
    for (int k = 0; k < CTMR_ST__COUNT; k++) stats->by_status[k] = hs.by_status[k];
    stats->n_new = hs.n_new + host_new;
    stats->n_dup = hs.n_dup + (hs.n_host - host_new);
    stats->n_host_set = hs.n_host;
    stats->payload_bytes = ends[1] - ends[0];  // entry view: the whole blob (leaf_input + extra_data)
    stats->map_launches = 1;
    if (prof) {
      (void)hipEventElapsedTime(&stats->ms_map, e->ev[0], e->ev[1]);
      (void)hipEventElapsedTime(&stats->ms_insert, e->ev[1], e->ev[2]);
      (void)hipEventElapsedTime(&stats->ms_resolve, e->ev[2], e->ev[3]);
      (void)hipEventElapsedTime(&stats->ms_compact, e->ev[3], e->ev[4]);
      (void)hipEventElapsedTime(&stats->ms_total, e->ev[0], e->ev[4]);
    }
  }
  return CTMR_OK;
}

int ctmr_map_batch_device(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets,
                          const uint32_t* d_issuer_idx, const uint8_t* d_entry_type, uint64_t n,
                          ctmr_record* d_records, uint64_t* d_new_idx, ctmr_batch_stats* stats) {
  if (!e || (n && (!d_payload || !d_offsets || !d_issuer_idx))) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  return map_device_locked(e, d_payload, d_offsets, d_issuer_idx, d_entry_type, n, d_records, d_new_idx, stats);
}

int ctmr_map_batch(ctmr_engine* e, const uint8_t* payload, const uint64_t* offsets,
                   const uint32_t* issuer_idx, const uint8_t* entry_type, uint64_t n,
                   ctmr_record* records, uint64_t* new_idx, ctmr_batch_stats* stats) {
  if (!e || (n && (!payload || !offsets || !issuer_idx))) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (stats) memset(stats, 0, sizeof *stats);
  if (n == 0) return CTMR_OK;
  for (uint64_t i = 0; i < n; i++)
    if (offsets[i + 1] < offsets[i]) return fail(e, CTMR_E_INVAL, "offsets not monotone at %llu", (unsigned long long)i);
  const uint64_t base = offsets[0], bytes = offsets[n] - base;
  // staging layout in SC_STAGE_A: payload | pad ; SC_STAGE_B: offsets | issuer_idx | entry_type | new_idx
  int r;
  if ((r = ensure(e, SC_STAGE_A, bytes + CTMR_PAYLOAD_PAD + 16))) return r;
  const size_t o_off = 0, o_iss = (n + 1) * 8, o_et = o_iss + n * 4, o_new = (o_et + n + 15) & ~(size_t)15;
  if ((r = ensure(e, SC_STAGE_B, o_new + n * 8))) return r;
  uint8_t* B = (uint8_t*)e->d_scratch[SC_STAGE_B];
  // rel[] lives in the engine: the uploads are ordered before the kernels on the stream and every caller buffer
  // is consumed before this call returns (the call ends with a synchronisation), so no sync is needed here
  std::vector<uint64_t>& rel = e->h_rel;
  rel.resize(n + 1);
  for (uint64_t i = 0; i <= n; i++) rel[i] = offsets[i] - base;
  HIPCHK(e, hipMemcpyAsync(e->d_scratch[SC_STAGE_A], payload + base, bytes, hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(B + o_off, rel.data(), (n + 1) * 8, hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(B + o_iss, issuer_idx, n * 4, hipMemcpyHostToDevice, e->stream));
  if (entry_type) HIPCHK(e, hipMemcpyAsync(B + o_et, entry_type, n, hipMemcpyHostToDevice, e->stream));
  ctmr_batch_stats st;
  r = map_device_locked(e, (const uint8_t*)e->d_scratch[SC_STAGE_A], (const uint64_t*)(B + o_off),
                        (const uint32_t*)(B + o_iss), entry_type ? B + o_et : nullptr, n, nullptr,
                        new_idx ? (uint64_t*)(B + o_new) : nullptr, &st, nullptr, 0, bytes);
  if (r) return r;
  if (records) HIPCHK(e, hipMemcpyAsync(records, e->d_scratch[SC_RECORDS], n * sizeof(ctmr_record), hipMemcpyDeviceToHost, e->stream));
  if (new_idx && st.n_new) HIPCHK(e, hipMemcpyAsync(new_idx, B + o_new, st.n_new * 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (stats) *stats = st;
  e->last_n = n; e->last_n_new = new_idx ? st.n_new : 0; e->last_o_off = o_off; e->last_o_new = o_new;
  e->last_is_view = false;
  return CTMR_OK;
}

// ------------------------------------------------------------------ CT get-entries decode (N2)

static int decode_locked(ctmr_engine* e, const uint8_t* d_blob, const uint64_t* d_bounds, uint64_t n,
                         const ctmr_entry_view* v, ctmr_decode_stats* stats) {
  HIPCHK(e, hipSetDevice(e->device));
  if (stats) memset(stats, 0, sizeof *stats);
  e->pending_issuers.clear();
  if (n == 0) return CTMR_OK;
  if (n >= 0xfffffff0ull) return fail(e, CTMR_E_INVAL, "batch too large (n < 2^32-16)");
  if (!v->cert_start || !v->cert_end || !v->issuer_idx || !v->entry_type)
    return fail(e, CTMR_E_INVAL, "entry view: cert_start, cert_end, issuer_idx and entry_type are required");
  int r;
  uint64_t* c0s = v->chain0_start;
  uint32_t* c0l = v->chain0_len;
  if (!c0s || !c0l) {
    if ((r = ensure(e, SC_VIEW, n * 12 + 64))) return r;
    if (!c0s) c0s = (uint64_t*)e->d_scratch[SC_VIEW];
    if (!c0l) c0l = (uint32_t*)((uint8_t*)e->d_scratch[SC_VIEW] + n * 8);
  }
  const bool prof = e->cfg.profile != 0;
  HIPCHK(e, hipMemsetAsync(e->d_dcount, 0, 64, e->stream));
  HIPCHK(e, hipMemsetAsync(e->d_pend, 0, (size_t)PEND_SLOTS * 8, e->stream));
  DecodeArgs da;
  da.blob = d_blob; da.bounds = d_bounds; da.n = n; da.cert_start = v->cert_start; da.cert_end = v->cert_end;
  da.entry_type = v->entry_type; da.timestamp = v->timestamp; da.chain0_start = c0s; da.chain0_len = c0l;
  da.counters = e->d_dcount;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (prof) HIPCHK(e, hipEventRecord(e->ev[5], e->stream));
  hipLaunchKernelGGL(k_entry_decode, dim3((unsigned)((n + DECODE_PER_BLOCK - 1) / DECODE_PER_BLOCK)), dim3(256), 0, e->stream, da);
  if (prof) HIPCHK(e, hipEventRecord(e->ev[6], e->stream));
  MatchArgs ma;
  ma.blob = d_blob; ma.chain0_start = c0s; ma.chain0_len = c0l; ma.entry_type = v->entry_type;
  ma.issuer_idx = v->issuer_idx; ma.n = n;
  ma.ht_mask = e->idb_ht_size - 1; ma.retry = 0;
  ma.pend = e->d_pend; ma.unreg_list = e->d_unreg; ma.unreg_cap = UNREG_CAP; ma.counters = e->d_dcount + 4;
  uint64_t added = 0;
  unsigned long long hc[8];
  for (int round = 0;; round++) {
    ma.idb_der = e->d_idb_der; ma.idb_off = e->d_idb_off; ma.idb_len = e->d_idb_len;
    ma.ht = e->d_idb_ht;
    hipLaunchKernelGGL(k_chain0_match, dim3(blocks), dim3(256), 0, e->stream, ma);
    if (round == 0 && prof) HIPCHK(e, hipEventRecord(e->ev[7], e->stream));
    HIPCHK(e, hipMemcpyAsync(hc, e->d_dcount, 64, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipGetLastError());
    if (hc[4] == 0) break;  // every Chain[0] is registered
    if (round > 64) return fail(e, CTMR_E_HIP, "Chain[0] registration does not converge");
    // ---- register the distinct unknown Chain[0] certificates (x509.ParseCertificate(Chain[0]) + NewIssuer once each)
    const uint32_t nl = (uint32_t)std::min<unsigned long long>(hc[5], UNREG_CAP);
    std::vector<uint32_t> list(nl);
    HIPCHK(e, hipMemcpy(list.data(), e->d_unreg, (size_t)nl * 4, hipMemcpyDeviceToHost));
    std::sort(list.begin(), list.end());  // ascending log index: registration order is deterministic
    std::vector<uint8_t> blob;
    std::vector<uint64_t> off{0};
    std::set<std::string> seen;
    for (uint32_t k = 0; k < nl; k++) {
      uint64_t lo;
      uint32_t len;
      HIPCHK(e, hipMemcpy(&lo, c0s + list[k], 8, hipMemcpyDeviceToHost));
      HIPCHK(e, hipMemcpy(&len, c0l + list[k], 4, hipMemcpyDeviceToHost));
      std::string der(len, '\0');
      HIPCHK(e, hipMemcpy(&der[0], d_blob + lo, len, hipMemcpyDeviceToHost));
      if (e->der_to_idx.count(der) || !seen.insert(der).second) continue;
      blob.insert(blob.end(), der.begin(), der.end());
      off.push_back(blob.size());
    }
    const uint32_t fresh = (uint32_t)off.size() - 1;
    if (fresh == 0) return fail(e, CTMR_E_HIP, "Chain[0] match reported unregistered certificates but none is new");
    if (!e->auto_register) {
      e->pending_issuers.clear();
      for (uint32_t k = 0; k < fresh; k++)
        e->pending_issuers.emplace_back((const char*)blob.data() + off[k], (size_t)(off[k + 1] - off[k]));
      return fail(e, CTMR_E_NOTFOUND, "%u Chain[0] certificate(s) are not registered (issuer auto-registration is off: "
                  "ctmr_pending_issuers, ctmr_add_issuers, then call again)", fresh);
    }
    blob.resize(blob.size() + CTMR_PAYLOAD_PAD);
    if ((r = add_issuers_locked(e, blob.data(), off.data(), fresh, nullptr))) return r;
    added += fresh;
    HIPCHK(e, hipMemsetAsync(e->d_dcount + 4, 0, 32, e->stream));
    HIPCHK(e, hipMemsetAsync(e->d_pend, 0, (size_t)PEND_SLOTS * 8, e->stream));
    ma.retry = 1;
  }
  if (stats) {
    stats->n = n;
    stats->n_x509 = hc[0]; stats->n_precert = hc[1]; stats->n_decode_error = hc[2]; stats->n_no_chain = hc[3];
    stats->n_issuers_added = added;
    uint64_t b[2];
    HIPCHK(e, hipMemcpy(&b[0], d_bounds, 8, hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(&b[1], d_bounds + 2 * n, 8, hipMemcpyDeviceToHost));
    stats->blob_bytes = b[1] - b[0];
    if (prof) {
      (void)hipEventElapsedTime(&stats->ms_decode, e->ev[5], e->ev[6]);
      (void)hipEventElapsedTime(&stats->ms_match, e->ev[6], e->ev[7]);
    }
  }
  return CTMR_OK;
}

int ctmr_decode_entries_device(ctmr_engine* e, const uint8_t* d_blob, const uint64_t* d_bounds, uint64_t n,
                               const ctmr_entry_view* d_view, ctmr_decode_stats* stats) {
  if (!e || !d_view || (n && (!d_blob || !d_bounds))) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  return decode_locked(e, d_blob, d_bounds, n, d_view, stats);
}

int ctmr_map_view_device(ctmr_engine* e, const uint8_t* d_blob, uint64_t blob_bytes, const ctmr_entry_view* v,
                         uint64_t n, ctmr_record* d_records, uint64_t* d_new_idx, ctmr_batch_stats* stats) {
  if (!e || !v || (n && (!d_blob || !v->cert_start || !v->cert_end || !v->issuer_idx))) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  return map_device_locked(e, d_blob, v->cert_start, v->issuer_idx, v->entry_type, n, d_records, d_new_idx, stats,
                           v->cert_end, blob_bytes);
}

// view arrays in SC_VIEW behind the chain0 scratch: start | end | issuer_idx | entry_type
static int view_in_scratch(ctmr_engine* e, uint64_t n, ctmr_entry_view* v, uint64_t* d_timestamp) {
  int r;
  const size_t o_c0 = 0, o_start = (n * 12 + 63) & ~(size_t)63, o_end = o_start + n * 8, o_iss = o_end + n * 8,
               o_et = o_iss + n * 4;
  if ((r = ensure(e, SC_VIEW, o_et + n + 64))) return r;
  uint8_t* V = (uint8_t*)e->d_scratch[SC_VIEW];
  v->chain0_start = (uint64_t*)(V + o_c0);
  v->chain0_len = (uint32_t*)(V + o_c0 + n * 8);
  v->cert_start = (uint64_t*)(V + o_start);
  v->cert_end = (uint64_t*)(V + o_end);
  v->issuer_idx = (uint32_t*)(V + o_iss);
  v->entry_type = V + o_et;
  v->timestamp = d_timestamp;
  return CTMR_OK;
}

static int map_entries_locked(ctmr_engine* e, const uint8_t* d_blob, const uint64_t* d_bounds, uint64_t n,
                              ctmr_record* d_records, uint64_t* d_new_idx, uint64_t* d_timestamp,
                              ctmr_decode_stats* dstats, ctmr_batch_stats* stats, ctmr_entry_view* view_out) {
  if (stats) memset(stats, 0, sizeof *stats);
  if (dstats) memset(dstats, 0, sizeof *dstats);
  if (n == 0) return CTMR_OK;
  ctmr_entry_view v;
  int r;
  if ((r = view_in_scratch(e, n, &v, d_timestamp))) return r;
  ctmr_decode_stats ds;
  if ((r = decode_locked(e, d_blob, d_bounds, n, &v, &ds))) return r;
  if (dstats) *dstats = ds;
  if (view_out) *view_out = v;
  return map_device_locked(e, d_blob, v.cert_start, v.issuer_idx, v.entry_type, n, d_records, d_new_idx, stats,
                           v.cert_end, ds.blob_bytes);
}

int ctmr_map_entries_device(ctmr_engine* e, const uint8_t* d_blob, const uint64_t* d_bounds, uint64_t n,
                            ctmr_record* d_records, uint64_t* d_new_idx, uint64_t* d_timestamp,
                            ctmr_decode_stats* dstats, ctmr_batch_stats* stats) {
  if (!e || (n && (!d_blob || !d_bounds))) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  if (n) {
    uint64_t b0 = 0;
    HIPCHK(e, hipMemcpy(&b0, d_bounds, 8, hipMemcpyDeviceToHost));
    if (b0 != 0) return fail(e, CTMR_E_INVAL, "bounds[0] must be 0 (bounds are relative to d_blob)");
  }
  return map_entries_locked(e, d_blob, d_bounds, n, d_records, d_new_idx, d_timestamp, dstats, stats, nullptr);
}

int ctmr_map_entries(ctmr_engine* e, const uint8_t* blob, const uint64_t* bounds, uint64_t n, ctmr_record* records,
                     uint64_t* new_idx, uint64_t* timestamp, ctmr_decode_stats* dstats, ctmr_batch_stats* stats) {
  if (!e || (n && (!blob || !bounds))) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (stats) memset(stats, 0, sizeof *stats);
  if (dstats) memset(dstats, 0, sizeof *dstats);
  if (n == 0) return CTMR_OK;
  for (uint64_t i = 0; i < 2 * n; i++)
    if (bounds[i + 1] < bounds[i]) return fail(e, CTMR_E_INVAL, "bounds not monotone at %llu", (unsigned long long)i);
  const uint64_t base = bounds[0], bytes = bounds[2 * n] - base;
  int r;
  if ((r = ensure(e, SC_STAGE_A, bytes + CTMR_PAYLOAD_PAD + 16))) return r;
  // SC_STAGE_B: bounds | new_idx | timestamp
  const size_t o_b = 0, o_new = (2 * n + 1) * 8, o_ts = o_new + n * 8;
  if ((r = ensure(e, SC_STAGE_B, o_ts + n * 8))) return r;
  uint8_t* B = (uint8_t*)e->d_scratch[SC_STAGE_B];
  std::vector<uint64_t> rel(2 * n + 1);
  for (uint64_t i = 0; i <= 2 * n; i++) rel[i] = bounds[i] - base;
  HIPCHK(e, hipMemcpyAsync(e->d_scratch[SC_STAGE_A], blob + base, bytes, hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemsetAsync((uint8_t*)e->d_scratch[SC_STAGE_A] + bytes, 0, CTMR_PAYLOAD_PAD, e->stream));
  HIPCHK(e, hipMemcpyAsync(B + o_b, rel.data(), (2 * n + 1) * 8, hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  ctmr_batch_stats st;
  ctmr_entry_view v;
  r = map_entries_locked(e, (const uint8_t*)e->d_scratch[SC_STAGE_A], (const uint64_t*)(B + o_b), n, nullptr,
                         new_idx ? (uint64_t*)(B + o_new) : nullptr, timestamp ? (uint64_t*)(B + o_ts) : nullptr,
                         dstats, &st, &v);
  if (r) return r;
  if (records) HIPCHK(e, hipMemcpy(records, e->d_scratch[SC_RECORDS], n * sizeof(ctmr_record), hipMemcpyDeviceToHost));
  if (new_idx && st.n_new) HIPCHK(e, hipMemcpy(new_idx, B + o_new, st.n_new * 8, hipMemcpyDeviceToHost));
  if (timestamp) HIPCHK(e, hipMemcpy(timestamp, B + o_ts, n * 8, hipMemcpyDeviceToHost));
  if (stats) *stats = st;
  // what ctmr_pem_new encodes: the view lives in SC_VIEW, the new list in SC_STAGE_B
  e->last_n = n; e->last_n_new = new_idx ? st.n_new : 0; e->last_o_new = o_new;
  e->last_is_view = true;
  e->last_o_start = (size_t)((uint8_t*)v.cert_start - (uint8_t*)e->d_scratch[SC_VIEW]);
  e->last_o_end = (size_t)((uint8_t*)v.cert_end - (uint8_t*)e->d_scratch[SC_VIEW]);
  return CTMR_OK;
}

int ctmr_set_issuer_autoregister(ctmr_engine* e, int on) {
  if (!e) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  e->auto_register = on != 0;
  return CTMR_OK;
}

int ctmr_pending_issuers(ctmr_engine* e, uint8_t* out, size_t cap, size_t* need, uint64_t* count) {
  if (!e) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  int rc = CTMR_OK;
  serialise(e->pending_issuers, out, cap, need, count, &rc);
  return rc;
}

// ------------------------------------------------------------------ IssuerMetadata on device (N3)

constexpr size_t META_HOUR_PAGE_BYTES = (size_t)META_HOUR_PAGE * (META_HOUR_BITS / 8);

// one (issuer, hour) bitmap page per META_HOUR_PAGE registered issuers
static int meta_hour_pages_locked(ctmr_engine* e) {
  const size_t need = (e->issuers.size() + META_HOUR_PAGE - 1) / META_HOUR_PAGE;
  if (e->meta_hour_pages.size() >= need) return CTMR_OK;
  if (!e->d_meta_hour_pages) {
    const size_t cap = ((size_t)e->max_issuers + META_HOUR_PAGE - 1) / META_HOUR_PAGE;
    if (hipMalloc(&e->d_meta_hour_pages, cap * sizeof(uint32_t*)) != hipSuccess) {
      (void)hipGetLastError();
      return fail(e, CTMR_E_NOMEM, "expDate bitmap page table");
    }
  }
  while (e->meta_hour_pages.size() < need) {
    uint32_t* p = nullptr;
    if (hipMalloc(&p, META_HOUR_PAGE_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return fail(e, CTMR_E_NOMEM, "expDate bitmap page (%zu bytes)", META_HOUR_PAGE_BYTES);
    }
    HIPCHK(e, hipMemsetAsync(p, 0, META_HOUR_PAGE_BYTES, e->stream));
    e->meta_hour_pages.push_back(p);
  }
  HIPCHK(e, hipMemcpyAsync(e->d_meta_hour_pages, e->meta_hour_pages.data(), e->meta_hour_pages.size() * sizeof(uint32_t*),
                           hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));  // the vector may move before the copy has read it
  return CTMR_OK;
}

static int meta_reset_locked(ctmr_engine* e) {
  if (!e->d_meta_slots) return CTMR_OK;
  HIPCHK(e, hipMemsetAsync(e->d_meta_slots, 0, e->n_meta_slots * sizeof(MetaSlot), e->stream));
  for (auto p : e->meta_hour_pages) HIPCHK(e, hipMemsetAsync(p, 0, META_HOUR_PAGE_BYTES, e->stream));
  HIPCHK(e, hipMemsetAsync(e->d_mcount, 0, 64, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->last_meta_valid = false;
  return CTMR_OK;
}

static int meta_device_locked(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets, const uint64_t* d_ends,
                              const ctmr_record* d_records, const uint64_t* d_new_idx, uint64_t n_new,
                              ctmr_meta_item* d_items, uint64_t items_cap, uint64_t* n_items) {
  HIPCHK(e, hipSetDevice(e->device));
  if (n_items) *n_items = 0;
  if (!e->cfg.collect_meta || !e->d_meta_slots) return fail(e, CTMR_E_INVAL, "engine created without collect_meta");
  if (n_new == 0) return CTMR_OK;
  if (!e->d_scratch[SC_META] || e->meta_n == 0) return fail(e, CTMR_E_INVAL, "no map call precedes ctmr_meta_new");
  MetaArgs a;
  a.payload = d_payload; a.offsets = d_offsets; a.ends = d_ends; a.records = d_records; a.canon = e->d_canon;
  a.meta_loc = (const uint2*)e->d_scratch[SC_META]; a.new_idx = d_new_idx; a.n_new = n_new;
  a.slots = e->d_meta_slots; a.mask = e->n_meta_slots - 1; a.arena = e->d_meta_arena; a.arena_cap = e->meta_arena_cap;
  a.counters = e->d_mcount; a.items = (MetaItem*)d_items; a.items_cap = items_cap;
  a.epoch = ++e->meta_epoch;
  int rp = meta_hour_pages_locked(e);
  if (rp) return rp;
  a.hour_pages = e->d_meta_hour_pages; a.n_hour_pages = (uint32_t)e->meta_hour_pages.size();
  HIPCHK(e, hipMemsetAsync(e->d_mcount + 1, 0, 8, e->stream));
  hipLaunchKernelGGL(k_meta_new, dim3((unsigned)((n_new + 255) / 256)), dim3(256), 0, e->stream, a);
  unsigned long long hc[3];
  HIPCHK(e, hipMemcpyAsync(hc, e->d_mcount, 24, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  HIPCHK(e, hipGetLastError());
  if (n_items) *n_items = hc[1];
  if (hc[1] > items_cap) {
    int r = meta_reset_locked(e);
    if (r) return r;
    return fail(e, CTMR_E_RANGE, "meta item buffer too small: %llu items (memo cleared; call again)", hc[1]);
  }
  return CTMR_OK;
}

int ctmr_meta_new_device(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets, const uint64_t* d_ends,
                         const ctmr_record* d_records, const uint64_t* d_new_idx, uint64_t n_new,
                         ctmr_meta_item* d_items, uint64_t items_cap, uint64_t* n_items) {
  if (!e || (n_new && (!d_payload || !d_offsets || !d_records || !d_new_idx || !d_items))) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  return meta_device_locked(e, d_payload, d_offsets, d_ends, d_records, d_new_idx, n_new, d_items, items_cap, n_items);
}

int ctmr_meta_reset(ctmr_engine* e) {
  if (!e) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  return meta_reset_locked(e);
}

int ctmr_meta_new(ctmr_engine* e, ctmr_meta_item* items, uint64_t items_cap, uint8_t* bytes, size_t bytes_cap,
                  uint64_t* n_items, size_t* bytes_need) {
  if (!e) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (n_items) *n_items = 0;
  if (bytes_need) *bytes_need = 0;
  const uint64_t nn = e->last_n_new;
  if (nn == 0) return CTMR_OK;
  uint8_t* B = (uint8_t*)e->d_scratch[SC_STAGE_B];
  const uint8_t* V = (const uint8_t*)e->d_scratch[SC_VIEW];
  const uint64_t* offs = e->last_is_view ? (const uint64_t*)(V + e->last_o_start) : (const uint64_t*)(B + e->last_o_off);
  const uint64_t* ends = e->last_is_view ? (const uint64_t*)(V + e->last_o_end) : nullptr;
  int r;
  if (!e->last_meta_valid) {
    uint64_t cap = std::min<uint64_t>(3 * nn + 1024, 16ull << 20);
    for (int attempt = 0;; attempt++) {
      if ((r = ensure(e, SC_ITEMS, cap * sizeof(ctmr_meta_item)))) return r;
      uint64_t got = 0;
      r = meta_device_locked(e, (const uint8_t*)e->d_scratch[SC_STAGE_A], offs, ends,
                             (const ctmr_record*)e->d_scratch[SC_RECORDS], (const uint64_t*)(B + e->last_o_new), nn,
                             (ctmr_meta_item*)e->d_scratch[SC_ITEMS], cap, &got);
      if (r == CTMR_OK) {
        e->last_meta_items = got;
        break;
      }
      if (r != CTMR_E_RANGE || attempt) return r;
      cap = got + 1024;  // the memo was cleared: the second run re-reports everything into a buffer that fits
    }
    e->last_meta_valid = true;
  }
  const uint64_t ni = e->last_meta_items;
  if (n_items) *n_items = ni;
  std::vector<ctmr_meta_item> h(ni);
  if (ni) HIPCHK(e, hipMemcpy(h.data(), e->d_scratch[SC_ITEMS], ni * sizeof(ctmr_meta_item), hipMemcpyDeviceToHost));
  size_t need = 0;
  for (auto& it : h) {
    if (it.kind == CTMR_MK_HOST || it.kind == CTMR_MK_EXPDATE) it.len = 0;
    need += it.len;
  }
  if (bytes_need) *bytes_need = need;
  if (ni > items_cap || need > bytes_cap || (ni && !items) || (need && !bytes))
    return fail(e, CTMR_E_RANGE, "meta buffers too small: %llu items, %llu bytes", (unsigned long long)ni, (unsigned long long)need);
  size_t at = 0;
  for (uint64_t k = 0; k < ni; k++) {
    if (h[k].len) {
      uint64_t lo = 0;
      HIPCHK(e, hipMemcpy(&lo, offs + h[k].entry, 8, hipMemcpyDeviceToHost));
      HIPCHK(e, hipMemcpy(bytes + at, (const uint8_t*)e->d_scratch[SC_STAGE_A] + lo + h[k].off, h[k].len, hipMemcpyDeviceToHost));
      at += h[k].len;
    }
    items[k] = h[k];
  }
  return CTMR_OK;
}

// ------------------------------------------------------------------ whole-certificate SHA-256 (auxiliary)
int ctmr_fingerprint_device(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets,
                            const uint64_t* d_ends, uint64_t n, uint8_t* d_digests, float* ms) {
  if (!e || (n && (!d_payload || !d_offsets || !d_digests))) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (ms) *ms = 0.f;
  if (n == 0) return CTMR_OK;
  if (((uintptr_t)d_digests & 15) != 0) return fail(e, CTMR_E_INVAL, "digests must be 16-byte aligned");
  HIPCHK(e, hipEventRecord(e->ev[5], e->stream));
  hipLaunchKernelGGL(k_fingerprint, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, d_payload, d_offsets,
                     d_ends, n, (uint32_t*)d_digests);
  HIPCHK(e, hipEventRecord(e->ev[6], e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  HIPCHK(e, hipGetLastError());
  if (ms) (void)hipEventElapsedTime(ms, e->ev[5], e->ev[6]);
  return CTMR_OK;
}

// ------------------------------------------------------------------ PEM write-back (N1)

static int pem_device_locked(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets,
                             const uint64_t* d_idx, uint64_t n_idx, uint8_t* d_pem, uint64_t pem_cap,
                             uint64_t* d_pem_offsets, uint64_t* pem_bytes, const uint64_t* d_ends = nullptr) {
  HIPCHK(e, hipSetDevice(e->device));
  if (pem_bytes) *pem_bytes = 0;
  if (n_idx == 0) {
    if (d_pem_offsets) HIPCHK(e, hipMemsetAsync(d_pem_offsets, 0, 8, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return CTMR_OK;
  }
  if (n_idx >= 0x7fffffffull) return fail(e, CTMR_E_INVAL, "too many certificates in one PEM call");
  hipLaunchKernelGGL(k_pem_len, dim3((unsigned)((n_idx + 1 + 255) / 256)), dim3(256), 0, e->stream, d_offsets, d_ends,
                     d_idx, n_idx, d_pem_offsets);
  size_t tmp_bytes = 0;
  HIPCHK(e, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_pem_offsets, d_pem_offsets, (int64_t)(n_idx + 1), e->stream));
  int r;
  if ((r = ensure(e, SC_TMP, tmp_bytes))) return r;
  HIPCHK(e, hipcub::DeviceScan::ExclusiveSum(e->d_scratch[SC_TMP], tmp_bytes, d_pem_offsets, d_pem_offsets,
                                             (int64_t)(n_idx + 1), e->stream));
  uint64_t total = 0;
  HIPCHK(e, hipMemcpyAsync(&total, d_pem_offsets + n_idx, 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (pem_bytes) *pem_bytes = total;
  if (!d_pem) return CTMR_OK;  // size query
  if (total > pem_cap) return fail(e, CTMR_E_RANGE, "PEM buffer too small: need %llu bytes", (unsigned long long)total);
  hipLaunchKernelGGL(k_pem_encode, dim3((unsigned)((n_idx + 4 * PEM_PER_WAVE - 1) / (4 * PEM_PER_WAVE))), dim3(256), 0, e->stream,
                     d_payload, d_offsets, d_ends, d_idx, n_idx, (const uint64_t*)d_pem_offsets, d_pem);
  HIPCHK(e, hipStreamSynchronize(e->stream));
  HIPCHK(e, hipGetLastError());
  return CTMR_OK;
}

int ctmr_pem_encode_device(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets,
                           const uint64_t* d_idx, uint64_t n_idx, uint8_t* d_pem, uint64_t pem_cap,
                           uint64_t* d_pem_offsets, uint64_t* pem_bytes) {
  if (!e || !d_pem_offsets || (n_idx && (!d_payload || !d_offsets || !d_idx))) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  return pem_device_locked(e, d_payload, d_offsets, d_idx, n_idx, d_pem, pem_cap, d_pem_offsets, pem_bytes);
}

int ctmr_pem_encode_view_device(ctmr_engine* e, const uint8_t* d_blob, const ctmr_entry_view* v,
                                const uint64_t* d_idx, uint64_t n_idx, uint8_t* d_pem, uint64_t pem_cap,
                                uint64_t* d_pem_offsets, uint64_t* pem_bytes) {
  if (!e || !v || !d_pem_offsets || (n_idx && (!d_blob || !v->cert_start || !v->cert_end || !d_idx))) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  return pem_device_locked(e, d_blob, v->cert_start, d_idx, n_idx, d_pem, pem_cap, d_pem_offsets, pem_bytes, v->cert_end);
}

int ctmr_pem_new(ctmr_engine* e, uint8_t* out, size_t cap, uint64_t* pem_offsets, size_t* need, uint64_t* count) {
  if (!e) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  const uint64_t nn = e->last_n_new;
  if (count) *count = nn;
  if (need) *need = 0;
  if (nn == 0) {
    if (pem_offsets) pem_offsets[0] = 0;
    return CTMR_OK;
  }
  int r;
  if ((r = ensure(e, SC_PEMOFF, (nn + 1) * 8))) return r;
  uint8_t* B = (uint8_t*)e->d_scratch[SC_STAGE_B];
  uint64_t* d_po = (uint64_t*)e->d_scratch[SC_PEMOFF];
  uint64_t total = 0;
  const uint8_t* V = (const uint8_t*)e->d_scratch[SC_VIEW];
  const uint64_t* offs = e->last_is_view ? (const uint64_t*)(V + e->last_o_start) : (const uint64_t*)(B + e->last_o_off);
  const uint64_t* ends = e->last_is_view ? (const uint64_t*)(V + e->last_o_end) : nullptr;
  r = pem_device_locked(e, (const uint8_t*)e->d_scratch[SC_STAGE_A], offs,
                        (const uint64_t*)(B + e->last_o_new), nn, nullptr, 0, d_po, &total, ends);
  if (r) return r;
  if (need) *need = total;
  if (!out || cap < total) return out ? fail(e, CTMR_E_RANGE, "PEM buffer too small: need %llu bytes", (unsigned long long)total) : CTMR_E_RANGE;
  if ((r = ensure(e, SC_PEM, total + 64))) return r;
  r = pem_device_locked(e, (const uint8_t*)e->d_scratch[SC_STAGE_A], offs,
                        (const uint64_t*)(B + e->last_o_new), nn, (uint8_t*)e->d_scratch[SC_PEM], total + 64, d_po, &total, ends);
  if (r) return r;
  HIPCHK(e, hipMemcpy(out, e->d_scratch[SC_PEM], total, hipMemcpyDeviceToHost));
  if (pem_offsets) HIPCHK(e, hipMemcpy(pem_offsets, d_po, (nn + 1) * 8, hipMemcpyDeviceToHost));
  return CTMR_OK;
}

// ------------------------------------------------------------------ cross-GPU key exchange

static int exchange_export_locked(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets,
                                  const uint32_t* d_issuer_idx, const uint8_t* d_entry_type, uint64_t n,
                                  ctmr_record* d_records, uint32_t world, void* d_keys_out, uint64_t* counts,
                                  const uint64_t* d_ends, uint64_t blob_bytes) {
  HIPCHK(e, hipSetDevice(e->device));
  for (uint32_t w = 0; w < world; w++) counts[w] = 0;
  if (n == 0) return CTMR_OK;
  if (n >= 0xfffffff0ull) return fail(e, CTMR_E_INVAL, "batch too large (n < 2^32-16)");
  int r;
  if ((r = launch_map(e, d_payload, d_offsets, d_issuer_idx, d_entry_type, n, d_records, false, nullptr, d_ends,
                      blob_bytes + CTMR_PAYLOAD_PAD))) return r;
  const uint64_t nb = (n + 1023) / 1024;
  const uint64_t ncnt = (uint64_t)world * nb;
  if ((r = ensure(e, SC_SLOTID, n))) return r;                 // owner byte per entry
  if ((r = ensure(e, SC_BLKNEW, (ncnt + 1) * 4))) return r;     // counts u32
  if ((r = ensure(e, SC_BLKBASE, (ncnt + 1) * 8))) return r;    // bases u64
  uint8_t* d_owner = (uint8_t*)e->d_scratch[SC_SLOTID];
  uint32_t* d_cnt = (uint32_t*)e->d_scratch[SC_BLKNEW];
  uint64_t* d_base = (uint64_t*)e->d_scratch[SC_BLKBASE];
  InsertArgs ia;
  ia.records = d_records; ia.payload = d_payload; ia.offsets = d_offsets; ia.ends = d_ends; ia.canon = e->d_canon;
  ia.table = e->table; ia.mask = e->nslots - 1; ia.slot_id = nullptr; ia.n = n; ia.epoch = e->epoch;
  HIPCHK(e, hipMemsetAsync(d_cnt, 0, (ncnt + 1) * 4, e->stream));
  hipLaunchKernelGGL(k_key_count, dim3((unsigned)nb), dim3(1024), 0, e->stream, ia, world, nb, d_owner, d_cnt);
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, e->stream, d_cnt, ncnt + 1, d_base);
  hipLaunchKernelGGL(k_key_scatter, dim3((unsigned)nb), dim3(1024), 0, e->stream, ia, world, nb,
                     (const uint8_t*)d_owner, (const uint64_t*)d_base, (KeyRec*)d_keys_out);
  std::vector<uint64_t> base(world + 1);
  for (uint32_t w = 0; w <= world; w++)
    HIPCHK(e, hipMemcpyAsync(&base[w], d_base + (uint64_t)w * nb, 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  HIPCHK(e, hipGetLastError());
  for (uint32_t w = 0; w < world; w++) counts[w] = base[w + 1] - base[w];
  return CTMR_OK;
}

int ctmr_exchange_export_device(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets,
                                const uint32_t* d_issuer_idx, const uint8_t* d_entry_type, uint64_t n,
                                ctmr_record* d_records, uint32_t world, void* d_keys_out,
                                uint64_t* counts) {
  if (!e || !d_records || !d_keys_out || !counts || world == 0 || world > MAX_WORLD ||
      (n && (!d_payload || !d_offsets || !d_issuer_idx)))
    return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  return exchange_export_locked(e, d_payload, d_offsets, d_issuer_idx, d_entry_type, n, d_records, world, d_keys_out,
                                counts, nullptr, 0);
}

int ctmr_exchange_export_view_device(ctmr_engine* e, const uint8_t* d_blob, uint64_t blob_bytes,
                                     const ctmr_entry_view* v, uint64_t n, ctmr_record* d_records,
                                     uint32_t world, void* d_keys_out, uint64_t* counts) {
  if (!e || !v || !d_records || !d_keys_out || !counts || world == 0 || world > MAX_WORLD ||
      (n && (!d_blob || !v->cert_start || !v->cert_end || !v->issuer_idx)))
    return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  return exchange_export_locked(e, d_blob, v->cert_start, v->issuer_idx, v->entry_type, n, d_records, world, d_keys_out,
                                counts, v->cert_end, blob_bytes);
}

int ctmr_exchange_insert_device(ctmr_engine* e, const void* d_keys, uint64_t n_keys, uint8_t* d_flags,
                                uint64_t* n_new) {
  if (!e || (n_keys && (!d_keys || !d_flags))) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (n_new) *n_new = 0;
  if (n_keys == 0) return CTMR_OK;
  if (n_keys >= 0xfffffff0ull) return fail(e, CTMR_E_INVAL, "too many keys");
  int r;
  if ((r = ensure(e, SC_SLOTID, n_keys * 4))) return r;
  uint32_t* d_slot = (uint32_t*)e->d_scratch[SC_SLOTID];
  e->epoch++;
  e->pairs_dirty = true;
  HIPCHK(e, hipMemsetAsync(e->d_stats, 0, sizeof(DevStats), e->stream));
  const KeyRec* keys = (const KeyRec*)d_keys;
  const unsigned b256 = (unsigned)((n_keys + 255) / 256);
  const uint64_t nb = (n_keys + 1023) / 1024;
  hipLaunchKernelGGL(k_keys_insert, dim3(b256), dim3(256), 0, e->stream, keys, n_keys, e->table, e->nslots - 1, e->epoch, d_slot);
  hipLaunchKernelGGL(k_keys_insert2, dim3(b256), dim3(256), 0, e->stream, keys, n_keys, e->table, e->nslots - 1, e->epoch, d_slot);
  hipLaunchKernelGGL(k_keys_resolve, dim3((unsigned)(nb < 512 ? nb : 512)), dim3(1024), 0, e->stream, keys, n_keys, nb,
                     (const Slot*)e->table, e->epoch, (const uint32_t*)d_slot, d_flags, e->issuer_counts, e->d_stats);
  DevStats hs;
  HIPCHK(e, hipMemcpyAsync(&hs, e->d_stats, sizeof hs, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  HIPCHK(e, hipGetLastError());
  if (hs.n_full) return fail(e, CTMR_E_FULL, "known-certificate table full (%llu slots)", (unsigned long long)e->nslots);
  if (n_new) *n_new = hs.n_new;
  return CTMR_OK;
}

int ctmr_exchange_apply_device(ctmr_engine* e, ctmr_record* d_records, uint64_t n, const void* d_keys_sent,
                               const uint8_t* d_flags, uint64_t n_keys, uint64_t* d_new_idx,
                               ctmr_batch_stats* stats) {
  if (!e || (n && !d_records) || (n_keys && (!d_keys_sent || !d_flags))) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (stats) memset(stats, 0, sizeof *stats);
  if (n == 0) return CTMR_OK;
  const uint64_t nb = (n + 1023) / 1024;
  int r;
  if ((r = ensure(e, SC_BLKNEW, nb * 4))) return r;
  if ((r = ensure(e, SC_BLKBASE, nb * 8))) return r;
  uint32_t* d_blk_new = (uint32_t*)e->d_scratch[SC_BLKNEW];
  uint64_t* d_blk_base = (uint64_t*)e->d_scratch[SC_BLKBASE];
  HIPCHK(e, hipMemsetAsync(d_blk_new, 0, nb * 4, e->stream));
  HIPCHK(e, hipMemsetAsync(e->d_stats, 0, sizeof(DevStats), e->stream));
  if (n_keys)
    hipLaunchKernelGGL(k_apply_flags, dim3((unsigned)((n_keys + 255) / 256)), dim3(256), 0, e->stream,
                       (const KeyRec*)d_keys_sent, d_flags, n_keys, d_records, d_blk_new);
  hipLaunchKernelGGL(k_status_hist, dim3((unsigned)(nb < 512 ? nb : 512)), dim3(1024), 0, e->stream,
                     (const ctmr_record*)d_records, n, nb, e->d_stats);
  if (d_new_idx) {
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, e->stream, d_blk_new, nb, d_blk_base);
    hipLaunchKernelGGL(k_compact, dim3((unsigned)nb), dim3(256), 0, e->stream, (const ctmr_record*)d_records,
                       (const uint32_t*)nullptr, n, (const uint64_t*)d_blk_base, d_new_idx);
  }
  DevStats hs;
  HIPCHK(e, hipMemcpyAsync(&hs, e->d_stats, sizeof hs, hipMemcpyDeviceToHost, e->stream));
  // NEW count = Σ blk_new
  std::vector<uint32_t> bn(nb);
  HIPCHK(e, hipMemcpyAsync(bn.data(), d_blk_new, nb * 4, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  HIPCHK(e, hipGetLastError());
  if (stats) {
    stats->n = n;
    for (int k = 0; k < CTMR_ST__COUNT; k++) stats->by_status[k] = hs.by_status[k];
    uint64_t nn = 0;
    for (auto v : bn) nn += v;
    stats->n_new = nn;
    stats->n_host_set = hs.n_host;
    stats->n_dup = hs.by_status[CTMR_ST_PASS] - nn - hs.n_host;
  }
  return CTMR_OK;
}

// ------------------------------------------------------------------ cross-GPU dedup, Bloom pre-filter variant

int ctmr_bloom_config(ctmr_engine* e, uint64_t bits, void* d_words) {
  if (!e) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (bits < 4096 || bits > (1ull << 40) || (bits & (bits - 1)))
    return fail(e, CTMR_E_INVAL, "Bloom filter size must be a power of two in [2^12, 2^40] bits");
  if ((uintptr_t)d_words & 7) return fail(e, CTMR_E_INVAL, "Bloom filter buffer must be 8-byte aligned");
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (e->d_bloom && e->bloom_owned) (void)hipFree(e->d_bloom);
  e->d_bloom = nullptr;
  e->bloom_words = 0;
  if (d_words) {
    e->d_bloom = (unsigned long long*)d_words;
    e->bloom_owned = false;
  } else {
    if (hipMalloc(&e->d_bloom, bits / 8) != hipSuccess) {
      (void)hipGetLastError();
      e->d_bloom = nullptr;
      return fail(e, CTMR_E_NOMEM, "Bloom filter: %llu bytes", (unsigned long long)(bits / 8));
    }
    e->bloom_owned = true;
  }
  e->bloom_words = bits / 64;
  e->bloom_round_epoch = 0;
  HIPCHK(e, hipMemsetAsync(e->d_bloom, 0, bits / 8, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return CTMR_OK;
}

int ctmr_bloom_device(ctmr_engine* e, void** d_words, uint64_t* n_words) {
  if (!e || !d_words) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  if (!e->d_bloom) return fail(e, CTMR_E_INVAL, "ctmr_bloom_config has not been called");
  *d_words = e->d_bloom;
  if (n_words) *n_words = e->bloom_words;
  return CTMR_OK;
}

static void bloom_args(ctmr_engine* e, InsertArgs& ia, const uint8_t* d_payload, const uint64_t* d_offsets,
                       const uint64_t* d_ends, uint64_t n, const ctmr_record* d_records) {
  ia.records = d_records; ia.payload = d_payload; ia.offsets = d_offsets; ia.ends = d_ends; ia.canon = e->d_canon;
  ia.table = e->table; ia.mask = e->nslots - 1; ia.slot_id = nullptr; ia.ent = nullptr; ia.n = n; ia.epoch = e->epoch;
}

int ctmr_bloom_add_device(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets, const uint64_t* d_ends,
                          uint64_t n, const ctmr_record* d_records) {
  if (!e || (n && (!d_payload || !d_offsets))) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->d_bloom) return fail(e, CTMR_E_INVAL, "ctmr_bloom_config has not been called");
  e->bloom_round_epoch = n ? e->epoch : 0;
  if (n == 0) return CTMR_OK;
  if (!d_records) d_records = (const ctmr_record*)e->d_scratch[SC_RECORDS];
  if (!d_records) return fail(e, CTMR_E_INVAL, "no records: run ctmr_map_*_device on this batch first");
  InsertArgs ia;
  bloom_args(e, ia, d_payload, d_offsets, d_ends, n, d_records);
  hipLaunchKernelGGL(k_bloom_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, ia, e->d_bloom,
                     e->bloom_words - 1);
  HIPCHK(e, hipStreamSynchronize(e->stream));  // the caller all-gathers the filter on its own stream next
  HIPCHK(e, hipGetLastError());
  return CTMR_OK;
}

int ctmr_bloom_probe_device(ctmr_engine* e, const uint8_t* d_payload, const uint64_t* d_offsets,
                            const uint64_t* d_ends, uint64_t n, const ctmr_record* d_records, const void* d_filters,
                            uint32_t world, uint32_t rank, uint64_t order_base, void* d_keys_out, uint64_t keys_cap,
                            uint64_t* counts) {
  if (!e || !counts || !d_filters || world == 0 || world > MAX_WORLD || rank >= world ||
      (n && (!d_payload || !d_offsets)) || (keys_cap && !d_keys_out))
    return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->d_bloom) return fail(e, CTMR_E_INVAL, "ctmr_bloom_config has not been called");
  for (uint32_t w = 0; w < world; w++) counts[w] = 0;
  if (n == 0) return CTMR_OK;
  if (!d_records) d_records = (const ctmr_record*)e->d_scratch[SC_RECORDS];
  if (!d_records) return fail(e, CTMR_E_INVAL, "no records: run ctmr_map_*_device on this batch first");
  const uint64_t nb = (n + 1023) / 1024;
  const uint64_t ncnt = (uint64_t)world * nb;
  int r;
  if ((r = ensure(e, SC_SLOTID, n * 4))) return r;              // hit mask u16 per entry
  if ((r = ensure(e, SC_BLKNEW, (ncnt + 1) * 4))) return r;
  if ((r = ensure(e, SC_BLKBASE, (ncnt + 1) * 8))) return r;
  uint16_t* d_hit = (uint16_t*)e->d_scratch[SC_SLOTID];
  uint32_t* d_cnt = (uint32_t*)e->d_scratch[SC_BLKNEW];
  uint64_t* d_base = (uint64_t*)e->d_scratch[SC_BLKBASE];
  InsertArgs ia;
  bloom_args(e, ia, d_payload, d_offsets, d_ends, n, d_records);
  HIPCHK(e, hipMemsetAsync(d_cnt, 0, (ncnt + 1) * 4, e->stream));
  hipLaunchKernelGGL(k_bloom_probe, dim3((unsigned)nb), dim3(1024), 0, e->stream, ia,
                     (const unsigned long long*)d_filters, e->bloom_words, world, rank, nb, d_hit, d_cnt);
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, e->stream, d_cnt, ncnt + 1, d_base);
  std::vector<uint64_t> base(world + 1);
  for (uint32_t w = 0; w <= world; w++)
    HIPCHK(e, hipMemcpyAsync(&base[w], d_base + (uint64_t)w * nb, 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  HIPCHK(e, hipGetLastError());
  for (uint32_t w = 0; w < world; w++) counts[w] = base[w + 1] - base[w];
  if (base[world] > keys_cap)
    return fail(e, CTMR_E_RANGE, "key buffer too small: %llu records needed", (unsigned long long)base[world]);
  if (base[world] == 0) return CTMR_OK;
  hipLaunchKernelGGL(k_bloom_scatter, dim3((unsigned)nb), dim3(1024), 0, e->stream, ia, world, nb,
                     (const uint16_t*)d_hit, (const uint64_t*)d_base, (unsigned long long)order_base,
                     (KeyRec*)d_keys_out);
  HIPCHK(e, hipStreamSynchronize(e->stream));
  HIPCHK(e, hipGetLastError());
  return CTMR_OK;
}

int ctmr_bloom_lookup_device(ctmr_engine* e, const void* d_keys, uint64_t n_keys, uint64_t order_base,
                             uint8_t* d_flags) {
  if (!e || (n_keys && (!d_keys || !d_flags))) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (n_keys == 0) return CTMR_OK;
  hipLaunchKernelGGL(k_keys_lookup, dim3((unsigned)((n_keys + 255) / 256)), dim3(256), 0, e->stream,
                     (const KeyRec*)d_keys, n_keys, (const Slot*)e->table, e->nslots - 1, e->bloom_round_epoch,
                     (unsigned long long)order_base, d_flags);
  HIPCHK(e, hipStreamSynchronize(e->stream));
  HIPCHK(e, hipGetLastError());
  return CTMR_OK;
}

int ctmr_bloom_apply_device(ctmr_engine* e, ctmr_record* d_records, uint64_t n, const void* d_keys_sent,
                            const uint8_t* d_flags, uint64_t n_keys, uint64_t* d_new_idx, ctmr_batch_stats* stats) {
  if (!e || (n_keys && (!d_keys_sent || !d_flags))) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (stats) memset(stats, 0, sizeof *stats);
  if (n == 0) return CTMR_OK;
  if (!d_records) d_records = (ctmr_record*)e->d_scratch[SC_RECORDS];
  if (!d_records) return fail(e, CTMR_E_INVAL, "no records: run ctmr_map_*_device on this batch first");
  const uint64_t nb = (n + 1023) / 1024;
  int r;
  if ((r = ensure(e, SC_BLKNEW, nb * 4))) return r;
  if ((r = ensure(e, SC_BLKBASE, nb * 8))) return r;
  uint32_t* d_blk_new = (uint32_t*)e->d_scratch[SC_BLKNEW];
  uint64_t* d_blk_base = (uint64_t*)e->d_scratch[SC_BLKBASE];
  HIPCHK(e, hipMemsetAsync(e->d_stats, 0, sizeof(DevStats), e->stream));
  if (n_keys) {
    e->pairs_dirty = true;
    hipLaunchKernelGGL(k_bloom_apply, dim3((unsigned)((n_keys + 255) / 256)), dim3(256), 0, e->stream,
                       (const KeyRec*)d_keys_sent, d_flags, n_keys, d_records, e->table, e->nslots - 1,
                       e->issuer_counts);
  }
  hipLaunchKernelGGL(k_count_new_flags, dim3((unsigned)nb), dim3(1024), 0, e->stream, (const ctmr_record*)d_records, n,
                     d_blk_new);
  hipLaunchKernelGGL(k_status_hist, dim3((unsigned)(nb < 512 ? nb : 512)), dim3(1024), 0, e->stream,
                     (const ctmr_record*)d_records, n, nb, e->d_stats);
  if (d_new_idx) {
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, e->stream, d_blk_new, nb, d_blk_base);
    hipLaunchKernelGGL(k_compact, dim3((unsigned)nb), dim3(256), 0, e->stream, (const ctmr_record*)d_records,
                       (const uint32_t*)nullptr, n, (const uint64_t*)d_blk_base, d_new_idx);
  }
  DevStats hs;
  HIPCHK(e, hipMemcpyAsync(&hs, e->d_stats, sizeof hs, hipMemcpyDeviceToHost, e->stream));
  std::vector<uint32_t> bn(nb);
  HIPCHK(e, hipMemcpyAsync(bn.data(), d_blk_new, nb * 4, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  HIPCHK(e, hipGetLastError());
  if (stats) {
    stats->n = n;
    for (int k = 0; k < CTMR_ST__COUNT; k++) stats->by_status[k] = hs.by_status[k];
    uint64_t nn = 0;
    for (auto v : bn) nn += v;
    stats->n_new = nn;                 // long-serial entries that were new in the host-side set keep their flag
    stats->n_host_set = hs.n_host;
    stats->n_dup = hs.by_status[CTMR_ST_PASS] - nn;
  }
  return CTMR_OK;
}

// ------------------------------------------------------------------ RemoteCache set methods

int ctmr_set_insert(ctmr_engine* e, const char* key, size_t kl, const uint8_t* m, size_t ml, int* was_new) {
  if (!e || !key || (ml && !m) || !was_new) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  int32_t eh; uint32_t canon;
  if (ml <= CTMR_MAX_SERIAL && table_key(e, key, kl, &eh, &canon)) return point_op(e, 0, eh, canon, m, ml, was_new);
  std::string k(key, kl);
  bool ins = e->hstore[k].insert(std::string((const char*)m, ml)).second;
  if (ins && table_key(e, key, kl, &eh, &canon)) e->host_issuer_counts[canon]++;
  *was_new = ins;
  return CTMR_OK;
}

int ctmr_set_contains(ctmr_engine* e, const char* key, size_t kl, const uint8_t* m, size_t ml, int* present) {
  if (!e || !key || (ml && !m) || !present) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  int32_t eh; uint32_t canon;
  if (ml <= CTMR_MAX_SERIAL && table_key(e, key, kl, &eh, &canon)) return point_op(e, 1, eh, canon, m, ml, present);
  auto it = e->hstore.find(std::string(key, kl));
  *present = it != e->hstore.end() && it->second.count(std::string((const char*)m, ml));
  return CTMR_OK;
}

int ctmr_set_remove(ctmr_engine* e, const char* key, size_t kl, const uint8_t* m, size_t ml, int* removed) {
  if (!e || !key || (ml && !m) || !removed) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  int32_t eh; uint32_t canon;
  if (ml <= CTMR_MAX_SERIAL && table_key(e, key, kl, &eh, &canon)) return point_op(e, 2, eh, canon, m, ml, removed);
  std::string k(key, kl);
  auto it = e->hstore.find(k);
  *removed = 0;
  if (it != e->hstore.end() && it->second.erase(std::string((const char*)m, ml))) {
    *removed = 1;
    if (table_key(e, key, kl, &eh, &canon)) e->host_issuer_counts[canon]--;
    if (it->second.empty()) e->hstore.erase(it);
  }
  return CTMR_OK;
}

int ctmr_set_cardinality(ctmr_engine* e, const char* key, size_t kl, int64_t* n) {
  if (!e || !key || !n) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  int64_t total = 0;
  auto it = e->hstore.find(std::string(key, kl));
  if (it != e->hstore.end()) total += (int64_t)it->second.size();
  int32_t eh; uint32_t canon;
  if (table_key(e, key, kl, &eh, &canon)) {
    uint64_t c;
    int r = pair_count(e, eh, canon, &c);
    if (r) return r;
    total += (int64_t)c;
  }
  *n = total;
  return CTMR_OK;
}

int ctmr_exists(ctmr_engine* e, const char* key, size_t kl, int* exists) {
  int64_t n;
  int r = ctmr_set_cardinality(e, key, kl, &n);
  if (r) return r;
  *exists = n > 0;
  return CTMR_OK;
}

int ctmr_set_members(ctmr_engine* e, const char* key, size_t kl, uint8_t* out, size_t cap, size_t* need, uint64_t* count) {
  if (!e || !key) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  std::vector<std::string> v;
  auto it = e->hstore.find(std::string(key, kl));
  if (it != e->hstore.end()) v.assign(it->second.begin(), it->second.end());
  int32_t eh; uint32_t canon;
  if (table_key(e, key, kl, &eh, &canon)) {
    size_t capn = 1 << 12;
    for (;;) {
      int r = ensure(e, SC_MISC, capn * 48);
      if (r) return r;
      HIPCHK(e, hipMemsetAsync(e->d_count, 0, 8, e->stream));
      hipLaunchKernelGGL(k_list, dim3((unsigned)((e->nslots + 255) / 256)), dim3(256), 0, e->stream, e->table,
                         e->nslots, (uint32_t)eh, canon, (uint8_t*)e->d_scratch[SC_MISC], (uint64_t)capn, e->d_count);
      unsigned long long cnt;
      HIPCHK(e, hipMemcpyAsync(&cnt, e->d_count, 8, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(e, hipStreamSynchronize(e->stream));
      if (cnt <= capn) {
        std::vector<uint8_t> buf(cnt * 48);
        if (cnt) HIPCHK(e, hipMemcpy(buf.data(), e->d_scratch[SC_MISC], cnt * 48, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < cnt; i++) {
          uint64_t l;
          memcpy(&l, &buf[i * 48], 8);
          v.emplace_back((const char*)&buf[i * 48 + 8], (size_t)l);
        }
        break;
      }
      capn = cnt + 64;
    }
  }
  std::sort(v.begin(), v.end());
  int rc;
  serialise(v, out, cap, need, count, &rc);
  return rc == CTMR_OK ? rc : fail(e, rc, "buffer too small");
}

int ctmr_keys(ctmr_engine* e, const char* pat, size_t pl, uint8_t* out, size_t cap, size_t* need, uint64_t* count) {
  if (!e || (pl && !pat)) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  std::set<std::string> keys;
  for (auto& kv : e->hstore) if (!kv.second.empty()) keys.insert(kv.first);
  std::vector<std::pair<unsigned long long, unsigned long long>> pr;
  int r = dump_pairs(e, &pr);
  if (r) return r;
  for (auto& p : pr) {
    const uint32_t canon = (uint32_t)(p.first >> 32) - 1;
    if (canon < e->issuers.size()) keys.insert(make_key(e, (int32_t)(uint32_t)p.first, canon));
  }
  std::vector<std::string> v;
  for (auto& k : keys) if (glob_match(pat, pl, k.data(), k.size())) v.push_back(k);
  int rc;
  serialise(v, out, cap, need, count, &rc);
  return rc == CTMR_OK ? rc : fail(e, rc, "buffer too small");
}

int ctmr_expire_at(ctmr_engine* e, const char* key, size_t kl, int64_t t) {
  if (!e || !key) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  e->expiry[std::string(key, kl)] = t;
  return CTMR_OK;
}

int ctmr_expire_sweep(ctmr_engine* e, int64_t now, uint64_t* removed) {
  if (!e) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  uint64_t total = 0;
  // (1) every table key carries ExpireAt(expDate hour) (knowncertificates.go:98-104), unless
  //     overridden by an explicit later ExpireAt → handled in (2)
  std::vector<std::pair<unsigned long long, unsigned long long>> pr;
  int r = dump_pairs(e, &pr);
  if (r) return r;
  HIPCHK(e, hipMemsetAsync(e->d_count, 0, 8, e->stream));
  bool any_override = false;
  for (auto& p : pr) {
    const uint32_t canon = (uint32_t)(p.first >> 32) - 1;
    if (canon < e->issuers.size() && e->expiry.count(make_key(e, (int32_t)(uint32_t)p.first, canon))) any_override = true;
  }
  const unsigned blocks = (unsigned)((e->nslots + 255) / 256);
  e->pairs_dirty = true;
  if (!any_override) {
    hipLaunchKernelGGL(k_sweep, dim3(blocks), dim3(256), 0, e->stream, e->table, e->nslots, 1, (long long)now, 0u, 0u,
                       e->issuer_counts, e->pairs, e->npairs - 1, e->d_count);
  } else {
    for (auto& p : pr) {
      const uint32_t canon = (uint32_t)(p.first >> 32) - 1;
      const int32_t eh = (int32_t)(uint32_t)p.first;
      if (canon >= e->issuers.size()) continue;
      auto it = e->expiry.find(make_key(e, eh, canon));
      const int64_t t = it != e->expiry.end() ? it->second : (int64_t)eh * 3600;
      if (t <= now)
        hipLaunchKernelGGL(k_sweep, dim3(blocks), dim3(256), 0, e->stream, e->table, e->nslots, 0, 0ll, (uint32_t)eh,
                           canon, e->issuer_counts, e->pairs, e->npairs - 1, e->d_count);
    }
  }
  unsigned long long cnt;
  HIPCHK(e, hipMemcpyAsync(&cnt, e->d_count, 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  total += cnt;
  // (2) host-side keys
  for (auto it = e->hstore.begin(); it != e->hstore.end();) {
    int64_t t;
    bool has = false;
    auto ex = e->expiry.find(it->first);
    int32_t eh; uint32_t canon;
    const bool tk = table_key(e, it->first.data(), it->first.size(), &eh, &canon);
    if (ex != e->expiry.end()) { t = ex->second; has = true; }
    else if (tk) { t = (int64_t)eh * 3600; has = true; }
    if (has && t <= now) {
      total += it->second.size();
      if (tk) e->host_issuer_counts[canon] -= it->second.size();
      it = e->hstore.erase(it);
    } else {
      ++it;
    }
  }
  for (auto it = e->expiry.begin(); it != e->expiry.end();)
    it = it->second <= now ? e->expiry.erase(it) : std::next(it);
  if (removed) *removed = total;
  return CTMR_OK;
}

int ctmr_issuer_counts(ctmr_engine* e, uint64_t* out, uint32_t n) {
  if (!e || (n && !out)) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  const uint32_t have = (uint32_t)e->issuers.size();
  std::vector<unsigned long long> c(have);
  if (have) HIPCHK(e, hipMemcpy(c.data(), e->issuer_counts, (size_t)have * 8, hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < n; i++) {
    if (i >= have) { out[i] = 0; continue; }
    const uint32_t canon = e->issuers[i].canon;
    uint64_t v = c[canon];
    auto it = e->host_issuer_counts.find(canon);
    if (it != e->host_issuer_counts.end()) v += it->second;
    out[i] = e->issuers[i].valid ? v : 0;
  }
  return CTMR_OK;
}

int ctmr_total_count(ctmr_engine* e, uint64_t* out) {
  if (!e || !out) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  const uint32_t have = (uint32_t)e->issuers.size();
  std::vector<unsigned long long> c(have);
  if (have) HIPCHK(e, hipMemcpy(c.data(), e->issuer_counts, (size_t)have * 8, hipMemcpyDeviceToHost));
  uint64_t t = 0;
  for (uint32_t i = 0; i < have; i++) if (e->issuers[i].canon == i) t += c[i];
  for (auto& kv : e->host_issuer_counts) t += kv.second;
  *out = t;
  return CTMR_OK;
}

int ctmr_issuer_counts_device(ctmr_engine* e, void** d, uint32_t* n) {
  if (!e || !d) return CTMR_E_INVAL;
  *d = e->issuer_counts;
  if (n) *n = e->max_issuers;
  return CTMR_OK;
}

int ctmr_reset_known(ctmr_engine* e) {
  if (!e) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, hipMemsetAsync(e->table, 0, e->nslots * sizeof(Slot), e->stream));
  HIPCHK(e, hipMemsetAsync(e->pairs, 0, e->npairs * sizeof(PairSlot), e->stream));
  e->pairs_dirty = false;
  HIPCHK(e, hipMemsetAsync(e->issuer_counts, 0, (size_t)e->max_issuers * 8, e->stream));
  for (auto it = e->hstore.begin(); it != e->hstore.end();)
    it = it->first.compare(0, 9, "serials::") == 0 ? e->hstore.erase(it) : std::next(it);
  e->host_issuer_counts.clear();
  if (e->d_bloom) HIPCHK(e, hipMemsetAsync(e->d_bloom, 0, e->bloom_words * 8, e->stream));
  e->bloom_round_epoch = 0;
  return CTMR_OK;
}

// ------------------------------------------------------------------ synthetic generator

uint32_t ctmr_synth_leaf_len(const ctmr_synth_config* c, uint64_t i) {
  const auto& cdf = host_cdf(c->n_issuers ? c->n_issuers : 1);
  SynthCfg s = to_synth(c, cdf.data());
  BackWriter w{nullptr, SYNTH_MAX_LEN};
  uint32_t iss; uint8_t et;
  synth_leaf_emit(s, i, w, iss, et);
  return SYNTH_MAX_LEN - w.pos;
}

uint32_t ctmr_synth_leaf(const ctmr_synth_config* c, uint64_t i, uint8_t* out, uint32_t cap,
                         uint32_t* issuer_idx, uint8_t* entry_type) {
  const auto& cdf = host_cdf(c->n_issuers ? c->n_issuers : 1);
  SynthCfg s = to_synth(c, cdf.data());
  uint8_t tmp[SYNTH_MAX_LEN];
  BackWriter w{tmp, SYNTH_MAX_LEN};
  uint32_t iss; uint8_t et;
  synth_leaf_emit(s, i, w, iss, et);
  const uint32_t len = SYNTH_MAX_LEN - w.pos;
  if (issuer_idx) *issuer_idx = iss;
  if (entry_type) *entry_type = et;
  if (out && len <= cap) memcpy(out, tmp + w.pos, len);
  return len;
}

uint32_t ctmr_synth_issuer(const ctmr_synth_config* c, uint32_t k, uint8_t* out, uint32_t cap) {
  SynthCfg s = to_synth(c, nullptr);
  uint8_t tmp[SYNTH_MAX_LEN];
  BackWriter w{tmp, SYNTH_MAX_LEN};
  synth_issuer_emit(s, k, w);
  const uint32_t len = SYNTH_MAX_LEN - w.pos;
  if (out && len <= cap) memcpy(out, tmp + w.pos, len);
  return len;
}

uint64_t ctmr_synth_host(const ctmr_synth_config* c, uint64_t first, uint64_t n, uint64_t* offsets,
                         uint8_t* payload, uint64_t cap, uint32_t* issuer_idx, uint8_t* entry_type) {
  const auto& cdf = host_cdf(c->n_issuers ? c->n_issuers : 1);
  SynthCfg s = to_synth(c, cdf.data());
  uint8_t tmp[SYNTH_MAX_LEN];
  uint64_t at = 0;
  if (offsets) offsets[0] = 0;
  for (uint64_t i = 0; i < n; i++) {
    BackWriter w{tmp, SYNTH_MAX_LEN};
    uint32_t iss; uint8_t et;
    synth_leaf_emit(s, first + i, w, iss, et);
    const uint32_t len = SYNTH_MAX_LEN - w.pos;
    if (payload && at + len <= cap) memcpy(payload + at, tmp + w.pos, len);
    at += len;
    if (offsets) offsets[i + 1] = at;
    if (issuer_idx) issuer_idx[i] = iss;
    if (entry_type) entry_type[i] = et;
  }
  return at;
}

int ctmr_synth_device(ctmr_engine* e, const ctmr_synth_config* c, uint64_t first, uint64_t n,
                      uint64_t* d_offsets, uint8_t* d_payload, uint64_t payload_cap,
                      uint32_t* d_issuer_idx, uint8_t* d_entry_type, uint64_t* payload_bytes) {
  if (!e || !c || !d_offsets) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (n == 0) return CTMR_OK;
  const uint32_t ni = c->n_issuers ? c->n_issuers : 1;
  const auto& cdf = host_cdf(ni);
  int r;
  if ((r = ensure(e, SC_MISC, (size_t)ni * 4))) return r;
  HIPCHK(e, hipMemcpyAsync(e->d_scratch[SC_MISC], cdf.data(), (size_t)ni * 4, hipMemcpyHostToDevice, e->stream));
  SynthCfg s = to_synth(c, (const uint32_t*)e->d_scratch[SC_MISC]);
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_synth_len, dim3(blocks), dim3(256), 0, e->stream, s, first, n, d_offsets);
  size_t tmp_bytes = 0;
  HIPCHK(e, hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, d_offsets + 1, d_offsets + 1, (int64_t)n, e->stream));
  if ((r = ensure(e, SC_STAGE_B, tmp_bytes))) return r;
  HIPCHK(e, hipcub::DeviceScan::InclusiveSum(e->d_scratch[SC_STAGE_B], tmp_bytes, d_offsets + 1, d_offsets + 1, (int64_t)n, e->stream));
  uint64_t total = 0;
  HIPCHK(e, hipMemcpyAsync(&total, d_offsets + n, 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (payload_bytes) *payload_bytes = total;
  if (!d_payload) return CTMR_OK;
  if (total + CTMR_PAYLOAD_PAD > payload_cap) return fail(e, CTMR_E_RANGE, "payload needs %llu bytes (+%d pad)", (unsigned long long)total, CTMR_PAYLOAD_PAD);
  if (!d_issuer_idx || !d_entry_type) return CTMR_E_INVAL;
  hipLaunchKernelGGL(k_synth_emit, dim3(blocks), dim3(256), 0, e->stream, s, first, n, (const uint64_t*)d_offsets,
                     d_payload, d_issuer_idx, d_entry_type);
  HIPCHK(e, hipStreamSynchronize(e->stream));
  HIPCHK(e, hipGetLastError());
  return CTMR_OK;
}


uint64_t ctmr_synth_entries_host(const ctmr_synth_config* c, uint64_t first, uint64_t n, uint64_t* bounds,
                                 uint8_t* blob, uint64_t cap) {
  const auto& cdf = host_cdf(c->n_issuers ? c->n_issuers : 1);
  SynthCfg s = to_synth(c, cdf.data());
  std::vector<uint8_t> tmp(SYNTH_ENTRY_MAX);
  uint64_t at = 0;
  if (bounds) bounds[0] = 0;
  for (uint64_t i = 0; i < n; i++) {
    BackWriter w{tmp.data(), SYNTH_ENTRY_MAX};
    const uint32_t leaf = synth_entry_emit(s, first + i, w);
    const uint32_t len = SYNTH_ENTRY_MAX - w.pos;
    if (blob && at + len <= cap) memcpy(blob + at, tmp.data() + w.pos, len);
    if (bounds) {
      bounds[2 * i + 1] = at + leaf;
      bounds[2 * i + 2] = at + len;
    }
    at += len;
  }
  return at;
}

int ctmr_synth_entries_device(ctmr_engine* e, const ctmr_synth_config* c, uint64_t first, uint64_t n,
                              uint64_t* d_bounds, uint8_t* d_blob, uint64_t blob_cap, uint64_t* blob_bytes) {
  if (!e || !c || !d_bounds) return CTMR_E_INVAL;
  std::lock_guard<std::mutex> g(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (n == 0) return CTMR_OK;
  const uint32_t ni = c->n_issuers ? c->n_issuers : 1;
  const auto& cdf = host_cdf(ni);
  int r;
  if ((r = ensure(e, SC_MISC, (size_t)ni * 4))) return r;
  HIPCHK(e, hipMemcpyAsync(e->d_scratch[SC_MISC], cdf.data(), (size_t)ni * 4, hipMemcpyHostToDevice, e->stream));
  SynthCfg s = to_synth(c, (const uint32_t*)e->d_scratch[SC_MISC]);
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_synth_entries_len, dim3(blocks), dim3(256), 0, e->stream, s, first, n, d_bounds);
  size_t tmp_bytes = 0;
  HIPCHK(e, hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, d_bounds + 1, d_bounds + 1, (int64_t)(2 * n), e->stream));
  if ((r = ensure(e, SC_TMP, tmp_bytes))) return r;
  HIPCHK(e, hipcub::DeviceScan::InclusiveSum(e->d_scratch[SC_TMP], tmp_bytes, d_bounds + 1, d_bounds + 1, (int64_t)(2 * n), e->stream));
  uint64_t total = 0;
  HIPCHK(e, hipMemcpyAsync(&total, d_bounds + 2 * n, 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (blob_bytes) *blob_bytes = total;
  if (!d_blob) return CTMR_OK;
  if (total + CTMR_PAYLOAD_PAD > blob_cap) return fail(e, CTMR_E_RANGE, "blob needs %llu bytes (+%d pad)", (unsigned long long)total, CTMR_PAYLOAD_PAD);
  hipLaunchKernelGGL(k_synth_entries_emit, dim3(blocks), dim3(256), 0, e->stream, s, first, n, (const uint64_t*)d_bounds, d_blob);
  HIPCHK(e, hipStreamSynchronize(e->stream));
  HIPCHK(e, hipGetLastError());
  return CTMR_OK;
}

}  // extern "C"
