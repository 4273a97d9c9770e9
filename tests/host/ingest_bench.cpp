// ingest_bench.cpp — PCIe-inclusive ingestion rate of the host-buffer entry points, driven from C++ (what a cgo host
// sees: no Python between the calls): get-entries-sized batches (1 001 entries, cmd/ct-fetch/ct-fetch.go:417-424)
// through (a) the synchronous ctmr_map_batch and (b) ctmr_submit_batch / ctmr_wait with WINDOW tickets in flight, from
// pageable and from page-locked (ctmr_alloc_pinned) payload buffers.  Prints one JSON line.  Not a test: a measurement
// (scripts/run_ingest.sh); DESIGN.md §7 quotes it.  bench.py's `value` never includes PCIe.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "ctmr.h"
#include "ctmr_bench.h"

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(call)                                                                          \
  do {                                                                                    \
    int _r = (call);                                                                      \
    if (_r != CTMR_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, _r, ctmr_last_error(e)); exit(1); } \
  } while (0)

struct HostBatch {
  uint8_t* payload;  // pageable or pinned
  std::vector<uint64_t> offsets;
  std::vector<uint32_t> iss;
  std::vector<uint8_t> et;
  uint64_t n, bytes;
};

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1001;
  const int calls = argc > 2 ? atoi(argv[2]) : 4000;
  const bool raw_mode = argc > 3 && std::string(argv[3]) == "raw";
  ctmr_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.table_slots = 1ull << 25;
  ctmr_engine* e = nullptr;
  if (ctmr_create(&cfg, &e) != CTMR_OK) { fprintf(stderr, "no GPU\n"); return 2; }
  ctmr_synth_config sc;
  memset(&sc, 0, sizeof sc);
  sc.seed = 20260925; sc.n_issuers = 256; sc.zipf = 1; sc.ca_permille = 10; sc.expired_permille = 10;
  {  // issuers
    std::vector<uint8_t> blob;
    std::vector<uint64_t> off(1, 0);
    std::vector<uint8_t> tmp(4096);
    for (uint32_t k = 0; k < 256; k++) {
      const uint32_t l = ctmr_synth_issuer(&sc, k, tmp.data(), 4096);
      blob.insert(blob.end(), tmp.begin(), tmp.begin() + l);
      off.push_back(blob.size());
    }
    uint32_t first;
    CK(ctmr_add_issuers(e, blob.data(), off.data(), 256, &first));
  }
  const char* filt = "Synth Issuer 0,Synth Issuer 1";
  CK(ctmr_set_filter(e, filt, strlen(filt), 0, 1767225600ll));
  printf("{\"entries_per_batch\": %llu, \"batches\": %d, \"form\": \"%s\", \"results\": [", (unsigned long long)n, calls,
         raw_mode ? "raw get-entries (leaf_input + extra_data)" : "packed certificates");
  bool first_out = true;
  if (raw_mode) {
    // RAW get-entries responses (what the downloader holds, ct-fetch.go:446-462): ctmr_map_entries per response against
    // ctmr_submit_entries / ctmr_wait_entries with WINDOW tickets in flight.  The engine registers the issuers itself.
    struct RawBatch { uint8_t* blob; std::vector<uint64_t> bounds; uint64_t bytes; };
    for (int pinned = 0; pinned < 2; pinned++) {
      const int NB = 16;
      std::vector<RawBatch> rb(NB);
      for (int k = 0; k < NB; k++) {
        RawBatch& b = rb[k];
        b.bounds.resize(2 * n + 1);
        b.bytes = ctmr_synth_entries_host(&sc, (uint64_t)k * n, n, b.bounds.data(), nullptr, 0);
        if (pinned) { void* p; CK(ctmr_alloc_pinned(e, b.bytes + 64, &p)); b.blob = (uint8_t*)p; }
        else b.blob = (uint8_t*)malloc(b.bytes + 64);
        ctmr_synth_entries_host(&sc, (uint64_t)k * n, n, b.bounds.data(), b.blob, b.bytes + 64);
      }
      std::vector<ctmr_record> rec(n);
      std::vector<uint64_t> nw(n), ts(n);
      ctmr_batch_stats st;
      ctmr_decode_stats ds;
      CK(ctmr_reset_known(e));
      CK(ctmr_map_entries(e, rb[0].blob, rb[0].bounds.data(), n, rec.data(), nw.data(), ts.data(), &ds, &st));
      CK(ctmr_reset_known(e));
      const int sync_calls = calls / 8 > 50 ? calls / 8 : 50;
      double t0 = now_s();
      for (int k = 0; k < sync_calls; k++) {
        RawBatch& b = rb[k % NB];
        CK(ctmr_map_entries(e, b.blob, b.bounds.data(), n, rec.data(), nw.data(), ts.data(), &ds, &st));
      }
      double dt = now_s() - t0;
      printf("%s{\"entry_point\": \"ctmr_map_entries\", \"payload_memory\": \"%s\", \"us_per_batch\": %.1f, \"entries_per_s\": %.0f, \"payload_GBps\": %.2f}",
             first_out ? "" : ", ", pinned ? "pinned" : "pageable", dt / sync_calls * 1e6, n * sync_calls / dt,
             (double)rb[0].bytes * sync_calls / dt / 1e9);
      first_out = false;
      for (int window : {8, 32, 64}) {
        // a raw super-batch closes at 96 MB ≈ 31 000 entries of 3 KB; at most 4 may be unfinished or uncollected
        if ((uint64_t)window * rb[0].bytes > 2 * (96ull << 20)) continue;
        CK(ctmr_reset_known(e));
        std::deque<ctmr_ticket> fl;
        uint64_t tot_new = 0;
        t0 = now_s();
        for (int k = 0; k < calls; k++) {
          RawBatch& b = rb[k % NB];
          ctmr_ticket t;
          CK(ctmr_submit_entries(e, b.blob, b.bounds.data(), n, &t));
          fl.push_back(t);
          if ((int)fl.size() > window) {
            CK(ctmr_wait_entries(e, fl.front(), rec.data(), nw.data(), ts.data(), &ds, &st));
            tot_new += st.n_new;
            fl.pop_front();
          }
        }
        while (!fl.empty()) {
          CK(ctmr_wait_entries(e, fl.front(), rec.data(), nw.data(), ts.data(), &ds, &st));
          tot_new += st.n_new;
          fl.pop_front();
        }
        dt = now_s() - t0;
        printf(", {\"entry_point\": \"ctmr_submit_entries/ctmr_wait_entries\", \"payload_memory\": \"%s\", \"tickets_in_flight\": %d, \"us_per_batch\": %.1f, "
               "\"entries_per_s\": %.0f, \"payload_GBps\": %.2f, \"n_new\": %llu}",
               pinned ? "pinned" : "pageable", window, dt / calls * 1e6, n * calls / dt, (double)rb[0].bytes * calls / dt / 1e9,
               (unsigned long long)tot_new);
      }
      for (auto& b : rb) {
        if (pinned) ctmr_free_pinned(e, b.blob);
        else free(b.blob);
      }
    }
    printf("]}\n");
    ctmr_destroy(e);
    return 0;
  }
  for (int pinned = 0; pinned < 2; pinned++) {
    const int NB = 16;  // distinct batches, reused round-robin (the table is reset per leg: the first pass inserts, later passes find duplicates)
    std::vector<HostBatch> hb(NB);
    for (int k = 0; k < NB; k++) {
      HostBatch& b = hb[k];
      b.n = n;
      b.offsets.resize(n + 1);
      b.iss.resize(n);
      b.et.resize(n);
      const uint64_t need = ctmr_synth_host(&sc, (uint64_t)k * n, n, b.offsets.data(), nullptr, 0, b.iss.data(), b.et.data());
      b.bytes = need;
      if (pinned) { void* p; CK(ctmr_alloc_pinned(e, need + 64, &p)); b.payload = (uint8_t*)p; }
      else b.payload = (uint8_t*)malloc(need + 64);
      ctmr_synth_host(&sc, (uint64_t)k * n, n, b.offsets.data(), b.payload, need + 64, b.iss.data(), b.et.data());
    }
    std::vector<ctmr_record> rec(n);
    std::vector<uint64_t> nw(n);
    ctmr_batch_stats st;
    // (a) synchronous
    CK(ctmr_reset_known(e));
    CK(ctmr_map_batch(e, hb[0].payload, hb[0].offsets.data(), hb[0].iss.data(), hb[0].et.data(), n, rec.data(), nw.data(), &st));
    CK(ctmr_reset_known(e));
    const int sync_calls = calls / 8 > 50 ? calls / 8 : 50;
    double t0 = now_s();
    for (int k = 0; k < sync_calls; k++) {
      HostBatch& b = hb[k % NB];
      CK(ctmr_map_batch(e, b.payload, b.offsets.data(), b.iss.data(), b.et.data(), n, rec.data(), nw.data(), &st));
    }
    double dt = now_s() - t0;
    printf("%s{\"entry_point\": \"ctmr_map_batch\", \"payload_memory\": \"%s\", \"us_per_batch\": %.1f, \"certs_per_s\": %.0f, \"payload_GBps\": %.2f}",
           first_out ? "" : ", ", pinned ? "pinned" : "pageable", dt / sync_calls * 1e6, n * sync_calls / dt,
           (double)hb[0].bytes * sync_calls / dt / 1e9);
    first_out = false;
    // (b) asynchronous, WINDOW tickets in flight
    for (int window : {8, 64, 192}) {
      if ((uint64_t)window * n > 196608) continue;  // at most 4 super-batches of 65 536 entries may be uncollected
      CK(ctmr_reset_known(e));
      std::deque<ctmr_ticket> fl;
      uint64_t tot_new = 0;
      t0 = now_s();
      for (int k = 0; k < calls; k++) {
        HostBatch& b = hb[k % NB];
        ctmr_ticket t;
        CK(ctmr_submit_batch(e, b.payload, b.offsets.data(), b.iss.data(), b.et.data(), n, &t));
        fl.push_back(t);
        if ((int)fl.size() > window) {
          CK(ctmr_wait(e, fl.front(), rec.data(), nw.data(), &st));
          tot_new += st.n_new;
          fl.pop_front();
        }
      }
      while (!fl.empty()) {
        CK(ctmr_wait(e, fl.front(), rec.data(), nw.data(), &st));
        tot_new += st.n_new;
        fl.pop_front();
      }
      dt = now_s() - t0;
      printf(", {\"entry_point\": \"ctmr_submit_batch/ctmr_wait\", \"payload_memory\": \"%s\", \"tickets_in_flight\": %d, \"us_per_batch\": %.1f, "
             "\"certs_per_s\": %.0f, \"payload_GBps\": %.2f, \"n_new\": %llu}",
             pinned ? "pinned" : "pageable", window, dt / calls * 1e6, n * calls / dt, (double)hb[0].bytes * calls / dt / 1e9,
             (unsigned long long)tot_new);
    }
    for (auto& b : hb) {
      if (pinned) ctmr_free_pinned(e, b.payload);
      else free(b.payload);
    }
  }
  printf("]}\n");
  ctmr_destroy(e);
  return 0;
}
