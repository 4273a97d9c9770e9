// host_writeback.cpp — the HOST half of FilesystemDatabase.Store's write-back for a batch the GPU has mapped
// (storage/filesystemdatabase.go:183-208): for every newly unknown certificate backend.StoreCertificatePEM(serial,
// expDate, issuer, pem) — the PEM block was encoded on the GPU (k_pem_encode) and copied to pinned host memory — and
// markDirty for the days the batch touched.  The backends are the C++ restatement of the reference's storage package
// (include/ctmr_storage.hpp: NoopBackend = storage/noopbackend.go, LocalDiskBackend = storage/localdiskbackend.go:188-199 with
// its path layout root/<expDate>/<issuerID>/<serialID> and the CWD-relative dirty marker of :89-91) — what a Go host keeps
// doing in Go; here it is native code behind a small C interface so that bench.py --stream --write-back can
// time the whole of BASELINE configs[4] and say how far a filesystem lags the GPU.  No GPU code, no libctmr dependency.
//
// Asynchronous: submit() hands a chunk (PEM bytes + offsets + the 32-byte records of the new certificates, all in host
// memory that stays untouched until wait()) to `threads` workers and returns; wait() joins them.  Two chunks may be in
// flight (double buffering: the GPU maps wave k+1 while the host stores wave k).
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ctmr_storage.hpp"

using namespace ctmr::storage;

namespace {
struct Job {
  std::vector<std::thread> workers;
  std::atomic<uint64_t> files{0}, bytes{0}, skipped{0}, failed{0};
  std::string err;
  std::mutex mu;
  std::chrono::steady_clock::time_point t0;
  double seconds = 0;
};
}  // namespace

struct ctmr_host_writer {
  std::unique_ptr<StorageBackend> backend;
  bool disk = false;
  int threads = 1;
  std::vector<Issuer> issuers;
  std::vector<std::unique_ptr<Job>> jobs;
  std::string err;
};

extern "C" {

// root == NULL or "": NoopBackend (engine.GetConfiguredStorage with an empty certPath, engine/engine.go:36-40); else
// LocalDiskBackend(0644, root).
ctmr_host_writer* ctmr_host_writer_open(const char* root, int threads) {
  auto* w = new ctmr_host_writer;
  w->threads = threads > 0 ? threads : 1;
  if (root && root[0]) {
    w->backend.reset(new LocalDiskBackend(0644, root));
    w->disk = true;
  } else {
    w->backend.reset(new NoopBackend);
  }
  return w;
}

// issuer IDs by issuer index: n strings of 44 characters (Issuer.ID(), types.go:124-130), back to back
int ctmr_host_writer_set_issuers(ctmr_host_writer* w, const char* ids44, uint32_t n) {
  if (!w || (n && !ids44)) return -1;
  w->issuers.clear();
  for (uint32_t k = 0; k < n; k++) w->issuers.push_back(Issuer::FromString(std::string(ids44 + (size_t)k * 44, 44)));
  return 0;
}

// One chunk of new certificates: PEM k = pem[pem_off[k], pem_off[k+1]), its record recs[k] (status PASS; issuer_idx,
// exp_hour and the serial octets name the file).  Serials beyond the 20 octets a record carries are SKIPPED and counted
// (the full binding parses those certificates on the host, FilesystemDatabase::afterMap).  Returns a job id (>= 0).
int ctmr_host_writer_submit(ctmr_host_writer* w, const uint8_t* pem, const uint64_t* pem_off, const ctmr_record* recs,
                            uint64_t count) {
  if (!w || (count && (!pem || !pem_off || !recs))) return -1;
  std::unique_ptr<Job> job(new Job);
  Job* j = job.get();
  j->t0 = std::chrono::steady_clock::now();
  const int T = w->threads;
  for (int t = 0; t < T; t++) {
    const uint64_t lo = count * (uint64_t)t / (uint64_t)T, hi = count * (uint64_t)(t + 1) / (uint64_t)T;
    if (lo == hi) continue;
    j->workers.emplace_back([w, j, pem, pem_off, recs, lo, hi]() {
      uint64_t files = 0, bytes = 0, skipped = 0, failed = 0;
      uint64_t k = lo;
      try {
        for (; k < hi; k++) {
          const ctmr_record& r = recs[k];
          if (r.serial_len > 20 || r.issuer_idx >= w->issuers.size()) {
            skipped++;
            continue;
          }
          const uint64_t len = pem_off[k + 1] - pem_off[k];
          w->backend->StoreCertificatePEM(Serial::FromBytes(std::string((const char*)r.serial, r.serial_len)),
                                          ExpDate::FromHour(r.exp_hour), w->issuers[r.issuer_idx],
                                          std::string((const char*)pem + pem_off[k], len));
          files++;
          bytes += len;
        }
      } catch (const std::exception& ex) {  // the backend threw at certificate k: it and the rest of this worker's range were not stored
        failed = hi - k;
        std::lock_guard<std::mutex> g(j->mu);
        j->err = ex.what();
      }
      j->files += files;
      j->bytes += bytes;
      j->skipped += skipped;
      j->failed += failed;
    });
  }
  // job ids are slots: a finished job's slot is free again (a double-buffered stream uses two for ever)
  for (size_t id = 0; id < w->jobs.size(); id++)
    if (!w->jobs[id]) {
      w->jobs[id] = std::move(job);
      return (int)id;
    }
  w->jobs.push_back(std::move(job));
  return (int)w->jobs.size() - 1;
}

// Joins the job's workers.  out[0] files handed to the backend, out[1] their PEM bytes, out[2] skipped (long serials);
// *seconds = wall time from submit to the last worker's end.  Returns 0, or -1 with ctmr_host_writer_error().
int ctmr_host_writer_wait(ctmr_host_writer* w, int job, uint64_t out[3], double* seconds) {
  if (!w || job < 0 || (size_t)job >= w->jobs.size() || !w->jobs[job]) return -1;
  Job* j = w->jobs[job].get();
  for (auto& t : j->workers) t.join();
  j->workers.clear();
  j->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - j->t0).count();
  if (out) {
    out[0] = j->files;
    out[1] = j->bytes;
    out[2] = j->skipped;
  }
  if (seconds) *seconds = j->seconds;
  const bool bad = !j->err.empty();
  if (bad) w->err = j->err + " (" + std::to_string((unsigned long long)j->failed) + " certificates of this job were not stored)";
  w->jobs[job].reset();
  return bad ? -1 : 0;
}

// FilesystemDatabase.markDirty (filesystemdatabase.go:141-144, 204-208) for the days (unix days) a batch touched:
// backend.MarkDirty("2006-01-02") — LocalDiskBackend writes <id>/dirty relative to the CURRENT DIRECTORY, not to its root
// (localdiskbackend.go:89-91): the quirk is the reference's.
int ctmr_host_writer_mark_dirty(ctmr_host_writer* w, const int32_t* days, uint32_t n) {
  if (!w || (n && !days)) return -1;
  try {
    for (uint32_t k = 0; k < n; k++) w->backend->MarkDirty(Time::Unix((int64_t)days[k] * 86400).Format(false));
  } catch (const std::exception& ex) {
    w->err = ex.what();
    return -1;
  }
  return 0;
}

const char* ctmr_host_writer_error(ctmr_host_writer* w) { return w ? w->err.c_str() : "null writer"; }

void ctmr_host_writer_close(ctmr_host_writer* w) {
  if (!w) return;
  for (auto& j : w->jobs)
    if (j)
      for (auto& t : j->workers) t.join();
  delete w;
}

}  // extern "C"
