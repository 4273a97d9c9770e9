// fake_rccl.cpp — a stand-in for librccl that lets SEVERAL RANKS SHARE ONE GPU, as threads of one process or as
// processes of their own (bench.py --gpus N spawns one process per rank), so that the RCCL transport of
// csrc/engine/group.inc (grouped ncclSend/ncclRecv all-to-all, in-place ncclAllGather of the control rows and of the
// Bloom filters, ncclAllReduce of the per-issuer counts) can be driven with a world of 2–4 where only one MI355X is
// reachable.  TEST INFRASTRUCTURE ONLY: the tests build it and hand it to the library through CTMR_RCCL_LIB; the product
// never loads it on its own.
//
// Transport: POSIX shared memory.  A communicator world is a control block /dev/shm/<id> (rank count, a sense-reversing
// barrier); every message, all-gather piece and all-reduce operand is staged through a shared-memory object of its own
// (device → host copy by the sender, host → device copy by the receiver).  Slow, and exactly as good for threads as
// for processes.
//
// Semantics kept from NCCL: a communicator per rank created collectively from one unique id; point-to-point operations
// between ncclGroupStart/ncclGroupEnd complete together; collectives are called by every rank with matching counts;
// in-place all-gather (sendbuff = recvbuff + rank * count) and in-place all-reduce.  Simplification: every operation is
// synchronous (the calling rank's stream is drained before data is exposed and after it has been copied) — ordering
// bugs of the CALLER that real NCCL's stream semantics would forgive are still forgiven, races it would expose are not
// reproduced.  What this catches is what can be wrong in group.inc itself: counts, offsets, peers, in-place rules —
// and in bench.py's launcher: ids handed to child processes, ranks, environment.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct ncclComm;
typedef ncclComm* ncclComm_t;
}

namespace {

constexpr int MAX_RANKS = 16;

struct Ctl {  // lives in shared memory
  std::atomic<uint32_t> ready;
  uint32_t n;
  std::atomic<uint32_t> joined, left;
  std::atomic<uint32_t> waiting, generation;
  uint64_t msg_bytes[MAX_RANKS][MAX_RANKS];  // [src][dst], valid between the barriers of one p2p round (+1: 0 = not posted)
};

void nap() {
  struct timespec ts = {0, 50000};
  nanosleep(&ts, nullptr);
}

void barrier(Ctl* c) {
  const uint32_t g = c->generation.load(std::memory_order_acquire);
  if (c->waiting.fetch_add(1, std::memory_order_acq_rel) + 1 == c->n) {
    c->waiting.store(0, std::memory_order_relaxed);
    c->generation.fetch_add(1, std::memory_order_acq_rel);
  } else {
    while (c->generation.load(std::memory_order_acquire) == g) nap();
  }
}

size_t dsize(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: return 4;
    default: return 8;
  }
}

// one staged piece of device memory in a shared-memory object
bool stage_out(const std::string& name, const void* d_src, size_t bytes) {
  const int fd = shm_open(name.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0600);
  if (fd < 0) return false;
  bool ok = true;
  if (bytes) {
    ok = ftruncate(fd, (off_t)bytes) == 0;
    void* p = ok ? mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : MAP_FAILED;
    ok = ok && p != MAP_FAILED;
    if (ok) {
      ok = hipMemcpy(p, d_src, bytes, hipMemcpyDeviceToHost) == hipSuccess;
      munmap(p, bytes);
    }
  }
  close(fd);
  return ok;
}
// map it for reading (the reader copies it wherever it likes)
const void* stage_map(const std::string& name, size_t bytes, int* fd_out) {
  *fd_out = -1;
  if (!bytes) return nullptr;
  const int fd = shm_open(name.c_str(), O_RDONLY, 0600);
  if (fd < 0) return nullptr;
  struct stat st;
  if (fstat(fd, &st) != 0 || (size_t)st.st_size != bytes) { close(fd); return nullptr; }
  void* p = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
  if (p == MAP_FAILED) { close(fd); return nullptr; }
  *fd_out = fd;
  return p;
}
void stage_unmap(const void* p, size_t bytes, int fd) {
  if (p) munmap((void*)p, bytes);
  if (fd >= 0) close(fd);
}

struct Op { bool send; const void* sp; void* rp; size_t bytes; int peer; hipStream_t stream; };
thread_local int tl_depth = 0;
thread_local std::vector<Op> tl_ops;
thread_local ncclComm* tl_comm = nullptr;

std::mutex g_mu;
uint64_t g_next_id = 1;

}  // namespace

struct ncclComm {
  Ctl* ctl;
  std::string id;  // name of the control block ("/fr-…")
  int rank;
  std::string piece(const char* kind, int a, int b = -1) const {
    char buf[200];
    snprintf(buf, sizeof buf, "%s-%s-%d-%d", id.c_str(), kind, a, b);
    return buf;
  }
};

namespace {

ncclResult_t run_p2p(ncclComm* c, std::vector<Op>& ops) {
  Ctl* w = c->ctl;
  hipStream_t st = nullptr;
  for (auto& o : ops) st = o.stream;
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;  // what I send is complete
  ncclResult_t rc = ncclSuccess;
  for (auto& o : ops)
    if (o.send) {
      if (w->msg_bytes[c->rank][o.peer]) { rc = ncclInvalidUsage; continue; }  // two sends to one peer in one group: group.inc never does that
      if (!stage_out(c->piece("p2p", c->rank, o.peer), o.sp, o.bytes)) rc = ncclSystemError;
      w->msg_bytes[c->rank][o.peer] = o.bytes + 1;
    }
  barrier(w);  // every send of this round is posted
  for (auto& o : ops)
    if (!o.send) {
      const uint64_t posted = w->msg_bytes[o.peer][c->rank];
      if (!posted || posted - 1 != o.bytes) { rc = ncclInvalidArgument; continue; }  // a recv without its send / size mismatch
      int fd;
      const void* p = stage_map(c->piece("p2p", o.peer, c->rank), o.bytes, &fd);
      if (o.bytes && !p) { rc = ncclSystemError; continue; }
      if (o.bytes && hipMemcpy(o.rp, p, o.bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
      stage_unmap(p, o.bytes, fd);
    }
  barrier(w);  // every copy is done: the staging objects may go
  for (auto& o : ops)
    if (o.send) {
      shm_unlink(c->piece("p2p", c->rank, o.peer).c_str());
      w->msg_bytes[c->rank][o.peer] = 0;
    }
  barrier(w);
  return rc;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::lock_guard<std::mutex> lk(g_mu);
  memset(id, 0, sizeof *id);
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  snprintf(id->internal, sizeof id->internal, "/fr-%d-%llu-%llx", (int)getpid(), (unsigned long long)g_next_id++,
           (unsigned long long)ts.tv_nsec ^ ((unsigned long long)ts.tv_sec << 20));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  const std::string name(id.internal, strnlen(id.internal, sizeof id.internal));
  if (name.size() < 4 || name[0] != '/') return ncclInvalidArgument;
  bool creator = true;
  int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) {
    creator = false;
    for (int tries = 0; tries < 200000 && fd < 0; tries++) {
      fd = shm_open(name.c_str(), O_RDWR, 0600);
      if (fd < 0) nap();
    }
    if (fd < 0) return ncclSystemError;
  }
  if (creator && ftruncate(fd, sizeof(Ctl)) != 0) { close(fd); return ncclSystemError; }
  if (!creator) {  // the creator sizes the object before anybody maps it
    struct stat st;
    for (int tries = 0; tries < 200000; tries++) {
      if (fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(Ctl)) break;
      nap();
    }
  }
  void* p = mmap(nullptr, sizeof(Ctl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  Ctl* ctl = (Ctl*)p;
  if (creator) {
    ctl->n = (uint32_t)nranks;  // (a fresh shared-memory object is zero-filled)
    ctl->ready.store(1, std::memory_order_release);
  } else {
    while (ctl->ready.load(std::memory_order_acquire) != 1) nap();
    if (ctl->n != (uint32_t)nranks) return ncclInvalidArgument;
  }
  ctl->joined.fetch_add(1);
  *comm = new ncclComm{ctl, name, rank};
  barrier(ctl);  // collective: returns when every rank has joined
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (!comm) return ncclSuccess;
  Ctl* ctl = comm->ctl;
  if (ctl->left.fetch_add(1) + 1 == ctl->n) shm_unlink(comm->id.c_str());  // the last one out
  munmap(ctl, sizeof(Ctl));
  delete comm;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "success";
    case ncclInvalidArgument: return "fake rccl: invalid argument (recv without matching send, or sizes differ)";
    case ncclInvalidUsage: return "fake rccl: invalid usage";
    case ncclUnhandledCudaError: return "fake rccl: HIP error";
    case ncclSystemError: return "fake rccl: shared-memory staging failed";
    default: return "fake rccl: error";
  }
}

ncclResult_t ncclGroupStart() {
  tl_depth++;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (tl_depth <= 0) return ncclInvalidUsage;
  if (--tl_depth > 0) return ncclSuccess;
  ncclResult_t rc = ncclSuccess;
  // every rank calls GroupEnd once per exchange round (group.inc), with or without operations of its own: the round's
  // barriers need all of them.  A rank without operations still has to know its communicator: it is the one used last.
  if (tl_comm) rc = run_p2p(tl_comm, tl_ops);
  tl_ops.clear();
  return rc;
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
  tl_comm = comm;
  tl_ops.push_back(Op{true, sendbuff, nullptr, count * dsize(t), peer, stream});
  if (tl_depth == 0) {
    const ncclResult_t rc = run_p2p(comm, tl_ops);
    tl_ops.clear();
    return rc;
  }
  return ncclSuccess;
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
  tl_comm = comm;
  tl_ops.push_back(Op{false, nullptr, recvbuff, count * dsize(t), peer, stream});
  if (tl_depth == 0) {
    const ncclResult_t rc = run_p2p(comm, tl_ops);
    tl_ops.clear();
    return rc;
  }
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t t, ncclComm_t comm,
                           hipStream_t stream) {
  tl_comm = comm;
  Ctl* w = comm->ctl;
  const size_t bytes = sendcount * dsize(t);
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  ncclResult_t rc = ncclSuccess;
  if (!stage_out(comm->piece("ag", comm->rank), sendbuff, bytes)) rc = ncclSystemError;
  barrier(w);
  for (int r = 0; r < (int)w->n; r++) {
    void* dst = (uint8_t*)recvbuff + (size_t)r * bytes;
    if (r == comm->rank && dst == sendbuff) continue;  // in place
    int fd;
    const void* p = stage_map(comm->piece("ag", r), bytes, &fd);
    if (bytes && !p) { rc = ncclSystemError; continue; }
    if (bytes && hipMemcpy(dst, p, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
    stage_unmap(p, bytes, fd);
  }
  barrier(w);
  shm_unlink(comm->piece("ag", comm->rank).c_str());
  barrier(w);
  return rc;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
  tl_comm = comm;
  if (t != ncclUint64 && t != ncclInt64) return ncclInvalidArgument;  // all the library reduces
  if (op != ncclSum && op != ncclMax) return ncclInvalidArgument;
  Ctl* w = comm->ctl;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  ncclResult_t rc = ncclSuccess;
  if (!stage_out(comm->piece("ar", comm->rank), sendbuff, count * 8)) rc = ncclSystemError;
  barrier(w);
  std::vector<uint64_t> acc(count, 0);
  for (int r = 0; r < (int)w->n; r++) {
    int fd;
    const uint64_t* p = (const uint64_t*)stage_map(comm->piece("ar", r), count * 8, &fd);
    if (count && !p) { rc = ncclSystemError; continue; }
    for (size_t k = 0; k < count; k++) acc[k] = op == ncclSum ? acc[k] + p[k] : (p[k] > acc[k] ? p[k] : acc[k]);
    stage_unmap(p, count * 8, fd);
  }
  barrier(w);  // everybody has read everybody's input: in-place results may be written, the operands may go
  shm_unlink(comm->piece("ar", comm->rank).c_str());
  if (count && hipMemcpy(recvbuff, acc.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
  barrier(w);
  return rc;
}

}  // extern "C"
