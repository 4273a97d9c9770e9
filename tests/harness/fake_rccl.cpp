// fake_rccl.cpp — a stand-in for librccl that lets SEVERAL RANKS LIVE AS THREADS OF ONE PROCESS ON ONE GPU, so that the
// RCCL transport of csrc/engine/group.inc (grouped ncclSend/ncclRecv all-to-all, in-place ncclAllGather of the counts
// matrix and of the Bloom filters, ncclAllReduce of the per-issuer counts) can be driven with a world of 2–4 where only
// one MI355X is reachable.  TEST INFRASTRUCTURE ONLY: tests/test_gpu_rccl_transport.py builds it and hands it to the
// library through CTMR_RCCL_LIB; the product never loads it on its own.
//
// Semantics kept from NCCL: a communicator per rank created collectively from one unique id; point-to-point operations
// between ncclGroupStart/ncclGroupEnd complete together; collectives are called by every rank with matching counts;
// in-place all-gather (sendbuff = recvbuff + rank * count) and in-place all-reduce.  Simplification: every operation is
// synchronous (the calling rank's stream is drained before data is exposed and after it has been copied) — ordering
// bugs of the CALLER that real NCCL's stream semantics would forgive are still forgiven, races it would expose are not
// reproduced.  What this catches is what can be wrong in group.inc itself: counts, offsets, peers, in-place rules.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
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

struct World {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  int joined = 0;
  // reusable barrier
  int waiting = 0;
  uint64_t generation = 0;
  // what every rank exposes for the operation in flight
  std::vector<const void*> ptr;                       // collectives: the rank's send buffer
  struct Msg { const void* p = nullptr; size_t bytes = 0; bool set = false; };
  std::vector<std::vector<Msg>> mail;                 // mail[src][dst]
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t g = generation;
    if (++waiting == n) {
      waiting = 0;
      generation++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != g; });
    }
  }
};

std::mutex g_mu;
std::map<std::string, World*> g_worlds;
uint64_t g_next_id = 1;

size_t dsize(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: return 4;
    default: return 8;
  }
}

struct Op { bool send; const void* sp; void* rp; size_t bytes; int peer; hipStream_t stream; };
thread_local int tl_depth = 0;
thread_local std::vector<Op> tl_ops;
thread_local ncclComm* tl_comm = nullptr;

}  // namespace

struct ncclComm {
  World* w;
  int rank;
};

namespace {

ncclResult_t run_p2p(ncclComm* c, std::vector<Op>& ops) {
  World* w = c->w;
  hipStream_t st = nullptr;
  for (auto& o : ops) st = o.stream;
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;  // what I send is complete
  {
    std::lock_guard<std::mutex> lk(w->mu);
    for (auto& o : ops)
      if (o.send) {
        World::Msg& m = w->mail[c->rank][o.peer];
        if (m.set) return ncclInvalidUsage;  // two sends to one peer in one group: group.inc never does that
        m = World::Msg{o.sp, o.bytes, true};
      }
  }
  w->barrier();  // every send of this round is posted
  ncclResult_t rc = ncclSuccess;
  for (auto& o : ops)
    if (!o.send) {
      World::Msg m;
      {
        std::lock_guard<std::mutex> lk(w->mu);
        m = w->mail[o.peer][c->rank];
      }
      if (!m.set || m.bytes != o.bytes) { rc = ncclInvalidArgument; continue; }  // a recv without its send / size mismatch
      if (hipMemcpyAsync(o.rp, m.p, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) rc = ncclUnhandledCudaError;
    }
  if (hipStreamSynchronize(st) != hipSuccess) rc = ncclUnhandledCudaError;
  w->barrier();  // every copy is done: senders may reuse their buffers
  {
    std::lock_guard<std::mutex> lk(w->mu);
    for (int d = 0; d < w->n; d++) w->mail[c->rank][d] = World::Msg{};  // my mailbox row is free for the next round
  }
  w->barrier();
  return rc;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::lock_guard<std::mutex> lk(g_mu);
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "fake-rccl-world-%llu", (unsigned long long)g_next_id++);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  World* w;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    World*& slot = g_worlds[std::string(id.internal, strnlen(id.internal, sizeof id.internal))];
    if (!slot) {
      slot = new World();
      slot->n = nranks;
      slot->ptr.assign(nranks, nullptr);
      slot->mail.assign(nranks, std::vector<World::Msg>(nranks));
    }
    w = slot;
    if (w->n != nranks) return ncclInvalidArgument;
  }
  *comm = new ncclComm{w, rank};
  w->barrier();  // collective: returns when every rank has joined
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  delete comm;  // worlds are leaked on purpose (a test process)
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "success";
    case ncclInvalidArgument: return "fake rccl: invalid argument (recv without matching send, or sizes differ)";
    case ncclInvalidUsage: return "fake rccl: invalid usage";
    case ncclUnhandledCudaError: return "fake rccl: HIP error";
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
  World* w = comm->w;
  const size_t bytes = sendcount * dsize(t);
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  {
    std::lock_guard<std::mutex> lk(w->mu);
    w->ptr[comm->rank] = sendbuff;
  }
  w->barrier();
  ncclResult_t rc = ncclSuccess;
  for (int r = 0; r < w->n; r++) {
    const void* src;
    {
      std::lock_guard<std::mutex> lk(w->mu);
      src = w->ptr[r];
    }
    void* dst = (uint8_t*)recvbuff + (size_t)r * bytes;
    if (dst == src) continue;  // in place
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) rc = ncclUnhandledCudaError;
  }
  if (hipStreamSynchronize(stream) != hipSuccess) rc = ncclUnhandledCudaError;
  w->barrier();
  return rc;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
  tl_comm = comm;
  if (t != ncclUint64 && t != ncclInt64) return ncclInvalidArgument;  // all the library reduces
  if (op != ncclSum && op != ncclMax) return ncclInvalidArgument;
  World* w = comm->w;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  {
    std::lock_guard<std::mutex> lk(w->mu);
    w->ptr[comm->rank] = sendbuff;
  }
  w->barrier();
  std::vector<uint64_t> acc(count, 0), tmp(count);
  ncclResult_t rc = ncclSuccess;
  for (int r = 0; r < w->n; r++) {
    const void* src;
    {
      std::lock_guard<std::mutex> lk(w->mu);
      src = w->ptr[r];
    }
    if (hipMemcpy(tmp.data(), src, count * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = ncclUnhandledCudaError;
    for (size_t k = 0; k < count; k++) acc[k] = op == ncclSum ? acc[k] + tmp[k] : (tmp[k] > acc[k] ? tmp[k] : acc[k]);
  }
  w->barrier();  // everybody has read everybody's input: in-place results may be written
  if (hipMemcpy(recvbuff, acc.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
  w->barrier();
  return rc;
}

}  // extern "C"
