// RCCL transport of the native communicator (N2 of SURVEY.md §2.4: the role ProcessGroupNCCL plays for the reference,
// /root/reference/classif.py:86): one RCCL communicator per process/GPU, a dedicated HIP stream for collectives,
// hipEvent fencing against the compute stream so a gradient bucket's all-reduce overlaps the rest of backward:
//
//   compute stream:  ... wgrad kernels of bucket k | record(ready[s]) | more backward kernels ... | wait(done[s]) | adam(k)
//   comm stream:                                    wait(ready[s]) | ncclAllReduce(bucket k) | record(done[s])
//
// s = the slot the front (comm_api.cpp) assigned to the collective; broadcasts take slots from the same ring.  All
// event pairs of the ring are created up front (prepare): the front's watchdog thread polls done[s] with hipEventQuery
// while the caller keeps issuing collectives, so nothing is ever reallocated.  RCCL itself has no notion of a timeout -
// a peer that never arrives leaves ncclAllReduce's kernel spinning - which is why the bound lives in that watchdog;
// abort() (ncclCommAbort) is what it calls to make the stuck kernel return.
// The ncclUniqueId travels through the C++ TCP store (csrc/host/tcpstore.cpp).  The comm stream has the SAME priority
// as the compute stream (as ProcessGroupNCCL's does): round 1 saw intermittent hangs of hipDeviceSynchronize with a
// high-priority comm stream and none since the priority is 0; round 2 could not reproduce the hang in either
// configuration (docs/COMM.md).  There is deliberately no knob for it.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>

#include "comm_core.h"

namespace dpa {
namespace {
struct RcclBackend : Backend {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int device = 0, world = 1;
  std::vector<hipEvent_t> ready, done;          // fixed after prepare()
  std::timed_mutex comm_mu;                     // the communicator handle: collectives enqueue under it, abort() takes it away
  std::atomic<bool> aborted{false};
  // test hook: a host function on the comm stream that blocks until released (a collective that never finishes)
  std::mutex stall_mu; std::condition_variable stall_cv; bool stall_armed = false, stall_released = false;

  int hip_fail(const char* what, hipError_t e) {
    char buf[160]; snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e)); error = buf; return 1;
  }
  int nccl_fail(const char* what, ncclResult_t r) {
    char buf[160]; snprintf(buf, sizeof(buf), "%s: %s", what, ncclGetErrorString(r)); error = buf; return 100 + (int)r;
  }
  int prepare(int ring) override {
    for (int i = 0; i < ring; ++i) {
      hipEvent_t a = nullptr, b = nullptr;
      hipError_t e = hipEventCreateWithFlags(&a, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&b, hipEventDisableTiming);
      if (e != hipSuccess) return hip_fail("hipEventCreateWithFlags", e);
      ready.push_back(a); done.push_back(b);
    }
    return 0;
  }
  static void stall_fn(void* self) {
    RcclBackend* b = (RcclBackend*)self;
    std::unique_lock<std::mutex> l(b->stall_mu);
    b->stall_cv.wait(l, [b] { return b->stall_released; });
  }
  void stall_next() override { std::lock_guard<std::mutex> l(stall_mu); stall_armed = true; }
  void release_stall() override {
    { std::lock_guard<std::mutex> l(stall_mu); stall_released = true; }
    stall_cv.notify_all();
  }
  // record(ready) on compute -> comm stream waits for it -> `body` enqueues the collective -> record(done)
  template <class F>
  int enqueue(int slot, void* compute, const char* what, F body) {
    if (slot < 0 || slot >= (int)done.size()) { error = "slot outside the prepared ring"; return 2; }
    std::unique_lock<std::timed_mutex> lk(comm_mu);
    if (aborted.load() || !comm) { error = "the communicator has been aborted"; return 125; }
    hipError_t e = hipEventRecord(ready[slot], (hipStream_t)compute);
    if (e != hipSuccess) return hip_fail("hipEventRecord(ready)", e);
    e = hipStreamWaitEvent(stream, ready[slot], 0);
    if (e != hipSuccess) return hip_fail("hipStreamWaitEvent(comm, ready)", e);
    bool stall = false;
    { std::lock_guard<std::mutex> l(stall_mu); stall = stall_armed && !stall_released; stall_armed = false; }
    if (stall) {
      e = hipLaunchHostFunc(stream, &RcclBackend::stall_fn, this);
      if (e != hipSuccess) return hip_fail("hipLaunchHostFunc(stall)", e);
    }
    ncclResult_t r = body();
    if (r != ncclSuccess) return nccl_fail(what, r);
    e = hipEventRecord(done[slot], stream);
    if (e != hipSuccess) return hip_fail("hipEventRecord(done)", e);
    // push the comm stream's queued markers to the hardware queue now (a query flushes; it never blocks): without a
    // flush the done-marker can sit in the host-side batch while the compute stream already waits on it
    // (not while `compute` is being captured into a hipGraph: the comm stream has joined the capture through the ready
    // event, and a query on a capturing stream is an error that invalidates the capture)
    if (trackable(compute)) (void)hipStreamQuery(stream);
    return 0;
  }
  int allreduce(int slot, float* ptr, long long count, void* compute) override {
    return enqueue(slot, compute, "ncclAllReduce",
                   [&] { return ncclAllReduce(ptr, ptr, (size_t)count, ncclFloat32, ncclSum, comm, stream); });
  }
  int broadcast(int slot, void* ptr, long long bytes, int root, void* compute) override {
    return enqueue(slot, compute, "ncclBroadcast",
                   [&] { return ncclBroadcast(ptr, ptr, (size_t)bytes, ncclChar, root, comm, stream); });
  }
  bool trackable(void* compute) override {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing((hipStream_t)compute, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
  }
  int wait(int slot, void* compute) override {
    if (slot < 0 || slot >= (int)done.size()) { error = "wait on a slot outside the prepared ring"; return 2; }
    hipError_t e = hipStreamWaitEvent((hipStream_t)compute, done[slot], 0);
    return e == hipSuccess ? 0 : hip_fail("hipStreamWaitEvent(compute, done)", e);
  }
  int query(int slot) override {
    if (slot < 0 || slot >= (int)done.size()) return -1;
    const hipError_t e = hipEventQuery(done[slot]);
    if (e == hipSuccess) return 0;
    if (e == hipErrorNotReady) return 1;
    (void)hipGetLastError();
    return -1;
  }
  int peer_count() override {
    std::unique_lock<std::timed_mutex> lk(comm_mu);
    int n = -1;
    if (!comm || ncclCommCount(comm, &n) != ncclSuccess) return -1;
    return n;
  }
  int async_error() override {
    if (aborted.load()) return 125;
    std::unique_lock<std::timed_mutex> lk(comm_mu);
    if (!comm) return -1;
    ncclResult_t st = ncclSuccess;
    if (ncclCommGetAsyncError(comm, &st) != ncclSuccess) return -1;
    return (int)st;
  }
  int sync() override { return hipStreamSynchronize(stream) == hipSuccess ? 0 : 1; }
  // Called from the watchdog thread (or the caller's error path).  The caller may be inside an enqueue: give it two
  // seconds to come out, then abort regardless - a thread blocked inside RCCL is exactly what abort is for.
  void abort() override {
    if (aborted.exchange(true)) return;
    std::unique_lock<std::timed_mutex> lk(comm_mu, std::defer_lock);
    const bool got = lk.try_lock_for(std::chrono::seconds(2));
    ncclComm_t c = comm;
    comm = nullptr;
    if (got) lk.unlock();
    if (c) ncclCommAbort(c);
  }
  const char* name() const override { return "rccl"; }
  ~RcclBackend() override {
    release_stall();
    if (stream && !aborted.load()) (void)hipStreamSynchronize(stream);
    if (comm) ncclCommDestroy(comm);
    for (auto e : ready) (void)hipEventDestroy(e);
    for (auto e : done) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
  }
};
}  // namespace

Backend* make_rccl_backend(const void* id_bytes, int world, int rank, int device, std::string* err) {
  RcclBackend* b = new RcclBackend();
  b->device = device; b->world = world;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&b->stream, hipStreamNonBlocking, 0);     // priority 0: see the header comment
  if (e != hipSuccess) { *err = std::string("HIP setup: ") + hipGetErrorString(e); delete b; return nullptr; }
  ncclUniqueId id; memcpy(&id, id_bytes, sizeof(id));
  ncclResult_t r = ncclCommInitRank(&b->comm, world, id, rank);
  if (r != ncclSuccess) { *err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); b->comm = nullptr; delete b; return nullptr; }
  return b;
}
}  // namespace dpa

DPA_API int dpa_comm_unique_id_size() { return (int)sizeof(ncclUniqueId); }
DPA_API int dpa_comm_unique_id(void* out) {
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return (int)r;
  memcpy(out, &id, sizeof(id));
  return 0;
}
