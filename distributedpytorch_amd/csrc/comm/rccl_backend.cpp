// RCCL transport of the native communicator (N2 of SURVEY.md §2.4: the role ProcessGroupNCCL plays for the reference,
// /root/reference/classif.py:86): one RCCL communicator per process/GPU, a dedicated HIP stream for collectives,
// hipEvent fencing against the compute stream so a gradient bucket's all-reduce overlaps the rest of backward:
//
//   compute stream:  ... wgrad kernels of bucket k | record(ready[s]) | more backward kernels ... | wait(done[s]) | adam(k)
//   comm stream:                                    wait(ready[s]) | ncclAllReduce(bucket k) | record(done[s])
//
// s = the slot the front (comm_api.cpp) assigned to the collective.  The ncclUniqueId travels through the C++ TCP
// store (csrc/host/tcpstore.cpp).  The comm stream has the SAME priority as the compute stream (as ProcessGroupNCCL's
// does): round 1 saw intermittent hangs of hipDeviceSynchronize with a high-priority comm stream and none since the
// priority is 0; round 2 could not reproduce the hang in either configuration (docs/COMM.md).  There is deliberately no
// knob for it.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>

#include "comm_core.h"

namespace dpa {
namespace {
struct RcclBackend : Backend {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int device = 0;
  std::vector<hipEvent_t> ready, done;
  hipEvent_t bc_ready = nullptr, bc_done = nullptr;

  int hip_fail(const char* what, hipError_t e) {
    char buf[160]; snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e)); error = buf; return 1;
  }
  int nccl_fail(const char* what, ncclResult_t r) {
    char buf[160]; snprintf(buf, sizeof(buf), "%s: %s", what, ncclGetErrorString(r)); error = buf; return 100 + (int)r;
  }
  int event_pair(int slot, hipEvent_t* r, hipEvent_t* d) {
    while ((int)ready.size() <= slot) {
      hipEvent_t a = nullptr, b = nullptr;
      hipError_t e = hipEventCreateWithFlags(&a, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&b, hipEventDisableTiming);
      if (e != hipSuccess) return hip_fail("hipEventCreateWithFlags", e);
      ready.push_back(a); done.push_back(b);
    }
    *r = ready[slot]; *d = done[slot];
    return 0;
  }
  int allreduce(int slot, float* ptr, long long count, void* compute) override {
    hipEvent_t rdy, dn;
    if (int rc = event_pair(slot, &rdy, &dn)) return rc;
    hipError_t e = hipEventRecord(rdy, (hipStream_t)compute);
    if (e != hipSuccess) return hip_fail("hipEventRecord(ready)", e);
    e = hipStreamWaitEvent(stream, rdy, 0);
    if (e != hipSuccess) return hip_fail("hipStreamWaitEvent(comm, ready)", e);
    ncclResult_t r = ncclAllReduce(ptr, ptr, (size_t)count, ncclFloat32, ncclSum, comm, stream);
    if (r != ncclSuccess) return nccl_fail("ncclAllReduce", r);
    e = hipEventRecord(dn, stream);
    if (e != hipSuccess) return hip_fail("hipEventRecord(done)", e);
    // push the comm stream's queued markers to the hardware queue now (a query flushes; it never blocks): without a
    // flush the done-marker can sit in the host-side batch while the compute stream already waits on it
    flush((hipStream_t)compute);
    return 0;
  }
  // (not while `compute` is being captured into a hipGraph: the comm stream has joined the capture through the ready
  // event, and a query on a capturing stream is an error that invalidates the capture)
  void flush(hipStream_t compute) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(compute, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) (void)hipStreamQuery(stream);
  }
  int wait(int slot, void* compute) override {
    if (slot >= (int)done.size()) { error = "wait on a slot that was never launched"; return 2; }
    hipError_t e = hipStreamWaitEvent((hipStream_t)compute, done[slot], 0);
    return e == hipSuccess ? 0 : hip_fail("hipStreamWaitEvent(compute, done)", e);
  }
  int broadcast(void* ptr, long long bytes, int root, void* compute) override {
    hipError_t e = hipEventRecord(bc_ready, (hipStream_t)compute);
    if (e != hipSuccess) return hip_fail("hipEventRecord(bc_ready)", e);
    e = hipStreamWaitEvent(stream, bc_ready, 0);
    if (e != hipSuccess) return hip_fail("hipStreamWaitEvent(comm, bc_ready)", e);
    ncclResult_t r = ncclBroadcast(ptr, ptr, (size_t)bytes, ncclChar, root, comm, stream);
    if (r != ncclSuccess) return nccl_fail("ncclBroadcast", r);
    e = hipEventRecord(bc_done, stream);
    if (e != hipSuccess) return hip_fail("hipEventRecord(bc_done)", e);
    flush((hipStream_t)compute);
    e = hipStreamWaitEvent((hipStream_t)compute, bc_done, 0);
    return e == hipSuccess ? 0 : hip_fail("hipStreamWaitEvent(compute, bc_done)", e);
  }
  int async_error() override {
    if (!comm) return -1;
    ncclResult_t st = ncclSuccess;
    if (ncclCommGetAsyncError(comm, &st) != ncclSuccess) return -1;
    return (int)st;
  }
  int sync() override { return hipStreamSynchronize(stream) == hipSuccess ? 0 : 1; }
  void abort() override { if (comm) { ncclCommAbort(comm); comm = nullptr; } }
  const char* name() const override { return "rccl"; }
  ~RcclBackend() override {
    if (stream) (void)hipStreamSynchronize(stream);
    if (comm) ncclCommDestroy(comm);
    for (auto e : ready) (void)hipEventDestroy(e);
    for (auto e : done) (void)hipEventDestroy(e);
    if (bc_ready) (void)hipEventDestroy(bc_ready);
    if (bc_done) (void)hipEventDestroy(bc_done);
    if (stream) (void)hipStreamDestroy(stream);
  }
};
}  // namespace

Backend* make_rccl_backend(const void* id_bytes, int world, int rank, int device, std::string* err) {
  RcclBackend* b = new RcclBackend();
  b->device = device;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&b->stream, hipStreamNonBlocking, 0);     // priority 0: see the header comment
  if (e == hipSuccess) e = hipEventCreateWithFlags(&b->bc_ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&b->bc_done, hipEventDisableTiming);
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
