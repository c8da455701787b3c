// Native RCCL communicator (N2 of SURVEY.md §2.4: the role ProcessGroupNCCL plays for the reference, classif.py:86):
// one RCCL communicator per process/GPU, a dedicated HIP stream for collectives, hipEvent fencing against
// the compute stream so a gradient bucket's all-reduce overlaps the rest of backward:
//
//   compute stream:  ... wgrad kernels of bucket k | record(ready[k]) | more backward kernels ...   | wait(done[k]) | adam(k)
//   comm stream:                                    wait(ready[k]) | ncclAllReduce(bucket k) | record(done[k])
//
// The ncclUniqueId travels through the C++ TCP store (tcpstore.cpp).  Opt-in backend (DPA_COMM_BACKEND=rccl_native);
// the default path uses torch.distributed's RCCL process group, which issues the same collectives.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define DPA_API extern "C" __attribute__((visibility("default")))

namespace {
struct Comm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int world = 1, rank = 0, device = 0;
  std::vector<hipEvent_t> ready, done;
  std::vector<int> bucket_slot;          // bucket -> ring index of its in-flight collective (-1: none)
  unsigned long long seq = 0;
  hipEvent_t bc_ready = nullptr, bc_done = nullptr;
  char last_error[256] = {0};
};
int fail(Comm* c, const char* what, int code) {
  if (c) snprintf(c->last_error, sizeof(c->last_error), "%s failed with code %d", what, code);
  return code ? code : -1;
}
hipEvent_t& slot(std::vector<hipEvent_t>& v, int i) {
  while ((int)v.size() <= i) {
    hipEvent_t e; hipEventCreateWithFlags(&e, hipEventDisableTiming); v.push_back(e);
  }
  return v[i];
}
}  // namespace

DPA_API int dpa_comm_unique_id_size() { return (int)sizeof(ncclUniqueId); }
DPA_API int dpa_comm_unique_id(void* out) {
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return (int)r;
  memcpy(out, &id, sizeof(id));
  return 0;
}
DPA_API void* dpa_comm_init(const void* id_bytes, int world, int rank, int device) {
  Comm* c = new Comm();
  c->world = world; c->rank = rank; c->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete c; return nullptr; }
  // Stream priority: NORMAL (0) by default, like the stream torch's ProcessGroupNCCL uses.  Measured on MI355X
  // (ROCm 7.0 runtime, GPU_MAX_HW_QUEUES=4): with a high (-1) or low (+1) priority comm queue, a barrier packet that
  // waits on the compute queue intermittently deadlocked against the compute queue's own wait on the comm queue
  // (hipDeviceSynchronize never returned); equal-priority queues do not.  DPA_COMM_PRIORITY=high|low overrides.
  int least = 0, greatest = 0;
  hipDeviceGetStreamPriorityRange(&least, &greatest);
  const char* sf = getenv("DPA_COMM_STREAM");
  const char* sp = getenv("DPA_COMM_PRIORITY");
  unsigned flags = (sf && sf[0] == 'b') ? hipStreamDefault : hipStreamNonBlocking;
  int prio = 0;
  if (sp && sp[0] == 'h') prio = greatest;
  if (sp && sp[0] == 'l') prio = least;
  if (hipStreamCreateWithPriority(&c->stream, flags, prio) != hipSuccess) { delete c; return nullptr; }
  hipEventCreateWithFlags(&c->bc_ready, hipEventDisableTiming);
  hipEventCreateWithFlags(&c->bc_done, hipEventDisableTiming);
  ncclUniqueId id; memcpy(&id, id_bytes, sizeof(id));
  if (ncclCommInitRank(&c->comm, world, id, rank) != ncclSuccess) { hipStreamDestroy(c->stream); delete c; return nullptr; }
  return c;
}
// In-place fp32 SUM all-reduce of `count` floats at `ptr`, ordered after everything already queued on `compute`.
// Events come from a ring (one fresh pair per collective) so an event is never re-recorded while an earlier
// wait on it may still be queued - iterations are enqueued back to back without host synchronisation.
DPA_API int dpa_comm_allreduce_async(void* h, float* ptr, long long count, int bucket, void* compute) {
  Comm* c = (Comm*)h;
  static const int ring = [] { const char* e = getenv("DPA_COMM_EVENT_RING"); int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
  int idx = (int)(c->seq++ % ring);
  if ((int)c->bucket_slot.size() <= bucket) c->bucket_slot.resize(bucket + 1, -1);
  c->bucket_slot[bucket] = idx;
  hipEvent_t& rdy = slot(c->ready, idx);
  hipEvent_t& dn = slot(c->done, idx);
  if (hipEventRecord(rdy, (hipStream_t)compute) != hipSuccess) return fail(c, "hipEventRecord", 1);
  if (hipStreamWaitEvent(c->stream, rdy, 0) != hipSuccess) return fail(c, "hipStreamWaitEvent", 1);
  ncclResult_t r = ncclAllReduce(ptr, ptr, (size_t)count, ncclFloat32, ncclSum, c->comm, c->stream);
  if (r != ncclSuccess) return fail(c, "ncclAllReduce", (int)r);
  if (hipEventRecord(dn, c->stream) != hipSuccess) return fail(c, "hipEventRecord", 1);
  // push the comm stream's queued markers to the hardware queue now (a query flushes; it never blocks): without a
  // flush the done-marker can sit in the host-side batch while the compute stream already waits on it
  (void)hipStreamQuery(c->stream);
  return 0;
}
// Make `compute` wait (on the device, not the host) for bucket's all-reduce.
DPA_API int dpa_comm_wait(void* h, int bucket, void* compute) {
  Comm* c = (Comm*)h;
  if (bucket >= (int)c->bucket_slot.size() || c->bucket_slot[bucket] < 0) return 0;
  hipEvent_t dn = c->done[c->bucket_slot[bucket]];
  c->bucket_slot[bucket] = -1;
  return hipStreamWaitEvent((hipStream_t)compute, dn, 0) == hipSuccess ? 0 : fail(c, "hipStreamWaitEvent", 1);
}
// Broadcast `bytes` from root, ordered after `compute`, and `compute` waits for it (used at wrap time and for BN buffers).
DPA_API int dpa_comm_broadcast(void* h, void* ptr, long long bytes, int root, void* compute) {
  Comm* c = (Comm*)h;
  if (hipEventRecord(c->bc_ready, (hipStream_t)compute) != hipSuccess) return fail(c, "hipEventRecord", 1);
  hipStreamWaitEvent(c->stream, c->bc_ready, 0);
  ncclResult_t r = ncclBroadcast(ptr, ptr, (size_t)bytes, ncclChar, root, c->comm, c->stream);
  if (r != ncclSuccess) return fail(c, "ncclBroadcast", (int)r);
  hipEventRecord(c->bc_done, c->stream);
  (void)hipStreamQuery(c->stream);
  hipStreamWaitEvent((hipStream_t)compute, c->bc_done, 0);
  return 0;
}
// 0 = healthy; otherwise the RCCL async error code (the trainer aborts and exits non-zero so the launcher tears down).
DPA_API int dpa_comm_async_error(void* h) {
  Comm* c = (Comm*)h;
  ncclResult_t st = ncclSuccess;
  if (ncclCommGetAsyncError(c->comm, &st) != ncclSuccess) return -1;
  return (int)st;
}
DPA_API const char* dpa_comm_last_error(void* h) { return ((Comm*)h)->last_error; }
DPA_API int dpa_comm_sync(void* h) { return hipStreamSynchronize(((Comm*)h)->stream) == hipSuccess ? 0 : 1; }
DPA_API void dpa_comm_abort(void* h) { Comm* c = (Comm*)h; if (c && c->comm) { ncclCommAbort(c->comm); c->comm = nullptr; } }
DPA_API void dpa_comm_destroy(void* h) {
  Comm* c = (Comm*)h;
  if (!c) return;
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->comm) ncclCommDestroy(c->comm);
  for (auto e : c->ready) hipEventDestroy(e);
  for (auto e : c->done) hipEventDestroy(e);
  if (c->bc_ready) hipEventDestroy(c->bc_ready);
  if (c->bc_done) hipEventDestroy(c->bc_done);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}
