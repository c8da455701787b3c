// Front of the native communicator: the C ABI the reducer core (reducer.cpp) and the Python layer call.  Everything
// that is NOT transport lives here and is therefore identical for RCCL and for the TCP transport the CPU tests run:
// collective sequence numbers, the slot ring (one slot = one in-flight collective = one ready/done event pair on the
// GPU), the bucket -> slot map, and the protocol errors (waiting for a bucket twice, re-using a slot that is still
// in flight because the ring wrapped).
#include "comm_core.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace dpa;

namespace {
int fail(Comm* c, int code, const char* what) {
  if (c) {
    const std::string& detail = c->be ? c->be->error : std::string();
    snprintf(c->last_error, sizeof(c->last_error), "%s failed (code %d)%s%s", what, code, detail.empty() ? "" : ": ", detail.c_str());
  }
  return code ? code : -1;
}
Comm* make_front(Backend* be, int world, int rank) {
  Comm* c = new Comm();
  c->be = be; c->world = world; c->rank = rank;
  const char* e = getenv("DPA_COMM_EVENT_RING");
  const int v = e ? atoi(e) : 0;
  c->ring = v > 0 ? v : 256;
  c->slot_busy.assign(c->ring, 0);
  return c;
}
thread_local std::string g_create_error;
}  // namespace

DPA_API const char* dpa_comm_create_error() { return g_create_error.c_str(); }

// ---- RCCL transport (rccl_backend.cpp) ---------------------------------------------------------------------------
DPA_API void* dpa_comm_init(const void* id_bytes, int world, int rank, int device) {
  std::string err;
  Backend* be = make_rccl_backend(id_bytes, world, rank, device, &err);
  if (!be) { g_create_error = err; return nullptr; }
  return make_front(be, world, rank);
}
// ---- TCP transport (tcp_backend.cpp): create -> publish port -> connect to every peer -----------------------------
// device < 0: buffers are host memory.  device >= 0: buffers are device memory staged through pinned host buffers.
DPA_API void* dpa_comm_init_tcp(int world, int rank, int device) {
  std::string err;
  Backend* be = make_tcp_backend(world, rank, device, &err);
  if (!be) { g_create_error = err; return nullptr; }
  return make_front(be, world, rank);
}
DPA_API int dpa_comm_tcp_port(void* h) { return tcp_backend_port(((Comm*)h)->be); }
// peers_csv: "addr:port,addr:port,..." indexed by rank (this rank's own entry is ignored)
DPA_API int dpa_comm_tcp_connect(void* h, const char* peers_csv, int timeout_ms) {
  Comm* c = (Comm*)h;
  const int rc = tcp_backend_connect(c->be, peers_csv, timeout_ms);
  return rc ? fail(c, rc, "tcp mesh connect") : 0;
}
DPA_API void dpa_comm_set_timeout_ms(void* h, int timeout_ms) {
  Comm* c = (Comm*)h;
  if (c && c->be && !strcmp(c->be->name(), "tcp")) tcp_backend_set_timeout(c->be, timeout_ms);
}

DPA_API const char* dpa_comm_backend_name(void* h) { return ((Comm*)h)->be->name(); }
DPA_API int dpa_comm_world(void* h) { return ((Comm*)h)->world; }
DPA_API int dpa_comm_rank(void* h) { return ((Comm*)h)->rank; }

// In-place fp32 SUM all-reduce of `count` floats at `ptr`, ordered after everything already queued on `compute`.
// Slots come from a ring (one fresh slot per collective) so that a GPU event is never re-recorded while an earlier
// wait on it may still be queued - iterations are enqueued back to back without host synchronisation.
DPA_API int dpa_comm_allreduce_async(void* h, float* ptr, long long count, int bucket, void* compute) {
  Comm* c = (Comm*)h;
  if (bucket < 0) return fail(c, 22, "all-reduce (negative bucket id)");
  if ((int)c->bucket_slot.size() <= bucket) c->bucket_slot.resize(bucket + 1, -1);
  if (c->bucket_slot[bucket] >= 0) return fail(c, 23, "all-reduce (bucket already has a collective in flight: missing wait)");
  const int idx = (int)(c->seq % (unsigned long long)c->ring);
  if (c->slot_busy[idx]) return fail(c, 24, "all-reduce (slot ring exhausted: more collectives in flight than DPA_COMM_EVENT_RING)");
  c->seq++;
  c->slot_busy[idx] = 1;
  c->bucket_slot[bucket] = idx;
  const int rc = c->be->allreduce(idx, ptr, count, compute);
  if (rc) { c->slot_busy[idx] = 0; c->bucket_slot[bucket] = -1; return fail(c, rc, "all-reduce"); }
  return 0;
}
// Make `compute` wait (on the device; the calling thread for host memory) for bucket's all-reduce.
DPA_API int dpa_comm_wait(void* h, int bucket, void* compute) {
  Comm* c = (Comm*)h;
  if (bucket < 0 || bucket >= (int)c->bucket_slot.size() || c->bucket_slot[bucket] < 0) return 0;   // nothing in flight
  const int idx = c->bucket_slot[bucket];
  c->bucket_slot[bucket] = -1;
  c->slot_busy[idx] = 0;
  const int rc = c->be->wait(idx, compute);
  return rc ? fail(c, rc, "wait") : 0;
}
DPA_API int dpa_comm_inflight(void* h) {
  Comm* c = (Comm*)h; int n = 0;
  for (char b : c->slot_busy) n += b;
  return n;
}
// Broadcast `bytes` from root, ordered after `compute`, and `compute` waits for it (wrap time, BatchNorm buffers).
DPA_API int dpa_comm_broadcast(void* h, void* ptr, long long bytes, int root, void* compute) {
  Comm* c = (Comm*)h;
  if (root < 0 || root >= c->world) return fail(c, 22, "broadcast (root out of range)");
  const int rc = c->be->broadcast(ptr, bytes, root, compute);
  return rc ? fail(c, rc, "broadcast") : 0;
}
// 0 = healthy; otherwise the transport's asynchronous error code (the trainer aborts and exits non-zero).
DPA_API int dpa_comm_async_error(void* h) { return ((Comm*)h)->be->async_error(); }
DPA_API const char* dpa_comm_last_error(void* h) { return ((Comm*)h)->last_error; }
DPA_API int dpa_comm_sync(void* h) { return ((Comm*)h)->be->sync(); }
DPA_API void dpa_comm_abort(void* h) { Comm* c = (Comm*)h; if (c && c->be) c->be->abort(); }
DPA_API void dpa_comm_destroy(void* h) {
  Comm* c = (Comm*)h;
  if (!c) return;
  delete c->be;
  delete c;
}
