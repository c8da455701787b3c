// Native communicator: ONE C ABI (dpa_comm_*), three transports behind it.
//
//   front (comm_api.cpp)   sequence numbers, the slot ring, bucket -> slot bookkeeping, protocol checks, error text, and
//                          the WATCHDOG: a host thread that follows every in-flight collective, bounds its run time,
//                          publishes this rank's progress through the rendezvous store and, on expiry, says which
//                          collective is stuck and which rank is behind, aborts the transport and ends the process
//   Backend                what moves the bytes:
//     RcclBackend  (rccl_backend.cpp)  RCCL over xGMI on its own HIP stream, hipEvent fencing against the compute stream
//     MeshBackend  (mesh_backend.hip)  the two-shot mesh all-reduce as OWN HIP kernels over hipIpc peer buffers (one node)
//     TcpBackend   (tcp_backend.cpp)   a socket mesh running the same two-shot algorithm through host memory; device
//                                      buffers are staged through pinned memory (several ranks may share ONE GPU)
//   WorkerBackend (worker_backend.h)   what Mesh and Tcp share: one worker thread = "the comm stream" of the transport
//
// This is the role ProcessGroupNCCL / ProcessGroupGloo play for the reference (/root/reference/classif.py:86-87,138):
// the reducer core (reducer.cpp) and the Python layer only see the front.
#pragma once
#include <cstddef>
#include <string>
#include <vector>

#define DPA_API extern "C" __attribute__((visibility("default")))

namespace dpa {

struct Backend {
  virtual ~Backend() {}
  // Create whatever per-slot state the transport needs for a ring of `ring` slots, once, before the first collective
  // (the watchdog thread reads per-slot state while the caller issues collectives: nothing may be reallocated later).
  virtual int prepare(int ring) { (void)ring; return 0; }
  // In-place fp32 SUM all-reduce of `count` floats at `ptr`, ordered after everything queued on `compute` (a
  // hipStream_t for device memory, ignored for host memory).  Asynchronous: completion is observed through wait(slot).
  virtual int allreduce(int slot, float* ptr, long long count, void* compute) = 0;
  // Make `compute` (device) or the calling thread (host) wait for the collective issued on `slot`.
  virtual int wait(int slot, void* compute) = 0;
  // Broadcast `bytes` from `root` on `slot`, ordered after `compute`.  Asynchronous like allreduce: wait(slot) orders
  // later work on `compute` behind it.
  virtual int broadcast(int slot, void* ptr, long long bytes, int root, void* compute) = 0;
  // Progress of the collective last issued on `slot`: 0 finished, 1 still running, < 0 cannot be told.  Called from the
  // watchdog thread.
  virtual int query(int slot) { (void)slot; return -1; }
  // false: a collective issued now on `compute` cannot be followed by query() (it is being captured into a hipGraph).
  virtual bool trackable(void* compute) { (void)compute; return true; }
  // Number of ranks the TRANSPORT itself believes it connects (ncclCommCount for RCCL); -1 unknown.
  virtual int peer_count() { return -1; }
  virtual int async_error() { return 0; }
  virtual int sync() { return 0; }
  virtual void abort() {}
  // Test hook (DPA_TEST_STALL_COLLECTIVE): make the NEXT collective hang inside the transport until release_stall().
  virtual void stall_next() {}
  virtual void release_stall() {}
  virtual const char* name() const = 0;
  std::string error;          // last error of this backend (text)
};

struct Comm;
struct Watchdog;

Backend* make_rccl_backend(const void* id_bytes, int world, int rank, int device, std::string* err);
Backend* make_tcp_backend(int world, int rank, int device, std::string* err);
int tcp_backend_port(Backend* b);
int tcp_backend_connect(Backend* b, const char* peers_csv, int timeout_ms);
void tcp_backend_set_timeout(Backend* b, int timeout_ms);
Backend* make_mesh_backend(int world, int rank, int device, long long capacity_bytes, std::string* err);
int mesh_backend_handle_size();
int mesh_backend_export(Backend* b, void* out);                       // this rank's IPC handles + shared-memory name
int mesh_backend_connect(Backend* b, const void* all_handles);       // every rank's export blob, indexed by rank
void mesh_backend_set_timeout(Backend* b, int timeout_ms);
void mesh_backend_unlink(Backend* b);

}  // namespace dpa
