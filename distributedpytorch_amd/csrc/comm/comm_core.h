// Native communicator: ONE C ABI (dpa_comm_*), two transports behind it.
//
//   front (comm_api.cpp)   sequence numbers, the slot ring, bucket -> slot bookkeeping, protocol checks, error text
//   Backend                what moves the bytes:
//     RcclBackend  (rccl_backend.cpp)  RCCL over xGMI on its own HIP stream, hipEvent fencing against the compute stream
//     TcpBackend   (tcp_backend.cpp)   a worker thread ("the comm stream") + a socket mesh running the two-shot
//                                      reduce-scatter / all-gather all-reduce; host memory, or device memory staged
//                                      through pinned buffers (several ranks may then share ONE GPU: the rehearsal
//                                      transport for boxes where RCCL cannot be given one device per rank)
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
  // In-place fp32 SUM all-reduce of `count` floats at `ptr`, ordered after everything queued on `compute` (a
  // hipStream_t for device memory, ignored for host memory).  Asynchronous: completion is observed through wait(slot).
  virtual int allreduce(int slot, float* ptr, long long count, void* compute) = 0;
  // Make `compute` (device) or the calling thread (host) wait for the collective issued on `slot`.
  virtual int wait(int slot, void* compute) = 0;
  // Broadcast `bytes` from `root`; on return the data is ordered before later work on `compute`.
  virtual int broadcast(void* ptr, long long bytes, int root, void* compute) = 0;
  virtual int async_error() { return 0; }
  virtual int sync() { return 0; }
  virtual void abort() {}
  virtual const char* name() const = 0;
  std::string error;          // last error of this backend (text)
};

struct Comm {
  Backend* be = nullptr;
  int world = 1, rank = 0;
  int ring = 256;                       // slots: one per in-flight collective
  unsigned long long seq = 0;
  std::vector<int> bucket_slot;         // bucket -> slot of its in-flight collective (-1: none)
  std::vector<char> slot_busy;          // slot launched and not yet waited for
  char last_error[512] = {0};
};

Backend* make_rccl_backend(const void* id_bytes, int world, int rank, int device, std::string* err);
Backend* make_tcp_backend(int world, int rank, int device, std::string* err);
int tcp_backend_port(Backend* b);
int tcp_backend_connect(Backend* b, const char* peers_csv, int timeout_ms);
void tcp_backend_set_timeout(Backend* b, int timeout_ms);

}  // namespace dpa
