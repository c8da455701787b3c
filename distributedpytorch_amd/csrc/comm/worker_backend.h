// What the socket transport (tcp_backend.cpp) and the IPC mesh transport (mesh_backend.hip) share: ONE worker thread
// per communicator that plays the role of RCCL's comm stream.  Same asynchronous contract as the RCCL transport:
//
//   allreduce / broadcast (slot, ...)  record the ready event on the compute stream (device memory), queue the job,
//                                      return at once; jobs run in issue order, overlapped with whatever the caller
//                                      does next
//   wait(slot, ...)                    blocks the caller until the worker has finished (host memory) or has QUEUED the
//                                      job's last device work and recorded the done event, which `compute` then waits for
//   query(slot)                        the job's pending flag - what the front's watchdog polls
//
// A transport implements execute(job): everything between "the producers are done" and "the result is in place".
#pragma once
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "comm_core.h"

namespace dpa {

struct Job {
  int kind = 0;            // 1 all-reduce, 2 broadcast
  int slot = -1;
  void* ptr = nullptr; long long n = 0; int root = 0;      // n: floats (all-reduce) / bytes (broadcast)
  hipEvent_t ready = nullptr, done = nullptr;
};

struct WorkerBackend : Backend {
  int world = 1, rank = 0, device = -1;
  int timeout_ms = 600000;
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::deque<Job> jobs;
  struct Slot { bool pending = false; int rc = 0; };
  std::vector<Slot> slots;               // indexed by the front's ring slot; fixed after prepare()
  bool stop = false;
  std::atomic<int> async_err{0};
  std::string worker_error;              // guarded by mu
  hipStream_t stream = nullptr;          // the transport's own stream (device memory only)
  std::vector<hipEvent_t> ev_ready, ev_done;
  bool stall_armed = false, stall_released = false;      // guarded by mu (test hook)

  virtual int execute(const Job& j, std::string* err) = 0;

  int setup_device(std::string* err) {
    if (device < 0) return 0;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, 0);
    if (e != hipSuccess) { *err = std::string("HIP setup: ") + hipGetErrorString(e); return 1; }
    return 0;
  }
  void start_worker() { worker = std::thread([this] { loop(); }); }

  int prepare(int ring) override {
    std::lock_guard<std::mutex> l(mu);
    slots.assign(ring, Slot());
    if (device >= 0) {
      for (int i = 0; i < ring; ++i) {
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) { error = "hipEventCreateWithFlags failed"; return 1; }
        ev_ready.push_back(a); ev_done.push_back(b);
      }
    }
    return 0;
  }

  void loop() {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> l(mu);
        cv_job.wait(l, [&] { return stop || !jobs.empty(); });
        if (jobs.empty()) return;          // stop requested and nothing queued
        j = jobs.front(); jobs.pop_front();
        if (stall_armed) {                 // test hook: this collective hangs inside the transport
          stall_armed = false;
          cv_job.wait(l, [&] { return stall_released || stop; });
        }
      }
      std::string err;
      int rc = 0;
      { std::lock_guard<std::mutex> l(mu); if (stop) { rc = 125; err = "communicator aborted"; } }
      if (!rc && async_err.load()) { rc = async_err.load(); err = "an earlier collective failed"; }
      if (!rc) rc = execute(j, &err);
      {
        std::lock_guard<std::mutex> l(mu);
        if (rc) { if (!async_err.load()) { async_err.store(rc); worker_error = err; } }
        slots[j.slot].pending = false; slots[j.slot].rc = rc;
      }
      cv_done.notify_all();
    }
  }

  int enqueue(Job j, void* compute) {
    if (j.slot < 0 || j.slot >= (int)slots.size()) { error = "slot outside the prepared ring"; return 2; }
    if (device >= 0) {
      j.ready = ev_ready[j.slot]; j.done = ev_done[j.slot];
      if (hipEventRecord(j.ready, (hipStream_t)compute) != hipSuccess) { error = "hipEventRecord(ready) failed"; return 1; }
    }
    std::lock_guard<std::mutex> l(mu);
    if (stop) { error = "the communicator has been aborted"; return 125; }
    Slot& s = slots[j.slot];
    if (s.pending) { error = "slot re-used while its collective is still running"; return 24; }
    s.pending = true; s.rc = 0;
    jobs.push_back(j);
    cv_job.notify_one();
    return 0;
  }
  int finish(int slot) {      // block until the worker is done with `slot`; returns its status
    std::unique_lock<std::mutex> l(mu);
    if (slot < 0 || slot >= (int)slots.size()) { error = "wait on a slot outside the prepared ring"; return 2; }
    Slot& s = slots[slot];
    cv_done.wait(l, [&] { return !s.pending || stop; });
    if (s.pending) { error = worker_error.empty() ? "communicator aborted" : worker_error; return 125; }
    if (s.rc) error = worker_error;
    return s.rc;
  }

  int allreduce(int slot, float* ptr, long long count, void* compute) override {
    Job j; j.kind = 1; j.slot = slot; j.ptr = ptr; j.n = count;
    return enqueue(j, compute);
  }
  int broadcast(int slot, void* ptr, long long bytes, int root, void* compute) override {
    Job j; j.kind = 2; j.slot = slot; j.ptr = ptr; j.n = bytes; j.root = root;
    return enqueue(j, compute);
  }
  int wait(int slot, void* compute) override {
    const int rc = finish(slot);
    if (rc) return rc;
    if (device >= 0) {
      // `done` was recorded by the worker before finish() returned, so the stream wait below is a real wait
      if (hipStreamWaitEvent((hipStream_t)compute, ev_done[slot], 0) != hipSuccess) { error = "hipStreamWaitEvent(compute, done) failed"; return 1; }
    }
    return 0;
  }
  int query(int slot) override {
    std::lock_guard<std::mutex> l(mu);
    if (slot < 0 || slot >= (int)slots.size()) return -1;
    return slots[slot].pending ? 1 : 0;
  }
  int async_error() override { return async_err.load(); }
  int sync() override {
    std::unique_lock<std::mutex> l(mu);
    cv_done.wait(l, [&] {
      if (stop) return true;
      if (!jobs.empty()) return false;
      for (auto& s : slots) if (s.pending) return false;
      return true;
    });
    return async_err.load();
  }
  void stall_next() override { std::lock_guard<std::mutex> l(mu); stall_armed = true; }
  void release_stall() override {
    { std::lock_guard<std::mutex> l(mu); stall_released = true; }
    cv_job.notify_all();
  }
  // wake everything that waits; derived classes additionally break their own blocking calls (sockets, spins)
  void abort() override {
    { std::lock_guard<std::mutex> l(mu); stop = true; }
    cv_job.notify_all();
    cv_done.notify_all();
  }
  // derived destructors call this FIRST (the worker uses their members)
  void join_worker() {
    { std::lock_guard<std::mutex> l(mu); stop = true; }
    cv_job.notify_all();
    cv_done.notify_all();
    if (worker.joinable()) worker.join();
  }
  void destroy_device() {
    if (device < 0) return;
    for (auto e : ev_ready) (void)hipEventDestroy(e);
    for (auto e : ev_done) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

}  // namespace dpa
