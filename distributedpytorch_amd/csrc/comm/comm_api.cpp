// Front of the native communicator: the C ABI the reducer core (reducer.cpp) and the Python layer call.  Everything
// that is NOT transport lives here and is therefore identical for RCCL, the IPC mesh and the TCP transport the CPU
// tests run: collective sequence numbers, the slot ring (one slot = one in-flight collective = one ready/done event
// pair on the GPU), the bucket -> slot map, the protocol errors (waiting for a bucket twice, re-using a slot that is
// still in flight because the ring wrapped) and the watchdog.
//
// Watchdog (the reference simply blocks, /root/reference/README.md:47-49, classif.py:86-87): every
// collective is entered into a queue with its sequence number, bucket, byte count and issue time.  One host thread
// asks the transport about the OLDEST entry (RCCL: hipEventQuery of its done event; worker transports: the job's
// pending flag), retires finished entries in order and publishes (issued, finished, time) for this rank under
// `hb/<rank>` in the rendezvous store every few seconds.  When the oldest entry has been running longer than the
// collective timeout the thread reads every rank's record, composes an error that names the collective AND the ranks
// that are behind or silent, prints it, aborts the transport (ncclCommAbort: a stuck device kernel returns, the
// compute stream drains, whatever host call was blocked comes back), and every later call of this ABI fails with that
// text.  If the process has not left by itself after a grace period the thread ends it with a non-zero status - a
// fresh process is the only safe continuation for a process that used the GPU.
#include "comm_core.h"

#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <mutex>
#include <thread>

using namespace dpa;

namespace dpa {

using Clock = std::chrono::steady_clock;
typedef int (*StoreSetFn)(void*, const char*, const void*, int);
typedef int (*StoreGetFn)(void*, const char*, void*, int, long long);

struct Tracked {
  unsigned long long seq;      // 1-based number of the collective on this communicator (the same on every rank)
  int slot, bucket, kind;      // kind: 1 all-reduce, 2 broadcast
  long long bytes;
  Clock::time_point t0;
};

struct Watchdog {
  Comm* comm = nullptr;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  bool stop = false;
  std::deque<Tracked> q;
  int timeout_ms = 600000, grace_ms = 10000, hb_ms = 2000;
  bool exit_on_trip = true;
  std::atomic<int> tripped{0};
  std::string text;                      // guarded by mu
  std::atomic<unsigned long long> issued{0}, finished{0};
  void* store = nullptr; StoreSetFn set_fn = nullptr; StoreGetFn get_fn = nullptr;
  void loop();
  void publish();
  void trip(const Tracked& t, long long age_ms);
};

struct Comm {
  Backend* be = nullptr;
  int world = 1, rank = 0;
  int ring = 256;                       // slots: one per in-flight collective
  unsigned long long seq = 0;
  std::vector<int> bucket_slot;         // bucket -> slot of its in-flight collective (-1: none)
  std::vector<char> slot_busy;          // slot launched and not yet waited for
  char last_error[1024] = {0};
  Watchdog* wd = nullptr;
  long long stall_at = -1;              // DPA_TEST_STALL_COLLECTIVE
};

static double unix_now() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

void Watchdog::publish() {
  if (!store || !set_fn) return;
  char key[48], val[96];
  snprintf(key, sizeof(key), "hb/%d", comm->rank);
  const int n = snprintf(val, sizeof(val), "%llu %llu %.3f", issued.load(), finished.load(), unix_now());
  (void)set_fn(store, key, val, n);
}

void Watchdog::trip(const Tracked& t, long long age_ms) {
  char head[400];
  snprintf(head, sizeof(head), "collective #%llu (%s, bucket %d, %lld bytes) on rank %d of %d has not finished after %.1f s "
           "(timeout %.1f s, transport %s; this rank has issued #%llu, finished #%llu)", t.seq, t.kind == 1 ? "all-reduce" : "broadcast",
           t.bucket, t.bytes, comm->rank, comm->world, age_ms / 1000.0, timeout_ms / 1000.0, comm->be->name(), issued.load(),
           finished.load());
  std::string msg = head;
  if (store && get_fn && comm->world > 1) {
    publish();
    std::string behind, silent, detail;
    const double now = unix_now();
    for (int r = 0; r < comm->world; ++r) {
      if (r == comm->rank) continue;
      char key[48], val[128] = {0};
      snprintf(key, sizeof(key), "hb/%d", r);
      const int n = get_fn(store, key, val, (int)sizeof(val) - 1, 500);
      unsigned long long iss = 0, fin = 0; double ts = 0;
      if (n <= 0 || sscanf(val, "%llu %llu %lf", &iss, &fin, &ts) != 3) {
        silent += (silent.empty() ? "" : ",") + std::to_string(r);
        detail += "; rank " + std::to_string(r) + ": no progress record";
        continue;
      }
      char d[160];
      const double age = now - ts;
      if (age > 3.0 * hb_ms / 1000.0 + 2.0) {
        silent += (silent.empty() ? "" : ",") + std::to_string(r);
        snprintf(d, sizeof(d), "; rank %d: last sign of life %.1f s ago at issued #%llu / finished #%llu (dead or frozen)", r, age, iss, fin);
      } else if (iss < t.seq) {
        behind += (behind.empty() ? "" : ",") + std::to_string(r);
        snprintf(d, sizeof(d), "; rank %d: has only issued #%llu (finished #%llu) - it never reached #%llu", r, iss, fin, t.seq);
      } else {
        snprintf(d, sizeof(d), "; rank %d: issued #%llu, finished #%llu", r, iss, fin);
      }
      detail += d;
    }
    if (!behind.empty()) msg += "; ranks BEHIND: [" + behind + "]";
    if (!silent.empty()) msg += "; ranks SILENT: [" + silent + "]";
    if (behind.empty() && silent.empty()) msg += "; every rank has issued it: the stall is inside the transport";
    msg += detail;
  }
  { std::lock_guard<std::mutex> l(mu); text = msg; }
  tripped.store(110);
  fprintf(stderr, "[dpa comm watchdog] %s\n", msg.c_str());
  fflush(stderr);
  comm->be->release_stall();        // (test hook only: a parked host function must not outlive the diagnosis)
  comm->be->abort();
  if (!exit_on_trip) return;
  // the caller gets `grace_ms` to notice (every call of this ABI now fails with the text above) and leave on its own
  std::unique_lock<std::mutex> l(mu);
  if (cv.wait_for(l, std::chrono::milliseconds(grace_ms), [&] { return stop; })) return;
  fprintf(stderr, "[dpa comm watchdog] rank %d did not leave within %.1f s of the abort: ending the process (exit 110)\n", comm->rank,
          grace_ms / 1000.0);
  fflush(stderr);
  _exit(110);
}

void Watchdog::loop() {
  auto next_hb = Clock::now();
  for (;;) {
    Tracked oldest{}; bool have = false; long long age_ms = 0;
    {
      std::unique_lock<std::mutex> l(mu);
      if (cv.wait_for(l, std::chrono::milliseconds(q.empty() ? 200 : 50), [&] { return stop; })) return;
      while (!q.empty()) {
        const int st = comm->be->query(q.front().slot);
        if (st == 1) break;
        finished.store(q.front().seq);         // finished (0) or not observable (< 0): retire it either way
        q.pop_front();
      }
      if (!q.empty()) {
        oldest = q.front(); have = true;
        age_ms = std::chrono::duration_cast<std::chrono::milliseconds>(Clock::now() - oldest.t0).count();
      }
    }
    if (Clock::now() >= next_hb) { publish(); next_hb = Clock::now() + std::chrono::milliseconds(hb_ms); }
    if (have && age_ms > timeout_ms && !tripped.load()) { trip(oldest, age_ms); return; }
  }
}

}  // namespace dpa

namespace {
int fail(Comm* c, int code, const char* what) {
  if (c) {
    const std::string& detail = c->be ? c->be->error : std::string();
    snprintf(c->last_error, sizeof(c->last_error), "%s failed (code %d)%s%s", what, code, detail.empty() ? "" : ": ", detail.c_str());
  }
  return code ? code : -1;
}
// 0, or 110 with the watchdog's diagnosis in last_error: the communicator is dead from here on
int tripped(Comm* c, const char* what) {
  if (!c->wd || !c->wd->tripped.load()) return 0;
  std::lock_guard<std::mutex> l(c->wd->mu);
  snprintf(c->last_error, sizeof(c->last_error), "%s refused: the communicator was aborted by its watchdog: %s", what, c->wd->text.c_str());
  return 110;
}
Comm* make_front(Backend* be, int world, int rank) {
  Comm* c = new Comm();
  c->be = be; c->world = world; c->rank = rank;
  const char* e = getenv("DPA_COMM_EVENT_RING");
  const int v = e ? atoi(e) : 0;
  c->ring = v > 0 ? v : 256;
  c->slot_busy.assign(c->ring, 0);
  const char* s = getenv("DPA_TEST_STALL_COLLECTIVE");
  const char* sr = getenv("DPA_TEST_STALL_RANK");
  if (s && (!sr || atoi(sr) == rank)) c->stall_at = atoll(s);
  return c;
}
// one slot of the ring for the next collective; -1 when the ring is exhausted
int take_slot(Comm* c) {
  const int idx = (int)(c->seq % (unsigned long long)c->ring);
  if (c->slot_busy[idx]) return -1;
  c->seq++;
  c->slot_busy[idx] = 1;
  if (c->stall_at >= 0 && (long long)c->seq == c->stall_at) c->be->stall_next();
  return idx;
}
void track(Comm* c, int slot, int bucket, int kind, long long bytes, void* compute) {
  Watchdog* w = c->wd;
  if (!w) return;
  w->issued.store(c->seq);
  if (!c->be->trackable(compute)) return;
  std::lock_guard<std::mutex> l(w->mu);
  if ((int)w->q.size() >= c->ring / 2)            // the host runs far ahead of the device: retire finished entries here too
    while (!w->q.empty() && c->be->query(w->q.front().slot) != 1) { w->finished.store(w->q.front().seq); w->q.pop_front(); }
  w->q.push_back(Tracked{c->seq, slot, bucket, kind, bytes, Clock::now()});
}
thread_local std::string g_create_error;
int finish_create(Comm* c) {
  if (int rc = c->be->prepare(c->ring)) { g_create_error = c->be->error; delete c->be; delete c; return rc; }
  return 0;
}
}  // namespace

DPA_API const char* dpa_comm_create_error() { return g_create_error.c_str(); }

// ---- RCCL transport (rccl_backend.cpp) ---------------------------------------------------------------------------
DPA_API void* dpa_comm_init(const void* id_bytes, int world, int rank, int device) {
  std::string err;
  Backend* be = make_rccl_backend(id_bytes, world, rank, device, &err);
  if (!be) { g_create_error = err; return nullptr; }
  Comm* c = make_front(be, world, rank);
  return finish_create(c) ? nullptr : c;
}
// ---- TCP transport (tcp_backend.cpp): create -> publish port -> connect to every peer -----------------------------
// device < 0: buffers are host memory.  device >= 0: buffers are device memory staged through pinned host buffers.
DPA_API void* dpa_comm_init_tcp(int world, int rank, int device) {
  std::string err;
  Backend* be = make_tcp_backend(world, rank, device, &err);
  if (!be) { g_create_error = err; return nullptr; }
  Comm* c = make_front(be, world, rank);
  return finish_create(c) ? nullptr : c;
}
DPA_API int dpa_comm_tcp_port(void* h) { return tcp_backend_port(((Comm*)h)->be); }
// peers_csv: "addr:port,addr:port,..." indexed by rank (this rank's own entry is ignored)
DPA_API int dpa_comm_tcp_connect(void* h, const char* peers_csv, int timeout_ms) {
  Comm* c = (Comm*)h;
  const int rc = tcp_backend_connect(c->be, peers_csv, timeout_ms);
  return rc ? fail(c, rc, "tcp mesh connect") : 0;
}
// ---- IPC mesh transport (mesh_backend.hip): create -> export handles -> connect with everyone's handles ------------
DPA_API void* dpa_comm_init_mesh(int world, int rank, int device, long long capacity_bytes) {
  std::string err;
  Backend* be = make_mesh_backend(world, rank, device, capacity_bytes, &err);
  if (!be) { g_create_error = err; return nullptr; }
  Comm* c = make_front(be, world, rank);
  return finish_create(c) ? nullptr : c;
}
DPA_API int dpa_comm_mesh_handle_size() { return mesh_backend_handle_size(); }
DPA_API int dpa_comm_mesh_export(void* h, void* out) {
  Comm* c = (Comm*)h;
  const int rc = mesh_backend_export(c->be, out);
  return rc ? fail(c, rc, "mesh export") : 0;
}
DPA_API int dpa_comm_mesh_connect(void* h, const void* all_handles) {
  Comm* c = (Comm*)h;
  const int rc = mesh_backend_connect(c->be, all_handles);
  return rc ? fail(c, rc, "mesh connect") : 0;
}
DPA_API void dpa_comm_mesh_unlink(void* h) { mesh_backend_unlink(((Comm*)h)->be); }

// The collective timeout: bounds every collective on every transport.  RCCL and the IPC mesh have no notion of time
// themselves - for them this is the watchdog's limit; the socket transport applies it per transfer as well (it can
// name the silent peer from the socket state), and its watchdog limit sits a little above so that diagnosis wins.
DPA_API void dpa_comm_set_timeout_ms(void* h, int timeout_ms) {
  Comm* c = (Comm*)h;
  if (!c || !c->be || timeout_ms <= 0) return;
  const bool tcp = !strcmp(c->be->name(), "tcp");
  if (tcp) tcp_backend_set_timeout(c->be, timeout_ms);
  if (!strcmp(c->be->name(), "mesh")) mesh_backend_set_timeout(c->be, timeout_ms);
  if (c->wd) {
    std::lock_guard<std::mutex> l(c->wd->mu);
    c->wd->timeout_ms = tcp ? timeout_ms + 2000 : timeout_ms;
    c->wd->hb_ms = timeout_ms / 4 < 2000 ? (timeout_ms / 4 > 100 ? timeout_ms / 4 : 100) : 2000;
  }
}
// Start the watchdog.  store / set_fn / get_fn: a rendezvous-store CLIENT of this thread's own (csrc/host/tcpstore.cpp:
// dpa_store_connect, dpa_store_set, dpa_store_get) for the progress records; null at world size 1.
// exit_grace_ms < 0: never end the process from the watchdog (tests that want to read the error).
DPA_API int dpa_comm_watchdog_start(void* h, int timeout_ms, int exit_grace_ms, void* store, void* set_fn, void* get_fn) {
  Comm* c = (Comm*)h;
  if (!c || c->wd) return 1;
  Watchdog* w = new Watchdog();
  w->comm = c;
  w->store = store; w->set_fn = (StoreSetFn)set_fn; w->get_fn = (StoreGetFn)get_fn;
  w->exit_on_trip = exit_grace_ms >= 0;
  w->grace_ms = exit_grace_ms >= 0 ? exit_grace_ms : 0;
  c->wd = w;
  dpa_comm_set_timeout_ms(h, timeout_ms > 0 ? timeout_ms : 600000);
  w->th = std::thread([w] { w->loop(); });
  return 0;
}
DPA_API int dpa_comm_watchdog_tripped(void* h) { Comm* c = (Comm*)h; return c && c->wd ? c->wd->tripped.load() : 0; }
// (issued, finished) sequence numbers as the watchdog sees them
DPA_API void dpa_comm_progress(void* h, unsigned long long* issued, unsigned long long* finished) {
  Comm* c = (Comm*)h;
  *issued = c->wd ? c->wd->issued.load() : c->seq;
  *finished = c->wd ? c->wd->finished.load() : 0;
}

DPA_API const char* dpa_comm_backend_name(void* h) { return ((Comm*)h)->be->name(); }
DPA_API int dpa_comm_world(void* h) { return ((Comm*)h)->world; }
DPA_API int dpa_comm_rank(void* h) { return ((Comm*)h)->rank; }
// How many ranks the transport itself connects (RCCL: ncclCommCount).  bench.py records it next to the number it prints.
DPA_API int dpa_comm_peer_count(void* h) { return ((Comm*)h)->be->peer_count(); }

// In-place fp32 SUM all-reduce of `count` floats at `ptr`, ordered after everything already queued on `compute`.
// Slots come from a ring (one fresh slot per collective) so that a GPU event is never re-recorded while an earlier
// wait on it may still be queued - iterations are enqueued back to back without host synchronisation.
DPA_API int dpa_comm_allreduce_async(void* h, float* ptr, long long count, int bucket, void* compute) {
  Comm* c = (Comm*)h;
  if (int rc = tripped(c, "all-reduce")) return rc;
  if (bucket < 0) return fail(c, 22, "all-reduce (negative bucket id)");
  if ((int)c->bucket_slot.size() <= bucket) c->bucket_slot.resize(bucket + 1, -1);
  if (c->bucket_slot[bucket] >= 0) return fail(c, 23, "all-reduce (bucket already has a collective in flight: missing wait)");
  const int idx = take_slot(c);
  if (idx < 0) return fail(c, 24, "all-reduce (slot ring exhausted: more collectives in flight than DPA_COMM_EVENT_RING)");
  c->bucket_slot[bucket] = idx;
  const int rc = c->be->allreduce(idx, ptr, count, compute);
  if (rc) { c->slot_busy[idx] = 0; c->bucket_slot[bucket] = -1; return fail(c, rc, "all-reduce"); }
  track(c, idx, bucket, 1, count * 4, compute);
  return 0;
}
// Make `compute` wait (on the device; the calling thread for host memory) for bucket's all-reduce.
DPA_API int dpa_comm_wait(void* h, int bucket, void* compute) {
  Comm* c = (Comm*)h;
  if (bucket < 0 || bucket >= (int)c->bucket_slot.size() || c->bucket_slot[bucket] < 0) return tripped(c, "wait");   // nothing in flight
  const int idx = c->bucket_slot[bucket];
  c->bucket_slot[bucket] = -1;
  c->slot_busy[idx] = 0;
  const int rc = c->be->wait(idx, compute);
  if (int t = tripped(c, "wait")) return t;
  return rc ? fail(c, rc, "wait") : 0;
}
DPA_API int dpa_comm_inflight(void* h) {
  Comm* c = (Comm*)h; int n = 0;
  for (char b : c->slot_busy) n += b;
  return n;
}
// Broadcast `bytes` from root, ordered after `compute`.  Returns a ticket (>= 0) for dpa_comm_broadcast_wait, which
// makes `compute` wait for it; < 0: -error code.  The pair lets the per-step BatchNorm-buffer broadcast run beside the
// first kernels of the forward pass instead of in front of them (X4 of SURVEY.md §2.6: "off the critical path").
DPA_API int dpa_comm_broadcast_async(void* h, void* ptr, long long bytes, int root, void* compute) {
  Comm* c = (Comm*)h;
  if (int rc = tripped(c, "broadcast")) return -rc;
  if (root < 0 || root >= c->world) return -fail(c, 22, "broadcast (root out of range)");
  const int idx = take_slot(c);
  if (idx < 0) return -fail(c, 24, "broadcast (slot ring exhausted: more collectives in flight than DPA_COMM_EVENT_RING)");
  const int rc = c->be->broadcast(idx, ptr, bytes, root, compute);
  if (rc) { c->slot_busy[idx] = 0; return -fail(c, rc, "broadcast"); }
  track(c, idx, -1, 2, bytes, compute);
  return idx;
}
DPA_API int dpa_comm_broadcast_wait(void* h, int ticket, void* compute) {
  Comm* c = (Comm*)h;
  if (ticket < 0 || ticket >= c->ring || !c->slot_busy[ticket]) return fail(c, 22, "broadcast wait (no such ticket)");
  c->slot_busy[ticket] = 0;
  const int rc = c->be->wait(ticket, compute);
  if (int t = tripped(c, "broadcast")) return t;
  return rc ? fail(c, rc, "broadcast") : 0;
}
DPA_API int dpa_comm_broadcast(void* h, void* ptr, long long bytes, int root, void* compute) {
  const int t = dpa_comm_broadcast_async(h, ptr, bytes, root, compute);
  return t < 0 ? -t : dpa_comm_broadcast_wait(h, t, compute);
}
// 0 = healthy; otherwise the transport's asynchronous error code or the watchdog's 110 (the trainer exits non-zero).
DPA_API int dpa_comm_async_error(void* h) {
  Comm* c = (Comm*)h;
  if (int t = tripped(c, "status")) return t;
  return c->be->async_error();
}
DPA_API const char* dpa_comm_last_error(void* h) { return ((Comm*)h)->last_error; }
DPA_API int dpa_comm_sync(void* h) { return ((Comm*)h)->be->sync(); }
DPA_API void dpa_comm_abort(void* h) { Comm* c = (Comm*)h; if (c && c->be) { c->be->abort(); c->be->release_stall(); } }
DPA_API void dpa_comm_destroy(void* h) {
  Comm* c = (Comm*)h;
  if (!c) return;
  if (c->wd) {
    { std::lock_guard<std::mutex> l(c->wd->mu); c->wd->stop = true; }
    c->wd->cv.notify_all();
    if (c->wd->th.joinable()) c->wd->th.join();
    delete c->wd;
  }
  c->be->release_stall();
  delete c->be;
  delete c;
}
