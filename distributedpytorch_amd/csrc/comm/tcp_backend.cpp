// TCP transport of the native communicator: the CPU backend of the dpa_comm_* C ABI (SURVEY.md §5.8 "Test backend"; the
// role ProcessGroupGloo plays for a CPU run of the reference).  Same asynchronous contract as the RCCL transport:
//
//   allreduce(slot, ...)  queues the collective on ONE worker thread (the analogue of the comm stream: collectives run
//                         in issue order, overlapped with whatever the caller does next) and returns at once
//   wait(slot, ...)       blocks the caller until that collective is done
//
// All-reduce algorithm = the two-shot mesh all-reduce SURVEY.md §5.8 describes for 8 fully connected GPUs, over a
// socket mesh: every rank owns 1/world of the buffer; phase 1 every rank sends each owner its piece (reduce-scatter),
// the owner adds the pieces IN RANK ORDER (so every rank computes bit-identical sums); phase 2 every owner sends its
// reduced piece to everyone (all-gather).  All peers are served concurrently by one poll() loop.
// Every message carries a header (magic, opcode, sequence number, byte count): ranks that disagree about which
// collective comes next fail loudly instead of adding the wrong buffers (race detection); a peer that
// stalls longer than the timeout or closes its socket is NAMED in the error (failure detection, §5.3).
//
// Device memory (device >= 0): the worker stages through pinned host buffers on its own HIP stream - ready/done events
// exactly like the RCCL transport - so several ranks can share ONE GPU.  That is the rehearsal transport for a
// single-GPU box, where RCCL cannot be given one device per rank.
#include <arpa/inet.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "worker_backend.h"

namespace dpa {
namespace {

struct Header { uint32_t magic, op; uint64_t seq, bytes; };
constexpr uint32_t kMagic = 0x44504131;   // "DPA1"
enum Op : uint32_t { kScatter = 1, kGather = 2, kBcast = 3 };

struct Xfer {            // one direction-pair with one peer: send sbuf[0..slen), receive rbuf[0..rlen)
  int peer = -1, fd = -1;
  Header sh{}, rh{};
  const char* sbuf = nullptr; size_t slen = 0, soff = 0;     // soff counts header + payload bytes sent
  char* rbuf = nullptr; size_t rlen = 0, roff = 0;
  bool send = false, recv = false, checked = false;
};

struct TcpBackend : WorkerBackend {
  int listen_fd = -1, port = 0;
  std::vector<int> fds;                  // socket to each peer (own entry -1)
  uint64_t seq = 0;                      // collective sequence number (worker thread only)
  std::vector<char> scratch;             // phase-1 receive area: one piece per peer
  std::vector<float> accum;
  char* pinned = nullptr; size_t pinned_bytes = 0;      // device staging

  const char* name() const override { return "tcp"; }

  // ------------------------------------------------------------------------------------------- socket plumbing
  static void set_nonblock(int fd) { int fl = fcntl(fd, F_GETFL, 0); fcntl(fd, F_SETFL, fl | O_NONBLOCK); }

  // Progress every transfer until all are complete.  Returns 0, or an error naming the peer.
  int run(std::vector<Xfer>& xs, std::string* err) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
    for (;;) {
      std::vector<pollfd> pf; std::vector<int> ix;
      for (int i = 0; i < (int)xs.size(); ++i) {
        Xfer& x = xs[i];
        short ev = 0;
        if (x.send && x.soff < sizeof(Header) + x.slen) ev |= POLLOUT;
        if (x.recv && x.roff < sizeof(Header) + x.rlen) ev |= POLLIN;
        if (ev) { pf.push_back(pollfd{x.fd, ev, 0}); ix.push_back(i); }
      }
      if (pf.empty()) return 0;
      const auto now = std::chrono::steady_clock::now();
      if (now >= deadline) {
        char b[200]; snprintf(b, sizeof(b), "collective #%llu timed out after %d ms waiting for rank %d", (unsigned long long)seq,
                              timeout_ms, xs[ix[0]].peer);
        *err = b; return 110;
      }
      int wait_ms = (int)std::chrono::duration_cast<std::chrono::milliseconds>(deadline - now).count();
      if (wait_ms > 200) wait_ms = 200;
      { std::lock_guard<std::mutex> l(mu); if (stop) { *err = "communicator aborted"; return 125; } }
      const int pr = ::poll(pf.data(), pf.size(), wait_ms);
      if (pr < 0) { if (errno == EINTR) continue; *err = "poll failed"; return 5; }
      for (size_t k = 0; k < pf.size(); ++k) {
        Xfer& x = xs[ix[k]];
        if (pf[k].revents & (POLLERR | POLLNVAL)) {
          char b[120]; snprintf(b, sizeof(b), "socket error on the connection to rank %d", x.peer); *err = b; return 104;
        }
        if (pf[k].revents & POLLOUT) {
          const size_t total = sizeof(Header) + x.slen;
          while (x.soff < total) {
            const char* p; size_t n;
            if (x.soff < sizeof(Header)) { p = (const char*)&x.sh + x.soff; n = sizeof(Header) - x.soff; }
            else { p = x.sbuf + (x.soff - sizeof(Header)); n = total - x.soff; }
            const ssize_t w = ::send(x.fd, p, n, MSG_NOSIGNAL);
            if (w > 0) { x.soff += (size_t)w; continue; }
            if (w < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) break;
            if (w < 0 && errno == EINTR) continue;
            char b[120]; snprintf(b, sizeof(b), "rank %d closed its connection (send)", x.peer); *err = b; return 104;
          }
        }
        if (pf[k].revents & (POLLIN | POLLHUP)) {
          const size_t total = sizeof(Header) + x.rlen;
          while (x.roff < total) {
            char* p; size_t n;
            if (x.roff < sizeof(Header)) { p = (char*)&x.rh + x.roff; n = sizeof(Header) - x.roff; }
            else { p = x.rbuf + (x.roff - sizeof(Header)); n = total - x.roff; }
            const ssize_t r = ::recv(x.fd, p, n, 0);
            if (r > 0) {
              x.roff += (size_t)r;
              if (!x.checked && x.roff >= sizeof(Header)) {
                x.checked = true;
                if (x.rh.magic != kMagic || x.rh.op != x.sh.op || x.rh.seq != x.sh.seq || x.rh.bytes != x.rlen) {
                  char b[240];
                  snprintf(b, sizeof(b), "collective mismatch with rank %d: it is in op %u #%llu (%llu bytes), this rank in op %u #%llu "
                           "(%llu bytes)", x.peer, x.rh.op, (unsigned long long)x.rh.seq, (unsigned long long)x.rh.bytes, x.sh.op,
                           (unsigned long long)x.sh.seq, (unsigned long long)x.rlen);
                  *err = b; return 71;
                }
              }
              continue;
            }
            if (r < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) break;
            if (r < 0 && errno == EINTR) continue;
            char b[120]; snprintf(b, sizeof(b), "rank %d closed its connection", x.peer); *err = b; return 104;
          }
        }
      }
    }
  }

  Xfer make(int peer, Op op, const void* s, size_t sl, void* r, size_t rl, bool do_send, bool do_recv) {
    Xfer x; x.peer = peer; x.fd = fds[peer];
    x.sh = Header{kMagic, (uint32_t)op, seq, (uint64_t)sl};
    x.sbuf = (const char*)s; x.slen = sl; x.rbuf = (char*)r; x.rlen = rl; x.send = do_send; x.recv = do_recv;
    return x;
  }

  // piece p of an n-float buffer: [lo, hi), pieces aligned to 16 floats so every owner adds whole cache lines
  static void piece(long long n, int world, int p, long long* lo, long long* hi) {
    long long per = (n + world - 1) / world;
    per = (per + 15) / 16 * 16;
    *lo = per * p < n ? per * p : n;
    *hi = per * (p + 1) < n ? per * (p + 1) : n;
  }

  int host_allreduce(float* buf, long long n, std::string* err) {
    ++seq;
    if (world == 1) return 0;
    long long mlo, mhi; piece(n, world, rank, &mlo, &mhi);
    const size_t mine = (size_t)(mhi - mlo);
    if (scratch.size() < mine * sizeof(float) * world) scratch.resize(mine * sizeof(float) * world);
    // phase 1: reduce-scatter.  To peer j: my values of piece j.  From peer j: its values of my piece.
    std::vector<Xfer> xs;
    for (int j = 0; j < world; ++j) {
      if (j == rank) continue;
      long long lo, hi; piece(n, world, j, &lo, &hi);
      xs.push_back(make(j, kScatter, buf + lo, (size_t)(hi - lo) * sizeof(float), scratch.data() + (size_t)j * mine * sizeof(float),
                        mine * sizeof(float), true, true));
    }
    if (int rc = run(xs, err)) return rc;
    if (mine) {      // add in rank order 0, 1, ..., world-1 (own contribution at its place): identical on every rank
      if (accum.size() < mine) accum.resize(mine);
      for (int j = 0; j < world; ++j) {
        const float* src = j == rank ? buf + mlo : (const float*)(scratch.data() + (size_t)j * mine * sizeof(float));
        if (j == 0) memcpy(accum.data(), src, mine * sizeof(float));
        else for (size_t i = 0; i < mine; ++i) accum[i] += src[i];
      }
      memcpy(buf + mlo, accum.data(), mine * sizeof(float));
    }
    // phase 2: all-gather of the reduced pieces, received in place
    xs.clear();
    for (int j = 0; j < world; ++j) {
      if (j == rank) continue;
      long long lo, hi; piece(n, world, j, &lo, &hi);
      xs.push_back(make(j, kGather, buf + mlo, mine * sizeof(float), buf + lo, (size_t)(hi - lo) * sizeof(float), true, true));
    }
    return run(xs, err);
  }

  int host_broadcast(char* buf, long long bytes, int root, std::string* err) {
    ++seq;
    if (world == 1) return 0;
    std::vector<Xfer> xs;
    if (rank == root) {
      for (int j = 0; j < world; ++j) if (j != rank) xs.push_back(make(j, kBcast, buf, (size_t)bytes, nullptr, 0, true, false));
    } else {
      xs.push_back(make(root, kBcast, nullptr, 0, buf, (size_t)bytes, false, true));
    }
    return run(xs, err);
  }

  // ------------------------------------------------------------------------------------------- worker ("comm stream")
  char* staging(size_t bytes) {
    if (pinned_bytes < bytes) {
      if (pinned) (void)hipHostFree(pinned);
      pinned = nullptr; pinned_bytes = 0;
      if (hipHostMalloc((void**)&pinned, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
      pinned_bytes = bytes;
    }
    return pinned;
  }

  int execute(const Job& j, std::string* err) override {
    const size_t bytes = j.kind == 1 ? (size_t)j.n * sizeof(float) : (size_t)j.n;
    if (device < 0) {
      return j.kind == 1 ? host_allreduce((float*)j.ptr, j.n, err) : host_broadcast((char*)j.ptr, j.n, j.root, err);
    }
    // device memory: wait for the producers (ready event), stage down, run the host collective, stage up, record done
    if (hipSetDevice(device) != hipSuccess) { *err = "hipSetDevice failed in the comm worker"; return 1; }
    char* h = staging(bytes);
    if (!h) { *err = "cannot allocate the pinned staging buffer"; return 1; }
    hipError_t e = hipStreamWaitEvent(stream, j.ready, 0);
    const bool need_down = !(j.kind == 2 && rank != j.root);
    if (e == hipSuccess && need_down) e = hipMemcpyAsync(h, j.ptr, bytes, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { *err = std::string("staging device -> host: ") + hipGetErrorString(e); return 1; }
    const int rc = j.kind == 1 ? host_allreduce((float*)h, j.n, err) : host_broadcast(h, j.n, j.root, err);
    if (rc) return rc;
    const bool need_up = !(j.kind == 2 && rank == j.root);
    if (need_up) e = hipMemcpyAsync(j.ptr, h, bytes, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipEventRecord(j.done, stream);
    // (the pinned buffer is reused by the next job: drain the upload before returning)
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { *err = std::string("staging host -> device: ") + hipGetErrorString(e); return 1; }
    return 0;
  }

  int peer_count() override {
    int n = 1;
    for (int fd : fds) n += fd >= 0;
    return n;
  }
  void abort() override {
    WorkerBackend::abort();
    for (int fd : fds) if (fd >= 0) ::shutdown(fd, SHUT_RDWR);
  }
  ~TcpBackend() override {
    join_worker();
    for (int fd : fds) if (fd >= 0) ::close(fd);
    if (listen_fd >= 0) ::close(listen_fd);
    if (device >= 0 && pinned) (void)hipHostFree(pinned);
    destroy_device();
  }
};

bool send_all(int fd, const void* b, size_t n) {
  const char* p = (const char*)b;
  while (n) { ssize_t w = ::send(fd, p, n, MSG_NOSIGNAL); if (w <= 0) { if (w < 0 && errno == EINTR) continue; return false; } p += w; n -= (size_t)w; }
  return true;
}
bool recv_all(int fd, void* b, size_t n, int timeout_ms) {
  char* p = (char*)b;
  while (n) {
    pollfd pf{fd, POLLIN, 0};
    if (::poll(&pf, 1, timeout_ms) <= 0) return false;
    ssize_t r = ::recv(fd, p, n, 0);
    if (r <= 0) { if (r < 0 && errno == EINTR) continue; return false; }
    p += r; n -= (size_t)r;
  }
  return true;
}
}  // namespace

Backend* make_tcp_backend(int world, int rank, int device, std::string* err) {
  if (world < 1 || rank < 0 || rank >= world) { *err = "rank/world out of range"; return nullptr; }
  TcpBackend* b = new TcpBackend();
  b->world = world; b->rank = rank; b->device = device;
  b->fds.assign(world, -1);
  if (b->setup_device(err)) { delete b; return nullptr; }
  if (world > 1) {
    b->listen_fd = ::socket(AF_INET, SOCK_STREAM, 0);
    int one = 1; setsockopt(b->listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in a{}; a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_ANY); a.sin_port = 0;
    socklen_t l = sizeof(a);
    if (::bind(b->listen_fd, (sockaddr*)&a, sizeof(a)) != 0 || ::listen(b->listen_fd, world) != 0 ||
        getsockname(b->listen_fd, (sockaddr*)&a, &l) != 0) { *err = "cannot open the mesh listening socket"; delete b; return nullptr; }
    b->port = ntohs(a.sin_port);
  }
  b->start_worker();
  return b;
}
int tcp_backend_port(Backend* be) { return ((TcpBackend*)be)->port; }
void tcp_backend_set_timeout(Backend* be, int timeout_ms) {
  // DPA_TCP_SOCKET_TIMEOUT_MS (tests): give the per-transfer limit a value of its own, so that the front's watchdog -
  // the only bound the RCCL and mesh transports have - can be exercised on sockets too
  const char* e = getenv("DPA_TCP_SOCKET_TIMEOUT_MS");
  if (e && atoi(e) > 0) timeout_ms = atoi(e);
  ((TcpBackend*)be)->timeout_ms = timeout_ms > 0 ? timeout_ms : 600000;
}

// Mesh bring-up: rank i connects to every j < i (announcing its rank) and accepts one connection from every j > i.
int tcp_backend_connect(Backend* be, const char* peers_csv, int timeout_ms) {
  TcpBackend* b = (TcpBackend*)be;
  if (b->world == 1) return 0;
  std::vector<std::string> peers;
  { std::string s(peers_csv); size_t p = 0; while (p <= s.size()) { size_t q = s.find(',', p); if (q == std::string::npos) q = s.size(); peers.push_back(s.substr(p, q - p)); p = q + 1; } }
  if ((int)peers.size() != b->world) { b->error = "peer list does not have one entry per rank"; return 22; }
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
  for (int j = 0; j < b->rank; ++j) {
    const size_t c = peers[j].rfind(':');
    if (c == std::string::npos) { b->error = "bad peer address " + peers[j]; return 22; }
    const std::string host = peers[j].substr(0, c); const int port = atoi(peers[j].c_str() + c + 1);
    sockaddr_in a{}; a.sin_family = AF_INET; a.sin_port = htons((uint16_t)port);
    if (inet_pton(AF_INET, host.c_str(), &a.sin_addr) != 1) {        // a name, not a dotted address: resolve it once
      addrinfo hints{}; hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
      addrinfo* res = nullptr;
      if (getaddrinfo(host.c_str(), nullptr, &hints, &res) != 0 || !res) { b->error = "cannot resolve the address of rank " + std::to_string(j) + ": " + host; return 22; }
      a.sin_addr = ((sockaddr_in*)res->ai_addr)->sin_addr;
      freeaddrinfo(res);
    }
    for (;;) {
      int fd = ::socket(AF_INET, SOCK_STREAM, 0);
      if (::connect(fd, (sockaddr*)&a, sizeof(a)) == 0) {
        const uint32_t me = htonl((uint32_t)b->rank);
        if (!send_all(fd, &me, 4)) { ::close(fd); b->error = "handshake failed"; return 104; }
        b->fds[j] = fd; break;
      }
      ::close(fd);
      if (std::chrono::steady_clock::now() > deadline) {
        char m[160]; snprintf(m, sizeof(m), "cannot connect to rank %d at %s within %d ms", j, peers[j].c_str(), timeout_ms);
        b->error = m; return 110;
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  }
  for (int need = b->world - 1 - b->rank; need > 0; --need) {
    int left = (int)std::chrono::duration_cast<std::chrono::milliseconds>(deadline - std::chrono::steady_clock::now()).count();
    pollfd pf{b->listen_fd, POLLIN, 0};
    if (left <= 0 || ::poll(&pf, 1, left) <= 0) {
      std::string missing;
      for (int j = b->rank + 1; j < b->world; ++j) if (b->fds[j] < 0) missing += (missing.empty() ? "" : ",") + std::to_string(j);
      b->error = "mesh bring-up timed out: no connection from ranks [" + missing + "]"; return 110;
    }
    int fd = ::accept(b->listen_fd, nullptr, nullptr);
    if (fd < 0) { ++need; continue; }
    uint32_t who = 0;
    if (!recv_all(fd, &who, 4, 5000)) { ::close(fd); ++need; continue; }
    who = ntohl(who);
    if ((int)who <= b->rank || (int)who >= b->world || b->fds[who] >= 0) { ::close(fd); b->error = "unexpected peer in the mesh handshake"; return 71; }
    b->fds[who] = fd;
  }
  for (int j = 0; j < b->world; ++j) {
    if (b->fds[j] < 0) continue;
    int one = 1; setsockopt(b->fds[j], IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    int sz = 4 << 20; setsockopt(b->fds[j], SOL_SOCKET, SO_SNDBUF, &sz, sizeof(sz)); setsockopt(b->fds[j], SOL_SOCKET, SO_RCVBUF, &sz, sizeof(sz));
    TcpBackend::set_nonblock(b->fds[j]);
  }
  return 0;
}
}  // namespace dpa
