// Host runtime, part 2: a small TCP key-value rendezvous store (the role c10d::TCPStore plays for the reference at
// /root/reference/classif.py:86-87 via MASTER_ADDR/MASTER_PORT, main.py:128-129).
// Rank 0 runs the server thread; every rank is a client.  Operations: SET key value, GET key (blocks until the key
// exists or the timeout expires), ADD key delta (atomic counter; returns the new value).  The launcher uses it to
// (a) check that every rank of DDTNodes has arrived and NAME the missing ranks on timeout instead of hanging,
// (b) run barriers outside the collective library, (c) publish small blobs (e.g. a communicator id).
// Wire format (all integers in network byte order, so nodes of different endianness interoperate):
//   u8 op, u32 klen, key, u32 vlen, value  ->  i32 status, u32 vlen, value;  64-bit payloads (timeout, counter) big-endian.
// The server binds the address it is given (MASTER_ADDR when that is a local address: loopback-only for single-node
// runs) and falls back to all interfaces only when that address is not local.  stop() shuts every client socket down,
// joins every worker and frees the server.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <endian.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define DPA_API extern "C" __attribute__((visibility("default")))

namespace {

bool send_all(int fd, const void* buf, size_t n) {
  const char* p = (const char*)buf;
  while (n) {
    ssize_t w = ::send(fd, p, n, MSG_NOSIGNAL);
    if (w <= 0) return false;
    p += w; n -= (size_t)w;
  }
  return true;
}
bool recv_all(int fd, void* buf, size_t n) {
  char* p = (char*)buf;
  while (n) {
    ssize_t r = ::recv(fd, p, n, 0);
    if (r <= 0) return false;
    p += r; n -= (size_t)r;
  }
  return true;
}

struct Server {
  int listen_fd = -1;
  std::atomic<bool> stop{false};
  std::thread accept_thread;
  std::vector<std::thread> workers;
  std::vector<int> client_fds;           // guarded by mu; shut down on stop so blocked workers return
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::string, std::string> kv;

  void serve(int fd) {
    for (;;) {
      uint8_t op; uint32_t klen, vlen;
      if (!recv_all(fd, &op, 1) || !recv_all(fd, &klen, 4)) break;
      klen = ntohl(klen);
      if (klen > (1u << 16)) break;
      std::string key(klen, '\0');
      if (klen && !recv_all(fd, &key[0], klen)) break;
      if (!recv_all(fd, &vlen, 4)) break;
      vlen = ntohl(vlen);
      if (vlen > (1u << 30)) break;
      std::string val(vlen, '\0');
      if (vlen && !recv_all(fd, &val[0], vlen)) break;
      int32_t status = 0;
      std::string out;
      if (op == 1) {                       // SET
        { std::lock_guard<std::mutex> l(mu); kv[key] = val; }
        cv.notify_all();
      } else if (op == 2) {                // GET with timeout (ms, big-endian i64, in value)
        int64_t ms = 0; if (val.size() == 8) { uint64_t t; memcpy(&t, val.data(), 8); ms = (int64_t)be64toh(t); }
        std::unique_lock<std::mutex> l(mu);
        bool ok = cv.wait_for(l, std::chrono::milliseconds(ms), [&] { return kv.count(key) > 0 || stop.load(); });
        if (ok && kv.count(key)) out = kv[key]; else status = -1;
      } else if (op == 3) {                // ADD (delta and result big-endian i64)
        int64_t d = 0; if (val.size() == 8) { uint64_t t; memcpy(&t, val.data(), 8); d = (int64_t)be64toh(t); }
        std::lock_guard<std::mutex> l(mu);
        int64_t cur = 0;
        auto it = kv.find(key);
        if (it != kv.end() && it->second.size() == 8) { uint64_t t; memcpy(&t, it->second.data(), 8); cur = (int64_t)be64toh(t); }
        cur += d;
        const uint64_t t = htobe64((uint64_t)cur);
        kv[key] = std::string((const char*)&t, 8);
        out = kv[key];
        cv.notify_all();
      } else if (op == 4) {                // CHECK (non-blocking exists)
        std::lock_guard<std::mutex> l(mu);
        status = kv.count(key) ? 0 : -1;
      } else if (op == 5) {                // DELETE
        std::lock_guard<std::mutex> l(mu);
        status = kv.erase(key) ? 0 : -1;
      } else if (op == 6) {                // COUNT: number of keys held (diagnostics: the store must not grow with the job)
        std::lock_guard<std::mutex> l(mu);
        const uint64_t t = htobe64((uint64_t)kv.size());
        out = std::string((const char*)&t, 8);
      } else status = -2;
      const uint32_t olen = htonl((uint32_t)out.size());
      const int32_t st_n = (int32_t)htonl((uint32_t)status);
      if (!send_all(fd, &st_n, 4) || !send_all(fd, &olen, 4) || (!out.empty() && !send_all(fd, out.data(), out.size()))) break;
    }
    {
      std::lock_guard<std::mutex> l(mu);
      for (auto& f : client_fds) if (f == fd) f = -1;
    }
    ::close(fd);
  }

  void accept_loop() {
    while (!stop.load()) {
      pollfd p{listen_fd, POLLIN, 0};
      int r = ::poll(&p, 1, 200);
      if (r <= 0) continue;
      int fd = ::accept(listen_fd, nullptr, nullptr);
      if (fd < 0) continue;
      int one = 1; setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
      std::lock_guard<std::mutex> l(mu);
      client_fds.push_back(fd);
      workers.emplace_back([this, fd] { serve(fd); });
    }
  }
};

struct Client { int fd = -1; std::mutex mu; std::string last; };

int request(Client* c, uint8_t op, const char* key, const void* val, uint32_t vlen, std::string* out) {
  std::lock_guard<std::mutex> lk(c->mu);          // one request at a time per connection (threads may share a client)
  const uint32_t klen = (uint32_t)strlen(key);
  const uint32_t klen_n = htonl(klen), vlen_n = htonl(vlen);
  if (!send_all(c->fd, &op, 1) || !send_all(c->fd, &klen_n, 4) || !send_all(c->fd, key, klen) || !send_all(c->fd, &vlen_n, 4) ||
      (vlen && !send_all(c->fd, val, vlen)))
    return -100;
  int32_t status; uint32_t olen;
  if (!recv_all(c->fd, &status, 4) || !recv_all(c->fd, &olen, 4)) return -101;
  status = (int32_t)ntohl((uint32_t)status); olen = ntohl(olen);
  std::string o(olen, '\0');
  if (olen && !recv_all(c->fd, &o[0], olen)) return -101;
  if (out) *out = o;
  return status;
}

}  // namespace

// bind_addr: dotted IPv4 address to listen on (null / "" / not a local address: all interfaces).  Names are resolved by
// the caller (parallel/store.py: socket.gethostbyname) - this layer speaks dotted IPv4 only.
DPA_API void* dpa_store_server_start(const char* bind_addr, int port) {
  auto* s = new Server();
  s->listen_fd = ::socket(AF_INET, SOCK_STREAM, 0);
  int one = 1; setsockopt(s->listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
  sockaddr_in a{}; a.sin_family = AF_INET; a.sin_port = htons((uint16_t)port);
  bool bound = false;
  if (bind_addr && bind_addr[0] && inet_pton(AF_INET, bind_addr, &a.sin_addr) == 1)
    bound = ::bind(s->listen_fd, (sockaddr*)&a, sizeof(a)) == 0;          // fails with EADDRNOTAVAIL when not local
  if (!bound) {
    a.sin_addr.s_addr = htonl(INADDR_ANY);
    bound = ::bind(s->listen_fd, (sockaddr*)&a, sizeof(a)) == 0;
  }
  if (!bound || ::listen(s->listen_fd, 256) != 0) {
    ::close(s->listen_fd); delete s; return nullptr;
  }
  s->accept_thread = std::thread([s] { s->accept_loop(); });
  return s;
}
DPA_API int dpa_store_server_port(void* h) {
  auto* s = (Server*)h; sockaddr_in a{}; socklen_t l = sizeof(a);
  getsockname(s->listen_fd, (sockaddr*)&a, &l);
  return ntohs(a.sin_port);
}
DPA_API void dpa_store_server_stop(void* h) {
  auto* s = (Server*)h;
  s->stop.store(true);
  s->cv.notify_all();
  if (s->accept_thread.joinable()) s->accept_thread.join();
  ::shutdown(s->listen_fd, SHUT_RDWR); ::close(s->listen_fd);
  std::vector<std::thread> workers;
  {
    std::lock_guard<std::mutex> l(s->mu);
    for (int fd : s->client_fds) if (fd >= 0) ::shutdown(fd, SHUT_RDWR);     // unblocks every worker's recv
    workers.swap(s->workers);
  }
  s->cv.notify_all();
  for (auto& w : workers) if (w.joinable()) w.join();
  delete s;
}
DPA_API void* dpa_store_connect(const char* host, int port, int timeout_ms) {
  auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
  for (;;) {
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    sockaddr_in a{}; a.sin_family = AF_INET; a.sin_port = htons((uint16_t)port);
    if (inet_pton(AF_INET, host, &a.sin_addr) == 1 && ::connect(fd, (sockaddr*)&a, sizeof(a)) == 0) {
      int one = 1; setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
      auto* c = new Client(); c->fd = fd; return c;
    }
    ::close(fd);
    if (std::chrono::steady_clock::now() > deadline) return nullptr;
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
  }
}
DPA_API void dpa_store_close(void* h) { auto* c = (Client*)h; if (c) { ::close(c->fd); delete c; } }
DPA_API int dpa_store_set(void* h, const char* key, const void* val, int len) {
  return request((Client*)h, 1, key, val, (uint32_t)len, nullptr);
}
// returns value length (>= 0) or negative status (-1 = timeout, -5 = larger than `capacity`: the value is kept and
// dpa_store_last_size / dpa_store_last_copy retrieve it)
DPA_API int dpa_store_get(void* h, const char* key, void* out, int capacity, long long timeout_ms) {
  Client* c = (Client*)h;
  std::string o;
  const uint64_t t = htobe64((uint64_t)timeout_ms);
  int st = request(c, 2, key, &t, 8, &o);
  if (st != 0) return st < 0 ? st : -1;
  if ((int)o.size() > capacity) { c->last.swap(o); return -5; }
  memcpy(out, o.data(), o.size());
  return (int)o.size();
}
DPA_API long long dpa_store_last_size(void* h) { return (long long)((Client*)h)->last.size(); }
DPA_API long long dpa_store_last_copy(void* h, void* out, long long capacity) {
  Client* c = (Client*)h;
  if ((long long)c->last.size() > capacity) return -5;
  memcpy(out, c->last.data(), c->last.size());
  const long long n = (long long)c->last.size();
  std::string().swap(c->last);
  return n;
}
DPA_API long long dpa_store_add(void* h, const char* key, long long delta) {
  std::string o;
  const uint64_t d = htobe64((uint64_t)delta);
  int st = request((Client*)h, 3, key, &d, 8, &o);
  if (st != 0 || o.size() != 8) return INT64_MIN;
  uint64_t v; memcpy(&v, o.data(), 8); return (long long)be64toh(v);
}
DPA_API int dpa_store_check(void* h, const char* key) { return request((Client*)h, 4, key, nullptr, 0, nullptr); }
DPA_API int dpa_store_delete(void* h, const char* key) { return request((Client*)h, 5, key, nullptr, 0, nullptr); }
DPA_API long long dpa_store_num_keys(void* h) {
  std::string o;
  if (request((Client*)h, 6, "", nullptr, 0, &o) != 0 || o.size() != 8) return -1;
  uint64_t v; memcpy(&v, o.data(), 8); return (long long)be64toh(v);
}
