// Host runtime, part 2: a small TCP key-value rendezvous store (the role c10d::TCPStore plays for the reference at
// /root/reference/classif.py:86-87 via MASTER_ADDR/MASTER_PORT, main.py:128-129).
// Rank 0 runs the server thread; every rank is a client.  Operations: SET key value, GET key (blocks until the key
// exists or the timeout expires), ADD key delta (atomic counter; returns the new value).  The launcher uses it to
// (a) check that every rank of DDTNodes has arrived and NAME the missing ranks on timeout instead of hanging,
// (b) run barriers outside the collective library, (c) publish small blobs (e.g. a communicator id).
// Wire format: u8 op, u32 klen, key, u32 vlen, value  ->  i32 status, u32 vlen, value.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

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
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::string, std::string> kv;

  void serve(int fd) {
    for (;;) {
      uint8_t op; uint32_t klen, vlen;
      if (!recv_all(fd, &op, 1) || !recv_all(fd, &klen, 4)) break;
      std::string key(klen, '\0');
      if (klen && !recv_all(fd, &key[0], klen)) break;
      if (!recv_all(fd, &vlen, 4)) break;
      std::string val(vlen, '\0');
      if (vlen && !recv_all(fd, &val[0], vlen)) break;
      int32_t status = 0;
      std::string out;
      if (op == 1) {                       // SET
        { std::lock_guard<std::mutex> l(mu); kv[key] = val; }
        cv.notify_all();
      } else if (op == 2) {                // GET with timeout (ms in value)
        int64_t ms = 0; memcpy(&ms, val.data(), val.size() < 8 ? val.size() : 8);
        std::unique_lock<std::mutex> l(mu);
        bool ok = cv.wait_for(l, std::chrono::milliseconds(ms), [&] { return kv.count(key) > 0 || stop.load(); });
        if (ok && kv.count(key)) out = kv[key]; else status = -1;
      } else if (op == 3) {                // ADD
        int64_t d = 0; memcpy(&d, val.data(), val.size() < 8 ? val.size() : 8);
        std::lock_guard<std::mutex> l(mu);
        int64_t cur = 0;
        auto it = kv.find(key);
        if (it != kv.end() && it->second.size() == 8) memcpy(&cur, it->second.data(), 8);
        cur += d;
        kv[key] = std::string((const char*)&cur, 8);
        out = kv[key];
        cv.notify_all();
      } else if (op == 4) {                // CHECK (non-blocking exists)
        std::lock_guard<std::mutex> l(mu);
        status = kv.count(key) ? 0 : -1;
      } else status = -2;
      uint32_t olen = (uint32_t)out.size();
      if (!send_all(fd, &status, 4) || !send_all(fd, &olen, 4) || (olen && !send_all(fd, out.data(), olen))) break;
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
      workers.emplace_back([this, fd] { serve(fd); });
    }
  }
};

struct Client { int fd = -1; };

int request(Client* c, uint8_t op, const char* key, const void* val, uint32_t vlen, std::string* out) {
  uint32_t klen = (uint32_t)strlen(key);
  if (!send_all(c->fd, &op, 1) || !send_all(c->fd, &klen, 4) || !send_all(c->fd, key, klen) || !send_all(c->fd, &vlen, 4) ||
      (vlen && !send_all(c->fd, val, vlen)))
    return -100;
  int32_t status; uint32_t olen;
  if (!recv_all(c->fd, &status, 4) || !recv_all(c->fd, &olen, 4)) return -101;
  std::string o(olen, '\0');
  if (olen && !recv_all(c->fd, &o[0], olen)) return -101;
  if (out) *out = o;
  return status;
}

}  // namespace

DPA_API void* dpa_store_server_start(int port) {
  auto* s = new Server();
  s->listen_fd = ::socket(AF_INET, SOCK_STREAM, 0);
  int one = 1; setsockopt(s->listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
  sockaddr_in a{}; a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_ANY); a.sin_port = htons((uint16_t)port);
  if (::bind(s->listen_fd, (sockaddr*)&a, sizeof(a)) != 0 || ::listen(s->listen_fd, 256) != 0) {
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
  for (auto& w : s->workers) if (w.joinable()) w.detach();   // clients close their sockets on exit
  // the Server object is intentionally leaked if workers are still blocked in recv; process exit reclaims it
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
// returns value length (>= 0) or negative status (-1 = timeout)
DPA_API int dpa_store_get(void* h, const char* key, void* out, int capacity, long long timeout_ms) {
  std::string o;
  int st = request((Client*)h, 2, key, &timeout_ms, 8, &o);
  if (st != 0) return st < 0 ? st : -1;
  if ((int)o.size() > capacity) return -5;
  memcpy(out, o.data(), o.size());
  return (int)o.size();
}
DPA_API long long dpa_store_add(void* h, const char* key, long long delta) {
  std::string o;
  int st = request((Client*)h, 3, key, &delta, 8, &o);
  if (st != 0 || o.size() != 8) return INT64_MIN;
  long long v; memcpy(&v, o.data(), 8); return v;
}
DPA_API int dpa_store_check(void* h, const char* key) { return request((Client*)h, 4, key, nullptr, 0, nullptr); }
