// IPC mesh transport of the native communicator: the two-shot all-reduce of SURVEY.md §5.8 as OWN HIP kernels over
// hipIpc peer buffers - the answer to "8 GPUs, 7 point-to-point xGMI links each": every rank reduces 1/world of the
// bucket and PULLS its inputs from all peers at once (all links busy), then every rank pulls the reduced pieces.
// (Role: what NCCL does behind /root/reference/classif.py:86 'nccl'; opt-in, DPA_COMM_TRANSPORT=mesh; one node.)
//
//   per rank, allocated once and exported with hipIpcGetMemHandle:   in[capacity]   this rank's contribution
//                                                                    red[capacity/world + pad]  its reduced piece
//   stage 1   copy the bucket chunk into in (local write)                                            -> publish 3k+1
//   stage 2   wait for everyone's 3k+1;  red[i] = in_0[lo+i] + in_1[lo+i] + ... (rank order, remote reads,
//             local write: every rank computes bit-identical sums, the same sums as the socket transport)  -> publish 3k+2
//   stage 3   wait for everyone's 3k+2;  bucket[piece j] = red_j for every j (remote reads, local writes) -> publish 3k+3
//
// Every write is LOCAL and every cross-rank read happens in a kernel that was launched after the writer's kernel had
// finished (the writer synchronises its stream before it publishes a stage): data crosses ranks only over kernel
// boundaries, the one hand-off whose visibility does not depend on in-kernel cache-coherence details (per-XCD L2s).
// The stage counters live in POSIX shared memory (one cache line per rank); the waits are host-side spins with the
// collective timeout, an abort flag every rank honours, and an error that NAMES the rank whose counter is behind.
// Before any data moves, every rank publishes (collective number, kind, size) and checks its peers': ranks that
// disagree about the next collective fail loudly instead of adding the wrong buffers.
//
// Buffer reuse needs no extra synchronisation: a rank overwrites `in` in stage 1 of collective k+1 only after it has
// seen everyone's 3k+2 (all reads of in are done), and `red` in stage 2 of k+1 only after everyone's 3(k+1)+1 > 3k+3.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "worker_backend.h"

namespace dpa {
namespace {

constexpr int kMaxRanks = 16;

struct PeerPtrs { const float* p[kMaxRanks]; };

// out[i] = sum over ranks, in rank order, of in_r[lo + i]   (i < count; lo and the bases are 64-byte aligned)
__global__ void __launch_bounds__(256) mesh_reduce_kernel(PeerPtrs in, float* __restrict__ out, long long lo, long long count, int world) {
  const long long n4 = count >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 acc = reinterpret_cast<const float4*>(in.p[0] + lo)[i];
    for (int r = 1; r < world; ++r) {
      const float4 v = reinterpret_cast<const float4*>(in.p[r] + lo)[i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(out)[i] = acc;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
    float acc = in.p[0][lo + i];
    for (int r = 1; r < world; ++r) acc += in.p[r][lo + i];
    out[i] = acc;
  }
}

// dst[e] = red_{e / per}[e % per]   (e < count; per is a multiple of 16 floats)
__global__ void __launch_bounds__(256) mesh_gather_kernel(PeerPtrs red, float* __restrict__ dst, long long per, long long count, int vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    const long long n4 = count >> 2;
    for (long long i = t; i < n4; i += stride) {
      const long long e = i << 2;
      const long long j = e / per;
      reinterpret_cast<float4*>(dst)[i] = *reinterpret_cast<const float4*>(red.p[j] + (e - j * per));
    }
    for (long long e = (n4 << 2) + t; e < count; e += stride) { const long long j = e / per; dst[e] = red.p[j][e - j * per]; }
  } else {
    for (long long e = t; e < count; e += stride) { const long long j = e / per; dst[e] = red.p[j][e - j * per]; }
  }
}

__global__ void __launch_bounds__(256) mesh_copy_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, long long bytes, int vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    const long long n16 = bytes >> 4;
    for (long long i = t; i < n16; i += stride) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    for (long long i = (n16 << 4) + t; i < bytes; i += stride) dst[i] = src[i];
  } else {
    for (long long i = t; i < bytes; i += stride) dst[i] = src[i];
  }
}

struct alignas(64) RankLine {
  volatile uint64_t stage;          // 3k+1 / 3k+2 / 3k+3 of the rank's latest collective k (host atomics below)
  volatile uint64_t coll, kind, size;     // what the rank believes collective `coll` is
  uint64_t pad[4];
};
struct MeshShm {
  RankLine line[kMaxRanks];
  volatile uint32_t abort_flag;     // any rank sets it: every spin ends
  volatile int32_t abort_rank;
};

struct MeshExport {
  hipIpcMemHandle_t in_h, red_h;
  char shm_name[64];
  long long capacity;
  int pid, device;
};

struct MeshBackend : WorkerBackend {
  long long capacity = 0;             // bytes of `in`
  long long red_capacity = 0;
  float* in_local = nullptr; float* red_local = nullptr;
  PeerPtrs in{}, red{};
  std::vector<void*> opened;
  MeshShm* shm = nullptr;
  std::string shm_name;
  bool shm_owner = false;
  uint64_t coll = 0;                  // worker thread only
  bool connected = false;

  const char* name() const override { return "mesh"; }
  int peer_count() override { return connected || world == 1 ? world : 1; }

  static void piece(long long n, int world, int p, long long* lo, long long* hi, long long* per_out) {
    long long per = (n + world - 1) / world;
    per = (per + 15) / 16 * 16;
    *lo = per * p < n ? per * p : n;
    *hi = per * (p + 1) < n ? per * (p + 1) : n;
    if (per_out) *per_out = per;
  }
  static int blocks_for(long long items) {
    long long b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
  }
  void publish(uint64_t v) { __atomic_store_n(&shm->line[rank].stage, v, __ATOMIC_RELEASE); }

  // Wait until every rank in [first, last] has reached `target`.  Names the rank that is behind on timeout.
  int wait_stage(uint64_t target, int first, int last, std::string* err) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (int j = first; j <= last; ++j) {
      if (j == rank) continue;
      for (;;) {
        const uint64_t v = __atomic_load_n(&shm->line[j].stage, __ATOMIC_ACQUIRE);
        if (v >= target) break;
        if (__atomic_load_n(&shm->abort_flag, __ATOMIC_ACQUIRE)) {
          char b[160]; snprintf(b, sizeof(b), "mesh collective #%llu aborted by rank %d", (unsigned long long)coll, (int)shm->abort_rank);
          *err = b; return 125;
        }
        if ((++spins & 0x3ff) == 0) {
          { std::lock_guard<std::mutex> l(mu); if (stop) { *err = "communicator aborted"; return 125; } }
          const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
          if (ms > timeout_ms) {
            char b[240];
            snprintf(b, sizeof(b), "mesh collective #%llu timed out after %d ms waiting for rank %d (its stage counter is %llu, expected >= %llu: "
                     "it is %s)", (unsigned long long)coll, timeout_ms, j, (unsigned long long)v, (unsigned long long)target,
                     (v ? (v - 1) / 3 : 0) < coll ? "behind by at least one collective" : "inside this collective");
            *err = b; return 110;
          }
          if (ms > 2) usleep(50);          // long waits: stop burning the core
        }
      }
    }
    return 0;
  }
  int check_meta(uint64_t kind, uint64_t size, std::string* err) {
    for (int j = 0; j < world; ++j) {
      if (j == rank) continue;
      const RankLine& L = shm->line[j];
      if (L.coll != coll || L.kind != kind || L.size != size) {
        char b[240];
        snprintf(b, sizeof(b), "collective mismatch with rank %d: it is in op %llu #%llu (%llu bytes), this rank in op %llu #%llu (%llu bytes)", j,
                 (unsigned long long)L.kind, (unsigned long long)L.coll, (unsigned long long)L.size, (unsigned long long)kind,
                 (unsigned long long)coll, (unsigned long long)size);
        *err = b; return 71;
      }
    }
    return 0;
  }
  int hip_err(const char* what, hipError_t e, std::string* err) { *err = std::string(what) + ": " + hipGetErrorString(e); return 1; }

  // one chunk of an all-reduce: m floats at p
  int allreduce_chunk(float* p, long long m, std::string* err) {
    ++coll;
    const uint64_t base = 3 * coll;
    RankLine& me = shm->line[rank];
    me.coll = coll; me.kind = 1; me.size = (uint64_t)m * 4;
    hipError_t e = hipMemcpyAsync(in_local, p, (size_t)m * 4, hipMemcpyDeviceToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_err("mesh stage 1 (copy in)", e, err);
    publish(base + 1);
    if (int rc = wait_stage(base + 1, 0, world - 1, err)) return rc;
    if (int rc = check_meta(1, (uint64_t)m * 4, err)) return rc;
    long long lo, hi, per; piece(m, world, rank, &lo, &hi, &per);
    if (hi > lo) {
      hipLaunchKernelGGL(mesh_reduce_kernel, dim3(blocks_for((hi - lo) / 4 + 1)), dim3(256), 0, stream, in, red_local, lo, hi - lo, world);
      e = hipGetLastError();
      if (e != hipSuccess) return hip_err("mesh_reduce_kernel launch", e, err);
    }
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_err("mesh stage 2 (reduce)", e, err);
    publish(base + 2);
    if (int rc = wait_stage(base + 2, 0, world - 1, err)) return rc;
    const int vec = ((uintptr_t)p & 15) == 0;
    hipLaunchKernelGGL(mesh_gather_kernel, dim3(blocks_for(m / 4 + 1)), dim3(256), 0, stream, red, p, per, m, vec);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_err("mesh_gather_kernel launch", e, err);
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_err("mesh stage 3 (gather)", e, err);
    publish(base + 3);
    return 0;
  }
  // one chunk of a broadcast: `bytes` at p, through the root's `in` buffer
  int broadcast_chunk(unsigned char* p, long long bytes, int root, std::string* err) {
    ++coll;
    const uint64_t base = 3 * coll;
    RankLine& me = shm->line[rank];
    me.coll = coll; me.kind = 2 + 16 * (uint64_t)root; me.size = (uint64_t)bytes;
    hipError_t e = hipSuccess;
    if (rank == root) {
      e = hipMemcpyAsync(in_local, p, (size_t)bytes, hipMemcpyDeviceToDevice, stream);
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      if (e != hipSuccess) return hip_err("mesh broadcast (copy in)", e, err);
    }
    publish(base + 1);
    if (int rc = wait_stage(base + 1, 0, world - 1, err)) return rc;
    if (int rc = check_meta(2 + 16 * (uint64_t)root, (uint64_t)bytes, err)) return rc;
    if (rank != root) {
      const unsigned char* src = (const unsigned char*)in.p[root];
      const int vec = ((uintptr_t)p & 15) == 0;
      hipLaunchKernelGGL(mesh_copy_kernel, dim3(blocks_for(bytes / 16 + 1)), dim3(256), 0, stream, src, p, bytes, vec);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(stream);
      if (e != hipSuccess) return hip_err("mesh broadcast (pull)", e, err);
    }
    publish(base + 2);
    if (int rc = wait_stage(base + 2, 0, world - 1, err)) return rc;       // the root may now overwrite `in`
    publish(base + 3);
    return 0;
  }

  int execute(const Job& j, std::string* err) override {
    if (hipSetDevice(device) != hipSuccess) { *err = "hipSetDevice failed in the mesh worker"; return 1; }
    hipError_t e = hipStreamWaitEvent(stream, j.ready, 0);
    if (e != hipSuccess) return hip_err("hipStreamWaitEvent(mesh, ready)", e, err);
    if (world > 1) {
      if (!connected) { *err = "mesh transport used before dpa_comm_mesh_connect"; return 22; }
      if (j.kind == 1) {
        const long long cap = capacity / 4 / (16 * world) * (16 * world);     // floats per chunk: whole aligned pieces
        for (long long off = 0; off < j.n; off += cap) {
          const long long m = j.n - off < cap ? j.n - off : cap;
          if (int rc = allreduce_chunk((float*)j.ptr + off, m, err)) return rc;
        }
      } else {
        for (long long off = 0; off < j.n; off += capacity) {
          const long long m = j.n - off < capacity ? j.n - off : capacity;
          if (int rc = broadcast_chunk((unsigned char*)j.ptr + off, m, j.root, err)) return rc;
        }
      }
    }
    e = hipEventRecord(j.done, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_err("mesh: record done", e, err);
    return 0;
  }

  void abort() override {
    if (shm) { shm->abort_rank = rank; __atomic_store_n(&shm->abort_flag, 1u, __ATOMIC_RELEASE); }
    WorkerBackend::abort();
  }
  ~MeshBackend() override {
    join_worker();
    if (device >= 0) (void)hipSetDevice(device);
    for (void* p : opened) (void)hipIpcCloseMemHandle(p);
    if (in_local) (void)hipFree(in_local);
    if (red_local) (void)hipFree(red_local);
    if (shm) munmap((void*)shm, sizeof(MeshShm));
    if (shm_owner && !shm_name.empty()) shm_unlink(shm_name.c_str());
    destroy_device();
  }
};
}  // namespace

Backend* make_mesh_backend(int world, int rank, int device, long long capacity_bytes, std::string* err) {
  if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world) { *err = "mesh transport: rank/world out of range (at most 16 ranks of one node)"; return nullptr; }
  if (device < 0) { *err = "the mesh transport needs a GPU device"; return nullptr; }
  MeshBackend* b = new MeshBackend();
  b->world = world; b->rank = rank; b->device = device;
  if (b->setup_device(err)) { delete b; return nullptr; }
  if (world > 1) {
    const long long unit = 64LL * world;                                 // bytes: whole 16-float pieces per rank
    b->capacity = (capacity_bytes > 0 ? capacity_bytes : (64LL << 20)) / unit * unit;
    if (b->capacity < unit) b->capacity = unit;
    b->red_capacity = b->capacity / world + 256;
    hipError_t e = hipMalloc((void**)&b->in_local, (size_t)b->capacity);
    if (e == hipSuccess) e = hipMalloc((void**)&b->red_local, (size_t)b->red_capacity);
    if (e != hipSuccess) { *err = std::string("mesh buffers: ") + hipGetErrorString(e); delete b; return nullptr; }
    if (rank == 0) {
      char nm[64]; snprintf(nm, sizeof(nm), "/dpa_mesh_%d_%llx", (int)getpid(), (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
      const int fd = shm_open(nm, O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, sizeof(MeshShm)) != 0) { *err = "cannot create the mesh's shared-memory segment"; if (fd >= 0) close(fd); delete b; return nullptr; }
      void* m = mmap(nullptr, sizeof(MeshShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if (m == MAP_FAILED) { *err = "cannot map the mesh's shared-memory segment"; shm_unlink(nm); delete b; return nullptr; }
      memset(m, 0, sizeof(MeshShm));
      b->shm = (MeshShm*)m; b->shm_name = nm; b->shm_owner = true;
    }
  }
  b->start_worker();
  return b;
}
int mesh_backend_handle_size() { return (int)sizeof(MeshExport); }
int mesh_backend_export(Backend* be, void* out) {
  MeshBackend* b = (MeshBackend*)be;
  MeshExport x; memset(&x, 0, sizeof(x));
  if (b->world > 1) {
    hipError_t e = hipIpcGetMemHandle(&x.in_h, b->in_local);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&x.red_h, b->red_local);
    if (e != hipSuccess) { b->error = std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e) + " (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set before HIP initialises)"; return 1; }
  }
  snprintf(x.shm_name, sizeof(x.shm_name), "%s", b->shm_name.c_str());
  x.capacity = b->capacity; x.pid = (int)getpid(); x.device = b->device;
  memcpy(out, &x, sizeof(x));
  return 0;
}
int mesh_backend_connect(Backend* be, const void* all_handles) {
  MeshBackend* b = (MeshBackend*)be;
  if (b->world == 1) { b->connected = true; return 0; }
  const MeshExport* xs = (const MeshExport*)all_handles;
  if (hipSetDevice(b->device) != hipSuccess) { b->error = "hipSetDevice failed"; return 1; }
  for (int j = 0; j < b->world; ++j) {
    if (xs[j].capacity != b->capacity) { b->error = "mesh buffer capacity differs between ranks (DPA_MESH_BUFFER_MB)"; return 22; }
    if (j == b->rank) { b->in.p[j] = b->in_local; b->red.p[j] = b->red_local; continue; }
    void* pi = nullptr; void* pr = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&pi, xs[j].in_h, hipIpcMemLazyEnablePeerAccess);
    if (e == hipSuccess) { b->opened.push_back(pi); e = hipIpcOpenMemHandle(&pr, xs[j].red_h, hipIpcMemLazyEnablePeerAccess); }
    if (e != hipSuccess) {
      b->error = "hipIpcOpenMemHandle of rank " + std::to_string(j) + "'s buffers: " + hipGetErrorString(e) + " (the mesh transport connects GPUs of ONE node)";
      return 1;
    }
    b->opened.push_back(pr);
    b->in.p[j] = (const float*)pi; b->red.p[j] = (const float*)pr;
  }
  if (b->rank != 0) {
    const int fd = shm_open(xs[0].shm_name, O_RDWR, 0600);
    if (fd < 0) { b->error = std::string("cannot open the mesh's shared-memory segment ") + xs[0].shm_name + " (ranks on different nodes?)"; return 1; }
    void* m = mmap(nullptr, sizeof(MeshShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { b->error = "cannot map the mesh's shared-memory segment"; return 1; }
    b->shm = (MeshShm*)m;
  }
  b->connected = true;
  return 0;
}
void mesh_backend_set_timeout(Backend* be, int timeout_ms) { ((MeshBackend*)be)->timeout_ms = timeout_ms > 0 ? timeout_ms : 600000; }
// After every rank has connected (the caller ran a barrier): the segment's NAME is no longer needed - unlink it now so
// that nothing is left in /dev/shm if the job is killed later.
void mesh_backend_unlink(Backend* be) {
  MeshBackend* b = (MeshBackend*)be;
  if (b->shm_owner && !b->shm_name.empty()) { shm_unlink(b->shm_name.c_str()); b->shm_owner = false; }
}
}  // namespace dpa
