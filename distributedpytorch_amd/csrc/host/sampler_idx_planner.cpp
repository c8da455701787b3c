// Host runtime, part 1 (C ABI, loaded with ctypes):
//   * DistributedSampler index selection with torch's exact semantics (torch/utils/data/distributed.py:98-141;
//     used by /root/reference/dataloader.py:147-152): pad the permutation by wrapping, then take rank::world.
//   * IDX (MNIST) reader / writer (big-endian magic 2051 images / 2049 labels), the on-disk format
//     torchvision.datasets.MNIST expects under <root>/MNIST/raw (dataloader.py:92-93).
//   * Gradient bucket planner for the DDP reducer: buckets are contiguous ranges of the gradient arena taken in
//     backward (reverse registration) order, a small first bucket so communication starts early, then a size
//     cap chosen for xGMI.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define DPA_API extern "C" __attribute__((visibility("default")))

DPA_API int dpa_host_abi_version() { return 2; }

// perm: permutation (or identity) of [0, n).  out must hold ceil(n / world) entries (drop_last = false).
DPA_API long long dpa_sampler_select(const long long* perm, long long n, int world, int rank, long long* out) {
  const long long num_samples = (n + world - 1) / world;
  const long long total = num_samples * world;
  long long k = 0;
  for (long long i = rank; i < total; i += world) {
    // positions >= n wrap around to the head of the permutation, repeatedly if the pad exceeds n
    out[k++] = perm[n ? i % n : 0];
  }
  return k;
}

static uint32_t be32(const unsigned char* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

// Reads the header: dims[0..ndim). Returns ndim (1 labels, 3 images) or a negative error code.
DPA_API int dpa_idx_header(const char* path, int* dims) {
  FILE* f = fopen(path, "rb");
  if (!f) return -1;
  unsigned char h[4];
  if (fread(h, 1, 4, f) != 4) { fclose(f); return -2; }
  if (h[0] != 0 || h[1] != 0 || h[2] != 0x08) { fclose(f); return -3; }   // unsigned byte payload only
  const int nd = h[3];
  if (nd < 1 || nd > 4) { fclose(f); return -4; }
  for (int i = 0; i < nd; ++i) {
    unsigned char d[4];
    if (fread(d, 1, 4, f) != 4) { fclose(f); return -2; }
    dims[i] = (int)be32(d);
  }
  fclose(f);
  return nd;
}

DPA_API long long dpa_idx_read(const char* path, unsigned char* out, long long capacity) {
  int dims[4];
  const int nd = dpa_idx_header(path, dims);
  if (nd < 0) return nd;
  long long n = 1;
  for (int i = 0; i < nd; ++i) n *= dims[i];
  if (n > capacity) return -5;
  FILE* f = fopen(path, "rb");
  if (!f) return -1;
  fseek(f, 4 + 4 * nd, SEEK_SET);
  const long long got = (long long)fread(out, 1, (size_t)n, f);
  fclose(f);
  return got == n ? n : -2;
}

DPA_API int dpa_idx_write(const char* path, const unsigned char* data, const int* dims, int ndim) {
  FILE* f = fopen(path, "wb");
  if (!f) return -1;
  unsigned char h[4] = {0, 0, 0x08, (unsigned char)ndim};
  fwrite(h, 1, 4, f);
  long long n = 1;
  for (int i = 0; i < ndim; ++i) {
    const uint32_t d = (uint32_t)dims[i];
    unsigned char b[4] = {(unsigned char)(d >> 24), (unsigned char)(d >> 16), (unsigned char)(d >> 8), (unsigned char)d};
    fwrite(b, 1, 4, f);
    n *= dims[i];
  }
  const long long w = (long long)fwrite(data, 1, (size_t)n, f);
  fclose(f);
  return w == n ? 0 : -2;
}

// sizes[i]: bytes of parameter i in registration order; trainable[i]: participates in all-reduce.
// Parameters are visited in REVERSE order (the order gradients become ready).  A bucket closes when adding the next
// parameter would exceed its cap (first bucket: first_bytes, later: cap_bytes) or when a non-trainable parameter
// interrupts the contiguous arena range.  bucket_of[i] = bucket index (0 = first to become ready) or -1.
DPA_API int dpa_plan_buckets(const long long* sizes, const int* trainable, int n, long long first_bytes, long long cap_bytes,
                             int* bucket_of) {
  int bucket = -1;
  long long fill = 0;
  bool open = false;
  for (int i = n - 1; i >= 0; --i) {
    if (!trainable[i]) { bucket_of[i] = -1; open = false; continue; }
    const long long limit = (bucket <= 0) ? first_bytes : cap_bytes;   // bucket 0 is the small, early one
    if (!open || (fill > 0 && fill + sizes[i] > limit)) {
      ++bucket; fill = 0; open = true;
    }
    fill += sizes[i];
    bucket_of[i] = bucket;
  }
  return bucket + 1;
}
