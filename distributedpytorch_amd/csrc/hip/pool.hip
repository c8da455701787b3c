// Pooling for NHWC activations on gfx950 (K9/K10 of SURVEY.md §2.5): MaxPool2d (incl. ceil_mode for SqueezeNet),
// AvgPool2d (count_include_pad as torch's default) and AdaptiveAvgPool2d, forward and backward.
// Memory-bound: one thread = one 16-byte channel chunk of one output pixel; backward is gather-style (each
// input pixel scans the few windows that cover it), so there are no atomics and results are deterministic.
#include "common.h"

template <typename T>
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          unsigned char* __restrict__ idx, PoolP p,
                                                          const float* __restrict__ scale, const float* __restrict__ shift) {
  // scale / shift != null: the pooled tensor is relu(x * scale[c] + shift[c]) - BatchNorm-apply + ReLU folded into the
  // window reads, so the full-resolution activation is never written or re-read (ResNet / DenseNet stems).
  constexpr int VEC = 16 / sizeof(T);
  const int cpr = p.C / VEC;
  const long long total = (long long)p.N * p.Ho * p.Wo * cpr;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % cpr) * VEC;
    long long t = i / cpr;
    const int wo = (int)(t % p.Wo); t /= p.Wo;
    const int ho = (int)(t % p.Ho);
    const int n = (int)(t / p.Ho);
    float best[VEC]; unsigned char bi[VEC];
    float sc[VEC], sf[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; bi[e] = 0; sc[e] = scale ? scale[c + e] : 1.f; sf[e] = scale ? shift[c + e] : 0.f; }
    bool first = true;
    for (int r = 0; r < p.kh; ++r) {
      const int hi = ho * p.sh - p.ph + r;
      if ((unsigned)hi >= (unsigned)p.H) continue;
      for (int s = 0; s < p.kw; ++s) {
        const int wi = wo * p.sw - p.pw + s;
        if ((unsigned)wi >= (unsigned)p.W) continue;
        Vec16<T> v; v.load(x + ((long long)(n * p.H + hi) * p.W + wi) * p.x_cs + c);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float f = v.get(e);
          if (scale) { f = fmaxf(f * sc[e] + sf[e], 0.f); if (sizeof(T) == 2) f = bf2f(f2bf(f)); }
          if (first || f > best[e] || f != f) { best[e] = f; bi[e] = (unsigned char)(r * p.kw + s); }
        }
        first = false;
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, best[e]);
    const long long opix = (long long)(n * p.Ho + ho) * p.Wo + wo;
    o.store(y + opix * p.y_cs + c);
    if (idx) {
      unsigned char* ip = idx + opix * p.C + c;
      if (VEC == 8) { uint2 pk; unsigned char* b = (unsigned char*)&pk;
#pragma unroll
        for (int e = 0; e < VEC; ++e) b[e] = bi[e];
        *reinterpret_cast<uint2*>(ip) = pk;
      } else { uint32_t pk; unsigned char* b = (unsigned char*)&pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = bi[e];
        *reinterpret_cast<uint32_t*>(ip) = pk;
      }
    }
  }
}

// dx[n,hi,wi,c] = sum over windows (ho,wo) covering (hi,wi) with argmax == this tap of dy[n,ho,wo,c]
template <typename T>
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                          T* __restrict__ dx, PoolP p, int dy_cs, int dx_cs) {
  constexpr int VEC = 16 / sizeof(T);
  const int cpr = p.C / VEC;
  const long long total = (long long)p.N * p.H * p.W * cpr;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int c, wi, hi, n;
    if (total <= 0x7fffffffLL) {               // 32-bit index arithmetic (64-bit divisions cost ~4x as many instructions)
      const uint32_t u = (uint32_t)i, t1 = u / (uint32_t)cpr, t2 = t1 / (uint32_t)p.W;
      c = (int)(u - t1 * (uint32_t)cpr) * VEC; wi = (int)(t1 - t2 * (uint32_t)p.W);
      n = (int)(t2 / (uint32_t)p.H); hi = (int)(t2 - (uint32_t)n * (uint32_t)p.H);
    } else {
      c = (int)(i % cpr) * VEC;
      long long t = i / cpr;
      wi = (int)(t % p.W); t /= p.W;
      hi = (int)(t % p.H);
      n = (int)(t / p.H);
    }
    float acc[VEC];
    maxpool_gather_grad<T, VEC>(dy, idx, p, dy_cs, n, hi, wi, c, acc);
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
    o.store(dx + ((long long)(n * p.H + hi) * p.W + wi) * dx_cs + c);
  }
}

// The 3x3 / stride-2 pool of the ResNet, DenseNet, SqueezeNet and Inception stems, specialised: one thread owns a 2x2 block of
// input pixels (x 16 bytes of channels).  The block is covered by exactly the four windows (A-1..A) x (B-1..B), so their
// dy / argmax chunks are loaded ONCE for four outputs (the generic gather loads 2.25 windows per pixel, with divergent trip
// counts 1 / 2 / 4 inside a wave and four integer divisions): 2.05 -> ~1 ms on the ResNet-50 stem at batch 1024.
// RED: the pool follows BatchNorm + ReLU (a network stem) and dx is the gradient w.r.t. that BatchNorm's output: the kernel
// also reads the BatchNorm's input z, masks with scale * z + shift > 0 and writes its block's partial sums of the
// BatchNorm-backward reduction (sum dz, sum dz * xhat) to slab[blockIdx][2][C] - the pass bn_bwd_reduce_kernel would make
// over dx and z right after (one read of a 3.3 GB tensor less on the ResNet-50 stem at batch 1024).  dx itself is written
// unmasked, as without RED (bn_bwd_dx_kernel masks from z again).  Needs 256 % (C / VEC) == 0: a thread's channel chunk
// is then the same in every iteration of its grid-stride loop.
struct PoolBn {
  const void* z; int z_cs, pad;
  const float *scale, *shift, *mean, *invstd;
  float* slab;
};
template <typename T, bool RED = false>
__global__ void __launch_bounds__(256) maxpool_bwd_k3s2_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                               T* __restrict__ dx, PoolP p, int dy_cs, int dx_cs, int nA, int nB,
                                                               PoolBn bn) {
  constexpr int VEC = 16 / sizeof(T);
  const uint32_t cpr = (uint32_t)(p.C / VEC);
  float rs[VEC], rq[VEC], mu[VEC], is[VEC], msc[VEC], msh[VEC];
  if constexpr (RED) {
    const int c0 = (int)(threadIdx.x % cpr) * VEC;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      rs[e] = 0.f; rq[e] = 0.f;
      mu[e] = bn.mean[c0 + e]; is[e] = bn.invstd[c0 + e]; msc[e] = bn.scale[c0 + e]; msh[e] = bn.shift[c0 + e];
    }
  }
  const uint32_t total = (uint32_t)p.N * nA * nB * cpr;          // < 2^31 (checked on the host)
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const uint32_t t1 = i / cpr, t2 = t1 / (uint32_t)nB, n = t2 / (uint32_t)nA;
    const int c = (int)(i - t1 * cpr) * VEC, B = (int)(t1 - t2 * (uint32_t)nB), A = (int)(t2 - n * (uint32_t)nA);
    float g[2][2][VEC];                        // dy of window (A-1+a, B-1+b); 0 where the window does not exist
    unsigned char tp[2][2][VEC];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ho = A - 1 + a, wo = B - 1 + b;
        const bool ok = ho >= 0 && ho < p.Ho && wo >= 0 && wo < p.Wo;
        const long long opix = ok ? ((long long)((int)n * p.Ho + ho) * p.Wo + wo) : 0;
        Vec16<T> v; v.load(dy + opix * dy_cs + c);
        const unsigned char* ip = idx + opix * p.C + c;
        if (VEC == 8) { uint2 pk = *reinterpret_cast<const uint2*>(ip); const unsigned char* q = (const unsigned char*)&pk;
#pragma unroll
          for (int e = 0; e < VEC; ++e) tp[a][b][e] = q[e];
        } else { uint32_t pk = *reinterpret_cast<const uint32_t*>(ip); const unsigned char* q = (const unsigned char*)&pk;
#pragma unroll
          for (int e = 0; e < 4; ++e) tp[a][b][e] = q[e];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) { g[a][b][e] = ok ? v.get(e) : 0.f; if (!ok) tp[a][b][e] = 255; }
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const int hi = 2 * A + j - p.ph, wi = 2 * B + l - p.pw;
        if (hi < 0 || hi >= p.H || wi < 0 || wi >= p.W) continue;
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          // window (A, B) sees this pixel under tap (j, l); (A-1, .) under row tap j + 2 (only j == 0), same for columns
          float acc = tp[1][1][e] == (unsigned char)(j * 3 + l) ? g[1][1][e] : 0.f;
          if (j == 0) acc += tp[0][1][e] == (unsigned char)(6 + l) ? g[0][1][e] : 0.f;
          if (l == 0) acc += tp[1][0][e] == (unsigned char)(j * 3 + 2) ? g[1][0][e] : 0.f;
          if (j == 0 && l == 0) acc += tp[0][0][e] == (unsigned char)8 ? g[0][0][e] : 0.f;
          o.set(e, acc);
        }
        const long long ipix = (long long)((int)n * p.H + hi) * p.W + wi;
        o.store(dx + ipix * dx_cs + c);
        if constexpr (RED) {
          Vec16<T> zv; zv.load(reinterpret_cast<const T*>(bn.z) + ipix * bn.z_cs + c);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float ze = zv.get(e);
            const float d = (ze * msc[e] + msh[e] > 0.f) ? o.get(e) : 0.f;      // (of the value as stored)
            rs[e] += d; rq[e] += d * (ze - mu[e]) * is[e];
          }
        }
      }
  }
  if constexpr (RED) {
    __shared__ float red[2][256][VEC > 4 ? 9 : 5];
    const int tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { red[0][tid][e] = rs[e]; red[1][tid][e] = rq[e]; }
    __syncthreads();
    if ((uint32_t)tid < cpr) {
      for (uint32_t l = 1; l < 256u / cpr; ++l)
#pragma unroll
        for (int e = 0; e < VEC; ++e) { rs[e] += red[0][l * cpr + tid][e]; rq[e] += red[1][l * cpr + tid][e]; }
      float* o = bn.slab + (size_t)blockIdx.x * 2 * p.C + (size_t)tid * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e) { o[e] = rs[e]; o[p.C + e] = rq[e]; }
    }
  }
}

// AvgPool2d, divisor = kh*kw (count_include_pad=True, torch default)
template <typename T>
__global__ void __launch_bounds__(256) avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, PoolP p) {
  constexpr int VEC = 16 / sizeof(T);
  const int cpr = p.C / VEC;
  const long long total = (long long)p.N * p.Ho * p.Wo * cpr;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const float inv = 1.0f / (float)(p.kh * p.kw);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % cpr) * VEC;
    long long t = i / cpr;
    const int wo = (int)(t % p.Wo); t /= p.Wo;
    const int ho = (int)(t % p.Ho);
    const int n = (int)(t / p.Ho);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int r = 0; r < p.kh; ++r) {
      const int hi = ho * p.sh - p.ph + r;
      if ((unsigned)hi >= (unsigned)p.H) continue;
      for (int s = 0; s < p.kw; ++s) {
        const int wi = wo * p.sw - p.pw + s;
        if ((unsigned)wi >= (unsigned)p.W) continue;
        Vec16<T> v; v.load(x + ((long long)(n * p.H + hi) * p.W + wi) * p.x_cs + c);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += v.get(e);
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, acc[e] * inv);
    o.store(y + ((long long)(n * p.Ho + ho) * p.Wo + wo) * p.y_cs + c);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, PoolP p,
                                                          int dy_cs, int dx_cs) {
  constexpr int VEC = 16 / sizeof(T);
  const int cpr = p.C / VEC;
  const long long total = (long long)p.N * p.H * p.W * cpr;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const float inv = 1.0f / (float)(p.kh * p.kw);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % cpr) * VEC;
    long long t = i / cpr;
    const int wi = (int)(t % p.W); t /= p.W;
    const int hi = (int)(t % p.H);
    const int n = (int)(t / p.H);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    int ho_lo = (hi + p.ph - p.kh + p.sh) / p.sh; if (hi + p.ph - p.kh + 1 <= 0) ho_lo = 0;
    int ho_hi = (hi + p.ph) / p.sh; if (ho_hi >= p.Ho) ho_hi = p.Ho - 1;
    int wo_lo = (wi + p.pw - p.kw + p.sw) / p.sw; if (wi + p.pw - p.kw + 1 <= 0) wo_lo = 0;
    int wo_hi = (wi + p.pw) / p.sw; if (wo_hi >= p.Wo) wo_hi = p.Wo - 1;
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
      const int r = hi + p.ph - ho * p.sh;
      if (r < 0 || r >= p.kh) continue;
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        const int s = wi + p.pw - wo * p.sw;
        if (s < 0 || s >= p.kw) continue;
        Vec16<T> g; g.load(dy + ((long long)(n * p.Ho + ho) * p.Wo + wo) * dy_cs + c);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += g.get(e);
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, acc[e] * inv);
    o.store(dx + ((long long)(n * p.H + hi) * p.W + wi) * dx_cs + c);
  }
}

// AdaptiveAvgPool2d: window i = [floor(i*H/Ho), ceil((i+1)*H/Ho))
__device__ __forceinline__ int ad_start(int i, int in, int out) { return (i * in) / out; }
__device__ __forceinline__ int ad_end(int i, int in, int out) { return ((i + 1) * in + out - 1) / out; }

template <typename T>
__global__ void __launch_bounds__(256) adaptive_avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, PoolP p) {
  constexpr int VEC = 16 / sizeof(T);
  const int cpr = p.C / VEC;
  const long long total = (long long)p.N * p.Ho * p.Wo * cpr;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % cpr) * VEC;
    long long t = i / cpr;
    const int wo = (int)(t % p.Wo); t /= p.Wo;
    const int ho = (int)(t % p.Ho);
    const int n = (int)(t / p.Ho);
    const int h0 = ad_start(ho, p.H, p.Ho), h1 = ad_end(ho, p.H, p.Ho);
    const int w0 = ad_start(wo, p.W, p.Wo), w1 = ad_end(wo, p.W, p.Wo);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int hi = h0; hi < h1; ++hi)
      for (int wi = w0; wi < w1; ++wi) {
        Vec16<T> v; v.load(x + ((long long)(n * p.H + hi) * p.W + wi) * p.x_cs + c);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += v.get(e);
      }
    const float inv = 1.0f / (float)((h1 - h0) * (w1 - w0));
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, acc[e] * inv);
    o.store(y + ((long long)(n * p.Ho + ho) * p.Wo + wo) * p.y_cs + c);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) adaptive_avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, PoolP p,
                                                                   int dy_cs, int dx_cs) {
  constexpr int VEC = 16 / sizeof(T);
  const int cpr = p.C / VEC;
  const long long total = (long long)p.N * p.H * p.W * cpr;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % cpr) * VEC;
    long long t = i / cpr;
    const int wi = (int)(t % p.W); t /= p.W;
    const int hi = (int)(t % p.H);
    const int n = (int)(t / p.H);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    // candidate output windows: those whose [start,end) contains hi
    int ho_lo = (hi * p.Ho) / p.H; if (ho_lo > 0) --ho_lo;
    int wo_lo = (wi * p.Wo) / p.W; if (wo_lo > 0) --wo_lo;
    for (int ho = ho_lo; ho < p.Ho; ++ho) {
      const int h0 = ad_start(ho, p.H, p.Ho), h1 = ad_end(ho, p.H, p.Ho);
      if (h0 > hi) break;
      if (hi >= h1) continue;
      for (int wo = wo_lo; wo < p.Wo; ++wo) {
        const int w0 = ad_start(wo, p.W, p.Wo), w1 = ad_end(wo, p.W, p.Wo);
        if (w0 > wi) break;
        if (wi >= w1) continue;
        const float inv = 1.0f / (float)((h1 - h0) * (w1 - w0));
        Vec16<T> g; g.load(dy + ((long long)(n * p.Ho + ho) * p.Wo + wo) * dy_cs + c);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += g.get(e) * inv;
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
    o.store(dx + ((long long)(n * p.H + hi) * p.W + wi) * dx_cs + c);
  }
}

#define POOL_DISPATCH(kern, total, ...)                                                                       \
  do {                                                                                                        \
    const int grid = dpa_grid(total, 256);                                                                    \
    if (dtype == 1) hipLaunchKernelGGL(kern<bf16_t>, dim3(grid), dim3(256), 0, s, __VA_ARGS__);               \
    else            hipLaunchKernelGGL(kern<float>, dim3(grid), dim3(256), 0, s, __VA_ARGS__);                \
  } while (0)

// 3x3 windows (any stride / padding): the nine loads of a window are issued together, unconditionally (clamped coordinates,
// a validity bit per tap), before the first comparison - the general kernel above walks the window with two runtime loops and
// a branch per tap, i.e. nine dependent round trips per output (ResNet-50 stem at batch 1024: 1.48 ms for 4.3 GB).
template <typename T>
__global__ void __launch_bounds__(256) maxpool_fwd_k3_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                             unsigned char* __restrict__ idx, PoolP p,
                                                             const float* __restrict__ scale, const float* __restrict__ shift) {
  constexpr int VEC = 16 / sizeof(T);
  const int cpr = p.C / VEC;
  const long long total = (long long)p.N * p.Ho * p.Wo * cpr;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % cpr) * VEC;
    long long t = i / cpr;
    const int wo = (int)(t % p.Wo); t /= p.Wo;
    const int ho = (int)(t % p.Ho);
    const int n = (int)(t / p.Ho);
    const int h0 = ho * p.sh - p.ph, w0 = wo * p.sw - p.pw;
    Vec16<T> v[9];
    unsigned valid = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int hi = h0 + r, wi = w0 + s;
        const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        valid |= (ok ? 1u : 0u) << (r * 3 + s);
        const int hc = min(max(hi, 0), p.H - 1), wc = min(max(wi, 0), p.W - 1);
        v[r * 3 + s].load(x + ((long long)(n * p.H + hc) * p.W + wc) * p.x_cs + c);
      }
    float best[VEC]; unsigned char bi[VEC];
    float sc[VEC], sf[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; bi[e] = 0; sc[e] = scale ? scale[c + e] : 1.f; sf[e] = scale ? shift[c + e] : 0.f; }
    bool first = true;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const bool ok = (valid >> k) & 1u;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float f = v[k].get(e);
        if (scale) { f = fmaxf(f * sc[e] + sf[e], 0.f); if (sizeof(T) == 2) f = bf2f(f2bf(f)); }
        if (ok && (first || f > best[e] || f != f)) { best[e] = f; bi[e] = (unsigned char)k; }
      }
      first = first && !ok;
    }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, best[e]);
    const long long opix = (long long)(n * p.Ho + ho) * p.Wo + wo;
    o.store(y + opix * p.y_cs + c);
    if (idx) {
      unsigned char* ip = idx + opix * p.C + c;
      if (VEC == 8) { uint2 pk; unsigned char* b = (unsigned char*)&pk;
#pragma unroll
        for (int e = 0; e < VEC; ++e) b[e] = bi[e];
        *reinterpret_cast<uint2*>(ip) = pk;
      } else { uint32_t pk; unsigned char* b = (unsigned char*)&pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = bi[e];
        *reinterpret_cast<uint32_t*>(ip) = pk;
      }
    }
  }
}

DPA_API int dpa_maxpool_fwd(const void* x, void* y, unsigned char* idx, const PoolP* pp, int dtype, hipStream_t s,
                             const float* scale, const float* shift) {
  PoolP p = *pp; const int vec = dtype == 1 ? 8 : 4;
  const long long total = (long long)p.N * p.Ho * p.Wo * (p.C / vec);
  static const int fast3 = [] { const char* e = getenv("DPA_POOL_K3"); return (e && e[0] == '0') ? 0 : 1; }();
  if (fast3 && p.kh == 3 && p.kw == 3) {
    if (dtype == 1) hipLaunchKernelGGL(maxpool_fwd_k3_kernel<bf16_t>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, idx, p, scale, shift);
    else            hipLaunchKernelGGL(maxpool_fwd_k3_kernel<float>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (const float*)x, (float*)y, idx, p, scale, shift);
    return dpa_launch_rc();
  }
  if (dtype == 1) hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, idx, p, scale, shift);
  else            hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (const float*)x, (float*)y, idx, p, scale, shift);
  return dpa_launch_rc();
}
DPA_API int dpa_maxpool_bwd(const void* dy, const unsigned char* idx, void* dx, const PoolP* pp, int dy_cs, int dx_cs,
                             int dtype, hipStream_t s) {
  PoolP p = *pp; const int vec = dtype == 1 ? 8 : 4;
  const long long total = (long long)p.N * p.H * p.W * (p.C / vec);
  static const int fast = [] { const char* e = getenv("DPA_POOL_K3S2"); return (e && e[0] == '0') ? 0 : 1; }();
  if (fast && p.kh == 3 && p.kw == 3 && p.sh == 2 && p.sw == 2 && p.ph <= 1 && p.pw <= 1) {
    // 2x2 input blocks: pixel hi = 2A + j - ph; the last block must reach row H - 1 (and window Ho - 1)
    const int nA = (p.H - 1 + p.ph) / 2 + 1 > p.Ho ? (p.H - 1 + p.ph) / 2 + 1 : p.Ho;
    const int nB = (p.W - 1 + p.pw) / 2 + 1 > p.Wo ? (p.W - 1 + p.pw) / 2 + 1 : p.Wo;
    const long long blocks = (long long)p.N * nA * nB * (p.C / vec);
    if (blocks < 0x7fffffffLL) {
      if (dtype == 1) hipLaunchKernelGGL(maxpool_bwd_k3s2_kernel<bf16_t>, dim3(dpa_grid(blocks, 256)), dim3(256), 0, s, (const bf16_t*)dy, idx, (bf16_t*)dx, p, dy_cs, dx_cs, nA, nB, PoolBn{});
      else            hipLaunchKernelGGL(maxpool_bwd_k3s2_kernel<float>, dim3(dpa_grid(blocks, 256)), dim3(256), 0, s, (const float*)dy, idx, (float*)dx, p, dy_cs, dx_cs, nA, nB, PoolBn{});
      return dpa_launch_rc();
    }
  }
  if (dtype == 1) hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (const bf16_t*)dy, idx, (bf16_t*)dx, p, dy_cs, dx_cs);
  else            hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (const float*)dy, idx, (float*)dx, p, dy_cs, dx_cs);
  return dpa_launch_rc();
}
// 3x3 / stride-2 max-pool backward + the partial sums of the BatchNorm(+ReLU) in front of the pool (see the kernel).  Returns
// the number of slab rows written (>= 1), or 0 when this geometry is not taken (nothing launched: the caller runs the two
// separate kernels), or a negative launch error.  slab: [>= 2048][2][C] floats.
DPA_API int dpa_maxpool_bwd_bn(const void* dy, const unsigned char* idx, void* dx, const PoolP* pp, int dy_cs, int dx_cs,
                                int dtype, hipStream_t s, const void* z, int z_cs, const float* scale, const float* shift,
                                const float* mean, const float* invstd, float* slab) {
  PoolP p = *pp; const int vec = dtype == 1 ? 8 : 4;
  static const int on = [] { const char* e = getenv("DPA_POOL_BN"); return (e && e[0] == '0') ? 0 : 1; }();
  if (!on || !(p.kh == 3 && p.kw == 3 && p.sh == 2 && p.sw == 2 && p.ph <= 1 && p.pw <= 1)) return 0;
  const int cpr = p.C / vec;
  if (cpr < 1 || cpr > 256 || 256 % cpr || p.C % vec) return 0;
  const int nA = (p.H - 1 + p.ph) / 2 + 1 > p.Ho ? (p.H - 1 + p.ph) / 2 + 1 : p.Ho;
  const int nB = (p.W - 1 + p.pw) / 2 + 1 > p.Wo ? (p.W - 1 + p.pw) / 2 + 1 : p.Wo;
  const long long blocks = (long long)p.N * nA * nB * cpr;
  if (blocks >= 0x7fffffffLL) return 0;
  const int grid = dpa_grid(blocks, 256);
  PoolBn bn{z, z_cs, 0, scale, shift, mean, invstd, slab};
  if (dtype == 1) hipLaunchKernelGGL((maxpool_bwd_k3s2_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, s, (const bf16_t*)dy, idx, (bf16_t*)dx, p, dy_cs, dx_cs, nA, nB, bn);
  else            hipLaunchKernelGGL((maxpool_bwd_k3s2_kernel<float, true>), dim3(grid), dim3(256), 0, s, (const float*)dy, idx, (float*)dx, p, dy_cs, dx_cs, nA, nB, bn);
  const int rc = dpa_launch_rc();
  return rc ? -rc : grid;
}
DPA_API int dpa_avgpool_fwd(const void* x, void* y, const PoolP* pp, int adaptive, int dtype, hipStream_t s) {
  PoolP p = *pp; const int vec = dtype == 1 ? 8 : 4;
  const long long total = (long long)p.N * p.Ho * p.Wo * (p.C / vec);
  const int grid = dpa_grid(total, 256);
  if (adaptive) {
    if (dtype == 1) hipLaunchKernelGGL(adaptive_avgpool_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, p);
    else            hipLaunchKernelGGL(adaptive_avgpool_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, p);
  } else {
    if (dtype == 1) hipLaunchKernelGGL(avgpool_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, p);
    else            hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, p);
  }
  return dpa_launch_rc();
}
DPA_API int dpa_avgpool_bwd(const void* dy, void* dx, const PoolP* pp, int dy_cs, int dx_cs, int adaptive, int dtype,
                             hipStream_t s) {
  PoolP p = *pp; const int vec = dtype == 1 ? 8 : 4;
  const long long total = (long long)p.N * p.H * p.W * (p.C / vec);
  const int grid = dpa_grid(total, 256);
  if (adaptive) {
    if (dtype == 1) hipLaunchKernelGGL(adaptive_avgpool_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, p, dy_cs, dx_cs);
    else            hipLaunchKernelGGL(adaptive_avgpool_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, (float*)dx, p, dy_cs, dx_cs);
  } else {
    if (dtype == 1) hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, p, dy_cs, dx_cs);
    else            hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, (float*)dx, p, dy_cs, dx_cs);
  }
  return dpa_launch_rc();
}
DPA_API int dpa_pool_params_size() { return (int)sizeof(PoolP); }
