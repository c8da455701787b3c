// Elementwise / small reduction kernels for NHWC activations on gfx950 (K8, K11, K13, K18b of SURVEY.md §2.5):
// ReLU backward mask, strided add (gradient accumulation at fan-out), Philox dropout forward/backward (the mask
// is regenerated from the counter in backward, never stored), filter transpose for dgrad, per-channel column
// sums (bias gradients), fills.  All 16 bytes per lane, grid-stride, capped grids.
#include "common.h"

template <typename T>
__global__ void __launch_bounds__(256) relu_bwd_kernel(T* __restrict__ dy, int dy_cs, const T* __restrict__ y, int y_cs,
                                                       long long M, int C) {
  constexpr int VEC = 16 / sizeof(T);
  const int cpr = C / VEC;
  const long long total = M * cpr, stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long m = i / cpr; const int c = (int)(i - m * cpr) * VEC;
    Vec16<T> g, yv; g.load(dy + m * dy_cs + c); yv.load(y + m * y_cs + c);
#pragma unroll
    for (int e = 0; e < VEC; ++e) if (!(yv.get(e) > 0.f)) g.set(e, 0.f);
    g.store(dy + m * dy_cs + c);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) relu_fwd_kernel(const T* __restrict__ x, int x_cs, T* __restrict__ y, int y_cs,
                                                       long long M, int C) {
  constexpr int VEC = 16 / sizeof(T);
  const int cpr = C / VEC;
  const long long total = M * cpr, stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long m = i / cpr; const int c = (int)(i - m * cpr) * VEC;
    Vec16<T> v; v.load(x + m * x_cs + c);
#pragma unroll
    for (int e = 0; e < VEC; ++e) v.set(e, fmaxf(v.get(e), 0.f));
    v.store(y + m * y_cs + c);
  }
}

// out = a + b (out may alias a)
template <typename T>
__global__ void __launch_bounds__(256) add_kernel(const T* __restrict__ a, int a_cs, const T* __restrict__ b, int b_cs,
                                                  T* __restrict__ out, int o_cs, long long M, int C) {
  constexpr int VEC = 16 / sizeof(T);
  const int cpr = C / VEC;
  const long long total = M * cpr, stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long m = i / cpr; const int c = (int)(i - m * cpr) * VEC;
    Vec16<T> av, bv; av.load(a + m * a_cs + c); bv.load(b + m * b_cs + c);
#pragma unroll
    for (int e = 0; e < VEC; ++e) av.set(e, av.get(e) + bv.get(e));
    av.store(out + m * o_cs + c);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) copy_kernel(const T* __restrict__ a, int a_cs, T* __restrict__ out, int o_cs,
                                                   long long M, int C) {
  constexpr int VEC = 16 / sizeof(T);
  const int cpr = C / VEC;
  const long long total = M * cpr, stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long m = i / cpr; const int c = (int)(i - m * cpr) * VEC;
    Vec16<T> av; av.load(a + m * a_cs + c);
    av.store(out + m * o_cs + c);
  }
}

// rng: [0] = seed, [1] = step counter (device resident so a captured graph replays fresh masks)
// element id = m*C + c ; one Philox call yields 4 uniforms -> chunk of VEC elements uses VEC/4 calls
template <typename T>
__global__ void __launch_bounds__(256) dropout_kernel(const T* __restrict__ x, int x_cs, T* __restrict__ y, int y_cs,
                                                      long long M, int C, float p, const unsigned long long* __restrict__ rng,
                                                      unsigned long long layer_id) {
  constexpr int VEC = 16 / sizeof(T);
  const int cpr = C / VEC;
  const long long total = M * cpr, stride = (long long)gridDim.x * blockDim.x;
  const Philox ph(rng[0]);
  const unsigned long long stream = (rng[1] << 16) ^ layer_id;
  const float scale = 1.0f / (1.0f - p);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long m = i / cpr; const int c = (int)(i - m * cpr) * VEC;
    Vec16<T> v; v.load(x + m * x_cs + c);
#pragma unroll
    for (int k = 0; k < VEC / 4; ++k) {
      uint32_t r[4];
      ph.gen((unsigned long long)i * (VEC / 4) + k, stream, r);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float keep = u32_to_unit(r[e]) >= p ? scale : 0.f;
        v.set(k * 4 + e, v.get(k * 4 + e) * keep);
      }
    }
    v.store(y + m * y_cs + c);
  }
}

// w[K][RS][C] -> wt[C][RS][K]
template <typename T>
__global__ void __launch_bounds__(256) transpose_krsc_kernel(const T* __restrict__ w, T* __restrict__ wt, int K, int RS, int C) {
  const long long total = (long long)K * RS * C, stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    // iterate in output order for coalesced writes
    const int k = (int)(i % K); long long t = i / K;
    const int rs = (int)(t % RS); const int c = (int)(t / RS);
    wt[i] = w[((long long)k * RS + rs) * C + c];
  }
}

// out[c] (+)= sum_m x[m][c]   (bias gradients); one block per 64 channels, fixed-order reduction
template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const T* __restrict__ x, int cs, long long M, int C, float* __restrict__ out,
                                                     int accumulate) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C) for (long long m = rl; m < M; m += 4) s += Elem<T>::ld(x + m * cs + c);
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    out[c] = accumulate ? out[c] + s : s;
  }
}

__global__ void __launch_bounds__(256) fill_f32_kernel(float* __restrict__ p, float v, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}
__global__ void rng_advance_kernel(unsigned long long* rng) { rng[1] += 1ull; }

#define EW_LAUNCH(kern, total, ...)                                                                     \
  do {                                                                                                  \
    const int grid_ = dpa_grid(total, 256);                                                             \
    if (dtype == 1) hipLaunchKernelGGL(kern<bf16_t>, dim3(grid_), dim3(256), 0, s, __VA_ARGS__);        \
    else            hipLaunchKernelGGL(kern<float>, dim3(grid_), dim3(256), 0, s, __VA_ARGS__);         \
  } while (0)

DPA_API int dpa_relu_bwd(void* dy, int dy_cs, const void* y, int y_cs, long long M, int C, int dtype, hipStream_t s) {
  const long long total = M * (C / (dtype == 1 ? 8 : 4));
  if (dtype == 1) hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (bf16_t*)dy, dy_cs, (const bf16_t*)y, y_cs, M, C);
  else            hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (float*)dy, dy_cs, (const float*)y, y_cs, M, C);
  return dpa_launch_rc();
}
DPA_API int dpa_relu_fwd(const void* x, int x_cs, void* y, int y_cs, long long M, int C, int dtype, hipStream_t s) {
  const long long total = M * (C / (dtype == 1 ? 8 : 4));
  if (dtype == 1) hipLaunchKernelGGL(relu_fwd_kernel<bf16_t>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (const bf16_t*)x, x_cs, (bf16_t*)y, y_cs, M, C);
  else            hipLaunchKernelGGL(relu_fwd_kernel<float>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (const float*)x, x_cs, (float*)y, y_cs, M, C);
  return dpa_launch_rc();
}
DPA_API int dpa_add(const void* a, int a_cs, const void* b, int b_cs, void* out, int o_cs, long long M, int C, int dtype,
                     hipStream_t s) {
  const long long total = M * (C / (dtype == 1 ? 8 : 4));
  if (dtype == 1) hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (const bf16_t*)a, a_cs, (const bf16_t*)b, b_cs, (bf16_t*)out, o_cs, M, C);
  else            hipLaunchKernelGGL(add_kernel<float>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (const float*)a, a_cs, (const float*)b, b_cs, (float*)out, o_cs, M, C);
  return dpa_launch_rc();
}
DPA_API int dpa_copy(const void* a, int a_cs, void* out, int o_cs, long long M, int C, int dtype, hipStream_t s) {
  const long long total = M * (C / (dtype == 1 ? 8 : 4));
  if (dtype == 1) hipLaunchKernelGGL(copy_kernel<bf16_t>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (const bf16_t*)a, a_cs, (bf16_t*)out, o_cs, M, C);
  else            hipLaunchKernelGGL(copy_kernel<float>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (const float*)a, a_cs, (float*)out, o_cs, M, C);
  return dpa_launch_rc();
}
DPA_API int dpa_dropout(const void* x, int x_cs, void* y, int y_cs, long long M, int C, float p,
                         const unsigned long long* rng, unsigned long long layer_id, int dtype, hipStream_t s) {
  const long long total = M * (C / (dtype == 1 ? 8 : 4));
  if (dtype == 1) hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (const bf16_t*)x, x_cs, (bf16_t*)y, y_cs, M, C, p, rng, layer_id);
  else            hipLaunchKernelGGL(dropout_kernel<float>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (const float*)x, x_cs, (float*)y, y_cs, M, C, p, rng, layer_id);
  return dpa_launch_rc();
}
DPA_API int dpa_rng_advance(unsigned long long* rng, hipStream_t s) {
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, s, rng);
  return dpa_launch_rc();
}
DPA_API int dpa_transpose_krsc(const void* w, void* wt, int K, int RS, int C, int dtype, hipStream_t s) {
  const long long total = (long long)K * RS * C;
  if (dtype == 1) hipLaunchKernelGGL(transpose_krsc_kernel<bf16_t>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (const bf16_t*)w, (bf16_t*)wt, K, RS, C);
  else            hipLaunchKernelGGL(transpose_krsc_kernel<float>, dim3(dpa_grid(total, 256)), dim3(256), 0, s, (const float*)w, (float*)wt, K, RS, C);
  return dpa_launch_rc();
}
DPA_API int dpa_colsum(const void* x, int cs, long long M, int C, float* out, int accumulate, int dtype, hipStream_t s) {
  const int grid = (C + 63) / 64;
  if (dtype == 1) hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, cs, M, C, out, accumulate);
  else            hipLaunchKernelGGL(colsum_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, cs, M, C, out, accumulate);
  return dpa_launch_rc();
}
DPA_API int dpa_fill_f32(float* p, float v, long long n, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(fill_f32_kernel, dim3(dpa_grid(n, 256)), dim3(256), 0, s, p, v, n);
  return dpa_launch_rc();
}
