// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
// wave = 64 lanes; every block size used here is a multiple of 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DPA_API extern "C" __attribute__((visibility("default")))

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// Plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950), unlike the integer-rounding trick.
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 16-byte vector of T: 4 floats or 8 bf16.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  float4 raw;
  __device__ __forceinline__ void load(const float* p) { raw = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = raw; }
  __device__ __forceinline__ float get(int i) const { return ((const float*)&raw)[i]; }
  __device__ __forceinline__ void set(int i, float v) { ((float*)&raw)[i] = v; }
};
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  uint4 raw;
  __device__ __forceinline__ void load(const bf16_t* p) { raw = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<uint4*>(p) = raw; }
  __device__ __forceinline__ float get(int i) const {
    uint32_t w = ((const uint32_t*)&raw)[i >> 1];
    return __uint_as_float((i & 1) ? (w & 0xffff0000u) : (w << 16));
  }
  __device__ __forceinline__ void set(int i, float v) {
    uint32_t* w = &((uint32_t*)&raw)[i >> 1];
    uint32_t b = f2bf(v);
    *w = (i & 1) ? ((*w & 0x0000ffffu) | (b << 16)) : ((*w & 0xffff0000u) | b);
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Every DPA_API entry point that launches kernels returns this: 0, or the hipError_t of a rejected launch (too much
// dynamic LDS, an oversize grid, ...).  Outputs are torch.empty buffers, so a launch that silently did not happen would
// train on uninitialised memory; the Python wrappers raise on a non-zero status (ops/native.py).
static inline int dpa_launch_rc() { return (int)hipGetLastError(); }

// Memory-bound grid sizing: cap at 256 CUs x 8 blocks and grid-stride the rest.
// Pooling geometry (pool.hip; also the fused BatchNorm + max-pool backward in bn.hip)
struct PoolP {
  int N, H, W, C, Ho, Wo;
  int kh, kw, sh, sw, ph, pw;
  int x_cs, y_cs;
};

#ifdef __HIPCC__
// Gradient of a max-pool w.r.t. input pixel (n, hi, wi), channels c .. c + VEC: the sum of dy over the (few) windows that
// cover the pixel and whose recorded argmax is this tap.  Gather style: no atomics, deterministic.
__device__ __forceinline__ int div_small(int a, int s) {      // a >= 0; strides are 1 or 2 in every model of the zoo
  return s == 1 ? a : (s == 2 ? (a >> 1) : a / s);
}
template <typename T, int VEC>
__device__ __forceinline__ void maxpool_gather_grad(const T* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                    const PoolP& p, int dy_cs, int n, int hi, int wi, int c, float (&acc)[VEC]) {
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  // ho*sh - ph + r == hi  with 0 <= r < kh
  int ho_lo = (hi + p.ph - p.kh + 1 <= 0) ? 0 : div_small(hi + p.ph - p.kh + p.sh, p.sh);
  int ho_hi = div_small(hi + p.ph, p.sh); if (ho_hi >= p.Ho) ho_hi = p.Ho - 1;
  int wo_lo = (wi + p.pw - p.kw + 1 <= 0) ? 0 : div_small(wi + p.pw - p.kw + p.sw, p.sw);
  int wo_hi = div_small(wi + p.pw, p.sw); if (wo_hi >= p.Wo) wo_hi = p.Wo - 1;
  for (int ho = ho_lo; ho <= ho_hi; ++ho) {
    const int r = hi + p.ph - ho * p.sh;
    if (r < 0 || r >= p.kh) continue;
    for (int wo = wo_lo; wo <= wo_hi; ++wo) {
      const int s = wi + p.pw - wo * p.sw;
      if (s < 0 || s >= p.kw) continue;
      const unsigned char tap = (unsigned char)(r * p.kw + s);
      const long long opix = (long long)(n * p.Ho + ho) * p.Wo + wo;
      Vec16<T> g; g.load(dy + opix * dy_cs + c);
      const unsigned char* ip = idx + opix * p.C + c;
      unsigned char bi[VEC];
      if (VEC == 8) { uint2 pk = *reinterpret_cast<const uint2*>(ip); const unsigned char* b = (const unsigned char*)&pk;
#pragma unroll
        for (int e = 0; e < VEC; ++e) bi[e] = b[e];
      } else { uint32_t pk = *reinterpret_cast<const uint32_t*>(ip); const unsigned char* b = (const unsigned char*)&pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) bi[e] = b[e];
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) if (bi[e] == tap) acc[e] += g.get(e);
    }
  }
}
#endif

static inline int dpa_grid(long long work_items, int block) {
  long long g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (int)g;
}

// Philox-4x32-10 counter RNG (dropout masks, augmentation parameters).
struct Philox {
  uint32_t k0, k1;
  __host__ __device__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
  __host__ __device__ static inline void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    uint64_t p = (uint64_t)a * b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
  }
  __host__ __device__ inline void gen(uint64_t ctr, uint64_t stream, uint32_t out[4]) const {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = (uint32_t)stream, c3 = (uint32_t)(stream >> 32);
    uint32_t a = k0, b = k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      uint32_t h0, l0, h1, l1;
      mulhilo(0xD2511F53u, c0, h0, l0);
      mulhilo(0xCD9E8D57u, c2, h1, l1);
      uint32_t n0 = h1 ^ c1 ^ a, n1 = l1, n2 = h0 ^ c3 ^ b, n3 = l0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      a += 0x9E3779B9u; b += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
  }
};
__host__ __device__ inline float u32_to_unit(uint32_t x) { return (x >> 8) * (1.0f / 16777216.0f); }
