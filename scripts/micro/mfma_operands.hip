// Does the bf16 MFMA rate depend on where the operands come from?  peak<0>: every MFMA reads the SAME A/B registers
// (scripts/micro/mfma_peak.hip); peak<1>: a 4x4 outer product over 4 distinct A and 4 distinct B fragments, as in the
// conv kernels (TM = TN = 4); peak<2>: the same with the fragments rotated every iteration (fresh registers per step).
// build: hipcc --offload-arch=gfx950 -O3 -w scripts/micro/mfma_operands.hip -o scripts/micro/bin/mfma_operands
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int MODE>
__global__ void __launch_bounds__(256) peak(float* out, const float* in, int iters) {
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[8], fb[8];
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 8; ++e) { fa[i][e] = (__bf16)in[(threadIdx.x + i * 8 + e) & 1023]; fb[i][e] = (__bf16)in[(threadIdx.x * 3 + i * 8 + e) & 1023]; }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[0], acc[i][j], 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    } else {
      const int o = (it & 1) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = (it & 1) ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[4 + i], fb[4 + j], acc[i][j], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
typedef __attribute__((ext_vector_type(16))) float f32x16;
// MODE 3: 2x2 outer product of 32x32x16 MFMAs (same 64x64 wave tile, half the operand registers read per FLOP)
// MODE 4: the same with one fixed A/B register pair (upper bound)
template <int MODE>
__global__ void __launch_bounds__(256) peak32(float* out, const float* in, int iters) {
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  bf16x8 fa[4], fb[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) { fa[i][e] = (__bf16)in[(threadIdx.x + i * 8 + e) & 1023]; fb[i][e] = (__bf16)in[(threadIdx.x * 3 + i * 8 + e) & 1023]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)       // K = 32 per iteration, like 16 of the 16x16x32 MFMAs
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = MODE == 3 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks * 2 + i], fb[ks * 2 + j], acc[i][j], 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float *out, *in; hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode)
    for (int blocks : {256, 512, 1024}) {
      const int iters = 8000;
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(peak<0>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
        else if (mode == 1) hipLaunchKernelGGL(peak<1>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
        else hipLaunchKernelGGL(peak<2>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      double flops = (double)blocks * 4 * iters * 16 * 16384.0;
      printf("mode %d (%s) blocks=%d (%.0f waves/SIMD): %.3f ms  %.1f TFLOP/s\n", mode,
             mode == 0 ? "same A/B registers" : (mode == 1 ? "4x4 outer product, fixed fragments" : "4x4 outer product, alternating fragments"),
             blocks, blocks / 256.0, ms, flops / ms / 1e9);
    }
  for (int mode = 3; mode < 5; ++mode)
    for (int blocks : {256, 512, 1024}) {
      const int iters = 8000;
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 3) hipLaunchKernelGGL(peak32<3>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
        else hipLaunchKernelGGL(peak32<4>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      double flops = (double)blocks * 4 * iters * 8 * 32768.0;
      printf("mode %d (32x32x16 %s) blocks=%d (%.0f waves/SIMD): %.3f ms  %.1f TFLOP/s\n", mode,
             mode == 3 ? "2x2 outer product, distinct fragments" : "same A/B registers", blocks, blocks / 256.0, ms, flops / ms / 1e9);
    }
  return 0;
}
