// MFMA peak microbenchmark for gfx950: register-only MFMA chains, no memory traffic.
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int MODE>
__global__ void __launch_bounds__(256) peak(float* out, int iters) {
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  bf16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)a; hb[i] = (__bf16)b; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      else           acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int blocks : {256, 512, 1024, 2048}) {
      int iters = mode == 0 ? 4000 : 8000;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(peak<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else           hipLaunchKernelGGL(peak<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 * iters * 16 * (mode == 0 ? 2048.0 : 16384.0);
        if (rep) printf("%s blocks=%d (%.1f waves/SIMD): %.3f ms  %.1f TFLOP/s\n", mode == 0 ? "f32 16x16x4 " : "bf16 16x16x32",
                        blocks, blocks / 256.0, ms, flops / ms / 1e9);
      }
    }
  return 0;
}
