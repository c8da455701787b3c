// LDS bank-conflict probe for the fragment read patterns of conv_x6p.hip (run under rocprofv3 --pmc
// SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS): each kernel issues only ONE read pattern, so the counters tell
// which pattern conflicts and which candidate layout does not.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/bin/lds_conflict scripts/micro/lds_conflict.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// MODE 0: A fragments (fp32 rows of 128 B, chunk ^ ((row >> 1) & 7))
// MODE 1: B plane fragments, production layout: rows of 64 B, slot ^ ((-(row >> 2)) & 3)
// MODE 2: B candidate: slot ^ ((row >> 2) & 3)
// MODE 3: B candidate: rows of 64 B, slot rotated: (slot + (row >> 2)) & 3
// MODE 4: B candidate: slot ^ ((row >> 2) & 3) ^ ((row >> 1) & 1) ... (see code)
// MODE 5: B unswizzled (expected: 4-way conflicts)
// MODE 6: A fragments with chunk ^ (((row >> 1) & 1) | (((row >> 2) & 1) << 2))
template <int MODE>
__global__ void __launch_bounds__(256) probe(unsigned* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 40960 / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    const int flip = (it & 1) * 0;      // keeps the address loop-dependent for the compiler without changing it
    if (MODE == 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint4 v = *reinterpret_cast<const uint4*>(smem + flip + lds_off(wave * 32 + i * 16 + fr, 2 * fq + ks));
          acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    } else if (MODE == 7) {        // general kernel (conv_igemm.hip): chunk = 4 ks + fq, chunk ^ ((row >> 1) & 7)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint4 v = *reinterpret_cast<const uint4*>(smem + flip + lds_off(wave * 32 + i * 16 + fr, 4 * ks + fq));
          acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    } else if (MODE == 8 || MODE == 9 || MODE == 10) {
      // grouping tests: distinct 16-byte segments exactly within a hypothesised group of 16 lanes, the same banks in every other group
      int off;
      if (MODE == 8) off = ((lane & 7) + 8 * ((lane >> 4) & 1)) * 16 + 256 * ((lane >> 3) & 1) + 512 * (lane >> 5);   // lanes 8m..8m+7 with 8m+16..8m+23
      else if (MODE == 9) off = (lane & 15) * 16 + 256 * (lane >> 4);                                                   // lanes 16g..16g+15
      else off = ((lane & 7) + 8 * (lane >> 5)) * 16 + 256 * ((lane >> 3) & 3);                                         // lanes 8m..8m+7 with 8m+32..8m+39
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint4 v = *reinterpret_cast<const uint4*>(smem + flip + off + r * 1024 + wave * 4096);
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
      }
    } else if (MODE == 6) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = wave * 32 + i * 16 + fr;
          const int off = row * 128 + (((2 * fq + ks) ^ (((row >> 1) & 1) | (((row >> 2) & 1) << 2))) << 4);
          const uint4 v = *reinterpret_cast<const uint4*>(smem + flip + off);
          acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    } else {
      int slot;
      if (MODE == 1) slot = fq ^ ((-(fr >> 2)) & 3);
      else if (MODE == 2) slot = fq ^ ((fr >> 2) & 3);
      else if (MODE == 3) slot = (fq + (fr >> 2)) & 3;
      else if (MODE == 4) slot = fq ^ ((fr >> 2) & 3) ^ (((fr >> 2) & 1) << 1);
      else slot = fq;
      const int b_frag = fr * 64 + (slot << 4);
#pragma unroll
      for (int part = 0; part < 3; ++part)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint4 v = *reinterpret_cast<const uint4*>(smem + flip + 16384 + part * 8192 + j * 1024 + b_frag);
          acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    asm volatile("" ::: "memory");
  }
  out[blockIdx.x * 256 + tid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

int main() {
  unsigned* out;
  hipMalloc(&out, 256 * 256 * 4);
  const int iters = 2000;
  hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 40960, 0, out, iters);
  hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 40960, 0, out, iters);
  hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 40960, 0, out, iters);
  hipLaunchKernelGGL(probe<3>, dim3(256), dim3(256), 40960, 0, out, iters);
  hipLaunchKernelGGL(probe<4>, dim3(256), dim3(256), 40960, 0, out, iters);
  hipLaunchKernelGGL(probe<5>, dim3(256), dim3(256), 40960, 0, out, iters);
  hipLaunchKernelGGL(probe<6>, dim3(256), dim3(256), 40960, 0, out, iters);
  hipLaunchKernelGGL(probe<7>, dim3(256), dim3(256), 40960, 0, out, iters);
  hipLaunchKernelGGL(probe<8>, dim3(256), dim3(256), 40960, 0, out, iters);
  hipLaunchKernelGGL(probe<9>, dim3(256), dim3(256), 40960, 0, out, iters);
  hipLaunchKernelGGL(probe<10>, dim3(256), dim3(256), 40960, 0, out, iters);
  hipDeviceSynchronize();
  printf("done %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
