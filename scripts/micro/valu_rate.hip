// VALU issue-rate microbenchmark for gfx950: how many cycles does a wave64 spend per instruction, alone and next to
// MFMA work from the same wave?  build: hipcc --offload-arch=gfx950 -O3 scripts/micro/valu_rate.hip -o scripts/micro/bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// MODE 0: v_cvt_pk_bf16_f32   1: v_and_b32   2: v_pk_add_f32   3: v_perm_b32   4: v_add_f32   5: mfma only
// MODE 10+m: VALU op m interleaved with 4 independent bf16 MFMAs per iteration
template <int MODE>
__global__ void __launch_bounds__(256) rate(float* out, int iters) {
  constexpr int OP = MODE % 10;
  constexpr bool WITH_MFMA = MODE >= 10 || OP == 5;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)v[i]; hb[i] = (__bf16)v[i + 8]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      if (OP == 0) {
        f32x2 r = {v[i], v[i + 1]};
        uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
        asm volatile("" : "+v"(h));
        v[i] = __uint_as_float(h);
      } else if (OP == 1) {
        uint32_t a = __float_as_uint(v[i]) & 0xffff0000u, b = __float_as_uint(v[i + 1]) & 0xffff0000u;
        asm volatile("" : "+v"(a), "+v"(b));
        v[i] = __uint_as_float(a); v[i + 1] = __uint_as_float(b);
      } else if (OP == 2) {
        f32x2 r = {v[i], v[i + 1]};
        f32x2 q = {v[(i + 2) & 15], v[(i + 3) & 15]};
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r) : "v"(q));
        v[i] = r.x; v[i + 1] = r.y;
      } else if (OP == 3) {
        uint32_t h = __builtin_amdgcn_perm(__float_as_uint(v[i]), __float_as_uint(v[i + 1]), 0x07060302u);
        asm volatile("" : "+v"(h));
        v[i] = __uint_as_float(h);
      } else if (OP == 4) {
        float a = v[i] + v[i + 1];
        asm volatile("" : "+v"(a));
        v[i] = a;
      }
      if (WITH_MFMA && (i % 4) == 0)
        acc[i / 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[i / 4], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, float* out, int valu_per_iter, int mfma_per_iter) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 512;     // 2 blocks of 4 waves per CU = 2 waves per SIMD
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  // per SIMD: 2 waves, each iters * (valu + mfma) instructions
  const double ns_per_iter_per_wave = ms * 1e6 / iters / 2.0;
  printf("%-34s %.3f ms  %.2f ns per wave-iteration (%d VALU + %d MFMA)\n", name, ms, ns_per_iter_per_wave, valu_per_iter, mfma_per_iter);
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  run<0>("cvt_pk_bf16_f32 x8", out, 8, 0);
  run<1>("and_b32 x16", out, 16, 0);
  run<2>("pk_add_f32 x8", out, 8, 0);
  run<3>("perm_b32 x8", out, 8, 0);
  run<4>("add_f32 x8", out, 8, 0);
  run<5>("mfma 16x16x32 bf16 x4", out, 0, 4);
  run<10>("cvt_pk x8 + mfma x4", out, 8, 4);
  run<11>("and x16 + mfma x4", out, 16, 4);
  run<12>("pk_add x8 + mfma x4", out, 8, 4);
  run<14>("add_f32 x8 + mfma x4", out, 8, 4);
  return 0;
}
