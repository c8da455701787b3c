// Fused multi-tensor optimizer kernels over flat fp32 arenas (K17/K18/K18b of SURVEY.md §2.5).
// Reference behaviour: torch.optim.Adam(lr=1e-3) / SGD(lr=1e-3, momentum=0.9) as selected at
// /root/reference/classif.py:123-128.  All parameters live in ONE contiguous arena, so a whole model
// (or one gradient bucket of it) is updated by a single launch: 16 B/lane vector access, grid-stride.
// Hyper-parameters and the step counter live in device memory so that a captured hipGraph replays
// correctly when lr (StepLR) or the step changes.
#include "common.h"

// hyper layout (12 floats): [0]=lr [1]=beta1 [2]=beta2 [3]=eps [4]=weight_decay [5]=step (as float)
//                           [6]=momentum [7]=grad_scale (1/world_size folded in here)
//                           [8]=float(1-beta1) [9]=float(1-beta2) [10]=beta1-(double)float(beta1) [11]=same for beta2
// The last four exist so that the kernel can reproduce torch.optim.Adam's scalar handling to the ulp (nn/optim.py:
// make_hyper): torch keeps the betas as Python doubles - the increments use float(1 - beta) (0.001f, not 1 - 0.999f =
// 0.00099998713), the bias corrections 1 - beta**step are evaluated in double - and only the per-element arithmetic is
// fp32.  With everything derived from float(beta2) instead, the update differed from torch's by 1e-5 relative
// (27 fp32 ulps on a parameter of the update's own size; tests/test_gpu_ops.py::test_optimizers_match_torch now bounds
// it at 2).  Built without -ffast-math (csrc/build.py).
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   bf16_t* __restrict__ p16, const float* __restrict__ hyper,
                                                   long long n) {
  const float lr = hyper[0], b2 = hyper[2], eps = hyper[3], wd = hyper[4];
  const float gs = hyper[7], omb1 = hyper[8], omb2 = hyper[9];
  const double step = (double)hyper[5] + 1.0;
  const double b1d = (double)hyper[1] + (double)hyper[10], b2d = (double)hyper[2] + (double)hyper[11];
  const double bc1 = 1.0 - pow(b1d, step);
  const float bc2s = (float)sqrt(1.0 - pow(b2d, step));
  const float step_size = (float)((double)lr / bc1);
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long long stride = (long long)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) {
    float4 pp = *reinterpret_cast<float4*>(p + i);
    float4 gg = *reinterpret_cast<const float4*>(g + i);
    float4 mm = *reinterpret_cast<float4*>(m + i);
    float4 vv = *reinterpret_cast<float4*>(v + i);
    float* pa = (float*)&pp; float* ga = (float*)&gg; float* ma = (float*)&mm; float* va = (float*)&vv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gr = ga[k] * gs + wd * pa[k];
      ma[k] = ma[k] + omb1 * (gr - ma[k]);            // torch: exp_avg.lerp_(grad, 1 - beta1)
      va[k] = b2 * va[k] + omb2 * gr * gr;            // torch: exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
      float denom = sqrtf(va[k]) / bc2s + eps;
      pa[k] -= step_size * (ma[k] / denom);           // torch: param.addcdiv_(exp_avg, denom, value=-step_size)
    }
    *reinterpret_cast<float4*>(p + i) = pp;
    *reinterpret_cast<float4*>(m + i) = mm;
    *reinterpret_cast<float4*>(v + i) = vv;
    if (p16) {
      uint2 o;
      o.x = (uint32_t)f2bf(pa[0]) | ((uint32_t)f2bf(pa[1]) << 16);
      o.y = (uint32_t)f2bf(pa[2]) | ((uint32_t)f2bf(pa[3]) << 16);
      *reinterpret_cast<uint2*>(p16 + i) = o;
    }
  }
  // tail (n not multiple of 4): handled by the thread whose i lands on it
  if (i < n) {
    for (long long j = i; j < n; ++j) {
      float gr = g[j] * gs + wd * p[j];
      float mj = m[j] + omb1 * (gr - m[j]);
      float vj = b2 * v[j] + omb2 * gr * gr;
      m[j] = mj; v[j] = vj;
      float pj = p[j] - step_size * (mj / (sqrtf(vj) / bc2s + eps));
      p[j] = pj;
      if (p16) p16[j] = f2bf(pj);
    }
  }
}

// torch SGD: buf = g on the first step, then buf = mu*buf + g ; p -= lr*buf  (dampening 0, no nesterov)
__global__ void __launch_bounds__(256) sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ buf, bf16_t* __restrict__ p16,
                                                  const float* __restrict__ hyper, long long n) {
  const float lr = hyper[0], wd = hyper[4], mu = hyper[6], gs = hyper[7];
  const bool first = hyper[5] < 0.5f;
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;      // 16 B per lane, like adam_kernel
  const long long stride = (long long)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) {
    float4 pp = *reinterpret_cast<float4*>(p + i);
    const float4 gg = *reinterpret_cast<const float4*>(g + i);
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mu != 0.0f && !first) bb = *reinterpret_cast<float4*>(buf + i);
    float* pa = (float*)&pp; const float* ga = (const float*)&gg; float* ba = (float*)&bb;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = ga[k] * gs + wd * pa[k];
      const float b = (mu != 0.0f && !first) ? mu * ba[k] + gr : gr;
      ba[k] = b;
      pa[k] -= lr * b;
    }
    *reinterpret_cast<float4*>(p + i) = pp;
    if (mu != 0.0f) *reinterpret_cast<float4*>(buf + i) = bb;
    if (p16) {
      uint2 o;
      o.x = (uint32_t)f2bf(pa[0]) | ((uint32_t)f2bf(pa[1]) << 16);
      o.y = (uint32_t)f2bf(pa[2]) | ((uint32_t)f2bf(pa[3]) << 16);
      *reinterpret_cast<uint2*>(p16 + i) = o;
    }
  }
  if (i < n) {                                                                // tail (n not a multiple of 4)
    for (long long j = i; j < n; ++j) {
      const float gr = g[j] * gs + wd * p[j];
      float b = gr;
      if (mu != 0.0f) { b = first ? gr : mu * buf[j] + gr; buf[j] = b; }
      const float pj = p[j] - lr * b;
      p[j] = pj;
      if (p16) p16[j] = f2bf(pj);
    }
  }
}

__global__ void advance_step_kernel(float* hyper) { hyper[5] += 1.0f; }

__global__ void __launch_bounds__(256) cast_f32_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) d[i] = f2bf(s[i]);
}
__global__ void __launch_bounds__(256) cast_bf16_f32_kernel(const bf16_t* __restrict__ s, float* __restrict__ d, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) d[i] = bf2f(s[i]);
}

DPA_API int dpa_adam_step(float* p, const float* g, float* m, float* v, bf16_t* p16, const float* hyper,
                           long long n, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(adam_kernel, dim3(dpa_grid((n + 3) / 4, 256)), dim3(256), 0, s, p, g, m, v, p16, hyper, n);
  return dpa_launch_rc();
}
DPA_API int dpa_sgd_step(float* p, const float* g, float* buf, bf16_t* p16, const float* hyper, long long n,
                          hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(sgd_kernel, dim3(dpa_grid((n + 3) / 4, 256)), dim3(256), 0, s, p, g, buf, p16, hyper, n);
  return dpa_launch_rc();
}
DPA_API int dpa_advance_step(float* hyper, hipStream_t s) {
  hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(1), 0, s, hyper);
  return dpa_launch_rc();
}
DPA_API int dpa_cast_f32_bf16(const float* src, bf16_t* dst, long long n, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(dpa_grid(n, 256)), dim3(256), 0, s, src, dst, n);
  return dpa_launch_rc();
}
DPA_API int dpa_cast_bf16_f32(const bf16_t* src, float* dst, long long n, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(dpa_grid(n, 256)), dim3(256), 0, s, src, dst, n);
  return dpa_launch_rc();
}
