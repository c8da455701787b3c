// Fused log-softmax + NLL loss + accuracy (+ gradient) for gfx950 (K14/K14b/K15/K16 of SURVEY.md §2.5).
// Reference behaviour: nn.CrossEntropyLoss / weighted CrossEntropyLoss / FocalLossN (/root/reference/classif.py:106-117,
// utils.py:142-156) followed by calculateAccuracy (utils.py:158-162) and two .item() syncs per iteration
// (classif.py:61-62).  Here ONE launch produces loss, accuracy, d(loss)/d(logits) and adds loss/acc into
// device-resident epoch accumulators, so the hot loop never synchronises with the host.
//   mode 0: mean_n( -log p[y] )                       mode 1: sum_n w[y](-log p[y]) / sum_n w[y]
//   mode 2: mean_n( -w[y] (1-p[y])^gamma log p[y] )   (FocalLossN: plain mean over N, utils.py:153-155)
#include "common.h"
#include <map>
#include <mutex>
#include <utility>

template <typename T>
__global__ void __launch_bounds__(1024) ce_loss_kernel(const T* __restrict__ logits, int ld, const long long* __restrict__ labels,
                                                       int N, int C, const float* __restrict__ cw, int mode, float gamma,
                                                       float scale, int aux, T* __restrict__ dlogits, float* __restrict__ out,
                                                       float* __restrict__ accum) {
  __shared__ float s_loss[16], s_w[16], s_corr[16];
  __shared__ float s_tot[3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float loss_acc = 0.f, w_acc = 0.f, corr = 0.f;
  for (int n = wave; n < N; n += nw) {
    const T* row = logits + (size_t)n * ld;
    const int y = (int)labels[n];
    float mx = -INFINITY; int arg = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
      const float v = Elem<T>::ld(row + c);
      if (v > mx || (v == mx && c < arg)) { mx = v; arg = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(mx, o, 64); const int oa = __shfl_xor(arg, o, 64);
      if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
    }
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += __expf(Elem<T>::ld(row + c) - mx);
    se = wave_sum(se);
    const float zy = Elem<T>::ld(row + y);
    const float logp = zy - mx - __logf(se);
    const float w = cw ? cw[y] : 1.f;
    float l;
    if (mode == 2) { const float p = __expf(logp); l = -w * powf(fmaxf(1.f - p, 0.f), gamma) * logp; }
    else l = -w * logp;
    loss_acc += l; w_acc += w; corr += (arg == y) ? 1.f : 0.f;
  }
  if (lane == 0) { s_loss[wave] = loss_acc; s_w[wave] = w_acc; s_corr[wave] = corr; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float L = 0.f, W = 0.f, Cc = 0.f;
    for (int i = 0; i < nw; ++i) { L += s_loss[i]; W += s_w[i]; Cc += s_corr[i]; }
    const float denom = (mode == 1) ? W : (float)N;
    s_tot[0] = L / denom; s_tot[1] = denom; s_tot[2] = Cc / (float)N;
    const float Ls = scale * s_tot[0];
    if (aux) { out[0] += Ls; if (accum) accum[0] += Ls; }
    else {
      out[0] = Ls; out[1] = s_tot[2];
      if (accum) { accum[0] += Ls; accum[1] += s_tot[2]; accum[2] += 1.f; }
    }
  }
  if (!dlogits) return;
  __syncthreads();
  const float inv = scale / s_tot[1];
  for (int n = wave; n < N; n += nw) {
    const T* row = logits + (size_t)n * ld;
    T* drow = dlogits + (size_t)n * ld;
    const int y = (int)labels[n];
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, Elem<T>::ld(row + c));
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += __expf(Elem<T>::ld(row + c) - mx);
    se = wave_sum(se);
    const float rse = 1.f / se;
    const float w = cw ? cw[y] : 1.f;
    float f = w;
    if (mode == 2) {
      const float py = __expf(Elem<T>::ld(row + y) - mx) * rse;
      const float om = fmaxf(1.f - py, 0.f);
      f = w * (powf(om, gamma) - gamma * powf(om, gamma - 1.f) * py * __logf(fmaxf(py, 1e-37f)));
    }
    for (int c = lane; c < ld; c += 64) {
      float g = 0.f;
      if (c < C) { const float p = __expf(Elem<T>::ld(row + c) - mx) * rse; g = f * inv * (p - (c == y ? 1.f : 0.f)); }
      Elem<T>::st(drow + c, g);
    }
  }
}

// ---- multi-block variant for large batches (ImageNet-shaped heads: 1024 x 1000 logits took 1.4 ms in one block) -------
// Phase 1: one wave per row computes loss / weight / correct AND (modes 0 and 2, whose denominator N is known up front)
// the gradient; per-block partial sums go to a small workspace.  Phase 2: one block adds the partials in a fixed order
// and updates the outputs / device accumulators exactly like the single-block kernel.  Mode 1 (weighted CE: the
// denominator is the sum of the weights) keeps the single-block kernel.
template <typename T>
__global__ void __launch_bounds__(1024) ce_rows_kernel(const T* __restrict__ logits, int ld, const long long* __restrict__ labels,
                                                       int N, int C, const float* __restrict__ cw, int mode, float gamma,
                                                       float scale, T* __restrict__ dlogits, float* __restrict__ partials) {
  __shared__ float s_loss[16], s_corr[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 16 + wave;
  float l = 0.f, corr = 0.f;
  if (n < N) {
    const T* row = logits + (size_t)n * ld;
    const int y = (int)labels[n];
    float mx = -INFINITY; int arg = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
      const float v = Elem<T>::ld(row + c);
      if (v > mx || (v == mx && c < arg)) { mx = v; arg = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(mx, o, 64); const int oa = __shfl_xor(arg, o, 64);
      if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
    }
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += __expf(Elem<T>::ld(row + c) - mx);
    se = wave_sum(se);
    const float zy = Elem<T>::ld(row + y);
    const float logp = zy - mx - __logf(se);
    const float w = cw ? cw[y] : 1.f;
    float f = w;
    if (mode == 2) {
      const float p = __expf(logp);
      const float om = fmaxf(1.f - p, 0.f);
      l = -w * powf(om, gamma) * logp;
      f = w * (powf(om, gamma) - gamma * powf(om, gamma - 1.f) * p * __logf(fmaxf(p, 1e-37f)));
    } else {
      l = -w * logp;
    }
    corr = (arg == y) ? 1.f : 0.f;
    if (dlogits) {
      T* drow = dlogits + (size_t)n * ld;
      const float rse = 1.f / se, inv = scale / (float)N;
      for (int c = lane; c < ld; c += 64) {
        float g = 0.f;
        if (c < C) { const float p = __expf(Elem<T>::ld(row + c) - mx) * rse; g = f * inv * (p - (c == y ? 1.f : 0.f)); }
        Elem<T>::st(drow + c, g);
      }
    }
  }
  if (lane == 0) { s_loss[wave] = l; s_corr[wave] = corr; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float L = 0.f, Cc = 0.f;
    for (int i = 0; i < 16; ++i) { L += s_loss[i]; Cc += s_corr[i]; }
    partials[2 * blockIdx.x] = L; partials[2 * blockIdx.x + 1] = Cc;
  }
}

__global__ void __launch_bounds__(64) ce_finish_kernel(const float* __restrict__ partials, int nblocks, int N, float scale, int aux,
                                                       float* __restrict__ out, float* __restrict__ accum) {
  if (threadIdx.x != 0) return;
  float L = 0.f, Cc = 0.f;
  for (int i = 0; i < nblocks; ++i) { L += partials[2 * i]; Cc += partials[2 * i + 1]; }
  const float Ls = scale * L / (float)N, acc = Cc / (float)N;
  if (aux) { out[0] += Ls; if (accum) accum[0] += Ls; }
  else {
    out[0] = Ls; out[1] = acc;
    if (accum) { accum[0] += Ls; accum[1] += acc; accum[2] += 1.f; }
  }
}

// Per-device scratch for the multi-block variant.  Allocated by dpa_init_workspaces() when the library is loaded (never
// inside a launch path that a hipGraph capture may be recording); a batch that outgrows it while the stream is being
// captured takes the single-block kernel instead of allocating.
static std::mutex ce_mu;
static std::map<int, std::pair<float*, int>> ce_ws;
static float* ce_partials(int need_floats, hipStream_t s) {
  std::lock_guard<std::mutex> lk(ce_mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  auto& e = ce_ws[dev];
  if (e.second < need_floats) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (s && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return nullptr;
    float* ptr = nullptr;
    const int cap = need_floats < 16384 ? 16384 : need_floats;
    if (hipMalloc((void**)&ptr, (size_t)cap * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    e = {ptr, cap};         // (an outgrown buffer is left alone: earlier launches may still read it)
  }
  return e.first;
}
extern "C" int dpa_init_ce_workspace() { return ce_partials(16384, nullptr) ? 0 : 1; }

DPA_API int dpa_ce_loss(const void* logits, int ld, const long long* labels, int N, int C, const float* class_w, int mode,
                         float gamma, float scale, int aux, void* dlogits, float* out, float* accum, int dtype, hipStream_t s) {
  if (mode != 1 && (long long)N * C >= (1 << 16) && N >= 64) {
    const int nb = (N + 15) / 16;
    float* part = ce_partials(2 * nb, s);
    if (part) {
      if (dtype == 1) hipLaunchKernelGGL(ce_rows_kernel<bf16_t>, dim3(nb), dim3(1024), 0, s, (const bf16_t*)logits, ld, labels, N, C,
                                         class_w, mode, gamma, scale, (bf16_t*)dlogits, part);
      else            hipLaunchKernelGGL(ce_rows_kernel<float>, dim3(nb), dim3(1024), 0, s, (const float*)logits, ld, labels, N, C,
                                         class_w, mode, gamma, scale, (float*)dlogits, part);
      hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(64), 0, s, part, nb, N, scale, aux, out, accum);
      return dpa_launch_rc();
    }
  }
  if (dtype == 1) hipLaunchKernelGGL(ce_loss_kernel<bf16_t>, dim3(1), dim3(1024), 0, s, (const bf16_t*)logits, ld, labels, N, C,
                                     class_w, mode, gamma, scale, aux, (bf16_t*)dlogits, out, accum);
  else            hipLaunchKernelGGL(ce_loss_kernel<float>, dim3(1), dim3(1024), 0, s, (const float*)logits, ld, labels, N, C,
                                     class_w, mode, gamma, scale, aux, (float*)dlogits, out, accum);
  return dpa_launch_rc();
}
