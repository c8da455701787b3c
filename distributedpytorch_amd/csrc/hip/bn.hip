// BatchNorm2d for NHWC activations on gfx950 (K5/K6/K7/K11 of SURVEY.md §2.5; the reference reaches these
// through torchvision's BatchNorm2d layers, /root/reference/utils.py:45-91).
//
// Training forward is split so that the statistics pass costs no extra read of the activation:
//   conv epilogue  -> per-block (sum, sumsq) slab        (conv_igemm.hip)       [or bn_stats for non-conv inputs]
//   bn_finalize    -> mean / invstd / running stats / folded (scale, shift)     (tiny)
//   bn_apply       -> y = relu?(x*scale + shift (+ residual))                   (one read, one write, 16 B/lane)
// Backward:
//   bn_bwd_reduce  -> dz = dy * (y > 0) written IN PLACE over dy, slab of (sum dz, sum dz*xhat)
//   bn_bwd_finalize-> dgamma, dbeta
//   bn_bwd_dx      -> dx = gamma*invstd*(dz - dbeta/M - xhat*dgamma/M)
// All slabs are reduced in a fixed order: results are bitwise reproducible (no float atomics).
#include "common.h"
#include <type_traits>

#define ROWS_L2 256  // rows of the second-level slab (= blocks along the row axis of slab_reduce: 32 left most CUs idle)

// MASK (bias gradient of conv + bias + ReLU): x is a gradient that is first masked IN PLACE with y > 0 - the ReLU backward
// and the column sum in one pass over the tensor instead of two kernels.
template <typename T, bool MASK = false>
__global__ void __launch_bounds__(256) bn_stats_kernel(T* __restrict__ x, long long M, int C, int cs,
                                                       float* __restrict__ slab, int rowsPerBlock,
                                                       const float* __restrict__ shift,
                                                       const T* __restrict__ y = nullptr, int y_cs = 0) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red[2][256][VEC > 4 ? 9 : 5];
  const int cpr = C / VEC;
  const int colsPerPass = cpr < 256 ? cpr : 256;
  const int rowLanes = 256 / colsPerPass;
  const int tid = threadIdx.x;
  const int col = tid % colsPerPass, rl = tid / colsPerPass;
  const long long r0 = (long long)blockIdx.x * rowsPerBlock;
  long long r1 = r0 + rowsPerBlock; if (r1 > M) r1 = M;
  for (int cb = 0; cb < cpr; cb += colsPerPass) {
    float s[VEC], q[VEC], c0[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s[e] = 0.f; q[e] = 0.f; c0[e] = 0.f; }
    const int cc = cb + col;
    if (cc < cpr && rl < rowLanes) {
      if (shift) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) c0[e] = shift[cc * VEC + e];
      }
      for (long long r = r0 + rl; r < r1; r += rowLanes) {
        Vec16<T> v; v.load(x + r * cs + (long long)cc * VEC);
        if constexpr (MASK) {
          Vec16<T> yv; yv.load(y + r * y_cs + (long long)cc * VEC);
#pragma unroll
          for (int e = 0; e < VEC; ++e) if (!(yv.get(e) > 0.f)) v.set(e, 0.f);
          v.store(x + r * cs + (long long)cc * VEC);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) { const float f = v.get(e) - c0[e]; s[e] += f; q[e] += f * f; }
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { red[0][tid][e] = s[e]; red[1][tid][e] = q[e]; }
    __syncthreads();
    if (rl == 0 && cc < cpr) {
      for (int l = 1; l < rowLanes; ++l)
#pragma unroll
        for (int e = 0; e < VEC; ++e) { s[e] += red[0][l * colsPerPass + col][e]; q[e] += red[1][l * colsPerPass + col][e]; }
      float* o = slab + (size_t)blockIdx.x * 2 * C + (size_t)cc * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e) { o[e] = s[e]; o[C + e] = q[e]; }
    }
    __syncthreads();
  }
}

// [rows][width] -> [groups][width]; out row j sums rows j, j + groups, ... in a FIXED order: a block owns 64 float4
// columns and four row lanes (lane l takes every fourth of its group's rows), combined lane 0..3 through LDS.
// (Round 3: one thread per column walked rows / groups = 196 rows one dependent load after the other - 32 us per call,
// 90 calls and 2.9 ms per ResNet-50 step at batch 1024 for 1 GB of traffic.  Now 24 independent 16-byte loads per thread.)
__global__ void __launch_bounds__(256) slab_reduce_kernel(const float* __restrict__ in, int rows, int width,
                                                          float* __restrict__ out, int groups) {
  __shared__ float4 red[4][64];
  const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = (blockIdx.y * 64 + cq) * 4;          // width % 4 == 0 (host)
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < width) {
#pragma unroll 4
    for (int r = blockIdx.x + rl * groups; r < rows; r += 4 * groups) {
      const float4 t = *reinterpret_cast<const float4*>(in + (size_t)r * width + c);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
  }
  red[rl][cq] = s;
  __syncthreads();
  if (rl == 0 && c < width) {
#pragma unroll
    for (int l = 1; l < 4; ++l) { const float4 t = red[l][cq]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    *reinterpret_cast<float4*>(out + (size_t)blockIdx.x * width + c) = s;
  }
}
// (width % 4 != 0: one thread per column)
__global__ void __launch_bounds__(256) slab_reduce_scalar_kernel(const float* __restrict__ in, int rows, int width,
                                                                 float* __restrict__ out, int groups) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= width) return;
  float s = 0.f;
  for (int r = blockIdx.x; r < rows; r += groups) s += in[(size_t)r * width + c];
  out[(size_t)blockIdx.x * width + c] = s;
}

// Column sums of a slab [rows][2][C] for the finalize kernels: a block is 64 channels x FIN_RL row lanes (lane l sums rows
// l, l + FIN_RL, ... - independent loads, 1/16 of the serial depth of one thread per channel), then lane 0 combines the
// lanes in a fixed order.  Returns true in the threads that own a channel (and hold its sums).  With it a slab of up to
// FIN_MAX_ROWS rows is finalized directly: no slab_reduce launch in front (batch 64: most BatchNorm layers of the zoo).
constexpr int FIN_RL = 16, FIN_T = 64 * FIN_RL, FIN_MAX_ROWS = 512;
__device__ __forceinline__ bool slab_sum2(const float* __restrict__ slab, int rows, int C, bool want_q, float& s, float& q) {
  __shared__ float red[2][FIN_RL][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a = 0.f, b = 0.f;
  if (c < C) {
#pragma unroll 4
    for (int r = rl; r < rows; r += FIN_RL) {
      a += slab[((size_t)r * 2) * C + c];
      if (want_q) b += slab[((size_t)r * 2 + 1) * C + c];
    }
  }
  red[0][rl][cl] = a; red[1][rl][cl] = b;
  __syncthreads();
  if (rl != 0 || c >= C) return false;
  s = 0.f; q = 0.f;
#pragma unroll
  for (int l = 0; l < FIN_RL; ++l) { s += red[0][l][cl]; q += red[1][l][cl]; }
  return true;
}

// slab [rows][2][C] -> batch statistics and the folded (scale, shift) of the normalisation
__global__ void __launch_bounds__(FIN_T) bn_finalize_kernel(const float* __restrict__ slab, int rows, int C, float count,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                                         long long* __restrict__ nbt, float momentum, float eps,
                                                         float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                         float* __restrict__ scale_out, float* __restrict__ shift_out,
                                                         float* __restrict__ stat_shift) {
  // Shifted-data variance: the slab holds sum(x - c) and sum((x - c)^2) about a per-channel shift c (the previous
  // step's batch mean, kept in `stat_shift` and refreshed here), so var = E[(x-c)^2] - (E[x-c])^2 subtracts two SMALL
  // numbers even when |mean| >> std (post-ReLU inputs of DenseNet's pre-activation BN, the inception stem) - the
  // plain E[x^2] - mean^2 in fp32 loses every digit of the variance there (mean 100, std 0.1: 10000.01 - 10000).
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  float s, q;
  if (!slab_sum2(slab, rows, C, true, s, q)) return;
  const float c0 = stat_shift ? stat_shift[c] : 0.f;
  const float dm = s / count;
  const float mean = c0 + dm;
  float var = q / count - dm * dm;
  if (var < 0.f) var = 0.f;
  if (stat_shift) stat_shift[c] = mean;
  const float invstd = rsqrtf(var + eps);
  mean_out[c] = mean; invstd_out[c] = invstd;
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale_out[c] = g * invstd;
  shift_out[c] = b - mean * g * invstd;
  if (running_mean) {
    const float unbiased = count > 1.f ? var * count / (count - 1.f) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

__global__ void __launch_bounds__(64) bn_eval_coeffs_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                                            float* __restrict__ scale_out, float* __restrict__ shift_out) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  const float invstd = rsqrtf(rv[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale_out[c] = g * invstd;
  shift_out[c] = b - rm[c] * g * invstd;
}

// Column-persistent elementwise layout shared by bn_apply / bn_bwd_dx: a thread owns ONE 16-byte channel chunk and
// walks rows, so the per-channel coefficients are loaded once into registers (the first version re-loaded 2-5 fp32
// vectors per element and was texture-address bound: 410 us vs 53 us for the same bytes, profiles/README.md).
template <typename T>
__global__ void __launch_bounds__(256) bn_apply_kernel(const T* __restrict__ x, int x_cs, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const T* __restrict__ res, int r_cs,
                                                       T* __restrict__ y, int y_cs, long long M, int C, int relu,
                                                       int rowsPerBlock, uint32_t* __restrict__ bits,
                                                       const float* __restrict__ rscale, const float* __restrict__ rshift) {
  // rscale / rshift (or null): the residual is the INPUT of a second BatchNorm (the downsample branch of a ResNet block,
  // ConvBNAct(defer_apply=True)): res * rscale + rshift is added - that BatchNorm's output never goes through memory.
  // bits (or null): [M][C / 32] words, bit b of word w = (y[row][32 w + b] > 0) - the ReLU mask a later fused dgrad
  // epilogue reads instead of y itself (1 / 32 of the bytes).  A thread owns VEC channels; the 32 / VEC adjacent lanes of a
  // word OR their bits together with shuffles (C % 32 == 0: a word's lanes are always active together).
  constexpr int VEC = 16 / sizeof(T);
  constexpr int LPW = 32 / VEC;                    // lanes per mask word
  auto put_bits = [&](long long m, int c, uint32_t mine) __attribute__((always_inline)) {
    uint32_t w = mine << ((c & 31));
#pragma unroll
    for (int o = 1; o < LPW; o <<= 1) w |= (uint32_t)__shfl_xor((int)w, o, 64);
    if (((c / VEC) & (LPW - 1)) == 0) bits[m * (C >> 5) + (c >> 5)] = w;
  };
  const int cpr = C / VEC;
  const int colsPerPass = cpr < 256 ? cpr : 256;
  const int rowLanes = 256 / colsPerPass;
  const int col = threadIdx.x % colsPerPass, rl = threadIdx.x / colsPerPass;
  if (rl >= rowLanes) return;
  const long long r0 = (long long)blockIdx.x * rowsPerBlock;
  long long r1 = r0 + rowsPerBlock; if (r1 > M) r1 = M;
  for (int cb = col; cb < cpr; cb += colsPerPass) {
    const int c = cb * VEC;
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e += 4) {
      const float4 a = *reinterpret_cast<const float4*>(scale + c + e);
      const float4 b = *reinterpret_cast<const float4*>(shift + c + e);
      sc[e] = a.x; sc[e + 1] = a.y; sc[e + 2] = a.z; sc[e + 3] = a.w;
      sh[e] = b.x; sh[e + 1] = b.y; sh[e + 2] = b.z; sh[e + 3] = b.w;
    }
    float rsc[VEC], rsh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { rsc[e] = rscale ? rscale[c + e] : 1.f; rsh[e] = rscale ? rshift[c + e] : 0.f; }
    // UNR rows per trip, every load issued before the first use: a streaming kernel needs the bytes in flight
    // (one row per trip ran at 5.1 TB/s where torch's 4-wide add reaches 6.0: scripts/micro/hbm_bw.py)
    constexpr int UNR = 4;
    long long m = r0 + rl;
    for (; m + (long long)(UNR - 1) * rowLanes < r1; m += (long long)UNR * rowLanes) {
      Vec16<T> v[UNR], rv[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u].load(x + (m + (long long)u * rowLanes) * x_cs + c);
      if (res) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) rv[u].load(res + (m + (long long)u * rowLanes) * r_cs + c);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        Vec16<T> o;
        uint32_t pos = 0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float f = v[u].get(e) * sc[e] + sh[e];
          if (res) f += rscale ? rv[u].get(e) * rsc[e] + rsh[e] : rv[u].get(e);
          if (relu) f = fmaxf(f, 0.f);
          o.set(e, f);
          pos |= (o.get(e) > 0.f ? 1u : 0u) << e;      // (of the value as stored)
        }
        o.store(y + (m + (long long)u * rowLanes) * y_cs + c);
        if (bits) put_bits(m + (long long)u * rowLanes, c, pos);
      }
    }
    for (; m < r1; m += rowLanes) {
      Vec16<T> v; v.load(x + m * x_cs + c);
      Vec16<T> rv; if (res) rv.load(res + m * r_cs + c);
      Vec16<T> o;
      uint32_t pos = 0;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float f = v.get(e) * sc[e] + sh[e];
        if (res) f += rscale ? rv.get(e) * rsc[e] + rsh[e] : rv.get(e);
        if (relu) f = fmaxf(f, 0.f);
        o.set(e, f);
        pos |= (o.get(e) > 0.f ? 1u : 0u) << e;
      }
      o.store(y + m * y_cs + c);
      if (bits) put_bits(m, c, pos);
    }
  }
}

// dz = dy * (y > 0) (in place over dy when `y` given); slab[block][2][C] = (sum dz, sum dz*xhat)
template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(T* __restrict__ dy, int dy_cs, const T* __restrict__ y, int y_cs,
                                                            const T* __restrict__ x, int x_cs, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, long long M, int C,
                                                            float* __restrict__ slab, int rowsPerBlock,
                                                            const float* __restrict__ mscale,
                                                            const float* __restrict__ mshift) {
  // ReLU mask: from `y` (dy is then masked IN PLACE for bn_bwd_dx and the residual branch), or - for BN+ReLU without a
  // residual - recomputed as scale*x+shift > 0 from the x this kernel reads anyway (mscale/mshift = the forward's
  // folded coefficients): no y read and no dy write, 2 of the 7 memory passes of the BN backward.
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red[2][256][VEC > 4 ? 9 : 5];
  const int cpr = C / VEC;
  const int colsPerPass = cpr < 256 ? cpr : 256;
  const int rowLanes = 256 / colsPerPass;
  const int tid = threadIdx.x;
  const int col = tid % colsPerPass, rl = tid / colsPerPass;
  const long long r0 = (long long)blockIdx.x * rowsPerBlock;
  long long r1 = r0 + rowsPerBlock; if (r1 > M) r1 = M;
  for (int cb = 0; cb < cpr; cb += colsPerPass) {
    float s[VEC], q[VEC], mu[VEC], is[VEC], msc[VEC], msh[VEC];
    const int cc = cb + col;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s[e] = 0.f; q[e] = 0.f; mu[e] = 0.f; is[e] = 0.f; msc[e] = 0.f; msh[e] = 1.f; }
    if (cc < cpr && rl < rowLanes) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) { mu[e] = mean[cc * VEC + e]; is[e] = invstd[cc * VEC + e]; }
      if (mscale) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { msc[e] = mscale[cc * VEC + e]; msh[e] = mshift[cc * VEC + e]; }
      }
      for (long long r = r0 + rl; r < r1; r += rowLanes) {
        Vec16<T> g; g.load(dy + r * dy_cs + (long long)cc * VEC);
        Vec16<T> xv; xv.load(x + r * x_cs + (long long)cc * VEC);
        if (mscale) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) if (!(xv.get(e) * msc[e] + msh[e] > 0.f)) g.set(e, 0.f);
        } else if (y) {
          Vec16<T> yv; yv.load(y + r * y_cs + (long long)cc * VEC);
#pragma unroll
          for (int e = 0; e < VEC; ++e) if (!(yv.get(e) > 0.f)) g.set(e, 0.f);
          g.store(dy + r * dy_cs + (long long)cc * VEC);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float d = g.get(e);
          s[e] += d; q[e] += d * (xv.get(e) - mu[e]) * is[e];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { red[0][tid][e] = s[e]; red[1][tid][e] = q[e]; }
    __syncthreads();
    if (rl == 0 && cc < cpr) {
      for (int l = 1; l < rowLanes; ++l)
#pragma unroll
        for (int e = 0; e < VEC; ++e) { s[e] += red[0][l * colsPerPass + col][e]; q[e] += red[1][l * colsPerPass + col][e]; }
      float* o = slab + (size_t)blockIdx.x * 2 * C + (size_t)cc * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e) { o[e] = s[e]; o[C + e] = q[e]; }
    }
    __syncthreads();
  }
}

// slab [rows][2][C] -> dbeta = sum dz, dgamma = sum dz*xhat (written or accumulated into the fp32 grad arena) and the
// folded per-channel coefficients of dx = A*dz + B*x + D  (coef[0..3C) = A | B | D):
//   A = gamma*invstd   B = -gamma*invstd^2*dgamma/M   D = -A*dbeta/M - B*mean
__global__ void __launch_bounds__(FIN_T) bn_bwd_finalize_kernel(const float* __restrict__ slab, int rows, int C,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ sums, int accumulate,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, float invM,
                                                             float* __restrict__ coef) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  float s, q;
  if (!slab_sum2(slab, rows, C, true, s, q)) return;
  sums[c] = s; sums[C + c] = q;
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + q : q;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + s : s;
  const float g = gamma ? gamma[c] : 1.f, is = invstd[c];
  const float A = g * is, B = -g * is * is * q * invM;
  coef[c] = A; coef[C + c] = B; coef[2 * C + c] = -A * s * invM - B * mean[c];
}

// ACC: dx += (instead of =): the gradient of a concat prefix that later layers already contributed to (DenseNet) - saves
// the temporary and the separate add kernel.
template <typename T, bool ACC = false>
__global__ void __launch_bounds__(256) bn_bwd_dx_kernel(const T* __restrict__ dz, int dz_cs, const T* x, int x_cs,
                                                        const float* __restrict__ coef, T* dx, int dx_cs,
                                                        long long M, int C, int rowsPerBlock,
                                                        const float* __restrict__ mscale,
                                                        const float* __restrict__ mshift) {
  constexpr int VEC = 16 / sizeof(T);
  const int cpr = C / VEC;
  const int colsPerPass = cpr < 256 ? cpr : 256;
  const int rowLanes = 256 / colsPerPass;
  const int col = threadIdx.x % colsPerPass, rl = threadIdx.x / colsPerPass;
  if (rl >= rowLanes) return;
  const long long r0 = (long long)blockIdx.x * rowsPerBlock;
  long long r1 = r0 + rowsPerBlock; if (r1 > M) r1 = M;
  for (int cb = col; cb < cpr; cb += colsPerPass) {
    const int c = cb * VEC;
    float A[VEC], B[VEC], D[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e += 4) {
      const float4 a = *reinterpret_cast<const float4*>(coef + c + e);
      const float4 b = *reinterpret_cast<const float4*>(coef + C + c + e);
      const float4 d = *reinterpret_cast<const float4*>(coef + 2 * C + c + e);
      A[e] = a.x; A[e + 1] = a.y; A[e + 2] = a.z; A[e + 3] = a.w;
      B[e] = b.x; B[e + 1] = b.y; B[e + 2] = b.z; B[e + 3] = b.w;
      D[e] = d.x; D[e + 1] = d.y; D[e + 2] = d.z; D[e + 3] = d.w;
    }
    float msc[VEC], msh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { msc[e] = mscale ? mscale[c + e] : 0.f; msh[e] = mscale ? mshift[c + e] : 1.f; }
    constexpr int UNR = 4;                      // rows per trip, loads first (see bn_apply_kernel)
    long long m = r0 + rl;
    for (; m + (long long)(UNR - 1) * rowLanes < r1; m += (long long)UNR * rowLanes) {
      Vec16<T> g[UNR], xv[UNR], av[UNR];        // dx may alias x: each thread reads its chunks before writing them
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        g[u].load(dz + (m + (long long)u * rowLanes) * dz_cs + c);
        xv[u].load(x + (m + (long long)u * rowLanes) * x_cs + c);
        if (ACC) av[u].load(dx + (m + (long long)u * rowLanes) * dx_cs + c);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float xe = xv[u].get(e);
          const float ge = (xe * msc[e] + msh[e] > 0.f) ? g[u].get(e) : 0.f;   // (no mask: 0*x + 1 > 0 always)
          { const float r = fmaf(A[e], ge, fmaf(B[e], xe, D[e])); o.set(e, ACC ? r + av[u].get(e) : r); }   // (order fixed: = bn_bwd_dx2_kernel)
        }
        o.store(dx + (m + (long long)u * rowLanes) * dx_cs + c);
      }
    }
    for (; m < r1; m += rowLanes) {
      Vec16<T> g; g.load(dz + m * dz_cs + c);
      Vec16<T> xv; xv.load(x + m * x_cs + c);
      Vec16<T> av; if (ACC) av.load(dx + m * dx_cs + c);
      Vec16<T> o;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float xe = xv.get(e);
        const float ge = (xe * msc[e] + msh[e] > 0.f) ? g.get(e) : 0.f;
        { const float r = fmaf(A[e], ge, fmaf(B[e], xe, D[e])); o.set(e, ACC ? r + av.get(e) : r); }
      }
      o.store(dx + m * dx_cs + c);
    }
  }
}

// Two BatchNorms under one gradient (a ResNet block's last BatchNorm and the one of its downsample branch: both take
// d(block output) as it is): dx1 = A1 dz + B1 x1 + D1 and dx2 = A2 dz + B2 x2 + D2 in one pass - dz is read once.  Same
// per-element operations as two bn_bwd_dx_kernel launches without a mask.
template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_dx2_kernel(const T* __restrict__ dz, int dz_cs, const T* x1, int x1_cs,
                                                         const float* __restrict__ coef1, T* dx1, int dx1_cs,
                                                         const T* x2, int x2_cs, const float* __restrict__ coef2, T* dx2,
                                                         int dx2_cs, long long M, int C, int rowsPerBlock) {
  constexpr int VEC = 16 / sizeof(T);
  const int cpr = C / VEC;
  const int colsPerPass = cpr < 256 ? cpr : 256;
  const int rowLanes = 256 / colsPerPass;
  const int col = threadIdx.x % colsPerPass, rl = threadIdx.x / colsPerPass;
  if (rl >= rowLanes) return;
  const long long r0 = (long long)blockIdx.x * rowsPerBlock;
  long long r1 = r0 + rowsPerBlock; if (r1 > M) r1 = M;
  for (int cb = col; cb < cpr; cb += colsPerPass) {
    const int c = cb * VEC;
    float A1[VEC], B1[VEC], D1[VEC], A2[VEC], B2[VEC], D2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      A1[e] = coef1[c + e]; B1[e] = coef1[C + c + e]; D1[e] = coef1[2 * C + c + e];
      A2[e] = coef2[c + e]; B2[e] = coef2[C + c + e]; D2[e] = coef2[2 * C + c + e];
    }
    auto rows = [&](auto unr_tag, long long m) __attribute__((always_inline)) {
      constexpr int U = decltype(unr_tag)::value;
      Vec16<T> g[U], xa[U], xb[U];               // dx may alias x: each thread reads its chunks before writing them
#pragma unroll
      for (int u = 0; u < U; ++u) {
        g[u].load(dz + (m + (long long)u * rowLanes) * dz_cs + c);
        xa[u].load(x1 + (m + (long long)u * rowLanes) * x1_cs + c);
        xb[u].load(x2 + (m + (long long)u * rowLanes) * x2_cs + c);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        Vec16<T> oa, ob;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float ge = g[u].get(e);
          oa.set(e, fmaf(A1[e], ge, fmaf(B1[e], xa[u].get(e), D1[e])));
          ob.set(e, fmaf(A2[e], ge, fmaf(B2[e], xb[u].get(e), D2[e])));
        }
        oa.store(dx1 + (m + (long long)u * rowLanes) * dx1_cs + c);
        ob.store(dx2 + (m + (long long)u * rowLanes) * dx2_cs + c);
      }
    };
    constexpr int UNR = 4;                      // rows per trip, loads first (see bn_apply_kernel)
    long long m = r0 + rl;
    for (; m + (long long)(UNR - 1) * rowLanes < r1; m += (long long)UNR * rowLanes) rows(std::integral_constant<int, UNR>{}, m);
    for (; m < r1; m += rowLanes) rows(std::integral_constant<int, 1>{}, m);
  }
}

// Row blocking of the streaming (elementwise) kernels: a block owns a CONTIGUOUS ~32 KiB of each operand and the grid
// walks memory in block order, so the blocks resident at any time cover one moving window of the tensor - the access
// pattern of torch's elementwise kernels, which stream at 6.0 TB/s here.  (The first version cut M into 2048 long
// ranges, i.e. 2048 concurrent streams 1.6 MB apart: 5.1 TB/s.  scripts/micro/hbm_bw.py, profiles/r2_hbm_bw.txt.)
static int ew_rows_per_block(long long M, int C, int elem_bytes, int* nblocks) {
  static const int target_kb = [] { const char* e = getenv("DPA_BN_EW_KB"); int v = e ? atoi(e) : 0; return v > 0 ? v : 16; }();
  // whole unrolled trips: the kernels move 4 rows per row-lane per trip (UNR in bn_apply_kernel / bn_bwd_dx_kernel)
  const int cpr = C / (16 / elem_bytes);
  const int row_lanes = 256 / (cpr < 256 ? (cpr > 0 ? cpr : 1) : 256);
  const long long unit = 4LL * (row_lanes > 0 ? row_lanes : 1);
  long long trips = ((long long)target_kb << 10) / (unit * C * elem_bytes);
  if (trips < 1) trips = 1;
  long long rpb = unit * trips;
  while ((M + rpb - 1) / rpb > 0x7fffffffLL) rpb *= 2;
  *nblocks = (int)((M + rpb - 1) / rpb);
  return (int)rpb;
}
static int rows_per_block(long long M, int* nblocks) {
  static const int target = [] { const char* e = getenv("DPA_BN_BLOCKS"); int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
  long long rpb = (M + target - 1) / target;
  if (rpb < 8) rpb = 8;
  *nblocks = (int)((M + rpb - 1) / rpb);
  return (int)rpb;
}

// Reduce a [rows][2][C] slab down to <= ROWS_L2 rows if it is tall. Returns the slab to finalize from.
static const float* shrink_slab(const float* slab, int* rows, int C, float* scratch, hipStream_t s) {
  if (*rows <= FIN_MAX_ROWS) return slab;
  // tall slabs (large batches: 25 088 rows at batch 1024) get 128 row groups so that the reduction covers the chip;
  // small ones stay at 32 (the finalize kernel walks the groups serially)
  const int groups = *rows >= 2048 ? ROWS_L2 : 32;
  if ((2 * C) % 4 == 0 && (reinterpret_cast<uintptr_t>(slab) & 15) == 0 && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0)
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(groups, (2 * C / 4 + 63) / 64), dim3(256), 0, s, slab, *rows, 2 * C, scratch, groups);
  else
    hipLaunchKernelGGL(slab_reduce_scalar_kernel, dim3(groups, (2 * C + 255) / 256), dim3(256), 0, s, slab, *rows, 2 * C, scratch, groups);
  *rows = groups;
  return scratch;
}

DPA_API int dpa_bn_slab_rows(long long M) { int nb; rows_per_block(M, &nb); return nb; }

// scratch: >= ROWS_L2*2*C floats.  slab: [rows][2][C]
DPA_API int dpa_bn_stats(const void* x, int dtype, long long M, int C, int cs, float* slab, const float* shift, hipStream_t s) {
  int nb; const int rpb = rows_per_block(M, &nb);
  if (dtype == 1) hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (bf16_t*)x, M, C, cs, slab, rpb, shift, (const bf16_t*)nullptr, 0);
  else            hipLaunchKernelGGL(bn_stats_kernel<float>, dim3(nb), dim3(256), 0, s, (float*)x, M, C, cs, slab, rpb, shift, (const float*)nullptr, 0);
  return dpa_launch_rc();
}
// Same slab, caller-chosen row blocking (the split-K conv path rebuilds the stats its fused epilogue would have written).
DPA_API int dpa_bn_stats_rows(const void* x, int dtype, long long M, int C, int cs, float* slab, int rowsPerBlock,
                               int nblocks, const float* shift, hipStream_t s) {
  if (dtype == 1) hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, dim3(nblocks), dim3(256), 0, s, (bf16_t*)x, M, C, cs, slab, rowsPerBlock, shift, (const bf16_t*)nullptr, 0);
  else            hipLaunchKernelGGL(bn_stats_kernel<float>, dim3(nblocks), dim3(256), 0, s, (float*)x, M, C, cs, slab, rowsPerBlock, shift, (const float*)nullptr, 0);
  return dpa_launch_rc();
}
DPA_API int dpa_bn_finalize(const float* slab, int rows, int C, long long count, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, long long* nbt, float momentum, float eps,
                             float* mean_out, float* invstd_out, float* scale_out, float* shift_out, float* scratch,
                             float* stat_shift, hipStream_t s) {
  const float* sl = shrink_slab(slab, &rows, C, scratch, s);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(FIN_T), 0, s, sl, rows, C, (float)count, gamma, beta,
                     running_mean, running_var, nbt, momentum, eps, mean_out, invstd_out, scale_out, shift_out, stat_shift);
  return dpa_launch_rc();
}
DPA_API int dpa_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                float* scale_out, float* shift_out, hipStream_t s) {
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 63) / 64), dim3(64), 0, s, C, gamma, beta, rm, rv, eps, scale_out, shift_out);
  return dpa_launch_rc();
}
DPA_API int dpa_bn_apply(const void* x, int x_cs, const float* scale, const float* shift, const void* res, int r_cs, void* y,
                          int y_cs, long long M, int C, int relu, int dtype, hipStream_t s, void* bits, const float* rscale,
                          const float* rshift) {
  if ((bits && (C % 32)) || (rscale && !res) || (!rscale != !rshift)) return 1;
  int nb; const int rpb = ew_rows_per_block(M, C, dtype == 1 ? 2 : 4, &nb);
  if (dtype == 1) hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)x, x_cs, scale, shift,
                                     (const bf16_t*)res, r_cs, (bf16_t*)y, y_cs, M, C, relu, rpb, (uint32_t*)bits, rscale, rshift);
  else            hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)x, x_cs, scale, shift,
                                     (const float*)res, r_cs, (float*)y, y_cs, M, C, relu, rpb, (uint32_t*)bits, rscale, rshift);
  return dpa_launch_rc();
}
// slab: >= dpa_bn_slab_rows(M)*2*C floats; scratch >= ROWS_L2*2*C; sums: 2*C floats; coef: 3*C floats (for bn_bwd_dx)
DPA_API int dpa_bn_bwd_reduce(void* dy, int dy_cs, const void* y, int y_cs, const void* x, int x_cs, const float* mean,
                               const float* invstd, const float* gamma, long long M, int C, float* slab, float* scratch,
                               float* dgamma, float* dbeta, float* sums, float* coef, int accumulate, int dtype,
                               hipStream_t s, const float* mscale, const float* mshift) {
  int nb; const int rpb = rows_per_block(M, &nb);
  if (dtype == 1) hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (bf16_t*)dy, dy_cs, (const bf16_t*)y,
                                     y_cs, (const bf16_t*)x, x_cs, mean, invstd, M, C, slab, rpb, mscale, mshift);
  else            hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(nb), dim3(256), 0, s, (float*)dy, dy_cs, (const float*)y,
                                     y_cs, (const float*)x, x_cs, mean, invstd, M, C, slab, rpb, mscale, mshift);
  int rows = nb;
  const float* sl = shrink_slab(slab, &rows, C, scratch, s);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(FIN_T), 0, s, sl, rows, C, dgamma, dbeta, sums, accumulate,
                     mean, invstd, gamma, 1.0f / (float)M, coef);
  return dpa_launch_rc();
}
// The second half of dpa_bn_bwd_reduce alone: the per-tile partial sums were written by a dgrad epilogue
// (ConvParams::bnr_slab, conv_common.h) - no pass over dy / y / x here.
DPA_API int dpa_bn_bwd_from_slab(const float* slab, int rows, const float* mean, const float* invstd, const float* gamma,
                                  long long M, int C, float* scratch, float* dgamma, float* dbeta, float* sums, float* coef,
                                  int accumulate, hipStream_t s) {
  const float* sl = shrink_slab(slab, &rows, C, scratch, s);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(FIN_T), 0, s, sl, rows, C, dgamma, dbeta, sums, accumulate,
                     mean, invstd, gamma, 1.0f / (float)M, coef);
  return dpa_launch_rc();
}
DPA_API int dpa_bn_bwd_dx(const void* dz, int dz_cs, const void* x, int x_cs, const float* coef, void* dx, int dx_cs,
                           long long M, int C, int dtype, hipStream_t s, const float* mscale, const float* mshift, int accumulate) {
  int nb; const int rpb = ew_rows_per_block(M, C, dtype == 1 ? 2 : 4, &nb);
#define DPA_BWD_DX(T, ACC) hipLaunchKernelGGL((bn_bwd_dx_kernel<T, ACC>), dim3(nb), dim3(256), 0, s, (const T*)dz, dz_cs, \
                                               (const T*)x, x_cs, coef, (T*)dx, dx_cs, M, C, rpb, mscale, mshift)
  if (dtype == 1) { if (accumulate) DPA_BWD_DX(bf16_t, true); else DPA_BWD_DX(bf16_t, false); }
  else            { if (accumulate) DPA_BWD_DX(float, true);  else DPA_BWD_DX(float, false); }
#undef DPA_BWD_DX
  return dpa_launch_rc();
}
DPA_API int dpa_bn_bwd_dx2(const void* dz, int dz_cs, const void* x1, int x1_cs, const float* coef1, void* dx1, int dx1_cs,
                            const void* x2, int x2_cs, const float* coef2, void* dx2, int dx2_cs, long long M, int C, int dtype,
                            hipStream_t s) {
  int nb; const int rpb = ew_rows_per_block(M, C, dtype == 1 ? 2 : 4, &nb);
  if (dtype == 1) hipLaunchKernelGGL(bn_bwd_dx2_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)dz, dz_cs, (const bf16_t*)x1, x1_cs,
                                     coef1, (bf16_t*)dx1, dx1_cs, (const bf16_t*)x2, x2_cs, coef2, (bf16_t*)dx2, dx2_cs, M, C, rpb);
  else            hipLaunchKernelGGL(bn_bwd_dx2_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)dz, dz_cs, (const float*)x1, x1_cs,
                                     coef1, (float*)dx1, dx1_cs, (const float*)x2, x2_cs, coef2, (float*)dx2, dx2_cs, M, C, rpb);
  return dpa_launch_rc();
}

// Per-channel column sum of a tall activation (conv bias gradients): stats slab -> fixed-order reduce.
__global__ void __launch_bounds__(FIN_T) slab_colsum_finalize_kernel(const float* __restrict__ slab, int rows, int C,
                                                                  float* __restrict__ out, int accumulate) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  float s, q;
  if (!slab_sum2(slab, rows, C, false, s, q)) return;
  out[c] = accumulate ? out[c] + s : s;
}
DPA_API int dpa_colsum_tall(const void* x, int dtype, long long M, int C, int cs, float* slab, float* scratch, float* out,
                             int accumulate, hipStream_t s) {
  const int rc0 = dpa_bn_stats(x, dtype, M, C, cs, slab, nullptr, s);
  if (rc0) return rc0;
  int rows = dpa_bn_slab_rows(M);
  const float* sl = shrink_slab(slab, &rows, C, scratch, s);
  hipLaunchKernelGGL(slab_colsum_finalize_kernel, dim3((C + 63) / 64), dim3(FIN_T), 0, s, sl, rows, C, out, accumulate);
  return dpa_launch_rc();
}

// Bias gradient of conv + bias + ReLU: dy *= (y > 0) in place and out (+)= column sums of the masked dy, one pass.
DPA_API int dpa_relu_bwd_colsum(void* dy, int dy_cs, const void* y, int y_cs, int dtype, long long M, int C, float* slab,
                                 float* scratch, float* out, int accumulate, hipStream_t s) {
  int nb; const int rpb = rows_per_block(M, &nb);
  if (dtype == 1) hipLaunchKernelGGL((bn_stats_kernel<bf16_t, true>), dim3(nb), dim3(256), 0, s, (bf16_t*)dy, M, C, dy_cs, slab, rpb,
                                     (const float*)nullptr, (const bf16_t*)y, y_cs);
  else            hipLaunchKernelGGL((bn_stats_kernel<float, true>), dim3(nb), dim3(256), 0, s, (float*)dy, M, C, dy_cs, slab, rpb,
                                     (const float*)nullptr, (const float*)y, y_cs);
  int rows = nb;
  const float* sl = shrink_slab(slab, &rows, C, scratch, s);
  hipLaunchKernelGGL(slab_colsum_finalize_kernel, dim3((C + 63) / 64), dim3(FIN_T), 0, s, sl, rows, C, out, accumulate);
  return dpa_launch_rc();
}
