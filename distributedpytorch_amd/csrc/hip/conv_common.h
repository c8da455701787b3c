// Implicit-GEMM convolution on gfx950 MFMA: shared parameter block + LDS/MFMA helpers.
//
// Data layout (MI355X-first, not the reference's NCHW): activations are NHWC with the channel dimension
// contiguous, weights are [Cout][kh][kw][Cin] ("KRSC"), so the GEMM reduction index (kh,kw,cin) is
// contiguous in 16-byte chunks for BOTH operands and every global load / LDS access is 16 B per lane.
// A tensor may be a channel slice of a wider buffer (channel stride `cs`, channel offset `co`): that is how
// concat (SqueezeNet Fire, DenseNet, Inception) is done zero-copy (K12 of SURVEY.md §2.5).
//
// LDS image (both operands, both dtypes): rows of 128 bytes = 8 chunks of 16 B (64 bf16 or 32 fp32 of the
// reduction dim), chunk index XOR-swizzled with (row>>1)&7 so that ds_write_b128 (8 lanes = one row) and
// the ds_read_b128 fragment reads of mfma 16x16 (lane -> row l&15, chunk l>>4) are bank-conflict-free.
#pragma once
#include "common.h"

struct FastDiv {  // n / d for 0 <= n < 2^31, d >= 1 : q = mulhi(n, mul) >> shift
  uint32_t mul, shift, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f; f.d = d;
  if (d == 1) { f.mul = 0; f.shift = 0; return f; }
  uint32_t l = 0; while ((1u << l) < d) ++l;           // ceil(log2 d)
  uint64_t m = ((1ull << 32) * ((1ull << l) - d)) / d + 1;
  f.mul = (uint32_t)m; f.shift = l; return f;
}
__device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv& f) {
  if (f.d == 1) return n;
  uint32_t t = __umulhi(n, f.mul);
  return (t + ((n - t) >> 1)) >> (f.shift - 1);
}

struct ConvParams {
  const void* src;   // gathered operand: x (fprop) or dy (dgrad)            [sN, sH, sW, s_cs] (+s_co)
  const void* wgt;   // [dC][R][S][sC] reduction-contiguous filter (fprop: W ; dgrad: W transposed)
  void* dst;         // y (fprop) or dx (dgrad)                              [sN, dH, dW, d_cs] (+d_co)
  const float* bias; // [dC] or null
  float* stats;      // [gridM][2][dC] per-block (sum, sumsq) slab of the fp32 accumulators, or null
  const void* addend; // optional tensor with dst geometry added in the epilogue (dgrad accumulation) or null
  const float* stats_shift; // [dC] or null: the slab holds sum(v - c), sum((v - c)^2) about this per-channel shift c
                            // (shifted-data variance: no cancellation when |mean| >> std, see bn_finalize_kernel)
  const void* wgt3;  // fp32 x6 fast path: the filter pre-split into three bf16 planes [3][dC][R][S][sC]
                     // (split3_planes_kernel), or null
  long long wgt3_plane;  // elements per plane
  // dgrad only: BatchNorm-backward reduction fused into the epilogue.  dst is the gradient w.r.t. the OUTPUT of a
  // BatchNorm(+ReLU) whose input was bnr_x: the epilogue applies the ReLU mask to the value it stores (from bnr_y > 0,
  // or from bnr_mscale * x + bnr_mshift > 0, or none) and writes per-M-tile partial sums of dz and dz * xhat into
  // bnr_slab [rows][2][dC] - the memory pass bn_bwd_reduce_kernel would otherwise make over dy, y and x.
  const void* bnr_x;         // BN input, same pixels as dst, channel stride bnr_cs; null = no fusion
  const void* bnr_y;         // BN(+residual)+ReLU output for the mask, channel stride bnr_y_cs; or null
  const float* bnr_mean;     // [dC]
  const float* bnr_invstd;   // [dC]
  const float* bnr_mscale;   // [dC] or null
  const float* bnr_mshift;   // [dC] or null
  float* bnr_slab;
  const void* bnr_bits;      // or null: the ReLU mask as bits, [pixels][dC / 32] words (bn_apply_kernel), read INSTEAD of
                             // bnr_y: 1 / 32 of the bytes (dC % 32 == 0)
  // A SECOND BatchNorm whose output was added to the first one's before the ReLU (the downsample branch of a ResNet
  // block): same gradient, same mask, so sum dz is shared and the epilogue only adds sum dz * xhat2 over its input
  // bnr2_x: slab rows (sum dz, sum dz * xhat2) go to bnr2_slab.  Only conv_x6p.hip, and only together with bnr_bits
  // (dpa_conv_igemm returns 5 before launching anything otherwise).
  const void* bnr2_x;
  const float* bnr2_mean;
  const float* bnr2_invstd;
  float* bnr2_slab;
  int bnr2_cs, bnr2_pad;
  int bnr_cs, bnr_y_cs;
  int bnr_row0;              // first slab row of this parity class (host)
  int bnr_pad;
  // Per-output-channel multiplier applied to the accumulator BEFORE the bias: out = acc * oscale[c] + bias[c] (+ addend)
  // (ReLU).  With oscale = gamma / sqrt(running_var + eps) and bias = beta - running_mean * oscale this is an inference
  // BatchNorm folded into the convolution (fprop only; null = 1).
  const float* oscale;
  // dgrad only: a second, COMPACT addend [sN, a2_H, a2_W, a2_cs]: output pixel (h, w) gets addend2[img][h / 2][w / 2] added
  // when h and w are both even, nothing otherwise - the gradient of a 1x1 / stride-2 convolution (a ResNet downsample
  // branch) without its three quarters of zeros.  Only the pre-split fp32 kernel (conv_x6p.hip) implements it; every
  // other path makes dpa_conv_igemm return 4 before anything is launched (the caller then adds a dense tensor).
  const void* addend2;
  int a2_H, a2_W, a2_cs, a2_pad;
  int sN, sH, sW, sC, s_cs, s_co;
  int dH, dW, dC, d_cs, d_co;
  int R, S, sh, sw, ph, pw;
  int M;             // sN*dH*dW
  int Kg;            // R*S*sC
  int relu;          // apply max(.,0) in the epilogue
  int no_uni;        // debug / A-B switch: force the general address path (DPA_CONV_UNI=0)
  int m_off;         // first GEMM row of this launch (a conv may be split into a big-tile and a small-tile launch)
  int stats_row0;    // first stats-slab row of this launch
  int splitK;        // > 1: blockIdx = split * tiles + tile; raw fp32 partial sums go to `ws`, epilogue in the reduce kernel
  int kt_per_split;  // K-tiles per split
  float* ws;         // split-K workspace [splitK][M][dC]
  int tilesM, tilesN;
  // dgrad "parity class" geometry: output pixels with (hi+ph) % sh == a, (wi+pw) % sw == b form a dense sub-problem
  // that only sees taps r = a + sh*t, s = b + sw*t (stride-1 convs are the single class a = b = 0).
  int cls_h0, cls_w0;      // first output pixel of the class along each dim
  int cls_nh, cls_nw;      // pixels of the class along each dim  (fprop: dH, dW)
  int base_y, base_x;      // source coordinate of class pixel 0 under tap 0 of the class
  int tap_r0, tap_s0;      // first tap of the class
  int nR, nS;              // taps of the class along each dim      (fprop: R, S)
  int stats_rows_used;     // OUT (host): rows of the stats slab this launch wrote (depends on the tile plan)
  FastDiv fd_nW, fd_nH, fd_sC, fd_nS;
};

template <typename T> struct MfmaT;
template <> struct MfmaT<bf16_t> {
  // one 16-byte chunk = 8 bf16 = one 16x16x32 k-quarter per lane
  static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct MfmaT<float> {
  // one 16-byte chunk = 4 fp32 -> four exact-f32 16x16x4 MFMAs (k slot l>>4 <-> element t of the chunk,
  // the same bijection on both operands, so the sum over k is unchanged)
  static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

// ---- fp32 through the bf16 matrix pipe ("x6" mode) ---------------------------------------------------------------
// x = p0 + p1 + p2 EXACTLY, each part a bf16 (round-to-nearest of the value, of the remainder, of the second
// remainder: 8 + 8 + 8 significant bits cover the 24-bit fp32 significand).  A product of two fp32 numbers is then
// evaluated as p0q0 + p0q1 + p1q0 + p0q2 + p2q0 + p1q1 on v_mfma_f32_16x16x32_bf16 (fp32 accumulation); the three
// dropped terms are below 2^-23 of the product, i.e. at the rounding of one fp32 multiply, and measured against an
// fp64 reference the result is slightly MORE accurate than the exact-fp32 16x16x4 MFMA chain (fewer accumulator
// roundings: 32 products per instruction instead of 4) - scripts/x6_accuracy.py, profiles/README.md.
// The bf16 pipe does 16x the MACs per cycle of the fp32 one, so 6 products still cost 2.7x fewer matrix cycles.
extern "C" int dpa_get_fp32_mode();      // 0 = exact fp32 MFMA, 1 = x6 (default); DPA_FP32_MODE / dpa_set_fp32_mode
// Two ways to cut x into three bf16 parts, both EXACT (8 + 8 + 8 significant bits):
//   round-to-nearest (default): p = bf16_rn(r), r -= p.  v_cvt_pk_bf16_f32 + v_pk_add_f32 per part;
//   truncation (-DDPA_SPLIT_TRUNC=1, an A/B build: csrc/build.py --libdir ... -D DPA_SPLIT_TRUNC=1): p = r & 0xffff0000
//     (the top 16 bits of an fp32 ARE a bf16), r -= p.  v_and_b32 + v_sub_f32 + one v_perm_b32 per pair - 11 instructions
//     per pair instead of 9, but none of the two kinds that profiles/r1_micro_valu_mfma_coissue.txt showed not overlapping
//     with MFMAs.  MEASURED in round 3 over the ResNet-50 layer table at batch 1024 (profiles/r3_split_trunc_ab.txt):
//     fprop 51.6 vs 52.3 ms, dgrad 66.7 vs 68.0, wgrad 52.9 vs 51.6 - 171.2 vs 171.8 ms in total, i.e. nothing: the
//     split's instruction kinds are not what holds the x6 kernels at 57-64 % matrix-pipe utilisation.  Truncated parts all
//     carry the sign of x, so the dropped cross terms become one-signed; with no speed to buy, the default stays RN.
#ifndef DPA_SPLIT_TRUNC
#define DPA_SPLIT_TRUNC 0
#endif
// round-to-nearest parts: the once-per-step filter planes (weight_prep_kernel) always use these - they are not in any K
// loop, and with zero-mean filter remainders the cross terms dropped against truncated ACTIVATION parts are zero-mean too
// (fprop / dgrad keep no one-signed bias at all; only wgrad, where both operands are split in the loop, has one)
__device__ __forceinline__ void split3_pair_rn(float x, float y, uint32_t (&o)[3]) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  f32x2_t r = {x, y};
#pragma unroll
  for (int part = 0; part < 3; ++part) {
    const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2_t));   // v_cvt_pk_bf16_f32
    o[part] = h;
    if (part < 2) {
      r.x -= __uint_as_float(h << 16);
      r.y -= __uint_as_float(h & 0xffff0000u);
    }
  }
}
__device__ __forceinline__ void split3_pair(float x, float y, uint32_t (&o)[3]) {
#if DPA_SPLIT_TRUNC
  uint32_t ux = __float_as_uint(x), uy = __float_as_uint(y);
#pragma unroll
  for (int part = 0; part < 3; ++part) {
    o[part] = __builtin_amdgcn_perm(uy, ux, 0x07060302u);      // (uy & 0xffff0000) | (ux >> 16)
    if (part < 2) {
      float rx, ry;      // (asm: two v_sub_f32, not one v_pk_add_f32 - the packed form is what does not overlap with MFMAs)
      asm("v_sub_f32 %0, %1, %2" : "=v"(rx) : "v"(__uint_as_float(ux)), "v"(__uint_as_float(ux & 0xffff0000u)));
      asm("v_sub_f32 %0, %1, %2" : "=v"(ry) : "v"(__uint_as_float(uy)), "v"(__uint_as_float(uy & 0xffff0000u)));
      ux = __float_as_uint(rx); uy = __float_as_uint(ry);
    }
  }
#else
  split3_pair_rn(x, y, o);
#endif
}
// order of the six partial products: small terms first
__device__ constexpr int kX6A[6] = {1, 2, 0, 1, 0, 0};
__device__ constexpr int kX6B[6] = {1, 0, 2, 0, 1, 0};

// ---- LDS-DMA from inline asm ----------------------------------------------------------------------------------------
// One 1-KiB LDS-DMA piece issued from inline asm: hipcc does not count it, so it puts no s_waitcnt vmcnt(0) in front of
// the LDS reads that follow (it does for the builtin: it cannot tell that they read another buffer) - the K loops that use it wait for its DMAs itself, once per K-tile.  M0 (the LDS destination) is saved and restored inside the
// statement (cdna_hip_programming.md §5.7).  lds_addr must be wave-uniform.
__device__ __forceinline__ void dpa_glds16(const void* g, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}

// Two pieces from a wave-uniform base (SGPR pair) + a 32-bit lane offset: bytes [0, 16) of every lane's address go to
// lds_a, bytes [128, 144) to lds_b.  No 64-bit vector address arithmetic at all (conv_x6s.hip: a row's two K-tile halves).
// The instruction's immediate offset is added to the LDS address as well as to the global one (observed: with M0 = lds_b
// the second piece landed 128 bytes further), so M0 for the second piece is lds_b - 128.
__device__ __forceinline__ void dpa_glds16x2_s(const void* sbase, uint32_t voff, uint32_t lds_a, uint32_t lds_b) {
  uint32_t keep;
  lds_b -= 128u;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
               "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:128\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_a), "s"(lds_b) : "memory");
}

// ---- fused BatchNorm-backward reduction (dgrad epilogues of conv_igemm.hip and conv_x6p.hip) --------------------------
// One lane owns VEC consecutive channels of a few rows of the tile.
template <typename T, int VEC>
struct BnReduceLane {
  float s[VEC], q[VEC], mu[VEC], is[VEC], msc[VEC], msh[VEC];
  __device__ __forceinline__ void init(const ConvParams& p, int col) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const bool ok = col + e < p.dC;
      s[e] = 0.f; q[e] = 0.f;
      mu[e] = ok ? p.bnr_mean[col + e] : 0.f; is[e] = ok ? p.bnr_invstd[col + e] : 0.f;
      msc[e] = (ok && p.bnr_mscale) ? p.bnr_mscale[col + e] : 0.f;
      msh[e] = (ok && p.bnr_mscale) ? p.bnr_mshift[col + e] : 1.f;
    }
  }
  // the same with the BatchNorm input x (and output y) already in registers
  __device__ __forceinline__ void visit_loaded(bool have_y, const float (&x)[VEC], const float (&y)[VEC], float (&v)[VEC]) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const bool keep = have_y ? (y[e] > 0.f) : (x[e] * msc[e] + msh[e] > 0.f);      // (no mask: 0 * x + 1 > 0)
      float d = keep ? v[e] : 0.f;
      v[e] = d;
      if (sizeof(T) == 2) d = bf2f(f2bf(d));
      s[e] += d; q[e] += d * (x[e] - mu[e]) * is[e];
    }
  }
  // v: the gradient values about to be stored for (pixel, col .. col + VEC); masked in place, sums updated
  __device__ __forceinline__ void visit(const ConvParams& p, size_t pixel, int col, float (&v)[VEC]) {
    Vec16<T> xv; xv.load(reinterpret_cast<const T*>(p.bnr_x) + pixel * (size_t)p.bnr_cs + col);
    if (p.bnr_bits) {         // the mask as bits (dC % 32 == 0; VEC consecutive channels lie inside one word): y is not read
      const uint32_t w = reinterpret_cast<const uint32_t*>(p.bnr_bits)[pixel * (size_t)(p.dC >> 5) + (size_t)(col >> 5)] >> (col & 31);
#pragma unroll
      for (int e = 0; e < VEC; ++e) if (!((w >> e) & 1u)) v[e] = 0.f;
    } else if (p.bnr_y) {
      Vec16<T> yv; yv.load(reinterpret_cast<const T*>(p.bnr_y) + pixel * (size_t)p.bnr_y_cs + col);
#pragma unroll
      for (int e = 0; e < VEC; ++e) if (!(yv.get(e) > 0.f)) v[e] = 0.f;
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) if (!(xv.get(e) * msc[e] + msh[e] > 0.f)) v[e] = 0.f;      // (no mask: 0 * x + 1 > 0)
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float d = v[e];
      if (sizeof(T) == 2) d = bf2f(f2bf(d));     // sums of the values as stored
      s[e] += d; q[e] += d * (xv.get(e) - mu[e]) * is[e];
    }
  }
};

// The same for conv_igemm.hip's epilogue, which prefetches its operands and has 168 registers (three blocks per CU): the
// mask coefficients (used only when the mask comes from x, a path without a y / bits load) are read when needed instead
// of living in 2 * VEC registers.  Same sums, same order as BnReduceLane::visit.
template <typename T, int VEC>
struct BnReduceLaneLite {
  float s[VEC], q[VEC], mu[VEC], is[VEC];
  __device__ __forceinline__ void init(const ConvParams& p, int col) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const bool ok = col + e < p.dC;
      s[e] = 0.f; q[e] = 0.f;
      mu[e] = ok ? p.bnr_mean[col + e] : 0.f; is[e] = ok ? p.bnr_invstd[col + e] : 0.f;
    }
  }
  // x: the BatchNorm input; keep: bit e = the mask from y or the bits (have_mask), else from mscale * x + mshift > 0, else all
  __device__ __forceinline__ void visit_loaded(const ConvParams& p, int col, bool have_mask, uint32_t keep_bits,
                                               const Vec16<T>& xv, float (&v)[VEC]) {
    if (!have_mask) {
      keep_bits = ~0u;
      if (p.bnr_mscale) {
        keep_bits = 0;
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          keep_bits |= (xv.get(e) * p.bnr_mscale[col + e] + p.bnr_mshift[col + e] > 0.f ? 1u : 0u) << e;
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float d = ((keep_bits >> e) & 1u) ? v[e] : 0.f;
      v[e] = d;
      if (sizeof(T) == 2) d = bf2f(f2bf(d));     // sums of the values as stored
      s[e] += d; q[e] += d * (xv.get(e) - mu[e]) * is[e];
    }
  }
};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// XCD-aware, bijective remap of the linear block id: blocks that share an XCD (id % 8) get a contiguous
// run of tiles, so neighbouring tiles (which share activation rows / filter panels) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
