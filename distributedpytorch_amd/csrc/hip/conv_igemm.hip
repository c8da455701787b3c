// Implicit-GEMM convolution forward (fprop) and data-gradient (dgrad) for gfx950 (K1/K2 of SURVEY.md §2.5;
// replaces the cuDNN convolutions the reference reaches through torchvision models at
// /root/reference/classif.py:55,59 and utils.py:45-91).
//
//   GEMM view:  D[m][n] = sum_k A[m][k] * B[n][k]
//     fprop : m = (img, ho, wo)  n = cout  k = (r, s, cin)   A = im2col(x) gathered on the fly, B = W[cout][r][s][cin]
//     dgrad : m = (img, hi, wi)  n = cin   k = (r, s, cout)  A = gathered dy (only taps with (hi+p-r) % stride == 0),
//                                                             B = Wt[cin][r][s][cout]
//   Block = 256 threads = 4 waves (WM x WN), each wave owns (TM*16) x (TN*16) of the output as TM x TN
//   accumulators of mfma 16x16 (bf16: v_mfma_f32_16x16x32_bf16, fp32: exact v_mfma_f32_16x16x4_f32).
//   K step = 128 bytes per row (64 bf16 / 32 fp32).  Register-staged software pipeline: the global loads of
//   K-tile t+1 are issued before the MFMAs of tile t and written to LDS after the barrier.
//   Epilogue (fused): + bias, + addend, ReLU, per-channel (sum, sumsq) slab for BatchNorm statistics, and the
//   tile is staged through LDS so global stores are 16 B per lane along the channel dimension, at an
//   arbitrary channel offset/stride of the destination buffer (zero-copy concat).
#include "conv_common.h"
#ifndef DPA_GLDS_EARLY
#define DPA_GLDS_EARLY 1      // 0: the round-2 loop (DMA issued behind the fragment reads through the builtin); A/B builds
#endif
#include <map>
#include <mutex>
#include <cstdlib>

// 64 zero bytes: the source of every padded / out-of-range 16-byte chunk of the LDS-DMA staging path
__device__ uint4 dpa_zero_chunk[4];

// Split one lane's 8 fp32 fragment values (two 16-byte chunks) into three bf16 planes (conv_common.h: split3_pair).
__device__ __forceinline__ void split3_bf16(const uint4& lo, const uint4& hi, uint4 (&pl)[3]) {
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  uint32_t o[4][3];
#pragma unroll
  for (int q = 0; q < 4; ++q) split3_pair(__uint_as_float(w[2 * q]), __uint_as_float(w[2 * q + 1]), o[q]);
#pragma unroll
  for (int part = 0; part < 3; ++part) pl[part] = make_uint4(o[0][part], o[1][part], o[2][part], o[3][part]);
}

// X6 (fp32 operands, LDS-DMA staging only): every fp32 product is evaluated as 6 bf16 x bf16 products of the exact
// 3-way bf16 splits of the operands (p0q0 + p0q1 + p1q0 + p0q2 + p2q0 + p1q1, fp32 accumulation in the MFMA); the
// dropped terms are below 2^-23 of the product, i.e. at the rounding of an fp32 multiply, and the bf16 matrix pipe
// is 16x the fp32 one per MAC (6 x 16-cycle 16x16x32 MFMAs replace 8 x 32-cycle 16x16x4 ones per 32 k).
// BNR: a data-gradient launch with the fused BatchNorm-backward reduction (p.bnr_x != null).  An instantiation of its own:
// its epilogue keeps a prefetched group of operands next to the reduction's per-channel state, which does not fit the 168
// registers of three blocks per CU (these launches are bound by their epilogue's memory traffic, not by the K loop).
template <typename T, int WM, int WN, int TM, int TN, bool DGRAD, bool GLDS, bool SPLIT = false, bool X6 = false, bool UNI = false,
          bool BNR = false>
__global__ void __launch_bounds__(256, ((GLDS || WM == 4 || X6 || BNR) ? 2 : 3)) conv_igemm_kernel(const ConvParams p) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int VEC = 16 / sizeof(T);        // elements per 16-byte chunk
  constexpr int BKE = 8 * VEC;               // reduction elements per K-tile (128-byte rows)
  constexpr int A_PASSES = BM / 32, B_PASSES = BN / 32;
  static_assert(WM * WN == 4, "4 waves per block");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE_BYTES = (BM + BN) * 128;      // one K-tile of both operands
  char* ldsA = smem;
  char* ldsB = smem + BM * 128;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int ntiles = p.tilesM * p.tilesN;
  const int split = SPLIT ? (int)blockIdx.x / ntiles : 0;      // block-uniform (SPLIT kernels only run with splitK > 1)
  const int tile = xcd_remap((int)blockIdx.x - split * ntiles, ntiles);
  const int tm = tile / p.tilesN, tn = tile - tm * p.tilesN;
  const int m0 = p.m_off + tm * BM, n0 = tn * BN;

  const T* __restrict__ src = reinterpret_cast<const T*>(p.src);
  const T* __restrict__ wgt = reinterpret_cast<const T*>(p.wgt);

  // ---- per-thread gather state: this thread stages chunk `ck` of rows (tid>>3) + 32*i -----------------
  // Register staging: the thread owns logical chunk tid&7 and writes it to the swizzled slot.  LDS-DMA staging
  // (global_load_lds writes wave-base + lane*16, no per-lane scatter): the thread owns the physical slot tid&7 and
  // fetches the logical chunk that the swizzle maps there; (row>>1)&7 == ((wave&1)<<2 | lane>>4) for every pass
  // because rows advance by 32 per pass, so the chunk -> tap decomposition is still per-thread constant.
  const int row0 = tid >> 3;
  const int ck = GLDS ? ((tid & 7) ^ ((row0 >> 1) & 7)) : (tid & 7);
  uint32_t a_base[A_PASSES];   // element offset of (img, 0, 0, s_co) in src; 0xffffffff when the row is out of range
  int a_y0[A_PASSES], a_x0[A_PASSES];
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int m = m0 + row0 + 32 * i;
    if (m < p.M) {
      const uint32_t t1 = fd_div((uint32_t)m, p.fd_nW);
      const int xo = m - (int)t1 * p.cls_nw;
      const uint32_t img = fd_div(t1, p.fd_nH);
      const int yo = (int)t1 - (int)img * p.cls_nh;
      a_base[i] = img * (uint32_t)(p.sH * p.sW) ;
      if (DGRAD) { a_y0[i] = yo + p.base_y; a_x0[i] = xo + p.base_x; }
      else       { a_y0[i] = yo * p.sh - p.ph; a_x0[i] = xo * p.sw - p.pw; }
    } else {
      a_base[i] = 0xffffffffu; a_y0[i] = 0; a_x0[i] = 0;
    }
  }
  uint32_t b_base[B_PASSES];
#pragma unroll
  for (int i = 0; i < B_PASSES; ++i) {
    const int n = n0 + row0 + 32 * i;
    b_base[i] = (n < p.dC) ? (uint32_t)n * (uint32_t)(p.R * p.S * p.sC) : 0xffffffffu;
  }

  // ---- uniform-tap fast path --------------------------------------------------------------------------------
  // When the channel count is a multiple of the K-tile (every ResNet/VGG/DenseNet/Inception layer but the stem) all
  // chunks of a K-tile belong to ONE filter tap, so (r, s) and the tap's address delta are wave-uniform (scalar
  // ALU) and the per-row validity of each tap is a precomputed bit mask: per pass the address generation is a bit
  // test, one 64-bit add and a select instead of two magic divisions, bounds compares and a multiply (the first
  // version spent 4.9 VALU instructions per MFMA here, profiles/README.md).
  const int ntaps = p.nR * p.nS;
  // (a dgrad parity class without taps has Kg == 0: it must take the general path, whose loads are all predicated off)
  // UNI: the host has checked the condition - the general path is not even compiled into the K loop
  const bool uni = UNI || ((p.sC % BKE == 0) && ntaps >= 1 && ntaps <= 32 && !p.no_uni);
  long long a_rowoff[A_PASSES];
  uint32_t a_mask[A_PASSES];
  if (uni) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      a_rowoff[i] = ((long long)a_base[i] + (long long)a_y0[i] * p.sW + a_x0[i]) * p.s_cs + p.s_co + ck * VEC;
      uint32_t mk = 0;
      if (a_base[i] != 0xffffffffu) {
        for (int t = 0; t < ntaps; ++t) {
          const int r = (int)fd_div((uint32_t)t, p.fd_nS), sx_ = t - r * p.nS;
          const int sy = DGRAD ? a_y0[i] - r : a_y0[i] + r;
          const int sx = DGRAD ? a_x0[i] - sx_ : a_x0[i] + sx_;
          if (((unsigned)sy < (unsigned)p.sH) && ((unsigned)sx < (unsigned)p.sW)) mk |= 1u << t;
        }
      }
      a_mask[i] = mk;
    }
  }
  const T* const zero = reinterpret_cast<const T*>(dpa_zero_chunk);

  // Source pointer of every chunk this thread stages for K-tile kt (the zero chunk for padding / out of range).
  const T* pa[A_PASSES];
  const T* pb[B_PASSES];
  auto gen_addresses = [&](int kt) __attribute__((always_inline)) {
    if (uni) {
      const uint32_t k0 = (uint32_t)kt * BKE;                       // wave-uniform from here on
      const uint32_t tap = fd_div(k0, p.fd_sC);
      const int cbase = (int)(k0 - tap * (uint32_t)p.sC);
      const int r = (int)fd_div(tap, p.fd_nS);
      const int s = (int)tap - r * p.nS;
      const long long delta = (long long)(DGRAD ? -(r * p.sW + s) : (r * p.sW + s)) * p.s_cs + cbase;
      const int woff = (DGRAD ? (((p.tap_r0 + p.sh * r) * p.S + p.tap_s0 + p.sw * s) * p.sC + cbase) : (int)k0) + ck * VEC;
      const uint32_t bit = 1u << tap;
#pragma unroll
      for (int i = 0; i < A_PASSES; ++i) pa[i] = (a_mask[i] & bit) ? src + (a_rowoff[i] + delta) : zero;
#pragma unroll
      for (int i = 0; i < B_PASSES; ++i)   // rows beyond dC alias row 0: their output columns are never stored
        pb[i] = wgt + (size_t)(b_base[i] == 0xffffffffu ? 0u : b_base[i]) + woff;
    } else {
    const int kidx = kt * BKE + ck * VEC;
    const bool kin = kidx < p.Kg;
    const uint32_t tap = fd_div((uint32_t)kidx, p.fd_sC);
    const int c = kidx - (int)tap * p.sC;
    const int r = (int)fd_div(tap, p.fd_nS);
    const int s = (int)tap - r * p.nS;
    // element offset of this chunk inside one filter row [R][S][sC]
    const int woff = DGRAD ? (((p.tap_r0 + p.sh * r) * p.S + p.tap_s0 + p.sw * s) * p.sC + c) : kidx;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      bool ok = kin && (a_base[i] != 0xffffffffu);
      const int sy = DGRAD ? a_y0[i] - r : a_y0[i] + r;
      const int sx = DGRAD ? a_x0[i] - s : a_x0[i] + s;
      ok = ok && ((unsigned)sy < (unsigned)p.sH) && ((unsigned)sx < (unsigned)p.sW);
      const size_t off = ((size_t)a_base[i] + (size_t)(sy * p.sW + sx)) * (size_t)p.s_cs + (size_t)(p.s_co + c);
      pa[i] = ok ? src + off : zero;
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i)
      pb[i] = (kin && b_base[i] != 0xffffffffu) ? wgt + (size_t)b_base[i] + woff : zero;
    }
  };

  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // native vector: stays in registers
  u32x4 ra[A_PASSES], rb[B_PASSES];
  auto load_tile = [&]() __attribute__((always_inline)) {     // after gen_addresses(kt)
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) ra[i] = *reinterpret_cast<const u32x4*>(pa[i]);
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) rb[i] = *reinterpret_cast<const u32x4*>(pb[i]);
  };
  auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) *reinterpret_cast<u32x4*>(ldsA + lds_off(row0 + 32 * i, ck)) = ra[i];
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) *reinterpret_cast<u32x4*>(ldsB + lds_off(row0 + 32 * i, ck)) = rb[i];
  };

  // LDS-DMA staging of K-tile kt into stage buffer `buf`: one global_load_lds_dwordx4 per 8 rows x 128 B per wave
  // (the destination is wave-base + lane*16, so the swizzle lives in the per-lane SOURCE chunk, see `ck`).
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void gbl_void;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto issue_tile = [&](int buf) __attribute__((always_inline)) {     // after gen_addresses(kt)
    char* stage = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void*)pa[i], (lds_void*)(stage + (i * 32 + wave_u * 8) * 128), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void*)pb[i], (lds_void*)(stage + BM * 128 + (i * 32 + wave_u * 8) * 128), 16, 0, 0);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt_all = (p.Kg + BKE - 1) / BKE;
  const int kt0 = SPLIT ? split * p.kt_per_split : 0;
  const int nkt = SPLIT ? min(nkt_all, kt0 + p.kt_per_split) : nkt_all;     // this block reduces K-tiles [kt0, nkt)
  const int fr = lane & 15, fq = lane >> 4;
  // x6: the 8 fp32 values a lane holds per fragment (both k-halves of the tile) become three bf16x8 planes = the
  // operands of 16x16x32 bf16 MFMAs; small terms first, j innermost so consecutive MFMAs hit independent accumulators
  auto mma_x6 = [&](const uint4 (&fa)[2][TM], const uint4 (&fb)[2][TN]) __attribute__((always_inline)) {
    uint4 bp[TN][3];
#pragma unroll
    for (int j = 0; j < TN; ++j) split3_bf16(fb[0][j], fb[1][j], bp[j]);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      uint4 ap[3];
      split3_bf16(fa[0][i], fa[1][i], ap);
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int j = 0; j < TN; ++j) MfmaT<bf16_t>::mma(ap[kX6A[t]], bp[j][kX6B[t]], acc[i][j]);
    }
  };
  auto compute_tile = [&](const char* tA, const char* tB) __attribute__((always_inline)) {
    if constexpr (X6) {
      uint4 fa[2][TM], fb[2][TN];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[ks][i] = *reinterpret_cast<const uint4*>(tA + lds_off(wm * TM * 16 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[ks][j] = *reinterpret_cast<const uint4*>(tB + lds_off(wn * TN * 16 + j * 16 + fr, ks * 4 + fq));
      }
      mma_x6(fa, fb);
      return;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = *reinterpret_cast<const uint4*>(tA + lds_off(wm * TM * 16 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[j] = *reinterpret_cast<const uint4*>(tB + lds_off(wn * TN * 16 + j * 16 + fr, ks * 4 + fq));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) MfmaT<T>::mma(fa[i], fb[j], acc[i][j]);
    }
  };
  if constexpr (GLDS) {
    // Two stage buffers, ONE barrier per K-tile.  Per iteration: [my DMA of tile kt landed] + barrier [everyone's
    // landed, everyone has read tile kt-1] -> read ALL fragments of tile kt into registers -> issue the DMA of tile
    // kt+1 into the buffer tile kt-1 used -> MFMAs.  The fragment reads come BEFORE the DMA issue on purpose: the
    // compiler puts s_waitcnt vmcnt(0) in front of every LDS read that follows an LDS-DMA (it cannot tell the two
    // buffers apart), which would wait for the prefetch right after issuing it and expose the whole global latency
    // every K-tile (measured: 63 % MFMA utilisation in fp32, profiles/README.md).
#if DPA_GLDS_EARLY
    // Round 3: the DMA of tile kt + 1 is issued FIRST in iteration kt, right behind the barrier, from inline asm
    // (dpa_glds16: hipcc does not count it, so the fragment reads behind it get no vmcnt(0)) - it has the whole iteration
    // to land instead of the MFMAs' 512 cycles, and its source addresses were generated under the previous tile's MFMAs.
    typedef __attribute__((address_space(3))) char lds_char;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_char*)smem);
    auto issue_asm = [&](int buf) __attribute__((always_inline)) {     // after gen_addresses(kt)
      const uint32_t stage = lds0 + buf * STAGE_BYTES;
#pragma unroll
      for (int i = 0; i < A_PASSES; ++i) dpa_glds16(pa[i], stage + (i * 32 + wave_u * 8) * 128);
#pragma unroll
      for (int i = 0; i < B_PASSES; ++i) dpa_glds16(pb[i], stage + BM * 128 + (i * 32 + wave_u * 8) * 128);
    };
    gen_addresses(kt0);      // (splits are never empty; a zero-tap dgrad class has all loads predicated to the zero chunk)
    issue_asm(0);
    if (kt0 + 1 < nkt) gen_addresses(kt0 + 1);
    for (int kt = kt0; kt < nkt; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < nkt) issue_asm((kt + 1 - kt0) & 1);
      const char* tA = smem + ((kt - kt0) & 1) * STAGE_BYTES;
      const char* tB = tA + BM * 128;
      uint4 fa[2][TM], fb[2][TN];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[ks][i] = *reinterpret_cast<const uint4*>(tA + lds_off(wm * TM * 16 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[ks][j] = *reinterpret_cast<const uint4*>(tB + lds_off(wn * TN * 16 + j * 16 + fr, ks * 4 + fq));
      }
      if (kt + 2 < nkt) gen_addresses(kt + 2);
      if constexpr (X6) {
        mma_x6(fa, fb);
      } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) MfmaT<T>::mma(fa[ks][i], fb[ks][j], acc[i][j]);
      }
    }
#else
    gen_addresses(kt0);      // (splits are never empty; a zero-tap dgrad class has all loads predicated to the zero chunk)
    issue_tile(0);
    for (int kt = kt0; kt < nkt; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const char* tA = smem + ((kt - kt0) & 1) * STAGE_BYTES;
      const char* tB = tA + BM * 128;
      uint4 fa[2][TM], fb[2][TN];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[ks][i] = *reinterpret_cast<const uint4*>(tA + lds_off(wm * TM * 16 + i * 16 + fr, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[ks][j] = *reinterpret_cast<const uint4*>(tB + lds_off(wn * TN * 16 + j * 16 + fr, ks * 4 + fq));
      }
      if (kt + 1 < nkt) { gen_addresses(kt + 1); issue_tile((kt + 1 - kt0) & 1); }
      if constexpr (X6) {
        mma_x6(fa, fb);
      } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) MfmaT<T>::mma(fa[ks][i], fb[ks][j], acc[i][j]);
      }
    }
#endif
    __syncthreads();
  } else {
    gen_addresses(kt0);
    load_tile();
    store_tile();
    __syncthreads();
    for (int kt = kt0; kt < nkt; ++kt) {
      if (kt + 1 < nkt) { gen_addresses(kt + 1); load_tile(); }
      compute_tile(ldsA, ldsB);
      __syncthreads();
      if (kt + 1 < nkt) {
        store_tile();
        __syncthreads();
      }
    }
  }

  // ---------------------------------------- epilogue --------------------------------------------------
  // acc[i][j][e] = D[row = wm*TM*16 + i*16 + fq*4 + e][col = wn*TN*16 + j*16 + fr]
  if constexpr (SPLIT) {     // raw partial sums; bias / addend / ReLU / store happen in splitk_reduce_kernel
    float* __restrict__ wsp = p.ws + (size_t)split * p.M * p.dC;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (TN * 16) + j * 16 + fr;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + wm * TM * 16 + i * 16 + fq * 4 + e;
          if (m < p.M && col < p.dC) wsp[(size_t)m * p.dC + col] = acc[i][j][e];
        }
      }
    return;
  }
  T* __restrict__ dst = reinterpret_cast<T*>(p.dst);
  const T* __restrict__ addend = reinterpret_cast<const T*>(p.addend);
  constexpr int WCOLS = TN * 16;               // columns owned by a wave
  constexpr int PITCH = WCOLS + 4;             // floats; keeps ds_write_b32 conflict-free and rows 16-B aligned
  float* stage = reinterpret_cast<float*>(smem) + wave * (16 * PITCH);
  float* red = reinterpret_cast<float*>(smem) + 4 * 16 * PITCH;   // [WM][2][BN] cross-wave stats scratch

  float csum[TN], csq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { csum[j] = 0.f; csq[j] = 0.f; }

  float bj[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * WCOLS + j * 16 + fr;
    bj[j] = (p.bias && col < p.dC) ? p.bias[col] : 0.f;
  }
  if constexpr (!SPLIT && !DGRAD) if (p.oscale) {      // folded inference BatchNorm: acc * scale[col] before the bias
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WCOLS + j * 16 + fr;
      const float sc = col < p.dC ? p.oscale[col] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] *= sc;
    }
  }
  float cj[TN];                                // statistics shift of this lane's columns
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * WCOLS + j * 16 + fr;
    cj[j] = (p.stats && p.stats_shift && col < p.dC) ? p.stats_shift[col] : 0.f;
  }
  constexpr int LPR = WCOLS / VEC;             // lanes per output row (16-B chunks per row)
  constexpr int RPI = 64 / LPR;                // rows per store instruction
  BnReduceLaneLite<T, VEC> bnr;                // fused BatchNorm-backward reduction (dgrad, p.bnr_x != null)
  if constexpr (BNR) bnr.init(p, n0 + wn * WCOLS + (lane % LPR) * VEC);
  // The global operands of the epilogue (addend, BatchNorm input, mask) of the NEXT group of RPI rows are requested before
  // the current group is consumed: one round trip to memory per group, 8 in a row per wave, was what bounded the fused
  // data-gradient launches (2.9 TB/s).
  constexpr int NRR = 16 / RPI;
  struct Pre { size_t off; Vec16<T> av, xv, yv; bool ok; };      // (yv.raw's first word holds the mask word when there are bits)
  const uint32_t* __restrict__ mbits = BNR ? reinterpret_cast<const uint32_t*>(p.bnr_bits) : nullptr;
  auto prefetch = [&](int k, Pre& t) __attribute__((always_inline)) {
    const int i = k / NRR, q = k % NRR;
    const int r = q * RPI + lane / LPR;
    const int m = m0 + wm * TM * 16 + i * 16 + r;
    const int col = n0 + wn * WCOLS + (lane % LPR) * VEC;
    t.ok = m < p.M && col < p.dC;
    if (!t.ok) return;
    size_t pixel = (size_t)m;
    if (DGRAD && (p.sh != 1 || p.sw != 1)) {      // class row -> (img, hi, wi) of the full gradient tensor
      const uint32_t t1 = fd_div((uint32_t)m, p.fd_nW);
      const int jx = m - (int)t1 * p.cls_nw;
      const uint32_t img = fd_div(t1, p.fd_nH);
      const int iy = (int)t1 - (int)img * p.cls_nh;
      pixel = ((size_t)img * p.dH + (size_t)(p.cls_h0 + p.sh * iy)) * p.dW + (size_t)(p.cls_w0 + p.sw * jx);
    }
    t.off = pixel * (size_t)p.d_cs + (size_t)(p.d_co + col);
    if (addend) t.av.load(addend + t.off);
    if constexpr (BNR) {
      t.xv.load(reinterpret_cast<const T*>(p.bnr_x) + pixel * (size_t)p.bnr_cs + col);
      if (mbits) *reinterpret_cast<uint32_t*>(&t.yv.raw) = mbits[pixel * (size_t)(p.dC >> 5) + (size_t)(col >> 5)] >> (col & 31);
      else if (p.bnr_y) t.yv.load(reinterpret_cast<const T*>(p.bnr_y) + pixel * (size_t)p.bnr_y_cs + col);
    }
  };
  constexpr int PD = 1;                        // groups of look-ahead (3 in the BNR instantiations measured the same: 85.2 ms bf16 step)
  Pre pre[PD + 1];
#pragma unroll
  for (int k = 0; k < PD && k < TM * NRR; ++k) prefetch(k, pre[k]);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    // wave-private staging region: no block barrier needed between iterations, only LDS ordering
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[i][j][e] + bj[j];
        stage[(fq * 4 + e) * PITCH + j * 16 + fr] = v;
      }
#pragma unroll
    for (int q = 0; q < NRR; ++q) {
      const int k = i * NRR + q;
      if (k + PD < TM * NRR) prefetch(k + PD, pre[(k + PD) % (PD + 1)]);
      if (q == 0) {
        __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
      }
      const int r = q * RPI + lane / LPR;
      const int cc = (lane % LPR) * VEC;
      const Pre& t = pre[k % (PD + 1)];
      float v[VEC];
#pragma unroll
      for (int e = 0; e < VEC; e += 4) {
        const float4 f = *reinterpret_cast<const float4*>(&stage[r * PITCH + cc + e]);
        v[e] = f.x; v[e + 1] = f.y; v[e + 2] = f.z; v[e + 3] = f.w;
      }
      if (t.ok) {
        if (addend) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[e] += t.av.get(e);
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (BNR) {
          uint32_t keep = *reinterpret_cast<const uint32_t*>(&t.yv.raw);
          if (!mbits && p.bnr_y) {
            keep = 0;
#pragma unroll
            for (int e = 0; e < VEC; ++e) keep |= (t.yv.get(e) > 0.f ? 1u : 0u) << e;
          }
          bnr.visit_loaded(p, n0 + wn * WCOLS + cc, mbits || p.bnr_y, keep, t.xv, v);
        }
        Vec16<T> ov;
#pragma unroll
        for (int e = 0; e < VEC; ++e) ov.set(e, v[e]);
        ov.store(dst + t.off);
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (p.stats) {
      if (m0 + BM <= p.M) {                        // interior tile (block-uniform): no per-element row check
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[i][j][e] + bj[j];
            if (sizeof(T) == 2) v = bf2f(f2bf(v));   // statistics of the values as stored
            v -= cj[j];
            csum[j] += v; csq[j] += v * v;
          }
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int m = m0 + wm * TM * 16 + i * 16 + fq * 4 + e;
            float v = acc[i][j][e] + bj[j];
            if (sizeof(T) == 2) v = bf2f(f2bf(v));
            v -= cj[j];
            if (m >= p.M) v = 0.f;
            csum[j] += v; csq[j] += v * v;
          }
      }
    }
  }
  if (p.stats) {
    __syncthreads();                             // staging region is reused below (block-wide)
    // All cross-lane sums first, a wait, then the (lane-masked) LDS writes.  With the natural loop - shuffles of column
    // group j, `if (fq == 0)` LDS write, shuffles of group j + 1, ... - the bf16 register-staged instantiations wrote a
    // sum of squares that was 1 ulp to 1 % off for a few 16-column groups of about one tile in 1500, differently from
    // launch to launch (scripts/diag/stats_race.py: 39 of 39 launches differed; output bits and plain sums unaffected).
    // A/B builds narrowed the trigger to the EXEC-masked write between the ds_bpermute pairs: the same loop with the write
    // done by all lanes (they hold the same totals) is clean, as is this order; separate ds_write_b32 instead of a
    // ds_write2, a wait after every write, a block barrier in front of the epilogue or a 256-register budget are not.
    // The generated code around the mask (s_and_saveexec / ds_write2st64 / s_or exec / ds_bpermute) reads correct, so
    // this is recorded as a hardware-level ordering effect between EXEC updates and ds_bpermute, avoided by construction.
    // Regression test: tests/test_gpu_conv.py::test_fused_statistics_are_the_same_from_launch_to_launch.
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      csum[j] += __shfl_xor(csum[j], 16, 64); csum[j] += __shfl_xor(csum[j], 32, 64);
      csq[j] += __shfl_xor(csq[j], 16, 64); csq[j] += __shfl_xor(csq[j], 32, 64);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      if (fq == 0) {
        red[(wm * 2 + 0) * BN + wn * WCOLS + j * 16 + fr] = csum[j];
        red[(wm * 2 + 1) * BN + wn * WCOLS + j * 16 + fr] = csq[j];
      }
    __syncthreads();
    for (int c = tid; c < 2 * BN; c += 256) {
      const int which = c / BN, col = c - which * BN;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) s += red[(w * 2 + which) * BN + col];
      if (n0 + col < p.dC) p.stats[((size_t)(p.stats_row0 + tm) * 2 + which) * p.dC + n0 + col] = s;
    }
  }
  if constexpr (BNR) {
    {                    // lanes that share columns (lane % LPR) -> one value per wave, waves added in fixed order
      __syncthreads();
#pragma unroll
      for (int e = 0; e < VEC; ++e) {            // (shuffles, wait, then the writes: see the statistics above)
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) { bnr.s[e] += __shfl_xor(bnr.s[e], o, 64); bnr.q[e] += __shfl_xor(bnr.q[e], o, 64); }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if (lane < LPR) {
          red[(wm * 2 + 0) * BN + wn * WCOLS + lane * VEC + e] = bnr.s[e];
          red[(wm * 2 + 1) * BN + wn * WCOLS + lane * VEC + e] = bnr.q[e];
        }
      __syncthreads();
      for (int c = tid; c < 2 * BN; c += 256) {
        const int which = c / BN, col = c - which * BN;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) s += red[(w * 2 + which) * BN + col];
        if (n0 + col < p.dC)
          p.bnr_slab[((size_t)(p.bnr_row0 + p.stats_row0 + tm) * 2 + which) * p.dC + n0 + col] = s;
      }
    }
  }
}

extern "C" int dpa_bn_stats_rows(const void* x, int dtype, long long M, int C, int cs, float* slab, int rowsPerBlock,
                                 int nblocks, const float* shift, hipStream_t s);

// Sum the split-K partials and apply the epilogue the conv kernel skipped: bias, residual addend, ReLU, store as T.
template <typename T>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const ConvParams p) {
  constexpr int VEC = 16 / sizeof(T);
  const int chunks = p.dC / VEC;                       // dC is a multiple of VEC (checked on the host)
  const long long total = (long long)p.M * chunks;
  T* __restrict__ dst = reinterpret_cast<T*>(p.dst);
  const T* __restrict__ addend = reinterpret_cast<const T*>(p.addend);
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int m = (int)(idx / chunks), col = (int)(idx - (long long)m * chunks) * VEC;
    float v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = 0.f;
    for (int s = 0; s < p.splitK; ++s) {
      const float* src = p.ws + ((size_t)s * p.M + m) * p.dC + col;
#pragma unroll
      for (int e = 0; e < VEC; e += 4) {
        const float4 t = *reinterpret_cast<const float4*>(src + e);
        v[e] += t.x; v[e + 1] += t.y; v[e + 2] += t.z; v[e + 3] += t.w;
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = v[e] * (p.oscale ? p.oscale[col + e] : 1.f) + (p.bias ? p.bias[col + e] : 0.f);
    const size_t off = (size_t)m * (size_t)p.d_cs + (size_t)(p.d_co + col);
    if (addend) {
      Vec16<T> av; av.load(addend + off);
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] += av.get(e);
    }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    Vec16<T> ov;
#pragma unroll
    for (int e = 0; e < VEC; ++e) ov.set(e, v[e]);
    ov.store(dst + off);
  }
}

// Per-device split-K workspace.  Allocated by dpa_init_workspaces() when the library is loaded - not inside a launch
// path, which a hipGraph capture may be recording; if it is missing while the stream is being captured the launch
// runs unsplit.  DPA_SPLITK_WS_MB sizes it (default 256 MiB of the 288 GB; a problem whose partial sums do not fit
// takes fewer splits).
static size_t splitk_ws_bytes() {
  static const size_t v = [] { const char* e = getenv("DPA_SPLITK_WS_MB"); long mb = e ? atol(e) : 0; return (size_t)(mb > 0 ? mb : 256) << 20; }();
  return v;
}
static float* splitk_workspace(hipStream_t s) {
  static std::mutex mu;
  static std::map<int, float*> ws;
  std::lock_guard<std::mutex> lk(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  auto it = ws.find(dev);
  if (it == ws.end()) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (s && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return nullptr;
    float* ptr = nullptr;
    if (hipMalloc((void**)&ptr, splitk_ws_bytes()) != hipSuccess) { (void)hipGetLastError(); ptr = nullptr; }
    it = ws.emplace(dev, ptr).first;
  }
  return it->second;
}
extern "C" int dpa_init_ce_workspace();
// Called once per device by ops/native.py right after the library is loaded on a machine with a GPU.
DPA_API int dpa_init_workspaces() {
  const int a = splitk_workspace(nullptr) ? 0 : 1;
  const int b = dpa_init_ce_workspace();
  return a | (b << 1);
}

// fp32 arithmetic mode of the convolution kernels (fprop / dgrad here, wgrad in conv_wgrad.hip): 0 = exact fp32 MFMA,
// 1 = x6 (fp32 through 6 bf16 MFMAs per product, see X6 above; the default).  Initial value from DPA_FP32_MODE=exact|x6; the
// Python layer switches it with dpa_set_fp32_mode (between launches; a captured hipGraph keeps the mode it was captured in).
static int g_fp32_mode = -1;
DPA_API int dpa_get_fp32_mode() {
  if (g_fp32_mode < 0) { const char* e = getenv("DPA_FP32_MODE"); g_fp32_mode = (e && e[0] == 'e') ? 0 : 1; }
  return g_fp32_mode;
}
DPA_API void dpa_set_fp32_mode(int mode) { g_fp32_mode = mode ? 1 : 0; }
static int fp32_x6() { return dpa_get_fp32_mode(); }
static int uni_ct() {       // DPA_CONV_UNICT=0: always launch the instantiations with both address paths (A/B switch)
  static const int v = [] { const char* e = getenv("DPA_CONV_UNICT"); return e ? atoi(e) : 1; }();
  return v;
}
static int x6_reg() {       // x6 also for the register-staged (short reduction) variant; DPA_X6_REG=0: A/B switch
  static const int v = [] { const char* e = getenv("DPA_X6_REG"); return (e && e[0] == '0') ? 0 : 1; }();
  return v;
}

// rows of the stats slab written by the launches of the current dpa_conv_igemm call (reported back in ConvParams)
static thread_local int g_stats_rows = 0;
static thread_local int g_class_rows = 0;       // dgrad: rows summed over the parity classes
bool dpa_x6p_eligible(const ConvParams& p);                       // conv_x6p.hip
int dpa_x6p_launch(ConvParams& p, int dgrad, hipStream_t s);

template <typename T, int WM, int WN, int TM, int TN>
static void launch_cfg(const ConvParams& pin, int dgrad, hipStream_t s) {
  ConvParams p = pin;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  p.tilesM = (p.M - p.m_off + BM - 1) / BM;
  p.tilesN = (p.dC + BN - 1) / BN;
  if (p.stats_row0 + p.tilesM > g_stats_rows) g_stats_rows = p.stats_row0 + p.tilesM;
  // staging policy: LDS-DMA + 2 stage buffers pays when there are enough K-tiles to pipeline; short reductions
  // (1x1 convs on few channels) run more blocks per CU with the single-buffer register-staged variant.
  static const int no_uni = [] { const char* e = getenv("DPA_CONV_UNI"); return (e && e[0] == '0') ? 1 : 0; }();
  p.no_uni = no_uni;
  static const int policy = [] { const char* e = getenv("DPA_CONV_STAGING"); return !e ? 0 : (e[0] == 'r' ? 1 : (e[0] == 'g' ? 2 : 0)); }();
  constexpr int BKE = 8 * (16 / (int)sizeof(T));
  const int nkt = (p.Kg + BKE - 1) / BKE;
  // (fp32 in x6 mode on the 256 x 64 / 128 x 32 tiles (WM = 4): their register-staged variant only exists in exact fp32 arithmetic - it would
  //  spill with the operand splits - so short reductions take the LDS-DMA x6 variant there too: the ResNet stem conv,
  //  7 K-tiles, 3.93 -> 2.77 ms at batch 1024)
  const bool x6_tall_short = sizeof(T) == 4 && WM == 4 && fp32_x6() && nkt >= 2;
  const bool use_glds = policy == 2 || (policy == 0 && (nkt >= 8 || x6_tall_short));
  size_t lds = (size_t)(BM + BN) * 128 * (use_glds ? 2 : 1);
  size_t epi = (size_t)(4 * 16 * (TN * 16 + 4) + WM * 2 * BN) * 4;
  if (epi > lds) lds = epi;
  // Split-K for skinny problems (Linear layers: 64 x 9216 x 4096 has 64 tiles for 256 CUs): every split streams its
  // share of the weights, raw partial sums go to a workspace and a small kernel finishes the epilogue.
  static const int allow_split = [] { const char* e = getenv("DPA_CONV_SPLITK"); return (e && e[0] == '0') ? 0 : 1; }();
  const int ntiles = p.tilesM * p.tilesN;
  p.splitK = 1; p.kt_per_split = nkt; p.ws = nullptr;
  constexpr int OCC = (BM * BN <= 64 * 64) ? 4 : (BM == 256 ? 1 : (BN <= 32 ? 3 : 2));     // resident blocks per CU
  const int slots = 256 * OCC;
  // Measured policy (scripts/bench_conv_layers.py --batch 64, scripts/bench_zoo.py): Linear-shaped problems (a few
  // hundred rows) always gain; small-spatial convs gain only with a long reduction (K >= 4096), and when the BN
  // statistics have to be rebuilt from the output afterwards only in bf16 (in fp32 that extra pass eats the gain).
  const bool linear_like = p.M <= 256 && ntiles * 2 <= 256 && nkt >= 32;
  const bool deep_conv = p.Kg >= 4096 && ntiles * 2 <= slots && (!p.stats || sizeof(T) == 2);
  if (allow_split && !p.bnr_x && p.sh == 1 && p.sw == 1 && p.m_off == 0 && (linear_like || deep_conv) &&
      (long long)p.Kg * p.dC >= (1 << 20)) {      // enough weight traffic to pay for the extra launches
    int want = slots / ntiles;
    if (want > nkt / 8) want = nkt / 8;
    const size_t need = (size_t)p.M * p.dC * sizeof(float);
    while (want > 1 && need * want > splitk_ws_bytes()) --want;
    float* ws = want > 1 ? splitk_workspace(s) : nullptr;
    if (ws) {
      p.kt_per_split = (nkt + want - 1) / want;
      p.splitK = (nkt + p.kt_per_split - 1) / p.kt_per_split;
      p.ws = ws;
    }
  }
  if (p.splitK > 1 && lds < (size_t)(BM + BN) * 128 * 2) lds = (size_t)(BM + BN) * 128 * 2;   // split kernels use LDS-DMA staging
  dim3 grid(ntiles * p.splitK), block(256);
  // The kernel's uniform-tap condition, evaluated here: those launches take instantiations whose K loop contains only
  // that address path (no loop-invariant branch, half the code in the loop: +2 % on the ResNet-50 step, measured A/B
  // on one box with DPA_CONV_UNICT=0|1).
  const bool host_uni = uni_ct() && (p.sC % BKE == 0) && p.nR * p.nS >= 1 && p.nR * p.nS <= 32 && !p.no_uni;
  if (p.splitK > 1) {   // split-K (always a long reduction, hence LDS-DMA staging); a separate instantiation so the
                        // single-pass kernels keep their register allocation
    if (sizeof(T) == 4 && fp32_x6()) {
      if constexpr (sizeof(T) == 4) {
        if (dgrad) hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, true, true, true, true>), grid, block, lds, s, p);
        else       hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, false, true, true, true>), grid, block, lds, s, p);
      }
    } else if (dgrad) hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, true, true, true>), grid, block, lds, s, p);
    else              hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, false, true, true>), grid, block, lds, s, p);
  } else if (sizeof(T) == 4 && fp32_x6() && (use_glds || (x6_reg() && WM != 4))) {   // (256 x 64 register-staged x6 would spill)
    if constexpr (sizeof(T) == 4) {
      if ((use_glds || WM == 4) && host_uni) {
        if (dgrad && p.bnr_x) hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, true, true, false, true, true, true>), grid, block, lds, s, p);
        else if (dgrad) hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, true, true, false, true, true>), grid, block, lds, s, p);
        else       hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, false, true, false, true, true>), grid, block, lds, s, p);
      } else if (use_glds || WM == 4) {
        if (dgrad && p.bnr_x) hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, true, true, false, true, false, true>), grid, block, lds, s, p);
        else if (dgrad) hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, true, true, false, true>), grid, block, lds, s, p);
        else       hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, false, true, false, true>), grid, block, lds, s, p);
      } else if constexpr (WM != 4) {     // (not even instantiated for the 256 x 64 tile)
        if (dgrad && p.bnr_x) hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, true, false, false, true, false, true>), grid, block, lds, s, p);
        else if (dgrad) hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, true, false, false, true>), grid, block, lds, s, p);
        else       hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, false, false, false, true>), grid, block, lds, s, p);
      }
    }
  } else if (use_glds && host_uni) {     // LDS-DMA staging, double-buffered, uniform-tap address path only (the usual case)
    if (dgrad && p.bnr_x) hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, true, true, false, false, true, true>), grid, block, lds, s, p);
    else if (dgrad) hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, true, true, false, false, true>), grid, block, lds, s, p);
    else       hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, false, true, false, false, true>), grid, block, lds, s, p);
  } else if (use_glds) {     // LDS-DMA staging, both address paths compiled in (stem, odd channel counts)
    if (dgrad && p.bnr_x) hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, true, true, false, false, false, true>), grid, block, lds, s, p);
    else if (dgrad) hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, true, true>), grid, block, lds, s, p);
    else       hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, false, true>), grid, block, lds, s, p);
  } else if (host_uni) {     // register staging, single buffer (short reductions), uniform-tap path only
    if (dgrad && p.bnr_x) hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, true, false, false, false, true, true>), grid, block, lds, s, p);
    else if (dgrad) hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, true, false, false, false, true>), grid, block, lds, s, p);
    else       hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, false, false, false, false, true>), grid, block, lds, s, p);
  } else {            // register staging, single buffer
    if (dgrad && p.bnr_x) hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, true, false, false, false, false, true>), grid, block, lds, s, p);
    else if (dgrad) hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, true, false>), grid, block, lds, s, p);
    else       hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, false, false>), grid, block, lds, s, p);
  }
  if (p.splitK > 1) {
    const long long work = (long long)p.M * (p.dC / (16 / (int)sizeof(T)));
    const int rb = (int)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048);
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(rb), dim3(256), 0, s, p);
    if (p.stats)      // the fused-epilogue statistics, rebuilt from the stored output (one slab row per M-tile)
      dpa_bn_stats_rows(reinterpret_cast<const T*>(p.dst) + p.d_co, sizeof(T) == 2 ? 1 : 0, p.M, p.dC, p.d_cs, p.stats, BM,
                        p.tilesM, p.stats_shift, s);
  }
}

static int big_min() {      // fewest 128 x 128 tiles for which the big-tile config is chosen (smaller problems: 64 x 64)
  static const int v = [] { const char* e = getenv("DPA_CONV_BIGMIN"); return e ? atoi(e) : 384; }();
  return v;
}
static int tile_override() {
  static const int v = [] { const char* e = getenv("DPA_CONV_TILE"); return e ? atoi(e) : 0; }();
  return v;
}

// Rows [0, split) go to 128 x 128 tiles, rows [split, M) to 64 x 64 tiles; 0 = no split.  Shared by the launcher and
// by dpa_conv_stats_rows (the stats slab has one row per M-tile of either size).
static int tail_split_rows(int M, int dC, int elem_bytes, int BM = 128, int BN = 128) {
  static const int enabled = [] { const char* e = getenv("DPA_CONV_TAILSPLIT"); return (e && e[0] == '0') ? 0 : 1; }();
  static const int rem_env = [] { const char* e = getenv("DPA_CONV_TAILSPLIT_REM"); return e ? atoi(e) : 0; }();
  if (!enabled) return 0;
  // measured (scripts/bench_conv_layers.py): fp32 gains up to a 320-tile tail (-2 % conv time), the LDS/L2-bound bf16
  // kernel only up to 128 (small tiles cost it more per FLOP)
  const int max_rem = rem_env > 0 ? rem_env : (elem_bytes == 4 ? 320 : 128);
  // The big-tile kernels (128 x 128, 256 x 64) are resident twice per CU and the dispatcher refills freed slots two at a
  // time, so the quantum that matters is a ROUND of 2 x 256 tiles (measured: 512 tiles 318 us, 784 tiles 595 us, 1024
  // tiles 592 us in fp32, scripts/conv_size_sweep.py).  The 64 x 64 kernel is resident 5 times per CU (1280 slots) and a
  // big tile is 4 small ones, so a last round of up to 320 big tiles fits ONE round of small tiles, which takes about
  // 0.75 of a big round (fp32): split whenever the partial round holds at most `max_rem` tiles.
  const int ROUND = 512;
  const long long tm = (M + BM - 1) / BM, tn = (dC + BN - 1) / BN, tiles = tm * tn;
  const long long full = tiles / ROUND;               // whole rounds
  const long long rem = tiles - full * ROUND;         // tiles of the last, partial round
  if (full < 1 || rem == 0 || rem > max_rem) return 0;
  const long long big_rows = (full * ROUND) / tn;     // M-tiles kept as big tiles
  if (big_rows <= 0 || big_rows >= tm) return 0;
  return (int)(big_rows * BM);
}

template <typename T>
static void launch_tiles(ConvParams& p, int dgrad, hipStream_t s) {
  p.fd_nW = make_fastdiv(p.cls_nw > 0 ? p.cls_nw : 1); p.fd_nH = make_fastdiv(p.cls_nh > 0 ? p.cls_nh : 1);
  p.fd_sC = make_fastdiv(p.sC); p.fd_nS = make_fastdiv(p.nS > 0 ? p.nS : 1);
  // tile selection: narrow outputs get a tall tile; small problems get small tiles so that the grid
  // still covers the 256 CUs.
  p.m_off = 0; p.stats_row0 = 0;
  if constexpr (sizeof(T) == 4) {
    // fp32, x6 arithmetic, filter available as pre-split planes: the kernel whose B operand costs no VALU (conv_x6p.hip)
    if (fp32_x6() && dpa_x6p_eligible(p)) {
      const int rows = dpa_x6p_launch(p, dgrad, s);
      if (rows > g_stats_rows) g_stats_rows = rows;
      return;
    }
  }
  const long long big_tiles = (long long)((p.M + 127) / 128) * ((p.dC + 127) / 128);
  if (p.dC <= 32)                     launch_cfg<T, 4, 1, 2, 2>(p, dgrad, s);   // 128 x 32
  else if (p.dC <= 64) {
    if ((p.M + 255) / 256 >= 256) {                                             // 256 x 64 (+ 64 x 64 tail)
      const int split = tail_split_rows(p.M, p.dC, (int)sizeof(T), 256, 64);
      if (split > 0) {
        ConvParams a = p; a.M = split;
        launch_cfg<T, 4, 1, 4, 4>(a, dgrad, s);
        ConvParams b = p; b.m_off = split; b.stats_row0 = split / 256;
        launch_cfg<T, 2, 2, 2, 2>(b, dgrad, s);
      } else {
        launch_cfg<T, 4, 1, 4, 4>(p, dgrad, s);
      }
    } else                            launch_cfg<T, 2, 2, 2, 2>(p, dgrad, s);   // 64 x 64
  } else if (tile_override() == 64)   launch_cfg<T, 2, 2, 2, 2>(p, dgrad, s);
  else if (tile_override() == 12864)  launch_cfg<T, 2, 2, 4, 2>(p, dgrad, s);   // 128 x 64
  else if (big_tiles >= big_min()) {
    // 128 x 128 tiles run in rounds of 512 (2 per CU); a nearly empty last round costs a full round.  Give whole
    // rounds to the big tiles and the few remaining rows to a second launch of 64 x 64 tiles (quarter-size work
    // items that spread over the chip): e.g. 1568 tiles = 3.06 rounds -> 3 rounds + a short tail instead of 4.
    const int split = tail_split_rows(p.M, p.dC, (int)sizeof(T));
    if (split > 0) {
      ConvParams a = p; a.M = split;
      launch_cfg<T, 2, 2, 4, 4>(a, dgrad, s);
      ConvParams b = p; b.m_off = split; b.stats_row0 = split / 128;
      launch_cfg<T, 2, 2, 2, 2>(b, dgrad, s);
    } else {
      launch_cfg<T, 2, 2, 4, 4>(p, dgrad, s);
    }
  } else                              launch_cfg<T, 2, 2, 2, 2>(p, dgrad, s);   // 64 x 64
}

template <typename T>
static int launch_conv(ConvParams& p, int dgrad, hipStream_t s) {
  constexpr int VEC = 16 / sizeof(T);
  if (p.sC % VEC || p.s_cs % VEC || p.s_co % VEC || p.dC % VEC || p.d_cs % VEC || p.d_co % VEC) return 1;
  if ((size_t)p.sN * p.sH * p.sW >= 0x7fffffffull || (size_t)p.sN * p.dH * p.dW >= 0x7fffffffull) return 2;
  if (!dgrad) {
    if (p.addend2) return 4;
    if (p.bnr2_x) return 5;
    p.cls_h0 = p.cls_w0 = 0; p.cls_nh = p.dH; p.cls_nw = p.dW;
    p.base_y = p.base_x = 0; p.tap_r0 = p.tap_s0 = 0; p.nR = p.R; p.nS = p.S;
    p.M = p.sN * p.dH * p.dW;
    p.Kg = p.R * p.S * p.sC;
    launch_tiles<T>(p, 0, s);
    return 0;
  }
  // dgrad: one dense sub-problem per (a, b) parity class; a stride-s conv costs 1x the fprop work instead of s^2 x
  auto make_class = [&](int a, int b, ConvParams& q) -> bool {      // false: the class has no pixels
    q = p;
    q.cls_h0 = ((a - p.ph) % p.sh + p.sh) % p.sh;
    q.cls_w0 = ((b - p.pw) % p.sw + p.sw) % p.sw;
    q.cls_nh = q.cls_h0 < p.dH ? (p.dH - q.cls_h0 + p.sh - 1) / p.sh : 0;
    q.cls_nw = q.cls_w0 < p.dW ? (p.dW - q.cls_w0 + p.sw - 1) / p.sw : 0;
    if (q.cls_nh == 0 || q.cls_nw == 0) return false;
    q.tap_r0 = a; q.tap_s0 = b;
    q.nR = a < p.R ? (p.R - a + p.sh - 1) / p.sh : 0;
    q.nS = b < p.S ? (p.S - b + p.sw - 1) / p.sw : 0;
    q.base_y = (q.cls_h0 + p.ph - a) / p.sh;
    q.base_x = (q.cls_w0 + p.pw - b) / p.sw;
    q.M = p.sN * q.cls_nh * q.cls_nw;
    q.Kg = q.nR * q.nS * p.sC;          // 0 taps: the kernel only runs its epilogue (zeros + addend)
    return true;
  };
  if (p.addend2 || p.bnr2_x) {
    // compact addend, second BatchNorm reduction: only conv_x6p.hip's epilogues know them.  Dry run over the classes BEFORE
    // anything is launched.
    const int refuse = p.addend2 ? 4 : 5;
    if (sizeof(T) != 4 || !fp32_x6()) return refuse;
    if (p.addend2 && (p.addend || p.a2_cs % 4 || p.a2_H != (p.dH + 1) / 2 || p.a2_W != (p.dW + 1) / 2)) return 4;
    if (p.bnr2_x && (!p.bnr_x || !p.bnr_bits || p.bnr2_cs % 4)) return 5;
    for (int a = 0; a < p.sh; ++a)
      for (int b = 0; b < p.sw; ++b) {
        ConvParams q;
        if (!make_class(a, b, q)) continue;
        q.m_off = 0; q.stats_row0 = 0;
        if (q.Kg == 0 || !dpa_x6p_eligible(q)) return refuse;
      }
  }
  for (int a = 0; a < p.sh; ++a)
    for (int b = 0; b < p.sw; ++b) {
      ConvParams q;
      if (!make_class(a, b, q)) continue;
      // ... and when the addend IS the destination (in-place accumulation into an existing gradient) a class without
      // taps leaves its pixels unchanged: nothing to launch (3 of the 4 classes of a 1x1 stride-2 downsample conv)
      if (q.Kg == 0 && p.addend == p.dst) {
        if (p.bnr_x) return 3;              // (the fused BN reduction has to visit every pixel; the caller checks R >= sh)
        continue;
      }
      const int row_base = g_class_rows;    // slab rows of the classes launched so far (fused BN-backward reduction)
      q.bnr_row0 = row_base;
      g_stats_rows = 0;
      launch_tiles<T>(q, 1, s);
      g_class_rows = row_base + g_stats_rows;
    }
  g_stats_rows = g_class_rows;
  return 0;
}

// dtype: 0 = fp32, 1 = bf16.  Returns 0 on success, non-zero for an unsupported geometry.
DPA_API int dpa_conv_igemm(ConvParams* p, int dtype, int dgrad, hipStream_t s) {
  ConvParams q = *p;
  g_stats_rows = 0; g_class_rows = 0;
  const int rc = dtype == 1 ? launch_conv<bf16_t>(q, dgrad, s) : launch_conv<float>(q, dgrad, s);
  p->stats_rows_used = g_stats_rows;
  if (rc) return rc;                       // 1, 2: unsupported geometry (nothing was launched)
  const int e = dpa_launch_rc();           // a rejected launch (LDS size, grid size, ...) must not pass silently
  return e ? 1000 + e : 0;
}
// rows of the per-block stats slab a launch with these dims will write (tilesM of the chosen config)
DPA_API int dpa_conv_stats_rows(int M, int dC, int dtype) {
  const int eb = dtype == 1 ? 2 : 4;
  const long long big_tiles = (long long)((M + 127) / 128) * ((dC + 127) / 128);
  if (dC <= 32) return (M + 127) / 128;
  if (dC <= 64) {
    if ((M + 255) / 256 < 256) return (M + 63) / 64;
    const int split = tail_split_rows(M, dC, eb, 256, 64);
    return split > 0 ? split / 256 + (M - split + 63) / 64 : (M + 255) / 256;
  }
  if (tile_override() == 64) return (M + 63) / 64;
  if (tile_override() == 12864) return (M + 127) / 128;
  if (big_tiles >= big_min()) {
    const int split = tail_split_rows(M, dC, eb);
    return split > 0 ? split / 128 + (M - split + 63) / 64 : (M + 127) / 128;
  }
  return (M + 63) / 64;
}
DPA_API int dpa_conv_params_size() { return (int)sizeof(ConvParams); }
