// Implicit-GEMM convolution weight-gradient for gfx950 (K3 of SURVEY.md §2.5; replaces cuDNN wgrad reached
// from loss.backward() at /root/reference/classif.py:59).
//
//   dW[k][r][s][c] = sum_{img,ho,wo} dy[img,ho,wo,k] * x[img, ho*sh-ph+r, wo*sw-pw+s, c]
//   GEMM view: D[m = k][n = (r,s,c)] = sum_{pix} A[pix][m] * B[pix][n]   -- the reduction index (pixel) is the
//   STRIDED dimension of both operands, so tiles are staged pixel-major ([pix][channels], 16 B per lane) and
//   the MFMA fragments are read transposed out of LDS: ds_read_b64_tr_b16 for bf16 (gfx950 hardware
//   transpose), plain ds_read_b32 for the exact-fp32 16x16x4 MFMA (one float per lane per operand).
//   The output is tiny and the reduction is huge (up to N*112*112 pixels), so the grid is split over the
//   pixel dimension (split-K) into fp32 partial slabs that a deterministic second kernel sums (no float
//   atomics: bitwise reproducible, cdna_hip_programming.md Guideline 12).
#include "conv_common.h"

struct WgradParams {
  const void* dy;    // [N, Ho, Wo, k_cs] (+k_co)   Cout = K
  const void* x;     // [N, H, W, c_cs] (+c_co)     Cin = C
  float* out;        // splitK == 1: dW [K][R][S][C] fp32 ; else slabs [splitK][K][R*S*C]
  int N, H, W, C, c_cs, c_co;
  int Ho, Wo, K, k_cs, k_co;
  int R, S, sh, sw, ph, pw;
  int P;             // N*Ho*Wo pixels (reduction length)
  int Ng;            // R*S*C
  int tilesM, tilesN, splitK, pixPerSplit;
  int accumulate;    // splitK == 1 only: out += result
  int xcd_order;     // XCD-contiguous block order (set by wg_launch)
  FastDiv fd_Wo, fd_Ho, fd_C, fd_S;
};

template <typename T> struct WgLds;
template <> struct WgLds<bf16_t> { static constexpr int PAD = 32; };   // bytes; shifts each pixel row by 8 banks
template <> struct WgLds<float>  { static constexpr int PAD = 64; };   // (pitch/4) % 32 == 16

// Bank swizzle of the bf16-plane tiles read with ds_read_b64_tr_b16 (x6 path, tile rows of 256 or 512 bytes = 128 / 256
// columns).  LDS serves 32 lanes x 8 bytes per pass over 64 banks x 4 bytes.  The padded layout (pitch 288 bytes) had
//   * transposed reads: a 16-lane group addresses 4 pixel rows x 32 bytes, and the second group of the pass sits 8 rows
//     further - 8 x 288 bytes is a multiple of the 256-byte bank line, so both groups hit the same banks;
//   * B-operand stores: a pass writes 4 consecutive rows x 64 bytes whose bank ranges overlap pairwise at that pitch
//   (PMC: SQ_LDS_BANK_CONFLICT = 43 % of SQ_LDS_IDX_ACTIVE).  Now: pitch = the row width (every row starts at bank 0) and
// the 32-byte block index inside a row is XOR-ed with swz(row) = 2 * (row & 3) | (row >> 3 & 1).  A transposed read then
// touches blocks blk ^ {0, 2, 4, 6} ^ {0, 1} - eight distinct 8-bank groups; a B store pass touches block pairs
// (2i ^ s, 2i + 1 ^ s) for s = 0, 2, 4, 6 - four distinct pairs; an A store pass is one full row.  Fragments (32 bytes)
// and store chunks (8 bytes) never straddle a block.  Writers and readers both go through tr_off().
template <int ROW_BYTES>
struct TrTile {
  static constexpr bool SWZ = ROW_BYTES == 256 || ROW_BYTES == 512;
  static constexpr int PITCH = SWZ ? ROW_BYTES : ROW_BYTES + WgLds<bf16_t>::PAD;
  static __device__ __forceinline__ int off(int row, int byte_in_row) {
    if constexpr (SWZ) return row * PITCH + (byte_in_row ^ ((((row & 3) << 1) | ((row >> 3) & 1)) << 5));
    else return row * PITCH + byte_in_row;
  }
};

// X6 (fp32 operands only): the staging threads split every value into three bf16 planes (conv_common.h, split3_pair)
// on the way into LDS - once per element - and the K loop runs the bf16 fragment path on the six plane pairs.
// Resident blocks per CU the register budget is set for: 3, or 2 for x6 (plane splits) and for the bf16 64 x 256 tile (at
// 3 it spilled 26 VGPRs to scratch: stem wgrad -24 %, 56x56 256->64 -9 % at 2).
#define WG_MINB(T, WN, TN, X6) ((X6 || (sizeof(T) == 2 && WN == 4 && TN == 4)) ? 2 : 3)
// ABL (diagnostic builds only, -DDPA_WG_DIAG; results are WRONG, timing only): bit 0 = no operand split, bit 1 = no global
// loads after the first tile, bit 2 = no MFMAs, bit 3 = no LDS stores after the first tile.
template <typename T, int WM, int WN, int TM, int TN, bool X6 = false, int ABL = 0>
__global__ void __launch_bounds__(256, WG_MINB(T, WN, TN, X6)) conv_wgrad_kernel(const WgradParams p) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int VEC = 16 / sizeof(T);
  constexpr int BKP = 8 * VEC;                          // pixels per K-tile (64 bf16 / 32 fp32)
  constexpr int ESZ = X6 ? 2 : (int)sizeof(T);          // bytes per element in LDS
  constexpr int LPAD = X6 ? WgLds<bf16_t>::PAD : WgLds<T>::PAD;
  typedef TrTile<BM * 2> TA;                          // x6 plane tiles (bank-swizzled where the row width allows)
  typedef TrTile<BN * 2> TB;
  // native bf16: swizzled only for the 128 x 128 tile (-5..-10 % on those layers, PMC: 43 % -> 0 conflict cycles); the
  // 64 x 256 instance is at its register limit and the per-store XOR addresses push it into scratch (+50 % time)
  constexpr bool TR = X6 || (sizeof(T) == 2 && BM == 128 && BN == 128);
  constexpr int PITCH_A = TR ? TA::PITCH : BM * ESZ + LPAD;
  constexpr int PITCH_B = TR ? TB::PITCH : BN * ESZ + LPAD;
  constexpr int PLANE_A = BKP * PITCH_A, PLANE_B = BKP * PITCH_B;   // X6: three planes per operand
  constexpr int CPR_A = BM / VEC, CPR_B = BN / VEC;     // 16-byte chunks per pixel row
  constexpr int A_PASSES = BKP * CPR_A / 256;
  constexpr int RPP_A = 256 / CPR_A;                    // pixel rows per pass (A)
  constexpr int TPR = 256 / BKP;                        // threads per pixel row (B): 4 bf16 / 8 fp32
  constexpr int B_CHUNKS = CPR_B / TPR;                 // chunks of ONE pixel row staged by a thread
  static_assert(WM * WN == 4, "4 waves per block");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsA = smem;
  char* ldsB = smem + (X6 ? 3 : 1) * PLANE_A;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware order: blocks that share an XCD (blockIdx % 8) get a contiguous run of the split-major work list, i.e.
  // one or two pixel ranges, so each XCD's L2 streams 1/8 of dy and x instead of all of it (8 x less HBM/MALL traffic:
  // the 3x3 layers had a 39 % L2 miss rate, profiles/r1_pmc_wgrad_14x14_256_3x3_fp32.txt)
  // Measured (scripts/bench_conv_layers.py, profiles/r1_pmc_wgrad_28x28_*): L2 hit rate 10 % -> 85 %, bf16 wgrad time
  // -8 %; neutral for the MFMA-bound fp32 kernel.
  const bool xcd_order = p.xcd_order != 0;               // chosen on the host (wg_launch)
  int bid = xcd_order ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int tilesMN = p.tilesM * p.tilesN;
  const int z = bid / tilesMN;
  bid -= z * tilesMN;
  const int tm = bid / p.tilesN, tn = bid - tm * p.tilesN;
  const int m0 = tm * BM, n0 = tn * BN;
  const int pix_begin = z * p.pixPerSplit;
  const int pix_end = min(p.P, pix_begin + p.pixPerSplit);

  const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);

  // A operand (dy, dense [pixel][cout]): this thread stages chunk ca of pixel rows ra0 + RPP_A*i
  const int ca = tid % CPR_A, ra0 = tid / CPR_A;
  const int a_col = m0 + ca * VEC;
  const bool a_ok = a_col < p.K;
  // B operand (x gathered): a thread owns ONE pixel row of the tile and B_CHUNKS channel chunks of it, so the
  // (img, ho, wo) decomposition is done once per thread per K-step; every chunk has a fixed tap (r, s, c) whose
  // address delta is precomputed (the first version decomposed the pixel once per chunk and was VALU bound).
  const int brow = tid / TPR, bsub = tid % TPR;
  int b_dr[B_CHUNKS], b_ds[B_CHUNKS], b_delta[B_CHUNKS];
#pragma unroll
  for (int i = 0; i < B_CHUNKS; ++i) {
    const int nidx = n0 + (bsub + TPR * i) * VEC;
    if (nidx < p.Ng) {
      const uint32_t tap = fd_div((uint32_t)nidx, p.fd_C);
      const int c = nidx - (int)tap * p.C;
      const int r = (int)fd_div(tap, p.fd_S);
      const int s = (int)tap - r * p.S;
      b_dr[i] = r; b_ds[i] = s;
      b_delta[i] = (r * p.W + s) * p.c_cs + p.c_co + c;
    } else {
      b_dr[i] = -0x40000000; b_ds[i] = 0; b_delta[i] = 0;     // fails the bounds check below -> zeros
    }
  }

  uint4 ra[A_PASSES], rb[B_CHUNKS];
  auto load_tile = [&](int pix0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int pix = pix0 + ra0 + RPP_A * i;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (a_ok && pix < pix_end) v = *reinterpret_cast<const uint4*>(dy + (size_t)pix * p.k_cs + p.k_co + a_col);
      ra[i] = v;
    }
    const int pix = pix0 + brow;
    const bool pin = pix < pix_end;
    const uint32_t t1 = fd_div((uint32_t)(pin ? pix : 0), p.fd_Wo);
    const int wo = (pin ? pix : 0) - (int)t1 * p.Wo;
    const uint32_t img = fd_div(t1, p.fd_Ho);
    const int ho = (int)t1 - (int)img * p.Ho;
    const int h0 = ho * p.sh - p.ph, w0 = wo * p.sw - p.pw;
    const T* base = x + ((size_t)img * (p.H * p.W) + (ptrdiff_t)(h0 * p.W + w0)) * p.c_cs;
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (pin && (unsigned)(h0 + b_dr[i]) < (unsigned)p.H && (unsigned)(w0 + b_ds[i]) < (unsigned)p.W)
        v = *reinterpret_cast<const uint4*>(base + b_delta[i]);
      rb[i] = v;
    }
  };
  auto store_tile = [&]() __attribute__((always_inline)) {
    if constexpr (X6) {
#pragma unroll
      for (int i = 0; i < A_PASSES; ++i) {
        uint32_t lo[3], hi[3];
        if constexpr (ABL & 1) { lo[0] = ra[i].x; lo[1] = ra[i].y; lo[2] = ra[i].x; hi[0] = ra[i].z; hi[1] = ra[i].w; hi[2] = ra[i].z; }
        else {
        split3_pair(__uint_as_float(ra[i].x), __uint_as_float(ra[i].y), lo);
        split3_pair(__uint_as_float(ra[i].z), __uint_as_float(ra[i].w), hi);
        }
#pragma unroll
        for (int part = 0; part < 3; ++part)
          *reinterpret_cast<uint2*>(ldsA + part * PLANE_A + TA::off(ra0 + RPP_A * i, ca * 8)) = make_uint2(lo[part], hi[part]);
      }
#pragma unroll
      for (int i = 0; i < B_CHUNKS; ++i) {
        uint32_t lo[3], hi[3];
        if constexpr (ABL & 1) { lo[0] = rb[i].x; lo[1] = rb[i].y; lo[2] = rb[i].x; hi[0] = rb[i].z; hi[1] = rb[i].w; hi[2] = rb[i].z; }
        else {
        split3_pair(__uint_as_float(rb[i].x), __uint_as_float(rb[i].y), lo);
        split3_pair(__uint_as_float(rb[i].z), __uint_as_float(rb[i].w), hi);
        }
#pragma unroll
        for (int part = 0; part < 3; ++part)
          *reinterpret_cast<uint2*>(ldsB + part * PLANE_B + TB::off(brow, (bsub + TPR * i) * 8)) = make_uint2(lo[part], hi[part]);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i)
      *reinterpret_cast<uint4*>(ldsA + (TR ? TA::off(ra0 + RPP_A * i, ca * 16) : (ra0 + RPP_A * i) * PITCH_A + ca * 16)) = ra[i];
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i)
      *reinterpret_cast<uint4*>(ldsB + (TR ? TB::off(brow, (bsub + TPR * i) * 16) : brow * PITCH_B + (bsub + TPR * i) * 16)) = rb[i];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  if (pix_begin < pix_end) {
    load_tile(pix_begin);
    store_tile();
    __syncthreads();
    for (int pix0 = pix_begin; pix0 < pix_end; pix0 += BKP) {
      const bool more = pix0 + BKP < pix_end;
      if (more && !(ABL & 2)) load_tile(pix0 + BKP);
      if constexpr (X6) {
        // one 32-pixel k-step per tile; the planes are bf16 tiles, read with the same transposed loads as the bf16 path
        typedef __attribute__((address_space(3))) bf16x4 lds_b4;
        const int q = fr >> 2, pp = fr & 3;
        const int krow = 8 * fq + q;
        bf16x8 fb[TN][3];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int part = 0; part < 3; ++part) {
            const char* pl = ldsB + part * PLANE_B;
            const int cb = (wn * TN * 16 + j * 16 + 4 * pp) * 2;
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(pl + TB::off(krow, cb)));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(pl + TB::off(krow + 4, cb)));
            fb[j][part] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          bf16x8 fa[3];
#pragma unroll
          for (int part = 0; part < 3; ++part) {
            const char* pl = ldsA + part * PLANE_A;
            const int cb = (wm * TM * 16 + i * 16 + 4 * pp) * 2;
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(pl + TA::off(krow, cb)));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(pl + TA::off(krow + 4, cb)));
            fa[part] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
          for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if constexpr (ABL & 4) asm volatile("" :: "v"(fa[kX6A[t]]), "v"(fb[j][kX6B[t]]));
              else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kX6A[t]], fb[j][kX6B[t]], acc[i][j], 0, 0, 0);
            }
        }
      } else if constexpr (sizeof(T) == 2) {
        // bf16: per 32-pixel k-step, lane needs k = 8*fq + j (j = 0..7) of column fr: two transposed
        // 4x16 block reads; lane (4q+pp) of each 16-lane group addresses block row q, columns 4pp..4pp+3.
        typedef __attribute__((address_space(3))) bf16x4 lds_b4;
        const int q = fr >> 2, pp = fr & 3;
#pragma unroll
        for (int ks = 0; ks < BKP / 32; ++ks) {
          bf16x8 fa[TM], fb[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int col = (wm * TM * 16 + i * 16 + 4 * pp) * 2;
            const int krow = ks * 32 + 8 * fq + q;
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(ldsA + (TR ? TA::off(krow, col) : krow * PITCH_A + col)));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(ldsA + (TR ? TA::off(krow + 4, col) : (krow + 4) * PITCH_A + col)));
            fa[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int col = (wn * TN * 16 + j * 16 + 4 * pp) * 2;
            const int krow = ks * 32 + 8 * fq + q;
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(ldsB + (TR ? TB::off(krow, col) : krow * PITCH_B + col)));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(ldsB + (TR ? TB::off(krow + 4, col) : (krow + 4) * PITCH_B + col)));
            fb[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
      } else {
        // fp32: 16x16x4, lane holds A[m = fr][k = fq] = tile[k][m]
#pragma unroll
        for (int ks = 0; ks < BKP / 4; ++ks) {
          float fa[TM], fb[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i)
            fa[i] = *reinterpret_cast<const float*>(ldsA + (ks * 4 + fq) * PITCH_A + (wm * TM * 16 + i * 16 + fr) * 4);
#pragma unroll
          for (int j = 0; j < TN; ++j)
            fb[j] = *reinterpret_cast<const float*>(ldsB + (ks * 4 + fq) * PITCH_B + (wn * TN * 16 + j * 16 + fr) * 4);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
      }
      __syncthreads();
      if (more) {
        if constexpr (!(ABL & 8)) store_tile();
        __syncthreads();
      }
    }
  }

  // epilogue: D[row = m0 + wm*TM*16 + i*16 + fq*4 + e][col = n0 + wn*TN*16 + j*16 + fr]
  float* out = p.out + (size_t)z * p.K * p.Ng;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * TN * 16 + j * 16 + fr;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = m0 + wm * TM * 16 + i * 16 + fq * 4 + e;
        if (row < p.K && col < p.Ng) {
          float* o = out + (size_t)row * p.Ng + col;
          *o = (p.accumulate && p.splitK == 1) ? (*o + acc[i][j][e]) : acc[i][j][e];
        }
      }
    }
}

__device__ uint4 dpa_zero_chunk_wg[1];

// ---- fp32 (x6) weight gradient, second structure ("x6q"): 8 waves, ONE block per CU, plane tiles double-buffered in LDS,
// ONE barrier per K-tile, staging interleaved with the matrix work --------------------------------------------------------
// Measured on the kernel above (profiles/r3_wgrad_ablation.txt, 14x14 256->256 3x3): its phases ADD - without the global
// loads it takes 75 % of its time, without the split + LDS stores 63 %, without MFMAs 60 %: MFMA phase, barrier, store
// phase, barrier, and the matrix pipe idles through every store phase (50 % busy).  Here the registers of a thread hold
// tile t + 1 (loaded during iteration t - 1) while the MFMAs of tile t run; their split and LDS stores go into the OTHER
// stage buffer between those MFMAs (every wave does both: two waves per SIMD, like two blocks of the kernel above), and the
// global loads of tile t + 2 are issued at the top of the iteration into the second register set.
// Tiles: 256 x 128 (WM = 4, WN = 2) or 128 x 256 (WM = 2, WN = 4), 64 x 64 per wave: per MFMA 25 % fewer split
// instructions and LDS-store bytes than the 128 x 128 tile.  LDS: 2 x 72 KB.
// (The same kernel on v_mfma_f32_32x32x16_bf16 - half the matrix instructions, three times the free issue slots - ran
// 13 % fewer cycles at a 13 % lower clock, i.e. the same wall time: profiles/r3_wgrad_x6q.txt.  Not kept.)
template <int WM, int WN>
__global__ void __launch_bounds__(512, 2) conv_wgrad_x6q_kernel(const WgradParams p) {
  static_assert(WM * WN == 8, "8 waves per block");
  constexpr int TM = 4, TN = 4, NT = 512, BKP = 32;
  constexpr int BM = WM * 64, BN = WN * 64;
  typedef TrTile<BM * 2> TA;
  typedef TrTile<BN * 2> TB;
  static_assert(TA::SWZ && TB::SWZ, "swizzled plane tiles");
  constexpr int PLANE_A = BKP * TA::PITCH, PLANE_B = BKP * TB::PITCH;
  constexpr int STAGE = 3 * (PLANE_A + PLANE_B);
  constexpr int CPR_A = BM / 4, CPR_B = BN / 4;        // 16-byte (4 x fp32) chunks per pixel row
  constexpr int A_PASSES = BKP * CPR_A / NT;           // 4 (BM = 256) or 2 (BM = 128)
  constexpr int RPP_A = NT / CPR_A;                    // pixel rows per pass (A)
  constexpr int TPR = NT / BKP;                        // 16 threads per pixel row (B)
  constexpr int B_CHUNKS = CPR_B / TPR;                // 2 (BN = 128) or 4 (BN = 256)
  constexpr int ITEMS = A_PASSES + B_CHUNKS;           // 16-byte chunks a thread stages per K-tile (6)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int bid = p.xcd_order ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int tilesMN = p.tilesM * p.tilesN;
  const int z = bid / tilesMN;
  bid -= z * tilesMN;
  const int tm = bid / p.tilesN, tn = bid - tm * p.tilesN;
  const int m0 = tm * BM, n0 = tn * BN;
  const int pix_begin = z * p.pixPerSplit;
  const int pix_end = min(p.P, pix_begin + p.pixPerSplit);
  const float* __restrict__ dy = reinterpret_cast<const float*>(p.dy);
  const float* __restrict__ x = reinterpret_cast<const float*>(p.x);

  const int ca = tid % CPR_A, ra0 = tid / CPR_A;
  const int a_col = m0 + ca * 4;
  const bool a_ok = a_col < p.K;
  const int brow = tid / TPR, bsub = tid % TPR;
  int b_dr[B_CHUNKS], b_ds[B_CHUNKS], b_delta[B_CHUNKS];
#pragma unroll
  for (int i = 0; i < B_CHUNKS; ++i) {
    const int nidx = n0 + (bsub + TPR * i) * 4;
    if (nidx < p.Ng) {
      const uint32_t tap = fd_div((uint32_t)nidx, p.fd_C);
      const int c = nidx - (int)tap * p.C;
      const int r = (int)fd_div(tap, p.fd_S);
      const int s = (int)tap - r * p.S;
      b_dr[i] = r; b_ds[i] = s;
      b_delta[i] = (r * p.W + s) * p.c_cs + p.c_co + c;
    } else {
      b_dr[i] = -0x40000000; b_ds[i] = 0; b_delta[i] = 0;     // fails the bounds check below -> zeros
    }
  }

  // tile at pix0 -> registers (zeros beyond pix_end / outside the image / beyond K, Ng).  UNCONDITIONAL loads through an
  // address select (the zero chunk for everything masked): a load under a branch makes hipcc wait vmcnt(0) where the
  // registers of the OTHER set are consumed, i.e. for the loads just issued (cdna_hip_programming.md §5, trap (c)).
  // In four parts, so that an iteration can spread them over its first MFMA segments.
  const float* const zero = reinterpret_cast<const float*>(dpa_zero_chunk_wg);
  struct BRow { const float* base; int h0, w0; bool pin; };
  auto load_A = [&](int pix0, uint4 (&r)[ITEMS]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const int pix = pix0 + ra0 + RPP_A * i;
      const float* ptr = (a_ok && pix < pix_end) ? dy + ((size_t)pix * p.k_cs + p.k_co + a_col) : zero;
      r[i] = *reinterpret_cast<const uint4*>(ptr);
    }
  };
  auto decompose_B = [&](int pix0) __attribute__((always_inline)) {
    BRow b;
    const int pix = pix0 + brow;
    b.pin = pix < pix_end;
    const uint32_t t1 = fd_div((uint32_t)(b.pin ? pix : 0), p.fd_Wo);
    const int wo = (b.pin ? pix : 0) - (int)t1 * p.Wo;
    const uint32_t img = fd_div(t1, p.fd_Ho);
    const int ho = (int)t1 - (int)img * p.Ho;
    b.h0 = ho * p.sh - p.ph; b.w0 = wo * p.sw - p.pw;
    b.base = x + ((size_t)img * (p.H * p.W) + (ptrdiff_t)(b.h0 * p.W + b.w0)) * p.c_cs;
    return b;
  };
  auto load_B = [&](const BRow& b, int i0, int i1, uint4 (&r)[ITEMS]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      const bool ok = b.pin && (unsigned)(b.h0 + b_dr[i]) < (unsigned)p.H && (unsigned)(b.w0 + b_ds[i]) < (unsigned)p.W;
      const float* ptr = ok ? b.base + b_delta[i] : zero;
      r[A_PASSES + i] = *reinterpret_cast<const uint4*>(ptr);
    }
  };
  auto load_tile = [&](int pix0, uint4 (&r)[ITEMS]) __attribute__((always_inline)) {
    load_A(pix0, r);
    const BRow b = decompose_B(pix0);
    load_B(b, 0, B_CHUNKS, r);
  };
  // one staged chunk (4 fp32) -> three bf16 planes in the stage buffer, in two halves (one split3_pair each) + the stores
  auto store_item = [&](int it, const uint32_t (&lo)[3], const uint32_t (&hi)[3], char* stage) __attribute__((always_inline)) {
    if (it < A_PASSES) {
#pragma unroll
      for (int part = 0; part < 3; ++part)
        *reinterpret_cast<uint2*>(stage + part * PLANE_A + TA::off(ra0 + RPP_A * it, ca * 8)) = make_uint2(lo[part], hi[part]);
    } else {
#pragma unroll
      for (int part = 0; part < 3; ++part)
        *reinterpret_cast<uint2*>(stage + 3 * PLANE_A + part * PLANE_B + TB::off(brow, (bsub + TPR * (it - A_PASSES)) * 8)) =
            make_uint2(lo[part], hi[part]);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  typedef __attribute__((address_space(3))) bf16x4 lds_b4;
  const int q = fr >> 2, pp = fr & 3;
  const int krow = 8 * fq + q;
  auto read_fa = [&](const char* ldsA, int i, bf16x8 (&fa)[3]) __attribute__((always_inline)) {
#pragma unroll
    for (int part = 0; part < 3; ++part) {
      const char* pl = ldsA + part * PLANE_A;
      const int cb = (wm * 64 + i * 16 + 4 * pp) * 2;
      bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(pl + TA::off(krow, cb)));
      bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(pl + TA::off(krow + 4, cb)));
      fa[part] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  // Iteration t.  MFMAs of the tile in `cur` in 16 segments of 6 (row tile i, quarter s), each fenced off from the next
  // (sched_barrier(0): hipcc's scheduler, left the whole iteration, clusters 60 MFMAs and 100 vector instructions) and each
  // with its own share of the other work, interleaved inside the segment by sched_group_barrier:
  //   row tile 0: the global loads of tile t + 2 into `rl` (addresses in the first three segments, B loads in the last two);
  //   row tiles 1-3: split + LDS stores of `rs` (tile t + 1, loaded a whole iteration ago) into `nxt`: one split3_pair per
  //                  segment, the three 8-byte stores of a chunk behind its second pair;
  //   segment 1 of every row tile: the fragment reads of the next row tile.
  // One barrier at the end.
  auto iteration = [&](int pix_next2, const char* cur, char* nxt, uint4 (&rl)[ITEMS], uint4 (&rs)[ITEMS]) __attribute__((always_inline)) {
    const char* ldsA = cur;
    const char* ldsB = cur + 3 * PLANE_A;
    bf16x8 fb[TN][3], fa[2][3];
    read_fa(ldsA, 0, fa[0]);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int part = 0; part < 3; ++part) {
        const char* pl = ldsB + part * PLANE_B;
        const int cb = (wn * 64 + j * 16 + 4 * pp) * 2;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(pl + TB::off(krow, cb)));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b4*)(pl + TB::off(krow + 4, cb)));
        fb[j][part] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    __builtin_amdgcn_sched_barrier(0);
    BRow brw;
    uint32_t lo3[3], hi3[3];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        int nread = 0, nvalu = 0, nwrite = 0, nvmem = 0;
        if (sg == 1 && i + 1 < TM) { read_fa(ldsA, i + 1, fa[(i + 1) & 1]); nread = 6; }
#pragma unroll
        for (int m = 6 * sg; m < 6 * sg + 6; ++m) {
          const int t = m / TN, j = m % TN;
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i & 1][kX6A[t]], fb[j][kX6B[t]], acc[i][j], 0, 0, 0);
        }
        if (i == 0) {
          if (sg == 0) { load_A(pix_next2, rl); nvalu = 6 * A_PASSES; nvmem = A_PASSES; }
          if (sg == 1) { brw = decompose_B(pix_next2); nvalu = 24; }
          if (sg == 2) { load_B(brw, 0, B_CHUNKS / 2, rl); nvalu = 5 * B_CHUNKS; nvmem = B_CHUNKS / 2; }
          if (sg == 3) { load_B(brw, B_CHUNKS / 2, B_CHUNKS, rl); nvalu = 5 * B_CHUNKS; nvmem = B_CHUNKS / 2; }
        } else {
          const int pair = (i - 1) * 4 + sg;          // 0 .. 11: chunk pair / 2, half pair & 1
          const int it = pair >> 1;
          if ((pair & 1) == 0) split3_pair(__uint_as_float(rs[it].x), __uint_as_float(rs[it].y), lo3);
          else {
            split3_pair(__uint_as_float(rs[it].z), __uint_as_float(rs[it].w), hi3);
            store_item(it, lo3, hi3, nxt);
            nwrite = 3;
          }
          nvalu = 10;
        }
        // the segment's interleave: fragment reads first, then vector instructions spread behind the six MFMAs (an MFMA
        // holds the issue port for 8 of its 16 cycles), loads / LDS stores in the second half
        if (nread) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          if (m == 1 && nvalu > 12) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          if (m == 3 && nvalu > 16) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          if (m >= 3 && nwrite) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          if (m >= 6 - nvmem && nvmem) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  };

  if (pix_begin < pix_end) {
    uint4 r0[ITEMS], r1[ITEMS];
    char* buf0 = smem;
    char* buf1 = smem + STAGE;
    load_tile(pix_begin, r0);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      uint32_t lo[3], hi[3];
      split3_pair(__uint_as_float(r0[it].x), __uint_as_float(r0[it].y), lo);
      split3_pair(__uint_as_float(r0[it].z), __uint_as_float(r0[it].w), hi);
      store_item(it, lo, hi, buf0);
    }
    load_tile(pix_begin + BKP, r1);
    __syncthreads();
    // (tiles beyond pix_end load as zeros and are staged like any other - no branch inside an iteration; the last
    // iteration's stores go into a buffer nobody reads any more)
    for (int pix0 = pix_begin; pix0 < pix_end; pix0 += 2 * BKP) {
      iteration(pix0 + 2 * BKP, buf0, buf1, r0, r1);
      if (pix0 + BKP < pix_end) iteration(pix0 + 3 * BKP, buf1, buf0, r1, r0);
    }
  }

  float* out = p.out + (size_t)z * p.K * p.Ng;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * 64 + j * 16 + fr;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = m0 + wm * 64 + i * 16 + fq * 4 + e;
        if (row < p.K && col < p.Ng) {
          float* o = out + (size_t)row * p.Ng + col;
          *o = (p.accumulate && p.splitK == 1) ? (*o + acc[i][j][e]) : acc[i][j][e];
        }
      }
    }
}

// Sum the split-K slabs in a fixed order (deterministic) into the gradient arena.  A block owns 64 float4 columns and
// splits the slabs over 4 lanes of z (then a fixed-order LDS combine), so small filters with many slabs still fill the chip.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                           long long n, int splitK, int accumulate) {
  __shared__ float4 red[4][64];
  const int col = threadIdx.x & 63, zl = threadIdx.x >> 6;
  const long long i = ((long long)blockIdx.x * 64 + col) * 4;      // n is a multiple of 4
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n) {
#pragma unroll 4
    for (int z = zl; z < splitK; z += 4) {
      const float4 t = *reinterpret_cast<const float4*>(slabs + (size_t)z * n + i);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
  }
  red[zl][col] = s;
  __syncthreads();
  if (zl == 0 && i < n) {
#pragma unroll
    for (int l = 1; l < 4; ++l) { const float4 t = red[l][col]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    if (accumulate) {
      const float4 o = *reinterpret_cast<const float4*>(out + i);
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    *reinterpret_cast<float4*>(out + i) = s;
  }
}

template <typename T, int WM, int WN, int TM, int TN>
static void wg_launch(WgradParams& p, float* grad, float* workspace, hipStream_t s) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, VEC = 16 / sizeof(T), BKP = 8 * VEC;
  p.tilesM = (p.K + BM - 1) / BM;
  p.tilesN = (p.Ng + BN - 1) / BN;
  const int tiles = p.tilesM * p.tilesN;
  const int ksteps = (p.P + BKP - 1) / BKP;
  // At most one resident round of blocks.  (Rounding UP here put every 3x3 layer at 513-576 blocks with 512 slots:
  // a second, almost empty round that cost 20-25 % - PMC showed the same cycles per MFMA as the 1x1 layers but lower
  // residency.)  Slab traffic grows with the split.
  static const int target_env = [] { const char* e = getenv("DPA_WGRAD_BLOCKS"); int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
  const bool x6 = sizeof(T) == 4 && dpa_get_fp32_mode() == 1;
  const int target = target_env ? target_env : ((x6 || WG_MINB(T, WN, TN, false) == 2) ? 512 : 768);
  int split = target / tiles;       // 768 = 3 resident blocks per CU (the kernel is built for 3 waves/SIMD; x6: 2)
  if (split > ksteps / 8) split = ksteps / 8;
  if (split < 1) split = 1;
  if (!workspace) split = 1;
  int steps_per = (ksteps + split - 1) / split;
  split = (ksteps + steps_per - 1) / steps_per;
  p.splitK = split;
  p.pixPerSplit = steps_per * BKP;
  // XCD-contiguous block order (see the kernel).  DPA_WGRAD_XCD=0|1 overrides the measured default.
  static const int xcd_env = [] { const char* e = getenv("DPA_WGRAD_XCD"); return e ? atoi(e) : -1; }();
  const bool xcd_order = xcd_env >= 0 ? xcd_env != 0 : true;
  p.xcd_order = xcd_order ? 1 : 0;
  p.out = split == 1 ? grad : workspace;
  if (x6) {
    if constexpr (sizeof(T) == 4) {
      const size_t lds = (size_t)BKP * 3 * (TrTile<BM * 2>::PITCH + TrTile<BN * 2>::PITCH);
#ifdef DPA_WG_DIAG
      if constexpr (WM == 2 && WN == 2 && TM == 4 && TN == 4) {
        const char* e = getenv("DPA_WG_ABL");
        const int abl = e ? atoi(e) : 0;
#define DPA_ABL_CASE(N) if (abl == N) { hipLaunchKernelGGL((conv_wgrad_kernel<T, WM, WN, TM, TN, true, N>), dim3(tiles * split), dim3(256), lds, s, p); goto launched; }
        DPA_ABL_CASE(1) DPA_ABL_CASE(2) DPA_ABL_CASE(3) DPA_ABL_CASE(4) DPA_ABL_CASE(8) DPA_ABL_CASE(9) DPA_ABL_CASE(12) DPA_ABL_CASE(14) DPA_ABL_CASE(15)
#undef DPA_ABL_CASE
      }
#endif
      hipLaunchKernelGGL((conv_wgrad_kernel<T, WM, WN, TM, TN, true>), dim3(tiles * split), dim3(256), lds, s, p);
#ifdef DPA_WG_DIAG
      launched:;
#endif
    }
  } else {
    const size_t lds = (sizeof(T) == 2 && BM == 128 && BN == 128)
                           ? (size_t)BKP * (TrTile<BM * 2>::PITCH + TrTile<BN * 2>::PITCH)
                           : (size_t)BKP * ((BM + BN) * sizeof(T) + 2 * WgLds<T>::PAD);
    hipLaunchKernelGGL((conv_wgrad_kernel<T, WM, WN, TM, TN>), dim3(tiles * split), dim3(256), lds, s, p);
  }
  if (split > 1) {
    const long long n = (long long)p.K * p.Ng;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, s, workspace, grad, n,
                       split, p.accumulate);
  }
}

// which: 1 = DPA_WGRAD_X6Q (default 1): the 8-wave double-buffered fp32 (x6) kernel for filters of more than 64 output
// channels (0 = the 4-wave kernel everywhere).  v >= 0 sets the knob and returns the previous value; v < 0 only reads it.
DPA_API int dpa_wgrad_knob(int which, int v) {
  static int cur[2] = {0, [] { const char* e = getenv("DPA_WGRAD_X6Q"); return e ? atoi(e) : 1; }()};
  if (which != 1) return -1;
  const int prev = cur[which];
  if (v >= 0) cur[which] = v;
  return prev;
}

template <int WM, int WN>
static int x6q_launch(WgradParams& p, float* grad, float* workspace, hipStream_t s) {
  constexpr int BM = WM * 64, BN = WN * 64, BKP = 32;
  constexpr size_t LDS = (size_t)2 * 3 * BKP * (TrTile<BM * 2>::PITCH + TrTile<BN * 2>::PITCH);      // 144 KB
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_x6q_kernel<WM, WN>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
  if (attr != hipSuccess) {      // no 144 KB of dynamic LDS on this device / runtime: the caller takes the 4-wave kernel
    (void)hipGetLastError();
    return -1;
  }
  p.tilesM = (p.K + BM - 1) / BM;
  p.tilesN = (p.Ng + BN - 1) / BN;
  const int tiles = p.tilesM * p.tilesN;
  const int ksteps = (p.P + BKP - 1) / BKP;
  // One block per CU.  The split-K factor fills whole rounds of the 256 block slots: the fewest rounds (at most three)
  // whose last one is at least 93 % full (512 -> 512 3x3: 72 tiles, 3 x 72 = 216 blocks leave 16 % of the CUs idle for the
  // whole kernel, 7 x 72 = 504 blocks fill two rounds to 98 %).
  int split = 256 / tiles;
  for (int r = 1; r <= 3; ++r) {
    const int sr = 256 * r / tiles;
    if (sr >= 1 && sr * tiles * 100 >= 93 * 256 * r) { split = sr; break; }
  }
  if (split > ksteps / 8) split = ksteps / 8;
  if (split < 1) split = 1;
  if (!workspace) split = 1;
  int steps_per = (ksteps + split - 1) / split;
  split = (ksteps + steps_per - 1) / steps_per;
  p.splitK = split;
  p.pixPerSplit = steps_per * BKP;
  p.xcd_order = 1;
  p.out = split == 1 ? grad : workspace;
  hipLaunchKernelGGL((conv_wgrad_x6q_kernel<WM, WN>), dim3(tiles * split), dim3(512), LDS, s, p);
  if (split > 1) {
    const long long n = (long long)p.K * p.Ng;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, s, workspace, grad, n,
                       split, p.accumulate);
  }
  return 0;
}

template <typename T>
static int wg_dispatch(WgradParams& p, float* grad, float* workspace, hipStream_t s) {
  constexpr int VEC = 16 / sizeof(T);
  p.P = p.N * p.Ho * p.Wo;
  p.Ng = p.R * p.S * p.C;
  if (p.C % VEC || p.c_cs % VEC || p.c_co % VEC || p.K % VEC || p.k_cs % VEC || p.k_co % VEC) return 1;
  if ((size_t)p.N * p.H * p.W >= 0x7fffffffull || (size_t)p.P >= 0x7fffffffull) return 2;
  p.fd_Wo = make_fastdiv(p.Wo); p.fd_Ho = make_fastdiv(p.Ho);
  p.fd_C = make_fastdiv(p.C); p.fd_S = make_fastdiv(p.S);
  if constexpr (sizeof(T) == 4) {
    // fp32 in x6 arithmetic, more than 64 filters: the 8-wave kernel with 256 x 128 or 128 x 256 tiles, where at most a
    // tenth of a tile dimension is padding (128 -> 128 3x3, Ng = 1152 = 4.5 tiles of 256: 3 % slower than the 4-wave kernel) (ResNet-50: everything but the 64-filter layers and the 64-channel 1x1 convs)
    if (dpa_get_fp32_mode() == 1 && dpa_wgrad_knob(1, -1) > 0 && p.Ng >= 128) {
      const int padM256 = (p.K + 255) / 256 * 256 - p.K, padN128 = (p.Ng + 127) / 128 * 128 - p.Ng;
      const int padM128 = (p.K + 127) / 128 * 128 - p.K, padN256 = (p.Ng + 255) / 256 * 256 - p.Ng;
      int rc = -1;
      if (p.K >= 256 && padM256 * 10 <= p.K && padN128 * 10 <= p.Ng) rc = x6q_launch<4, 2>(p, grad, workspace, s);
      else if (p.K > 64 && p.Ng >= 256 && padM128 * 10 <= p.K && padN256 * 10 <= p.Ng) rc = x6q_launch<2, 4>(p, grad, workspace, s);
      if (rc >= 0) return rc;
    }
  }
  if (p.Ng <= 64 && p.K >= 256 && sizeof(T) == 4)
                       wg_launch<T, 4, 1, 4, 4>(p, grad, workspace, s);   // 256 x  64: 1x1 expand convs on 64 channels (fp32: the
                                                                          // operand splits are paid per staged element, and this
                                                                          // tile does 2.3x the MFMAs per element of 64 x 64)
  else if (p.Ng <= 64) wg_launch<T, 2, 2, 2, 2>(p, grad, workspace, s);   //  64 x  64
  else if (p.K <= 32 && p.Ng % 192 == 0 && p.Ng % 256 != 0)
                       wg_launch<T, 1, 4, 2, 3>(p, grad, workspace, s);   //  32 x 192: DenseNet growth convs (32 filters, 3x3 x 128):
  else if (p.K <= 32)  wg_launch<T, 1, 4, 2, 4>(p, grad, workspace, s);   //  32 x 256  no MFMA rows spent on padding filters
  else if (p.K <= 64 && p.Ng % 192 == 0 && p.Ng % 256 != 0)
                       wg_launch<T, 1, 4, 4, 3>(p, grad, workspace, s);   //  64 x 192: 3x3 x 64 channels without padding
  else if (p.K <= 64)  wg_launch<T, 1, 4, 4, 4>(p, grad, workspace, s);   //  64 x 256: narrow Cout, all waves share the dy tile
  else                 wg_launch<T, 2, 2, 4, 4>(p, grad, workspace, s);   // 128 x 128
  return 0;
}

// workspace: fp32 scratch of at least dpa_conv_wgrad_workspace() floats, or null to force splitK = 1.
DPA_API int dpa_conv_wgrad(WgradParams* p, float* grad, float* workspace, int dtype, hipStream_t s) {
  WgradParams q = *p;
  const int rc = dtype == 1 ? wg_dispatch<bf16_t>(q, grad, workspace, s) : wg_dispatch<float>(q, grad, workspace, s);
  if (rc) return rc;
  const int e = dpa_launch_rc();
  return e ? 1000 + e : 0;
}
// upper bound on workspace floats for a layer: split <= 1024 tiles-worth, i.e. <= ~1024*BM*BN + K*Ng
DPA_API long long dpa_conv_wgrad_workspace(int K, int Ng) {
  const int bm = (Ng <= 64 && K >= 256) ? 256 : ((Ng > 64 && K <= 32) ? 32 : ((Ng <= 64 || K <= 64) ? 64 : 128));
  const int bn = Ng <= 64 ? 64 : (K <= 64 ? ((Ng % 192 == 0 && Ng % 256 != 0) ? 192 : 256) : 128);
  const long long tiles = (long long)((K + bm - 1) / bm) * ((Ng + bn - 1) / bn);
  long long split = (1024 + tiles - 1) / tiles;      // upper bound of what wg_launch may choose
  const long long tq1 = (long long)((K + 255) / 256) * ((Ng + 127) / 128), tq2 = (long long)((K + 127) / 128) * ((Ng + 255) / 256);
  const long long split_q = 768 / (tq1 < tq2 ? tq1 : tq2);      // x6q_launch: at most three rounds of 256 blocks
  if (split < split_q) split = split_q;
  return split * (long long)K * Ng;
}
DPA_API int dpa_wgrad_params_size() { return (int)sizeof(WgradParams); }
