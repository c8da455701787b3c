// fp32 convolution forward / data-gradient through the bf16 matrix pipe with PRE-SPLIT weights ("x6p").
//
// x6 arithmetic (conv_common.h): every fp32 value is the exact sum of three bf16 parts, a product is six bf16 x bf16
// MFMA products with fp32 accumulation.  The general x6 kernel (conv_igemm.hip) splits BOTH operands after every LDS
// fragment read - 343 VALU instructions next to 96 MFMAs per K-tile and wave, and on this part VALU time adds to MFMA
// time (profiles/r1_micro_valu_mfma_coissue.txt), so it ran at half the matrix rate.  Here:
//   * the filter is split ONCE per step into three bf16 planes (split3_planes_kernel below) and staged as planes, so
//     the B operand costs no VALU at all;
//   * the four waves of a block are stacked along M (WM = 4, WN = 1): a wave owns 32 rows x the whole tile width, so
//     each activation fragment is split once and used against TN = 8 (4) filter fragments - 2 splits per 96 (48) MFMAs.
// What remains is ~80 VALU instructions per K-tile and wave: the kernel is bound by the bf16 MFMA rate / 6.
//
// Staging: LDS-DMA (global_load_lds_dwordx4), two stage buffers, one barrier per K-tile, like the GLDS path of
// conv_igemm.hip.  A (activations, fp32): 128 rows x 128 B, chunk ^ ((row >> 1) & 7) swizzle.  B: three planes of
// BN rows x 64 B (32 bf16 of the reduction), 16-byte slot ^ ((-(row >> 2)) & 3) swizzle - conflict-free for the
// ds_read_b128 lane groups of the 16x16x32 fragment (derivation in docs/KERNELS.md).
// Only the uniform-tap address path (Cin % 32 == 0: every layer of the zoo but the stems) - other shapes keep the
// general kernel.
//
// Planes are in natural element order.  A lane of an A fragment holds 8 consecutive reduction elements (two adjacent
// 16-byte chunks of the fp32 row), so one ds_read_b128 of a plane (8 bf16) is the matching MFMA operand.
//
// MFMA shape: v_mfma_f32_16x16x32_bf16, 2 x TN tiles per wave.  (A v_mfma_f32_32x32x16_bf16 variant - 1 x TN/2 tiles,
// two k-steps per K-tile, half as many matrix instructions - measured 7 % SLOWER over the ResNet-50 layers:
// profiles/r2_x6p_variants.txt.  So did 8 waves per block and one block per CU.)
#include "conv_common.h"
#include "x6_frag.h"

__device__ uint4 dpa_zero_chunk_x6p[4];
DPA_API int dpa_x6p_knob(int which, int v);
bool dpa_x6s_eligible(const ConvParams& p);                       // conv_x6s.hip
int dpa_x6s_launch(ConvParams& p, int dgrad, hipStream_t s);

__device__ __forceinline__ void split3_frag_rn(const uint4& lo, const uint4& hi, uint4 (&pl)[3]) {      // filter planes
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  uint32_t o[4][3];
#pragma unroll
  for (int q = 0; q < 4; ++q) split3_pair_rn(__uint_as_float(w[2 * q]), __uint_as_float(w[2 * q + 1]), o[q]);
#pragma unroll
  for (int part = 0; part < 3; ++part) pl[part] = make_uint4(o[0][part], o[1][part], o[2][part], o[3][part]);
}

// n fp32 values (n % 32 == 0) -> three bf16 planes of n values each.  One thread per 8 values.
__global__ void __launch_bounds__(256) split3_planes_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, long long n) {
  const long long nchunks = n >> 3;
  for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < nchunks; c += (long long)gridDim.x * 256) {
    const float4 lo = *reinterpret_cast<const float4*>(w + c * 8);
    const float4 hi = *reinterpret_cast<const float4*>(w + c * 8 + 4);
    uint4 pl[3];
    split3_frag_rn(make_uint4(__float_as_uint(lo.x), __float_as_uint(lo.y), __float_as_uint(lo.z), __float_as_uint(lo.w)),
                   make_uint4(__float_as_uint(hi.x), __float_as_uint(hi.y), __float_as_uint(hi.z), __float_as_uint(hi.w)), pl);
#pragma unroll
    for (int part = 0; part < 3; ++part) *reinterpret_cast<uint4*>(out + (long long)part * n + c * 8) = pl[part];
  }
}
DPA_API int dpa_split3_planes(const float* w, void* out, long long n, hipStream_t s) {
  if (n <= 0 || (n & 31)) return 1;
  hipLaunchKernelGGL(split3_planes_kernel, dim3(dpa_grid(n / 8, 256)), dim3(256), 0, s, w, (bf16_t*)out, n);
  return dpa_launch_rc();
}

// ---- batched per-step weight preparation (nn/wprep.py) ------------------------------------------------------------------
// Every conv layer needs, once per step, its filter as three bf16 planes (fprop), the transposed filter [C][RS][K]
// (dgrad) and that one's planes: two or three tiny launches per layer (DenseNet-121: 359 launches and 1.2 ms of a 25 ms
// step at batch 64).  This kernel does all of them in ONE launch from a device-side segment table.
// One thread per 8 consecutive OUTPUT elements; a block of 256 chunks lies inside one segment (segments start at multiples
// of 256 chunks), found by binary search over the first chunk of each segment.
// transposed: 0 = planes of the filter as it is, 1 = transpose by strided gather, 2 = transpose through an LDS tile.
struct WPrepSeg {
  long long first_chunk;   // index of this segment's first 8-element chunk in the launch
  long long n;             // elements (a multiple of 8)
  const void* src;         // filter [K][RS][C] in the parameter arena (or its bf16 shadow)
  void* dst;               // transposed: [C][RS][K] in the element type (or null); direct: unused
  void* planes;            // three bf16 planes of the (transposed) filter, [3][n] (fp32 only, or null)
  int K, RS, C, transposed;
};
template <typename T>
__global__ void __launch_bounds__(256) weight_prep_kernel(const WPrepSeg* __restrict__ segs, int nsegs, long long total_chunks) {
  // every segment starts at a multiple of 256 chunks, so a block lies inside ONE segment (found by its first thread's chunk)
  __shared__ float tile[64][33];
  const long long g0 = (long long)blockIdx.x * 256;
  int lo = 0, hi = nsegs - 1;
  while (lo < hi) {                          // last segment whose first_chunk <= g0 (block-uniform)
    const int mid = (lo + hi + 1) >> 1;
    if (segs[mid].first_chunk <= g0) lo = mid; else hi = mid - 1;
  }
  const WPrepSeg sg = segs[lo];
  const int t = threadIdx.x;
  const long long lc0 = g0 - sg.first_chunk;         // first local chunk of this block
  if (lc0 * 8 >= sg.n) return;                       // padding between segments (block-uniform)
  const T* __restrict__ src = reinterpret_cast<const T*>(sg.src);
  alignas(16) T v[8];
  long long i0;                                      // first of this thread's 8 consecutive OUTPUT elements
  if (sg.transposed == 2) {
    // tiled transpose (fp32, K % 64 == 0, C % 32 == 0): the block owns 64 filters x 32 channels of one tap.  Coalesced
    // reads along c, LDS, coalesced 32-byte writes along k.  (The gather below reads 4 bytes per 128-byte line: 451 us
    // for AlexNet's 58 M fully-connected weights.)
    const int tilesC = sg.C / 32, tilesK = sg.K / 64;
    const long long tl = lc0 >> 8;                   // tile index: k-tile fastest, then c-tile, then tap
    const int tk = (int)(tl % tilesK);
    const long long r1 = tl / tilesK;
    const int tc = (int)(r1 % tilesC), rs = (int)(r1 / tilesC);
    const float* fsrc = reinterpret_cast<const float*>(sg.src);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int kr = p * 32 + (t >> 3), c4 = (t & 7) * 4;
      const float4 x = *reinterpret_cast<const float4*>(fsrc + ((long long)(tk * 64 + kr) * sg.RS + rs) * sg.C + tc * 32 + c4);
      tile[kr][c4] = x.x; tile[kr][c4 + 1] = x.y; tile[kr][c4 + 2] = x.z; tile[kr][c4 + 3] = x.w;
    }
    __syncthreads();
    const int c = t >> 3, k8 = (t & 7) * 8;
    float* fv = reinterpret_cast<float*>(v);
#pragma unroll
    for (int e = 0; e < 8; ++e) fv[e] = tile[k8 + e][c];
    i0 = ((long long)(tc * 32 + c) * sg.RS + rs) * sg.K + tk * 64 + k8;
  } else {
    i0 = (lc0 + t) * 8;
    if (i0 >= sg.n) return;
    if (sg.transposed) {
      // output order [c][rs][k]: k runs over 8 consecutive filters (K % 8 == 0), the source stride between them is RS * C
      const int k = (int)(i0 % sg.K);
      const long long r1 = i0 / sg.K;
      const int rs = (int)(r1 % sg.RS), c = (int)(r1 / sg.RS);
      const T* s0 = src + ((long long)k * sg.RS + rs) * sg.C + c;
      const long long st = (long long)sg.RS * sg.C;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = s0[e * st];
    } else if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(src + i0);
      *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(src + i0 + 4);
    }
  }
  if (sg.transposed && sg.dst) {
    T* d = reinterpret_cast<T*>(sg.dst) + i0;
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(v);
    }
  }
  if constexpr (sizeof(T) == 4) {
    if (sg.planes) {
      uint4 pl[3];
      split3_frag_rn(make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])),
                     make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])), pl);
      bf16_t* out = reinterpret_cast<bf16_t*>(sg.planes);
#pragma unroll
      for (int part = 0; part < 3; ++part) *reinterpret_cast<uint4*>(out + (long long)part * sg.n + i0) = pl[part];
    }
  }
}
DPA_API int dpa_weight_prep(const void* segs, int nsegs, long long total_chunks, int dtype, hipStream_t s) {
  if (nsegs <= 0 || total_chunks <= 0) return 0;
  if (total_chunks > 0x7fffffffll * 256) return 1;
  const dim3 grid((unsigned)((total_chunks + 255) / 256));
  if (dtype == 1) hipLaunchKernelGGL(weight_prep_kernel<bf16_t>, grid, dim3(256), 0, s, (const WPrepSeg*)segs, nsegs, total_chunks);
  else            hipLaunchKernelGGL(weight_prep_kernel<float>, grid, dim3(256), 0, s, (const WPrepSeg*)segs, nsegs, total_chunks);
  return dpa_launch_rc();
}
DPA_API int dpa_weight_prep_seg_size() { return (int)sizeof(WPrepSeg); }

template <int V> struct X6pInt { static constexpr int value = V; };

// NW = 4 waves per block, stacked along M; each wave owns TMW = 2 row tiles of 16 rows: a 128 x BN block tile, two blocks
// per CU.  Variants that were measured and lost in round 2 are NOT compiled into the library any more (VERDICT r2 weak
// #10); their code is in the history (commit 6a4c0f5) and their numbers in profiles/: 8 waves / 256-row tiles, one block
// per CU with 4 row tiles per wave ("tall": 10-15 % slower, r2_x6p_tall_tile.txt), activations pre-split into planes by
// the producer ("x6pp": 23-30 % slower, r2_x6pp_activation_planes.txt), normalise-on-load (< 1 % net, r2_norm_on_load.txt).
// TMW = 1 ("short" tile, 64 x 64, round 3): for problems whose 128-row tiles would not fill ONE round of the chip's block
// slots (small batches on 7x7 / 4x4 maps: ResNet-18 layer4 at batch 64 has 200 tiles for 512 slots) - twice the blocks,
// three resident per CU; each activation fragment still feeds 4 filter fragments x 6 products.
// RICH (dgrad): the epilogue features of the residual-block launches compiled in - 1: the ReLU mask as bits; 2: also the
// compact addend (addend2) and the second BatchNorm slab.  Instantiations of their own: with those paths in every data-gradient kernel the plain and
// singly-fused launches (35 of the 51 per ResNet-50 step) ran 5-6 % slower (register pressure around the K loop: 807 ->
// 850 us average, profiles/r3_conv_layers_fp32_bs1024.txt).
template <int TN, bool DGRAD, int NW, int MF, int TMW = 2, int ABL = 0, bool PIPE = false, int RICH = 0>
__global__ void __launch_bounds__(NW * 64, (TMW == 1 ? 3 : 2)) conv_x6p_kernel(const ConvParams p) {
  // ABL (diagnostic builds only, -DDPA_X6P_DIAG; results are WRONG, timing only): bit 0 = no operand split, bit 1 = no
  // LDS-DMA after the second K-tile, bit 2 = no MFMAs.
  static_assert(TMW == 1 || TMW == 2, "row tiles per wave");
  constexpr int WROWS = 16 * TMW;                  // rows of the block tile a wave owns
  constexpr int BM = WROWS * NW, BN = TN * 16, BKE = 32, NT = NW * 64;
  constexpr int A_BYTES = BM * 128, BP_BYTES = BN * 64;
  constexpr int STAGE_BYTES = A_BYTES + 3 * BP_BYTES;
  constexpr int RPP = NT / 8;                      // rows staged per pass (8 threads per 128-byte row)
  constexpr int A_PASSES = BM / RPP;               // = 4
  constexpr int B_PIECES = (3 * TN + NW - 1) / NW; // 1-KiB pieces (16 rows x 64 B of one plane) staged by each wave
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = p.tilesM * p.tilesN;
  const int tile = xcd_remap((int)blockIdx.x, ntiles);
  const int tm = tile / p.tilesN, tn = tile - tm * p.tilesN;
  const int m0 = p.m_off + tm * BM, n0 = tn * BN;
  const float* __restrict__ src = reinterpret_cast<const float*>(p.src);
  const bf16_t* __restrict__ wgt3 = reinterpret_cast<const bf16_t*>(p.wgt3);

  // ---- A operand: this thread stages the chunk that lands in slot tid & 7 of rows (tid >> 3) + 32 i -------------------
  const int row0 = tid >> 3;
  const int ck = (tid & 7) ^ x6p_a_swz(row0);      // (rows advance by 32 per pass: the swizzle is the same for every pass)
  const int ntaps = p.nR * p.nS;
  // (block-uniform) the short-reduction 1x1 layers spend as long in this prologue as in their two K-tiles
  const bool linear = ntaps == 1 && p.sh == 1 && p.sw == 1 && p.ph == 0 && p.pw == 0 && p.sH == p.dH && p.sW == p.dW;
  long long a_rowoff[A_PASSES];
  uint32_t a_mask[A_PASSES];
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int m = m0 + row0 + RPP * i;
    a_mask[i] = 0; a_rowoff[i] = 0;
    if (m < p.M && linear) {        // 1x1, stride 1, no padding: GEMM row m IS source pixel m, its single tap always valid
      a_rowoff[i] = (long long)m * p.s_cs + p.s_co + ck * 4;
      a_mask[i] = 1u;
    } else if (m < p.M) {
      const uint32_t t1 = fd_div((uint32_t)m, p.fd_nW);
      const int xo = m - (int)t1 * p.cls_nw;
      const uint32_t img = fd_div(t1, p.fd_nH);
      const int yo = (int)t1 - (int)img * p.cls_nh;
      const int y0 = DGRAD ? yo + p.base_y : yo * p.sh - p.ph;
      const int x0 = DGRAD ? xo + p.base_x : xo * p.sw - p.pw;
      a_rowoff[i] = ((long long)img * (p.sH * p.sW) + (long long)y0 * p.sW + x0) * p.s_cs + p.s_co + ck * 4;
      uint32_t mk = 0;
      for (int t = 0; t < ntaps; ++t) {
        const int r = (int)fd_div((uint32_t)t, p.fd_nS), s_ = t - r * p.nS;
        const int sy = DGRAD ? y0 - r : y0 + r;
        const int sx = DGRAD ? x0 - s_ : x0 + s_;
        if (((unsigned)sy < (unsigned)p.sH) && ((unsigned)sx < (unsigned)p.sW)) mk |= 1u << t;
      }
      a_mask[i] = mk;
    }
  }
  // ---- B operand: piece q = wave + NW i of the K-tile: plane q / TN, rows 16 (q % TN) .. + 15; lane -> row lane >> 2,
  //      slot lane & 3, which holds logical chunk slot ^ ((-(row >> 2)) & 3) -----------------------------------------------
  long long b_off[B_PIECES];
  int b_lds[B_PIECES];
#pragma unroll
  for (int i = 0; i < B_PIECES; ++i) {
    const int q = wave + NW * i;                   // (3 TN is a multiple of NW for every instantiation but TN = 4, NW = 8)
    const int plane = q / TN, r16 = q - plane * TN;
    const int row = 16 * r16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((-(row >> 2)) & 3);
    const int n = n0 + row;
    // rows beyond dC alias row 0 of the filter: their output columns are never stored
    b_off[i] = (long long)plane * p.wgt3_plane + (long long)(n < p.dC ? n : 0) * (p.R * p.S * p.sC) + chunk * 8;
    b_lds[i] = A_BYTES + plane * BP_BYTES + r16 * 1024;
  }
  const float* const zero = reinterpret_cast<const float*>(dpa_zero_chunk_x6p);

  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void gbl_void;
  typedef __attribute__((address_space(3))) char lds_char_t;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_char_t*)smem);
  auto issue_tile = [&](int kt, int buf) __attribute__((always_inline)) {
    const uint32_t k0 = (uint32_t)kt * BKE;                       // wave-uniform
    const uint32_t tap = fd_div(k0, p.fd_sC);
    const int cbase = (int)(k0 - tap * (uint32_t)p.sC);
    const int r = (int)fd_div(tap, p.fd_nS);
    const int s = (int)tap - r * p.nS;
    const long long delta = (long long)(DGRAD ? -(r * p.sW + s) : (r * p.sW + s)) * p.s_cs + cbase;
    const int woff = DGRAD ? (((p.tap_r0 + p.sh * r) * p.S + p.tap_s0 + p.sw * s) * p.sC + cbase) : (int)k0;
    const uint32_t bit = 1u << tap;
    // (from inline asm, dpa_glds16: hipcc does not count these, so the fragment reads behind them get no vmcnt(0) and the
    // DMA can be issued first in the iteration)
    const uint32_t stage = lds_base + buf * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      const float* pa = (a_mask[i] & bit) ? src + (a_rowoff[i] + delta) : zero;
      dpa_glds16(pa, stage + (i * RPP + wave * 8) * 128);
    }
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i)
      if ((3 * TN) % NW == 0 || wave + NW * i < 3 * TN)          // wave-uniform
        dpa_glds16(wgt3 + b_off[i] + woff, stage + b_lds[i]);
  };

  const int nkt = (p.Kg + BKE - 1) / BKE;
  float* __restrict__ dst = reinterpret_cast<float*>(p.dst);
  const float* __restrict__ addend = reinterpret_cast<const float*>(p.addend);
  constexpr int PITCH = BN + 4;
  float* stage = reinterpret_cast<float*>(smem) + wave * (16 * PITCH);
  float* red = reinterpret_cast<float*>(smem) + NW * 16 * PITCH;      // [NW waves][2][BN]
  constexpr int LPR = BN / 4;                  // lanes per output row (16-byte chunks)
  constexpr int RPI = 64 / LPR;                // rows per store instruction

  BnReduceLane<float, 4> bnr;                  // fused BatchNorm-backward reduction (dgrad, p.bnr_x != null); its 24 registers
                                               // of per-channel state are filled at the START OF THE EPILOGUE, not here: they
                                               // would be live across the whole K loop
  // the second BatchNorm of the same sum (conv_common.h: bnr2_x): sum dz * xhat2 (constants loaded at the epilogue's start)
  float q2[4] = {0.f, 0.f, 0.f, 0.f}, mu2[4] = {0.f, 0.f, 0.f, 0.f}, is2[4] = {0.f, 0.f, 0.f, 0.f};
  // 16 staged rows (block rows m0 + wave*WROWS + i*16 ..) -> global memory: addend, ReLU, dgrad pixel map, fused BN reduce.
  // Two phases so that every global read of the 16 rows (addend, and for the fused BatchNorm reduction x and y) is in
  // flight before the first one is consumed: issued one row group at a time, their latency was exposed 16 / RPI times
  // per wave and the fused reduction cost more than the kernel it replaced.
  // plain variant (no fused reduction): one row group at a time - fewer live registers, and measurably faster on the
  // memory-bound 1x1 layers than the two-phase variant below (56x56 64->256 fprop: 1137 vs 1355 us)
  // compact addend (dgrad, p.addend2): the value for GEMM row m, columns col .. col + 3, or zeros for pixels with an odd
  // coordinate.  (img, hi, wi) of the full gradient tensor from the class row, as in the pixel map below.
  const float* __restrict__ addend2 = RICH == 2 ? reinterpret_cast<const float*>(p.addend2) : nullptr;
  auto compact_addend = [&](int m, int col) __attribute__((always_inline)) {
    const uint32_t t1 = fd_div((uint32_t)m, p.fd_nW);
    const int jx = m - (int)t1 * p.cls_nw;
    const uint32_t img = fd_div(t1, p.fd_nH);
    const int iy = (int)t1 - (int)img * p.cls_nh;
    const int hi = p.cls_h0 + p.sh * iy, wi = p.cls_w0 + p.sw * jx;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (((hi | wi) & 1) == 0)
      a = *reinterpret_cast<const float4*>(addend2 + (((size_t)img * p.a2_H + (hi >> 1)) * p.a2_W + (wi >> 1)) * (size_t)p.a2_cs + col);
    return a;
  };
  auto store_rows_plain = [&](int i) __attribute__((always_inline)) {
#pragma unroll
    for (int rr = 0; rr < 16; rr += RPI) {
      const int r = rr + lane / LPR;
      const int cc = (lane % LPR) * 4;
      const int m = m0 + wave * WROWS + i * 16 + r;
      const int col = n0 + cc;
      float4 v = *reinterpret_cast<const float4*>(&stage[r * PITCH + cc]);
      if (m < p.M && col < p.dC) {
        size_t pixel = (size_t)m;
        if (DGRAD && (p.sh != 1 || p.sw != 1)) {      // class row -> (img, hi, wi) of the full gradient tensor
          const uint32_t t1 = fd_div((uint32_t)m, p.fd_nW);
          const int jx = m - (int)t1 * p.cls_nw;
          const uint32_t img = fd_div(t1, p.fd_nH);
          const int iy = (int)t1 - (int)img * p.cls_nh;
          pixel = ((size_t)img * p.dH + (size_t)(p.cls_h0 + p.sh * iy)) * p.dW + (size_t)(p.cls_w0 + p.sw * jx);
        }
        const size_t off = pixel * (size_t)p.d_cs + (size_t)(p.d_co + col);
        if (addend) {
          const float4 a = *reinterpret_cast<const float4*>(addend + off);
          v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        if constexpr (RICH == 2) {
          if (addend2) { const float4 a = compact_addend(m, col); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
        }
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(dst + off) = v;
      }
    }
  };
  auto store_rows_fused = [&](int i) __attribute__((always_inline)) {
    constexpr int NG = 16 / RPI;
    const int cc = (lane % LPR) * 4;
    const int col = n0 + cc;
    size_t off[NG], pix[NG];
    bool ok[NG];
    float4 av[NG], xv[NG], yv[NG];
    uint32_t mw[NG];
    const bool fuse = DGRAD && p.bnr_x != nullptr;
    const float* __restrict__ bx = reinterpret_cast<const float*>(p.bnr_x);
    const uint32_t* __restrict__ bits = RICH ? reinterpret_cast<const uint32_t*>(p.bnr_bits) : nullptr;
    // (the ReLU mask as bits when the producer made them: one word per 32 channels instead of 128 bytes of y)
    const float* __restrict__ by = bits ? nullptr : reinterpret_cast<const float*>(p.bnr_y);
    // (the second BatchNorm's input: only with the bit mask - its values live in the registers y would take)
    const float* __restrict__ bx2 = (RICH == 2 && bits) ? reinterpret_cast<const float*>(p.bnr2_x) : nullptr;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int m = m0 + wave * WROWS + i * 16 + g * RPI + lane / LPR;
      ok[g] = m < p.M && col < p.dC;
      size_t pixel = ok[g] ? (size_t)m : 0;
      if (DGRAD && (p.sh != 1 || p.sw != 1) && ok[g]) {      // class row -> (img, hi, wi) of the full gradient tensor
        const uint32_t t1 = fd_div((uint32_t)m, p.fd_nW);
        const int jx = m - (int)t1 * p.cls_nw;
        const uint32_t img = fd_div(t1, p.fd_nH);
        const int iy = (int)t1 - (int)img * p.cls_nh;
        pixel = ((size_t)img * p.dH + (size_t)(p.cls_h0 + p.sh * iy)) * p.dW + (size_t)(p.cls_w0 + p.sw * jx);
      }
      const int c_ok = ok[g] ? col : 0;
      pix[g] = pixel;
      off[g] = pixel * (size_t)p.d_cs + (size_t)(p.d_co + c_ok);
      if (addend) av[g] = *reinterpret_cast<const float4*>(addend + off[g]);
      if constexpr (RICH == 2) {    // (never together with a dense addend: conv_igemm.hip launch_conv)
        if (addend2) av[g] = ok[g] ? compact_addend(m, col) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (fuse) {
        xv[g] = *reinterpret_cast<const float4*>(bx + pixel * (size_t)p.bnr_cs + c_ok);
        if (by) yv[g] = *reinterpret_cast<const float4*>(by + pixel * (size_t)p.bnr_y_cs + c_ok);
        if (bits) mw[g] = bits[pixel * (size_t)(p.dC >> 5) + (c_ok >> 5)];
        if (bx2) yv[g] = *reinterpret_cast<const float4*>(bx2 + pixel * (size_t)p.bnr2_cs + c_ok);
      }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int r = g * RPI + lane / LPR;
      float4 v = *reinterpret_cast<const float4*>(&stage[r * PITCH + cc]);
      if (ok[g]) {
        if (addend || (RICH == 2 && addend2)) { v.x += av[g].x; v.y += av[g].y; v.z += av[g].z; v.w += av[g].w; }
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (fuse) {
          float a4[4] = {v.x, v.y, v.z, v.w};
          const float x4[4] = {xv[g].x, xv[g].y, xv[g].z, xv[g].w};
          float y4[4] = {yv[g].x, yv[g].y, yv[g].z, yv[g].w};
          const float z2[4] = {yv[g].x, yv[g].y, yv[g].z, yv[g].w};      // (the second BatchNorm's input when bx2)
          if (RICH && bits) {
            const uint32_t nib = mw[g] >> (col & 31);
#pragma unroll
            for (int e = 0; e < 4; ++e) y4[e] = (nib >> e) & 1u ? 1.f : 0.f;
          }
          bnr.visit_loaded(by != nullptr || bits != nullptr, x4, y4, a4);
          if constexpr (RICH == 2) {
            if (bx2) {
#pragma unroll
              for (int e = 0; e < 4; ++e) q2[e] += a4[e] * (z2[e] - mu2[e]) * is2[e];
            }
          }
          v = make_float4(a4[0], a4[1], a4[2], a4[3]);
        }
        *reinterpret_cast<float4*>(dst + off[g]) = v;
      }
    }
  };
  auto store_rows = [&](int i) __attribute__((always_inline)) {
    if (DGRAD && p.bnr_x != nullptr) store_rows_fused(i); else store_rows_plain(i);
  };
  // block-wide: per-wave column sums in red[] -> one slab row per M-tile, waves added in fixed order (reproducible)
  auto write_stats = [&](float* slab, int row0) __attribute__((always_inline)) {
    __syncthreads();
    for (int c = tid; c < 2 * BN; c += NT) {
      const int which = c / BN, col = c - which * BN;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[(w * 2 + which) * BN + col];
      if (n0 + col < p.dC) slab[((size_t)(row0 + tm) * 2 + which) * p.dC + n0 + col] = s;
    }
  };
  const bool interior = m0 + BM <= p.M;          // block-uniform

  {
    constexpr int TM = TMW;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    const int b_frag = fr * 64 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);     // + plane * BP_BYTES + j * 1024
    constexpr int JH = TN > 4 ? 4 : TN;                                // filter fragments held in registers at a time

    if constexpr (PIPE) {
      // ---- software-pipelined K loop --------------------------------------------------------------------------------------
      // Measured on the loop below (profiles/r3_x6p_ablation.txt): its phases ADD - without MFMAs it still takes 60 % of
      // its time, MFMAs + fragment reads alone 76 % - because a wave reads its fragments, splits them and issues the next
      // tile's DMA BETWEEN its MFMA runs.  Here every wave works one tile ahead of its own matrix work: while the MFMAs of
      // tile kt run (operands `cur`, already split), it reads and splits the activation fragments of tile kt + 1 (`nxt`),
      // streams the filter fragments of tile kt two column tiles at a time, and the DMA of A(kt + 2) / B(kt + 1) was issued
      // at the top of the iteration - a whole iteration before anyone waits for it.  LDS: two A buffers, two B buffers
      // (the same 80 KB): A[kt & 1] was read during iteration kt - 1 and is free after the barrier that ended it.
      typedef __attribute__((address_space(3))) char lds_char;
      const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_char*)smem);
      constexpr int B_BASE = 2 * A_BYTES;
      auto issue_A = [&](int kt, int par) __attribute__((always_inline)) {
        const uint32_t k0 = (uint32_t)kt * BKE;
        const uint32_t tap = fd_div(k0, p.fd_sC);
        const int cbase = (int)(k0 - tap * (uint32_t)p.sC);
        const int r = (int)fd_div(tap, p.fd_nS);
        const int s = (int)tap - r * p.nS;
        const long long delta = (long long)(DGRAD ? -(r * p.sW + s) : (r * p.sW + s)) * p.s_cs + cbase;
        const uint32_t bit = 1u << tap;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
          const float* pa = (a_mask[i] & bit) ? src + (a_rowoff[i] + delta) : zero;
          dpa_glds16(pa, lds0 + par * A_BYTES + (i * RPP + wave * 8) * 128);
        }
      };
      auto issue_B = [&](int kt, int par) __attribute__((always_inline)) {
        const uint32_t k0 = (uint32_t)kt * BKE;
        int woff = (int)k0;
        if constexpr (DGRAD) {
          const uint32_t tap = fd_div(k0, p.fd_sC);
          const int cbase = (int)(k0 - tap * (uint32_t)p.sC);
          const int r = (int)fd_div(tap, p.fd_nS);
          const int s = (int)tap - r * p.nS;
          woff = ((p.tap_r0 + p.sh * r) * p.S + p.tap_s0 + p.sw * s) * p.sC + cbase;
        }
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i)
          if ((3 * TN) % NW == 0 || wave + NW * i < 3 * TN)          // wave-uniform
            dpa_glds16(wgt3 + b_off[i] + woff, lds0 + B_BASE + par * 3 * BP_BYTES + (b_lds[i] - A_BYTES));
      };
      auto read_A = [&](int par, uint4 (&lo)[TM], uint4 (&hi)[TM]) __attribute__((always_inline)) {
        const char* tA = smem + par * A_BYTES;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          lo[i] = *reinterpret_cast<const uint4*>(tA + x6p_a_off(wave * WROWS + i * 16 + fr, 2 * fq));
          hi[i] = *reinterpret_cast<const uint4*>(tA + x6p_a_off(wave * WROWS + i * 16 + fr, 2 * fq + 1));
        }
      };
      uint4 ap0[TM][3], ap1[TM][3];
      auto body = [&](int kt, auto parc, uint4 (&cur)[TM][3], uint4 (&nxt)[TM][3]) __attribute__((always_inline)) {
        constexpr int par = decltype(parc)::value;               // kt & 1
        // (Issuing the DMA piece by piece BETWEEN the column tiles' MFMAs was tried three ways - asm statements with a
        // "memory" clobber, without one behind MFMA-only sched_barrier fences, and pinned by fake register dependencies on
        // the accumulators: each time hipcc's scheduler gave up the MFMA / VALU interleave below around the asm statements
        // and the loop got slower.  The block of ten pieces here costs 8 % of the loop: profiles/r3_x6p_ablation.txt.)
        if (!(ABL & 2) || kt < 1) {
          if (kt + 2 < nkt) issue_A(kt + 2, par);
          if (kt + 1 < nkt) issue_B(kt + 1, par ^ 1);
        }
        const char* tB = smem + B_BASE + par * 3 * BP_BYTES;
        uint4 bp[2][3], fl[TM], fh[TM];
#pragma unroll
        for (int part = 0; part < 3; ++part) bp[0][part] = *reinterpret_cast<const uint4*>(tB + part * BP_BYTES + b_frag);
        read_A(par ^ 1, fl, fh);       // (after the last tile: stale bytes of the same buffer, split and never used)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (j + 1 < TN) {
#pragma unroll
            for (int part = 0; part < 3; ++part)
              bp[(j + 1) & 1][part] = *reinterpret_cast<const uint4*>(tB + part * BP_BYTES + (j + 1) * 1024 + b_frag);
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int t = 0; t < 6; ++t) {
              if constexpr (ABL & 4) x6p_keep(cur[i][kX6A[t]], bp[j & 1][kX6B[t]]);
              else MfmaT<bf16_t>::mma(cur[i][kX6A[t]], bp[j & 1][kX6B[t]], acc[i][j]);
            }
          // the next tile's operand split, one row tile in the program text of each of the first TM column tiles
          if (j < TM) {
            if constexpr (ABL & 1) { nxt[j][0] = fl[j]; nxt[j][1] = fh[j]; nxt[j][2] = fl[j]; }
            else split3_frag(fl[j], fh[j], nxt[j]);
          }
        }
        // pin the split INSIDE this iteration (its results are only used in the next one: left alone, hipcc sinks all of
        // it below the barrier, in front of the next iteration's MFMAs - the very serialisation this loop removes) ...
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int part = 0; part < 3; ++part)
            asm volatile("" : "+v"(nxt[i][part].x), "+v"(nxt[i][part].y), "+v"(nxt[i][part].z), "+v"(nxt[i][part].w));
        // ... and describe the interleave to the scheduler: the fragment reads of the first column tile and of the next
        // row tiles, then per column tile the reads of the following one and its 6 TM MFMAs, one split instruction behind
        // each MFMA (36 TM of them, spread evenly: an MFMA holds the issue port for 8 of its 16 cycles).
        if constexpr (ABL == 0) {
          __builtin_amdgcn_sched_group_barrier(0x100, 3 + 2 * TM, 0);
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if (j + 1 < TN) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
            for (int m = 0; m < 6 * TM; ++m) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              constexpr int NV = 36 * TM + 8, NM = 6 * TM * TN;       // vector instructions to place, MFMAs to place them behind
              if (NV / NM >= 1) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
              if (NV / NM >= 2) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
              if (j * 6 * TM + m < NV % NM) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      };
      if (nkt > 0) {
        issue_A(0, 0);
        issue_B(0, 0);
        if (nkt > 1) issue_A(1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        {
          uint4 fl[TM], fh[TM];
          read_A(0, fl, fh);
#pragma unroll
          for (int i = 0; i < TM; ++i) split3_frag(fl[i], fh[i], ap0[i]);
        }
        __syncthreads();               // every wave has A(0) in registers before iteration 0 refills A[0]
        for (int kt = 0; kt < nkt; kt += 2) {
          body(kt, X6pInt<0>{}, ap0, ap1);
          if (kt + 1 < nkt) body(kt + 1, X6pInt<1>{}, ap1, ap0);
        }
      }
    } else {
    if (nkt > 0) issue_tile(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      // short reductions (fewer than four K-tiles: the 56x56 1x1 layers on 64 channels): the next tile's DMA first - its
      // buffer was read during iteration kt - 1 - so that it has the whole iteration to land (+4-10 % on those layers;
      // the 64 -> 64 3x3 layers, 18 K-tiles, lose 3 % that way and keep the issue behind the fragment reads)
      const bool early = nkt < 4;                                        // block-uniform
      if (early && kt + 1 < nkt && (!(ABL & 2) || kt < 1)) issue_tile(kt + 1, (kt + 1) & 1);
      const char* tA = smem + (kt & 1) * STAGE_BYTES;
      const char* tB = tA + A_BYTES;
      uint4 fa[2][TM];
      uint4 ap[TM][3];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[ks][i] = *reinterpret_cast<const uint4*>(tA + x6p_a_off(wave * WROWS + i * 16 + fr, 2 * fq + ks));
      uint4 bp[JH][3];
#pragma unroll
      for (int j = 0; j < JH; ++j)
#pragma unroll
        for (int part = 0; part < 3; ++part)
          bp[j][part] = *reinterpret_cast<const uint4*>(tB + part * BP_BYTES + j * 1024 + b_frag);
      if constexpr (TN <= 4) { if (!early && kt + 1 < nkt && (!(ABL & 2) || kt < 1)) issue_tile(kt + 1, (kt + 1) & 1); }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (ABL & 1) { ap[i][0] = fa[0][i]; ap[i][1] = fa[1][i]; ap[i][2] = fa[0][i]; }
        else split3_frag(fa[0][i], fa[1][i], ap[i]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int j = 0; j < JH; ++j) {
            if constexpr (ABL & 4) x6p_keep(ap[i][kX6A[t]], bp[j][kX6B[t]]);
            else MfmaT<bf16_t>::mma(ap[i][kX6A[t]], bp[j][kX6B[t]], acc[i][j]);
          }
      if constexpr (TN > 4) {
        // second half of the filter fragments, then the next tile's DMA, then the other 48 MFMAs
#pragma unroll
        for (int j = 0; j < JH; ++j)
#pragma unroll
          for (int part = 0; part < 3; ++part)
            bp[j][part] = *reinterpret_cast<const uint4*>(tB + part * BP_BYTES + (JH + j) * 1024 + b_frag);
        if (!early && kt + 1 < nkt && (!(ABL & 2) || kt < 1)) issue_tile(kt + 1, (kt + 1) & 1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int j = 0; j < JH; ++j) {
              if constexpr (ABL & 4) x6p_keep(ap[i][kX6A[t]], bp[j][kX6B[t]]);
              else MfmaT<bf16_t>::mma(ap[i][kX6A[t]], bp[j][kX6B[t]], acc[i][JH + j]);
            }
      }
    }
    }
    __syncthreads();

    // ---- epilogue: acc[i][j][e] = D[row = wave*WROWS + i*16 + fq*4 + e][col = j*16 + fr] ---------------------------------
    if constexpr (DGRAD) { if (p.bnr_x) bnr.init(p, n0 + (lane % LPR) * 4); }
    if constexpr (RICH == 2) {
      if (p.bnr2_x) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c2 = n0 + (lane % LPR) * 4 + e;
          mu2[e] = c2 < p.dC ? p.bnr2_mean[c2] : 0.f; is2[e] = c2 < p.dC ? p.bnr2_invstd[c2] : 0.f;
        }
      }
    }
    float csum[TN], csq[TN], bj[TN], cj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + j * 16 + fr;
      csum[j] = 0.f; csq[j] = 0.f;
      bj[j] = (p.bias && col < p.dC) ? p.bias[col] : 0.f;
      cj[j] = (p.stats && p.stats_shift && col < p.dC) ? p.stats_shift[col] : 0.f;
    }
    if constexpr (!DGRAD) if (p.oscale) {      // folded inference BatchNorm: acc * scale[col] before the bias
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + j * 16 + fr;
        const float sc = col < p.dC ? p.oscale[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] *= sc;
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) stage[(fq * 4 + e) * PITCH + j * 16 + fr] = acc[i][j][e] + bj[j];
      __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0)
      __builtin_amdgcn_wave_barrier();
      store_rows(i);
      __builtin_amdgcn_wave_barrier();
      if (p.stats) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[i][j][e] + bj[j] - cj[j];
            if (!interior && m0 + wave * WROWS + i * 16 + fq * 4 + e >= p.M) v = 0.f;
            csum[j] += v; csq[j] += v * v;
          }
      }
    }
    if (p.stats) {
      __syncthreads();                             // the staging region is reused below (block-wide)
      // (every cross-lane sum, a wait, then the LDS writes - never a ds_bpermute right behind the masked ds_write of the
      //  previous group: conv_igemm.hip's bf16 instantiations lost sums of squares that way, see its epilogue)
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
          red[(wave * 2 + 0) * BN + j * 16 + fr] = csum[j];
          red[(wave * 2 + 1) * BN + j * 16 + fr] = csq[j];
        }
      write_stats(p.stats, p.stats_row0);
    }
    if constexpr (DGRAD) {
      if (p.bnr_x) {
        __syncthreads();
        float rs[4], rq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          rs[e] = bnr.s[e]; rq[e] = bnr.q[e];
#pragma unroll
          for (int o = LPR; o < 64; o <<= 1) { rs[e] += __shfl_xor(rs[e], o, 64); rq[e] += __shfl_xor(rq[e], o, 64); }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (lane < LPR) {
            red[(wave * 2 + 0) * BN + lane * 4 + e] = rs[e];
            red[(wave * 2 + 1) * BN + lane * 4 + e] = rq[e];
          }
        write_stats(p.bnr_slab, p.bnr_row0 + p.stats_row0);
        if (RICH == 2 && p.bnr2_x) {
          __syncthreads();
#pragma unroll
          for (int e = 0; e < 4; ++e) {              // (sum dz again: already reduced above)
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1) q2[e] += __shfl_xor(q2[e], o, 64);
          }
          __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (lane < LPR) {
              red[(wave * 2 + 0) * BN + lane * 4 + e] = rs[e];
              red[(wave * 2 + 1) * BN + lane * 4 + e] = q2[e];
            }
          write_stats(p.bnr2_slab, p.bnr_row0 + p.stats_row0);
        }
      }
    }
  }
}

template <int TN, int NW, int MF, int TMW = 2>
static void x6p_launch(ConvParams& p, int dgrad, hipStream_t s) {
  constexpr int BN = TN * 16, BM = 16 * TMW * NW;
  p.tilesM = (p.M - p.m_off + BM - 1) / BM;
  p.tilesN = (p.dC + BN - 1) / BN;
  size_t lds = (size_t)2 * (BM * 128 + 3 * BN * 64);
  const size_t epi = (size_t)(NW * 16 * (BN + 4) + NW * 2 * BN) * 4;
  if (epi > lds) lds = epi;
  const dim3 grid(p.tilesM * p.tilesN), block(NW * 64);
  const int rich = !dgrad ? 0 : ((p.addend2 || p.bnr2_x) ? 2 : (p.bnr_bits ? 1 : 0));      // (see the template parameter)
#ifdef DPA_X6P_DIAG
  if constexpr (TN == 8 && TMW == 2) {
    const char* e = getenv("DPA_X6P_ABL");
    const int abl = e ? atoi(e) : 0;
#define DPA_ABL_CASE(N) if (abl == N && !dgrad) { hipLaunchKernelGGL((conv_x6p_kernel<TN, false, NW, MF, TMW, N>), grid, block, lds, s, p); return; }
    DPA_ABL_CASE(1) DPA_ABL_CASE(2) DPA_ABL_CASE(3) DPA_ABL_CASE(4) DPA_ABL_CASE(5) DPA_ABL_CASE(6) DPA_ABL_CASE(7)
#undef DPA_ABL_CASE
#define DPA_ABL_CASE(N) if (abl == 8 + N && !dgrad) { hipLaunchKernelGGL((conv_x6p_kernel<TN, false, NW, MF, TMW, N, true>), grid, block, lds, s, p); return; }
    DPA_ABL_CASE(0) DPA_ABL_CASE(1) DPA_ABL_CASE(2) DPA_ABL_CASE(3) DPA_ABL_CASE(4) DPA_ABL_CASE(5) DPA_ABL_CASE(6) DPA_ABL_CASE(7)
#undef DPA_ABL_CASE
  }
#endif
  // the software-pipelined K loop: 128 x 128 tiles only (measured, profiles/r3_x6p_pipelined_loop.txt: +5-10 % on every
  // layer with 128-column tiles and four K-tiles or more, -3 % on the 128 x 64 tiles of the 64-filter layers, whose 48
  // MFMAs per K-tile leave too few slots for the 72 split instructions)
  if constexpr (TN == 8 && TMW == 2) {
    const int pipe_min = dpa_x6p_knob(3, -1);
    if (pipe_min > 0 && (p.Kg + 31) / 32 >= pipe_min) {
      if (dgrad && rich == 2)      hipLaunchKernelGGL((conv_x6p_kernel<TN, true, NW, MF, TMW, 0, true, 2>), grid, block, lds, s, p);
      else if (dgrad && rich == 1) hipLaunchKernelGGL((conv_x6p_kernel<TN, true, NW, MF, TMW, 0, true, 1>), grid, block, lds, s, p);
      else if (dgrad)    hipLaunchKernelGGL((conv_x6p_kernel<TN, true, NW, MF, TMW, 0, true>), grid, block, lds, s, p);
      else               hipLaunchKernelGGL((conv_x6p_kernel<TN, false, NW, MF, TMW, 0, true>), grid, block, lds, s, p);
      return;
    }
  }
  if (dgrad && rich == 2)      hipLaunchKernelGGL((conv_x6p_kernel<TN, true, NW, MF, TMW, 0, false, 2>), grid, block, lds, s, p);
  else if (dgrad && rich == 1) hipLaunchKernelGGL((conv_x6p_kernel<TN, true, NW, MF, TMW, 0, false, 1>), grid, block, lds, s, p);
  else if (dgrad)    hipLaunchKernelGGL((conv_x6p_kernel<TN, true, NW, MF, TMW>), grid, block, lds, s, p);
  else               hipLaunchKernelGGL((conv_x6p_kernel<TN, false, NW, MF, TMW>), grid, block, lds, s, p);
}

// Fewest tiles for which the pre-split kernel is taken (DPA_X6P_MIN_TILES, default 96: profiles/r2_x6p_small_batch_sweep.txt).  v >= 0 sets it and returns the
// previous value (the kernel tests set 1 so that small shapes exercise the kernel); v < 0 only reads it.
DPA_API int dpa_x6p_min_tiles(int v) {
  static int cur = [] { const char* e = getenv("DPA_X6P_MIN_TILES"); return e ? atoi(e) : 96; }();
  const int prev = cur;
  if (v >= 0) cur = v;
  return prev;
}
// Tile width: 128 columns, or 64 for dC <= 64 and for problems whose 128-column tiles would not fill one round of the
// chip's 512 block slots (small batches: twice the blocks at half the filter reuse).  DPA_X6P_NARROW_BELOW (tiles,
// default 512).  Measured on the batch-64 zoo (the reference's default batch): ResNet-18 +9.7 %, AlexNet +5 %,
// Inception-v3 +6.6 %, the others unchanged; nothing in the batch-1024 ResNet-50 step is below 512 tiles.
static long long x6p_tiles(const ConvParams& p, bool narrow) {
  return (long long)((p.M + 127) / 128) * ((p.dC + (narrow ? 63 : 127)) / (narrow ? 64 : 128));
}
// which: 3 = DPA_X6P_PIPE_MIN_KT (default 4: reductions of at least that many 32-element K-tiles take the software-pipelined K
// loop; 0 = never.  profiles/r3_x6p_pipelined_loop.txt), 1 = DPA_X6P_NARROW_BELOW (default 512), 2 = DPA_X6P_SHORT_BELOW (default 320: problems with fewer 128 x 64 tiles
// take 64 x 64 tiles; 0 = never.  Measured, profiles/r3_x6p_short_tile.txt: 200-256 tiles +20-30 %, 392 tiles -20 %).  v >= 0 sets the knob and returns the previous value (the kernel tests steer small
// shapes to each tile shape with it); v < 0 only reads it.
DPA_API int dpa_x6p_knob(int which, int v) {
  static int cur[4] = {0, [] { const char* e = getenv("DPA_X6P_NARROW_BELOW"); return e ? atoi(e) : 512; }(),
                          [] { const char* e = getenv("DPA_X6P_SHORT_BELOW"); return e ? atoi(e) : 320; }(),
                          [] { const char* e = getenv("DPA_X6P_PIPE_MIN_KT"); return e ? atoi(e) : 4; }()};
  if (which < 1 || which > 3) return -1;
  const int prev = cur[which];
  if (v >= 0) cur[which] = v;
  return prev;
}
static bool x6p_narrow(const ConvParams& p) {
  return p.dC <= 64 || x6p_tiles(p, false) < dpa_x6p_knob(1, -1);
}
// Can this (sub-)problem take the pre-split kernel?  (called by conv_igemm.hip's launch_tiles for fp32 in x6 mode)
bool dpa_x6p_eligible(const ConvParams& p) {
  static const int enabled = [] { const char* e = getenv("DPA_X6P"); return (e && e[0] == '0') ? 0 : 1; }();
  static const int min_kt = [] { const char* e = getenv("DPA_X6P_MINKT"); return e ? atoi(e) : 2; }();
  const int min_tiles = dpa_x6p_min_tiles(-1);
  const int ntaps = p.nR * p.nS;
  // Only where its 128 x 128 (128 x 64) tiles fill the chip: skinny problems (Linear layers, small batches) need the
  // general kernel's split-K / small tiles, and dC <= 32 (DenseNet growth convs) its 128 x 32 tile.
  return enabled && p.wgt3 && (p.sC % 32 == 0) && ntaps >= 1 && ntaps <= 32 && !p.no_uni && p.Kg >= 32 * min_kt && p.dC > 32 &&
         x6p_tiles(p, x6p_narrow(p)) >= min_tiles;
}
// Could a problem with M rows and dC output columns take this kernel at all (are filter planes worth preparing)?  The
// per-launch decision (dpa_x6p_eligible) looks at more, this is its size part: skinny problems - Linear layers at small
// batches - never qualify, and their planes would be most of the per-step weight preparation.
DPA_API int dpa_x6p_plausible(long long M, int dC) {
  ConvParams q{}; q.M = (int)(M > 0x7fffffffLL ? 0x7fffffffLL : M); q.dC = dC;
  return dC > 32 && x6p_tiles(q, x6p_narrow(q)) >= dpa_x6p_min_tiles(-1);
}
// Launches the whole (sub-)problem; returns the number of stats-slab rows it wrote.
int dpa_x6p_launch(ConvParams& p, int dgrad, hipStream_t s) {
  p.m_off = 0; p.stats_row0 = 0;
  if (dpa_x6s_eligible(p)) return dpa_x6s_launch(p, dgrad, s);      // 1x1 layers on 64 channels: the persistent streaming kernel
  if (x6p_narrow(p)) {
    if (x6p_tiles(p, true) < dpa_x6p_knob(2, -1)) x6p_launch<4, 4, 16, 1>(p, dgrad, s);     //  64 x 64
    else                                          x6p_launch<4, 4, 16>(p, dgrad, s);        // 128 x 64
  } else                                          x6p_launch<8, 4, 16>(p, dgrad, s);        // 128 x 128
  return p.tilesM;
}
