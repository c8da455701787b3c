// fp32 (x6 arithmetic) 1x1 / stride-1 convolution with a 64-element reduction as a PERSISTENT streaming kernel ("x6s").
//
// The layers it is for (ResNet-50 at 56x56: 64 -> 256 forward, 256 -> 64 data gradient, 64 -> 64 both ways) read or write
// 3.3 GB per call at batch 1024 and do two K-tiles of matrix work per output tile.  In conv_x6p.hip every tile is a block
// of its own: address set-up, 32 KB of activations AND 48 KB of filter planes through the LDS-DMA, a wait for all of it,
// 192 MFMAs per wave, an epilogue staged through LDS - a chain of latencies that two resident blocks per CU overlap only
// by chance (profiles/r2_conv_layers_fp32_bs1024.txt: 85 TF-equivalent, 1434 us of which 1015 us without the stores and
// 545 us for the stores alone: the phases ADD).  Here:
//   * a block is resident for the whole launch and walks over M-tiles tm = slot, slot + nslots, ... of ONE column tile: the
//     filter planes of that column tile (all 64 reduction elements, 48 KB) are staged once;
//   * the activation tile of the NEXT M-tile is in flight (LDS-DMA, issued right after every wave has its fragments of the
//     current one in registers) while the current tile is multiplied and stored;
//   * the epilogue stores straight from the accumulator layout - lanes 0..15 of a quarter wave hold 16 consecutive
//     channels of one pixel: 64-byte segments, scalar base + 32-bit lane offset + immediate - no LDS staging, so the
//     activation buffer is free for the next tile's DMA during the whole epilogue;
//   * BatchNorm statistics are carried in registers across the block's tiles and written once: nslots slab rows instead
//     of one per M-tile.
// Same arithmetic as conv_x6p.hip (same fragment layouts, same order of the six partial products and of the K-tiles): the
// outputs are bit-identical to that kernel's.
#include "conv_common.h"
#include "x6_frag.h"

__device__ uint4 dpa_zero_chunk_x6s[1];

template <int V> struct X6sInt { static constexpr int value = V; };

template <int TN>
__global__ void __launch_bounds__(256, 2) conv_x6s_kernel(const ConvParams p, const int nslots) {
  constexpr int NW = 4, TM = 2, BM = 128, BN = TN * 16, NT = 256;
  constexpr int A_HALF = BM * 128;                 // one K-tile (32 fp32) of the activation tile
  constexpr int A_BYTES = 2 * A_HALF;
  constexpr int BP_BYTES = BN * 64;                // one plane of one K-tile of the filter tile
  constexpr int B_KT = 3 * BP_BYTES;
  constexpr int B_PIECES = 6 * TN / NW;            // 1-KiB pieces staged by each wave, once
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char lds_char;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  // blocks that share an XCD (blockIdx % 8) and are neighbours in it work on the same M-tiles of different column tiles
  const int bq = (int)blockIdx.x >> 3;
  const int tn = bq % p.tilesN;
  const int slot = (bq / p.tilesN) * 8 + ((int)blockIdx.x & 7);
  const int n0 = tn * BN;
  const float* __restrict__ src = reinterpret_cast<const float*>(p.src);
  const bf16_t* __restrict__ wgt3 = reinterpret_cast<const bf16_t*>(p.wgt3);
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_char*)smem);

  // ---- filter planes of this column tile, both K-tiles: once ----------------------------------------------------------
#pragma unroll
  for (int i = 0; i < B_PIECES; ++i) {
    const int q = wave + NW * i;
    const int kt = q / (3 * TN), r1 = q - kt * 3 * TN;
    const int plane = r1 / TN, r16 = r1 - plane * TN;
    const int row = 16 * r16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((-(row >> 2)) & 3);
    dpa_glds16(wgt3 + (long long)plane * p.wgt3_plane + (long long)(n0 + row) * 64 + kt * 32 + chunk * 8,
               lds0 + A_BYTES + kt * B_KT + plane * BP_BYTES + r16 * 1024);
  }
  // ---- activation tile: thread -> slot tid & 7 of rows (tid >> 3) + 32 i, both K-tiles ----------------------------------
  const int row0 = tid >> 3;
  const int ck = (tid & 7) ^ x6p_a_swz(row0);
  const float* const zero = reinterpret_cast<const float*>(dpa_zero_chunk_x6s);
  auto issue_A = [&](int tm) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long m = (long long)tm * BM + row0 + 32 * i;
      const float* pa = src + (m * p.s_cs + p.s_co + ck * 4);
      const bool ok = m < p.M;
      dpa_glds16(ok ? pa : zero, lds0 + (i * 32 + wave * 8) * 128);
      dpa_glds16(ok ? pa + 32 : zero, lds0 + A_HALF + (i * 32 + wave * 8) * 128);
    }
  };
  if (slot < p.tilesM) issue_A(slot);

  // ---- per-lane epilogue constants: lane (fr, fq) holds rows fq * 4 + e, column j * 16 + fr of every 16 x 16 tile --------
  float bj[TN], sj[TN], csum[TN], csq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + j * 16 + fr;
    bj[j] = p.bias ? p.bias[col] : 0.f;
    sj[j] = bj[j] - ((p.stats && p.stats_shift) ? p.stats_shift[col] : 0.f);      // statistics are about the shift
    csum[j] = 0.f; csq[j] = 0.f;
  }
  char* const dst = reinterpret_cast<char*>(p.dst);
  const char* const addend = reinterpret_cast<const char*>(p.addend);
  const uint32_t voff = (uint32_t)(fq * 4 * p.d_cs + fr) * 4u;                     // lane part of every store address
  const int b_frag = fr * 64 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);

  // The loop is rotated: the fragments of tile t + 1 are fetched BETWEEN the MFMAs and the stores of tile t.  The wait in
  // front of them then covers a DMA that was issued before the MFMAs of tile t and the stores of tile t - 1 - nothing
  // recent - and the stores of tile t drain while tile t + 1 is multiplied.  (Not rotated, every iteration began by
  // waiting for the stores it had just issued: 967 instead of 880 us on 56x56 64 -> 256 at batch 1024; conv_x6p.hip: 1250.)
  uint4 fa[2][TM][2];
  auto fetch_frags = [&](int tm_after) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tile's activations (and, the first time, the filter planes)
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          fa[kt][i][h] = *reinterpret_cast<const uint4*>(smem + kt * A_HALF + x6p_a_off(wave * 32 + i * 16 + fr, 2 * fq + h));
    __syncthreads();                                      // (waits for the reads) every wave has its fragments: the buffer is free
    if (tm_after < p.tilesM) issue_A(tm_after);
  };
  if (slot < p.tilesM) fetch_frags(slot + nslots);
  for (int tm = slot; tm < p.tilesM; tm += nslots) {
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      uint4 ap[TM][3];
#pragma unroll
      for (int i = 0; i < TM; ++i) split3_frag(fa[kt][i][0], fa[kt][i][1], ap[i]);
      const char* tB = smem + A_BYTES + kt * B_KT + b_frag;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        uint4 bp[3];
#pragma unroll
        for (int part = 0; part < 3; ++part) bp[part] = *reinterpret_cast<const uint4*>(tB + part * BP_BYTES + j * 1024);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int t = 0; t < 6; ++t) MfmaT<bf16_t>::mma(ap[i][kX6A[t]], bp[kX6B[t]], acc[i][j]);
      }
    }

    if (tm + nslots < p.tilesM) fetch_frags(tm + 2 * nslots);      // block-uniform

    // ---- epilogue: straight from the accumulators ---------------------------------------------------------------------
    const long long mrow = (long long)tm * BM + wave * 32;                        // wave-uniform
    const bool interior = (long long)tm * BM + BM <= p.M;                         // block-uniform
    const size_t ubase = ((size_t)mrow * p.d_cs + p.d_co + n0) * 4;               // scalar part of the store addresses
    auto epilogue = [&](auto edge_c, auto add_c) __attribute__((always_inline)) {
      constexpr bool EDGE = decltype(edge_c)::value != 0, ADD = decltype(add_c)::value != 0;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = !EDGE || mrow + i * 16 + fq * 4 + e < p.M;
          // (rows beyond M: every address is clamped to the tile's first row, nothing is stored)
          const size_t urow = ubase + (ok ? (size_t)((i * 16 + e) * p.d_cs) * 4 + voff : 0);
          float av[TN];
          if constexpr (ADD) {
#pragma unroll
            for (int j = 0; j < TN; ++j) av[j] = *reinterpret_cast<const float*>(addend + urow + j * 64);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float v = acc[i][j][e] + bj[j];
            if constexpr (ADD) {
              asm volatile("" : "+v"(v));      // (-ffast-math would re-associate bias + addend: keep conv_x6p.hip's rounding)
              v += av[j];
            }
            if (p.relu) v = fmaxf(v, 0.f);
            if (ok) *reinterpret_cast<float*>(dst + urow + j * 64) = v;
          }
        }
      }
      if (p.stats) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = !EDGE || mrow + i * 16 + fq * 4 + e < p.M;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const float sv = ok ? acc[i][j][e] + sj[j] : 0.f;
              csum[j] += sv; csq[j] += sv * sv;
            }
          }
      }
    };
    if (interior) { if (addend) epilogue(X6sInt<0>{}, X6sInt<1>{}); else epilogue(X6sInt<0>{}, X6sInt<0>{}); }
    else          { if (addend) epilogue(X6sInt<1>{}, X6sInt<1>{}); else epilogue(X6sInt<1>{}, X6sInt<0>{}); }
  }

  // ---- the block's column sums -> one slab row per slot; waves added in fixed order ------------------------------------
  float* red = reinterpret_cast<float*>(smem);             // [NW][2][BN] (the activation buffer: no DMA is in flight now)
  auto write_slab = [&](float (&s_)[TN], float (&q_)[TN], float* slab, int row_first) __attribute__((always_inline)) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s = s_[j], q = q_[j];
      s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
      if (fq == 0) {
        red[(wave * 2 + 0) * BN + j * 16 + fr] = s;
        red[(wave * 2 + 1) * BN + j * 16 + fr] = q;
      }
    }
    __syncthreads();
    for (int c = tid; c < 2 * BN; c += NT) {
      const int which = c / BN, col = c - which * BN;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[(w * 2 + which) * BN + col];
      slab[((size_t)(row_first + slot) * 2 + which) * p.dC + n0 + col] = s;
    }
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (p.stats) write_slab(csum, csq, p.stats, p.stats_row0);
}

// which = 1: DPA_X6S (default 1; 0 = these layers stay on conv_x6p.hip's kernel); 2: DPA_X6S_BLOCKS (default 512: resident
// blocks of a launch, two on each of the 256 CUs; the kernel tests set 16 so that small problems take the kernel).
// v >= 0 sets the knob and returns the previous value, v < 0 only reads it.
DPA_API int dpa_x6s_knob(int which, int v) {
  static int cur[3] = {0, [] { const char* e = getenv("DPA_X6S"); return e ? atoi(e) : 1; }(),
                          [] { const char* e = getenv("DPA_X6S_BLOCKS"); return e ? atoi(e) : 512; }()};
  if (which < 1 || which > 2) return -1;
  const int prev = cur[which];
  if (v >= 0) cur[which] = v;
  return prev;
}
static int x6s_slots(int tilesN) {
  const int n = (dpa_x6s_knob(2, -1) / tilesN) & ~7;
  return n < 8 ? 8 : n;
}

// 1x1, stride 1, no padding, same source and destination maps, 64 reduction elements, full column tiles, at least two
// M-tiles for every resident block.
bool dpa_x6s_eligible(const ConvParams& p) {
  if (dpa_x6s_knob(1, -1) <= 0) return false;
  const bool linear = p.nR * p.nS == 1 && p.R == 1 && p.S == 1 && p.sh == 1 && p.sw == 1 && p.ph == 0 && p.pw == 0 &&
                      p.sH == p.dH && p.sW == p.dW;
  // (a data gradient with the fused BatchNorm-backward reduction stays on conv_x6p.hip: its epilogue reads the BatchNorm's
  // input and output besides the addend - 14 GB per call on 56x56 x 256 channels at batch 1024 - and runs at 90 % of the
  // memory roofline there)
  if (!linear || !p.wgt3 || p.sC != 64 || p.Kg != 64 || p.no_uni || p.oscale || p.splitK > 1 || p.bnr_x) return false;
  if (!(p.dC == 64 || p.dC % 128 == 0)) return false;
  if ((p.s_cs | p.s_co) & 3) return false;
  const int tilesN = p.dC == 64 ? 1 : p.dC / 128;
  if (tilesN > 8) return false;
  const long long tilesM = ((long long)p.M + 127) / 128;
  // (lane offsets of the epilogue are 32-bit)
  if ((long long)p.d_cs * 16 * 4 >= 0x7fffffffLL) return false;
  return tilesM >= 2 * x6s_slots(tilesN);
}

// Launches the whole problem; returns the number of stats-slab rows it wrote.
int dpa_x6s_launch(ConvParams& p, int dgrad, hipStream_t s) {
  const bool narrow = p.dC == 64;
  p.tilesM = (p.M + 127) / 128;
  p.tilesN = narrow ? 1 : p.dC / 128;
  const int nslots = x6s_slots(p.tilesN);           // two resident blocks on each of the 256 CUs
  const dim3 grid(nslots * p.tilesN), block(256);
  const size_t lds = (size_t)2 * 128 * 128 + (size_t)6 * (narrow ? 64 : 128) * 64;
  (void)dgrad;      // (1x1, stride 1: the data gradient is the same GEMM on the transposed filter)
  if (narrow) hipLaunchKernelGGL((conv_x6s_kernel<4>), grid, block, lds, s, p, nslots);
  else        hipLaunchKernelGGL((conv_x6s_kernel<8>), grid, block, lds, s, p, nslots);
  return nslots;
}
