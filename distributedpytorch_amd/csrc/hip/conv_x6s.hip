// fp32 (x6 arithmetic) 1x1 / stride-1 convolution with a 64-element reduction as a PERSISTENT streaming kernel ("x6s").
//
// The layers it is for (ResNet-50 at 56x56: 64 -> 256 forward, 256 -> 64 data gradient, 64 -> 64 both ways) read or write
// 3.3 GB per call at batch 1024 and do two K-tiles of matrix work per output tile.  In conv_x6p.hip every tile is a block
// of its own: address set-up, 32 KB of activations AND 48 KB of filter planes through the LDS-DMA, a wait for all of it,
// 192 MFMAs per wave, an epilogue staged through LDS - a chain of latencies that two resident blocks per CU overlap only
// by chance (profiles/r2_conv_layers_fp32_bs1024.txt: 85 TF-equivalent, 1434 us of which 1015 us without the stores and
// 545 us for the stores alone: the phases ADD).  Here:
//   * a block is resident for the whole launch and bound to ONE column tile: the filter planes of that column tile (all 64
//     reduction elements, 48 KB) are staged once; its waves walk over 32-row strips of the activations;
//   * the activation strip AFTER the next one is in flight (LDS-DMA into the wave's own buffer, issued as soon as the wave
//     has the next strip's fragments in registers) while the current strip is multiplied and stored;
//   * the epilogue stores straight from the accumulator layout - lanes 0..15 of a quarter wave hold 16 consecutive
//     channels of one pixel: 64-byte segments, one address per row + the column tile in the immediate offset - no LDS
//     staging, so the activation buffer belongs to the DMA during the whole epilogue;
//   * BatchNorm statistics are carried in registers across a wave's strips and written once per block: nslots slab rows
//     instead of one per M-tile.
// Same arithmetic as conv_x6p.hip (same fragment layouts, same order of the six partial products and of the K-tiles): the
// outputs are bit-identical to that kernel's.
#include "conv_common.h"
#include "x6_frag.h"

template <int V> struct X6sInt { static constexpr int value = V; };

// Four waves per block, each with a PRIVATE 8-KiB activation buffer: a wave stages, reads and multiplies its own 32-row
// strips (strip = ws, ws + nws, ...: ws = slot * 4 + wave) and never waits for another wave - there is no barrier in the
// loop, the waves drift apart and their loads, MFMAs and stores interleave on the CU.
// STATS / ADD: BatchNorm statistics / an addend in the epilogue (host: p.stats, p.addend).
// Measured and dropped (profiles/r3_x6s_streaming_kernel.txt): 8 waves in one block per CU (the same), 12 waves (spills;
// 10 % slower), 64-column tiles with 8 or 12 waves per CU (the operand split is done twice as often: 19 % slower), the
// loop without the rotation below (9 % slower with block-wide barriers, 2 % with private buffers), a counted vmcnt wait
// that leaves the previous strip's stores in flight (the same).
template <int TN, bool STATS, bool ADD>
__global__ void __launch_bounds__(256, 2) conv_x6s_kernel(const ConvParams p, const int nslots) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int NW = 4, TM = 2, BN = TN * 16, NT = NW * 64;
  constexpr int A_WAVE = 32 * 256;                 // a wave's strip: 32 rows x 64 fp32, as two K-tile halves of 4 KiB
  constexpr int A_BYTES = NW * A_WAVE;
  constexpr int BP_BYTES = BN * 64;                // one plane of one K-tile of the filter tile
  constexpr int B_KT = 3 * BP_BYTES;
  constexpr int B_PIECES = 6 * TN / NW;            // 1-KiB pieces staged by each wave, once
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) char lds_char;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  // blocks that share an XCD (blockIdx % 8) and are neighbours in it work on the same strips of different column tiles
  const int bq = (int)blockIdx.x >> 3;
  const int tn = bq % p.tilesN;
  const int slot = (bq / p.tilesN) * 8 + ((int)blockIdx.x & 7);
  const int n0 = tn * BN;
  const int nstrips = p.tilesM;                    // (host: strips of 32 rows)
  const int nws = nslots * NW, ws = slot * NW + wave;
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
  // ---- the wave's strip: lane -> slot lane & 7 of rows (lane >> 3) + 8 g, both K-tile halves
  //      ----------------------------------------------------------------------------------------------------------------
  const int row0 = lane >> 3;
  const int ck = (lane & 7) ^ x6p_a_swz(row0);     // (rows advance by 8 per piece: the swizzle is the same for every piece)
  const uint32_t ldsA = lds0 + wave * A_WAVE;
  // (wave-uniform base + 32-bit lane offset: the addresses cost scalar instructions only)
  const uint32_t lane_off = (uint32_t)(row0 * p.s_cs + p.s_co + ck * 4) * 4u;
  const char* const srcb = reinterpret_cast<const char*>(p.src);
  int nxt = ws;                                    // the strip the next issue_A stages
  auto issue_A = [&]() __attribute__((always_inline)) {
    const char* base = srcb + (size_t)nxt * 32 * (size_t)p.s_cs * 4;
    if ((long long)nxt * 32 + 32 <= p.M) {         // wave-uniform
#pragma unroll
      for (int g = 0; g < 4; ++g)
        dpa_glds16x2_s(base + (size_t)(8 * g) * (size_t)p.s_cs * 4, lane_off, ldsA + g * 1024, ldsA + 4096 + g * 1024);
    } else {                                       // rows beyond M re-read row M - 1: their results are never used
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long long m = (long long)nxt * 32 + row0 + 8 * g;
        const int r = (int)((m < p.M ? m : (long long)p.M - 1) - (long long)nxt * 32);      // 0 <= r < 32: the strip starts below M
        dpa_glds16x2_s(base, (uint32_t)(r * p.s_cs + p.s_co + ck * 4) * 4u, ldsA + g * 1024, ldsA + 4096 + g * 1024);
      }
    }
    nxt += nws;
  };
  if (nxt < nstrips) issue_A();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                 // the filter planes (the only block-wide wait before the final reduction)

  // ---- per-lane epilogue constants: lane (fr, fq) holds rows fq * 4 + e, column j * 16 + fr of every 16 x 16 tile --------
  float bj[TN], sj[TN], csum[TN], csq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + j * 16 + fr;
    bj[j] = p.bias ? p.bias[col] : 0.f;
    sj[j] = bj[j] - ((STATS && p.stats_shift) ? p.stats_shift[col] : 0.f);      // statistics are about the shift
    csum[j] = 0.f; csq[j] = 0.f;
  }
  char* const dst = reinterpret_cast<char*>(p.dst);
  const char* const addend = reinterpret_cast<const char*>(p.addend);
  const uint32_t voff = (uint32_t)(fq * 4 * p.d_cs + fr) * 4u;                     // lane part of every store address
  const int b_frag = fr * 64 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);

  // The loop is rotated: the fragments of strip t + 1 are fetched BETWEEN the MFMAs and the stores of strip t.  The wait in
  // front of them then covers a DMA that was issued before the MFMAs of strip t and the stores of strip t - 1 - nothing
  // recent - and the stores of strip t drain while strip t + 1 is multiplied.
  uint4 fa[2][TM][2];
  const char* const myA = smem + wave * A_WAVE;
  auto fetch_frags = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the strip's activations have landed (this wave's own DMA)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          fa[kt][i][h] = *reinterpret_cast<const uint4*>(myA + kt * 4096 + x6p_a_off(i * 16 + fr, 2 * fq + h));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // ... and are in registers: the buffer is free for the next strip
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          asm volatile("" : "+v"(fa[kt][i][h].x), "+v"(fa[kt][i][h].y), "+v"(fa[kt][i][h].z), "+v"(fa[kt][i][h].w));
    if (nxt < nstrips) issue_A();
  };
  if (ws < nstrips) fetch_frags();
  for (int strip = ws; strip < nstrips; strip += nws) {
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

    if (strip + nws < nstrips) fetch_frags();      // wave-uniform

    // ---- epilogue: straight from the accumulators; packed fp32 arithmetic on the row pairs (e, e + 1) ------------------
    const long long mrow = (long long)strip * 32;                                 // wave-uniform
    const size_t ubase = ((size_t)mrow * p.d_cs + p.d_co + n0) * 4;               // scalar part of the store addresses
    auto epilogue = [&](auto edge_c, auto relu_c) __attribute__((always_inline)) {
      constexpr bool EDGE = decltype(edge_c)::value != 0, RELU = decltype(relu_c)::value != 0;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        bool ok[4];
        size_t urow[4];                            // wave-uniform part of the row's addresses
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ok[e] = !EDGE || mrow + i * 16 + fq * 4 + e < p.M;
          urow[e] = ubase + (size_t)((i * 16 + e) * p.d_cs) * 4;
        }
        float av[4][TN];
        if constexpr (ADD) {      // every load of the row tile before the first store (the addend may BE the destination)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              av[e][j] = ok[e] ? *reinterpret_cast<const float*>(addend + urow[e] + (size_t)voff + j * 64) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const f32x2 lo = acc[i][j].lo, hi = acc[i][j].hi;
          f32x2 v0 = lo + bj[j], v1 = hi + bj[j];
          float v[4] = {v0.x, v0.y, v1.x, v1.y};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (ADD) {
              asm volatile("" : "+v"(v[e]));      // (-ffast-math would re-associate bias + addend: keep conv_x6p.hip's rounding)
              v[e] += av[e][j];
            }
            if constexpr (RELU) v[e] = fmaxf(v[e], 0.f);
            if (ok[e]) *reinterpret_cast<float*>(dst + urow[e] + (size_t)voff + j * 64) = v[e];
          }
        }
      }
      if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          bool ok[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) ok[e] = !EDGE || mrow + i * 16 + fq * 4 + e < p.M;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            f32x2 s0 = acc[i][j].lo + sj[j], s1 = acc[i][j].hi + sj[j];
            if constexpr (EDGE) {
              s0.x = ok[0] ? s0.x : 0.f; s0.y = ok[1] ? s0.y : 0.f; s1.x = ok[2] ? s1.x : 0.f; s1.y = ok[3] ? s1.y : 0.f;
            }
            const f32x2 t = s0 + s1, q = s0 * s0 + s1 * s1;      // (packed: rows e and e + 2, e + 1 and e + 3)
            csum[j] += t.x + t.y; csq[j] += q.x + q.y;
          }
        }
      }
    };
    if (mrow + 32 <= p.M) { if (p.relu) epilogue(X6sInt<0>{}, X6sInt<1>{}); else epilogue(X6sInt<0>{}, X6sInt<0>{}); }
    else                  { if (p.relu) epilogue(X6sInt<1>{}, X6sInt<1>{}); else epilogue(X6sInt<1>{}, X6sInt<0>{}); }
  }

  // ---- the block's column sums -> one slab row per slot; waves added in fixed order ------------------------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (STATS) {
    float* red = reinterpret_cast<float*>(smem);           // [NW][2][BN] (an activation buffer: no DMA is in flight now)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {                // (all shuffles, a wait, then the writes: see conv_igemm.hip's epilogue)
      csum[j] += __shfl_xor(csum[j], 16, 64); csum[j] += __shfl_xor(csum[j], 32, 64);
      csq[j] += __shfl_xor(csq[j], 16, 64); csq[j] += __shfl_xor(csq[j], 32, 64);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      if (fq == 0) {
        red[(wave * 2 + 0) * BN + j * 16 + fr] = csum[j];
        red[(wave * 2 + 1) * BN + j * 16 + fr] = csq[j];
      }
    __syncthreads();
    for (int c = tid; c < 2 * BN; c += NT) {
      const int which = c / BN, col = c - which * BN;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[(w * 2 + which) * BN + col];
      p.stats[((size_t)(p.stats_row0 + slot) * 2 + which) * p.dC + n0 + col] = s;
    }
  }
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
// 32-row strips for every resident wave.
bool dpa_x6s_eligible(const ConvParams& p) {
  if (dpa_x6s_knob(1, -1) <= 0) return false;
  const bool linear = p.nR * p.nS == 1 && p.R == 1 && p.S == 1 && p.sh == 1 && p.sw == 1 && p.ph == 0 && p.pw == 0 &&
                      p.sH == p.dH && p.sW == p.dW;
  // (a data gradient with the fused BatchNorm-backward reduction stays on conv_x6p.hip: its epilogue reads the BatchNorm's
  // input and output besides the addend - 14 GB per call on 56x56 x 256 channels at batch 1024 - and runs at 90 % of the
  // memory roofline there)
  if (!linear || !p.wgt3 || p.sC != 64 || p.Kg != 64 || p.no_uni || p.oscale || p.splitK > 1 || p.bnr_x || p.addend2) return false;
  if (!(p.dC == 64 || p.dC % 128 == 0)) return false;
  if (p.stats && p.addend) return false;      // (no layer of the zoo asks for both; the variant would spill registers)
  if ((p.s_cs | p.s_co) & 3) return false;
  const int tilesN = p.dC == 64 ? 1 : p.dC / 128;
  if (tilesN > 8) return false;
  const long long strips = ((long long)p.M + 31) / 32;
  // (lane offsets of the epilogue are 32-bit)
  if ((long long)p.d_cs * 16 * 4 >= 0x7fffffffLL || ((long long)p.s_cs * 32 + p.s_co + 64) * 4 >= 0x7fffffffLL) return false;
  return strips >= 2LL * x6s_slots(tilesN) * 4;
}

template <int TN>
static void x6s_launch(const ConvParams& p, int nslots, hipStream_t s) {
  constexpr size_t LDS = (size_t)4 * 32 * 256 + (size_t)6 * TN * 16 * 64;
  const dim3 grid(nslots * p.tilesN), block(256);
  if (p.stats) {      // (never with an addend: dpa_x6s_eligible)
    hipLaunchKernelGGL((conv_x6s_kernel<TN, true, false>), grid, block, LDS, s, p, nslots);
  } else {
    if (p.addend) hipLaunchKernelGGL((conv_x6s_kernel<TN, false, true>), grid, block, LDS, s, p, nslots);
    else          hipLaunchKernelGGL((conv_x6s_kernel<TN, false, false>), grid, block, LDS, s, p, nslots);
  }
}

// Launches the whole problem; returns the number of stats-slab rows it wrote.
int dpa_x6s_launch(ConvParams& p, int dgrad, hipStream_t s) {
  (void)dgrad;      // (1x1, stride 1: the data gradient is the same GEMM on the transposed filter)
  p.tilesM = (p.M + 31) / 32;                       // strips of 32 rows, one wave each
  p.tilesN = p.dC == 64 ? 1 : p.dC / 128;
  const int nslots = x6s_slots(p.tilesN);
  if (p.dC == 64) x6s_launch<4>(p, nslots, s);
  else            x6s_launch<8>(p, nslots, s);
  return nslots;
}
