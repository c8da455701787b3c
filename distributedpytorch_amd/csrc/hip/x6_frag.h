// Fragment helpers shared by the pre-split fp32 (x6) kernels: conv_x6p.hip, conv_x6s.hip.
#pragma once
#include "conv_common.h"

// LDS offset of 16-byte chunk `chunk` of row `row` of the fp32 activation tile (128-byte rows).  Measured, not derived
// (scripts/micro/lds_conflict.hip under PMC, profiles/r2_lds_read_b128_lane_groups.txt): with this kernel's fragment
// addressing (lane (fr, fq) reads chunk 2 fq + ks of row fr) the generic chunk ^ ((row >> 1) & 7) swizzle of
// conv_common.h costs 8 LDS cycles per ds_read_b128 instead of 4 (12 % of this kernel's LDS cycles were bank conflicts);
// the row-pair index in chunk bits 0 and 2, leaving bit 1 to fq, costs 4.
__device__ __forceinline__ int x6p_a_swz(int row) { return ((row >> 1) & 1) | (((row >> 2) & 1) << 2); }
__device__ __forceinline__ int x6p_a_off(int row, int chunk) { return row * 128 + ((chunk ^ x6p_a_swz(row)) << 4); }

__device__ __forceinline__ void x6p_keep(const uint4& a, const uint4& b) {      // (diagnostic builds: keeps operands live)
  asm volatile("" :: "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w));
}
__device__ __forceinline__ void split3_frag(const uint4& lo, const uint4& hi, uint4 (&pl)[3]) {
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  uint32_t o[4][3];
#pragma unroll
  for (int q = 0; q < 4; ++q) split3_pair(__uint_as_float(w[2 * q]), __uint_as_float(w[2 * q + 1]), o[q]);
#pragma unroll
  for (int part = 0; part < 3; ++part) pl[part] = make_uint4(o[0][part], o[1][part], o[2][part], o[3][part]);
}

