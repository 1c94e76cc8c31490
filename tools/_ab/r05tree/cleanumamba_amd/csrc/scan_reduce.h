// Cross-lane reductions of the selective-scan backward kernels (scan_bwd.hip, scan_bwd_small.hip): per-step sums of the
// dB / dC contributions over the 64 channels of a wave.
#pragma once
#include "scan_common.h"

namespace cum {

constexpr int NP2 = NS / 2;   // state pairs per wave (f2: scan_common.h)
static_assert(NS == 8, "s_a holds the 8 states of a lane as two float4");
__device__ __forceinline__ f2 exp2_2(f2 v) {
  f2 r;
  r.x = __builtin_amdgcn_exp2f(v.x);
  r.y = __builtin_amdgcn_exp2f(v.y);
  return r;
}

// Sum 8 per-lane values (4 pairs) over the 64 lanes.  On return lanes of row q = lane>>4 hold in
// r[0], r[1] the totals of v[2q], v[2q+1].  (Kept on scalars: one asm block per exchange and scalar adds cost
// fewer register copies than grouped exchanges with packed adds.)
__device__ __forceinline__ void wave_reduce_scatter8(const f2 (&v2)[NP2], float (&r)[2]) {
  const float v[NS] = {v2[0].x, v2[0].y, v2[1].x, v2[1].y, v2[2].x, v2[2].y, v2[3].x, v2[3].y};
  float h[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // after the exchange a + b = pair sums of v[i] (lanes < 32) / of v[4+i] (lanes >= 32)
    float a = v[i], b = v[4 + i];
    swap32(a, b);
    h[i] = a + b;
  }
  float q[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    // even rows keep the sums of h[i], odd rows those of h[2+i]
    float a = h[i], b = h[2 + i];
    swap16(a, b);
    q[i] = a + b;
  }
  // row q now holds 4-lane sums of v[4*(q>>1) + 2*(q&1) + i] = v[2q + i]
  r[0] = row16_allsum(q[0]);
  r[1] = row16_allsum(q[1]);
}

// Both per-step reductions (dB and dC) at once: same exchanges and the same result placement as two calls of
// wave_reduce_scatter8, but the eight v_permlane32_swap (and the four v_permlane16_swap) are issued back to back from
// ONE asm block each.  The "VALU write -> permlane swap read" hazard needs two wait states only in front of the first
// swap of a block (later swaps touch registers no neighbour wrote), so a step pays 2 s_nop instead of 12
// (192 -> 32 per 16-step chunk; the s_nop 1 pads were ~14 % of the kernel's issue cycles).
__device__ __forceinline__ void wave_reduce_scatter8x2(const f2 (&b2)[NP2], const f2 (&c2)[NP2], float (&rB)[2],
                                                       float (&rC)[2]) {
  float a[8] = {b2[0].x, b2[0].y, b2[1].x, b2[1].y, c2[0].x, c2[0].y, c2[1].x, c2[1].y};
  float b[8] = {b2[2].x, b2[2].y, b2[3].x, b2[3].y, c2[2].x, c2[2].y, c2[3].x, c2[3].y};
  // a[i] <-> v[i], b[i] <-> v[4 + i] of each array (indices 0-3: dB, 4-7: dC)
  asm volatile(
      "s_nop 1\n\t"
      "v_permlane32_swap_b32 %0, %8\n\t"
      "v_permlane32_swap_b32 %1, %9\n\t"
      "v_permlane32_swap_b32 %2, %10\n\t"
      "v_permlane32_swap_b32 %3, %11\n\t"
      "v_permlane32_swap_b32 %4, %12\n\t"
      "v_permlane32_swap_b32 %5, %13\n\t"
      "v_permlane32_swap_b32 %6, %14\n\t"
      "v_permlane32_swap_b32 %7, %15"
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b[0]),
        "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
  float h[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = a[i] + b[i];
  // per array: even rows keep the sums of h[i], odd rows those of h[2 + i]
  float e[4] = {h[0], h[1], h[4], h[5]}, o[4] = {h[2], h[3], h[6], h[7]};
  asm volatile(
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %0, %4\n\t"
      "v_permlane16_swap_b32 %1, %5\n\t"
      "v_permlane16_swap_b32 %2, %6\n\t"
      "v_permlane16_swap_b32 %3, %7"
      : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
  rB[0] = row16_allsum(e[0] + o[0]);
  rB[1] = row16_allsum(e[1] + o[1]);
  rC[0] = row16_allsum(e[2] + o[2]);
  rC[1] = row16_allsum(e[3] + o[3]);
}

// Same exchanges, but the last four values are reduce-SCATTERED over the 16 lanes of a row instead of summed four times:
// DPP adds whose bank mask writes only the quads that keep the value (row_mirror: lanes 0-7 keep the dB pair, 8-15 the
// dC pair; row_half_mirror: even quads keep the first of the pair), then two quad steps -- 8 DPP adds instead of 16 and
// no result copies.  On return lane l of row q holds, in every lane of its quad, the total of
//   dB[2q] (l & 15 in 0-3), dB[2q + 1] (4-7), dC[2q] (8-11), dC[2q + 1] (12-15).
__device__ __forceinline__ float wave_reduce_scatter8x2q_s(float (&a)[8], float (&b)[8]);
__device__ __forceinline__ float wave_reduce_scatter8x2q(const f2 (&b2)[NP2], const f2 (&c2)[NP2]) {
  float a[8] = {b2[0].x, b2[0].y, b2[1].x, b2[1].y, c2[0].x, c2[0].y, c2[1].x, c2[1].y};
  float b[8] = {b2[2].x, b2[2].y, b2[3].x, b2[3].y, c2[2].x, c2[2].y, c2[3].x, c2[3].y};
  return wave_reduce_scatter8x2q_s(a, b);
}
// The same on sixteen scalars: a[0..3] = dB values 0-3, a[4..7] = dC values 0-3, b[0..3] = dB values 4-7, b[4..7] = dC values
// 4-7.  (Callers that form the values with scalar multiplies hand the exchanges sixteen free-standing registers; halves of
// packed results cost a register copy each for eight of them.)
__device__ __forceinline__ float wave_reduce_scatter8x2q_s(float (&a)[8], float (&b)[8]) {
  // ONE asm block, instructions ordered so that every hazard distance (VALU write -> permlane swap / DPP read: two wait
  // states) is covered by the neighbouring instructions: four s_nop per step instead of six, and no compiler-placed adds
  // between blocks.  h_i = a_i + b_i after the half exchange; e = {h0, h1, h4, h5}, o = {h2, h3, h6, h7} meet in the row
  // exchange; q0..q3 = a0, a1, a4, a5 enter the bank-masked DPP tail.
  // (b0, b1, b2 are dead after the first adds and serve as the tail's two intermediates and its result: no register beyond
  //  the sixteen operands)
  asm volatile(
      "s_nop 1\n\t"
      "v_permlane32_swap_b32 %0, %8\n\t"
      "v_permlane32_swap_b32 %1, %9\n\t"
      "v_permlane32_swap_b32 %2, %10\n\t"
      "v_permlane32_swap_b32 %3, %11\n\t"
      "v_permlane32_swap_b32 %4, %12\n\t"
      "v_permlane32_swap_b32 %5, %13\n\t"
      "v_permlane32_swap_b32 %6, %14\n\t"
      "v_permlane32_swap_b32 %7, %15\n\t"
      "v_add_f32 %0, %0, %8\n\t"
      "v_add_f32 %1, %1, %9\n\t"
      "v_add_f32 %2, %2, %10\n\t"
      "v_add_f32 %3, %3, %11\n\t"
      "v_add_f32 %4, %4, %12\n\t"
      "v_add_f32 %5, %5, %13\n\t"
      "v_add_f32 %6, %6, %14\n\t"
      "v_add_f32 %7, %7, %15\n\t"
      "v_permlane16_swap_b32 %0, %2\n\t"
      "v_permlane16_swap_b32 %1, %3\n\t"
      "v_permlane16_swap_b32 %4, %6\n\t"
      "v_permlane16_swap_b32 %5, %7\n\t"
      "v_add_f32 %0, %0, %2\n\t"
      "v_add_f32 %1, %1, %3\n\t"
      "v_add_f32 %4, %4, %6\n\t"
      "v_add_f32 %5, %5, %7\n\t"
      "v_add_f32_dpp %8, %0, %0 row_mirror row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %9, %1, %1 row_mirror row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %8, %4, %4 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %9, %5, %5 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %10, %8, %8 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %10, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %10, %10, %10 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %10, %10, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b[0]),
        "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
  return b[2];
}

}  // namespace cum
