// Last decoder layer of CleanUMamba, fused, for gfx950:  Conv1d(64 -> 128, 1x1) + GLU + ConvTranspose1d(64 -> 1, k 4, s 2)
// (src/network/CleanUMamba.py:121-130 with channels_output 1, channels_H 64: the E6 / E8 configurations of configs/exp/).
//
// With one output channel the transposed conv is four multiply-adds per GLU value, and the layer's 64-channel GLU
// output g (164 MB in f16 at E8 B = 16) is cheaper to REBUILD from the layer input u than to store and read back.  The
// generic path (gemm.hip / gemm_tn.hip) at E8 B = 16 (profiles/r02_gemm_table.txt, M = 1 282 048 rows):
//   forward   1x1+GLU GEMM writes g and the gate (101 us), the transposed-conv GEMM reads g (75 us)
//   backward  dWt = dY^T g (126 us) | dZ = GLU'(dY Wt) (reads gate and g, writes the 128-column dZ: 237 us) |
//             dW1 = dZ^T u (140 us) | dU = dZ W1 with the ReLU mask of the layer below (179 us): 682 us, 2.1 GB
// cum_dec7_fwd reads u once and writes the one-channel output; cum_dec7_bwd reads u, dY (one channel) and the mask bits
// once and writes dU and its masked copy: per 32 rows a workgroup rebuilds (a | b) = W1 u + b1 on the matrix cores,
// forms dG = dY x taps and dZ on the VALU into an LDS tile, and its four waves run dU = dZ W1, dW1 += dZ^T u,
// db1 += column sums of dZ (one more MFMA against ones), dWt / dbt += g x dY on the VALU.  Partial sums per workgroup go
// to f32 slabs; two small launches add them in a fixed order (deterministic) and write the two weight-gradient slots of
// the decoder stack's arena in the layouts cum_gemm_tn would have produced.
#include "outer_common.h"

namespace cum {

constexpr int D7_H = 64;          // channels of u and of the GLU output
constexpr int D7_J = 128;         // 1x1 output rows (packed GLU order: per 32 rows 16 a | 16 b)
constexpr int D7_R = 32;          // rows per step
constexpr int D7_ZS = 288;        // dZ tile row stride in bytes (256 + 32)
constexpr int D7_US = 144;        // 64-channel tile row stride in bytes (128 + 16)
constexpr int D7_SLAB = D7_J * D7_H + D7_J + D7_H * 4 + 2;   // dW1 | db1 | dWt[c][tap] | dbt (even, odd output rows)

// ---------------------------------------------------------------------------------------------------- forward
struct Dec7FwdParams {
  const void *u;       // input row buffer [1 + M + slack][64]
  const void *w1p;     // [128][64] T: the 1x1 weight in the forward GEMM's packed row order
  const float *b1p;    // [128] f32, packed order
  const float *wt;     // (64, 1, 4) f32: the transposed conv's weight as stored
  const float *bt;     // (1) f32
  void *out;           // output row buffer [1 + 2 M + slack][8]: channel 0 = the signal, channels 1-7 zero
  int64_t M;           // input rows (clips x pitch)
  int pitch, valid;    // rows per clip, real rows per clip (T); pair row p <= T produces output rows 2 p, 2 p + 1
  int64_t zero_tail;   // elements to clear behind output row 2 M
  int steps_per_wg;
};

// One workgroup walks a contiguous run of 32-row steps: output row 2 m + q needs the taps q of row m and q + 2 of row m - 1,
// so the last row's taps are carried from step to step (the run's first step is preceded by one silent step).
template <typename T>
__global__ __launch_bounds__(256) void dec7_fwd_kernel(const Dec7FwdParams p) {
  static_assert(sizeof(T) == 2, "16-bit element types only");
  __shared__ __attribute__((aligned(16))) unsigned char ut_all[2][D7_R * D7_US];
  __shared__ float4 pv_all[2][4][D7_R];                  // [step parity][wave][row]: the wave's 16 channels x 4 taps
  __shared__ float carry[2][2];                          // taps 2, 3 of the step's last row

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = uniform(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const T *u = static_cast<const T *>(p.u) + D7_H;      // row 0 of the buffer is the leading zero row
  T *out = static_cast<T *>(p.out) + 8;
  if (blockIdx.x == 0) {                                 // framing rows of the output buffer
    if (tid < 8) out[tid - 8] = (T)0.f;
    for (int64_t i = tid; i < p.zero_tail; i += 256) out[2 * p.M * 8 + i] = (T)0.f;
  }
  // A operand: packed weight rows j = 32 w + 16 jt + r, k = h = 32 ks + 8 g .. + 7
  e0_u32x4 wA[2][2];
  float ba[4], bb[4], wtr[4][4];
  {
    const T *w1 = static_cast<const T *>(p.w1p);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        wA[jt][ks] = *reinterpret_cast<const e0_u32x4 *>(w1 + (int64_t)(32 * w + 16 * jt + r) * D7_H + 32 * ks + 8 * g);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      ba[rr] = p.b1p[32 * w + 4 * g + rr];
      bb[rr] = p.b1p[32 * w + 16 + 4 * g + rr];
#pragma unroll
      for (int k = 0; k < 4; ++k) wtr[rr][k] = (float)(T)p.wt[(16 * w + 4 * g + rr) * 4 + k];   // the GEMM multiplies rounded weights
    }
  }
  const float btv = p.bt[0];
  const int64_t nsteps = (p.M + D7_R - 1) / D7_R;
  const int64_t s_begin = (int64_t)blockIdx.x * p.steps_per_wg;
  int64_t s_end = s_begin + p.steps_per_wg;
  s_end = s_end < nsteps ? s_end : nsteps;
  if (s_begin >= s_end) return;
  if (tid < 2) carry[1][tid] = 0.f;                      // run at row 0: row -1 is the leading zero row

  constexpr int PF = 4;
  uint4 uc[PF];
  auto fetch = [&](int64_t s, uint4 &c) {
    int64_t m = s * D7_R + (tid >> 3);
    m = m < p.M ? m : p.M - 1;                           // clamped row; masked by the validity test
    c = *reinterpret_cast<const uint4 *>(u + m * D7_H + (tid & 7) * 8);
  };
  int64_t s = s_begin > 0 ? s_begin - 1 : s_begin;       // the silent step
#pragma unroll
  for (int i = 0; i < PF; ++i) fetch(s + i < nsteps ? s + i : nsteps - 1, uc[i]);
  int par = 0;
  for (; s < s_end; ++s, par ^= 1) {
    unsigned char *ut = ut_all[par];
    const int64_t m0 = s * D7_R;
    const unsigned t0 = (unsigned)uniform((int)((unsigned)m0 % (unsigned)p.pitch));
    *reinterpret_cast<uint4 *>(ut + (tid >> 3) * D7_US + (tid & 7) * 16) = uc[0];
    __syncthreads();
    {
#pragma unroll
      for (int i = 0; i + 1 < PF; ++i) uc[i] = uc[i + 1];
      if (s + PF < nsteps) fetch(s + PF, uc[PF - 1]);
    }
    // (a | b)[j][row] = sum_h W1p[j][h] u[row][h]; lane: row 16 mt + r, channels 16 w + 4 g + rr
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      e0_f32x4 da = e0_f32x4{0.f, 0.f, 0.f, 0.f}, db = da;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const e0_u32x4 ub = *reinterpret_cast<const e0_u32x4 *>(ut + (16 * mt + r) * D7_US + (4 * ks + g) * 16);
        da = e0_mfma<T>(wA[0][ks], ub, da);
        db = e0_mfma<T>(wA[1][ks], ub, db);
      }
      const bool ok = e0_row_ok(t0, 16 * mt + r, m0, p.M, (unsigned)p.pitch, (unsigned)p.valid);
      float pv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float gv = e0_keep(ok, (float)(T)((da[rr] + ba[rr]) * sigmoidf_(db[rr] + bb[rr])));   // g as the GEMM would read it
#pragma unroll
        for (int k = 0; k < 4; ++k) pv[k] = fmaf(gv, wtr[rr][k], pv[k]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {                      // the four channel groups g of a row sit in lanes r, r + 16, r + 32, r + 48
        pv[k] += __shfl_xor(pv[k], 16, 64);
        pv[k] += __shfl_xor(pv[k], 32, 64);
      }
      if (g == 0) pv_all[par][w][16 * mt + r] = make_float4(pv[0], pv[1], pv[2], pv[3]);
    }
    __syncthreads();
    if (tid < 64) {
      const int row = tid >> 1, q = tid & 1;
      float cur[4] = {0.f, 0.f, 0.f, 0.f}, prv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) {
        const float4 c = pv_all[par][ww][row];
        cur[0] += c.x; cur[1] += c.y; cur[2] += c.z; cur[3] += c.w;
        if (row > 0) {
          const float4 d = pv_all[par][ww][row - 1];
          prv[2] += d.z; prv[3] += d.w;
        }
      }
      if (row == 0) {
        prv[2] = carry[par ^ 1][0];
        prv[3] = carry[par ^ 1][1];
      }
      if (row == D7_R - 1) carry[par][q] = cur[2 + q];
      const int64_t m = m0 + row;
      if (s >= s_begin && m < p.M) {
        const bool ok = e0_row_ok(t0, row, m0, p.M, (unsigned)p.pitch, (unsigned)p.valid + 1u);
        const float v = ok ? cur[q] + prv[2 + q] + btv : 0.f;
        *reinterpret_cast<uint4 *>(out + (2 * m + q) * 8) = make_uint4(e0_pack2<T>(v, 0.f), 0u, 0u, 0u);
      }
    }
    // (double-buffered: the next step writes the other tile / tap array / carry; its barriers order the reuse of these)
  }
}

// ---------------------------------------------------------------------------------------------------- backward
struct Dec7BwdParams {
  const void *u;       // input row buffer [1 + M + slack][64]
  const void *dy;      // gradient of the output row buffer [1 + 2 M + slack][8] (channel 0)
  const unsigned char *mask;   // sign bits of the layer below's ReLU, data row 0 first: 16 bytes per row (4 channels per byte, low nibble)
  const void *w1p;     // [128][64] T packed
  const float *b1p;    // [128] f32 packed
  const float *wt;     // (64, 1, 4) f32
  void *du;            // gradient of u: row buffer [1 + M + slack][64]
  void *dpre;          // du where the mask bit is set, else 0: same geometry
  float *slabs;        // [gridDim.x][D7_SLAB]
  int64_t M;
  int pitch, valid;
  int64_t zero_tail;   // elements to clear behind row M of du / dpre
};

template <typename T>
__global__ __launch_bounds__(256) void dec7_bwd_kernel(const Dec7BwdParams p) {
  static_assert(sizeof(T) == 2, "16-bit element types only");
  // one array: [2 buffers][u tile | dZ tile | dU tile | dpre tile | dY samples 80 f32 | mask 512 B]
  constexpr int BUF = D7_R * D7_US + D7_R * D7_ZS + 2 * D7_R * D7_US + 320 + 512;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = uniform(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const T *u = static_cast<const T *>(p.u) + D7_H;
  const T *dy = static_cast<const T *>(p.dy);
  T *du = static_cast<T *>(p.du) + D7_H, *dpre = static_cast<T *>(p.dpre) + D7_H;
  if (blockIdx.x == 0) {
    for (int i = tid; i < D7_H; i += 256) du[i - D7_H] = dpre[i - D7_H] = (T)0.f;
    for (int64_t i = tid; i < p.zero_tail; i += 256) du[p.M * D7_H + i] = dpre[p.M * D7_H + i] = (T)0.f;
  }
  e0_u32x4 wA[2][2], wB[4];
  float ba[4], bb[4], wtr[4][4];
  {
    const T *w1 = static_cast<const T *>(p.w1p);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        wA[jt][ks] = *reinterpret_cast<const e0_u32x4 *>(w1 + (int64_t)(32 * w + 16 * jt + r) * D7_H + 32 * ks + 8 * g);
    // B operand of dU = dZ W1: k = j = 32 ks + 8 g .. + 7, n = h = 16 w + r
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      unsigned v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = 32 * ks + 8 * g + 2 * q;
        const unsigned short lo = __builtin_bit_cast(unsigned short, w1[(int64_t)j * D7_H + 16 * w + r]);
        const unsigned short hi = __builtin_bit_cast(unsigned short, w1[(int64_t)(j + 1) * D7_H + 16 * w + r]);
        v[q] = (unsigned)lo | ((unsigned)hi << 16);
      }
      wB[ks] = e0_u32x4{v[0], v[1], v[2], v[3]};
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      ba[rr] = p.b1p[32 * w + 4 * g + rr];
      bb[rr] = p.b1p[32 * w + 16 + 4 * g + rr];
#pragma unroll
      for (int k = 0; k < 4; ++k) wtr[rr][k] = (float)(T)p.wt[(16 * w + 4 * g + rr) * 4 + k];
    }
  }
  const unsigned one2 = __is_same(T, f16) ? 0x3C003C00u : 0x3F803F80u;
  const e0_u32x4 ones = e0_u32x4{one2, one2, one2, one2};

  e0_f32x4 acc1[2][4];     // dW1[j = 32 w + 16 it + 4 g + rr][h = 16 nt + r]
  e0_f32x4 accb[2];        // db1[j = 32 w + 16 it + 4 g + rr] (every column holds the same sum)
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    accb[it] = e0_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc1[it][nt] = e0_f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float awt[4][4];         // dWt[c = 16 w + 4 g + rr][tap] partial sums over this lane's rows
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int k = 0; k < 4; ++k) awt[rr][k] = 0.f;
  float sdy = 0.f;         // wave 0: sum of dY over output rows of this lane's parity

  const int64_t nsteps = (p.M + D7_R - 1) / D7_R;
  constexpr int PF = 3;
  uint4 uc[PF];
  float xs[PF];
  unsigned mk[PF];
  auto fetch = [&](int64_t s, uint4 &c, float &x, unsigned &mw) {
    const int64_t m0 = s * D7_R;
    int64_t m = m0 + (tid >> 3);
    m = m < p.M ? m : p.M - 1;
    c = *reinterpret_cast<const uint4 *>(u + m * D7_H + (tid & 7) * 8);
    x = 0.f;
    if (tid < 72) {                                      // output rows 2 m0 .. 2 m0 + 65 (+ a few: clamped)
      int64_t row = 1 + 2 * m0 + tid;
      const int64_t last = 2 * p.M + 2;                  // the buffer holds at least 1 + 2 M + 2 rows
      row = row < last ? row : last;
      x = (float)dy[row * 8];
    }
    mw = 0u;
    if (tid >= 128) {                                    // 32 rows x 4 words of sign bits
      int64_t mm = m0 + ((tid - 128) >> 2);
      mm = mm < p.M ? mm : p.M - 1;
      mw = *reinterpret_cast<const unsigned *>(p.mask + mm * 16 + (tid & 3) * 4);
    }
  };
  int64_t s = blockIdx.x;
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int64_t sf = s + (int64_t)i * gridDim.x;
    fetch(sf < nsteps ? sf : nsteps - 1, uc[i], xs[i], mk[i]);
  }
  int par = 0;
  for (; s < nsteps; s += gridDim.x, par ^= 1) {
    unsigned char *ut = lds + par * BUF, *zt = ut + D7_R * D7_US, *dut = zt + D7_R * D7_ZS, *dpt = dut + D7_R * D7_US;
    float *xt = reinterpret_cast<float *>(dpt + D7_R * D7_US);
    unsigned *mt_ = reinterpret_cast<unsigned *>(xt + 80);
    const int64_t m0 = s * D7_R;
    const unsigned t0 = (unsigned)uniform((int)((unsigned)m0 % (unsigned)p.pitch));
    *reinterpret_cast<uint4 *>(ut + (tid >> 3) * D7_US + (tid & 7) * 16) = uc[0];
    if (tid < 72) xt[tid] = 2 * m0 + tid < 2 * p.M ? xs[0] : 0.f;
    if (tid >= 128) mt_[tid - 128] = mk[0];
    __syncthreads();
    {
#pragma unroll
      for (int i = 0; i + 1 < PF; ++i) {
        uc[i] = uc[i + 1];
        xs[i] = xs[i + 1];
        mk[i] = mk[i + 1];
      }
      const int64_t sf = s + (int64_t)PF * gridDim.x;
      if (sf < nsteps) fetch(sf, uc[PF - 1], xs[PF - 1], mk[PF - 1]);
    }
    if (w == 0) sdy += xt[lane];                          // output rows 2 m0 + lane: parity = lane & 1
    // ---- rebuild (a | b), g; dG = dY x taps; dZ -> LDS; dWt += g x dY
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      e0_f32x4 da = e0_f32x4{0.f, 0.f, 0.f, 0.f}, db = da;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const e0_u32x4 ub = *reinterpret_cast<const e0_u32x4 *>(ut + (16 * mt + r) * D7_US + (4 * ks + g) * 16);
        da = e0_mfma<T>(wA[0][ks], ub, da);
        db = e0_mfma<T>(wA[1][ks], ub, db);
      }
      const int row = 16 * mt + r;
      const bool ok = e0_row_ok(t0, row, m0, p.M, (unsigned)p.pitch, (unsigned)p.valid);
      const float d0 = xt[2 * row], d1 = xt[2 * row + 1], d2 = xt[2 * row + 2], d3 = xt[2 * row + 3];
      float za[4], zb[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float sg = sigmoidf_(db[rr] + bb[rr]);
        const float gv = e0_keep(ok, (float)(T)((da[rr] + ba[rr]) * sg));      // the stored GLU output of the generic path
        float dg = d0 * wtr[rr][0];
        dg = fmaf(d1, wtr[rr][1], dg);
        dg = fmaf(d2, wtr[rr][2], dg);
        dg = fmaf(d3, wtr[rr][3], dg);
        za[rr] = e0_keep(ok, dg * sg);                   // da = d sig(b)
        zb[rr] = dg * gv * (1.f - sg);                   // db = d y (1 - sig(b)), y = the rounded output (0 on padding rows)
        awt[rr][0] = fmaf(gv, d0, awt[rr][0]);
        awt[rr][1] = fmaf(gv, d1, awt[rr][1]);
        awt[rr][2] = fmaf(gv, d2, awt[rr][2]);
        awt[rr][3] = fmaf(gv, d3, awt[rr][3]);
      }
      *reinterpret_cast<uint2 *>(zt + row * D7_ZS + (32 * w + 4 * g) * 2) = make_uint2(e0_pack2<T>(za[0], za[1]), e0_pack2<T>(za[2], za[3]));
      *reinterpret_cast<uint2 *>(zt + row * D7_ZS + (32 * w + 16 + 4 * g) * 2) = make_uint2(e0_pack2<T>(zb[0], zb[1]), e0_pack2<T>(zb[2], zb[3]));
    }
    __syncthreads();
    // ---- dU[row][h = 16 w + r] = sum_j dZ[row][j] W1p[j][h]: A rows 16 mt + r, k chunk 4 ks + g
    {
      e0_f32x4 d1[2] = {e0_f32x4{0.f, 0.f, 0.f, 0.f}, e0_f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const e0_u32x4 a = *reinterpret_cast<const e0_u32x4 *>(zt + (16 * mt + r) * D7_ZS + (4 * ks + g) * 16);
          d1[mt] = e0_mfma<T>(a, wB[ks], d1[mt]);
        }
      // d1[mt][rr] = row 16 mt + 4 g + rr, channel h = 16 w + r
      const int h = 16 * w + r;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = 16 * mt + 4 * g + rr;
          const unsigned bit = (reinterpret_cast<const unsigned char *>(mt_)[row * 16 + (h >> 2)] >> (h & 3)) & 1u;
          const T v = (T)d1[mt][rr];
          *reinterpret_cast<T *>(dut + row * D7_US + h * 2) = v;
          *reinterpret_cast<T *>(dpt + row * D7_US + h * 2) = bit ? v : (T)0.f;
        }
    }
    // ---- dW1[j][h] += sum_rows dZ[row][j] u[row][h]: transposing reads of the two row-major tiles
    {
      const int q = r >> 2, pp = r & 3;
      e0_u32x4 af[2], bf[4];
#pragma unroll
      for (int it = 0; it < 2; ++it) af[it] = e0_tr_read<D7_ZS>(zt + (8 * g + q) * D7_ZS + (32 * w + 16 * it + 4 * pp) * 2);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bf[nt] = e0_tr_read<D7_US>(ut + (8 * g + q) * D7_US + (16 * nt + 4 * pp) * 2);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc1[it][nt] = e0_mfma<T>(af[it], bf[nt], acc1[it][nt]);
        accb[it] = e0_mfma<T>(af[it], ones, accb[it]);
      }
    }
    __syncthreads();
    {
      const int row = tid >> 3, ch = tid & 7;            // one 128-byte row per 8 lanes
      const int64_t m = m0 + row;
      if (m < p.M) {
        *reinterpret_cast<uint4 *>(du + m * D7_H + ch * 8) = *reinterpret_cast<const uint4 *>(dut + row * D7_US + ch * 16);
        *reinterpret_cast<uint4 *>(dpre + m * D7_H + ch * 8) = *reinterpret_cast<const uint4 *>(dpt + row * D7_US + ch * 16);
      }
    }
    // (the next step writes the OTHER buffer; its barriers order this step's reads before the step after it)
  }

  // ---- slabs: dW1 | db1 | dWt | dbt of this workgroup
  float *slab = p.slabs + (int64_t)blockIdx.x * D7_SLAB;
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int j = 32 * w + 16 * it + 4 * g + rr;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) slab[j * D7_H + 16 * nt + r] = acc1[it][nt][rr];
      if (r == 0) slab[D7_J * D7_H + j] = accb[it][rr];
    }
  // dWt: a channel's rows sit in the 16 lanes r of one lane group g
  float *s1 = slab + D7_J * D7_H + D7_J;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float v = row16_allsum(awt[rr][k]);
      if (r == 0) s1[(16 * w + 4 * g + rr) * 4 + k] = v;
    }
  if (w == 0) {                                          // dbt: even / odd output rows (lanes of one parity)
    sdy += __shfl_xor(sdy, 2, 64);
    sdy += __shfl_xor(sdy, 4, 64);
    sdy += __shfl_xor(sdy, 8, 64);
    sdy += __shfl_xor(sdy, 16, 64);
    sdy += __shfl_xor(sdy, 32, 64);
    if (lane < 2) s1[D7_H * 4 + lane] = sdy;
  }
}

// slabs -> the two arena slots of the decoder stack (layouts of cum_gemm_tn): slot_w1 = dW1 [128][64] then db1 [128];
// slot_wt = dW [16][128] then db [16] of the transposed conv seen as a GEMM with N = (output-row parity q, 8 channels),
// K = (pair half hh: row m - 1 | row m, 64 channels): element (8 q, 64 hh + c) = tap q + 2 - 2 hh of channel c, db[8 q] =
// the sum of dY over output rows of parity q; the rows of the 7 padding channels are zero.  Two passes in index order.
constexpr int D7_RG = 16;
__global__ __launch_bounds__(256) void dec7_bwd_reduce1_kernel(const float *__restrict__ slabs, int nslabs,
                                                              float *__restrict__ part) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= D7_SLAB) return;
  const int per = (nslabs + D7_RG - 1) / D7_RG;
  const int lo = blockIdx.y * per, hi = lo + per < nslabs ? lo + per : nslabs;
  float a = 0.f, b = 0.f;
  int i = lo;
  for (; i + 1 < hi; i += 2) {
    a += slabs[(int64_t)i * D7_SLAB + e];
    b += slabs[(int64_t)(i + 1) * D7_SLAB + e];
  }
  if (i < hi) a += slabs[(int64_t)i * D7_SLAB + e];
  part[(int64_t)blockIdx.y * D7_SLAB + e] = a + b;
}

__global__ __launch_bounds__(256) void dec7_bwd_reduce2_kernel(const float *__restrict__ part,
                                                              float *__restrict__ slot_w1, float *__restrict__ slot_wt) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  constexpr int n1 = D7_J * D7_H + D7_J, nt = 16 * 128 + 16;
  if (e >= n1 + nt) return;
  int src = e;                                           // element of the slab this output is the sum of (-1: zero)
  if (e >= n1) {
    const int o = e - n1;
    if (o < 16 * 128) {
      const int n = o >> 7, k = o & 127;
      const int q = n >> 3, hh = k >> 6, c = k & 63;
      src = (n & 7) == 0 ? n1 + c * 4 + q + 2 - 2 * hh : -1;
    } else {
      const int n = o - 16 * 128;
      src = (n & 7) == 0 ? n1 + D7_H * 4 + (n >> 3) : -1;
    }
  }
  float v = 0.f;
  if (src >= 0) {
#pragma unroll
    for (int i = 0; i < D7_RG; ++i) v += part[(int64_t)i * D7_SLAB + src];
  }
  if (e < n1) slot_w1[e] = v;
  else slot_wt[e - n1] = v;
}

}  // namespace cum

using namespace cum;

static int dec7_fwd_workgroups(int64_t M) {
  const int64_t steps = (M + D7_R - 1) / D7_R;
  return (int)(steps < 1024 ? steps : 1024);             // four resident workgroups per CU (112 VGPRs)
}

extern "C" int cum_dec7_fwd(int32_t dtype, int64_t M, int32_t pitch, int32_t valid, const void *u, const void *w1p,
                            const float *b1p, const float *wt, const float *bt, void *out, int64_t zero_tail, void *stream) {
  CUM_REQUIRE(is16(dtype), "dec7_fwd: 16-bit element types only (f32 takes the generic path)");
  CUM_REQUIRE(M > 0 && M < 1073741823LL && pitch >= 32 && valid > 0 && valid < pitch && zero_tail >= 0, "dec7_fwd: bad geometry");
  CUM_REQUIRE(u && w1p && b1p && wt && bt && out, "dec7_fwd: null tensor");
  CUM_REQUIRE(((uintptr_t)u & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)w1p & 15) == 0,
              "dec7_fwd: u, out and w1p must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  Dec7FwdParams p{};
  p.u = u; p.w1p = w1p; p.b1p = b1p; p.wt = wt; p.bt = bt; p.out = out; p.M = M; p.pitch = pitch; p.valid = valid;
  p.zero_tail = zero_tail;
  const int64_t steps = (M + D7_R - 1) / D7_R;
  int nwg = dec7_fwd_workgroups(M);
  p.steps_per_wg = (int)((steps + nwg - 1) / nwg);
  nwg = (int)((steps + p.steps_per_wg - 1) / p.steps_per_wg);
  if (dtype == CUM_F16)
    hipLaunchKernelGGL(dec7_fwd_kernel<f16>, dim3(nwg), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(dec7_fwd_kernel<__bf16>, dim3(nwg), dim3(256), 0, st, p);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

extern "C" int32_t cum_dec7_bwd_workgroups(int64_t M) {
  const int64_t steps = (M + D7_R - 1) / D7_R;
  return (int32_t)(steps < 1 ? 1 : (steps > 512 ? 512 : steps));    // two resident workgroups per CU
}

extern "C" int64_t cum_dec7_bwd_workspace_elems(int64_t M) {
  return ((int64_t)cum_dec7_bwd_workgroups(M) + D7_RG) * D7_SLAB;
}

extern "C" int cum_dec7_bwd(int32_t dtype, int64_t M, int32_t pitch, int32_t valid, const void *dY, const void *u,
                            const void *mask, const void *w1p, const float *b1p, const float *wt, void *dU, void *dpre,
                            int64_t zero_tail, float *slot_w1, float *slot_wt, float *workspace, void *stream) {
  CUM_REQUIRE(is16(dtype), "dec7_bwd: 16-bit element types only (f32 takes the generic path)");
  CUM_REQUIRE(M > 0 && M < 1073741823LL && pitch >= 32 && valid > 0 && valid < pitch && zero_tail >= 0, "dec7_bwd: bad geometry");
  CUM_REQUIRE(dY && u && mask && w1p && b1p && wt && dU && dpre && slot_w1 && slot_wt && workspace, "dec7_bwd: null tensor");
  CUM_REQUIRE(((uintptr_t)u & 15) == 0 && ((uintptr_t)dU & 15) == 0 && ((uintptr_t)dpre & 15) == 0 && ((uintptr_t)w1p & 15) == 0 &&
                  ((uintptr_t)mask & 3) == 0, "dec7_bwd: u, dU, dpre and w1p must be 16-byte aligned, mask 4-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  Dec7BwdParams p{};
  p.u = u; p.dy = dY; p.mask = static_cast<const unsigned char *>(mask); p.w1p = w1p; p.b1p = b1p; p.wt = wt; p.du = dU;
  p.dpre = dpre; p.slabs = workspace; p.M = M; p.pitch = pitch; p.valid = valid; p.zero_tail = zero_tail;
  const int nwg = cum_dec7_bwd_workgroups(M);
  if (dtype == CUM_F16)
    hipLaunchKernelGGL(dec7_bwd_kernel<f16>, dim3(nwg), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(dec7_bwd_kernel<__bf16>, dim3(nwg), dim3(256), 0, st, p);
  CUM_CHECK_LAUNCH();
  float *part = workspace + (int64_t)nwg * D7_SLAB;
  hipLaunchKernelGGL(dec7_bwd_reduce1_kernel, dim3((D7_SLAB + 255) / 256, D7_RG), dim3(256), 0, st, workspace, nwg, part);
  constexpr int nout = D7_J * D7_H + D7_J + 16 * 128 + 16;
  hipLaunchKernelGGL(dec7_bwd_reduce2_kernel, dim3((nout + 255) / 256), dim3(256), 0, st, part, slot_w1, slot_wt);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}
