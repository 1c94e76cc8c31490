// First encoder layer of CleanUMamba, fused, for gfx950:  Conv1d(1 -> 64, k 4, s 2) + ReLU + Conv1d(64 -> 128, 1x1) + GLU
// (src/network/CleanUMamba.py:108-113 with channels_input 1, channels_H 64: the E6 / E8 configurations of configs/exp/).
//
// With one input channel the strided conv is four multiply-adds per output value: its ReLU output y1 (64 channels x
// 80 126 steps x 16 clips = 164 MB in f16) is cheaper to REBUILD from the 4 input samples than to read back.  The
// generic path (gemm.hip / gemm_tn.hip) stores y1 in the forward and moves it three more times in the backward:
//   forward   conv GEMM (K = 32, 7/8 of it zero padding) writes y1, the 1x1+GLU GEMM reads it
//   backward  dW2 = dZ^T y1 (reads dZ, y1) | dY1 = (dZ W2) . [y1 > 0] (reads dZ, y1, writes dY1) | dW1 = dY1^T X
//             (reads dY1): 1.6 GB in three launches, 445 us at E8 B = 16 (profiles/r02_gemm_table.txt)
// cum_enc0_bwd reads dZ ONCE (328 MB) and nothing else of size: per 32 rows a workgroup rebuilds y1 on the VALU into LDS,
// and its four waves run  dY1 = dZ W2 (each wave 16 of the 64 channels, masked by y1 > 0),  dW2 += dZ^T y1 (each wave 32
// of the 128 rows of dW2, operands by ds_read_b64_tr_b16 from the row-major tiles),  db2 += column sums of dZ (one more
// MFMA against a ones operand),  dW1 / db1 += dY1 x taps (40 FMAs per lane: the layer has no input gradient).
// Partial sums per workgroup go to f32 slabs, a second launch adds them in a fixed order (deterministic) and writes
// the two weight-gradient slots of the conv stack's arena in the layouts cum_gemm_tn would have produced.
#include "outer_common.h"

namespace cum {

constexpr int E0_H = 64;          // conv output channels
constexpr int E0_J = 128;         // 1x1 output rows (packed GLU order: per 32 rows 16 a | 16 b)
constexpr int E0_R = 32;          // rows per step
constexpr int E0_ZS = 288;        // dZ tile row stride in bytes (256 + 32: spreads the 16 rows of a fragment over the banks)
constexpr int E0_YS = 144;        // y1 tile row stride in bytes (128 + 16)
constexpr int E0_SLAB = E0_J * E0_H + E0_J + E0_H * 4 + E0_H;   // dW2 | db2 | dW1 | db1

struct Enc0BwdParams {
  const void *dz;      // [M][128] element type T, packed GLU column order
  const void *xin;     // input row buffer of the layer: [1 + 2 M + ...][8], column 0 = sample
  const float *w1;     // (64, 1, 4) f32
  const float *b1;     // (64) f32
  const void *w2p;     // [128][64] T: the 1x1 weight in the forward GEMM's packed row order
  float *slabs;        // [gridDim.x][E0_SLAB]
  int64_t M;           // output rows (clips x pitch)
  int pitch, valid;    // rows per clip, real rows per clip
};

template <typename T>
__global__ __launch_bounds__(256) void enc0_bwd_kernel(const Enc0BwdParams p) {
  static_assert(sizeof(T) == 2, "16-bit element types only");
  // one array (cdna_hip_programming.md 5.7): [2 buffers][dZ tile 32 x 288 B | y1 tile 32 x 144 B | x samples 80 f32]
  constexpr int BUF = E0_R * E0_ZS + E0_R * E0_YS + 320;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = uniform(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const T *dz = static_cast<const T *>(p.dz);
  const T *xin = static_cast<const T *>(p.xin);

  // ---- constants in registers
  // y1 rebuild: lane (row = lane & 31, hh = lane >> 5) computes channels 16 w + 8 hh + e of its row
  const int yrow = lane & 31, yh0 = 16 * w + 8 * (lane >> 5);
  float w1r[8][4], b1r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int k = 0; k < 4; ++k) w1r[e][k] = (float)(T)p.w1[(yh0 + e) * 4 + k];   // the forward multiplies rounded weights
    b1r[e] = p.b1[yh0 + e];
  }
  // B operand of dY1 = dZ W2: k = j = 32 ks + 8 g .. + 7, n = h = 16 w + r
  e0_u32x4 wB[4];
  {
    const T *w2 = static_cast<const T *>(p.w2p);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      unsigned v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = 32 * ks + 8 * g + 2 * q;
        const unsigned short lo = __builtin_bit_cast(unsigned short, w2[(int64_t)j * E0_H + 16 * w + r]);
        const unsigned short hi = __builtin_bit_cast(unsigned short, w2[(int64_t)(j + 1) * E0_H + 16 * w + r]);
        v[q] = (unsigned)lo | ((unsigned)hi << 16);
      }
      wB[ks] = e0_u32x4{v[0], v[1], v[2], v[3]};
    }
  }
  const unsigned one2 = __is_same(T, f16) ? 0x3C003C00u : 0x3F803F80u;
  const e0_u32x4 ones = e0_u32x4{one2, one2, one2, one2};

  e0_f32x4 acc2[2][4];     // dW2[j = 32 w + 16 it + 4 g + rr][h = 16 nt + r]
  e0_f32x4 accb[2];        // db2[j = 32 w + 16 it + 4 g + rr] (every column n holds the same sum)
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    accb[it] = e0_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc2[it][nt] = e0_f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float a1[4] = {0.f, 0.f, 0.f, 0.f}, ab1 = 0.f;   // dW1[h][k], db1[h] partial sums of this lane's rows

  const int64_t nsteps = (p.M + E0_R - 1) / E0_R;
  // staging registers: this thread's two 16-byte chunks of the dZ tiles of the next PF steps, and (tid < 72) one input
  // sample of each.  One step of compute is far shorter than a trip to HBM: with a single step in flight the kernel ran
  // at the memory latency (1.6 TB/s); PF steps x 2 workgroups per CU keep ~50 KB per CU on the way.
  constexpr int PF = 3;
  uint4 zc[PF][2];
  float xs[PF];
  auto fetch = [&](int64_t s, uint4 (&z)[2], float &x) {
    const int64_t m0 = s * E0_R;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = tid + 256 * it;
      int64_t m = m0 + (c >> 4);
      m = m < p.M ? m : p.M - 1;                       // clamped row; masked by the validity test below
      z[it] = *reinterpret_cast<const uint4 *>(dz + m * E0_J + (c & 15) * 8);
    }
    x = 0.f;
    if (tid < 72) {                                    // samples 2 m0 .. 2 m0 + 65 (+ a few: clamped)
      int64_t row = 1 + 2 * m0 + tid;                  // input buffer row (row 0 is the leading zero row)
      const int64_t last = 2 * p.M + 2;                // the input buffer holds at least 1 + 2 M + 2 rows
      row = row < last ? row : last;
      x = (float)xin[row * 8];
    }
  };
  int64_t s = blockIdx.x;
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int64_t sf = s + (int64_t)i * gridDim.x;
    fetch(sf < nsteps ? sf : nsteps - 1, zc[i], xs[i]);
  }
  int par = 0;
  for (; s < nsteps; s += gridDim.x, par ^= 1) {
    unsigned char *zt = lds + par * BUF, *yt = zt + E0_R * E0_ZS;
    float *xt = reinterpret_cast<float *>(yt + E0_R * E0_YS);
    const int64_t m0 = s * E0_R;
    const unsigned t0 = (unsigned)uniform((int)((unsigned)m0 % (unsigned)p.pitch));
    // ---- tiles of this step -> LDS
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = tid + 256 * it;
      // rows past the end (fetched from a clamped address) and the zero rows between clips contribute nothing --
      // db2 is a plain column sum of this tile
      const bool ok = e0_row_ok(t0, c >> 4, m0, p.M, (unsigned)p.pitch, (unsigned)p.valid);
      *reinterpret_cast<uint4 *>(zt + (c >> 4) * E0_ZS + (c & 15) * 16) = ok ? zc[0][it] : make_uint4(0, 0, 0, 0);
    }
    if (tid < 72) xt[tid] = xs[0];
    // y1 needs this step's samples: the waves that loaded them (0 and part of 1) are not the only readers
    __syncthreads();
    {
      const bool ok = e0_row_ok(t0, yrow, m0, p.M, (unsigned)p.pitch, (unsigned)p.valid);
      const float x0 = xt[2 * yrow], x1 = xt[2 * yrow + 1], x2 = xt[2 * yrow + 2], x3 = xt[2 * yrow + 3];
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = b1r[e];
        v = fmaf(w1r[e][0], x0, v);
        v = fmaf(w1r[e][1], x1, v);
        v = fmaf(w1r[e][2], x2, v);
        v = fmaf(w1r[e][3], x3, v);
        y[e] = ok && v > 0.f ? v : 0.f;
      }
      const uint4 pk = make_uint4(e0_pack2<T>(y[0], y[1]), e0_pack2<T>(y[2], y[3]), e0_pack2<T>(y[4], y[5]), e0_pack2<T>(y[6], y[7]));
      *reinterpret_cast<uint4 *>(yt + yrow * E0_YS + yh0 * 2) = pk;
    }
    __syncthreads();
    {                                                      // rotate the staging registers, fetch step s + PF
#pragma unroll
      for (int i = 0; i + 1 < PF; ++i) {
        zc[i][0] = zc[i + 1][0];
        zc[i][1] = zc[i + 1][1];
        xs[i] = xs[i + 1];
      }
      const int64_t sf = s + (int64_t)PF * gridDim.x;
      if (sf < nsteps) fetch(sf, zc[PF - 1], xs[PF - 1]);
    }

    // ---- dY1[row][h = 16 w + r] = sum_j dZ[row][j] W2[j][h]: A rows 16 mt + r, k chunk 4 ks + g
    e0_f32x4 d1[2] = {e0_f32x4{0.f, 0.f, 0.f, 0.f}, e0_f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const e0_u32x4 a = *reinterpret_cast<const e0_u32x4 *>(zt + (16 * mt + r) * E0_ZS + (4 * ks + g) * 16);
        d1[mt] = e0_mfma<T>(a, wB[ks], d1[mt]);
      }
    // result layout: d1[mt][rr] = row 16 mt + 4 g + rr, channel h = 16 w + r.  Gate by y1 > 0, then dW1 / db1.
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = 16 * mt + 4 * g + rr;
        const unsigned short yv = *reinterpret_cast<const unsigned short *>(yt + row * E0_YS + (16 * w + r) * 2);
        const float dv = (yv & 0x7FFFu) != 0 ? d1[mt][rr] : 0.f;     // y1 >= 0: non-zero <=> positive
        ab1 += dv;
        a1[0] = fmaf(dv, xt[2 * row], a1[0]);
        a1[1] = fmaf(dv, xt[2 * row + 1], a1[1]);
        a1[2] = fmaf(dv, xt[2 * row + 2], a1[2]);
        a1[3] = fmaf(dv, xt[2 * row + 3], a1[3]);
      }
    // ---- dW2[j][h] += sum_rows dZ[row][j] y1[row][h]: both operands K(row)-major -> transposing reads of the row-major
    //      tiles: lane (g, q = r >> 2, pp = r & 3) addresses row 8 g + q (+ 4), columns 16 blk + 4 pp .. + 3 and receives
    //      rows 8 g .. 8 g + 7 of column 16 blk + r
    const int q = r >> 2, pp = r & 3;
    e0_u32x4 af[2], bf[4];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const unsigned char *a0 = zt + (8 * g + q) * E0_ZS + (32 * w + 16 * it + 4 * pp) * 2;
      e0_u32x2 lo, hi;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"((unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)a0) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"((unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)a0), "n"(4 * E0_ZS) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi) : : "memory");
      af[it] = e0_u32x4{lo.x, lo.y, hi.x, hi.y};
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const unsigned char *b0 = yt + (8 * g + q) * E0_YS + (16 * nt + 4 * pp) * 2;
      e0_u32x2 lo, hi;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"((unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)b0) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"((unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)b0), "n"(4 * E0_YS) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi) : : "memory");
      bf[nt] = e0_u32x4{lo.x, lo.y, hi.x, hi.y};
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc2[it][nt] = e0_mfma<T>(af[it], bf[nt], acc2[it][nt]);
      accb[it] = e0_mfma<T>(af[it], ones, accb[it]);
    }
    // (the next step writes the OTHER buffer; its first barrier orders this step's reads before the step after it)
  }

  // ---- slabs: dW2 | db2 | dW1 | db1 of this workgroup
  float *slab = p.slabs + (int64_t)blockIdx.x * E0_SLAB;
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int j = 32 * w + 16 * it + 4 * g + rr;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) slab[j * E0_H + 16 * nt + r] = acc2[it][nt][rr];
      if (r == 0) slab[E0_J * E0_H + j] = accb[it][rr];
    }
  // dW1 / db1: the four row groups g of a channel sit in lanes r, r + 16, r + 32, r + 48
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    a1[k] += __shfl_xor(a1[k], 16, 64);
    a1[k] += __shfl_xor(a1[k], 32, 64);
  }
  ab1 += __shfl_xor(ab1, 16, 64);
  ab1 += __shfl_xor(ab1, 32, 64);
  if (g == 0) {
    const int h = 16 * w + r;
    float *s1 = slab + E0_J * E0_H + E0_J;
#pragma unroll
    for (int k = 0; k < 4; ++k) s1[h * 4 + k] = a1[k];
    s1[E0_H * 4 + h] = ab1;
  }
}

// slabs -> the two arena slots of the conv stack (layouts of cum_gemm_tn): slot_w2 = dW2 [128][64] then db2 [128];
// slot_w1 = dW1 [64][32] with element (h, 8 k) = tap k (the other columns belong to the 7 padding channels: zero) then
// db1 [64].  Two passes, both in index order: 16 groups of slabs -> 16 partial slabs (behind the workgroup slabs in the
// workspace), then those -> the slots.  (One pass with a thread per output walked 512 slabs serially: 78 us.)
constexpr int E0_RG = 16;
__global__ __launch_bounds__(256) void enc0_bwd_reduce1_kernel(const float *__restrict__ slabs, int nslabs,
                                                              float *__restrict__ part) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E0_SLAB) return;
  const int per = (nslabs + E0_RG - 1) / E0_RG;
  const int lo = blockIdx.y * per, hi = lo + per < nslabs ? lo + per : nslabs;
  float a = 0.f, b = 0.f;
  int i = lo;
  for (; i + 1 < hi; i += 2) {
    a += slabs[(int64_t)i * E0_SLAB + e];
    b += slabs[(int64_t)(i + 1) * E0_SLAB + e];
  }
  if (i < hi) a += slabs[(int64_t)i * E0_SLAB + e];
  part[(int64_t)blockIdx.y * E0_SLAB + e] = a + b;
}

__global__ __launch_bounds__(256) void enc0_bwd_reduce2_kernel(const float *__restrict__ part,
                                                              float *__restrict__ slot_w2, float *__restrict__ slot_w1) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E0_SLAB) return;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < E0_RG; ++i) v += part[(int64_t)i * E0_SLAB + e];
  constexpr int n2 = E0_J * E0_H + E0_J;
  if (e < n2) {
    slot_w2[e] = v;
  } else if (e < n2 + E0_H * 4) {
    const int h = (e - n2) >> 2, k = (e - n2) & 3;
    float *row = slot_w1 + h * 32 + 8 * k;
    row[0] = v;
#pragma unroll
    for (int c = 1; c < 8; ++c) row[c] = 0.f;          // the 7 padding channels of the tap
  } else {
    slot_w1[E0_H * 32 + (e - n2 - E0_H * 4)] = v;
  }
}

// ---------------------------------------------------------------------------------------------------- forward
// out[row][c] = a_c sigmoid(b_c), (a | b) = W2 relu(conv(x)) + b2, c < 64; the gate pre-activation b is kept for the
// backward ([M][64], output-column order), y1 never leaves the CU.  Per 32 rows: y1 rebuilt into LDS (as above), wave w
// multiplies the two packed weight tiles of channels 16 w .. 16 w + 15 (a-rows, b-rows) against it -- weights are the MFMA
// A operand, so a lane ends with the (a, b) pair of 4 consecutive channels of one row --, the GLU runs on the
// accumulators, and the results leave through an LDS tile as whole 128-byte rows (16 bytes per lane).
struct Enc0FwdParams {
  const void *xin;
  const float *w1, *b1;
  const void *w2p;     // [128][64] T
  const float *b2p;    // [128] f32, packed order
  void *out;           // row buffer of the output geometry: row 0 + [M][64] + slack rows
  void *gate;          // [M][64] T or null (inference)
  int64_t M;
  int pitch, valid;
  int64_t zero_tail;   // elements to clear behind out row M (slack rows)
};

template <typename T>
__global__ __launch_bounds__(256) void enc0_fwd_kernel(const Enc0FwdParams p) {
  static_assert(sizeof(T) == 2, "16-bit element types only");
  // [2 buffers][y1 tile 32 x 144 B | out tile 32 x 144 B | gate tile 32 x 144 B | x samples 80 f32]
  constexpr int BUF = 3 * E0_R * E0_YS + 320;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = uniform(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const T *xin = static_cast<const T *>(p.xin);
  T *out = static_cast<T *>(p.out) + E0_H;            // row 0 of the buffer is the leading zero row
  T *gate = static_cast<T *>(p.gate);
  if (blockIdx.x == 0) {                               // framing rows of the output buffer
    for (int i = tid; i < E0_H; i += 256) out[i - E0_H] = (T)0.f;
    for (int64_t i = tid; i < p.zero_tail; i += 256) out[p.M * E0_H + i] = (T)0.f;
  }
  const int yrow = lane & 31, yh0 = 16 * w + 8 * (lane >> 5);
  float w1r[8][4], b1r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int k = 0; k < 4; ++k) w1r[e][k] = (float)(T)p.w1[(yh0 + e) * 4 + k];
    b1r[e] = p.b1[yh0 + e];
  }
  // A operand: packed weight rows j = 32 w + 16 jt + r, k = h = 32 ks + 8 g .. + 7
  e0_u32x4 wA[2][2];
  float ba[4], bb[4];
  {
    const T *w2 = static_cast<const T *>(p.w2p);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        wA[jt][ks] = *reinterpret_cast<const e0_u32x4 *>(w2 + (int64_t)(32 * w + 16 * jt + r) * E0_H + 32 * ks + 8 * g);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      ba[rr] = p.b2p[32 * w + 4 * g + rr];
      bb[rr] = p.b2p[32 * w + 16 + 4 * g + rr];
    }
  }
  const int64_t nsteps = (p.M + E0_R - 1) / E0_R;
  constexpr int PF = 4;                                  // steps whose input samples are in flight (2-byte strided loads)
  float xs[PF];
  auto fetch = [&](int64_t s, float &x) {
    x = 0.f;
    if (tid < 72) {
      int64_t row = 1 + 2 * s * E0_R + tid;
      const int64_t last = 2 * p.M + 2;
      row = row < last ? row : last;
      x = (float)xin[row * 8];
    }
  };
  int64_t s = blockIdx.x;
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int64_t sf = s + (int64_t)i * gridDim.x;
    fetch(sf < nsteps ? sf : nsteps - 1, xs[i]);
  }
  int par = 0;
  for (; s < nsteps; s += gridDim.x, par ^= 1) {
    unsigned char *yt = lds + par * BUF, *ot = yt + E0_R * E0_YS, *gt = ot + E0_R * E0_YS;
    float *xt = reinterpret_cast<float *>(gt + E0_R * E0_YS);
    const int64_t m0 = s * E0_R;
    const unsigned t0 = (unsigned)uniform((int)((unsigned)m0 % (unsigned)p.pitch));
    if (tid < 72) xt[tid] = xs[0];
    __syncthreads();
    {
#pragma unroll
      for (int i = 0; i + 1 < PF; ++i) xs[i] = xs[i + 1];
      const int64_t sf = s + (int64_t)PF * gridDim.x;
      if (sf < nsteps) fetch(sf, xs[PF - 1]);
    }
    {
      const bool ok = e0_row_ok(t0, yrow, m0, p.M, (unsigned)p.pitch, (unsigned)p.valid);
      const float x0 = xt[2 * yrow], x1 = xt[2 * yrow + 1], x2 = xt[2 * yrow + 2], x3 = xt[2 * yrow + 3];
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = b1r[e];
        v = fmaf(w1r[e][0], x0, v);
        v = fmaf(w1r[e][1], x1, v);
        v = fmaf(w1r[e][2], x2, v);
        v = fmaf(w1r[e][3], x3, v);
        y[e] = ok && v > 0.f ? v : 0.f;
      }
      *reinterpret_cast<uint4 *>(yt + yrow * E0_YS + yh0 * 2) =
          make_uint4(e0_pack2<T>(y[0], y[1]), e0_pack2<T>(y[2], y[3]), e0_pack2<T>(y[4], y[5]), e0_pack2<T>(y[6], y[7]));
    }
    __syncthreads();
    // z[j][row] = sum_h W2p[j][h] y1[row][h]: B operand = y1 rows 16 mt + r, k chunk 4 ks + g (plain 16-byte reads)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      e0_f32x4 da = e0_f32x4{0.f, 0.f, 0.f, 0.f}, db = da;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const e0_u32x4 yb = *reinterpret_cast<const e0_u32x4 *>(yt + (16 * mt + r) * E0_YS + (4 * ks + g) * 16);
        da = e0_mfma<T>(wA[0][ks], yb, da);
        db = e0_mfma<T>(wA[1][ks], yb, db);
      }
      // lane: row 16 mt + r, channels 16 w + 4 g + rr
      const bool ok = e0_row_ok(t0, 16 * mt + r, m0, p.M, (unsigned)p.pitch, (unsigned)p.valid);
      float o[4], gv[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float a = da[rr] + ba[rr], b = db[rr] + bb[rr];
        o[rr] = e0_keep(ok, a * sigmoidf_(b));
        gv[rr] = ok ? b : 0.f;
      }
      const int off = (16 * mt + r) * E0_YS + (16 * w + 4 * g) * 2;
      *reinterpret_cast<uint2 *>(ot + off) = make_uint2(e0_pack2<T>(o[0], o[1]), e0_pack2<T>(o[2], o[3]));
      *reinterpret_cast<uint2 *>(gt + off) = make_uint2(e0_pack2<T>(gv[0], gv[1]), e0_pack2<T>(gv[2], gv[3]));
    }
    __syncthreads();
    {
      const int row = tid >> 3, ch = tid & 7;          // 32 rows x 8 chunks of 16 bytes = one 128-byte row per 8 lanes
      const int64_t m = m0 + row;
      if (m < p.M) {
        *reinterpret_cast<uint4 *>(out + m * E0_H + ch * 8) = *reinterpret_cast<const uint4 *>(ot + row * E0_YS + ch * 16);
        if (gate) *reinterpret_cast<uint4 *>(gate + m * E0_H + ch * 8) = *reinterpret_cast<const uint4 *>(gt + row * E0_YS + ch * 16);
      }
    }
    // (double-buffered tiles: the next step writes the other buffer, its barriers order the reuse of this one)
  }
}

}  // namespace cum

using namespace cum;

extern "C" int cum_enc0_fwd(int32_t dtype, int64_t M, int32_t pitch, int32_t valid, const void *xin, const float *w1,
                            const float *b1, const void *w2p, const float *b2p, void *out, int64_t zero_tail, void *gate,
                            void *stream) {
  CUM_REQUIRE(is16(dtype), "enc0_fwd: 16-bit element types only (f32 takes the generic path)");
  CUM_REQUIRE(M > 0 && M < 2147483647LL && pitch > 0 && valid > 0 && valid <= pitch && zero_tail >= 0, "enc0_fwd: bad geometry");
  CUM_REQUIRE(xin && w1 && b1 && w2p && b2p && out, "enc0_fwd: null tensor");
  CUM_REQUIRE(((uintptr_t)out & 15) == 0 && ((uintptr_t)gate & 15) == 0 && ((uintptr_t)w2p & 15) == 0,
              "enc0_fwd: out, gate and w2p must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  Enc0FwdParams p{};
  p.xin = xin; p.w1 = w1; p.b1 = b1; p.w2p = w2p; p.b2p = b2p; p.out = out; p.gate = gate; p.M = M; p.pitch = pitch;
  p.valid = valid; p.zero_tail = zero_tail;
  const int64_t steps = (M + E0_R - 1) / E0_R;
  const int nwg = (int)(steps < 1024 ? steps : 1024);
  if (dtype == CUM_F16)
    hipLaunchKernelGGL(enc0_fwd_kernel<f16>, dim3(nwg), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(enc0_fwd_kernel<__bf16>, dim3(nwg), dim3(256), 0, st, p);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

extern "C" int32_t cum_enc0_bwd_workgroups(int64_t M) {
  const int64_t steps = (M + E0_R - 1) / E0_R;
  return (int32_t)(steps < 1 ? 1 : (steps > 512 ? 512 : steps));    // two resident workgroups per CU (224 VGPRs)
}

extern "C" int64_t cum_enc0_bwd_workspace_elems(int64_t M) {
  return ((int64_t)cum_enc0_bwd_workgroups(M) + E0_RG) * E0_SLAB;
}

extern "C" int cum_enc0_bwd(int32_t dtype, int64_t M, int32_t pitch, int32_t valid, const void *dZ, const void *xin,
                            const float *w1, const float *b1, const void *w2p, float *slot_w2, float *slot_w1,
                            float *workspace, void *stream) {
  CUM_REQUIRE(is16(dtype), "enc0_bwd: 16-bit element types only (f32 takes the generic path)");
  CUM_REQUIRE(M > 0 && M < 2147483647LL && pitch > 0 && valid > 0 && valid <= pitch, "enc0_bwd: bad geometry");
  CUM_REQUIRE(dZ && xin && w1 && b1 && w2p && slot_w2 && slot_w1 && workspace, "enc0_bwd: null tensor");
  CUM_REQUIRE(((uintptr_t)dZ & 15) == 0, "enc0_bwd: dZ must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  Enc0BwdParams p{};
  p.dz = dZ; p.xin = xin; p.w1 = w1; p.b1 = b1; p.w2p = w2p; p.slabs = workspace; p.M = M; p.pitch = pitch; p.valid = valid;
  const int nwg = cum_enc0_bwd_workgroups(M);
  if (dtype == CUM_F16)
    hipLaunchKernelGGL(enc0_bwd_kernel<f16>, dim3(nwg), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(enc0_bwd_kernel<__bf16>, dim3(nwg), dim3(256), 0, st, p);
  CUM_CHECK_LAUNCH();
  float *part = workspace + (int64_t)nwg * E0_SLAB;
  hipLaunchKernelGGL(enc0_bwd_reduce1_kernel, dim3((E0_SLAB + 255) / 256, E0_RG), dim3(256), 0, st, workspace, nwg, part);
  hipLaunchKernelGGL(enc0_bwd_reduce2_kernel, dim3((E0_SLAB + 255) / 256), dim3(256), 0, st, part, slot_w2, slot_w1);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}
