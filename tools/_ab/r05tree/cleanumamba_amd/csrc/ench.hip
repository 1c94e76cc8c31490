// One encoder layer of width 128 as ONE forward kernel for gfx950:
//   Conv1d(64 -> 128, k 4, s 2) + ReLU + Conv1d(128 -> 256, 1x1) + GLU      (src/network/CleanUMamba.py:108-113 at
//   channels_H = 64: the second encoder layer of E6 / E8)
// on row buffers (network/convstack.py Geo), 16-bit element types.  What it replaces: the conv GEMM launch (EPI_RELU with
// sign nibbles) + the 1x1 GEMM launch (EPI_GLU, gate-only save), i.e. one write and one read of the hidden activation
// y1 (M x 128) and a launch; it writes exactly what those two write -- y1 (the 1x1's weight-gradient operand), the sign
// nibbles of the ReLU, the layer output, the gate pre-activation -- so the backward is unchanged.  With gate == NULL and
// y1 == NULL (no backward to come) only the output leaves the chip.
//
// Design: both weight matrices live in the REGISTERS of a persistent 8-wave workgroup -- wave w owns hidden
// channels 16 w .. 16 w + 15 (W1: 8 MFMA A-operand fragments) and GLU rows 32 w .. 32 w + 31 = 16 (a, b) pairs (W2: 2 x 4
// fragments), loaded once.  A tile is 128 output rows: its 258 input rows (64 channels, contiguous in the row buffer) are
// one padded image in LDS (register-staged one tile ahead), every wave multiplies its W1 slice against the whole image
// (weights = A operand, rows = B operand: a lane ends with four consecutive channels of one row), bias + ReLU + padding
// mask, hidden tile -> LDS as the B operand of the second GEMM and, as whole 256-byte rows, out to HBM; second GEMM, GLU
// on the accumulators (a and b of a channel sit in the same lane), 8-byte stores.  Row strides of both images are 16 bytes
// past a multiple of 128 so that the 16-lane groups of ds_read_b128 fall on distinct banks.  Two barriers per tile.
#include "outer_common.h"

namespace cum {

constexpr int EH_R = 128;              // output rows per tile
constexpr int EH_XR = 2 * EH_R + 2;    // input rows per tile
constexpr int EH_XS = 144;             // x image row stride in bytes (64 channels + 16)
constexpr int EH_YS = 288;             // y1 image row stride in bytes (128 channels + 32)
constexpr int EH_XCH = EH_XR * 8;      // 16-byte chunks of the x image
constexpr int EH_PF = (EH_XCH + 511) / 512;

struct EncHParams {
  const void *x, *w1p, *w2p;
  const float *b1p, *b2p;
  void *y1, *y, *gate;
  unsigned char *bits;
  int64_t M, x_rows, y1_tail, y_tail;
  int pitch, valid;
};

template <typename T>
__global__ __launch_bounds__(512) void ench_fwd_kernel(const EncHParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char ximg[EH_XR * EH_XS];
  __shared__ __attribute__((aligned(16))) unsigned char yimg[EH_R * EH_YS];
  __shared__ __attribute__((aligned(16))) unsigned char bimg[EH_R * 32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = uniform(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const T *x = static_cast<const T *>(p.x);
  T *y1 = static_cast<T *>(p.y1), *y = static_cast<T *>(p.y), *gate = static_cast<T *>(p.gate);

  if (blockIdx.x == 0) {                 // framing rows of the two row buffers (as the GEMM launches do)
    for (int64_t i = tid; i < 128; i += 512) {
      if (y1) y1[-1 - i] = (T)0.f;
      y[-1 - i] = (T)0.f;
    }
    if (y1)
      for (int64_t i = tid; i < p.y1_tail; i += 512) y1[p.M * 128 + i] = (T)0.f;
    for (int64_t i = tid; i < p.y_tail; i += 512) y[p.M * 128 + i] = (T)0.f;
  }

  // ---- weights and biases of this wave, once
  e0_u32x4 wA1[8], wA2a[4], wA2b[4];
  {
    const T *w1 = static_cast<const T *>(p.w1p) + (int64_t)(16 * w + r) * 256 + 8 * g;
#pragma unroll
    for (int s = 0; s < 8; ++s) wA1[s] = *reinterpret_cast<const e0_u32x4 *>(w1 + 32 * s);
    const T *w2 = static_cast<const T *>(p.w2p) + (int64_t)(32 * w + r) * 128 + 8 * g;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      wA2a[s] = *reinterpret_cast<const e0_u32x4 *>(w2 + 32 * s);
      wA2b[s] = *reinterpret_cast<const e0_u32x4 *>(w2 + 16 * 128 + 32 * s);
    }
  }
  const float4 b1 = *reinterpret_cast<const float4 *>(p.b1p + 16 * w + 4 * g);
  const float4 ba = *reinterpret_cast<const float4 *>(p.b2p + 32 * w + 4 * g);
  const float4 bb = *reinterpret_cast<const float4 *>(p.b2p + 32 * w + 16 + 4 * g);

  const int64_t ntiles = (p.M + EH_R - 1) / EH_R;
  e0_u32x4 pf[EH_PF];
  auto fetch = [&](int64_t tile) {
    const int64_t r0 = 2 * tile * EH_R;
#pragma unroll
    for (int it = 0; it < EH_PF; ++it) {
      const int idx = it * 512 + tid;
      if (idx < EH_XCH) {
        int64_t row = r0 + (idx >> 3);
        row = row < p.x_rows ? row : p.x_rows - 1;
        pf[it] = *reinterpret_cast<const e0_u32x4 *>(x + row * 64 + (idx & 7) * 8);
      }
    }
  };
  int64_t tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    const int64_t m0 = tile * EH_R;
#pragma unroll
    for (int it = 0; it < EH_PF; ++it) {
      const int idx = it * 512 + tid;
      if (idx < EH_XCH) *reinterpret_cast<e0_u32x4 *>(ximg + (idx >> 3) * EH_XS + (idx & 7) * 16) = pf[it];
    }
    __syncthreads();                                   // the image is complete; the previous tile's readers are done
    if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
    const unsigned t0 = (unsigned)((uint64_t)m0 % (unsigned)p.pitch);
    // ---- hidden tile: y1[row][16 w + 4 g + j] for the 8 groups of 16 rows
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const int lr = 16 * n + r;
      e0_f32x4 acc = e0_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const e0_u32x4 bf = *reinterpret_cast<const e0_u32x4 *>(ximg + (2 * lr + (s >> 1)) * EH_XS + (4 * (s & 1) + g) * 16);
        acc = e0_mfma<T>(wA1[s], bf, acc);
      }
      unsigned tt = t0 + (unsigned)lr;                 // (t0 < pitch, lr < 128: a loop of conditional subtractions is
      while (tt >= (unsigned)p.pitch) tt -= (unsigned)p.pitch;   //  at most 128 / pitch + 1 long; pitch >= 3)
      const bool real = m0 + lr < p.M && tt < (unsigned)p.valid;
      float v[4] = {acc[0] + b1.x, acc[1] + b1.y, acc[2] + b1.z, acc[3] + b1.w};
      unsigned nib = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = real ? fmaxf(v[j], 0.f) : 0.f;
        nib |= (v[j] > 0.f ? 1u : 0u) << j;
      }
      *reinterpret_cast<uint2 *>(yimg + lr * EH_YS + (16 * w + 4 * g) * 2) = make_uint2(e0_pack2<T>(v[0], v[1]), e0_pack2<T>(v[2], v[3]));
      bimg[lr * 32 + 4 * w + g] = (unsigned char)nib;
    }
    __syncthreads();                                   // the hidden tile is complete
    // ---- hidden tile and its sign nibbles out, whole rows
    if (y1) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = it * 512 + tid, row = idx >> 4, ch = idx & 15;
        if (m0 + row < p.M)
          *reinterpret_cast<e0_u32x4 *>(y1 + (m0 + row) * 128 + ch * 8) = *reinterpret_cast<const e0_u32x4 *>(yimg + row * EH_YS + ch * 16);
      }
      if (p.bits && tid < 256) {
        const int row = tid >> 1, hf = tid & 1;
        if (m0 + row < p.M)
          *reinterpret_cast<e0_u32x4 *>(p.bits + (m0 + row) * 32 + hf * 16) = *reinterpret_cast<const e0_u32x4 *>(bimg + row * 32 + hf * 16);
      }
    }
    // ---- 1x1 + GLU: out[row][16 w + 4 g + j]
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const int lr = 16 * n + r;
      e0_f32x4 aa = e0_f32x4{0.f, 0.f, 0.f, 0.f}, ab = aa;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const e0_u32x4 bf = *reinterpret_cast<const e0_u32x4 *>(yimg + lr * EH_YS + (4 * s + g) * 16);
        aa = e0_mfma<T>(wA2a[s], bf, aa);
        ab = e0_mfma<T>(wA2b[s], bf, ab);
      }
      unsigned tt = t0 + (unsigned)lr;
      while (tt >= (unsigned)p.pitch) tt -= (unsigned)p.pitch;
      const bool real = tt < (unsigned)p.valid;
      if (m0 + lr < p.M) {
        const float a[4] = {aa[0] + ba.x, aa[1] + ba.y, aa[2] + ba.z, aa[3] + ba.w};
        const float b[4] = {ab[0] + bb.x, ab[1] + bb.y, ab[2] + bb.z, ab[3] + bb.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = e0_keep(real, a[j] * sigmoidf_(b[j]));
        const int64_t off = (m0 + lr) * 128 + 16 * w + 4 * g;
        *reinterpret_cast<uint2 *>(y + off) = make_uint2(e0_pack2<T>(o[0], o[1]), e0_pack2<T>(o[2], o[3]));
        if (gate) *reinterpret_cast<uint2 *>(gate + off) = make_uint2(e0_pack2<T>(b[0], b[1]), e0_pack2<T>(b[2], b[3]));
      }
    }
    // (the next tile's image is written by waves that have passed the second barrier, i.e. after every wave's last read
    //  of this one; its hidden tile is written after the next first barrier, i.e. after every wave's reads above)
  }
}

}  // namespace cum

using namespace cum;

extern "C" int cum_ench_fwd(int32_t dtype, int64_t M, int32_t pitch, int32_t valid, const void *xin, int64_t x_rows,
                            const void *w1p, const float *b1p, const void *w2p, const float *b2p, void *y1, int64_t y1_tail,
                            void *bits, void *out, int64_t out_tail, void *gate, void *stream) {
  CUM_REQUIRE(is16(dtype), "ench_fwd: 16-bit element types only");
  CUM_REQUIRE(M >= 0 && pitch >= 3 && valid >= 0 && valid <= pitch && x_rows >= 1 && y1_tail >= 0 && out_tail >= 0,
              "ench_fwd: bad sizes");
  CUM_REQUIRE(xin && w1p && b1p && w2p && b2p && out, "ench_fwd: null tensor");
  CUM_REQUIRE((((uintptr_t)xin | (uintptr_t)w1p | (uintptr_t)w2p | (uintptr_t)b1p | (uintptr_t)b2p | (uintptr_t)y1 |
                (uintptr_t)bits | (uintptr_t)out | (uintptr_t)gate) & 15) == 0, "ench_fwd: pointers must be 16-byte aligned");
  CUM_REQUIRE(!bits || y1, "ench_fwd: sign nibbles are written with the hidden activation");
  if (M == 0) return CUM_OK;
  EncHParams p{};
  p.x = xin; p.w1p = w1p; p.w2p = w2p; p.b1p = b1p; p.b2p = b2p;
  p.y1 = y1; p.y = out; p.gate = gate; p.bits = static_cast<unsigned char *>(bits);
  p.M = M; p.x_rows = x_rows; p.y1_tail = y1_tail; p.y_tail = out_tail; p.pitch = pitch; p.valid = valid;
  const int64_t ntiles = (M + EH_R - 1) / EH_R;
  const int grid = (int)(ntiles < 256 ? ntiles : 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CUM_F16)
    hipLaunchKernelGGL(ench_fwd_kernel<f16>, dim3(grid), dim3(512), 0, st, p);
  else
    hipLaunchKernelGGL(ench_fwd_kernel<__bf16>, dim3(grid), dim3(512), 0, st, p);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}
