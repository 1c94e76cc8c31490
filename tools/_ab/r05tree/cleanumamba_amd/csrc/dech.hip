// One decoder layer of width 128 as ONE forward kernel for gfx950:
//   Conv1d(128 -> 256, 1x1) + GLU + ConvTranspose1d(128 -> 64, k 4, s 2) + ReLU (+ the skip connection of the next layer)
//   (src/network/CleanUMamba.py:121-130, 313-316 at channels_H = 64: the second-to-last decoder layer of E6 / E8)
// on row buffers (network/convstack.py Geo), 16-bit element types.  It replaces the 1x1 GEMM launch (EPI_GLU, gate-only
// save) + the transposed-conv GEMM launch (EPI_RELU with residual and sign nibbles) and writes exactly what those two
// write -- the GLU output g (the transposed conv's weight-gradient operand), its gate pre-activation, the layer output,
// the sign nibbles of the ReLU -- so the backward is unchanged.  With g == NULL (no backward to come) only the output leaves.
//
// Design: that of ench.hip.  Both weight matrices live in the registers of a persistent 8-wave workgroup:
// wave w owns GLU rows 32 w .. 32 w + 31 of the 1x1 (16 (a, b) pairs, 2 x 4 MFMA A-operand fragments) and rows 16 w ..
// 16 w + 15 of the transposed conv seen as a GEMM with N = (output-row parity, 64 channels), K = (row m - 1 | row m, 128
// channels) (8 fragments).  A tile is 128 GEMM rows = 256 output rows; it needs g of buffer rows m0 .. m0 + 128, so the
// GLU is computed for 129 (of 144) input rows -- row m0 is the halo the previous tile also computes -- into a padded LDS
// image, which is the B operand of the second GEMM and leaves for HBM as whole rows; bias + ReLU + padding mask + skip on
// the accumulators, 8-byte stores, sign nibbles through a small LDS image.  Three barriers per tile.
#include "outer_common.h"

namespace cum {

constexpr int DH_R = 128;              // GEMM rows per tile (two output rows each)
constexpr int DH_UR = 144;             // input rows per tile in LDS (9 groups of 16; 129 are needed)
constexpr int DH_S = 288;              // image row stride in bytes (128 channels + 32)
constexpr int DH_UCH = DH_UR * 16;     // 16-byte chunks of the u image
constexpr int DH_PF = (DH_UCH + 511) / 512;

struct DecHParams {
  const void *u, *w1p, *wtp, *skip;
  const float *b1p, *btp;
  void *g, *gate, *out;
  unsigned char *bits;
  int64_t M, u_rows, g_tail, out_tail;
  int pitch, valid;                    // of the input rows: data row d is real iff (d mod pitch) < valid
};

template <typename T>
__global__ __launch_bounds__(512) void dech_fwd_kernel(const DecHParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char uimg[DH_UR * DH_S];
  __shared__ __attribute__((aligned(16))) unsigned char gimg[DH_UR * DH_S];
  __shared__ __attribute__((aligned(16))) unsigned char bimg[DH_R * 32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = uniform(tid >> 6);
  const int g4 = lane >> 4, r = lane & 15;
  const T *u = static_cast<const T *>(p.u);             // buffer row 0 (the leading zero row)
  const T *skip = static_cast<const T *>(p.skip);       // row 1 of the skip buffer (output geometry), may be null
  T *gb = static_cast<T *>(p.g), *gate = static_cast<T *>(p.gate), *out = static_cast<T *>(p.out);

  if (blockIdx.x == 0) {                 // framing rows of the two row buffers (as the GEMM launches do)
    for (int64_t i = tid; i < 128; i += 512)
      if (gb) gb[i] = (T)0.f;                            // g: buffer row 0
    if (gb)
      for (int64_t i = tid; i < p.g_tail; i += 512) gb[(p.M + 1) * 128 + i] = (T)0.f;
    for (int64_t i = tid; i < 64; i += 512) out[-1 - i] = (T)0.f;        // out points at row 1; rows are 64 wide
    for (int64_t i = tid; i < p.out_tail; i += 512) out[p.M * 128 + i] = (T)0.f;
  }

  // ---- weights and biases of this wave, once
  e0_u32x4 wA1a[4], wA1b[4], wA2[8];
  {
    const T *w1 = static_cast<const T *>(p.w1p) + (int64_t)(32 * w + r) * 128 + 8 * g4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      wA1a[s] = *reinterpret_cast<const e0_u32x4 *>(w1 + 32 * s);
      wA1b[s] = *reinterpret_cast<const e0_u32x4 *>(w1 + 16 * 128 + 32 * s);
    }
    const T *wt = static_cast<const T *>(p.wtp) + (int64_t)(16 * w + r) * 256 + 8 * g4;
#pragma unroll
    for (int s = 0; s < 8; ++s) wA2[s] = *reinterpret_cast<const e0_u32x4 *>(wt + 32 * s);
  }
  const float4 ba = *reinterpret_cast<const float4 *>(p.b1p + 32 * w + 4 * g4);
  const float4 bb = *reinterpret_cast<const float4 *>(p.b1p + 32 * w + 16 + 4 * g4);
  const float4 bt = *reinterpret_cast<const float4 *>(p.btp + 16 * w + 4 * g4);

  const int64_t ntiles = (p.M + DH_R - 1) / DH_R;
  e0_u32x4 pf[DH_PF];
  auto fetch = [&](int64_t tile) {
    const int64_t r0 = tile * DH_R;
#pragma unroll
    for (int it = 0; it < DH_PF; ++it) {
      const int idx = it * 512 + tid;
      if (idx < DH_UCH) {
        int64_t row = r0 + (idx >> 4);
        row = row < p.u_rows ? row : p.u_rows - 1;
        pf[it] = *reinterpret_cast<const e0_u32x4 *>(u + row * 128 + (idx & 15) * 8);
      }
    }
  };
  int64_t tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    const int64_t m0 = tile * DH_R;
#pragma unroll
    for (int it = 0; it < DH_PF; ++it) {
      const int idx = it * 512 + tid;
      if (idx < DH_UCH) *reinterpret_cast<e0_u32x4 *>(uimg + (idx >> 4) * DH_S + (idx & 15) * 16) = pf[it];
    }
    __syncthreads();                                   // the image is complete; the previous tile's readers are done
    if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
    // the skip values of this lane's outputs, requested before the first GEMM
    uint2 sk[8];
    if (skip) {
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        int64_t m = m0 + 16 * n + r;
        m = m < p.M ? m : p.M - 1;
        sk[n] = *reinterpret_cast<const uint2 *>(skip + m * 128 + 16 * w + 4 * g4);
      }
    }
    const unsigned t0 = (unsigned)((uint64_t)m0 % (unsigned)p.pitch);      // position of buffer row m0 + 1's data row ...
    // (buffer row b carries data row b - 1; GEMM row m pairs buffer rows m, m + 1 = data rows m - 1, m.  With tt = m mod
    //  pitch: data row m - 1 is real iff 1 <= tt <= valid, GEMM row m has real outputs iff tt <= valid.)
    // ---- GLU tile: g[buffer row m0 + i][16 w + 4 g4 + j] for the 9 groups of 16 rows (i = 0: halo, i > 128: unused)
#pragma unroll
    for (int n = 0; n < 9; ++n) {
      const int i = 16 * n + r;
      e0_f32x4 aa = e0_f32x4{0.f, 0.f, 0.f, 0.f}, ab = aa;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const e0_u32x4 bf = *reinterpret_cast<const e0_u32x4 *>(uimg + i * DH_S + (4 * s + g4) * 16);
        aa = e0_mfma<T>(wA1a[s], bf, aa);
        ab = e0_mfma<T>(wA1b[s], bf, ab);
      }
      unsigned tt = t0 + (unsigned)i;
      while (tt >= (unsigned)p.pitch) tt -= (unsigned)p.pitch;
      const int64_t b = m0 + i;                        // buffer row
      const bool real = b >= 1 && b <= p.M && tt >= 1u && tt <= (unsigned)p.valid;
      const float a[4] = {aa[0] + ba.x, aa[1] + ba.y, aa[2] + ba.z, aa[3] + ba.w};
      const float bg[4] = {ab[0] + bb.x, ab[1] + bb.y, ab[2] + bb.z, ab[3] + bb.w};
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = e0_keep(real, a[j] * sigmoidf_(bg[j]));
      *reinterpret_cast<uint2 *>(gimg + i * DH_S + (16 * w + 4 * g4) * 2) = make_uint2(e0_pack2<T>(o[0], o[1]), e0_pack2<T>(o[2], o[3]));
      if (gate && i >= 1 && i <= DH_R && b <= p.M)     // gate of data row b - 1 (owned rows only; stored unmasked)
        *reinterpret_cast<uint2 *>(gate + (b - 1) * 128 + 16 * w + 4 * g4) = make_uint2(e0_pack2<T>(bg[0], bg[1]), e0_pack2<T>(bg[2], bg[3]));
    }
    __syncthreads();                                   // the GLU tile is complete
    // ---- the tile's own g rows (buffer rows m0 + 1 .. m0 + 128) out, whole rows
    if (gb) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = it * 512 + tid, i = 1 + (idx >> 4), ch = idx & 15;
        if (m0 + i <= p.M)
          *reinterpret_cast<e0_u32x4 *>(gb + (m0 + i) * 128 + ch * 8) = *reinterpret_cast<const e0_u32x4 *>(gimg + i * DH_S + ch * 16);
      }
    }
    // ---- transposed conv as a GEMM + bias + ReLU + mask (+ skip): out[2 m + q][co], n = 64 q + co = 16 w + 4 g4 + j
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const int lr = 16 * n + r;
      e0_f32x4 acc = e0_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const e0_u32x4 bf = *reinterpret_cast<const e0_u32x4 *>(gimg + (lr + (s >> 2)) * DH_S + (4 * (s & 3) + g4) * 16);
        acc = e0_mfma<T>(wA2[s], bf, acc);
      }
      unsigned tt = t0 + (unsigned)lr;
      while (tt >= (unsigned)p.pitch) tt -= (unsigned)p.pitch;
      const int64_t m = m0 + lr;
      const bool real = m < p.M && tt <= (unsigned)p.valid;
      float v[4] = {acc[0] + bt.x, acc[1] + bt.y, acc[2] + bt.z, acc[3] + bt.w};
      unsigned nib = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = real ? fmaxf(v[j], 0.f) : 0.f;
        nib |= (v[j] > 0.f ? 1u : 0u) << j;
      }
      bimg[lr * 32 + 4 * w + g4] = (unsigned char)nib;
      if (skip) {
        typedef T V2 __attribute__((ext_vector_type(2)));
        const V2 s01 = __builtin_bit_cast(V2, sk[n].x), s23 = __builtin_bit_cast(V2, sk[n].y);
        v[0] = real ? v[0] + (float)s01[0] : 0.f;
        v[1] = real ? v[1] + (float)s01[1] : 0.f;
        v[2] = real ? v[2] + (float)s23[0] : 0.f;
        v[3] = real ? v[3] + (float)s23[1] : 0.f;
      }
      if (m < p.M)
        *reinterpret_cast<uint2 *>(out + m * 128 + 16 * w + 4 * g4) = make_uint2(e0_pack2<T>(v[0], v[1]), e0_pack2<T>(v[2], v[3]));
    }
    if (p.bits) {
      __syncthreads();                                 // the nibble image is complete
      if (tid < 256) {
        const int row = tid >> 1, hf = tid & 1;
        if (m0 + row < p.M)
          *reinterpret_cast<e0_u32x4 *>(p.bits + (m0 + row) * 32 + hf * 16) = *reinterpret_cast<const e0_u32x4 *>(bimg + row * 32 + hf * 16);
      }
    }
    // (next tile: the u image is rewritten by waves past the second barrier = after every wave's last read of it; the g
    //  image after the next first barrier = after every wave's reads above; the nibble image after the next second one)
  }
}

}  // namespace cum

using namespace cum;

extern "C" int cum_dech_fwd(int32_t dtype, int64_t M, int32_t pitch, int32_t valid, const void *u, int64_t u_rows,
                            const void *w1p, const float *b1p, const void *wtp, const float *btp, const void *skip, void *g,
                            int64_t g_tail, void *gate, void *out, int64_t out_tail, void *bits, void *stream) {
  CUM_REQUIRE(is16(dtype), "dech_fwd: 16-bit element types only");
  CUM_REQUIRE(M >= 0 && pitch >= 3 && valid >= 0 && valid < pitch && u_rows >= 1 && g_tail >= 0 && out_tail >= 0,
              "dech_fwd: bad sizes");
  CUM_REQUIRE(u && w1p && b1p && wtp && btp && out, "dech_fwd: null tensor");
  CUM_REQUIRE((((uintptr_t)u | (uintptr_t)w1p | (uintptr_t)wtp | (uintptr_t)b1p | (uintptr_t)btp | (uintptr_t)skip |
                (uintptr_t)g | (uintptr_t)gate | (uintptr_t)out | (uintptr_t)bits) & 15) == 0,
              "dech_fwd: pointers must be 16-byte aligned");
  CUM_REQUIRE((gate == nullptr) == (g == nullptr), "dech_fwd: g and its gate are stored together");
  if (M == 0) return CUM_OK;
  DecHParams p{};
  p.u = u; p.w1p = w1p; p.wtp = wtp; p.skip = skip; p.b1p = b1p; p.btp = btp;
  p.g = g; p.gate = gate; p.out = out; p.bits = static_cast<unsigned char *>(bits);
  p.M = M; p.u_rows = u_rows; p.g_tail = g_tail; p.out_tail = out_tail; p.pitch = pitch; p.valid = valid;
  const int64_t ntiles = (M + DH_R - 1) / DH_R;
  const int grid = (int)(ntiles < 256 ? ntiles : 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CUM_F16)
    hipLaunchKernelGGL(dech_fwd_kernel<f16>, dim3(grid), dim3(512), 0, st, p);
  else
    hipLaunchKernelGGL(dech_fwd_kernel<__bf16>, dim3(grid), dim3(512), 0, st, p);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}
