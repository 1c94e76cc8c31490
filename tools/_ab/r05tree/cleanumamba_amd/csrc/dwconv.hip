// Causal depthwise conv1d (+SiLU) forward / backward / streaming update for gfx950.
//
// Replaces causal_conv1d_cuda.{causal_conv1d_fwd,_bwd,_update} of causal-conv1d 1.1.0,
// reached from Mamba.forward / Mamba.step (call pattern: src/network/S4/MambaS4.py:454-463;
// torch equivalent act(conv1d(x, padding=W-1)[..., :L]) at :455).  SURVEY.md Appendix A.4.
//
// Pure HBM-bandwidth kernels.  lane <-> channel (channel-contiguous rows are read as
// coalesced 256-B segments), each thread walks TC consecutive time steps with the W-1
// halo rows in registers, so every input row is fetched 1 + (W-1)/TC times.
#include "common.h"

namespace cum {

constexpr int TC = 16;    // time steps per thread
constexpr int MAXW = 4;   // kernel width supported (reference uses d_conv = 4)

struct ConvParams {
  cum_conv_shape s;
  const void *x, *dy;    // x, y, dy, dx: elements of s.io_dtype
  const float *w, *bias;
  void *y, *dx;
  float *ws;   // ws: [batch * nchunks][MAXW + 1][dim] partial dweight / dbias
  int64_t dx_sb, dx_sd, dx_sl;
  int nchunks;
};

template <int W, typename TIO>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const ConvParams p) {
  const int lane = threadIdx.x & 63;
  const int d = blockIdx.x * 64 + lane;
  const int chunk = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  const int L = p.s.len;
  const int t0 = chunk * TC;
  if (d >= p.s.dim || t0 >= L) return;
  float wk[W];
#pragma unroll
  for (int k = 0; k < W; ++k) wk[k] = p.w[d * W + k];
  const float bs = p.bias ? p.bias[d] : 0.f;
  const TIO *xp = static_cast<const TIO *>(p.x) + b * p.s.x_sb + d * p.s.x_sd;
  TIO *yp = static_cast<TIO *>(p.y) + b * p.s.y_sb + d * p.s.y_sd;
  float xv[TC + W - 1];
#pragma unroll
  for (int i = 0; i < TC + W - 1; ++i) {
    const int t = t0 - (W - 1) + i;
    const int tc = t < 0 ? 0 : (t < L ? t : L - 1);
    const float v = (float)xp[(int64_t)tc * p.s.x_sl];
    xv[i] = (t >= 0 && t < L) ? v : 0.f;
  }
#pragma unroll
  for (int i = 0; i < TC; ++i) {
    const int t = t0 + i;
    float acc = bs;
#pragma unroll
    for (int k = 0; k < W; ++k) acc = fmaf(wk[k], xv[i + k], acc);
    if (p.s.silu) acc = acc * sigmoidf_(acc);
    if (t < L) yp[(int64_t)t * p.s.y_sl] = (TIO)acc;
  }
}

// bf16, channel-contiguous rows: one lane walks TWO adjacent channels (4-byte loads / stores: a wave instruction moves
// 256 B instead of 128 B, half as many memory instructions) and the arithmetic runs on float2 pairs (v_pk_* ops).
typedef float f2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ f2 h2_to_f2(uint32_t v);
template <>
__device__ __forceinline__ f2 h2_to_f2<__bf16>(uint32_t v) {
  f2 r;
  r.x = __builtin_bit_cast(float, v << 16);
  r.y = __builtin_bit_cast(float, v & 0xffff0000u);
  return r;
}
template <>
__device__ __forceinline__ f2 h2_to_f2<f16>(uint32_t v) {
  typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
  const f16x2 h = __builtin_bit_cast(f16x2, v);
  return f2{(float)h.x, (float)h.y};
}
template <typename T>
__device__ __forceinline__ uint32_t f2_to_h2(f2 v) {
  typedef __attribute__((ext_vector_type(2))) T hx2;
  const hx2 o = {(T)v.x, (T)v.y};
  return __builtin_bit_cast(uint32_t, o);
}

template <int W, typename TH>
__global__ __launch_bounds__(256) void dwconv_fwd2_kernel(const ConvParams p) {
  const int lane = threadIdx.x & 63;
  const int d = blockIdx.x * 128 + 2 * lane;
  const int chunk = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  const int L = p.s.len;
  const int t0 = chunk * TC;
  if (d >= p.s.dim || t0 >= L) return;
  f2 wk[W];
#pragma unroll
  for (int k = 0; k < W; ++k) wk[k] = f2{p.w[d * W + k], p.w[(d + 1) * W + k]};
  const f2 bs = p.bias ? f2{p.bias[d], p.bias[d + 1]} : f2{0.f, 0.f};
  const TH *xp = static_cast<const TH *>(p.x) + b * p.s.x_sb + d;
  TH *yp = static_cast<TH *>(p.y) + b * p.s.y_sb + d;
  f2 xv[TC + W - 1];
#pragma unroll
  for (int i = 0; i < TC + W - 1; ++i) {
    const int t = t0 - (W - 1) + i;
    const int tc = t < 0 ? 0 : (t < L ? t : L - 1);
    const f2 v = h2_to_f2<TH>(*reinterpret_cast<const uint32_t *>(xp + (int64_t)tc * p.s.x_sl));
    xv[i] = (t >= 0 && t < L) ? v : f2{0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < TC; ++i) {
    const int t = t0 + i;
    f2 acc = bs;
#pragma unroll
    for (int k = 0; k < W; ++k) acc = wk[k] * xv[i + k] + acc;
    if (p.s.silu) {
      acc.x = acc.x * sigmoidf_(acc.x);
      acc.y = acc.y * sigmoidf_(acc.y);
    }
    if (t < L) *reinterpret_cast<uint32_t *>(yp + (int64_t)t * p.s.y_sl) = f2_to_h2<TH>(acc);
  }
}

template <int W, typename TIO>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const ConvParams p) {
  const int lane = threadIdx.x & 63;
  const int d = blockIdx.x * 64 + lane;
  const int chunk = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  const int L = p.s.len;
  const int t0 = chunk * TC;
  if (d >= p.s.dim || chunk >= p.nchunks) return;
  float wk[W];
#pragma unroll
  for (int k = 0; k < W; ++k) wk[k] = p.w[d * W + k];
  const float bs = p.bias ? p.bias[d] : 0.f;
  const TIO *xp = static_cast<const TIO *>(p.x) + b * p.s.x_sb + d * p.s.x_sd;
  const TIO *dyp = static_cast<const TIO *>(p.dy) + b * p.s.y_sb + d * p.s.y_sd;
  TIO *dxp = static_cast<TIO *>(p.dx) + b * p.dx_sb + d * p.dx_sd;
  // x rows t0-(W-1) .. t0+TC+W-2, dy rows t0 .. t0+TC+W-2
  float xv[TC + 2 * (W - 1)], g[TC + W - 1];
#pragma unroll
  for (int i = 0; i < TC + 2 * (W - 1); ++i) {
    const int t = t0 - (W - 1) + i;
    const int tc = t < 0 ? 0 : (t < L ? t : L - 1);
    const float v = (float)xp[(int64_t)tc * p.s.x_sl];
    xv[i] = (t >= 0 && t < L) ? v : 0.f;
  }
#pragma unroll
  for (int i = 0; i < TC + W - 1; ++i) {
    const int s = t0 + i;
    const int sc = s < L ? s : L - 1;
    const float v = (float)dyp[(int64_t)sc * p.s.y_sl];
    float gi = s < L ? v : 0.f;
    if (p.s.silu) {
      float pre = bs;
#pragma unroll
      for (int k = 0; k < W; ++k) pre = fmaf(wk[k], xv[i + k], pre);
      const float sg = sigmoidf_(pre);
      gi *= sg * (1.f + pre * (1.f - sg));
    }
    g[i] = gi;
  }
  float dwk[W], db = 0.f;
#pragma unroll
  for (int k = 0; k < W; ++k) dwk[k] = 0.f;
#pragma unroll
  for (int i = 0; i < TC; ++i) {
    const int t = t0 + i;
    // dx[t] = sum_k w[k] g[t + (W-1) - k]
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < W; ++k) acc = fmaf(wk[k], g[i + (W - 1) - k], acc);
    if (t < L) dxp[(int64_t)t * p.dx_sl] = (TIO)acc;
    // dw[k] += g[s] x[s-(W-1)+k] for s = t (each s counted by exactly one chunk)
#pragma unroll
    for (int k = 0; k < W; ++k) dwk[k] = fmaf(g[i], xv[i + k], dwk[k]);
    db += g[i];
  }
  float *ws = p.ws + ((int64_t)(b * p.nchunks + chunk) * (MAXW + 1)) * p.s.dim + d;
#pragma unroll
  for (int k = 0; k < W; ++k) ws[(int64_t)k * p.s.dim] = dwk[k];
  ws[(int64_t)MAXW * p.s.dim] = db;
}

template <int W, typename TH>
__global__ __launch_bounds__(256) void dwconv_bwd2_kernel(const ConvParams p) {
  const int lane = threadIdx.x & 63;
  const int d = blockIdx.x * 128 + 2 * lane;
  const int chunk = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  const int L = p.s.len;
  const int t0 = chunk * TC;
  if (d >= p.s.dim || chunk >= p.nchunks) return;
  f2 wk[W];
#pragma unroll
  for (int k = 0; k < W; ++k) wk[k] = f2{p.w[d * W + k], p.w[(d + 1) * W + k]};
  const f2 bs = p.bias ? f2{p.bias[d], p.bias[d + 1]} : f2{0.f, 0.f};
  const TH *xp = static_cast<const TH *>(p.x) + b * p.s.x_sb + d;
  const TH *dyp = static_cast<const TH *>(p.dy) + b * p.s.y_sb + d;
  TH *dxp = static_cast<TH *>(p.dx) + b * p.dx_sb + d;
  f2 xv[TC + 2 * (W - 1)], g[TC + W - 1];
#pragma unroll
  for (int i = 0; i < TC + 2 * (W - 1); ++i) {
    const int t = t0 - (W - 1) + i;
    const int tc = t < 0 ? 0 : (t < L ? t : L - 1);
    const f2 v = h2_to_f2<TH>(*reinterpret_cast<const uint32_t *>(xp + (int64_t)tc * p.s.x_sl));
    xv[i] = (t >= 0 && t < L) ? v : f2{0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < TC + W - 1; ++i) {
    const int s = t0 + i;
    const int sc = s < L ? s : L - 1;
    const f2 v = h2_to_f2<TH>(*reinterpret_cast<const uint32_t *>(dyp + (int64_t)sc * p.s.y_sl));
    f2 gi = s < L ? v : f2{0.f, 0.f};
    if (p.s.silu) {
      f2 pre = bs;
#pragma unroll
      for (int k = 0; k < W; ++k) pre = wk[k] * xv[i + k] + pre;
      f2 sg;
      sg.x = sigmoidf_(pre.x);
      sg.y = sigmoidf_(pre.y);
      gi *= sg * (1.f + pre * (1.f - sg));
    }
    g[i] = gi;
  }
  f2 dwk[W], db = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < W; ++k) dwk[k] = f2{0.f, 0.f};
#pragma unroll
  for (int i = 0; i < TC; ++i) {
    const int t = t0 + i;
    f2 acc = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < W; ++k) acc = wk[k] * g[i + (W - 1) - k] + acc;
    if (t < L) *reinterpret_cast<uint32_t *>(dxp + (int64_t)t * p.dx_sl) = f2_to_h2<TH>(acc);
#pragma unroll
    for (int k = 0; k < W; ++k) dwk[k] = g[i] * xv[i + k] + dwk[k];
    db += g[i];
  }
  float *ws = p.ws + ((int64_t)(b * p.nchunks + chunk) * (MAXW + 1)) * p.s.dim + d;
#pragma unroll
  for (int k = 0; k < W; ++k) *reinterpret_cast<f2 *>(ws + (int64_t)k * p.s.dim) = dwk[k];
  *reinterpret_cast<f2 *>(ws + (int64_t)MAXW * p.s.dim) = db;
}

// dweight[d][k] = sum over (batch, chunk) slabs; dbias likewise.  A workgroup owns 64 consecutive (k, d) outputs;
// its 4 waves each add every 4th slab (fixed order) and the four partial sums meet in LDS: deterministic, and
// 4 x more loads in flight than one thread per output.
__global__ __launch_bounds__(256) void dwconv_bwd_finalize_kernel(const float *ws, int nslabs, int dim, int W,
                                                                  float *dweight, float *dbias) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const bool live = i < (MAXW + 1) * dim;
  const int k = live ? i / dim : 0, d = live ? i % dim : 0;
  // eight loads in flight per thread (624 slabs at the E8 bottleneck: the two-deep form spent 25 us on load latency)
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (live && (k < W || k == MAXW)) {
    const float *base = ws + (int64_t)k * dim + d;
    const int64_t stride = (int64_t)(MAXW + 1) * dim;
    int j = sl;
    for (; j + 28 < nslabs; j += 32) {
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] += base[(j + 4 * q) * stride];
    }
    for (int q = 0; j < nslabs; j += 4, ++q) a[q & 7] += base[j * stride];
  }
  red[sl][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (sl == 0 && live) {
    const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (k == MAXW) {
      if (dbias) dbias[d] = s;
    } else if (k < W) {
      dweight[d * W + k] = s;
    }
  }
}

__global__ void dwconv_update_kernel(int batch, int dim, int W, float *__restrict__ state, const float *__restrict__ x,
                                     const float *__restrict__ w, const float *__restrict__ bias, int silu,
                                     float *__restrict__ y) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)batch * dim) return;
  const int d = i % dim;
  float *st = state + i * W;
  float acc = bias ? bias[d] : 0.f;
  for (int k = 0; k < W; ++k) {
    const float v = (k + 1 < W) ? st[k + 1] : x[i];
    st[k] = v;
    acc = fmaf(w[d * W + k], v, acc);
  }
  if (silu) acc = acc * sigmoidf_(acc);
  y[i] = acc;
}

}  // namespace cum

using namespace cum;

static int conv_check(const cum_conv_shape *s) {
  CUM_REQUIRE(s != nullptr, "conv: null shape");
  CUM_REQUIRE(s->batch >= 0 && s->dim >= 1 && s->len >= 0, "conv: bad batch/dim/len");
  CUM_REQUIRE(s->width >= 1 && s->width <= MAXW, "conv: width must be in [1, 4]");
  CUM_REQUIRE(dtype_ok(s->io_dtype), "conv: io_dtype must be CUM_F32, CUM_BF16 or CUM_F16");
  CUM_REQUIRE(s->batch <= 65535, "conv: batch > 65535");
  return CUM_OK;
}

extern "C" int cum_causal_conv1d_fwd(const cum_conv_shape *s, const void *x, const float *weight, const float *bias,
                                     void *y, void *stream) {
  if (int rc = conv_check(s)) return rc;
  CUM_REQUIRE(x && weight && y, "conv_fwd: null tensor");
  if (s->batch == 0 || s->len == 0) return CUM_OK;
  ConvParams p{};
  p.s = *s; p.x = x; p.w = weight; p.bias = bias; p.y = y;
  p.nchunks = (s->len + TC - 1) / TC;
  dim3 grid((s->dim + 63) / 64, (p.nchunks + 3) / 4, s->batch), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int io = s->io_dtype;
  const bool pair = is16(io) && s->dim % 2 == 0 && s->x_sd == 1 && s->y_sd == 1 && s->x_sl % 2 == 0 && s->y_sl % 2 == 0 &&
                    s->x_sb % 2 == 0 && s->y_sb % 2 == 0 && ((uintptr_t)x & 3) == 0 && ((uintptr_t)y & 3) == 0;
#define CUM_DW_FWD(W)                                                                                         \
  do {                                                                                                        \
    if (pair && io == CUM_BF16) hipLaunchKernelGGL((dwconv_fwd2_kernel<W, __bf16>), grid2, block, 0, st, p);  \
    else if (pair) hipLaunchKernelGGL((dwconv_fwd2_kernel<W, f16>), grid2, block, 0, st, p);                  \
    else if (io == CUM_BF16) hipLaunchKernelGGL((dwconv_fwd_kernel<W, __bf16>), grid, block, 0, st, p);       \
    else if (io == CUM_F16) hipLaunchKernelGGL((dwconv_fwd_kernel<W, f16>), grid, block, 0, st, p);           \
    else hipLaunchKernelGGL((dwconv_fwd_kernel<W, float>), grid, block, 0, st, p);                            \
  } while (0)
  dim3 grid2((s->dim / 2 + 63) / 64, grid.y, grid.z);
  switch (s->width) {
    case 1: CUM_DW_FWD(1); break;
    case 2: CUM_DW_FWD(2); break;
    case 3: CUM_DW_FWD(3); break;
    default: CUM_DW_FWD(4); break;
  }
#undef CUM_DW_FWD
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

extern "C" int64_t cum_conv_bwd_workspace_elems(int32_t batch, int32_t dim, int32_t len, int32_t width) {
  (void)width;
  const int64_t nchunks = (len + TC - 1) / TC;
  return (int64_t)batch * nchunks * (MAXW + 1) * dim;
}

extern "C" int cum_causal_conv1d_bwd(const cum_conv_shape *s, const void *x, const float *weight, const float *bias,
                                     const void *dy, void *dx, int64_t dx_sb, int64_t dx_sd, int64_t dx_sl,
                                     float *dweight, float *dbias, float *workspace, void *stream) {
  if (int rc = conv_check(s)) return rc;
  CUM_REQUIRE(x && weight && dy && dx && dweight, "conv_bwd: null tensor");
  hipStream_t st = (hipStream_t)stream;
  if (s->batch == 0 || s->len == 0) {
    (void)hipMemsetAsync(dweight, 0, sizeof(float) * (size_t)s->dim * s->width, st);
    if (dbias) (void)hipMemsetAsync(dbias, 0, sizeof(float) * s->dim, st);
    return CUM_OK;
  }
  CUM_REQUIRE(workspace, "conv_bwd: workspace required");
  ConvParams p{};
  p.s = *s; p.x = x; p.w = weight; p.bias = bias; p.dy = dy; p.dx = dx; p.ws = workspace;
  p.dx_sb = dx_sb; p.dx_sd = dx_sd; p.dx_sl = dx_sl;
  p.nchunks = (s->len + TC - 1) / TC;
  dim3 grid((s->dim + 63) / 64, (p.nchunks + 3) / 4, s->batch), block(256);
  const int io = s->io_dtype;
  // two channels per lane when every row of x, dy and dx starts 4-byte aligned with unit channel stride
  const bool pair = is16(io) && s->dim % 2 == 0 && s->x_sd == 1 && s->y_sd == 1 && dx_sd == 1 && s->x_sl % 2 == 0 &&
                    s->y_sl % 2 == 0 && dx_sl % 2 == 0 && s->x_sb % 2 == 0 && s->y_sb % 2 == 0 && dx_sb % 2 == 0 &&
                    ((uintptr_t)x & 3) == 0 && ((uintptr_t)dy & 3) == 0 && ((uintptr_t)dx & 3) == 0;
#define CUM_DW_BWD(W)                                                                                         \
  do {                                                                                                        \
    if (pair && io == CUM_BF16) hipLaunchKernelGGL((dwconv_bwd2_kernel<W, __bf16>), grid2, block, 0, st, p);  \
    else if (pair) hipLaunchKernelGGL((dwconv_bwd2_kernel<W, f16>), grid2, block, 0, st, p);                  \
    else if (io == CUM_BF16) hipLaunchKernelGGL((dwconv_bwd_kernel<W, __bf16>), grid, block, 0, st, p);       \
    else if (io == CUM_F16) hipLaunchKernelGGL((dwconv_bwd_kernel<W, f16>), grid, block, 0, st, p);           \
    else hipLaunchKernelGGL((dwconv_bwd_kernel<W, float>), grid, block, 0, st, p);                            \
  } while (0)
  dim3 grid2((s->dim / 2 + 63) / 64, (p.nchunks + 3) / 4, s->batch);
  switch (s->width) {
    case 1: CUM_DW_BWD(1); break;
    case 2: CUM_DW_BWD(2); break;
    case 3: CUM_DW_BWD(3); break;
    default: CUM_DW_BWD(4); break;
  }
#undef CUM_DW_BWD
  CUM_CHECK_LAUNCH();
  const int total = (MAXW + 1) * s->dim;
  hipLaunchKernelGGL(dwconv_bwd_finalize_kernel, dim3((total + 63) / 64), dim3(256), 0, st, workspace,
                     s->batch * p.nchunks, s->dim, s->width, dweight, dbias);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

extern "C" int cum_causal_conv1d_update(int32_t batch, int32_t dim, int32_t width, float *conv_state, const float *x,
                                        const float *weight, const float *bias, int32_t silu, float *y,
                                        void *stream) {
  CUM_REQUIRE(batch >= 0 && dim >= 1 && width >= 1, "conv_update: bad sizes");
  CUM_REQUIRE(conv_state && x && weight && y, "conv_update: null tensor");
  if (batch == 0) return CUM_OK;
  const int64_t total = (int64_t)batch * dim;
  hipLaunchKernelGGL(dwconv_update_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     batch, dim, width, conv_state, x, weight, bias, silu, y);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}
