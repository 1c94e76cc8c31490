// Residual add + LayerNorm of the Mamba blocks, forward and backward, one pass each.
//
// Reference: mamba-ssm Block.forward as CleanUMamba runs it (fused_add_norm=False, residual_in_fp32=True;
// src/network/CleanUMamba.py:156-189, 288-294):  residual = hidden + residual (fp32);  hidden = LayerNorm(residual).
// On PyTorch that is an add, a cast and the LayerNorm kernels forward, and five kernels backward, each a pass over
// (B, L, d_model).  Here one wave owns a row: the row stays in registers between the add, the statistics and the
// normalisation; the backward produces the input gradient (= gradient of `hidden` and of the incoming residual) in
// one pass and accumulates the weight / bias gradients per workgroup (f32 slabs + fixed-order finalize: no atomics).
#include "common.h"

namespace cum {

constexpr int LN_MAXCH = 4;     // 8-element chunks per lane: d_model <= 64 * 8 * 4 = 2048

template <typename T>
__device__ __forceinline__ void ld8(const T *p, float (&v)[8]);
template <>
__device__ __forceinline__ void ld8<float>(const float *p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void ld8<__bf16>(const __bf16 *p, float (&v)[8]) {
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
  const bf16x8 t = *reinterpret_cast<const bf16x8 *>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
template <>
__device__ __forceinline__ void ld8<f16>(const f16 *p, float (&v)[8]) {
  typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
  const f16x8 t = *reinterpret_cast<const f16x8 *>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
template <typename T>
__device__ __forceinline__ void st8(T *p, const float (&v)[8]);
template <>
__device__ __forceinline__ void st8<float>(float *p, const float (&v)[8]) {
  *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4 *>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <>
__device__ __forceinline__ void st8<__bf16>(__bf16 *p, const float (&v)[8]) {
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
  bf16x8 t;
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = (__bf16)v[i];
  *reinterpret_cast<bf16x8 *>(p) = t;
}

template <>
__device__ __forceinline__ void st8<f16>(f16 *p, const float (&v)[8]) {
  typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
  f16x8 t;
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = (f16)v[i];
  *reinterpret_cast<f16x8 *>(p) = t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct LnParams {
  const void *x;            // hidden (batch, len, dim), element strides x_sb, x_sl, 1
  const float *res;         // incoming residual, contiguous f32, or null
  const float *w, *b;       // LayerNorm weight / bias (bias may be null)
  float *res_out;           // x + res, contiguous f32
  void *y;                  // normalised output, contiguous
  float *mean, *rstd;       // per row
  int64_t x_sb, x_sl;
  int64_t rows;             // batch * len
  int len, dim, nch;        // nch = dim / 8
  float eps;
};

// grid: rows / 4 workgroups of 4 waves; wave = one row
// MAXCH: 8-element chunks per lane -- 1 for d_model <= 512 (every shipped configuration: a quarter of the registers of the
// general form, which keeps rows of up to 2048 in registers), else LN_MAXCH
template <typename TX, typename TY, int MAXCH>
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const LnParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const TX *x = static_cast<const TX *>(p.x) + (row / p.len) * p.x_sb + (row % p.len) * p.x_sl;
  float v[MAXCH][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int c = lane + 64 * i;
    if (c < p.nch) {
      ld8<TX>(x + 8 * c, v[i]);
      if (p.res) {
        float r[8];
        ld8<float>(p.res + row * p.dim + 8 * c, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] += r[j];
      }
      st8<float>(p.res_out + row * p.dim + 8 * c, v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[i][j];
    }
  }
  const float mean = wave_sum(s) / p.dim;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i)
    if (lane + 64 * i < p.nch) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = v[i][j] - mean;
        q += d * d;
      }
    }
  const float rstd = rsqrtf(wave_sum(q) / p.dim + p.eps);
  if (lane == 0) {
    p.mean[row] = mean;
    p.rstd[row] = rstd;
  }
  TY *y = static_cast<TY *>(p.y) + row * p.dim;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int c = lane + 64 * i;
    if (c < p.nch) {
      float w[8], b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, o[8];
      ld8<float>(p.w + 8 * c, w);
      if (p.b) ld8<float>(p.b + 8 * c, b);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * w[j] + b[j];
      st8<TY>(y + 8 * c, o);
    }
  }
}

struct LnBwdParams {
  const void *dy;           // gradient of the normalised output, contiguous (rows, dim)
  const float *dres;        // gradient of res_out arriving from the next block, contiguous f32, or null
  const float *xr;          // res_out saved by the forward
  const float *mean, *rstd, *w;
  float *dx32;              // input gradient, f32 (gradient of the incoming residual), or null
  void *dxh;                // the same values in the dtype of `hidden`, or null
  float *slab;              // [gridDim.x][2][dim] partial dweight | dbias
  int64_t rows;
  int dim, nch;
};

// persistent workgroups: wave w of workgroup g walks rows g*4 + w, + 4*gridDim.x, ...
template <typename TY, typename TH, int MAXCH>
__global__ __launch_bounds__(256) void add_layernorm_bwd_kernel(const LnBwdParams p) {
  __shared__ float red[4][2][MAXCH * 64 * 8];           // per wave: dweight | dbias partials (first `dim` used): <= 64 KB
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float aw[MAXCH][8], ab[MAXCH][8];
#pragma unroll
  for (int i = 0; i < MAXCH; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) aw[i][j] = ab[i][j] = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wv; row < p.rows; row += (int64_t)gridDim.x * 4) {
    const float mean = p.mean[row], rstd = p.rstd[row];
    float g[MAXCH][8], xh[MAXCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int c = lane + 64 * i;
      if (c < p.nch) {
        float d[8], x[8], w[8];
        ld8<TY>(static_cast<const TY *>(p.dy) + row * p.dim + 8 * c, d);
        ld8<float>(p.xr + row * p.dim + 8 * c, x);
        ld8<float>(p.w + 8 * c, w);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xh[i][j] = (x[j] - mean) * rstd;
          g[i][j] = d[j] * w[j];
          s1 += g[i][j];
          s2 += g[i][j] * xh[i][j];
          aw[i][j] += d[j] * xh[i][j];
          ab[i][j] += d[j];
        }
      }
    }
    const float c1 = wave_sum(s1) / p.dim, c2 = wave_sum(s2) / p.dim;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int c = lane + 64 * i;
      if (c < p.nch) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rstd * (g[i][j] - c1 - xh[i][j] * c2);
        if (p.dres) {
          float r[8];
          ld8<float>(p.dres + row * p.dim + 8 * c, r);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += r[j];
        }
        if (p.dx32) st8<float>(p.dx32 + row * p.dim + 8 * c, o);
        if (p.dxh) st8<TH>(static_cast<TH *>(p.dxh) + row * p.dim + 8 * c, o);
      }
    }
  }
  // four waves -> one partial per workgroup, fixed order
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int c = lane + 64 * i;
    if (c < p.nch) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[wv][0][8 * c + j] = aw[i][j];
        red[wv][1][8 * c + j] = ab[i][j];
      }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * p.dim; e += 256) {
    const int k = e / p.dim, c = e % p.dim;
    p.slab[((int64_t)blockIdx.x * 2 + k) * p.dim + c] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
  }
}

// 64 outputs per workgroup; its 4 waves each add every 4th slab (fixed order), partial sums meet in LDS
__global__ __launch_bounds__(256) void add_layernorm_bwd_finalize_kernel(const float *slab, int nslab, int dim, float *dw,
                                                                         float *db) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  const bool live = e < 2 * dim;
  const int k = live ? e / dim : 0, c = live ? e % dim : 0;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // eight loads in flight per thread
  if (live) {
    int s = sl;
    for (; s + 28 < nslab; s += 32) {
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] += slab[((int64_t)(s + 4 * q) * 2 + k) * dim + c];
    }
    for (int q = 0; s < nslab; s += 4, ++q) a[q & 7] += slab[((int64_t)s * 2 + k) * dim + c];
  }
  red[sl][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (sl == 0 && live) {
    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (k == 0)
      dw[c] = v;
    else if (db)
      db[c] = v;
  }
}

// ---- any d_model (pruned checkpoints: 55, 114, 477 ...): rows are not 16-byte aligned, so the same arithmetic with
// element accesses, lane l owning elements l, l + 64, ... (<= 32 per lane).  Same launch geometry and slab layout.
constexpr int LN_MAXE = 32;

template <typename T>
__device__ __forceinline__ float ln_ld(const void *p, int64_t i) { return (float)static_cast<const T *>(p)[i]; }
template <typename T>
__device__ __forceinline__ void ln_st(void *p, int64_t i, float v) { static_cast<T *>(p)[i] = (T)v; }

template <typename TX, typename TY>
__global__ __launch_bounds__(256) void add_layernorm_fwd_any_kernel(const LnParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const int64_t xo = (row / p.len) * p.x_sb + (row % p.len) * p.x_sl;
  float v[LN_MAXE];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXE; ++i) {
    const int e = lane + 64 * i;
    v[i] = 0.f;
    if (e < p.dim) {
      v[i] = ln_ld<TX>(p.x, xo + e) + (p.res ? p.res[row * p.dim + e] : 0.f);
      p.res_out[row * p.dim + e] = v[i];
      s += v[i];
    }
  }
  const float mean = wave_sum(s) / p.dim;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXE; ++i)
    if (lane + 64 * i < p.dim) {
      const float d = v[i] - mean;
      q += d * d;
    }
  const float rstd = rsqrtf(wave_sum(q) / p.dim + p.eps);
  if (lane == 0) {
    p.mean[row] = mean;
    p.rstd[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < LN_MAXE; ++i) {
    const int e = lane + 64 * i;
    if (e < p.dim) ln_st<TY>(p.y, row * p.dim + e, (v[i] - mean) * rstd * p.w[e] + (p.b ? p.b[e] : 0.f));
  }
}

template <typename TY, typename TH>
__global__ __launch_bounds__(256) void add_layernorm_bwd_any_kernel(const LnBwdParams p) {
  __shared__ float red[4][2][LN_MAXE * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float aw[LN_MAXE], ab[LN_MAXE];
#pragma unroll
  for (int i = 0; i < LN_MAXE; ++i) aw[i] = ab[i] = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wv; row < p.rows; row += (int64_t)gridDim.x * 4) {
    const float mean = p.mean[row], rstd = p.rstd[row];
    float g[LN_MAXE], xh[LN_MAXE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXE; ++i) {
      const int e = lane + 64 * i;
      g[i] = xh[i] = 0.f;
      if (e < p.dim) {
        const float d = ln_ld<TY>(p.dy, row * p.dim + e);
        xh[i] = (p.xr[row * p.dim + e] - mean) * rstd;
        g[i] = d * p.w[e];
        s1 += g[i];
        s2 += g[i] * xh[i];
        aw[i] += d * xh[i];
        ab[i] += d;
      }
    }
    const float c1 = wave_sum(s1) / p.dim, c2 = wave_sum(s2) / p.dim;
#pragma unroll
    for (int i = 0; i < LN_MAXE; ++i) {
      const int e = lane + 64 * i;
      if (e < p.dim) {
        float o = rstd * (g[i] - c1 - xh[i] * c2);
        if (p.dres) o += p.dres[row * p.dim + e];
        if (p.dx32) p.dx32[row * p.dim + e] = o;
        if (p.dxh) ln_st<TH>(p.dxh, row * p.dim + e, o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LN_MAXE; ++i) {
    const int e = lane + 64 * i;
    if (e < p.dim) {
      red[wv][0][e] = aw[i];
      red[wv][1][e] = ab[i];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * p.dim; e += 256) {
    const int k = e / p.dim, c = e % p.dim;
    p.slab[((int64_t)blockIdx.x * 2 + k) * p.dim + c] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
  }
}

constexpr int kLnBwdGroups = 256;

}  // namespace cum

using namespace cum;

#define LN_FWD(TX, TY)                                                                              \
  do {                                                                                              \
    if (p.nch <= 64)                                                                                \
      hipLaunchKernelGGL((add_layernorm_fwd_kernel<TX, TY, 1>), grid, block, 0, st, p);            \
    else                                                                                            \
      hipLaunchKernelGGL((add_layernorm_fwd_kernel<TX, TY, LN_MAXCH>), grid, block, 0, st, p);     \
  } while (0)
#define LN_BWD(TY, TH)                                                                              \
  do {                                                                                              \
    if (p.nch <= 64)                                                                                \
      hipLaunchKernelGGL((add_layernorm_bwd_kernel<TY, TH, 1>), grid, block, 0, st, p);            \
    else                                                                                            \
      hipLaunchKernelGGL((add_layernorm_bwd_kernel<TY, TH, LN_MAXCH>), grid, block, 0, st, p);     \
  } while (0)

static int ln_check(int64_t batch, int32_t len, int32_t dim) {
  CUM_REQUIRE(batch >= 0 && len >= 0 && dim >= 1 && dim <= 64 * 8 * LN_MAXCH, "add_layernorm: d_model must be 1 ... 2048");
  return CUM_OK;
}

extern "C" int cum_add_layernorm_fwd(int32_t x_dtype, int32_t y_dtype, int64_t batch, int32_t len, int32_t dim,
                                     const void *x, int64_t x_sb, int64_t x_sl, const float *residual,
                                     const float *weight, const float *bias, float eps, float *residual_out, void *y,
                                     float *mean, float *rstd, void *stream) {
  if (int rc = ln_check(batch, len, dim)) return rc;
  CUM_REQUIRE(dtype_ok(x_dtype) && dtype_ok(y_dtype) && !(is16(x_dtype) && is16(y_dtype) && x_dtype != y_dtype),
              "add_layernorm: dtypes must be CUM_F32 / CUM_BF16 / CUM_F16 (one 16-bit type per call)");
  const int64_t rows = batch * len;
  if (rows == 0) return CUM_OK;
  CUM_REQUIRE(x && weight && residual_out && y && mean && rstd, "add_layernorm_fwd: null pointer");
  // vector path: 16-byte aligned rows everywhere; else the element-access kernels (any d_model, any strides)
  const bool vec = dim % 8 == 0 && x_sb % 8 == 0 && x_sl % 8 == 0 && ((uintptr_t)x & 15) == 0;
  LnParams p{};
  p.x = x; p.res = residual; p.w = weight; p.b = bias; p.res_out = residual_out; p.y = y; p.mean = mean; p.rstd = rstd;
  p.x_sb = x_sb; p.x_sl = x_sl; p.rows = rows; p.len = len; p.dim = dim; p.nch = dim / 8; p.eps = eps;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (!vec) {
    if (x_dtype == CUM_F16 && y_dtype == CUM_F16)
      hipLaunchKernelGGL((add_layernorm_fwd_any_kernel<f16, f16>), grid, block, 0, st, p);
    else if (x_dtype == CUM_F16)
      hipLaunchKernelGGL((add_layernorm_fwd_any_kernel<f16, float>), grid, block, 0, st, p);
    else if (y_dtype == CUM_F16)
      hipLaunchKernelGGL((add_layernorm_fwd_any_kernel<float, f16>), grid, block, 0, st, p);
    else if (x_dtype == CUM_BF16 && y_dtype == CUM_BF16)
      hipLaunchKernelGGL((add_layernorm_fwd_any_kernel<__bf16, __bf16>), grid, block, 0, st, p);
    else if (x_dtype == CUM_BF16)
      hipLaunchKernelGGL((add_layernorm_fwd_any_kernel<__bf16, float>), grid, block, 0, st, p);
    else if (y_dtype == CUM_BF16)
      hipLaunchKernelGGL((add_layernorm_fwd_any_kernel<float, __bf16>), grid, block, 0, st, p);
    else
      hipLaunchKernelGGL((add_layernorm_fwd_any_kernel<float, float>), grid, block, 0, st, p);
    CUM_CHECK_LAUNCH();
    return CUM_OK;
  }
  if (x_dtype == CUM_F16 && y_dtype == CUM_F16)
    LN_FWD(f16, f16);
  else if (x_dtype == CUM_F16)
    LN_FWD(f16, float);
  else if (y_dtype == CUM_F16)
    LN_FWD(float, f16);
  else if (x_dtype == CUM_BF16 && y_dtype == CUM_BF16)
    LN_FWD(__bf16, __bf16);
  else if (x_dtype == CUM_BF16)
    LN_FWD(__bf16, float);
  else if (y_dtype == CUM_BF16)
    LN_FWD(float, __bf16);
  else
    LN_FWD(float, float);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

extern "C" int64_t cum_add_layernorm_bwd_workspace_elems(int32_t dim) { return (int64_t)kLnBwdGroups * 2 * dim; }

extern "C" int cum_add_layernorm_bwd(int32_t y_dtype, int32_t h_dtype, int64_t rows, int32_t dim, const void *dy,
                                     const float *dres_out, const float *residual_out, const float *mean,
                                     const float *rstd, const float *weight, float *dx32, void *dxh, float *dweight,
                                     float *dbias, float *workspace, void *stream) {
  if (int rc = ln_check(rows, 1, dim)) return rc;
  CUM_REQUIRE(dtype_ok(y_dtype) && dtype_ok(h_dtype) && !(is16(y_dtype) && is16(h_dtype) && y_dtype != h_dtype),
              "add_layernorm: dtypes must be CUM_F32 / CUM_BF16 / CUM_F16 (one 16-bit type per call)");
  CUM_REQUIRE(dweight && workspace, "add_layernorm_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) {
    (void)hipMemsetAsync(dweight, 0, sizeof(float) * dim, st);
    if (dbias) (void)hipMemsetAsync(dbias, 0, sizeof(float) * dim, st);
    return CUM_OK;
  }
  CUM_REQUIRE(dy && residual_out && mean && rstd && weight, "add_layernorm_bwd: null pointer");
  LnBwdParams p{};
  p.dy = dy; p.dres = dres_out; p.xr = residual_out; p.mean = mean; p.rstd = rstd; p.w = weight;
  p.dx32 = dx32; p.dxh = dxh; p.slab = workspace; p.rows = rows; p.dim = dim; p.nch = dim / 8;
  const int groups = (int)((rows + 3) / 4 < kLnBwdGroups ? (rows + 3) / 4 : kLnBwdGroups);
  dim3 grid(groups), block(256);
  if (dim % 8 != 0) {
    if (y_dtype == CUM_F16 && h_dtype == CUM_F16)
      hipLaunchKernelGGL((add_layernorm_bwd_any_kernel<f16, f16>), grid, block, 0, st, p);
    else if (y_dtype == CUM_F16)
      hipLaunchKernelGGL((add_layernorm_bwd_any_kernel<f16, float>), grid, block, 0, st, p);
    else if (h_dtype == CUM_F16)
      hipLaunchKernelGGL((add_layernorm_bwd_any_kernel<float, f16>), grid, block, 0, st, p);
    else if (y_dtype == CUM_BF16 && h_dtype == CUM_BF16)
      hipLaunchKernelGGL((add_layernorm_bwd_any_kernel<__bf16, __bf16>), grid, block, 0, st, p);
    else if (y_dtype == CUM_BF16)
      hipLaunchKernelGGL((add_layernorm_bwd_any_kernel<__bf16, float>), grid, block, 0, st, p);
    else if (h_dtype == CUM_BF16)
      hipLaunchKernelGGL((add_layernorm_bwd_any_kernel<float, __bf16>), grid, block, 0, st, p);
    else
      hipLaunchKernelGGL((add_layernorm_bwd_any_kernel<float, float>), grid, block, 0, st, p);
  } else
  if (y_dtype == CUM_F16 && h_dtype == CUM_F16)
    LN_BWD(f16, f16);
  else if (y_dtype == CUM_F16)
    LN_BWD(f16, float);
  else if (h_dtype == CUM_F16)
    LN_BWD(float, f16);
  else if (y_dtype == CUM_BF16 && h_dtype == CUM_BF16)
    LN_BWD(__bf16, __bf16);
  else if (y_dtype == CUM_BF16)
    LN_BWD(__bf16, float);
  else if (h_dtype == CUM_BF16)
    LN_BWD(float, __bf16);
  else
    LN_BWD(float, float);
  CUM_CHECK_LAUNCH();
  hipLaunchKernelGGL(add_layernorm_bwd_finalize_kernel, dim3((2 * dim + 63) / 64), dim3(256), 0, st, workspace, groups,
                     dim, dweight, dbias);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}
