// One Mamba block for one token of every stream, in ONE launch (streaming inference of small / pruned models).
//
// Replaces, for the step path of the reference (Block.forward + Mamba.step, reached from
// CleanUMamba._denoise_frame, src/network/CleanUMamba.py:451-454): residual add + LayerNorm, in_proj, the causal
// conv update, x_proj, dt_proj + softplus, the selective state update with the D skip and the silu(z) gate, out_proj
// -- a dozen launches of tiny kernels per block and hop.  A workgroup owns one stream; every intermediate vector lives
// in LDS; matrix rows are read with one wave per output row (coalesced, reduced with DPP shuffles).  Meant for models
// whose projection matrices are small enough to be re-read from L2 by every stream (the host checks the sizes).
#include "common.h"

namespace cum {

constexpr int kStepMaxModel = 1024;   // d_model
constexpr int kStepMaxInner = 512;    // d_inner
constexpr int kStepMaxXdb = 256;      // dt_rank + 2 d_state

struct StepParams {
  int streams, d_model, d_inner, d_state, dt_rank, d_conv;
  float eps;
  const float *hidden_in, *residual_in, *norm_w, *norm_b;
  const float *in_w, *in_b;
  float *conv_state;
  const float *conv_w, *conv_b;
  const float *xproj_w, *dtproj_w, *dtproj_b, *A, *D;
  float *ssm_state;
  const float *out_w, *out_b;
  float *hidden_out, *residual_out;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// out[j] = bias[j] + sum_k W[j][k] * vec[k] for j in [0, rows): one wave per row, lanes over k
__device__ __forceinline__ void matvec(const float *__restrict__ W, const float *__restrict__ bias, const float *vec,
                                       float *out, int rows, int cols, int wave, int lane, int nwaves) {
  for (int j = wave; j < rows; j += nwaves) {
    const float *w = W + (int64_t)j * cols;
    float acc = 0.f;
    for (int k = lane; k < cols; k += 64) acc = fmaf(w[k], vec[k], acc);
    acc = wave_sum(acc);
    if (lane == 0) out[j] = acc + (bias ? bias[j] : 0.f);
  }
}

// Small matrices (rows * (cols + 1) <= kStepStage floats): stage W into LDS with coalesced reads (row pitch cols + 1:
// conflict-free column walks), then one thread per output row -- two barriers instead of rows / 4 dependent
// load -> shuffle-reduce rounds per wave.  Larger ones keep the wave-per-row form.  Ends with a barrier either way.
constexpr int kStepStage = 12288;
__device__ __forceinline__ void matvec_small(const float *__restrict__ W, const float *__restrict__ bias, const float *vec,
                                             float *out, int rows, int cols, float *s_w, int tid, int wave, int lane) {
  if (rows * (cols + 1) <= kStepStage) {
    const int total = rows * cols;
    for (int e = tid; e < total; e += 256) s_w[(e / cols) * (cols + 1) + e % cols] = W[e];
    __syncthreads();
    for (int j = tid; j < rows; j += 256) {
      const float *w = s_w + j * (cols + 1);
      float a0 = bias ? bias[j] : 0.f, a1 = 0.f;
      int k = 0;
      for (; k + 1 < cols; k += 2) {
        a0 = fmaf(w[k], vec[k], a0);
        a1 = fmaf(w[k + 1], vec[k + 1], a1);
      }
      if (k < cols) a0 = fmaf(w[k], vec[k], a0);
      out[j] = a0 + a1;
    }
  } else {
    matvec(W, bias, vec, out, rows, cols, wave, lane, 4);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void mamba_step_kernel(const StepParams p) {
  __shared__ float s_w[kStepStage];
  __shared__ float s_h[kStepMaxModel];            // normed hidden, later reused for nothing else
  __shared__ float s_xz[2 * kStepMaxInner];
  __shared__ float s_x[kStepMaxInner];
  __shared__ float s_y[kStepMaxInner];
  __shared__ float s_dt[kStepMaxInner];
  __shared__ float s_xdb[kStepMaxXdb];
  __shared__ float s_red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t s = blockIdx.x;
  const int dm = p.d_model, di = p.d_inner, N = p.d_state, R = p.dt_rank, W = p.d_conv;

  // ---- residual add + LayerNorm (two-pass: mean, then variance of the centred values)
  const float *hin = p.hidden_in + s * dm;
  const float *rin = p.residual_in ? p.residual_in + s * dm : nullptr;
  float *rout = p.residual_out + s * dm;
  float part = 0.f;
  for (int k = tid; k < dm; k += 256) {
    const float r = hin[k] + (rin ? rin[k] : 0.f);
    rout[k] = r;
    s_h[k] = r;
    part += r;
  }
  part = wave_sum(part);
  if (lane == 0) s_red[0][wave] = part;
  __syncthreads();
  const float mean = (s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3]) / dm;
  part = 0.f;
  for (int k = tid; k < dm; k += 256) {
    const float d = s_h[k] - mean;
    part += d * d;
  }
  part = wave_sum(part);
  if (lane == 0) s_red[1][wave] = part;
  __syncthreads();
  const float rstd = rsqrtf((s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3]) / dm + p.eps);
  for (int k = tid; k < dm; k += 256) s_h[k] = (s_h[k] - mean) * rstd * p.norm_w[k] + (p.norm_b ? p.norm_b[k] : 0.f);
  __syncthreads();

  // ---- in_proj: xz = W_in h
  matvec_small(p.in_w, p.in_b, s_h, s_xz, 2 * di, dm, s_w, tid, wave, lane);

  // ---- causal conv update + SiLU (state shifted in place)
  for (int d = tid; d < di; d += 256) {
    float *st = p.conv_state + (s * di + d) * W;
    float acc = p.conv_b ? p.conv_b[d] : 0.f;
    for (int k = 0; k < W; ++k) {
      const float v = (k + 1 < W) ? st[k + 1] : s_xz[d];
      st[k] = v;
      acc = fmaf(p.conv_w[d * W + k], v, acc);
    }
    s_x[d] = acc * sigmoidf_(acc);
  }
  __syncthreads();

  // ---- x_proj -> (dt_low, B, C); dt = softplus(W_dt dt_low + b_dt)
  matvec_small(p.xproj_w, nullptr, s_x, s_xdb, R + 2 * N, di, s_w, tid, wave, lane);
  for (int d = tid; d < di; d += 256) {
    const float *w = p.dtproj_w + (int64_t)d * R;
    float acc = p.dtproj_b ? p.dtproj_b[d] : 0.f;
    for (int r = 0; r < R; ++r) acc = fmaf(w[r], s_xdb[r], acc);
    s_dt[d] = softplus20(acc);
  }
  __syncthreads();

  // ---- state update: one wave per channel, lanes over the states
  const float *Bv = s_xdb + R, *Cv = s_xdb + R + N;
  for (int d = wave; d < di; d += 4) {
    const float dt = s_dt[d], x = s_x[d];
    float *st = p.ssm_state + (s * di + d) * N;
    float acc = 0.f;
    for (int n = lane; n < N; n += 64) {
      const float a = __builtin_amdgcn_exp2f(dt * p.A[(int64_t)d * N + n] * kLog2e);
      const float v = fmaf(a, st[n], dt * Bv[n] * x);
      st[n] = v;
      acc = fmaf(Cv[n], v, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      const float z = s_xz[di + d];
      s_y[d] = (acc + (p.D ? p.D[d] : 0.f) * x) * (z * sigmoidf_(z));
    }
  }
  __syncthreads();

  // ---- out_proj
  matvec_small(p.out_w, p.out_b, s_y, p.hidden_out + s * dm, dm, di, s_w, tid, wave, lane);
}

// out[s][j] = bias[j] + sum_k W[j][k] x[s][k]: the 1x1 bottleneck convolutions of a hop on one-column inputs (a
// library GEMM takes 25 us for a 256 x 85 x 114 problem); one workgroup per stream, W staged in LDS.
__global__ __launch_bounds__(256) void small_linear_kernel(const float *__restrict__ x, int64_t x_stride,
                                                           const float *__restrict__ W, const float *__restrict__ bias,
                                                           float *__restrict__ out, int64_t out_stride, int rows,
                                                           int cols) {
  __shared__ float s_w[kStepStage];
  __shared__ float s_v[kStepMaxModel];
  const int tid = threadIdx.x;
  const float *xv = x + blockIdx.x * x_stride;
  for (int k = tid; k < cols; k += 256) s_v[k] = xv[k];
  __syncthreads();
  matvec_small(W, bias, s_v, out + blockIdx.x * out_stride, rows, cols, s_w, tid, tid >> 6, tid & 63);
}

}  // namespace cum

using namespace cum;

extern "C" int cum_small_linear(int32_t streams, int32_t rows, int32_t cols, const float *x, int64_t x_stride,
                                const float *W, const float *bias, float *out, int64_t out_stride, void *stream) {
  CUM_REQUIRE(streams >= 0 && rows >= 1 && cols >= 1 && cols <= kStepMaxModel && x_stride >= cols && out_stride >= rows,
              "small_linear: bad shape (cols <= 1024)");
  if (streams == 0) return CUM_OK;
  CUM_REQUIRE(x && W && out, "small_linear: null pointer");
  hipLaunchKernelGGL(small_linear_kernel, dim3(streams), dim3(256), 0, (hipStream_t)stream, x, x_stride, W, bias, out,
                     out_stride, rows, cols);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}


extern "C" int cum_mamba_step_supported(int32_t d_model, int32_t d_inner, int32_t d_state, int32_t dt_rank,
                                        int32_t d_conv) {
  return d_model >= 1 && d_model <= kStepMaxModel && d_inner >= 1 && d_inner <= kStepMaxInner && d_state >= 1 &&
         dt_rank >= 1 && dt_rank + 2 * d_state <= kStepMaxXdb && d_conv >= 1 && d_conv <= 8 &&
         (int64_t)d_model * d_inner <= 65536;
}

extern "C" int cum_mamba_step(int32_t streams, int32_t d_model, int32_t d_inner, int32_t d_state, int32_t dt_rank,
                              int32_t d_conv, float eps, const float *hidden_in, const float *residual_in,
                              const float *norm_w, const float *norm_b, const float *in_proj_w, const float *in_proj_b,
                              float *conv_state, const float *conv_w, const float *conv_b, const float *x_proj_w,
                              const float *dt_proj_w, const float *dt_proj_b, const float *A, const float *D,
                              float *ssm_state, const float *out_proj_w, const float *out_proj_b, float *hidden_out,
                              float *residual_out, void *stream) {
  CUM_REQUIRE(streams >= 0 && cum_mamba_step_supported(d_model, d_inner, d_state, dt_rank, d_conv),
              "mamba_step: sizes outside the fused step's limits (cum_mamba_step_supported)");
  if (streams == 0) return CUM_OK;
  CUM_REQUIRE(hidden_in && norm_w && in_proj_w && conv_state && conv_w && x_proj_w && dt_proj_w && A && ssm_state &&
                  out_proj_w && hidden_out && residual_out,
              "mamba_step: null pointer");
  StepParams p{};
  p.streams = streams; p.d_model = d_model; p.d_inner = d_inner; p.d_state = d_state; p.dt_rank = dt_rank;
  p.d_conv = d_conv; p.eps = eps;
  p.hidden_in = hidden_in; p.residual_in = residual_in; p.norm_w = norm_w; p.norm_b = norm_b;
  p.in_w = in_proj_w; p.in_b = in_proj_b; p.conv_state = conv_state; p.conv_w = conv_w; p.conv_b = conv_b;
  p.xproj_w = x_proj_w; p.dtproj_w = dt_proj_w; p.dtproj_b = dt_proj_b; p.A = A; p.D = D; p.ssm_state = ssm_state;
  p.out_w = out_proj_w; p.out_b = out_proj_b; p.hidden_out = hidden_out; p.residual_out = residual_out;
  hipLaunchKernelGGL(mamba_step_kernel, dim3(streams), dim3(256), 0, (hipStream_t)stream, p);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}
