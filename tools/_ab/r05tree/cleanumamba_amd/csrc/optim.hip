// Gradient clipping + Adam (+ dynamic loss scaling) over FLAT parameter / gradient buffers, for gfx950.
//
// Replaces the optimizer section of the reference's hot loop (src/training/train.py:303-310:
// scaler.unscale_ -> clip_grad_norm_(10) -> scaler.step(Adam) -> scaler.update, optimizer built at :145-148,
// GradScaler at :158-160).  On PyTorch that is ~10 multi-tensor launches over 103 tensors plus host
// bookkeeping (and, with GradScaler, a device->host sync per step).  Here the parameters, their gradients and
// both Adam moments live in four flat f32 buffers (training/flat_optim.py), so the section is three launches
// with no host synchronisation, which also makes the whole train step capturable in a hipGraph:
//   cum_optim_sumsq    per-block partial sums of g^2 (fixed order: deterministic)
//   cum_optim_prepare  one workgroup: total norm, inf/nan check, clip coefficient, loss-scale update, Adam step
//                      count and bias corrections -> `state` (device memory; nothing returns to the host)
//   cum_optim_adam     p, m, v updated from g * state.grad_mult; skipped as a whole when state.found_inf
// Pure HBM streaming: 16 B read + 12 B written per element.
#include "common.h"

namespace cum {

// state vector (f32), shared by the three kernels and read by the host side only for logging
enum { ST_NORM = 0, ST_MULT = 1, ST_FOUND_INF = 2, ST_SCALE = 3, ST_TRACKER = 4, ST_STEP = 5, ST_BC1 = 6,
       ST_BC2_SQRT = 7, ST_LR = 8, ST_SKIPPED = 9 };

__global__ __launch_bounds__(256) void optim_sumsq_kernel(const float *__restrict__ g, int64_t n, float *__restrict__ partials) {
  __shared__ float red[4];
  const int64_t n4 = n >> 2;
  float a0 = 0.f, a1 = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {     // two independent chains keep two 16-byte loads in flight
    const float4 x = reinterpret_cast<const float4 *>(g)[i], y = reinterpret_cast<const float4 *>(g)[i + stride];
    a0 += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    a1 += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
  }
  if (i < n4) {
    const float4 x = reinterpret_cast<const float4 *>(g)[i];
    a0 += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float x = g[(n4 << 2) + threadIdx.x];
    a1 += x * x;
  }
  float s = a0 + a1;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct PrepareArgs {
  float *state;
  const float *partials;
  int nparts;
  float max_norm;          // <= 0: no clipping
  double beta1, beta2;
  int use_scaler;          // 1: gradients carry state[ST_SCALE]; dynamic scale update as torch.amp.GradScaler
  float growth, backoff;
  int growth_interval;
};

__global__ __launch_bounds__(256) void optim_prepare_kernel(const PrepareArgs a) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < a.nparts; i += 256) s += (double)a.partials[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  float *st = a.state;
  const double sumsq = red[0];
  const float scale = a.use_scaler ? st[ST_SCALE] : 1.f;
  const bool bad = !(sumsq == sumsq) || sumsq > 1.7e308;   // nan / inf (an f32 partial that overflowed is inf)
  const float norm = bad ? __builtin_inff() : (float)(sqrt(sumsq) / (double)scale);  // norm of the UNSCALED gradient
  float coef = 1.f;
  if (a.max_norm > 0.f && !bad) {
    coef = a.max_norm / (norm + 1e-6f);            // torch.nn.utils.clip_grad_norm_
    coef = coef < 1.f ? coef : 1.f;
  }
  st[ST_NORM] = norm;
  st[ST_MULT] = coef / scale;
  st[ST_FOUND_INF] = bad ? 1.f : 0.f;
  if (bad) {
    st[ST_SKIPPED] += 1.f;
  } else {
    const float t = st[ST_STEP] + 1.f;
    st[ST_STEP] = t;
    st[ST_BC1] = (float)(1.0 - pow(a.beta1, (double)t));
    st[ST_BC2_SQRT] = (float)sqrt(1.0 - pow(a.beta2, (double)t));
  }
  if (a.use_scaler) {                              // torch.amp.GradScaler.update()
    if (bad) {
      st[ST_SCALE] = scale * a.backoff;
      st[ST_TRACKER] = 0.f;
    } else {
      const float tr = st[ST_TRACKER] + 1.f;
      if (tr >= (float)a.growth_interval) {
        st[ST_SCALE] = scale * a.growth;
        st[ST_TRACKER] = 0.f;
      } else {
        st[ST_TRACKER] = tr;
      }
    }
  }
}

struct AdamArgs {
  float *p, *m, *v;
  const float *g;
  int64_t n;
  const float *state;
  float beta1, beta2, omb1, omb2, eps, weight_decay;    // omb = 1 - beta, rounded from the double-precision difference
};

__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, float mult, float lr_bc1, float inv_bc2s,
                                      const AdamArgs &a) {
  g *= mult;
  if (a.weight_decay != 0.f) g = fmaf(a.weight_decay, p, g);      // Adam's L2 form (torch.optim.Adam)
  m = fmaf(a.beta1, m, a.omb1 * g);
  v = fmaf(a.beta2, v, a.omb2 * (g * g));
  const float denom = sqrtf(v) * inv_bc2s + a.eps;
  p -= lr_bc1 * (m / denom);
}

__global__ __launch_bounds__(256) void optim_adam_kernel(const AdamArgs a) {
  const float *st = a.state;
  if (st[ST_FOUND_INF] != 0.f) return;           // the whole step is skipped (GradScaler semantics)
  const float mult = st[ST_MULT], lr_bc1 = st[ST_LR] / st[ST_BC1], inv_bc2s = 1.f / st[ST_BC2_SQRT];
  const int64_t n4 = a.n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 p = reinterpret_cast<float4 *>(a.p)[i], m = reinterpret_cast<float4 *>(a.m)[i];
    float4 v = reinterpret_cast<float4 *>(a.v)[i];
    const float4 g = reinterpret_cast<const float4 *>(a.g)[i];
    adam1(p.x, g.x, m.x, v.x, mult, lr_bc1, inv_bc2s, a);
    adam1(p.y, g.y, m.y, v.y, mult, lr_bc1, inv_bc2s, a);
    adam1(p.z, g.z, m.z, v.z, mult, lr_bc1, inv_bc2s, a);
    adam1(p.w, g.w, m.w, v.w, mult, lr_bc1, inv_bc2s, a);
    reinterpret_cast<float4 *>(a.p)[i] = p;
    reinterpret_cast<float4 *>(a.m)[i] = m;
    reinterpret_cast<float4 *>(a.v)[i] = v;
  }
  if (blockIdx.x == 0 && (int64_t)threadIdx.x < (a.n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    adam1(a.p[i], a.g[i], a.m[i], a.v[i], mult, lr_bc1, inv_bc2s, a);
  }
}

}  // namespace cum

using namespace cum;

extern "C" int32_t cum_optim_state_elems(void) { return 16; }

extern "C" int32_t cum_optim_sumsq_parts(int64_t n) {
  const int64_t want = (n / 4 + 2047) / 2048;       // >= 8 float4 per thread
  return (int32_t)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
}

extern "C" int cum_optim_sumsq(const float *g, int64_t n, float *partials, void *stream) {
  CUM_REQUIRE(g && partials && n >= 0 && ((uintptr_t)g & 15) == 0, "optim_sumsq: null or misaligned pointer");
  hipLaunchKernelGGL(optim_sumsq_kernel, dim3(cum_optim_sumsq_parts(n)), dim3(256), 0, (hipStream_t)stream, g, n, partials);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

extern "C" int cum_optim_prepare(float *state, const float *partials, int32_t nparts, float max_norm, double beta1,
                                 double beta2, int32_t use_scaler, float growth, float backoff, int32_t growth_interval,
                                 void *stream) {
  CUM_REQUIRE(state && partials && nparts > 0, "optim_prepare: bad argument");
  CUM_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "optim_prepare: betas must be in [0, 1)");
  PrepareArgs a{state, partials, nparts, max_norm, beta1, beta2, use_scaler, growth, backoff, growth_interval};
  hipLaunchKernelGGL(optim_prepare_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

extern "C" int cum_optim_adam(float *p, const float *g, float *m, float *v, int64_t n, const float *state, double beta1,
                              double beta2, float eps, float weight_decay, void *stream) {
  CUM_REQUIRE(p && g && m && v && state && n >= 0, "optim_adam: null pointer");
  CUM_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "optim_adam: buffers must be 16-byte aligned");
  if (n == 0) return CUM_OK;
  AdamArgs a{p, m, v, g, n, state, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps, weight_decay};
  const int64_t want = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(optim_adam_kernel, dim3((unsigned)(want < 1 ? 1 : (want > 4096 ? 4096 : want))), dim3(256), 0,
                     (hipStream_t)stream, a);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}
