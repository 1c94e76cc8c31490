// Batched 1-D FFTs of the STFT loss straight on hipFFT / rocFFT (reference: torch.stft inside
// src/util/stft_loss.py:16-38 and its autograd).  torch.fft.rfft / irfft clone their input on ROCm because rocFFT's
// real transforms may overwrite it; here the inputs are scratch buffers of the loss (windowed frames, spectrum
// gradient), so they are handed over as destroyable and the six clones per step disappear.
//
// No state lives in this library: a plan is an object the CALLER creates, keeps and destroys (cum_fft_plan_create /
// _destroy), and its work area is caller memory handed over with every transform (hipfftSetAutoAllocation(0) +
// hipfftSetWorkArea) -- the rule of include/cleanumamba_hip.h, "kernels never allocate, no global mutable state", holds
// for the transforms too (round 3 kept a process-global, mutex-guarded plan cache here and let hipFFT allocate the work
// areas on first use).  What rocFFT allocates inside hipfftMakePlanMany (twiddle tables) belongs to the plan object and is
// released by cum_fft_plan_destroy.
#include <hipfft/hipfft.h>

#include "common.h"

namespace {

struct FftPlan {
  hipfftHandle h;
  int32_t kind, n;
  int64_t batch;
  size_t work_bytes;
};

}  // namespace

// kind: 0 = real -> complex (n real in, n/2 + 1 complex out), 1 = complex -> real, 2 = complex -> complex.
// *work_bytes: size of the work area every cum_fft_exec of this plan must be given (may be 0).
extern "C" int cum_fft_plan_create(int32_t kind, int32_t n, int64_t batch, void **plan, int64_t *work_bytes) {
  CUM_REQUIRE(plan && work_bytes, "fft_plan_create: null argument");
  CUM_REQUIRE(kind >= 0 && kind <= 2, "fft_plan_create: kind must be 0 (r2c), 1 (c2r) or 2 (c2c)");
  CUM_REQUIRE(n >= 2 && (kind == 2 || n % 2 == 0) && batch >= 1 && batch < 2147483647LL, "fft_plan_create: bad length or batch");
  const hipfftType type = kind == 0 ? HIPFFT_R2C : kind == 1 ? HIPFFT_C2R : HIPFFT_C2C;
  FftPlan *p = new (std::nothrow) FftPlan{};
  CUM_REQUIRE(p, "fft_plan_create: out of host memory");
  p->kind = kind; p->n = n; p->batch = batch;
  int len[1] = {n};
  size_t ws = 0;
  if (hipfftCreate(&p->h) != HIPFFT_SUCCESS) {
    delete p;
    cum_set_error("fft_plan_create: hipfftCreate failed");
    return CUM_ELAUNCH;
  }
  if (hipfftSetAutoAllocation(p->h, 0) != HIPFFT_SUCCESS ||
      hipfftMakePlanMany(p->h, 1, len, nullptr, 1, 0, nullptr, 1, 0, type, (int)batch, &ws) != HIPFFT_SUCCESS) {
    hipfftDestroy(p->h);
    delete p;
    cum_set_error("fft_plan_create: hipfftMakePlanMany failed");
    return CUM_ELAUNCH;
  }
  p->work_bytes = ws;
  *plan = p;
  *work_bytes = (int64_t)ws;
  return CUM_OK;
}

extern "C" int cum_fft_plan_destroy(void *plan) {
  if (!plan) return CUM_OK;
  FftPlan *p = static_cast<FftPlan *>(plan);
  hipfftDestroy(p->h);
  delete p;
  return CUM_OK;
}

// One batched transform, unnormalised in both directions.  in MAY BE OVERWRITTEN by the real transforms (pass scratch);
// c2c allows in == out.  work: device memory of the plan's work_bytes (NULL if 0), 16-byte aligned, not shared with a
// transform that may run concurrently.  A plan is not re-entrant: one thread / one stream at a time (the stream is
// recorded in the plan for the duration of the call).
extern "C" int cum_fft_exec(void *plan, float *in, float *out, int32_t inverse, void *work, void *stream) {
  CUM_REQUIRE(plan && in && out, "fft_exec: null argument");
  FftPlan *p = static_cast<FftPlan *>(plan);
  CUM_REQUIRE(p->work_bytes == 0 || work, "fft_exec: this plan needs its work area");
  hipfftResult rc = hipfftSetStream(p->h, (hipStream_t)stream);
  if (rc == HIPFFT_SUCCESS && p->work_bytes) rc = hipfftSetWorkArea(p->h, work);
  if (rc == HIPFFT_SUCCESS) {
    if (p->kind == 0)
      rc = hipfftExecR2C(p->h, in, reinterpret_cast<hipfftComplex *>(out));
    else if (p->kind == 1)
      rc = hipfftExecC2R(p->h, reinterpret_cast<hipfftComplex *>(in), out);
    else
      rc = hipfftExecC2C(p->h, reinterpret_cast<hipfftComplex *>(in), reinterpret_cast<hipfftComplex *>(out),
                         inverse ? HIPFFT_BACKWARD : HIPFFT_FORWARD);
  }
  if (rc != HIPFFT_SUCCESS) {
    cum_set_error("fft_exec: hipFFT call failed");
    return CUM_ELAUNCH;
  }
  return CUM_OK;
}
