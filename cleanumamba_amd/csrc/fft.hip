// Batched real FFTs of the STFT loss straight on hipFFT / rocFFT (reference: torch.stft inside
// src/util/stft_loss.py:16-38 and its autograd).  torch.fft.rfft / irfft clone their input on ROCm because rocFFT's
// real transforms may overwrite it; here the inputs are scratch buffers of the loss (windowed frames, spectrum
// gradient), so they are handed over as destroyable and the six clones per step disappear.  Plans are cached per
// (length, batch, direction); their work areas are allocated by hipFFT once, at plan creation.
#include <hipfft/hipfft.h>

#include <map>
#include <mutex>
#include <tuple>

#include "common.h"

namespace {

std::mutex g_mu;
std::map<std::tuple<int, int, long long, int>, hipfftHandle> g_plans;   // (device, n, batch, type)

int get_plan(int n, long long batch, hipfftType type, hipfftHandle *out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return CUM_ELAUNCH;
  const auto key = std::make_tuple(dev, n, batch, (int)type);
  auto it = g_plans.find(key);
  if (it == g_plans.end()) {
    hipfftHandle h;
    int len[1] = {n};
    if (hipfftPlanMany(&h, 1, len, nullptr, 1, 0, nullptr, 1, 0, type, (int)batch) != HIPFFT_SUCCESS) {
      cum_set_error("fft: hipfftPlanMany failed");
      return CUM_ELAUNCH;
    }
    it = g_plans.emplace(key, h).first;
  }
  *out = it->second;
  return CUM_OK;
}

}  // namespace

// in: [batch][n] real (MAY BE OVERWRITTEN); out: [batch][n/2 + 1] interleaved complex.  Unnormalised.
extern "C" int cum_rfft(int32_t n, int64_t batch, float *in, float *out, void *stream) {
  CUM_REQUIRE(n >= 2 && n % 2 == 0 && batch >= 0 && batch < 2147483647LL, "rfft: bad length or batch");
  if (batch == 0) return CUM_OK;
  CUM_REQUIRE(in && out, "rfft: null pointer");
  std::lock_guard<std::mutex> lock(g_mu);
  hipfftHandle h;
  if (int rc = get_plan(n, batch, HIPFFT_R2C, &h)) return rc;
  if (hipfftSetStream(h, (hipStream_t)stream) != HIPFFT_SUCCESS ||
      hipfftExecR2C(h, in, reinterpret_cast<hipfftComplex *>(out)) != HIPFFT_SUCCESS) {
    cum_set_error("rfft: hipfftExecR2C failed");
    return CUM_ELAUNCH;
  }
  return CUM_OK;
}

// in: [batch][n/2 + 1] interleaved complex (MAY BE OVERWRITTEN); out: [batch][n] real.  Unnormalised (no 1/n).
extern "C" int cum_irfft(int32_t n, int64_t batch, float *in, float *out, void *stream) {
  CUM_REQUIRE(n >= 2 && n % 2 == 0 && batch >= 0 && batch < 2147483647LL, "irfft: bad length or batch");
  if (batch == 0) return CUM_OK;
  CUM_REQUIRE(in && out, "irfft: null pointer");
  std::lock_guard<std::mutex> lock(g_mu);
  hipfftHandle h;
  if (int rc = get_plan(n, batch, HIPFFT_C2R, &h)) return rc;
  if (hipfftSetStream(h, (hipStream_t)stream) != HIPFFT_SUCCESS ||
      hipfftExecC2R(h, reinterpret_cast<hipfftComplex *>(in), out) != HIPFFT_SUCCESS) {
    cum_set_error("irfft: hipfftExecC2R failed");
    return CUM_ELAUNCH;
  }
  return CUM_OK;
}

// in, out: [batch][n] interleaved complex (in == out allowed).  Unnormalised in both directions.
extern "C" int cum_cfft(int32_t n, int64_t batch, float *in, float *out, int32_t inverse, void *stream) {
  CUM_REQUIRE(n >= 2 && batch >= 0 && batch < 2147483647LL, "cfft: bad length or batch");
  if (batch == 0) return CUM_OK;
  CUM_REQUIRE(in && out, "cfft: null pointer");
  std::lock_guard<std::mutex> lock(g_mu);
  hipfftHandle h;
  if (int rc = get_plan(n, batch, HIPFFT_C2C, &h)) return rc;
  if (hipfftSetStream(h, (hipStream_t)stream) != HIPFFT_SUCCESS ||
      hipfftExecC2C(h, reinterpret_cast<hipfftComplex *>(in), reinterpret_cast<hipfftComplex *>(out),
                    inverse ? HIPFFT_BACKWARD : HIPFFT_FORWARD) != HIPFFT_SUCCESS) {
    cum_set_error("cfft: hipfftExecC2C failed");
    return CUM_ELAUNCH;
  }
  return CUM_OK;
}
