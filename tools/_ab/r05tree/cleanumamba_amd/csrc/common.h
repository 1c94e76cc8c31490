// Shared device/host helpers for libcleanumamba_hip (gfx950 only: wave64, 4 SIMDs/CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cleanumamba_hip.h"

#define CUM_WAVE 64

extern "C" void cum_set_error(const char *msg);

#define CUM_REQUIRE(cond, msg)  \
  do {                          \
    if (!(cond)) {              \
      cum_set_error(msg);       \
      return CUM_EINVAL;        \
    }                           \
  } while (0)

#define CUM_CHECK_LAUNCH()                      \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) {                    \
      cum_set_error(hipGetErrorString(e__));    \
      return CUM_ELAUNCH;                       \
    }                                           \
  } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Experiment switches.  The shipped library (plain `make`) reads NO environment variable and carries only the kernels the
// dispatch rules below choose: every alternative kernel kept for same-box A/B runs, and every CUM_* knob that selects
// one, exists only in a `make AB=1` build (-DCUM_AB).  cum_knob(name, dflt) is the knob's value there and the constant
// `dflt` here.
#ifdef CUM_AB
#include <stdlib.h>
static inline int64_t cum_knob(const char *name, int64_t dflt) {
  const char *e = getenv(name);
  return e ? atoll(e) : dflt;
}
#else
#define cum_knob(name, dflt) ((int64_t)(dflt))
#endif

namespace cum {

typedef _Float16 f16;
// 16-bit element types share every tiling decision (8 elements per 16-byte chunk, K tile 64)
static inline bool is16(int dt) { return dt == CUM_BF16 || dt == CUM_F16; }
static inline bool dtype_ok(int dt) { return dt == CUM_F32 || dt == CUM_BF16 || dt == CUM_F16; }

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// Hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, 1 ulp each).  The libm
// expf/log1pf expand to ~100 instructions apiece, which would cost as much per (t, d)
// as the 64 state updates they feed.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// log1p(e) for e >= 0 with the u = 1 + e compensation: log(u) * e / (u - 1).
__device__ __forceinline__ float fast_log1p(float e) {
  const float u = 1.f + e;
  const float l = __builtin_amdgcn_logf(u) * kLn2;
  const float um1 = u - 1.f;
  return um1 == 0.f ? e : l * (e * fast_rcp(um1));
}
// softplus with the upstream threshold: x <= 20 ? log1p(exp(x)) : x
__device__ __forceinline__ float softplus20(float x) { return x <= 20.f ? fast_log1p(fast_exp(x)) : x; }
__device__ __forceinline__ float sigmoidf_(float x) { return fast_rcp(1.f + fast_exp(-x)); }

// Hide a wave-uniform pointer from the optimiser so that per-step address arithmetic
// stays inside its step instead of being precomputed (and spilled) for a whole chunk.
template <typename T>
__device__ __forceinline__ T *opaque(T *p) {
  asm volatile("" : "+s"(p));
  return p;
}

// wave-uniform value -> SGPR
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// DPP helpers (row = 16 lanes).  ctrl: quad_perm 0x00-0xFF, row_ror:n = 0x120+n.
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Sum `v` over the 16 lanes of each row; every lane of the row ends with the sum.
__device__ __forceinline__ float row16_allsum(float v) {
  v += dpp<0x128>(v);  // row_ror:8
  v += dpp<0x124>(v);  // row_ror:4
  v += dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp<0xB1>(v);   // quad_perm [1,0,3,2]
  return v;
}

// Half / row exchanges.  hipcc (ROCm 7.2) miscompiles the two-result builtins
// __builtin_amdgcn_permlane{32,16}_swap when both results feed one expression (it reuses
// result 0 for result 1: `v_permlane32_swap v3, v7; v_add_f32 v3, v3, v3`), so the
// instruction is emitted directly.  The s_nop 1 covers the "VALU write -> permlane
// swap read" hazard (2 wait states), which hipcc does not pad inside asm.
//   swap32(a, b): lanes 32-63 of a  <->  lanes 0-31 of b
//   swap16(a, b): odd 16-lane rows of a  <->  even rows of b
__device__ __forceinline__ void swap32(float &a, float &b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap16(float &a, float &b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

}  // namespace cum
