// Diagnostic (not part of the library): the issue roof of the selective-scan inner loop, measured.
//
// Every wave runs the bare instruction mix of scan_fwd_lds_kernel's state update on registers only -- per state PAIR
// v_pk_mul_f32 (delta * A'), 2 x v_exp_f32, v_pk_mul_f32 (delta*u * B), v_pk_fma_f32 (state), v_pk_fma_f32 (C * x) --
// with no memory traffic, NW waves per SIMD on every CU, for a few milliseconds.  Stamps around the loop give
//   clock  = d(s_memtime) / d(s_memrealtime) * 100 MHz      (MI355X_MICROARCH.md, DVFS give-back item 6)
//   rate   = state updates per clock and SIMD
// so that DESIGN.md 3.1 can quote the scan against a MEASURED issue roof instead of an assumed clock.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/build/clock_probe tools/clock_probe.hip
// run:   tools/build/clock_probe [waves_per_simd=6] [iters=20000]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <bool PACKED>
__global__ __launch_bounds__(512) void probe(float *sink, unsigned long long *stamps, int iters, float seed) {
  const int lane = threadIdx.x & 63;
  f2 Ap[4], x[4], bv[4], cv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    Ap[j] = f2{-0.01f * (2 * j + 1) - seed, -0.01f * (2 * j + 2) - seed};
    x[j] = f2{0.f, 0.f};
    bv[j] = f2{0.5f + 0.1f * j + seed * lane, 0.25f + 0.01f * j};     // distinct per state: no common subexpressions
    cv[j] = f2{1.f + 0.1f * j, -1.f + seed + 0.01f * j};
  }
  float dt = 0.1f + 1e-3f * lane, du = 0.3f;
  f2 y = {0.f, 0.f};
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (PACKED) {
        const f2 e = dt * Ap[j];
        f2 a;
        a.x = __builtin_amdgcn_exp2f(e.x);
        a.y = __builtin_amdgcn_exp2f(e.y);
        x[j] = a * x[j] + du * bv[j];
        y = cv[j] * x[j] + y;
      } else {
        const float ex = dt * Ap[j].x, ey = dt * Ap[j].y;
        const float ax = __builtin_amdgcn_exp2f(ex), ay = __builtin_amdgcn_exp2f(ey);
        x[j].x = fmaf(ax, x[j].x, du * bv[j].x);
        x[j].y = fmaf(ay, x[j].y, du * bv[j].y);
        y.x = fmaf(cv[j].x, x[j].x, y.x);
        y.y = fmaf(cv[j].y, x[j].y, y.y);
      }
    }
    dt += 1e-7f;     // delta and delta*u change every step in the real kernel: nothing may be hoisted out of the
    du -= 1e-7f;     // loop (two extra VALU ops per 8 updates)
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    const int w = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    stamps[2 * w] = c1 - c0;
    stamps[2 * w + 1] = r1 - r0;
  }
  if (y.x + y.y + x[0].x + x[1].y + x[2].x + x[3].y == 12345.f) sink[0] = y.x;
}

template <bool PACKED>
static void run(int wps, int iters, const char *name) {
  const int cus = 256, waves_per_block = 8;
  const int blocks = cus * wps * 4 / waves_per_block;
  float *sink;
  unsigned long long *stamps;
  (void)hipMalloc(&sink, 4);
  (void)hipMalloc(&stamps, sizeof(unsigned long long) * 2 * blocks * waves_per_block);
  for (int warm = 0; warm < 3; ++warm) hipLaunchKernelGGL(probe<PACKED>, dim3(blocks), dim3(512), 0, 0, sink, stamps, iters, 0.f);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(probe<PACKED>, dim3(blocks), dim3(512), 0, 0, sink, stamps, iters, 0.f);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * blocks * waves_per_block);
  (void)hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::vector<double> clk, cyc;
  for (size_t w = 0; w < h.size() / 2; ++w) {
    if (h[2 * w + 1] == 0) continue;
    clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1);   // GHz: memrealtime ticks at 100 MHz
    cyc.push_back((double)h[2 * w]);
  }
  std::sort(clk.begin(), clk.end());
  std::sort(cyc.begin(), cyc.end());
  const double updates = (double)blocks * 512 * 8.0 * iters;        // 8 state updates per lane and iteration
  const double med_clk = clk[clk.size() / 2], med_cyc = cyc[cyc.size() / 2];
  // wall-clock rate (kernel duration from HIP events, clock from the stamps); the waves' own stamped durations are
  // shorter than the kernel (workgroups of a 4-per-CU grid do not all start together) and are not used for the rate
  (void)med_cyc;
  const double per_clk_simd = updates / (ms * 1e-3 * med_clk * 1e9 * 1024.0);
  printf("{\"variant\": \"%s\", \"waves_per_simd\": %d, \"iters\": %d, \"kernel_ms\": %.3f, \"clock_GHz_median\": %.3f, "
         "\"clock_GHz_p10\": %.3f, \"clock_GHz_p90\": %.3f, \"updates_per_clk_per_simd\": %.3f, "
         "\"state_updates_T_per_s\": %.3f}\n",
         name, wps, iters, ms, med_clk, clk[clk.size() / 10], clk[clk.size() * 9 / 10], per_clk_simd,
         updates / (ms * 1e-3) / 1e12);
  (void)hipFree(sink);
  (void)hipFree(stamps);
}

int main(int argc, char **argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 6;
  const int iters = argc > 2 ? atoi(argv[2]) : 20000;
  run<true>(wps, iters, "packed (v_pk_mul/v_pk_fma + 2 v_exp_f32 per pair)");
  run<false>(wps, iters, "scalar (v_mul/v_fma + v_exp_f32 per state)");
  return 0;
}
