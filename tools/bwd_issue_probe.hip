// Diagnostic (not part of the library): the issue roof of the d_state > 16 BACKWARD selective scan, measured.
//
// Every wave runs, on registers only (no memory, no LDS, no barrier), the bare arithmetic of one 8-step half of
// scan_bwd_kernel (csrc/scan_bwd.hip) for its 8 states x 64 channels:
//   * 7 recomputed forward steps: per state pair v_pk_mul (delta' A'), 2 v_exp_f32, v_pk_mul (delta' u B), v_pk_fma (state);
//   * 8 reverse steps: per pair v_pk_fma (dx = C dy + carry), v_pk_mul (carry = a dx), v_pk_mul (g = carry x_{t-1}),
//     v_pk_fma (dA), v_pk_fma (p1 += g A'), v_pk_fma (p2 += dx B), the decay factor again for the half's last step only
//     (the floor: 7/8 + 1/8 = 1 v_exp_f32 per state update, every other factor taken as parked; the kernel parks 5 of 8: 1.25), 2 + 2 scalar products
//     (dB, dC contributions) and, per step, the 16 DPP adds + 2 pair sums of the xor-scatter reduce (xor_reduce16) and
//     1.5 exchange + add pairs of the row-level reduce;
// NW waves per SIMD on every CU for a few milliseconds.  state updates / s of this loop is the issue roof the backward
// rows of bench.py are priced against (as tools/clock_probe.hip is for the forward): what is left between it and the
// kernel is LDS traffic (operands, parked factors, partial sums), phase A / C, barriers and loads.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/build/bwd_issue_probe tools/bwd_issue_probe.hip
// run:   tools/build/bwd_issue_probe [waves_per_simd=2] [iters=4000]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 exp2_2(f2 v) { return f2{__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y)}; }

// the reduce of one reverse step: 16 DPP adds + 2 pair sums, as csrc/scan_bwd.hip::xor_reduce16 (products formed outside)
__device__ __forceinline__ float reduce16(float (&b)[8], float (&c)[8], float p1x, float p1y, float p2x, float p2y,
                                          float &q0, float &q1) {
  float X;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %3, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %5, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %7, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %9, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %10, %11, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %12, %13, %12 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %14, %15, %14 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %2, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %6, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %10, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %12, %14, %12 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %4, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32 %17, %19, %20\n\t"
      "v_add_f32_dpp %8, %12, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32 %18, %21, %22\n\t"
      "v_add_f32_dpp %16, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %16, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xc"
      : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(c[0]),
        "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "=&v"(X), "=&v"(q0), "=&v"(q1)
      : "v"(p1x), "v"(p1y), "v"(p2x), "v"(p2y));
  return X;
}
__device__ __forceinline__ float swap_add32(float a, float b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float swap_add16(float a, float b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

__global__ __launch_bounds__(512) void probe(float *sink, unsigned long long *stamps, int iters, float seed) {
  const int lane = threadIdx.x & 63;
  f2 Ap[4], x[4], bv[4], cv[4], dxc[4], dA[4], xs[8][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    Ap[j] = f2{-0.01f * (2 * j + 1) - seed, -0.01f * (2 * j + 2) - seed};
    x[j] = f2{0.1f, 0.2f};
    bv[j] = f2{0.5f + 0.1f * j + seed * lane, 0.25f + 0.01f * j};
    cv[j] = f2{1.f + 0.1f * j, -1.f + seed + 0.01f * j};
    dxc[j] = f2{0.f, 0.f};
    dA[j] = f2{0.f, 0.f};
  }
  float dt = 0.1f + 1e-3f * lane, du = 0.3f, dy = 0.7f, acc = 0.f;
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    f2 a7[4];
#pragma unroll
    for (int s = 0; s < 8; ++s) {                 // states before every step; 7 recomputed steps
#pragma unroll
      for (int j = 0; j < 4; ++j) xs[s][j] = x[j];
      if (s < 7) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f2 a = exp2_2(dt * Ap[j]);
          x[j] = a * x[j] + du * bv[j];
          if (s == 6) a7[j] = a;                   // (stands for a factor read back from LDS)
        }
        dt += 1e-7f;
      }
    }
    float Xo = 0.f, Yo = 0.f;
#pragma unroll
    for (int s = 7; s >= 0; --s) {
      f2 p1, p2;
      float rb[8], rc[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f2 a = s == 7 ? exp2_2(dt * Ap[j]) : a7[j];
        const f2 xt = s < 7 ? xs[s + 1][j] : a * xs[s][j] + du * bv[j];
        const f2 dx = cv[j] * dy + dxc[j];
        rb[2 * j] = dx.x * du; rb[2 * j + 1] = dx.y * du;
        rc[2 * j] = dy * xt.x; rc[2 * j + 1] = dy * xt.y;
        dxc[j] = a * dx;
        const f2 gg = dxc[j] * xs[s][j];
        dA[j] = gg * dt + dA[j];
        if (j == 0) {
          p1 = gg * Ap[j];
          p2 = dx * bv[j];
        } else {
          p1 = gg * Ap[j] + p1;
          p2 = dx * bv[j] + p2;
        }
      }
      float q0, q1;
      const float X = reduce16(rb, rc, p1.x, p1.y, p2.x, p2.y, q0, q1);
      acc += q0 + q1;
      if (s & 1) {
        Xo = X;
      } else {
        const float Y = swap_add32(Xo, X);
        if (s & 2) Yo = Y;
        else acc += swap_add16(Yo, Y);
      }
      dy += 1e-7f;
      du -= 1e-7f;
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    const int w = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    stamps[2 * w] = c1 - c0;
    stamps[2 * w + 1] = r1 - r0;
  }
  if (acc + dA[0].x + dA[1].y + dA[2].x + dA[3].y + x[0].x + dxc[1].y == 12345.f) sink[0] = acc;
}

int main(int argc, char **argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 2;
  const int iters = argc > 2 ? atoi(argv[2]) : 4000;
  const int cus = 256, waves_per_block = 8;
  const int blocks = cus * wps * 4 / waves_per_block;
  float *sink;
  unsigned long long *stamps;
  (void)hipMalloc(&sink, 4);
  (void)hipMalloc(&stamps, sizeof(unsigned long long) * 2 * blocks * waves_per_block);
  for (int warm = 0; warm < 3; ++warm) hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, sink, stamps, iters, 0.f);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, sink, stamps, iters, 0.f);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * blocks * waves_per_block);
  (void)hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (size_t w = 0; w < h.size() / 2; ++w)
    if (h[2 * w + 1]) clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1);
  std::sort(clk.begin(), clk.end());
  const double updates = (double)blocks * 512 * 8.0 * 8.0 * iters;   // 8 states x 8 steps per lane and iteration
  printf("{\"variant\": \"backward half: 7 recomputed + 8 reverse steps + xor-scatter reduce, registers only\", "
         "\"waves_per_simd\": %d, \"iters\": %d, \"kernel_ms\": %.3f, \"clock_GHz_median\": %.3f, "
         "\"state_updates_T_per_s\": %.3f}\n",
         wps, iters, ms, clk[clk.size() / 2], updates / (ms * 1e-3) / 1e12);
  return 0;
}
