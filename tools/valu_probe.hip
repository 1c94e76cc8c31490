// Diagnostic (not part of the library): issue cost of the VALU instruction kinds the scan kernels are made of.
// Each variant runs a register-only loop of 32 independent instructions of one kind (8 accumulators x 4) per iteration
// on every SIMD of the chip at 1 / 2 / 4 waves per SIMD; cycles per instruction and SIMD = clock * time * SIMDs /
// (instructions issued).  build: hipcc --offload-arch=gfx950 -O3 -o tools/build/valu_probe tools/valu_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f2 __attribute__((ext_vector_type(2)));

// KIND 0: v_fma_f32   1: v_pk_fma_f32   2: v_exp_f32   3: v_fma_f32 + s_add (1:1)   4: v_pk_fma + v_exp 2:1 (scan mix)
// 5: v_add_f32 dpp row_ror   6: v_pk_mul_f32   7: v_fma + s_nop 0 (1:1)
template <int KIND>
__global__ __launch_bounds__(256) void probe(float *sink, int iters, float seed) {
  float a[8], m = 1.0001f + seed, c = 1e-9f + seed;
  f2 p[8], pm = {1.0001f + seed, 0.9999f + seed}, pc = {1e-9f, 2e-9f};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = 1.f + j + threadIdx.x * 1e-3f;
    p[j] = f2{a[j], a[j] + 0.5f};
  }
  int sacc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(m), "v"(c));
        if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(pm), "v"(pc));
        if constexpr (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[j]));
        if constexpr (KIND == 3) asm volatile("v_fma_f32 %0, %0, %2, %3\n\ts_add_i32 %1, %1, 1" : "+v"(a[j]), "+s"(sacc) : "v"(m), "v"(c));
        if constexpr (KIND == 4) {
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(pm), "v"(pc));
          if (j & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[j]));
        }
        if constexpr (KIND == 5) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a[j]));
        if constexpr (KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(pm));
        if constexpr (KIND == 7) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_nop 0" : "+v"(a[j]) : "v"(m), "v"(c));
      }
  }
  float s = sacc;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += a[j] + p[j].x + p[j].y;
  if (s == 12345.f) sink[0] = s;
}

template <int KIND>
static void run(const char *name, double per_iter) {
  float *sink;
  (void)hipMalloc(&sink, 4);
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int blocks = 256 * wps;   // 4 waves per block: one per SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, sink, iters, 0.f);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, sink, iters, 0.f);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // instructions per SIMD = wps * iters * per_iter; cycles at an assumed 2.35 GHz (clock_probe measures 2.30-2.37)
    const double cyc = ms * 1e-3 * 2.35e9 / (wps * (double)iters * per_iter);
    printf("%-34s waves/SIMD %d  %.3f ms  %.2f cycles per instruction and SIMD (at 2.35 GHz)\n", name, wps, ms, cyc);
  }
}

int main() {
  run<0>("v_fma_f32", 32);
  run<1>("v_pk_fma_f32", 32);
  run<6>("v_pk_mul_f32", 32);
  run<2>("v_exp_f32", 32);
  run<5>("v_add_f32 dpp row_ror", 32);
  run<3>("v_fma_f32 + s_add_i32 (per pair)", 32);
  run<7>("v_fma_f32 + s_nop 0 (per pair)", 32);
  run<4>("2 v_pk_fma + 1 v_exp (per triple)", 16);
  return 0;
}
