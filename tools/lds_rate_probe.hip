// Diagnostic (not part of the library): LDS read rate per CU by access width.  One workgroup of NW waves per CU, every
// wave issuing ITER x 8 independent reads of its own 1 KB (b128) / 512 B (b64) / 256 B (b32) slices, conflict-free.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/build/lds_rate_probe tools/lds_rate_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int W>   // bytes per lane: 4, 8, 16
__global__ __launch_bounds__(512) void probe(float *sink, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
  for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) reinterpret_cast<float *>(lds)[i] = i;
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
  const unsigned a = base + (threadIdx.x & 63) * W + (threadIdx.x >> 6) * 4096;
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  u4 acc4 = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    if constexpr (W == 16) {
      u4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(v[k]) : "v"(a + 1024 * (k & 3)));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
      for (int k = 0; k < 8; ++k) acc4 ^= v[k];
    } else if constexpr (W == 8) {
      u2 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("ds_read_b64 %0, %1 offset:0" : "=v"(v[k]) : "v"(a + 512 * (k & 7)));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
      for (int k = 0; k < 8; ++k) { acc4.x ^= v[k].x; acc4.y ^= v[k].y; }
    } else {
      unsigned v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("ds_read_b32 %0, %1 offset:0" : "=v"(v[k]) : "v"(a + 256 * (k & 7)));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
      for (int k = 0; k < 8; ++k) acc4.x ^= v[k];
    }
  }
  if ((acc4.x ^ acc4.y ^ acc4.z ^ acc4.w) == 0x12345678u) sink[0] = 1.f;
}

template <int W>
static void run(int nw) {
  float *sink;
  (void)hipMalloc(&sink, 4);
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(probe<W>, dim3(256), dim3(64 * nw), 0, 0, sink, iters);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(probe<W>, dim3(256), dim3(64 * nw), 0, 0, sink, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)nw * iters * 8 * 64 * W;          // per CU
  printf("ds_read_b%-3d %d waves per CU: %.3f ms, %.1f bytes per clock and CU (at 2.35 GHz)\n", 8 * W, nw, ms, bytes / (ms * 1e-3 * 2.35e9));
}

int main() {
  for (int nw : {4, 8}) { run<16>(nw); run<8>(nw); run<4>(nw); }
  return 0;
}
