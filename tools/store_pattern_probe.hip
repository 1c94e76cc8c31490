// Probe: HBM write rate of the GEMM epilogue's store pattern (8 B per lane, 16 rows x 32 B per wave-instruction)
// against row-contiguous 16-B stores, on a [M][N] bf16 output.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// one wave per 64 x 64 tile, MFMA-layout stores: lane (g, r) -> row 16 mi + r, cols 16 ni + 4 g
__global__ __launch_bounds__(256) void pattern_mfma(__bf16 *out, int64_t M, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, r = lane & 15;
  const int64_t tiles_n = N / 64;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  const int64_t m0 = (tile / tiles_n) * 64, n0 = (tile % tiles_n) * 64;
  if (m0 >= M) return;
  bf16x4 v = {(__bf16)1.f, (__bf16)2.f, (__bf16)3.f, (__bf16)lane};
  for (int mi = 0; mi < 4; ++mi)
    for (int ni = 0; ni < 4; ++ni)
      *reinterpret_cast<bf16x4 *>(out + (m0 + mi * 16 + r) * N + n0 + ni * 16 + 4 * g) = v;
}
// same tile, row-contiguous: 8 lanes x 16 B = one 128-B row segment, 8 rows per instruction
__global__ __launch_bounds__(256) void pattern_rows(__bf16 *out, int64_t M, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tiles_n = N / 64;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  const int64_t m0 = (tile / tiles_n) * 64, n0 = (tile % tiles_n) * 64;
  if (m0 >= M) return;
  bf16x8 v = {(__bf16)1.f, (__bf16)2.f, (__bf16)3.f, (__bf16)lane, (__bf16)1.f, (__bf16)2.f, (__bf16)3.f, (__bf16)4.f};
  for (int i = 0; i < 8; ++i)
    *reinterpret_cast<bf16x8 *>(out + (m0 + i * 8 + (lane >> 3)) * N + n0 + (lane & 7) * 8) = v;
}
int main() {
  const int64_t M = 1282048;
  for (int N : {64, 128, 256, 512}) {
    __bf16 *d;
    hipMalloc(&d, M * N * 2);
    const int64_t tiles = (M / 64) * (N / 64);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int k = 0; k < 2; ++k) {
      float ms;
      for (int w = 0; w < 2; ++w) {
        if (k == 0) pattern_mfma<<<(tiles + 3) / 4, 256>>>(d, M, N); else pattern_rows<<<(tiles + 3) / 4, 256>>>(d, M, N);
      }
      hipEventRecord(a);
      for (int it = 0; it < 10; ++it) {
        if (k == 0) pattern_mfma<<<(tiles + 3) / 4, 256>>>(d, M, N); else pattern_rows<<<(tiles + 3) / 4, 256>>>(d, M, N);
      }
      hipEventRecord(b); hipEventSynchronize(b);
      hipEventElapsedTime(&ms, a, b);
      printf("N=%4d %s: %.1f us  %.2f TB/s\n", N, k == 0 ? "mfma-layout 8B stores " : "row-contiguous 16B   ", ms * 100, M * N * 2 / (ms / 10 * 1e-3) / 1e12);
    }
    hipFree(d);
  }
  return 0;
}
