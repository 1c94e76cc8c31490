// Diagnostic (not part of the library): the memory access pattern of the GLU-backward epilogue of gemm_nt8_kernel in
// isolation -- one 512-thread workgroup per CU (128 KB of LDS claimed), every wave reading three M x N bf16 tensors and
// writing one M x 2N tensor for a 256 x 256 tile at a time, no MFMA work.
//   variant A: the MFMA result layout (8 bytes per lane: 16 rows x 32 bytes per instruction), as the kernel does it
//   variant B: the same bytes row-contiguous (16 bytes per lane: 2 rows x 512 bytes per instruction)
//   variant C: each wave its own 128 x 64 sub-tile, 16 bytes per lane: loads 8 rows x 128 B, stores 4 rows x 256 B
// build: hipcc --offload-arch=gfx950 -O3 -o tools/build/epi_pattern_probe tools/epi_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef unsigned short u16;

template <int VAR>
__global__ __launch_bounds__(512) void probe(const u16 *ext, const u16 *b, const u16 *y, u16 *out, int M, int N, int ntiles) {
  __shared__ char pad[128 * 1024];
  if (threadIdx.x == 9999) pad[threadIdx.x] = 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int NB = N / 256;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int m0 = (t / NB) * 256, n0 = (t % NB) * 256;
    if constexpr (VAR == 0) {
      const int wr = wave >> 2, wc = wave & 3, g = lane >> 4, r = lane & 15;
      for (int s = 0; s < 8; ++s) {
        const int64_t m = m0 + 128 * wr + 16 * s + r;
        uint2 e[4], bb[4], yy[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int n = n0 + 64 * wc + 16 * ni + 4 * g;
          e[ni] = *reinterpret_cast<const uint2 *>(ext + m * N + n);
          bb[ni] = *reinterpret_cast<const uint2 *>(b + m * N + n);
          yy[ni] = *reinterpret_cast<const uint2 *>(y + m * N + n);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int64_t zc = 2 * (int64_t)(n0 + 64 * wc + 16 * ni) + 4 * g;
          uint2 o0 = {e[ni].x ^ bb[ni].x, e[ni].y + yy[ni].y}, o1 = {e[ni].x + yy[ni].x, bb[ni].y ^ yy[ni].y};
          *reinterpret_cast<uint2 *>(out + m * 2 * N + zc) = o0;
          *reinterpret_cast<uint2 *>(out + m * 2 * N + zc + 16) = o1;
        }
      }
    } else if constexpr (VAR == 2) {
      // variant C: the wave's own 128 x 64 sub-tile, row-contiguous: loads 8 rows x 128 B, stores 4 rows x 256 B per instruction
      const int wr = wave >> 2, wc = wave & 3;
      for (int s = 0; s < 8; ++s) {
        uint4 e[2], bb[2], yy[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int64_t m = m0 + 128 * wr + 16 * s + 8 * k + (lane >> 3);
          const int n = n0 + 64 * wc + 8 * (lane & 7);
          e[k] = *reinterpret_cast<const uint4 *>(ext + m * N + n);
          bb[k] = *reinterpret_cast<const uint4 *>(b + m * N + n);
          yy[k] = *reinterpret_cast<const uint4 *>(y + m * N + n);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int64_t m = m0 + 128 * wr + 16 * s + 4 * k + (lane >> 4);
          const int64_t zc = 2 * (int64_t)(n0 + 64 * wc) + 8 * (lane & 15);
          uint4 o0 = {e[k & 1].x ^ bb[k & 1].x, e[k & 1].y + yy[k & 1].y, e[k >> 1].z, bb[k >> 1].w};
          *reinterpret_cast<uint4 *>(out + m * 2 * N + zc) = o0;
        }
      }
    } else {
      // wave w: rows 32 w .. 32 w + 31 of the tile; one instruction = 2 rows x 256 columns (512 B each)
      for (int s = 0; s < 16; s += 4) {
        uint4 e[4], bb[4], yy[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int64_t m = m0 + 32 * wave + 2 * (s + k) + (lane >> 5);
          const int n = n0 + 8 * (lane & 31);
          e[k] = *reinterpret_cast<const uint4 *>(ext + m * N + n);
          bb[k] = *reinterpret_cast<const uint4 *>(b + m * N + n);
          yy[k] = *reinterpret_cast<const uint4 *>(y + m * N + n);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int64_t m = m0 + 32 * wave + 2 * (s + k) + (lane >> 5);
          const int64_t zc = 2 * (int64_t)n0 + 16 * (lane & 31);
          uint4 o0 = {e[k].x ^ bb[k].x, e[k].y + yy[k].y, e[k].z, bb[k].w}, o1 = {e[k].x + yy[k].x, bb[k].y ^ yy[k].y, yy[k].z, e[k].w};
          *reinterpret_cast<uint4 *>(out + m * 2 * N + zc) = o0;
          *reinterpret_cast<uint4 *>(out + m * 2 * N + zc + 8) = o1;
        }
      }
    }
  }
}

int main() {
  const int M = 80128 / 256 * 256, N = 768, ntiles = (M / 256) * (N / 256);
  u16 *ext, *b, *y, *out;
  (void)hipMalloc(&ext, (size_t)M * N * 2); (void)hipMalloc(&b, (size_t)M * N * 2); (void)hipMalloc(&y, (size_t)M * N * 2);
  (void)hipMalloc(&out, (size_t)M * N * 4);
  (void)hipMemset(ext, 1, (size_t)M * N * 2); (void)hipMemset(b, 2, (size_t)M * N * 2); (void)hipMemset(y, 3, (size_t)M * N * 2);
  for (int var = 0; var < 3; ++var)
    for (int grid : {256, 128, 939}) {
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      for (int it = 0; it < 3; ++it) {
        if (it == 2) (void)hipEventRecord(e0, 0);
        if (var == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 0, 0, ext, b, y, out, M, N, ntiles);
        else if (var == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(512), 0, 0, ext, b, y, out, M, N, ntiles);
        else hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(512), 0, 0, ext, b, y, out, M, N, ntiles);
      }
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)M * N * 2 * 5;
      printf("variant %c grid %4d: %.1f us for %d tiles = %.1f us per tile and CU-slot, %.2f TB/s\n", "ABC"[var], grid, ms * 1e3, ntiles,
             ms * 1e3 / ((ntiles + grid - 1) / grid), bytes / ms / 1e9);
    }
  return 0;
}
