// Gather used to re-pack weights into the GEMM operand layouts (and weight gradients back out of them):
//   dst[i] = idx[i] < 0 ? 0 : convert(src[idx[i]])
// The layouts are pure index permutations with zero padding (network/convstack.py lay_*), rebuilt from the fp32
// master weights every step; a 32-bit index and the dtype conversion in the same pass halve the bytes the
// equivalent cast + 64-bit index_select moved.
#include "common.h"

namespace cum {

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void gather_kernel(const TS *__restrict__ src, const int32_t *__restrict__ idx,
                                                     TD *__restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += stride) {
    if (i0 + 3 < n) {
      const int4 j = *reinterpret_cast<const int4 *>(idx + i0);
      const float v0 = j.x < 0 ? 0.f : (float)src[j.x], v1 = j.y < 0 ? 0.f : (float)src[j.y];
      const float v2 = j.z < 0 ? 0.f : (float)src[j.z], v3 = j.w < 0 ? 0.f : (float)src[j.w];
      dst[i0] = (TD)v0; dst[i0 + 1] = (TD)v1; dst[i0 + 2] = (TD)v2; dst[i0 + 3] = (TD)v3;
    } else {
      for (int64_t i = i0; i < n; ++i) {
        const int32_t j = idx[i];
        dst[i] = (TD)(j < 0 ? 0.f : (float)src[j]);
      }
    }
  }
}

}  // namespace cum

using namespace cum;

extern "C" int cum_gather(int32_t src_dtype, const void *src, const int32_t *idx, int64_t n, int32_t dst_dtype,
                          void *dst, void *stream) {
  CUM_REQUIRE((src_dtype == CUM_F32 || src_dtype == CUM_BF16) && (dst_dtype == CUM_F32 || dst_dtype == CUM_BF16),
              "gather: dtypes must be CUM_F32 or CUM_BF16");
  CUM_REQUIRE(n >= 0, "gather: negative length");
  if (n == 0) return CUM_OK;
  CUM_REQUIRE(src && idx && dst && ((uintptr_t)idx & 15) == 0, "gather: null or misaligned pointer");
  const int64_t want = (n + 1023) / 1024;
  dim3 grid((unsigned)(want < 4096 ? want : 4096)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (src_dtype == CUM_F32 && dst_dtype == CUM_F32)
    hipLaunchKernelGGL((gather_kernel<float, float>), grid, block, 0, st, (const float *)src, idx, (float *)dst, n);
  else if (src_dtype == CUM_F32)
    hipLaunchKernelGGL((gather_kernel<float, __bf16>), grid, block, 0, st, (const float *)src, idx, (__bf16 *)dst, n);
  else if (dst_dtype == CUM_F32)
    hipLaunchKernelGGL((gather_kernel<__bf16, float>), grid, block, 0, st, (const __bf16 *)src, idx, (float *)dst, n);
  else
    hipLaunchKernelGGL((gather_kernel<__bf16, __bf16>), grid, block, 0, st, (const __bf16 *)src, idx, (__bf16 *)dst, n);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}
