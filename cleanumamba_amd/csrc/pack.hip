// Gather used to re-pack weights into the GEMM operand layouts (and weight gradients back out of them):
//   dst[i] = idx[i] < 0 ? 0 : convert(src[idx[i]])
// The layouts are pure index permutations with zero padding (network/convstack.py lay_*), rebuilt from the fp32
// master weights every step; a 32-bit index and the dtype conversion in the same pass halve the bytes the
// equivalent cast + 64-bit index_select moved.
#include "common.h"

namespace cum {

template <typename TD>
struct Vec4;
template <>
struct Vec4<float> {
  typedef float4 type;
  static __device__ __forceinline__ type make(float a, float b, float c, float d) { return make_float4(a, b, c, d); }
};
template <>
struct Vec4<__bf16> {
  typedef __attribute__((ext_vector_type(4))) __bf16 type;
  static __device__ __forceinline__ type make(float a, float b, float c, float d) {
    type v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
    return v;
  }
};

template <>
struct Vec4<f16> {
  typedef __attribute__((ext_vector_type(4))) _Float16 type;
  static __device__ __forceinline__ type make(float a, float b, float c, float d) {
    type v = {(f16)a, (f16)b, (f16)c, (f16)d};
    return v;
  }
};

// 8 elements per thread and sweep: two 16-byte index loads, eight independent gathers in flight, two vector stores
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void gather_kernel(const TS *__restrict__ src, const int32_t *__restrict__ idx,
                                                     TD *__restrict__ dst, int64_t n) {
  typedef typename Vec4<TD>::type V;
  const int64_t stride = (int64_t)gridDim.x * 256 * 8;
  for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i0 < n; i0 += stride) {
    if (i0 + 7 < n) {
      const int4 j = *reinterpret_cast<const int4 *>(idx + i0), k = *reinterpret_cast<const int4 *>(idx + i0 + 4);
      const float v0 = j.x < 0 ? 0.f : (float)src[j.x], v1 = j.y < 0 ? 0.f : (float)src[j.y];
      const float v2 = j.z < 0 ? 0.f : (float)src[j.z], v3 = j.w < 0 ? 0.f : (float)src[j.w];
      const float v4 = k.x < 0 ? 0.f : (float)src[k.x], v5 = k.y < 0 ? 0.f : (float)src[k.y];
      const float v6 = k.z < 0 ? 0.f : (float)src[k.z], v7 = k.w < 0 ? 0.f : (float)src[k.w];
      *reinterpret_cast<V *>(dst + i0) = Vec4<TD>::make(v0, v1, v2, v3);
      *reinterpret_cast<V *>(dst + i0 + 4) = Vec4<TD>::make(v4, v5, v6, v7);
    } else {
      for (int64_t i = i0; i < n; ++i) {
        const int32_t jj = idx[i];
        dst[i] = (TD)(jj < 0 ? 0.f : (float)src[jj]);
      }
    }
  }
}

}  // namespace cum

using namespace cum;

extern "C" int cum_gather(int32_t src_dtype, const void *src, const int32_t *idx, int64_t n, int32_t dst_dtype,
                          void *dst, void *stream) {
  CUM_REQUIRE(dtype_ok(src_dtype) && dtype_ok(dst_dtype) && !(is16(src_dtype) && is16(dst_dtype) && src_dtype != dst_dtype),
              "gather: dtypes must be CUM_F32 / CUM_BF16 / CUM_F16 (no bf16 <-> f16 conversion)");
  CUM_REQUIRE(n >= 0, "gather: negative length");
  if (n == 0) return CUM_OK;
  CUM_REQUIRE(src && idx && dst && ((uintptr_t)idx & 15) == 0 && ((uintptr_t)dst & 15) == 0,
              "gather: null or misaligned pointer");
  const int64_t want = (n + 2047) / 2048;
  dim3 grid((unsigned)(want < 8192 ? want : 8192)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (src_dtype == CUM_F32 && dst_dtype == CUM_F16)
    hipLaunchKernelGGL((gather_kernel<float, f16>), grid, block, 0, st, (const float *)src, idx, (f16 *)dst, n);
  else if (src_dtype == CUM_F16 && dst_dtype == CUM_F32)
    hipLaunchKernelGGL((gather_kernel<f16, float>), grid, block, 0, st, (const f16 *)src, idx, (float *)dst, n);
  else if (src_dtype == CUM_F16)
    hipLaunchKernelGGL((gather_kernel<f16, f16>), grid, block, 0, st, (const f16 *)src, idx, (f16 *)dst, n);
  else if (src_dtype == CUM_F32 && dst_dtype == CUM_F32)
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
