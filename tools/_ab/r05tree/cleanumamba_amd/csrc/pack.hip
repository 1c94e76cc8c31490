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

// ---------------------------------------------------------------- index-free re-pack of SEPARABLE layouts
// Every weight layout of the conv stack and of the projections is a 2-D matrix whose source offset separates:
//   dst[r][c] = src[rowoff[r] + coloff[c]]      (either table entry = PACK_PAD: zero padding)
// (permutations of (h, c, tap) with zero-padded channel counts, GLU row interleaves, transposes).  Two small tables per
// operand replace the 4-byte-per-element index of gather_kernel, and where consecutive ROWS of the destination are the
// near neighbours in the source (the data-gradient layouts: transposes) the 64 x 64 tile goes through LDS so that both
// the reads and the writes are coalesced.  One launch packs every operand of a group: workgroup -> (job, tile) from a
// tile list built once per model.
constexpr int PACK_PAD = -2147483647 - 1;
struct PackJob {
  int64_t dst_off;      // elements from the start of the group's buffer
  int32_t rows, cols;   // destination matrix (cols: multiple of 8)
  int32_t row_tab, col_tab;   // positions of the two tables in `tables`
  int32_t transpose;    // 1: source is fast along destination rows
  int32_t runs8;        // 1: every aligned group of 8 destination columns is 8 consecutive source elements (or all padding)
                        // 2: ... and every such run starts on a 16-byte boundary of the source (two float4 loads)
};

template <typename TD>
__global__ __launch_bounds__(256) void pack2d_kernel(const float *__restrict__ src, const PackJob *__restrict__ jobs,
                                                    const int32_t *__restrict__ tiles, const int32_t *__restrict__ tables,
                                                    TD *__restrict__ dst) {
  typedef typename Vec4<TD>::type V;
  __shared__ __attribute__((aligned(16))) TD tile[64][72];
  const int tid = threadIdx.x;
  const PackJob j = jobs[tiles[3 * blockIdx.x]];
  const int r0 = tiles[3 * blockIdx.x + 1] * 64, c0 = tiles[3 * blockIdx.x + 2] * 64;
  const int32_t *rt = tables + j.row_tab, *ct = tables + j.col_tab;
  TD *out = dst + j.dst_off;
  if (j.transpose) {
    // read phase: lanes along destination rows (near neighbours in the source), wave w takes columns 16 w .. 16 w + 15
    const int r = r0 + (tid & 63), cw = c0 + 16 * (tid >> 6);
    const int ro = r < j.rows ? rt[r] : PACK_PAD;
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
      const int c = cw + cc;
      const int co = c < j.cols ? ct[c] : PACK_PAD;
      const float v = (ro == PACK_PAD || co == PACK_PAD) ? 0.f : src[(int64_t)ro + co];
      tile[tid & 63][16 * (tid >> 6) + cc] = (TD)v;
    }
    __syncthreads();
  }
  // write phase: a thread owns 8 consecutive columns of rows (tid / 8) and (tid / 8) + 32
  const int cg = 8 * (tid & 7);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int rl = (tid >> 3) + 32 * h, r = r0 + rl, c = c0 + cg;
    if (r >= j.rows || c >= j.cols) continue;
    if (j.transpose) {
      *reinterpret_cast<uint4 *>(out + (int64_t)r * j.cols + c) = *reinterpret_cast<const uint4 *>(&tile[rl][cg]);
      if constexpr (sizeof(TD) == 4)
        *reinterpret_cast<uint4 *>(out + (int64_t)r * j.cols + c + 4) = *reinterpret_cast<const uint4 *>(&tile[rl][cg + 4]);
    } else {
      const int ro = rt[r];
      float v[8];
      if (j.runs8) {          // one table entry and a contiguous run instead of eight entries and eight gathers
        const int co = ct[c];
        const bool pad = ro == PACK_PAD || co == PACK_PAD;
        const float *q = src + (pad ? 0 : (int64_t)ro + co);
        if (j.runs8 == 2) {
          const float4 a = pad ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4 *>(q);
          const float4 b = pad ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4 *>(q + 4);
          v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = pad ? 0.f : q[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int co = ct[c + i];
          v[i] = (ro == PACK_PAD || co == PACK_PAD) ? 0.f : src[(int64_t)ro + co];
        }
      }
      *reinterpret_cast<V *>(out + (int64_t)r * j.cols + c) = Vec4<TD>::make(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<V *>(out + (int64_t)r * j.cols + c + 4) = Vec4<TD>::make(v[4], v[5], v[6], v[7]);
    }
  }
}

}  // namespace cum

using namespace cum;

extern "C" int cum_pack2d(const float *src, const void *jobs, const int32_t *tiles, int32_t n_tiles, const int32_t *tables,
                          int32_t dst_dtype, void *dst, void *stream) {
  CUM_REQUIRE(dtype_ok(dst_dtype), "pack2d: dst dtype must be CUM_F32 / CUM_BF16 / CUM_F16");
  CUM_REQUIRE(n_tiles >= 0, "pack2d: negative tile count");
  if (n_tiles == 0) return CUM_OK;
  CUM_REQUIRE(src && jobs && tiles && tables && dst && ((uintptr_t)dst & 15) == 0 && ((uintptr_t)src & 15) == 0,
              "pack2d: null or misaligned pointer");
  hipStream_t st = (hipStream_t)stream;
  const PackJob *pj = static_cast<const PackJob *>(jobs);
  if (dst_dtype == CUM_F16)
    hipLaunchKernelGGL(pack2d_kernel<f16>, dim3(n_tiles), dim3(256), 0, st, src, pj, tiles, tables, (f16 *)dst);
  else if (dst_dtype == CUM_BF16)
    hipLaunchKernelGGL(pack2d_kernel<__bf16>, dim3(n_tiles), dim3(256), 0, st, src, pj, tiles, tables, (__bf16 *)dst);
  else
    hipLaunchKernelGGL(pack2d_kernel<float>, dim3(n_tiles), dim3(256), 0, st, src, pj, tiles, tables, (float *)dst);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

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
