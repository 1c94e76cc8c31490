// Streaming decoder glue (reference: CleanUMamba._denoise_frame, src/network/CleanUMamba.py:476-488, with the skip
// order fixed as SURVEY fact 9 describes).  Per hop and decoder layer the transposed conv of a frame yields 2L + 2
// time steps per stream; the first `stride` = 2 overlap the previous frame's tail, the last 2 are this frame's tail
// for the next hop.  One kernel (one launch) does what the reference spells as slice / add / cat / clone / relu / add-skip:
//   out[s][t]  = act( y[s][t] + (t < 2 ? tail[s][t] : 0) ) + skip[s][t]          t < 2L
//   tail[s][t] = y[s][2L + t] - bias                                               t < 2   (bias re-added next hop)
// on channels-last row buffers (csrc/gemm.hip layout), S streams in lock-step.
#include "common.h"

namespace cum {

template <typename T>
__global__ __launch_bounds__(256) void stream_overlap_add_kernel(const T *__restrict__ y, int64_t y_pitch,
                                                                 T *__restrict__ tail, const float *__restrict__ bias,
                                                                 const T *__restrict__ skip, int64_t skip_pitch,
                                                                 T *__restrict__ out, int64_t out_pitch, int streams,
                                                                 int L2, int Cp, int C, int relu) {
  const int64_t total = (int64_t)streams * (L2 + 2) * Cp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = i % Cp;
    const int64_t r = i / Cp;
    const int t = r % (L2 + 2);
    const int64_t s = r / (L2 + 2);
    const float v = (float)y[(s * y_pitch + t) * Cp + c];
    if (t < L2) {
      float o = v;
      if (t < 2) {
        // the thread that consumes the old tail element also replaces it (rows L2, L2 + 1 of this frame, bias taken
        // out because the next frame's transposed conv adds it again): no other thread touches tail[s][t][c]
        o += (float)tail[(s * 2 + t) * Cp + c];
        const float b = (bias && c < C) ? bias[c] : 0.f;
        tail[(s * 2 + t) * Cp + c] = (T)(c < C ? (float)y[(s * y_pitch + L2 + t) * Cp + c] - b : 0.f);
      }
      if (relu) o = fmaxf(o, 0.f);
      if (skip) o += (float)skip[(s * skip_pitch + t) * Cp + c];
      out[(s * out_pitch + t) * Cp + c] = (T)(c < C ? o : 0.f);
    }
  }
}

// Per-layer activation window of the streaming encoder: drop the n oldest rows of every stream, append the n newest
// rows of `fresh` (the layer recomputed over the whole window).  Older rows keep the values they got in the hop that
// first produced them, exactly like the reference's per-layer caches (src/network/CleanUMamba.py:425-447).
// Out of place into `tmp`, then copied back by the second kernel (stream order makes the shift race-free).
template <typename T>
__global__ __launch_bounds__(256) void stream_window_shift_kernel(const T *__restrict__ window, const T *__restrict__ fresh,
                                                                  T *__restrict__ tmp, int64_t pitch, int64_t fresh_pitch,
                                                                  int fresh_row0, int streams, int rows, int n_new, int Cp) {
  const int64_t total = (int64_t)streams * rows * Cp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = i % Cp;
    const int64_t r = i / Cp;
    const int t = r % rows;
    const int64_t s = r / rows;
    tmp[i] = t < rows - n_new ? window[(s * pitch + t + n_new) * Cp + c] : fresh[(s * fresh_pitch + t - fresh_row0) * Cp + c];
  }
}
template <typename T>
__global__ __launch_bounds__(256) void stream_window_store_kernel(const T *__restrict__ tmp, T *__restrict__ window,
                                                                  int64_t pitch, int streams, int rows, int Cp) {
  const int64_t total = (int64_t)streams * rows * Cp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = i % Cp;
    const int64_t r = i / Cp;
    window[((r / rows) * pitch + r % rows) * Cp + c] = tmp[i];
  }
}

// The same update in ONE launch and in place: a workgroup owns (stream, chunk of CC 16-byte channel vectors), pulls the
// rows it keeps into registers (<= 8 vectors per thread), and only after its barrier writes them back n_new rows
// earlier and appends the fresh rows -- no other workgroup touches those addresses.  Everything moves as 16-byte
// vectors (Cp is a multiple of 8 elements).
constexpr int kShiftRegs = 8;
__global__ __launch_bounds__(256) void stream_window_inplace_kernel(uint4 *__restrict__ window, const uint4 *__restrict__ fresh,
                                                                    int64_t pitch, int64_t fresh_pitch, int fresh_row0,
                                                                    int rows, int n_new, int Cv, int CC,
                                                                    uint4 *__restrict__ tail_dst, int64_t tail_pitch) {
  const int nchunk = (Cv + CC - 1) / CC;
  const int64_t s = blockIdx.x / nchunk;
  const int c0 = (blockIdx.x % nchunk) * CC;
  const int cc = min(CC, Cv - c0);
  const int keep = rows - n_new, total = keep * cc;
  uint4 *w = window + s * pitch * Cv + c0;
  uint4 v[kShiftRegs];
#pragma unroll
  for (int k = 0; k < kShiftRegs; ++k) {
    const int e = threadIdx.x + k * 256;
    if (e < total) v[k] = w[(int64_t)(e / cc + n_new) * Cv + e % cc];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kShiftRegs; ++k) {
    const int e = threadIdx.x + k * 256;
    if (e < total) w[(int64_t)(e / cc) * Cv + e % cc] = v[k];
  }
  const uint4 *f = fresh + s * fresh_pitch * Cv + c0;
  // tail_dst: the n_new + 2 newest rows of the updated window as a compact clip buffer = the next layer's input of an
  // incremental hop (two carried rows + the new ones), written here instead of by a separate launch
  uint4 *td = tail_dst ? tail_dst + s * tail_pitch * Cv + c0 : nullptr;
  for (int e = threadIdx.x; e < n_new * cc; e += 256) {
    const int t = keep + e / cc;
    const uint4 v_new = f[(int64_t)(t - fresh_row0) * Cv + e % cc];
    w[(int64_t)t * Cv + e % cc] = v_new;
    if (td) td[(int64_t)(t - keep + 2) * Cv + e % cc] = v_new;
  }
  if (td) {
    // carried rows = rows keep - 2, keep - 1 of the updated window = old rows rows - 2, rows - 1 (still in registers)
#pragma unroll
    for (int k = 0; k < kShiftRegs; ++k) {
      const int e = threadIdx.x + k * 256;
      if (e < total && e / cc >= keep - 2) td[(int64_t)(e / cc - (keep - 2)) * Cv + e % cc] = v[k];
    }
  }
}

// newest rows of a window -> compact input of the next layer's incremental step
template <typename T>
__global__ __launch_bounds__(256) void stream_tail_rows_kernel(const T *__restrict__ src, int64_t src_pitch, int src_row0,
                                                               T *__restrict__ dst, int64_t dst_pitch, int streams,
                                                               int rows, int Cp) {
  const int64_t total = (int64_t)streams * rows * Cp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = i % Cp;
    const int64_t r = i / Cp;
    const int t = r % rows;
    const int64_t s = r / rows;
    dst[(s * dst_pitch + t) * Cp + c] = src[(s * src_pitch + src_row0 + t) * Cp + c];
  }
}

}  // namespace cum

using namespace cum;

extern "C" int cum_stream_window_update(int32_t dtype, int32_t streams, int32_t rows, int32_t n_new, int32_t Cp,
                                        void *window, const void *fresh, int64_t pitch, int64_t fresh_pitch,
                                        int32_t fresh_row0, void *tmp, void *tail_dst, int64_t tail_pitch,
                                        void *stream) {
  CUM_REQUIRE(dtype_ok(dtype), "stream_window_update: dtype must be CUM_F32, CUM_BF16 or CUM_F16");
  CUM_REQUIRE(streams >= 0 && rows > 0 && n_new > 0 && n_new <= rows && Cp > 0 && pitch >= rows,
              "stream_window_update: bad shape");
  CUM_REQUIRE(fresh_row0 >= 0 && fresh_row0 <= rows - n_new && fresh_pitch >= rows - fresh_row0,
              "stream_window_update: fresh rows out of range");
  if (streams == 0) return CUM_OK;
  CUM_REQUIRE(window && fresh, "stream_window_update: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int keep = rows - n_new;
  CUM_REQUIRE(!tail_dst || (keep >= 2 && tail_pitch >= n_new + 2), "stream_window_update: tail_dst needs two carried rows");
  const int esz = is16(dtype) ? 2 : 4;
  const bool vec_ok = (Cp * esz) % 16 == 0 && ((uintptr_t)window & 15) == 0 && ((uintptr_t)fresh & 15) == 0 &&
                      ((uintptr_t)tail_dst & 15) == 0;
  if (keep <= kShiftRegs * 256 && vec_ok) {   // in place, one launch: every workgroup's kept rows fit its registers
    const int Cv = Cp * esz / 16;
    int CC = keep > 0 ? (kShiftRegs * 256) / keep : Cv;
    CC = CC < Cv ? CC : Cv;
    const int nchunk = (Cv + CC - 1) / CC;
    hipLaunchKernelGGL(stream_window_inplace_kernel, dim3(streams * nchunk), dim3(256), 0, st, (uint4 *)window,
                       (const uint4 *)fresh, pitch, fresh_pitch, fresh_row0, rows, n_new, Cv, CC, (uint4 *)tail_dst,
                       tail_pitch);
    CUM_CHECK_LAUNCH();
    return CUM_OK;
  }
  CUM_REQUIRE(!tail_dst, "stream_window_update: tail_dst needs the in-place path (16-byte aligned rows, <= 2048 kept rows)");
  CUM_REQUIRE(tmp, "stream_window_update: this window needs the scratch buffer (tmp)");
  const int64_t total = (int64_t)streams * rows * Cp;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (is16(dtype)) {     // pure data movement: one 16-bit instantiation serves bf16 and f16
    hipLaunchKernelGGL(stream_window_shift_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, (const __bf16 *)window,
                       (const __bf16 *)fresh, (__bf16 *)tmp, pitch, fresh_pitch, fresh_row0, streams, rows, n_new, Cp);
    hipLaunchKernelGGL(stream_window_store_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, (const __bf16 *)tmp,
                       (__bf16 *)window, pitch, streams, rows, Cp);
  } else {
    hipLaunchKernelGGL(stream_window_shift_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float *)window,
                       (const float *)fresh, (float *)tmp, pitch, fresh_pitch, fresh_row0, streams, rows, n_new, Cp);
    hipLaunchKernelGGL(stream_window_store_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float *)tmp,
                       (float *)window, pitch, streams, rows, Cp);
  }
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

extern "C" int cum_stream_tail_rows(int32_t dtype, int32_t streams, int32_t rows, int32_t Cp, const void *src,
                                    int64_t src_pitch, int32_t src_row0, void *dst, int64_t dst_pitch, void *stream) {
  CUM_REQUIRE(dtype_ok(dtype), "stream_tail_rows: dtype must be CUM_F32, CUM_BF16 or CUM_F16");
  CUM_REQUIRE(streams >= 0 && rows > 0 && Cp > 0 && src_row0 >= 0 && src_pitch >= src_row0 + rows && dst_pitch >= rows,
              "stream_tail_rows: bad shape");
  if (streams == 0) return CUM_OK;
  CUM_REQUIRE(src && dst, "stream_tail_rows: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)streams * rows * Cp;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (is16(dtype))         // pure data movement
    hipLaunchKernelGGL(stream_tail_rows_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, (const __bf16 *)src, src_pitch,
                       src_row0, (__bf16 *)dst, dst_pitch, streams, rows, Cp);
  else
    hipLaunchKernelGGL(stream_tail_rows_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float *)src, src_pitch,
                       src_row0, (float *)dst, dst_pitch, streams, rows, Cp);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

extern "C" int cum_stream_overlap_add(int32_t dtype, int32_t streams, int32_t L2, int32_t Cp, int32_t C, const void *y,
                                      int64_t y_pitch, void *tail, const float *bias, const void *skip,
                                      int64_t skip_pitch, void *out, int64_t out_pitch, int32_t relu, void *stream) {
  CUM_REQUIRE(dtype_ok(dtype), "stream_overlap_add: dtype must be CUM_F32, CUM_BF16 or CUM_F16");
  CUM_REQUIRE(streams >= 0 && L2 >= 2 && Cp > 0 && C > 0 && C <= Cp, "stream_overlap_add: bad shape");
  if (streams == 0) return CUM_OK;
  CUM_REQUIRE(y && tail && out && y_pitch >= L2 + 2 && out_pitch >= L2, "stream_overlap_add: bad buffer");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)streams * (L2 + 2) * Cp;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == CUM_BF16) {
    hipLaunchKernelGGL(stream_overlap_add_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, (const __bf16 *)y, y_pitch,
                       (__bf16 *)tail, bias, (const __bf16 *)skip, skip_pitch, (__bf16 *)out, out_pitch, streams, L2, Cp, C,
                       relu);
  } else if (dtype == CUM_F16) {
    hipLaunchKernelGGL(stream_overlap_add_kernel<f16>, dim3(blocks), dim3(256), 0, st, (const f16 *)y, y_pitch,
                       (f16 *)tail, bias, (const f16 *)skip, skip_pitch, (f16 *)out, out_pitch, streams, L2, Cp, C,
                       relu);
  } else {
    hipLaunchKernelGGL(stream_overlap_add_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float *)y, y_pitch,
                       (float *)tail, bias, (const float *)skip, skip_pitch, (float *)out, out_pitch, streams, L2, Cp, C,
                       relu);
  }
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}
