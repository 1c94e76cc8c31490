// Fused "convolution as GEMM" kernels for the encoder / decoder stack on gfx950.
//
// Replaces cuDNN/cuBLAS behind nn.Conv1d / nn.ConvTranspose1d + ReLU / GLU / skip-add in
// the reference's encoder and decoder layers (src/network/CleanUMamba.py:108-113, 121-130,
// 313-316; GLU src/network/layers.py:26-33).
//
// Layout idea (DESIGN.md "conv stack"): activations are channels-last [B, T+2, Cpad] with
// two zero rows closing every clip.  Then
//   * Conv1d(k=4, s=2) output row t reads input rows 2t..2t+3 = 4*C CONTIGUOUS elements
//     starting at row 2t: a GEMM whose A rows overlap (row stride 2*C < K = 4*C);
//   * ConvTranspose1d(k=4, s=2) output rows (2t, 2t+1) read input rows t-1, t = 2*C
//     contiguous elements: again an overlapping-row GEMM, N = 2*Cout;
//   * 1x1 convs are plain GEMMs;
// and every backward-data pass is one of the same forms with re-packed weights.  One NT
// GEMM kernel, out[m][n] = epi(sum_k A[m*lda + k] * W[n*ldw + k]), therefore covers every
// fused layer: epilogues bias / ReLU / GLU (+ residual add, + row masking that keeps the
// closing rows zero), pre-activation side output for the backward.
//
// MFMA mapping: 128x128 block tile, 4 waves (2x2) of 64x64, v_mfma_f32_16x16x32_bf16
// (bf16 in, f32 acc) or v_mfma_f32_16x16x4_f32 (exact f32 for the parity path).  The
// weight tile is the MFMA "A" operand and the activation tile the "B" operand, so a lane
// ends up with 4 CONSECUTIVE output channels of one row: 8-/16-byte stores, and the GLU
// pair (a_j, b_j) sits in the same lane of two adjacent 16-column tiles (weights are packed
// [16 a-rows | 16 b-rows] per 32 rows).  LDS tiles are [128 rows][8 x 16 B] with the 16-B
// chunk index XOR-swizzled by (row & 7).  Tiles go HBM -> LDS directly (global_load_lds_dwordx4,
// no staging VGPRs; the swizzle is applied to the per-lane SOURCE address because the LDS side of
// the DMA is linear), single-buffered: 32 KB of LDS and ~110 VGPRs per workgroup let 4 workgroups
// share a CU, and their interleaving hides the load latency (CDNA guide: the 128x128 "step-3"
// structure).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace cum {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

enum { EPI_BIAS = 0, EPI_RELU = 1, EPI_GLU = 2, EPI_MASK = 3, EPI_GLU_BWD = 4 };

struct GemmParams {
  const void *A, *W;
  const float *bias;       // [N] (padded), may be null
  const void *res;         // residual [M][ldr], may be null; added AFTER the activation
                           //   EPI_MASK: the ReLU output whose sign gates the result; EPI_GLU_BWD: added BEFORE the GLU backward
  void *out;               // [M][ldc]
  void *aux;               // GLU: pre-activation [M][ldz] (N columns); BIAS/RELU: activation before the residual add;
                           //   EPI_MASK: the ungated result (second output); EPI_GLU_BWD: the saved pre-activation Z (input)
  const void *aux2;        // gate_only GLU_BWD: the GLU output y saved by the forward [M][ldy]
  int64_t lda, ldw, ldc, ldr, ldz, ldy;
  int gate_only;           // GLU / GLU_BWD: aux holds only the gate pre-activation b ([M][ldz], output-column order)
  int allow_split_k;       // few-tile launches may use gemm_nt_splitk_kernel
  int mask_bits;           // RELU: aux receives the SIGN (value > 0) of each element instead of the activation, four
                           // consecutive channels per byte (low nibble; byte index (m * ld + n) / 4) -- what a lane
                           // holds after the MFMA, so no cross-lane packing; MASK: res is such an array.
  int M, N, K;             // N multiple of 16 (32 for GLU), K multiple of the K tile
  int pitch, valid;        // row m is real iff (m % pitch) < valid; other rows are stored as zeros
  int n_store;             // number of output columns to store (<= N, or N/2 for GLU); multiple of 4
  int64_t zero_head, zero_tail;
  int rows_epilogue;       // gemm_nt8_kernel: GLU_BWD epilogue through LDS (CUM_NT8_ROWS=0 turns it off for A/B runs)
  int group_m;             // gemm_nt9_kernel: m-tiles an XCD walks side by side (launch_gemm_nt9)
};

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int EPC = 4;  // elements per 16-byte chunk
  static __device__ __forceinline__ float to_f(float v) { return v; }
  static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <>
struct Elem<__bf16> {
  static constexpr int EPC = 8;
  static __device__ __forceinline__ float to_f(__bf16 v) { return (float)v; }
  static __device__ __forceinline__ __bf16 from_f(float v) { return (__bf16)v; }
};

template <>
struct Elem<f16> {
  static constexpr int EPC = 8;
  static __device__ __forceinline__ float to_f(f16 v) { return (float)v; }
  static __device__ __forceinline__ f16 from_f(float v) { return (f16)v; }
};

template <typename T>
__device__ __forceinline__ void store4(T *p, const float (&v)[4]);
template <>
__device__ __forceinline__ void store4<float>(float *p, const float (&v)[4]) {
  *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void store4<__bf16>(__bf16 *p, const float (&v)[4]) {
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
  bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  *reinterpret_cast<bf16x4 *>(p) = o;
}
template <>
__device__ __forceinline__ void store4<f16>(f16 *p, const float (&v)[4]) {
  typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
  f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
  *reinterpret_cast<f16x4 *>(p) = o;
}
template <typename T>
__device__ __forceinline__ void load4(const T *p, float (&v)[4]);
template <>
__device__ __forceinline__ void load4<float>(const float *p, float (&v)[4]) {
  const float4 t = *reinterpret_cast<const float4 *>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <>
__device__ __forceinline__ void load4<__bf16>(const __bf16 *p, float (&v)[4]) {
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
  const bf16x4 t = *reinterpret_cast<const bf16x4 *>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (float)t[i];
}

template <>
__device__ __forceinline__ void load4<f16>(const f16 *p, float (&v)[4]) {
  typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
  const f16x4 t = *reinterpret_cast<const f16x4 *>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (float)t[i];
}

// One K-step of a wave's 64x64 sub-tile from the staged tiles (ldsA / ldsW: [row * 8 + chunk], chunk index
// XOR-swizzled by row & 7).
template <typename T>
__device__ __forceinline__ void nt_compute(const uint4 *ldsA, const uint4 *ldsW, f32x4 (&acc)[4][4], int wm, int wn,
                                           int g, int r) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 wf[4], af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int wrow = wn * 64 + i * 16 + r;
        const int arow = wm * 64 + i * 16 + r;
        const int cl = ks * 4 + g;
        wf[i] = ldsW[wrow * 8 + (cl ^ (wrow & 7))];
        af[i] = ldsA[arow * 8 + (cl ^ (arow & 7))];
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          if constexpr (__is_same(T, f16))
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[ni]),
                                                                 __builtin_bit_cast(f16x8, af[mi]), acc[ni][mi], 0, 0, 0);
          else
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ni]),
                                                                  __builtin_bit_cast(bf16x8, af[mi]), acc[ni][mi], 0, 0, 0);
        }
    }
  } else {
    // f32: lane reads k = 8g .. 8g+7 (two 16-byte chunks) of its row; MFMA k-slot g at sub-step s of chunk h is
    // k = 8g + 4h + s for BOTH operands, so the dot product is a permutation of the same 32 products.  One chunk at
    // a time: 32 fragment registers beside the 64 accumulators (both chunks at once spilled 6-49 VGPRs at the
    // 128-register budget of four waves per SIMD).
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float wf[4][4], af[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int wrow = wn * 64 + i * 16 + r;
        const int arow = wm * 64 + i * 16 + r;
        const int cl = 2 * g + h;
        const uint4 wv = ldsW[wrow * 8 + (cl ^ (wrow & 7))];
        const uint4 av = ldsA[arow * 8 + (cl ^ (arow & 7))];
        wf[i][0] = __builtin_bit_cast(float, wv.x); wf[i][1] = __builtin_bit_cast(float, wv.y);
        wf[i][2] = __builtin_bit_cast(float, wv.z); wf[i][3] = __builtin_bit_cast(float, wv.w);
        af[i][0] = __builtin_bit_cast(float, av.x); af[i][1] = __builtin_bit_cast(float, av.y);
        af[i][2] = __builtin_bit_cast(float, av.z); af[i][3] = __builtin_bit_cast(float, av.w);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ni][s], af[mi][s], acc[ni][mi], 0, 0, 0);
    }
  }
}

// The 16 bias values a lane adds in the epilogue (columns n0 + 64 wn + 16 ni + 4 g + j), fetched with four 16-byte
// loads BEFORE the K loop: loaded inside the epilogue they were 64 dependent L2 round trips per lane, which made the
// epilogue -- not HBM -- the bound of every layer with few K steps.
__device__ __forceinline__ void nt_load_bias(const GemmParams &p, int n0, int wn, int g, float (&bv)[4][4]) {
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = n0 + wn * 64 + ni * 16 + 4 * g;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && n + 3 < p.N) t = *reinterpret_cast<const float4 *>(p.bias + n);
    bv[ni][0] = t.x; bv[ni][1] = t.y; bv[ni][2] = t.z; bv[ni][3] = t.w;
  }
}

// Raw 4-element vector of T: loads are issued first and converted only where they are consumed, so that the loads of
// a whole 16-row slab are in flight together.
template <typename T>
struct Raw4;
template <>
struct Raw4<float> {
  float4 v;
  __device__ __forceinline__ void ld(const float *p) { v = *reinterpret_cast<const float4 *>(p); }
  __device__ __forceinline__ void get(float (&o)[4]) const { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
};
template <>
struct Raw4<__bf16> {
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
  bf16x4 v;
  __device__ __forceinline__ void ld(const __bf16 *p) { v = *reinterpret_cast<const bf16x4 *>(p); }
  __device__ __forceinline__ void get(float (&o)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (float)v[i];
  }
};

template <>
struct Raw4<f16> {
  typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
  f16x4 v;
  __device__ __forceinline__ void ld(const f16 *p) { v = *reinterpret_cast<const f16x4 *>(p); }
  __device__ __forceinline__ void get(float (&o)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (float)v[i];
  }
};

// Epilogue of a wave's 64x64 sub-tile at (m0 + 64 wm, n0 + 64 wn).
//
// Per 16-row slab (mi) every global load the slab needs (residual / ReLU mask / gate / saved output) is issued from
// a clamped, always-valid address with no branch in between, and (where registers allow) the loads of slab mi+1 are
// issued BEFORE the stores of slab mi (vmcnt counts in order: waiting for those loads never waits for younger
// stores).  The
// straightforward form -- load, wait, compute, store per 16x16 tile inside per-lane `continue`s -- serialised 16
// (MASK) to 48 (GLU_BWD) memory round trips per tile.
// NH: 64-row halves of the wave's sub-tile (1, or 2 for gemm_nt8_kernel's 128 x 64); PIPE: slabs whose loads are
// issued ahead of the slab being finished (-1: the default of the 128-VGPR kernels, see the end of the function).
template <typename T, int EPI, int NH = 1, int PIPE = -1>
__device__ __forceinline__ void nt_epilogue(const GemmParams &p, const f32x4 (*accp)[4][4], const float (&bv)[4][4],
                                            int m0, int n0, int wm0, int wn, int g, int r) {
  // lane holds D[n = nb + 4g + j][m = mb + r], j = 0..3 -> 4 consecutive channels of row m.
  // No two of these buffers overlap.
  T *__restrict__ out = static_cast<T *>(p.out);
  T *__restrict__ aux = static_cast<T *>(p.aux);
  const T *__restrict__ res = static_cast<const T *>(p.res);
  const T *__restrict__ aux2 = static_cast<const T *>(p.aux2);
  const int nw0 = n0 + wn * 64;
  constexpr bool kMask = EPI == EPI_MASK;
  constexpr int NL = EPI == EPI_GLU ? 2 : 4;           // loads of one kind per slab
  struct Slab {
    Raw4<T> r[NL], a[EPI == EPI_GLU_BWD ? 4 : 1], b[EPI == EPI_GLU_BWD ? 4 : 1];
    unsigned char mw[4];       // MASK with mask_bits: this lane's sign nibbles of the slab's four tiles
  };
  // slab sl = 16 rows: half sl / 4 (64 rows each), 16-row tile sl % 4
  auto row_of = [&](int sl, bool &live) {
    const int m_raw = m0 + wm0 * 64 + sl * 16 + r;
    live = m_raw < p.M;
    return (int64_t)(live ? m_raw : p.M - 1);          // clamped row for the loads
  };
  auto issue = [&](int sl, Slab &s) {
    bool live;
    const int64_t m = row_of(sl, live);
    if constexpr (EPI == EPI_GLU) {
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const int oc = nw0 / 2 + pi * 16 + 4 * g;
        if (res) s.r[pi].ld(res + m * p.ldr + (oc < p.n_store ? oc : 0));
      }
    } else if constexpr (EPI == EPI_GLU_BWD) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = nw0 + ni * 16 + 4 * g;
        const bool nok = n < p.n_store;
        const int nc = nok ? n : 0;
        const int64_t zc = nok ? 2 * (int64_t)(nw0 + ni * 16) + 4 * g : 4 * g;
        if (res) s.r[ni].ld(res + m * p.ldr + nc);
        if (p.gate_only) {
          s.b[ni].ld(aux + m * p.ldz + nc);
          s.a[ni].ld(aux2 + m * p.ldy + nc);       // the saved output y = a * sig(b)
        } else {
          s.a[ni].ld(aux + m * p.ldz + zc);
          s.b[ni].ld(aux + m * p.ldz + zc + 16);
        }
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = nw0 + ni * 16 + 4 * g;
        if (kMask && p.mask_bits) {
          s.mw[ni] = reinterpret_cast<const unsigned char *>(p.res)[(m * p.ldr + (n < p.n_store ? n : 0)) >> 2];
        } else if (kMask || res) {
          s.r[ni].ld(res + m * p.ldr + (n < p.n_store ? n : 0));
        }
      }
    }
  };
  auto finish = [&](int sl, const Slab &s) {
    const f32x4 (&acc)[4][4] = accp[sl / 4];
    const int mi = sl % 4;
    bool live;
    const int64_t m = row_of(sl, live);
    const bool real = live && (m % p.pitch) < p.valid;
    if constexpr (EPI == EPI_GLU) {
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {  // tile pair (2*pi, 2*pi+1) = (a, b)
        const int na = nw0 + (2 * pi) * 16 + 4 * g;
        const int nb = na + 16;
        const int oc = nw0 / 2 + pi * 16 + 4 * g;
        float a[4], b[4], o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a[j] = acc[2 * pi][mi][j] + bv[2 * pi][j];
          b[j] = acc[2 * pi + 1][mi][j] + bv[2 * pi + 1][j];
          o[j] = real ? a[j] * sigmoidf_(b[j]) : 0.f;
        }
        if (aux && live && na < p.N) {
          if (p.gate_only) {
            store4<T>(aux + m * p.ldz + oc, b);
          } else {
            store4<T>(aux + m * p.ldz + na, a);
            store4<T>(aux + m * p.ldz + nb, b);
          }
        }
        if (res) {
          float rr[4];
          s.r[pi].get(rr);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = real ? o[j] + rr[j] : 0.f;
        }
        if (live && oc < p.n_store && na < p.N) store4<T>(out + m * p.ldc + oc, o);
      }
    } else if constexpr (EPI == EPI_GLU_BWD) {
      // d = acc (+ res) is the gradient of a GLU output; the 16-column tile t of row m pairs with columns
      // [32t, 32t+16) (a) and [32t+16, 32t+32) (b) of Z row m, and dZ is written in Z's layout.
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = nw0 + ni * 16 + 4 * g;
        const int64_t zc = 2 * (int64_t)(nw0 + ni * 16) + 4 * g;
        float d[4], a[4], b[4], da[4], db[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = acc[ni][mi][j];
        if (res) {
          float rr[4];
          s.r[ni].get(rr);
#pragma unroll
          for (int j = 0; j < 4; ++j) d[j] += rr[j];
        }
        s.a[ni].get(a);
        s.b[ni].get(b);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float sg = sigmoidf_(b[j]);
          const float dj = real ? d[j] : 0.f;
          da[j] = dj * sg;
          db[j] = p.gate_only ? dj * a[j] * (1.f - sg) : dj * a[j] * sg * (1.f - sg);
        }
        if (live && n < p.n_store) {
          store4<T>(out + m * p.ldc + zc, da);
          store4<T>(out + m * p.ldc + zc + 16, db);
        }
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = nw0 + ni * 16 + 4 * g;
        const bool on = live && n < p.n_store;
        float v[4], rr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[ni][mi][j] + bv[ni][j];
          if (EPI == EPI_RELU) v[j] = fmaxf(v[j], 0.f);
          v[j] = real ? v[j] : 0.f;
        }
        if (kMask && p.mask_bits) {
          const unsigned nib = s.mw[ni];
#pragma unroll
          for (int j = 0; j < 4; ++j) rr[j] = (nib >> j) & 1u ? 1.f : 0.f;
        } else if (kMask || res) {
          s.r[ni].get(rr);
        }
        if (aux) {
          if (!kMask && p.mask_bits) {
            unsigned w = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) w |= (v[j] > 0.f ? 1u : 0u) << j;
            if (on) reinterpret_cast<unsigned char *>(p.aux)[(m * p.ldz + n) >> 2] = (unsigned char)w;
          } else if (on) {
            store4<T>(aux + m * p.ldz + n, v);      // ungated / pre-residual value
          }
        }
        if constexpr (kMask) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = rr[j] > 0.f ? v[j] : 0.f;
        } else if (res) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = real ? v[j] + rr[j] : 0.f;
        }
        if (on) store4<T>(out + m * p.ldc + n, v);
      }
    }
  };
  constexpr int NSL = 4 * NH;
  // default depth: 12 loads per slab (GLU_BWD) or 16-byte f32 operands: two slabs in flight would spill at the 128-VGPR
  // budget of 4 waves per SIMD (the f32 instantiations spilled 6-49 VGPRs with the pipelined form); else one ahead
  constexpr int AHEAD = PIPE >= 0 ? PIPE : ((EPI == EPI_GLU_BWD || sizeof(T) == 4) ? 0 : 1);
  Slab ring[AHEAD + 1];
#pragma unroll
  for (int sl = 0; sl < AHEAD && sl < NSL; ++sl) issue(sl, ring[sl]);
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    if (sl + AHEAD < NSL) issue(sl + AHEAD, ring[(sl + AHEAD) % (AHEAD + 1)]);
    finish(sl, ring[sl % (AHEAD + 1)]);
  }
}

// GLU-backward epilogue of gemm_nt8_kernel through LDS (gate-only form, full-width tiles, 16-byte aligned rows).
//
// In the MFMA result layout a lane touches 8 bytes of 16 different rows per instruction: 16 rows x 32 bytes.  One CU
// sustains that pattern at 44 us per 256 x 256 tile of this epilogue however idle the rest of the chip is, against 19 us
// for the same bytes moved 16 bytes per lane along the rows (tools/epi_pattern_probe.hip; in-kernel time stamps put the
// epilogue at 44 us per tile beside a 55 us K loop).  After the K loop the 128 KB of LDS are free, so every wave
// transposes its own 128 x 64 sub-tile through a private 11 KB region, one 16-row slab at a time: the three operand
// slabs arrive with 16-byte row-contiguous loads (8 rows x 128 B per instruction, issued one slab ahead), are re-read in
// the MFMA layout (row stride 144 B: conflict-free 8-byte reads), and dZ leaves through a row-major staging slab (stride
// 272 B) as 4 rows x 256 B per store instruction.  Wave-private LDS traffic needs no barrier: the LDS executes a wave's
// instructions in order.
// NH: 64-row halves of the wave's sub-tile (gemm_nt8_kernel: 2, gemm_nt_kernel: 1); AHEAD: slabs whose operand loads are
// issued before the slab being finished (1 where registers allow, 0 in the 128-VGPR kernels).
template <typename T, int NH, int AHEAD>
__device__ __forceinline__ void nt_epilogue_glu_bwd_rows(const GemmParams &p, const f32x4 (*accp)[4][4], int mw0, int nw0,
                                                        int lane, unsigned char *lw) {
  constexpr int NSL = 4 * NH;
  constexpr int IS = 144, OS = 272;
  const T *__restrict__ gb = static_cast<const T *>(p.aux);
  // without a residual the gate is read in its place (valid memory) and weighted by zero: no branch around the loads
  const T *__restrict__ res = p.res ? static_cast<const T *>(p.res) : gb;
  const int64_t ldr = p.res ? p.ldr : p.ldz;
  const float ew = p.res ? 1.f : 0.f;
  const T *__restrict__ yy = static_cast<const T *>(p.aux2);
  T *__restrict__ out = static_cast<T *>(p.out);
  const int g = lane >> 4, r = lane & 15;
  unsigned char *const le = lw, *const lb = lw + 16 * IS, *const ly = lw + 32 * IS, *const lo = lw + 48 * IS;
  const int lrow = lane >> 3, lch = lane & 7;      // loads: 8 rows x 8 chunks of 16 bytes
  const int srow = lane >> 4, sch = lane & 15;     // stores: 4 rows x 16 chunks
  struct In {
    u32x4 e[2], b[2], y[2];
  };
  auto issue = [&](int sl, In &v) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int m_raw = mw0 + 16 * sl + 8 * k + lrow;
      const int64_t m = m_raw < p.M ? m_raw : p.M - 1;           // clamped row, never stored
      v.e[k] = *reinterpret_cast<const u32x4 *>(res + m * ldr + nw0 + 8 * lch);
      v.b[k] = *reinterpret_cast<const u32x4 *>(gb + m * p.ldz + nw0 + 8 * lch);
      v.y[k] = *reinterpret_cast<const u32x4 *>(yy + m * p.ldy + nw0 + 8 * lch);
    }
  };
  In ring[AHEAD + 1];
  if (AHEAD) issue(0, ring[0]);
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    if (!AHEAD) issue(sl, ring[0]);
    else if (sl + 1 < NSL) issue(sl + 1, ring[(sl + 1) & 1]);
    const In &cur = ring[AHEAD ? sl & 1 : 0];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int off = (8 * k + lrow) * IS + 16 * lch;
      *reinterpret_cast<u32x4 *>(le + off) = cur.e[k];
      *reinterpret_cast<u32x4 *>(lb + off) = cur.b[k];
      *reinterpret_cast<u32x4 *>(ly + off) = cur.y[k];
    }
    const f32x4 (&acc)[4][4] = accp[sl / 4];
    const int mi = sl % 4;
    const int m_raw = mw0 + 16 * sl + r;
    const bool real = m_raw < p.M && (m_raw % p.pitch) < p.valid;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int io = r * IS + 32 * ni + 8 * g;
      float e[4], b[4], a[4], da[4], db[4];
      load4<T>(reinterpret_cast<const T *>(le + io), e);
      load4<T>(reinterpret_cast<const T *>(lb + io), b);
      load4<T>(reinterpret_cast<const T *>(ly + io), a);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sg = sigmoidf_(b[j]);
        const float dj = real ? fmaf(ew, e[j], acc[ni][mi][j]) : 0.f;
        da[j] = dj * sg;
        db[j] = dj * a[j] * (1.f - sg);
      }
      store4<T>(reinterpret_cast<T *>(lo + r * OS + 64 * ni + 8 * g), da);
      store4<T>(reinterpret_cast<T *>(lo + r * OS + 64 * ni + 8 * g + 32), db);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = 4 * k + srow;
      const int ms = mw0 + 16 * sl + row;
      const u32x4 v = *reinterpret_cast<const u32x4 *>(lo + row * OS + 16 * sch);
      if (ms < p.M) *reinterpret_cast<u32x4 *>(out + (int64_t)ms * p.ldc + 2 * nw0 + 8 * sch) = v;
    }
  }
}

// The same transposition for the bias / ReLU / ReLU-gate epilogues (EPI_BIAS, EPI_RELU, EPI_MASK) of gemm_nt8_kernel:
// residual or gating activation in (T, or sign nibbles: one byte per four channels), result out, optional second
// output (the pre-residual / ungated value as T, or the sign nibbles of a ReLU).  Arithmetic and masking are those of
// nt_epilogue; only the route of the bytes differs.  The launcher-side conditions are in nt8_rows_ok().
template <typename T, int EPI, int NH, int AHEAD>
__device__ __forceinline__ void nt_epilogue_rows(const GemmParams &p, const f32x4 (*accp)[4][4], const float (&bv)[4][4],
                                                 int mw0, int nw0, int lane, unsigned char *lw) {
  constexpr int NSL = 4 * NH;
  constexpr int RS = 144;                          // LDS row stride of a 128-byte row segment
  constexpr bool kMask = EPI == EPI_MASK;
  const T *__restrict__ res = static_cast<const T *>(p.res);
  T *__restrict__ out = static_cast<T *>(p.out);
  T *__restrict__ aux = static_cast<T *>(p.aux);
  const int g = lane >> 4, r = lane & 15;
  unsigned char *const l_res = lw, *const l_out = lw + 16 * RS, *const l_aux = lw + 32 * RS;
  unsigned char *const l_bin = lw + 48 * RS, *const l_bout = l_bin + 16 * 16;     // sign nibbles in / out: [16 rows][16 bytes]
  const bool res_bits = kMask && p.mask_bits, res_t = !res_bits && (kMask || p.res != nullptr);
  const bool aux_bits = !kMask && p.mask_bits && p.aux != nullptr, aux_t = !aux_bits && p.aux != nullptr;
  const int lrow = lane >> 3, lch = lane & 7;      // 16-byte accesses: 8 rows x 128 B
  const int brow = lane >> 2, bch = lane & 3;      // nibble bytes: 16 rows x 16 B, 4 bytes per lane
  struct In {
    u32x4 v[2];
    unsigned bits;
  };
  auto issue = [&](int sl, In &in) {
    if (res_t) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int m_raw = mw0 + 16 * sl + 8 * k + lrow;
        const int64_t m = m_raw < p.M ? m_raw : p.M - 1;
        in.v[k] = *reinterpret_cast<const u32x4 *>(res + m * p.ldr + nw0 + 8 * lch);
      }
    }
    if (res_bits) {
      const int m_raw = mw0 + 16 * sl + brow;
      const int64_t m = m_raw < p.M ? m_raw : p.M - 1;
      in.bits = *reinterpret_cast<const unsigned *>(reinterpret_cast<const unsigned char *>(p.res) + ((m * p.ldr + nw0) >> 2) + 4 * bch);
    }
  };
  In ring[AHEAD + 1];
  if (AHEAD) issue(0, ring[0]);
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    if (!AHEAD) issue(sl, ring[0]);
    else if (sl + 1 < NSL) issue(sl + 1, ring[(sl + 1) & 1]);
    const In &cur = ring[AHEAD ? sl & 1 : 0];
    if (res_t) {
#pragma unroll
      for (int k = 0; k < 2; ++k) *reinterpret_cast<u32x4 *>(l_res + (8 * k + lrow) * RS + 16 * lch) = cur.v[k];
    }
    if (res_bits) *reinterpret_cast<unsigned *>(l_bin + 16 * brow + 4 * bch) = cur.bits;
    const f32x4 (&acc)[4][4] = accp[sl / 4];
    const int mi = sl % 4;
    const int m_raw = mw0 + 16 * sl + r;
    const bool real = m_raw < p.M && (m_raw % p.pitch) < p.valid;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int io = r * RS + 32 * ni + 8 * g;
      float v[4], rr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = acc[ni][mi][j] + bv[ni][j];
        if (EPI == EPI_RELU) v[j] = fmaxf(v[j], 0.f);
        v[j] = real ? v[j] : 0.f;
      }
      if (res_bits) {
        const unsigned nib = l_bin[16 * r + 4 * ni + g];
#pragma unroll
        for (int j = 0; j < 4; ++j) rr[j] = (nib >> j) & 1u ? 1.f : 0.f;
      } else if (res_t) {
        load4<T>(reinterpret_cast<const T *>(l_res + io), rr);
      }
      if (aux_bits) {
        unsigned w = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) w |= (v[j] > 0.f ? 1u : 0u) << j;
        l_bout[16 * r + 4 * ni + g] = (unsigned char)w;
      } else if (aux_t) {
        store4<T>(reinterpret_cast<T *>(l_aux + io), v);        // ungated / pre-residual value
      }
      if constexpr (kMask) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rr[j] > 0.f ? v[j] : 0.f;
      } else if (res_t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = real ? v[j] + rr[j] : 0.f;
      }
      store4<T>(reinterpret_cast<T *>(l_out + io), v);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int row = 8 * k + lrow;
      const int ms = mw0 + 16 * sl + row;
      const u32x4 vo = *reinterpret_cast<const u32x4 *>(l_out + row * RS + 16 * lch);
      if (ms < p.M) *reinterpret_cast<u32x4 *>(out + (int64_t)ms * p.ldc + nw0 + 8 * lch) = vo;
      if (aux_t) {
        const u32x4 va = *reinterpret_cast<const u32x4 *>(l_aux + row * RS + 16 * lch);
        if (ms < p.M) *reinterpret_cast<u32x4 *>(aux + (int64_t)ms * p.ldz + nw0 + 8 * lch) = va;
      }
    }
    if (aux_bits) {
      const int ms = mw0 + 16 * sl + brow;
      const unsigned vb = *reinterpret_cast<const unsigned *>(l_bout + 16 * brow + 4 * bch);
      if (ms < p.M)
        *reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(p.aux) + (((int64_t)ms * p.ldz + nw0) >> 2) + 4 * bch) = vb;
    }
  }
}

// ... and for the GLU epilogue (EPI_GLU, gate-only saves or none): a wave's 64 accumulator columns are 32 (a, b) pairs
// = 32 output channels, i.e. 64-byte row segments of the output, of the saved gate and of the residual: one 16-byte
// access per lane covers 16 rows x 64 B.
template <typename T, int NH, int AHEAD>
__device__ __forceinline__ void nt_epilogue_glu_rows(const GemmParams &p, const f32x4 (*accp)[4][4], const float (&bv)[4][4],
                                                     int mw0, int nw0, int lane, unsigned char *lw) {
  constexpr int NSL = 4 * NH;
  constexpr int RS = 80;                           // LDS row stride of a 64-byte row segment
  const T *__restrict__ res = static_cast<const T *>(p.res);
  T *__restrict__ out = static_cast<T *>(p.out);
  T *__restrict__ aux = static_cast<T *>(p.aux);
  const int g = lane >> 4, r = lane & 15;
  const int ow0 = nw0 / 2;                         // first output channel of the wave
  unsigned char *const l_res = lw, *const l_out = lw + 16 * RS, *const l_gate = lw + 32 * RS;
  const int lrow = lane >> 2, lch = lane & 3;      // 16 rows x 4 chunks of 16 bytes
  const bool has_res = p.res != nullptr, has_gate = p.aux != nullptr;
  u32x4 ring[AHEAD + 1];
  auto issue = [&](int sl, u32x4 &v) {
    if (has_res) {
      const int m_raw = mw0 + 16 * sl + lrow;
      const int64_t m = m_raw < p.M ? m_raw : p.M - 1;
      v = *reinterpret_cast<const u32x4 *>(res + m * p.ldr + ow0 + 8 * lch);
    }
  };
  if (AHEAD) issue(0, ring[0]);
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    if (!AHEAD) issue(sl, ring[0]);
    else if (sl + 1 < NSL) issue(sl + 1, ring[(sl + 1) & 1]);
    if (has_res) *reinterpret_cast<u32x4 *>(l_res + lrow * RS + 16 * lch) = ring[AHEAD ? sl & 1 : 0];
    const f32x4 (&acc)[4][4] = accp[sl / 4];
    const int mi = sl % 4;
    const int m_raw = mw0 + 16 * sl + r;
    const bool real = m_raw < p.M && (m_raw % p.pitch) < p.valid;
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const int io = r * RS + 32 * pi + 8 * g;
      float a[4], b[4], o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[j] = acc[2 * pi][mi][j] + bv[2 * pi][j];
        b[j] = acc[2 * pi + 1][mi][j] + bv[2 * pi + 1][j];
        o[j] = real ? a[j] * sigmoidf_(b[j]) : 0.f;
      }
      if (has_gate) store4<T>(reinterpret_cast<T *>(l_gate + io), b);
      if (has_res) {
        float rr[4];
        load4<T>(reinterpret_cast<const T *>(l_res + io), rr);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = real ? o[j] + rr[j] : 0.f;
      }
      store4<T>(reinterpret_cast<T *>(l_out + io), o);
    }
    const int ms = mw0 + 16 * sl + lrow;
    const u32x4 vo = *reinterpret_cast<const u32x4 *>(l_out + lrow * RS + 16 * lch);
    if (ms < p.M) *reinterpret_cast<u32x4 *>(out + (int64_t)ms * p.ldc + ow0 + 8 * lch) = vo;
    if (has_gate) {
      const u32x4 vg = *reinterpret_cast<const u32x4 *>(l_gate + lrow * RS + 16 * lch);
      if (ms < p.M) *reinterpret_cast<u32x4 *>(aux + (int64_t)ms * p.ldz + ow0 + 8 * lch) = vg;
    }
  }
}

// Per-wave conditions of the LDS-transposed epilogues: all 64 columns of the wave's sub-tile are stored, rows of every
// tensor touched as T are 16-byte aligned, groups of sign nibbles 4-byte aligned.  (Anything else takes nt_epilogue.)
__device__ __forceinline__ bool nt_rows_ok(const GemmParams &p, int epi, int nw0) {
  if (!p.rows_epilogue) return false;
  if (epi == EPI_GLU) {
    if (nw0 + 64 > p.N || nw0 / 2 + 32 > p.n_store) return false;
    bool ok = (p.ldc & 7) == 0 && ((uintptr_t)p.out & 15) == 0;
    if (p.res) ok = ok && (p.ldr & 7) == 0 && ((uintptr_t)p.res & 15) == 0;
    if (p.aux) ok = ok && p.gate_only && (p.ldz & 7) == 0 && ((uintptr_t)p.aux & 15) == 0;
    return ok;
  }
  if (nw0 + 64 > p.n_store) return false;
  if (epi == EPI_GLU_BWD)
    return p.gate_only && (((p.res ? p.ldr : 0) | p.ldz | p.ldy | p.ldc) & 7) == 0 &&
           ((((uintptr_t)p.res) | ((uintptr_t)p.aux) | ((uintptr_t)p.aux2) | ((uintptr_t)p.out)) & 15) == 0;
  const bool kMask = epi == EPI_MASK;
  const bool res_bits = kMask && p.mask_bits, res_t = !res_bits && (kMask || p.res != nullptr);
  const bool aux_bits = !kMask && p.mask_bits && p.aux != nullptr, aux_t = !aux_bits && p.aux != nullptr;
  // a store-only epilogue (no operand to read, one output) gains nothing from the detour: measured 85 -> 99 us on the
  // 1 282 048 x 64 x 64 launch, +-2 % on the deep ones
  if (!res_t && !res_bits && !aux_t && !aux_bits) return false;
  bool ok = (p.ldc & 7) == 0 && ((uintptr_t)p.out & 15) == 0;
  if (res_t) ok = ok && (p.ldr & 7) == 0 && ((uintptr_t)p.res & 15) == 0;
  if (res_bits) ok = ok && (p.ldr & 15) == 0 && ((uintptr_t)p.res & 3) == 0;
  if (aux_t) ok = ok && (p.ldz & 7) == 0 && ((uintptr_t)p.aux & 15) == 0;
  if (aux_bits) ok = ok && (p.ldz & 15) == 0 && ((uintptr_t)p.aux & 3) == 0;
  return ok;
}

// bytes of wave-private LDS an LDS-transposed epilogue needs
constexpr int nt_rows_lds(int epi) { return epi == EPI_GLU_BWD ? 48 * 144 + 16 * 272 : epi == EPI_GLU ? 48 * 80 : 48 * 144 + 2 * 256; }

// One wave's epilogue: through LDS when nt_rows_ok, else the generic one.  The caller has made sure (barrier) that
// nobody still reads the K loop's LDS tiles.
template <typename T, int EPI, int NH, int AHEAD>
__device__ __forceinline__ void nt_epilogue_any(const GemmParams &p, const f32x4 (*accp)[4][4], const float (&bv)[4][4], int m0,
                                                int n0, int wm0, int wn, int lane, unsigned char *lw) {
  const int mw0 = m0 + 64 * wm0, nw0 = n0 + 64 * wn;
  if constexpr (sizeof(T) == 2) {
    if (nt_rows_ok(p, EPI, nw0)) {
      if constexpr (EPI == EPI_GLU_BWD) nt_epilogue_glu_bwd_rows<T, NH, AHEAD>(p, accp, mw0, nw0, lane, lw);
      else if constexpr (EPI == EPI_GLU) nt_epilogue_glu_rows<T, NH, AHEAD>(p, accp, bv, mw0, nw0, lane, lw);
      else nt_epilogue_rows<T, EPI, NH, AHEAD>(p, accp, bv, mw0, nw0, lane, lw);
      return;
    }
  }
  nt_epilogue<T, EPI, NH>(p, accp, bv, m0, n0, wm0, wn, lane >> 4, lane & 15);
}

// Block tiles BM x BN, one wave per 64x64 sub-tile:
//   128x128 (4 waves, 32 KB LDS, 4 workgroups/CU), 256x128 (8 waves, 48 KB, 2-3 workgroups/CU): single LDS
//   buffer, the interleaving of the co-resident workgroups hides the load latency;
//   256x256 (16 waves, one workgroup per CU): two LDS buffers (128 KB), the DMA of step k+1 runs under the
//   MFMAs of step k, one barrier per step.
// The kernel is bound by L2 -> LDS bandwidth (a 128x128x64 tile moves 32 KB per 2.1 MFLOP = 64 flop/B; 256x128:
// 85 flop/B; 256x256: 128 flop/B), so the largest tile that still fills the chip wins.
template <typename T, int EPI, int BM, int BN>
// 16-bit element types: four waves per SIMD (128 VGPRs).  f32 (the parity path) carries 16-byte operand registers
// through the epilogues and needs up to ~170: it is allowed down to two waves per SIMD instead of spilling.  So is the
// GLU-backward epilogue (three operand streams): held to 128 it parked 84 values in AGPRs (v_accvgpr moves in the loop,
// "desired occupancy 4, final 2"); allowed to choose, it takes 119 VGPRs and no AGPR -- three waves per SIMD (its LDS
// staging caps it there anyway): 1.06 -> 0.90 ms per step over its five launches, same box.  (The MASK epilogue at 128
// VGPRs spills 9 dwords; given 140 registers at three waves it ran 6 % slower: it stays at four.)
__global__ __launch_bounds__(BM * BN / 64) __attribute__((amdgpu_waves_per_eu((sizeof(T) == 4 || EPI == EPI_GLU_BWD) ? 2 : 4, 4))) void gemm_nt_kernel(const GemmParams p) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BK = 8 * EPC;  // 64 bf16 / 32 f32: LDS rows are 128 B either way
  constexpr int NT = BM * BN / 64;   // threads: one wave per 64x64 sub-tile
  constexpr int WN = BN / 64;        // waves along n
  constexpr int ACH = BM * 8 / NT;   // activation chunks per thread
  constexpr int WCH = BN * 8 / NT;   // weight chunks per thread
  constexpr bool DB = (BM == 256 && BN == 256);
  constexpr int STAGE = (BM + BN) * 8;
  // per stage [row * 8 + chunk]: activations, then weights; after the K loop the same memory is the waves' private
  // transposition space of the LDS-routed epilogues (16-bit types; the GLU-backward one needs 11 KB per wave)
  constexpr int TILE_CHUNKS = (DB ? 2 : 1) * STAGE;
  // LDS-routed epilogues: the 128 x 128 tile of the 16-bit types (the 16-wave tile is the A/B fallback of gemm_nt8_kernel
  // and the 256 x 128 tile would drop to one or two workgroups per CU: both keep the generic epilogue)
  constexpr bool ROWS = sizeof(T) == 2 && BM == 128 && BN == 128;
  constexpr int EPI_CHUNKS = ROWS ? (NT / 64) * nt_rows_lds(EPI) / 16 : 0;
  __shared__ uint4 lds_all[TILE_CHUNKS > EPI_CHUNKS ? TILE_CHUNKS : EPI_CHUNKS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int g = lane >> 4, r = lane & 15;
  // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (each with its own 4 MB L2), so
  // ids b and b+8 share an L2.  All n-tiles of one m-tile get ids that are 8 apart: they run back to back on
  // ONE XCD and the activation panel (128 x K) is fetched from HBM once instead of once per n-tile; the small
  // weight matrix is served from the Infinity Cache.  Placement affects speed only.
  const int NB = (p.N + BN - 1) / BN;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int m_tile = (local / NB) * 8 + xcd;
  const int n0 = (local % NB) * BN;
  const int m0 = m_tile * BM;
  if (m0 >= p.M) return;
  // The first workgroup also clears the rows that frame the output buffer (leading zero row, slack rows), so the
  // host never issues fill kernels for them.
  if (blockIdx.x == 0) {
    T *o = static_cast<T *>(p.out);
    T *x = (EPI != EPI_GLU && EPI != EPI_GLU_BWD && !(EPI == EPI_RELU && p.mask_bits)) ? static_cast<T *>(p.aux) : nullptr;
    for (int64_t i = threadIdx.x; i < p.zero_head; i += NT) {
      o[-1 - i] = Elem<T>::from_f(0.f);
      if (x) x[-1 - i] = Elem<T>::from_f(0.f);
    }
    const int64_t tail0 = (int64_t)p.M * p.ldc, tailx = (int64_t)p.M * p.ldz;
    for (int64_t i = threadIdx.x; i < p.zero_tail; i += NT) {
      o[tail0 + i] = Elem<T>::from_f(0.f);
      if (x) x[tailx + i] = Elem<T>::from_f(0.f);
    }
  }
  const T *A = static_cast<const T *>(p.A);
  const T *W = static_cast<const T *>(p.W);

  // ---- HBM -> LDS: thread handles linear LDS chunk positions it*NT + tid of each tile
  const T *ga[ACH], *gw[WCH];
#pragma unroll
  for (int it = 0; it < ACH; ++it) {
    const int pos = it * NT + tid;
    const int row = pos >> 3, cphys = pos & 7;
    const int clog = cphys ^ (row & 7);
    int am = m0 + row;
    am = am < p.M ? am : p.M - 1;
    ga[it] = A + (int64_t)am * p.lda + clog * EPC;
  }
#pragma unroll
  for (int it = 0; it < WCH; ++it) {
    const int pos = it * NT + tid;
    const int row = pos >> 3, cphys = pos & 7;
    const int clog = cphys ^ (row & 7);
    int wr = n0 + row;
    wr = wr < p.N ? wr : p.N - 1;
    gw[it] = W + (int64_t)wr * p.ldw + clog * EPC;
  }
  typedef __attribute__((address_space(3))) void *lds_ptr;
  typedef const __attribute__((address_space(1))) void *glb_ptr;
  const int wave_u = uniform(wave);
  // one wave-instruction fills 1 KiB = 8 LDS rows; lane L writes chunk position it*NT + wave*64 + L
#define CUM_GLDS(k0, stage)                                                                                    \
  do {                                                                                                         \
    _Pragma("unroll") for (int it = 0; it < ACH; ++it)                                                        \
      __builtin_amdgcn_global_load_lds((glb_ptr)(ga[it] + (k0)),                                               \
                                       (lds_ptr)(&lds_all[(stage) * STAGE + it * NT + wave_u * 64]), 16, 0, 0);  \
    _Pragma("unroll") for (int it = 0; it < WCH; ++it)                                                        \
      __builtin_amdgcn_global_load_lds((glb_ptr)(gw[it] + (k0)),                                               \
                                       (lds_ptr)(&lds_all[(stage) * STAGE + BM * 8 + it * NT + wave_u * 64]), 16, 0, 0); \
  } while (0)

  f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float bv[4][4];
  nt_load_bias(p, n0, wn, g, bv);
  const int nk = p.K / BK;
  if constexpr (DB) CUM_GLDS(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    if constexpr (!DB) CUM_GLDS(kt * BK, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // DB: stage kt has landed for every wave, and every wave is done reading stage kt-1
    if constexpr (DB) {
      if (kt + 1 < nk) CUM_GLDS((kt + 1) * BK, (kt + 1) & 1);
    }
    const uint4 *const ldsA = lds_all + (DB ? (kt & 1) * STAGE : 0), *const ldsW = ldsA + BM * 8;
    nt_compute<T>(ldsA, ldsW, acc, wm, wn, g, r);
    if constexpr (!DB) __syncthreads();  // every wave is done reading before the next tile overwrites the buffer
  }
#undef CUM_GLDS

  if constexpr (ROWS)   // (the single-buffered loop ends with a barrier: the LDS is free; 128 VGPRs: no slab of loads ahead)
    nt_epilogue_any<T, EPI, 1, (EPI == EPI_GLU_BWD && sizeof(T) == 2) ? 1 : 0>(p, &acc, bv, m0, n0, wm, wn, lane,
                                  reinterpret_cast<unsigned char *>(lds_all) + wave * nt_rows_lds(EPI));
  else
    nt_epilogue<T, EPI>(p, &acc, bv, m0, n0, wm, wn, g, r);
}

#ifdef CUM_AB   // gemm_nt8_kernel: the predecessor of gemm_nt9_kernel, kept for same-box A/B runs (CUM_NT9=0)
// ---------------------------------------------------------------- 256 x 256 tile, 8 waves, DMA in flight across barriers
// The 16-wave 256x256 kernel above waits `vmcnt(0)` + `__syncthreads()` at the top of every K-step: one LDS-DMA stage in
// flight, every wave stalled while it lands, 16 waves x 64x64 sub-tiles (0.5 fragment reads per MFMA).  This variant
// follows the structure cdna_hip_programming.md section 5 measures at 1.3-1.45x such a loop (256^2 tile, 8 waves of
// 128 x 64, K-step 64, raw s_barrier, counted vmcnt, never 0 in the loop), with its own unit schedule:
//   * a K-tile is four 16 KB UNITS -- activation rows 0-127 / 128-255 (A0, A1), weight rows 0-127 / 128-255 (W0, W1);
//     two K-tiles of units = 128 KB, ONE __shared__ array;
//   * wave (wr, wc) owns rows [128 wr, +128) x channels [64 wc, +64): it reads unit A_wr whole at the start of the
//     K-tile (16 fragment reads, kept in registers) and W_(wc >> 1) in two halves (phases 1 and 3);
//   * so K-tile t's A units are free after phase 1 and its W units after phase 3, and the units of K-tile t + 2 are
//     DMA'd into them one per phase (A0, A1, W0, W1) while K-tile t's 64 MFMAs per wave run: at the top of K-tile t + 1
//     a counted `s_waitcnt vmcnt(8)` retires K-tile t + 1's units and leaves all eight DMAs of K-tile t + 2 in flight --
//     every unit has one to two K-tiles (2-4 k cycles) of cover instead of at most one K-step;
//   * three barriers per K-tile: B1 (K-tile landed, before the first read), B2 (A units read by every wave), B3 (W units
//     read by every wave).  A DMA'd unit is read only after the wait that retires it AND a barrier; a unit is re-staged
//     only after a barrier that follows every wave's lgkmcnt(0) on its reads.
template <typename T, int EPI>
__global__ __launch_bounds__(512) void gemm_nt8_kernel(const GemmParams p) {
  static_assert(sizeof(T) == 2, "16-bit element types only");
  constexpr int EPC = 8, BK = 64;
  constexpr int UNIT = 128 * 8;                       // 16-byte chunks of one unit: [128 rows][8 chunks], swizzled
  __shared__ uint4 lds_all[2 * 4 * UNIT];             // [K-tile parity][A0, A1, W0, W1]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int g = lane >> 4, r = lane & 15;
  const int NB = (p.N + 255) / 256;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int m_tile = (local / NB) * 8 + xcd;
  const int n0 = (local % NB) * 256;
  const int m0 = m_tile * 256;
  if (m0 >= p.M) return;
  if (blockIdx.x == 0) {                              // framing rows of the output buffer (see gemm_nt_kernel)
    T *o = static_cast<T *>(p.out);
    T *x = (EPI != EPI_GLU && EPI != EPI_GLU_BWD && !(EPI == EPI_RELU && p.mask_bits)) ? static_cast<T *>(p.aux) : nullptr;
    for (int64_t i = threadIdx.x; i < p.zero_head; i += 512) {
      o[-1 - i] = Elem<T>::from_f(0.f);
      if (x) x[-1 - i] = Elem<T>::from_f(0.f);
    }
    const int64_t tail0 = (int64_t)p.M * p.ldc, tailx = (int64_t)p.M * p.ldz;
    for (int64_t i = threadIdx.x; i < p.zero_tail; i += 512) {
      o[tail0 + i] = Elem<T>::from_f(0.f);
      if (x) x[tailx + i] = Elem<T>::from_f(0.f);
    }
  }
  const T *A = static_cast<const T *>(p.A);
  const T *W = static_cast<const T *>(p.W);
  // per-thread DMA sources: unit u (0 A0, 1 A1, 2 W0, 3 W1), instruction it (0, 1): linear LDS chunk it*512 + tid of the unit
  const T *src[4][2];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int pos = it * 512 + tid;
      const int row = pos >> 3, cphys = pos & 7;
      const int clog = cphys ^ (row & 7);
      if (u < 2) {
        int am = m0 + 128 * u + row;
        am = am < p.M ? am : p.M - 1;
        src[u][it] = A + (int64_t)am * p.lda + clog * EPC;
      } else {
        int wn_ = n0 + 128 * (u - 2) + row;
        wn_ = wn_ < p.N ? wn_ : p.N - 1;
        src[u][it] = W + (int64_t)wn_ * p.ldw + clog * EPC;
      }
    }
  typedef __attribute__((address_space(3))) void *lds_ptr;
  typedef const __attribute__((address_space(1))) void *glb_ptr;
#define CUM_STAGE(u, kt, par)                                                                                   \
  do {                                                                                                          \
    _Pragma("unroll") for (int it = 0; it < 2; ++it)                                                           \
      __builtin_amdgcn_global_load_lds((glb_ptr)(src[u][it] + (kt) * BK),                                      \
                                       (lds_ptr)(&lds_all[((par) * 4 + (u)) * UNIT + it * 512 + wave * 64]), 16, 0, 0); \
  } while (0)

  f32x4 acc[2][4][4];                                 // [m half][ni][mi]: rows 128 wr + 64 h + 16 mi, channels 64 wc + 16 ni
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bv[4][4];
  nt_load_bias(p, n0, wc, g, bv);

  const int nk = p.K / BK;
#pragma unroll
  for (int u = 0; u < 4; ++u) CUM_STAGE(u, 0, 0);
  if (nk > 1) {
#pragma unroll
    for (int u = 0; u < 4; ++u) CUM_STAGE(u, 1, 1);
  }
  // Fragment reads are inline asm: hipcc's wait insertion sees every ds_read of an array that LDS-DMAs are in flight
  // into as a reason for `s_waitcnt vmcnt(0)` (it cannot tell the units apart), which would drain the pipeline twice
  // per K-tile.  An asm read is not counted by the compiler: each group of reads is followed by one wait statement
  // that names every destination "+v" (cdna_hip_programming.md 5.7, form ii), so no consumer is scheduled above it.
  // Addresses: byte offset of (row, 16-byte chunk) inside a unit = row * 128 + ((chunk ^ (row & 7)) << 4); rows 16 apart
  // share the swizzle term, so one base register per K half (ks) + immediate offsets serves all row tiles.
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)lds_all;
  unsigned aA[2], aW[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int cl = ks * 4 + g;
    aA[ks] = lds0 + (unsigned)((wr * UNIT + r * 8 + (cl ^ (r & 7))) * 16);
    aW[ks] = lds0 + (unsigned)(((2 + (wc >> 1)) * UNIT + ((wc & 1) * 64 + r) * 8 + (cl ^ (r & 7))) * 16);
  }
#define CUM_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")
  for (int kt = 0; kt < nk; ++kt) {
    const int par = kt & 1;
    const unsigned pb = (unsigned)par * (4 * UNIT * 16);
    const unsigned a0 = aA[0] + pb, a1 = aA[1] + pb, w0 = aW[0] + pb, w1 = aW[1] + pb;
    const bool more = kt + 2 < nk;
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // K-tile kt landed; K-tile kt + 1 stays in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");                              // B1
    // ---- phase 1: W fragments of channels 0-31 and all A fragments of the K-tile, issued in the order the MFMAs
    //      consume them (LDS returns in order): the first half-quadrant starts after 6 of the 20 reads, the rest land
    //      under MFMAs
    u32x4 af[2][8], wf[2][2];        // native vectors: an asm operand of the HIP uint4 struct would go through memory
    CUM_DSR(wf[0][0], w0, 0);     CUM_DSR(wf[0][1], w0, 2048);
    CUM_DSR(af[0][0], a0, 0);     CUM_DSR(af[0][1], a0, 2048);  CUM_DSR(af[0][2], a0, 4096);  CUM_DSR(af[0][3], a0, 6144);
    CUM_DSR(wf[1][0], w1, 0);     CUM_DSR(wf[1][1], w1, 2048);
    CUM_DSR(af[1][0], a1, 0);     CUM_DSR(af[1][1], a1, 2048);  CUM_DSR(af[1][2], a1, 4096);  CUM_DSR(af[1][3], a1, 6144);
    CUM_DSR(af[0][4], a0, 8192);  CUM_DSR(af[0][5], a0, 10240); CUM_DSR(af[0][6], a0, 12288); CUM_DSR(af[0][7], a0, 14336);
    CUM_DSR(af[1][4], a1, 8192);  CUM_DSR(af[1][5], a1, 10240); CUM_DSR(af[1][6], a1, 12288); CUM_DSR(af[1][7], a1, 14336);
#define CUM_HALFQ(h, nlo, ks)                                                                                  \
  do {                                                                                                         \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                           \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                                       \
        if constexpr (__is_same(T, f16))                                                                       \
          acc[h][(nlo) + ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                                     \
              __builtin_bit_cast(f16x8, wf[ks][ni]), __builtin_bit_cast(f16x8, af[ks][4 * (h) + mi]),          \
              acc[h][(nlo) + ni][mi], 0, 0, 0);                                                                \
        else                                                                                                   \
          acc[h][(nlo) + ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                    \
              __builtin_bit_cast(bf16x8, wf[ks][ni]), __builtin_bit_cast(bf16x8, af[ks][4 * (h) + mi]),        \
              acc[h][(nlo) + ni][mi], 0, 0, 0);                                                                \
      }                                                                                                        \
  } while (0)
#define CUM_QUAD(h, nlo)         \
  do {                           \
    CUM_HALFQ(h, nlo, 0);        \
    CUM_HALFQ(h, nlo, 1);        \
  } while (0)
    asm volatile("s_waitcnt lgkmcnt(14)"
                 : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]) : : "memory");
    __builtin_amdgcn_s_setprio(1);
    CUM_HALFQ(0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);     // keep these 8 MFMAs in front of the next wait: they cover the reads it waits for
    asm volatile("s_waitcnt lgkmcnt(8)"
                 : "+v"(wf[1][0]), "+v"(wf[1][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][3]) : : "memory");
    CUM_HALFQ(0, 0, 1);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(af[0][4]), "+v"(af[0][5]), "+v"(af[0][6]), "+v"(af[0][7]), "+v"(af[1][4]), "+v"(af[1][5]),
                   "+v"(af[1][6]), "+v"(af[1][7]) : : "memory");
    asm volatile("s_barrier" ::: "memory");                              // B2: the A units of this parity are free
    // Waves w and w + 4 share a SIMD.  Issuing an LDS-DMA costs the issuing wave ~100 cycles apiece; if both partners
    // issue theirs at the same point of the K-tile the SIMD's matrix pipe idles meanwhile.  So the two halves of the
    // workgroup take the DMA issue at different points: waves 0-3 stage first and compute after, waves 4-7 compute
    // first (same barriers, same DMA count between the counted waits).
    // ---- phase 2
    const bool early = more && wr == 0, late = more && wr != 0;
    if (early) {
      CUM_STAGE(0, kt + 2, par);
      CUM_STAGE(1, kt + 2, par);
    }
    __builtin_amdgcn_s_setprio(1);
    CUM_QUAD(1, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (late) {
      CUM_STAGE(0, kt + 2, par);
      CUM_STAGE(1, kt + 2, par);
    }
    // ---- phase 3: W fragments of channels 32-63 (rows + 32 of the unit: + 4096 bytes)
    CUM_DSR(wf[0][0], w0, 4096);  CUM_DSR(wf[0][1], w0, 6144);  CUM_DSR(wf[1][0], w1, 4096);  CUM_DSR(wf[1][1], w1, 6144);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[1][0]), "+v"(wf[1][1]) : : "memory");
    asm volatile("s_barrier" ::: "memory");                              // B3: the W units of this parity are free
    if (early) {
      CUM_STAGE(2, kt + 2, par);
      CUM_STAGE(3, kt + 2, par);
    }
    __builtin_amdgcn_s_setprio(1);
    CUM_QUAD(1, 2);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (late) {
      CUM_STAGE(2, kt + 2, par);
      CUM_STAGE(3, kt + 2, par);
    }
    // ---- phase 4
    __builtin_amdgcn_s_setprio(1);
    CUM_QUAD(0, 2);
    __builtin_amdgcn_s_setprio(0);
  }
#undef CUM_HALFQ
#undef CUM_DSR
#undef CUM_QUAD
#undef CUM_STAGE
  // Epilogue: through wave-private LDS where the conditions hold (nt_epilogue_any); the fragment registers are dead here,
  // so one slab of operand loads is issued ahead.  (Deeper pipelining of the generic epilogue measured neutral: it is
  // bound by its access pattern, not by loads in flight.)
  asm volatile("s_barrier" ::: "memory");          // every wave is done reading the K loop's LDS units
  nt_epilogue_any<T, EPI, 2, 1>(p, acc, bv, m0, n0, 2 * wr, wc, lane,
                                reinterpret_cast<unsigned char *>(lds_all) + wave * nt_rows_lds(EPI));
}

#endif  // CUM_AB

// ---------------------------------------------------------------- 256 x 256 tile, 8 waves, two wave groups in ping-pong
// Same tile, units, DMA scheme and epilogues as gemm_nt8_kernel; what changes is WHEN the two halves of the workgroup do
// what.  In gemm_nt8_kernel all eight waves run the same phase at the same time: both waves of a SIMD issue their 20
// fragment reads together, wait for them together, want the matrix pipe together and meet at the same three barriers --
// the pipe measured 48 % busy.  Here a K-tile is eight SLOTS (one barrier each), alternately a LOAD slot (issue the
// fragment reads of the next 16 MFMAs and this slot's share of the LDS-DMAs, then wait at the barrier) and a COMPUTE slot
// (16 MFMAs = one 64 x 32 quadrant over the K-tile), and waves 4-7 run ONE SLOT BEHIND waves 0-3: on every SIMD one wave
// computes while its partner loads (MI355X_MICROARCH.md "Two waves per SIMD", cdna_hip_programming.md 5 "8-phase").
//   group 0 (rows 0-127):   L1 C1 L2 C2 L3 C3 L4 C4 | L1 ...        group 1 (rows 128-255): .. L1 C1 L2 C2 L3 C3 L4 C4 | ...
//   L1: A rows 0-63 + W channels 0-31 (12 reads)   C1: (rows 0-63,  ch 0-31)
//   L2: A rows 64-127 (8 reads)                    C2: (rows 64-127, ch 0-31)
//   L3: W channels 32-63 (4 reads)                 C3: (rows 64-127, ch 32-63)
//   L4: -                                          C4: (rows 0-63,  ch 32-63)
// Unit lifetimes (K-tile t, slots counted in barriers 8 t + i of group 0): A0 is read by group 0 only, in L1 / L2 -> free
// after barrier 8t+4; A1 by group 1 only -> free after 8t+5; W0 / W1 by both, last in group 1's L3 -> free after 8t+7.
// LDS-DMA of K-tile t + 2 into the freed units: group 0 issues A0 in L3, A1 in L4, W0 + W1 in the next K-tile's L1;
// group 1 issues A0 + A1 in L3, W0 + W1 in L4.  One counted wait per K-tile and wave (end of C4 / in L4): vmcnt(4) = K-tile
// t + 1 has landed, the four A instructions of K-tile t + 2 stay in flight; the barrier behind it publishes K-tile t + 1.
// (Spreading the DMA issue evenly over the load slots -- two instructions in each -- measured the same to +-2 %.)
// Measured on MI355X (tools/bench_gemm.py, bf16): enc3-enc6 conv 0.84 / 1.01 / 1.09 / 1.08 -> 0.90 / 1.09 / 1.20 / 1.17
// PFLOP/s, plain 8192^3 1.14 -> 1.22-1.25; SQ counters on the plain GEMM (tools/pmc_gemm_plain.sh): matrix pipe busy
// 59 -> 68 % of the kernel's cycles while the chip's clock under this load fell 1.58 -> 1.51 GHz (power: part of every
// gain in MFMA density is given back as clock, MI355X_MICROARCH.md "DVFS give-back").
template <typename T, int EPI>
__global__ __launch_bounds__(512) void gemm_nt9_kernel(const GemmParams p) {
  static_assert(sizeof(T) == 2, "16-bit element types only");
  constexpr int EPC = 8, BK = 64;
  constexpr int UNIT = 128 * 8;
  __shared__ uint4 lds_all[2 * 4 * UNIT];             // [K-tile parity][A0, A1, W0, W1]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int g = lane >> 4, r = lane & 15;
  // Tile order: workgroup ids b, b + 8, ... run on one XCD (one L2).  An XCD walks `group_m` of its m-tiles side by side,
  // n-tile after n-tile: its 32 resident workgroups then cover group_m x (32 / group_m) tiles and pull group_m A panels +
  // 32 / group_m W panels per K-tile through its L2 instead of 1 + 32 (group_m = 1: one m-tile after the other, which is what
  // the layers' N <= 1 536 = 6 n-tiles want; a square GEMM with 32 n-tiles streams all of W per m-tile that way).
  const int NB = (p.N + 255) / 256;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int per = p.group_m * NB, grp = local / per, within = local - grp * per;
  const int left = (int)gridDim.x / (8 * NB) - grp * p.group_m;
  const int gm = left < p.group_m ? left : p.group_m;
  const int m_tile = (grp * p.group_m + within % gm) * 8 + xcd;
  const int n0 = (within / gm) * 256;
  const int m0 = m_tile * 256;
  if (m0 >= p.M) return;
  if (blockIdx.x == 0) {                              // framing rows of the output buffer (see gemm_nt_kernel)
    T *o = static_cast<T *>(p.out);
    T *x = (EPI != EPI_GLU && EPI != EPI_GLU_BWD && !(EPI == EPI_RELU && p.mask_bits)) ? static_cast<T *>(p.aux) : nullptr;
    for (int64_t i = threadIdx.x; i < p.zero_head; i += 512) {
      o[-1 - i] = Elem<T>::from_f(0.f);
      if (x) x[-1 - i] = Elem<T>::from_f(0.f);
    }
    const int64_t tail0 = (int64_t)p.M * p.ldc, tailx = (int64_t)p.M * p.ldz;
    for (int64_t i = threadIdx.x; i < p.zero_tail; i += 512) {
      o[tail0 + i] = Elem<T>::from_f(0.f);
      if (x) x[tailx + i] = Elem<T>::from_f(0.f);
    }
  }
  const T *A = static_cast<const T *>(p.A);
  const T *W = static_cast<const T *>(p.W);
  const T *src[4][2];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int pos = it * 512 + tid;
      const int row = pos >> 3, cphys = pos & 7;
      const int clog = cphys ^ (row & 7);
      if (u < 2) {
        int am = m0 + 128 * u + row;
        am = am < p.M ? am : p.M - 1;
        src[u][it] = A + (int64_t)am * p.lda + clog * EPC;
      } else {
        int wn_ = n0 + 128 * (u - 2) + row;
        wn_ = wn_ < p.N ? wn_ : p.N - 1;
        src[u][it] = W + (int64_t)wn_ * p.ldw + clog * EPC;
      }
    }
  typedef __attribute__((address_space(3))) void *lds_ptr;
  typedef const __attribute__((address_space(1))) void *glb_ptr;
#define CUM_STAGE(u, kt, par)                                                                                   \
  do {                                                                                                          \
    _Pragma("unroll") for (int it = 0; it < 2; ++it)                                                           \
      __builtin_amdgcn_global_load_lds((glb_ptr)(src[u][it] + (kt) * BK),                                      \
                                       (lds_ptr)(&lds_all[((par) * 4 + (u)) * UNIT + it * 512 + wave * 64]), 16, 0, 0); \
  } while (0)

  f32x4 acc[2][4][4];                                 // [m half][ni][mi]: rows 128 wr + 64 h + 16 mi, channels 64 wc + 16 ni
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bv[4][4];
  nt_load_bias(p, n0, wc, g, bv);

  const int nk = p.K / BK;
#pragma unroll
  for (int u = 0; u < 4; ++u) CUM_STAGE(u, 0, 0);
  if (nk > 1) {
#pragma unroll
    for (int u = 0; u < 4; ++u) CUM_STAGE(u, 1, 1);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)lds_all;
  unsigned aA[2], aW[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int cl = ks * 4 + g;
    aA[ks] = lds0 + (unsigned)((wr * UNIT + r * 8 + (cl ^ (r & 7))) * 16);
    aW[ks] = lds0 + (unsigned)(((2 + (wc >> 1)) * UNIT + ((wc & 1) * 64 + r) * 8 + (cl ^ (r & 7))) * 16);
  }
#define CUM_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")
#ifdef CUM_NT9_TIMING_NO_W_READS   // timing-only build (wrong results): the W fragment reads = the third of the fragment
#define CUM_DSRW(dst, addr, off) asm volatile("" : "=v"(dst) : "v"(addr))   // bytes four waves of 128 x 128 would not read
#else
#define CUM_DSRW(dst, addr, off) CUM_DSR(dst, addr, off)
#endif
#define CUM_BAR()                               \
  do {                                          \
    __builtin_amdgcn_sched_barrier(0);          \
    asm volatile("s_barrier" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);          \
  } while (0)
#define CUM_HALFQ(h, nlo, ks)                                                                                  \
  do {                                                                                                         \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                           \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                                       \
        if constexpr (__is_same(T, f16))                                                                       \
          acc[h][(nlo) + ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                                     \
              __builtin_bit_cast(f16x8, wf[ks][ni]), __builtin_bit_cast(f16x8, af[ks][4 * (h) + mi]),          \
              acc[h][(nlo) + ni][mi], 0, 0, 0);                                                                \
        else                                                                                                   \
          acc[h][(nlo) + ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                    \
              __builtin_bit_cast(bf16x8, wf[ks][ni]), __builtin_bit_cast(bf16x8, af[ks][4 * (h) + mi]),        \
              acc[h][(nlo) + ni][mi], 0, 0, 0);                                                                \
      }                                                                                                        \
  } while (0)
#define CUM_QUAD(h, nlo)                 \
  do {                                   \
    __builtin_amdgcn_s_setprio(1);       \
    CUM_HALFQ(h, nlo, 0);                \
    CUM_HALFQ(h, nlo, 1);                \
    __builtin_amdgcn_s_setprio(0);       \
  } while (0)

  if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // K-tile 0 landed; K-tile 1 stays in flight
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CUM_BAR();                                                           // K-tile 0 is visible to every wave
  if (wr != 0) CUM_BAR();                                              // group 1 runs one slot behind group 0
  u32x4 af[2][8], wf[2][2];
  for (int kt = 0; kt < nk; ++kt) {
    const int par = kt & 1;
    const unsigned pb = (unsigned)par * (4 * UNIT * 16);
    const unsigned a0 = aA[0] + pb, a1 = aA[1] + pb, w0 = aW[0] + pb, w1 = aW[1] + pb;
    const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
    // ---- L1: W channels 0-31 and A rows 0-63; group 0 stages the W units of K-tile kt + 1 (freed one slot ago)
    CUM_DSRW(wf[0][0], w0, 0);    CUM_DSRW(wf[0][1], w0, 2048);
    CUM_DSR(af[0][0], a0, 0);     CUM_DSR(af[0][1], a0, 2048);  CUM_DSR(af[0][2], a0, 4096);  CUM_DSR(af[0][3], a0, 6144);
    CUM_DSRW(wf[1][0], w1, 0);    CUM_DSRW(wf[1][1], w1, 2048);
    CUM_DSR(af[1][0], a1, 0);     CUM_DSR(af[1][1], a1, 2048);  CUM_DSR(af[1][2], a1, 4096);  CUM_DSR(af[1][3], a1, 6144);
    if (wr == 0 && kt >= 1 && more1) {
      CUM_STAGE(2, kt + 1, par ^ 1);
      CUM_STAGE(3, kt + 1, par ^ 1);
    }
    CUM_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[1][0]), "+v"(wf[1][1]), "+v"(af[0][0]), "+v"(af[0][1]),
                   "+v"(af[0][2]), "+v"(af[0][3]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][3]) : : "memory");
    CUM_QUAD(0, 0);                                                       // C1
    CUM_BAR();
    // ---- L2: A rows 64-127
    CUM_DSR(af[0][4], a0, 8192);  CUM_DSR(af[0][5], a0, 10240); CUM_DSR(af[0][6], a0, 12288); CUM_DSR(af[0][7], a0, 14336);
    CUM_DSR(af[1][4], a1, 8192);  CUM_DSR(af[1][5], a1, 10240); CUM_DSR(af[1][6], a1, 12288); CUM_DSR(af[1][7], a1, 14336);
    CUM_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(af[0][4]), "+v"(af[0][5]), "+v"(af[0][6]), "+v"(af[0][7]), "+v"(af[1][4]), "+v"(af[1][5]),
                   "+v"(af[1][6]), "+v"(af[1][7]) : : "memory");
    CUM_QUAD(1, 0);                                                       // C2
    CUM_BAR();
    // ---- L3: W channels 32-63; the A units of this parity are free: K-tile kt + 2
    CUM_DSRW(wf[0][0], w0, 4096); CUM_DSRW(wf[0][1], w0, 6144); CUM_DSRW(wf[1][0], w1, 4096); CUM_DSRW(wf[1][1], w1, 6144);
    if (more2) {
      CUM_STAGE(0, kt + 2, par);
      if (wr != 0) CUM_STAGE(1, kt + 2, par);
    }
    CUM_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[1][0]), "+v"(wf[1][1]) : : "memory");
    CUM_QUAD(1, 2);                                                       // C3
    CUM_BAR();
    // ---- L4: no reads.  group 0: A1 of K-tile kt + 2; group 1: K-tile kt + 1 must have landed, then W of K-tile kt + 2
    if (wr == 0) {
      if (more2) CUM_STAGE(1, kt + 2, par);
    } else {
      if (more1) {
        if (more2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (more2) {
        CUM_STAGE(2, kt + 2, par);
        CUM_STAGE(3, kt + 2, par);
      }
    }
    CUM_BAR();
    CUM_QUAD(0, 2);                                                       // C4
    if (wr == 0 && more1) {
      if (more2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    CUM_BAR();
  }
  if (wr == 0) CUM_BAR();                                                // group 1's last slot
#undef CUM_HALFQ
#undef CUM_DSRW
#undef CUM_DSR
#undef CUM_QUAD
#undef CUM_STAGE
#undef CUM_BAR
  asm volatile("s_barrier" ::: "memory");          // every wave is done reading the K loop's LDS units
  nt_epilogue_any<T, EPI, 2, 1>(p, acc, bv, m0, n0, 2 * wr, wc, lane,
                                reinterpret_cast<unsigned char *>(lds_all) + wave * nt_rows_lds(EPI));
}

// ---------------------------------------------------------------- 128 x 256 tile, 8 waves, three-stage LDS-DMA ring
// For launches with FEW tiles and a long K axis (M ~ 10 000: the bottleneck's projections with N = 512, the innermost conv
// layers with N = 768): 312 tiles of 128 x 128 leave a CU one or two single-buffered workgroups, each paying a whole L2
// round trip per K-step (in_proj's data gradient 76 us where the vendor's stream-K kernel takes 45), and 256 x 256 tiles
// fill half the chip.  Here a workgroup owns 128 x 256 (<= 256 workgroups: one resident round, one per CU) and walks K
// through a ring of three 48 KB stages filled by LDS-DMA TWO K-steps ahead: 96 KB per CU in flight behind one counted
// vmcnt and one barrier per K-step (the stage refilled at step kt is the one every wave finished reading before it arrived
// at kt's barrier).  Wave tile, fragment layout, MFMA order and epilogues are those of gemm_nt_kernel<T, EPI, 128, 128>
// (results are bit-identical to it); every LDS read of the loop is inline asm (a compiler-visible read behind an LDS-DMA is
// given an s_waitcnt vmcnt(0), which would drain the ring at every step).
template <typename T, int EPI>
__global__ __launch_bounds__(512) void gemm_nt_ring_kernel(const GemmParams p) {
  static_assert(sizeof(T) == 2, "16-bit element types only");
#ifndef CUM_RING_NST
#define CUM_RING_NST 3      // (2: one K-step in flight -- the timing experiment of profiles/r06_nt_ring_ab.txt)
#endif
  constexpr int EPC = 8, BK = 64, BM = 128, BN = 256, NST = CUM_RING_NST;
  static_assert(NST == 2 || NST == 3, "two or three stages");
  constexpr int STAGE = (BM + BN) * 8;                // 16-byte chunks per stage: A rows, then W rows; 48 KB
  __shared__ uint4 lds_all[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int g = lane >> 4, r = lane & 15;
  const int NB = (p.N + BN - 1) / BN;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int m_tile = (local / NB) * 8 + xcd;
  const int n0 = (local % NB) * BN;
  const int m0 = m_tile * BM;
  if (m0 >= p.M) return;
  if (blockIdx.x == 0) {                              // framing rows of the output buffer (see gemm_nt_kernel)
    T *o = static_cast<T *>(p.out);
    T *x = (EPI != EPI_GLU && EPI != EPI_GLU_BWD && !(EPI == EPI_RELU && p.mask_bits)) ? static_cast<T *>(p.aux) : nullptr;
    for (int64_t i = threadIdx.x; i < p.zero_head; i += 512) {
      o[-1 - i] = Elem<T>::from_f(0.f);
      if (x) x[-1 - i] = Elem<T>::from_f(0.f);
    }
    const int64_t tail0 = (int64_t)p.M * p.ldc, tailx = (int64_t)p.M * p.ldz;
    for (int64_t i = threadIdx.x; i < p.zero_tail; i += 512) {
      o[tail0 + i] = Elem<T>::from_f(0.f);
      if (x) x[tailx + i] = Elem<T>::from_f(0.f);
    }
  }
  const T *A = static_cast<const T *>(p.A);
  const T *W = static_cast<const T *>(p.W);
  const T *ga[2], *gw[4];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int pos = it * 512 + tid, row = pos >> 3, clog = (pos & 7) ^ (row & 7);
    int am = m0 + row;
    am = am < p.M ? am : p.M - 1;
    ga[it] = A + (int64_t)am * p.lda + clog * EPC;
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int pos = it * 512 + tid, row = pos >> 3, clog = (pos & 7) ^ (row & 7);
    int wr = n0 + row;
    wr = wr < p.N ? wr : p.N - 1;
    gw[it] = W + (int64_t)wr * p.ldw + clog * EPC;
  }
  typedef __attribute__((address_space(3))) void *lds_ptr;
  typedef const __attribute__((address_space(1))) void *glb_ptr;
  auto issue = [&](int kt, int buf) {
    uint4 *st = lds_all + buf * STAGE;
#pragma unroll
    for (int it = 0; it < 2; ++it)
      __builtin_amdgcn_global_load_lds((glb_ptr)(ga[it] + kt * BK), (lds_ptr)(&st[it * 512 + wave * 64]), 16, 0, 0);
#pragma unroll
    for (int it = 0; it < 4; ++it)
      __builtin_amdgcn_global_load_lds((glb_ptr)(gw[it] + kt * BK), (lds_ptr)(&st[BM * 8 + it * 512 + wave * 64]), 16, 0, 0);
  };

  f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bv[4][4];
  nt_load_bias(p, n0, wn, g, bv);

  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)lds_all;
  unsigned aA[2], aW[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int cl = (ks * 4 + g) ^ (r & 7);
    aA[ks] = (unsigned)(((wm * 64 + r) * 8 + cl) * 16);
    aW[ks] = (unsigned)(((BM + wn * 64 + r) * 8 + cl) * 16);
  }
#define CUM_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define CUM_FWAIT(n, KS)                                                                                              \
  asm volatile("s_waitcnt lgkmcnt(%8)"                                                                               \
               : "+v"(wf[KS][0]), "+v"(wf[KS][1]), "+v"(wf[KS][2]), "+v"(wf[KS][3]), "+v"(af[KS][0]), "+v"(af[KS][1]), \
                 "+v"(af[KS][2]), "+v"(af[KS][3]) : "n"(n) : "memory")
  const int nk = p.K / BK;
  issue(0, 0);
  if (NST == 3 && nk > 1) issue(1, 1);
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (NST == 3 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // K-step kt landed; kt + 1 stays in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");            // step kt is visible; every wave is done with step kt - 1
    if (kt + NST - 1 < nk) issue(kt + NST - 1, buf == 0 ? NST - 1 : buf - 1);
    const unsigned sb = lds0 + (unsigned)(buf * STAGE * 16);
    u32x4 af[2][4], wf[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned a = sb + aA[ks], w = sb + aW[ks];
      CUM_DSR(wf[ks][0], w, 0); CUM_DSR(wf[ks][1], w, 2048); CUM_DSR(wf[ks][2], w, 4096); CUM_DSR(wf[ks][3], w, 6144);
      CUM_DSR(af[ks][0], a, 0); CUM_DSR(af[ks][1], a, 2048); CUM_DSR(af[ks][2], a, 4096); CUM_DSR(af[ks][3], a, 6144);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      __builtin_amdgcn_sched_barrier(0);
      if (ks == 0) CUM_FWAIT(8, 0);
      else CUM_FWAIT(0, 1);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          if constexpr (__is_same(T, f16))
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[ks][ni]),
                                                                 __builtin_bit_cast(f16x8, af[ks][mi]), acc[ni][mi], 0, 0, 0);
          else
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ks][ni]),
                                                                  __builtin_bit_cast(bf16x8, af[ks][mi]), acc[ni][mi], 0, 0, 0);
        }
    }
    buf = buf == NST - 1 ? 0 : buf + 1;
  }
#undef CUM_DSR
#undef CUM_FWAIT
  asm volatile("s_barrier" ::: "memory");              // every wave is done reading the ring: it is the epilogues' space now
  nt_epilogue_any<T, EPI, 1, (EPI == EPI_GLU_BWD) ? 1 : 0>(p, &acc, bv, m0, n0, wm, wn, lane,
                                reinterpret_cast<unsigned char *>(lds_all) + wave * nt_rows_lds(EPI));
}

#ifdef CUM_AB
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// ---------------------------------------------------------------- 256 x 256 tile, FOUR waves of 128 x 128 (experiment)
// Same tile, LDS units and LDS-DMA as gemm_nt9_kernel; one wave per SIMD, its 64 accumulator blocks in AGPRs.  What it is
// for: a wave of 128 x 128 reads (128 + 128) x 64 x 2 B = 32 KB of fragments per K-tile, 128 KB per CU, where eight waves of
// 128 x 64 read 192 KB -- with the 64 KB the DMA writes, gemm_nt9_kernel keeps the LDS as busy as the matrix pipe (256 KB =
// 2 048 clk at 128 B/clk against 2 048 MFMA cycles per SIMD), and a timing-only build of it without the W fragment reads
// runs the plain 8192^3 GEMM at 1.67 PFLOP/s against 1.30 (profiles/r06_nt_asm_ab.txt).  Every instruction of the K loop
// is a volatile asm statement, so the order written here is the order issued:
//   phase A (64 MFMAs on K-half 0): the 32 fragment reads of K-half 1 under MFMAs 0-31; after MFMA 39 lgkmcnt(0) + barrier
//     (every wave is done reading this parity's units), then the 16 LDS-DMAs of K-tile kt + 2 under MFMAs 40-55;
//   phase B (64 MFMAs on K-half 1): after MFMA 23 vmcnt(16) + barrier (K-tile kt + 1 has landed for every wave), then the
//     32 fragment reads of its K-half 0 under MFMAs 24-55, lgkmcnt(0) behind MFMA 63.
template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm_nt4_kernel(const GemmParams p) {
  static_assert(sizeof(T) == 2, "16-bit element types only");
  constexpr int EPC = 8, BK = 64;
  constexpr int UNIT = 128 * 8;
  __shared__ uint4 lds_all[2 * 4 * UNIT];             // [K-tile parity][A0, A1, W0, W1]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int g = lane >> 4, r = lane & 15;
  const int NB = (p.N + 255) / 256;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int per = p.group_m * NB, grp = local / per, within = local - grp * per;
  const int left = (int)gridDim.x / (8 * NB) - grp * p.group_m;
  const int gm = left < p.group_m ? left : p.group_m;
  const int m_tile = (grp * p.group_m + within % gm) * 8 + xcd;
  const int n0 = (within / gm) * 256;
  const int m0 = m_tile * 256;
  if (m0 >= p.M) return;
  if (blockIdx.x == 0) {                              // framing rows of the output buffer (see gemm_nt_kernel)
    T *o = static_cast<T *>(p.out);
    T *x = (EPI != EPI_GLU && EPI != EPI_GLU_BWD && !(EPI == EPI_RELU && p.mask_bits)) ? static_cast<T *>(p.aux) : nullptr;
    for (int64_t i = threadIdx.x; i < p.zero_head; i += 256) {
      o[-1 - i] = Elem<T>::from_f(0.f);
      if (x) x[-1 - i] = Elem<T>::from_f(0.f);
    }
    const int64_t tail0 = (int64_t)p.M * p.ldc, tailx = (int64_t)p.M * p.ldz;
    for (int64_t i = threadIdx.x; i < p.zero_tail; i += 256) {
      o[tail0 + i] = Elem<T>::from_f(0.f);
      if (x) x[tailx + i] = Elem<T>::from_f(0.f);
    }
  }
  const T *A = static_cast<const T *>(p.A);
  const T *W = static_cast<const T *>(p.W);
  const T *src[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int pos = it * 256 + tid;
      const int row = pos >> 3, cphys = pos & 7;
      const int clog = cphys ^ (row & 7);
      if (u < 2) {
        int am = m0 + 128 * u + row;
        am = am < p.M ? am : p.M - 1;
        src[u][it] = A + (int64_t)am * p.lda + clog * EPC;
      } else {
        int wn_ = n0 + 128 * (u - 2) + row;
        wn_ = wn_ < p.N ? wn_ : p.N - 1;
        src[u][it] = W + (int64_t)wn_ * p.ldw + clog * EPC;
      }
    }
  typedef __attribute__((address_space(3))) void *lds_ptr;
  typedef const __attribute__((address_space(1))) void *glb_ptr;
#define CUM_DMA(u, it, kt, par)                                                                       \
  __builtin_amdgcn_global_load_lds((glb_ptr)(src[u][it] + (kt) * BK),                                 \
                                   (lds_ptr)(&lds_all[((par) * 4 + (u)) * UNIT + (it) * 256 + wave * 64]), 16, 0, 0)

  f32x4 acc[8][8];                                    // [ni][mi] in AGPRs: channels 128 wc + 16 ni, rows 128 wr + 16 mi
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int it = 0; it < 4; ++it) CUM_DMA(u, it, 0, 0);
  {
    const int k1 = nk > 1 ? 1 : 0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int it = 0; it < 4; ++it) CUM_DMA(u, it, k1, 1);
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)lds_all;
  unsigned aA[2], aW[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int cl = ks * 4 + g;
    aA[ks] = lds0 + (unsigned)((wr * UNIT + r * 8 + (cl ^ (r & 7))) * 16);
    aW[ks] = lds0 + (unsigned)(((2 + wc) * UNIT + r * 8 + (cl ^ (r & 7))) * 16);
  }
#define CUM_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define CUM_WAIT_FRAGS(ks)                                                                                               \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                    \
               : "+v"(af[ks][0]), "+v"(af[ks][1]), "+v"(af[ks][2]), "+v"(af[ks][3]), "+v"(af[ks][4]), "+v"(af[ks][5]),   \
                 "+v"(af[ks][6]), "+v"(af[ks][7]), "+v"(wf[ks][0]), "+v"(wf[ks][1]), "+v"(wf[ks][2]), "+v"(wf[ks][3]),   \
                 "+v"(wf[ks][4]), "+v"(wf[ks][5]), "+v"(wf[ks][6]), "+v"(wf[ks][7]) : : "memory")
#define CUM_MFMA(ks, ni, mi)                                                                                             \
  do {                                                                                                                   \
    if constexpr (__is_same(T, f16))                                                                                     \
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[ni][mi]) : "v"(wf[ks][ni]), "v"(af[ks][mi]));     \
    else                                                                                                                 \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[ni][mi]) : "v"(wf[ks][ni]), "v"(af[ks][mi]));    \
  } while (0)
  // fragment read number J (0-15) of K-half KS at parity offset pb: A block J / 2 (even J) or W block J / 2 (odd J)
#define CUM_FRAG(KS, J, pb)                                                             \
  do {                                                                                  \
    if constexpr (((J) & 1) != 0) CUM_DSR(wf[KS][(J) >> 1], aW[KS] + (pb), ((J) >> 1) * 2048); \
    else CUM_DSR(af[KS][(J) >> 1], aA[KS] + (pb), ((J) >> 1) * 2048);                   \
  } while (0)

  u32x4 af[2][8], wf[2][8];
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                     // K-tile 0 landed; K-tile 1 stays in flight
  asm volatile("s_barrier" ::: "memory");
  static_for<0, 16>([&](auto ic) {
    constexpr int j = decltype(ic)::value;
    CUM_FRAG(0, j, 0u);
  });
  CUM_WAIT_FRAGS(0);
  // The loop body is branch-free (with accumulators in asm operands every branch costs the allocator its grip on them): the
  // last two K-tiles re-fetch K-tile nk - 1 into the free parity and read fragments nobody uses.
#ifndef CUM_NT4_PA
#define CUM_NT4_PA 40      // MFMAs of phase A in front of the "units free" barrier (16 ... 48)
#endif
#ifndef CUM_NT4_PB
#define CUM_NT4_PB 24      // MFMAs of phase B in front of the "next K-tile landed" barrier (0 ... 48)
#endif
#ifdef CUM_NT4_NOBAR       // timing-only: no barriers (races)
#define CUM_NT4_BAR() asm volatile("" ::: "memory")
#else
#define CUM_NT4_BAR() asm volatile("s_barrier" ::: "memory")
#endif
#ifdef CUM_NT4_NODMA        // timing-only: no LDS-DMA in the loop
#define CUM_DMA_L(u, it, kt, par) asm volatile("" ::: "memory")
#else
#define CUM_DMA_L(u, it, kt, par) CUM_DMA(u, it, kt, par)
#endif
#ifdef CUM_NT4_NOREAD       // timing-only: no fragment reads in the loop
#define CUM_FRAG_L(KS, J, pb) asm volatile("" ::: "memory")
#else
#define CUM_FRAG_L(KS, J, pb) CUM_FRAG(KS, J, pb)
#endif
  constexpr int PA = CUM_NT4_PA, PB = CUM_NT4_PB;
  for (int kt = 0; kt < nk; ++kt) {
    const int par = kt & 1;
    const unsigned pb = (unsigned)par * (4 * UNIT * 16), pn = pb ^ (4 * UNIT * 16);
    const int kt2 = kt + 2 < nk ? kt + 2 : nk - 1;
    // ---- phase A: K-half 0
    static_for<0, PA>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      CUM_MFMA(0, i >> 3, i & 7);
      if constexpr (i < 16) CUM_FRAG_L(1, i, pb);
    });
    CUM_WAIT_FRAGS(1);
    CUM_NT4_BAR();                                                       // this parity's units are free
    static_for<PA, PA + 16>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      CUM_MFMA(0, i >> 3, i & 7);
      CUM_DMA_L((i - PA) >> 2, (i - PA) & 3, kt2, par);
    });
    static_for<PA + 16, 64>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      CUM_MFMA(0, i >> 3, i & 7);
    });
    // ---- phase B: K-half 1
    static_for<0, PB>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      CUM_MFMA(1, i >> 3, i & 7);
    });
    #ifndef CUM_NT4_NODMA
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#endif
    CUM_NT4_BAR();                                                       // K-tile kt + 1 is visible to every wave
    static_for<PB, PB + 16>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      CUM_MFMA(1, i >> 3, i & 7);
      CUM_FRAG_L(0, i - PB, pn);
    });
    static_for<PB + 16, 64>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      CUM_MFMA(1, i >> 3, i & 7);
    });
    CUM_WAIT_FRAGS(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef CUM_DSR
#undef CUM_WAIT_FRAGS
#undef CUM_MFMA
#undef CUM_FRAG
#undef CUM_DMA
  // the last MFMAs' results must be in the AGPRs before anything the compiler emits reads them (it sees asm outputs as
  // ready at once): the blocks of the last eight MFMAs go through the nops
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_barrier"
               : "+a"(acc[7][0]), "+a"(acc[7][1]), "+a"(acc[7][2]), "+a"(acc[7][3]), "+a"(acc[7][4]), "+a"(acc[7][5]),
                 "+a"(acc[7][6]), "+a"(acc[7][7]) : : "memory");
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float bv[4][4];
    nt_load_bias(p, n0, 2 * wc + h, g, bv);
    f32x4 part[2][4][4];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) part[hh][ni][mi] = acc[4 * h + ni][4 * hh + mi];
    nt_epilogue_any<T, EPI, 2, 1>(p, part, bv, m0, n0, 2 * wr, 2 * wc + h, lane,
                                  reinterpret_cast<unsigned char *>(lds_all) + wave * nt_rows_lds(EPI));
  }
}
#endif  // CUM_AB

// ---------------------------------------------------------------- small-M variant (streaming hops)
// Launches with only a few dozen 128x128 tiles (M = streams x a handful of rows) leave most of the chip idle while
// every workgroup walks the whole K axis at one exposed DMA latency per step.  Here a workgroup owns a 64x64 tile and
// its four waves split the K steps among themselves (wave w takes steps w, w + 4, ...), each with its own
// double-buffered LDS stage; the four partial tiles meet in LDS and wave 0 runs the usual epilogue.  4x the tiles and
// 4x shorter K chains per tile; results differ from the 128x128 kernel only in summation order.
template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_splitk_kernel(const GemmParams p) {
  constexpr int EPC = Elem<T>::EPC, BK = 8 * EPC;
  constexpr int WSTG = 128 * 8;                        // 16-byte chunks of one wave's stage: 64 A rows + 64 W rows
  __shared__ uint4 lds_all[4 * 2 * WSTG];              // [wave][stage]: 128 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int NB = (p.N + 63) / 64;
  const int m0 = ((int)blockIdx.x / NB) * 64, n0 = ((int)blockIdx.x % NB) * 64;
  if (blockIdx.x == 0) {
    T *o = static_cast<T *>(p.out);
    T *x = (EPI != EPI_GLU && EPI != EPI_GLU_BWD && !(EPI == EPI_RELU && p.mask_bits)) ? static_cast<T *>(p.aux) : nullptr;
    for (int64_t i = threadIdx.x; i < p.zero_head; i += 256) {
      o[-1 - i] = Elem<T>::from_f(0.f);
      if (x) x[-1 - i] = Elem<T>::from_f(0.f);
    }
    const int64_t tail0 = (int64_t)p.M * p.ldc, tailx = (int64_t)p.M * p.ldz;
    for (int64_t i = threadIdx.x; i < p.zero_tail; i += 256) {
      o[tail0 + i] = Elem<T>::from_f(0.f);
      if (x) x[tailx + i] = Elem<T>::from_f(0.f);
    }
  }
  const T *A = static_cast<const T *>(p.A);
  const T *W = static_cast<const T *>(p.W);
  // a wave instruction fills 1 KB = 8 rows: instruction `it` covers rows 8 it .. 8 it + 7, lane L = (row, chunk)
  const T *ga[8], *gw[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = 8 * it + (lane >> 3), cphys = lane & 7, clog = cphys ^ (row & 7);
    int am = m0 + row, wr = n0 + row;
    am = am < p.M ? am : p.M - 1;
    wr = wr < p.N ? wr : p.N - 1;
    ga[it] = A + (int64_t)am * p.lda + clog * EPC;
    gw[it] = W + (int64_t)wr * p.ldw + clog * EPC;
  }
  typedef __attribute__((address_space(3))) void *lds_ptr;
  typedef const __attribute__((address_space(1))) void *glb_ptr;
  uint4 *mine = lds_all + wave * 2 * WSTG;
  auto issue = [&](int kt, int stage) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      __builtin_amdgcn_global_load_lds((glb_ptr)(ga[it] + kt * BK), (lds_ptr)(&mine[stage * WSTG + it * 64]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr)(gw[it] + kt * BK), (lds_ptr)(&mine[stage * WSTG + 512 + it * 64]), 16, 0, 0);
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bv[4][4];
  nt_load_bias(p, n0, 0, g, bv);
  const int nk = p.K / BK;
  int stage = 0;
  if (wave < nk) issue(wave, 0);
  for (int kt = wave; kt < nk; kt += 4) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stage landed (wave-private LDS: no barrier)
    if (kt + 4 < nk) issue(kt + 4, stage ^ 1);
    nt_compute<T>(mine + stage * WSTG, mine + stage * WSTG + 512, acc, 0, 0, g, r);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // fragment reads done before the stage is refilled
    stage ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- meet in LDS: waves 1..3 park their partial tiles (16 KB each, in their own stage area), wave 0 adds them
  f32x4 *park = reinterpret_cast<f32x4 *>(mine);
  if (wave != 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) park[(i * 4 + j) * 64 + lane] = acc[i][j];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(lds_all + w * 2 * WSTG);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] += src[(i * 4 + j) * 64 + lane];
  }
  nt_epilogue<T, EPI>(p, &acc, bv, m0, n0, 0, 0, g, r);
}

template <typename T>
static int launch_gemm_splitk(const GemmParams &p, int epi, hipStream_t st) {
  dim3 grid(((p.M + 63) / 64) * ((p.N + 63) / 64)), block(256);
  switch (epi) {
    case EPI_BIAS: hipLaunchKernelGGL((gemm_nt_splitk_kernel<T, EPI_BIAS>), grid, block, 0, st, p); break;
    case EPI_RELU: hipLaunchKernelGGL((gemm_nt_splitk_kernel<T, EPI_RELU>), grid, block, 0, st, p); break;
    case EPI_MASK: hipLaunchKernelGGL((gemm_nt_splitk_kernel<T, EPI_MASK>), grid, block, 0, st, p); break;
    case EPI_GLU_BWD: hipLaunchKernelGGL((gemm_nt_splitk_kernel<T, EPI_GLU_BWD>), grid, block, 0, st, p); break;
    default: hipLaunchKernelGGL((gemm_nt_splitk_kernel<T, EPI_GLU>), grid, block, 0, st, p); break;
  }
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

// ---------------------------------------------------------------- elementwise backward
// GLU backward on the packed pre-activation: Z [M][ldz] holds per 32 columns 16 a then 16 b;
// dOut [M][ldo] holds the 16 matching output channels per group.  dZ has Z's layout.
template <typename T>
__global__ void glu_bwd_kernel(const T *__restrict__ Z, const T *__restrict__ dO, T *__restrict__ dZ, int64_t M,
                               int ngroups, int64_t ldz, int64_t ldo, int n_out) {
  const int64_t total = M * ngroups * 4;  // one thread per (row, group, quad of 4 channels)
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int q = i & 3;
    const int64_t t = i >> 2;
    const int grp = t % ngroups;
    const int64_t m = t / ngroups;
    const int oc = grp * 16 + q * 4;
    float a[4], b[4], d[4] = {0.f, 0.f, 0.f, 0.f}, da[4], db[4];
    load4<T>(Z + m * ldz + grp * 32 + q * 4, a);
    load4<T>(Z + m * ldz + grp * 32 + 16 + q * 4, b);
    if (oc < n_out) load4<T>(dO + m * ldo + oc, d);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float s = sigmoidf_(b[j]);
      da[j] = d[j] * s;
      db[j] = d[j] * a[j] * s * (1.f - s);
    }
    store4<T>(dZ + m * ldz + grp * 32 + q * 4, da);
    store4<T>(dZ + m * ldz + grp * 32 + 16 + q * 4, db);
  }
}

// Same from the gate pre-activation alone: Bg [M][ldb] (16 per group) and the saved GLU output Y [M][ldy]:
// da = d * sig(b), db = d * y * (1 - sig(b)).  dZ [M][ldz] is written in the packed (16 a | 16 b) layout.
template <typename T>
__global__ void glu_bwd_gate_kernel(const T *__restrict__ Bg, const T *__restrict__ Y, const T *__restrict__ dO,
                                    T *__restrict__ dZ, int64_t M, int ngroups, int64_t ldb, int64_t ldy, int64_t ldo,
                                    int64_t ldz, int n_out) {
  const int64_t total = M * ngroups * 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int q = i & 3;
    const int64_t t = i >> 2;
    const int grp = t % ngroups;
    const int64_t m = t / ngroups;
    const int oc = grp * 16 + q * 4;
    float y[4] = {0.f, 0.f, 0.f, 0.f}, b[4], d[4] = {0.f, 0.f, 0.f, 0.f}, da[4], db[4];
    load4<T>(Bg + m * ldb + oc, b);
    if (oc < n_out) {
      load4<T>(dO + m * ldo + oc, d);
      load4<T>(Y + m * ldy + oc, y);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float s = sigmoidf_(b[j]);
      da[j] = d[j] * s;
      db[j] = d[j] * y[j] * (1.f - s);
    }
    store4<T>(dZ + m * ldz + grp * 32 + q * 4, da);
    store4<T>(dZ + m * ldz + grp * 32 + 16 + q * 4, db);
  }
}

// dZ = dOut * (Y > 0)   (Y = ReLU output before any residual add); 4 elements per thread
template <typename T>
__global__ void relu_bwd_kernel(const T *__restrict__ Y, const T *__restrict__ dO, T *__restrict__ dZ, int64_t M,
                                int ncol4, int64_t ldy, int64_t ldo, int64_t ldz, int64_t zero_head, int64_t zero_tail) {
  if (blockIdx.x == 0) {
    for (int64_t i = threadIdx.x; i < zero_head; i += blockDim.x) dZ[-1 - i] = Elem<T>::from_f(0.f);
    for (int64_t i = threadIdx.x; i < zero_tail; i += blockDim.x) dZ[M * ldz + i] = Elem<T>::from_f(0.f);
  }
  const int64_t total = M * ncol4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (i % ncol4) * 4;
    const int64_t m = i / ncol4;
    float y[4], d[4];
    load4<T>(Y + m * ldy + c, y);
    load4<T>(dO + m * ldo + c, d);
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = y[j] > 0.f ? d[j] : 0.f;
    store4<T>(dZ + m * ldz + c, d);
  }
}

// Column sums of X [M][ld] (first n columns) -> out[n] (f32), two deterministic stages.
template <typename T>
__global__ void colsum_stage1(const T *__restrict__ X, int64_t M, int n, int64_t ld, int rows_per_block,
                              float *__restrict__ part) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  r1 = r1 < M ? r1 : M;
  float s = 0.f;
  for (int64_t m = r0; m < r1; ++m) s += Elem<T>::to_f(X[m * ld + c]);
  part[(int64_t)blockIdx.y * n + c] = s;
}
__global__ void colsum_stage2(const float *__restrict__ part, int nparts, int n, float *__restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  float s = 0.f;
  for (int i = 0; i < nparts; ++i) s += part[(int64_t)i * n + c];
  out[c] = s;
}

#ifdef CUM_AB
template <typename T>
static int launch_gemm_nt8(const GemmParams &p, int epi, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    const int NB = (p.N + 255) / 256, MB = (p.M + 255) / 256;
    dim3 grid(8 * NB * ((MB + 7) / 8)), block(512);
    switch (epi) {
      case EPI_BIAS: hipLaunchKernelGGL((gemm_nt8_kernel<T, EPI_BIAS>), grid, block, 0, st, p); break;
      case EPI_RELU: hipLaunchKernelGGL((gemm_nt8_kernel<T, EPI_RELU>), grid, block, 0, st, p); break;
      case EPI_MASK: hipLaunchKernelGGL((gemm_nt8_kernel<T, EPI_MASK>), grid, block, 0, st, p); break;
      case EPI_GLU_BWD: hipLaunchKernelGGL((gemm_nt8_kernel<T, EPI_GLU_BWD>), grid, block, 0, st, p); break;
      default: hipLaunchKernelGGL((gemm_nt8_kernel<T, EPI_GLU>), grid, block, 0, st, p); break;
    }
    CUM_CHECK_LAUNCH();
  }
  return CUM_OK;
}

#endif  // CUM_AB

template <typename T>
static int launch_gemm_nt9(const GemmParams &p0, int epi, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    GemmParams p = p0;
    const int NB = (p.N + 255) / 256, MB = (p.M + 255) / 256;
    dim3 grid(8 * NB * ((MB + 7) / 8)), block(512);
    p.group_m = (int)cum_knob("CUM_NT_GROUPM", NB >= 24 ? 4 : 1);
    if (p.group_m < 1) p.group_m = 1;
#ifdef CUM_AB
#ifdef CUM_NT4_ONLY_BIAS                              // (tuning builds: one instantiation, a fifth of the compile time)
    if (cum_knob("CUM_NT4", 0) != 0 && epi == EPI_BIAS) {
      hipLaunchKernelGGL((gemm_nt4_kernel<T, EPI_BIAS>), grid, dim3(256), 0, st, p);
      CUM_CHECK_LAUNCH();
      return CUM_OK;
    }
#else
    if (cum_knob("CUM_NT4", 0) != 0) {                 // the four-wave experiment (gemm_nt4_kernel)
      const dim3 b4(256);
      switch (epi) {
        case EPI_BIAS: hipLaunchKernelGGL((gemm_nt4_kernel<T, EPI_BIAS>), grid, b4, 0, st, p); break;
        case EPI_RELU: hipLaunchKernelGGL((gemm_nt4_kernel<T, EPI_RELU>), grid, b4, 0, st, p); break;
        case EPI_MASK: hipLaunchKernelGGL((gemm_nt4_kernel<T, EPI_MASK>), grid, b4, 0, st, p); break;
        case EPI_GLU_BWD: hipLaunchKernelGGL((gemm_nt4_kernel<T, EPI_GLU_BWD>), grid, b4, 0, st, p); break;
        default: hipLaunchKernelGGL((gemm_nt4_kernel<T, EPI_GLU>), grid, b4, 0, st, p); break;
      }
      CUM_CHECK_LAUNCH();
      return CUM_OK;
    }
#endif
#endif
    switch (epi) {
      case EPI_BIAS: hipLaunchKernelGGL((gemm_nt9_kernel<T, EPI_BIAS>), grid, block, 0, st, p); break;
      case EPI_RELU: hipLaunchKernelGGL((gemm_nt9_kernel<T, EPI_RELU>), grid, block, 0, st, p); break;
      case EPI_MASK: hipLaunchKernelGGL((gemm_nt9_kernel<T, EPI_MASK>), grid, block, 0, st, p); break;
      case EPI_GLU_BWD: hipLaunchKernelGGL((gemm_nt9_kernel<T, EPI_GLU_BWD>), grid, block, 0, st, p); break;
      default: hipLaunchKernelGGL((gemm_nt9_kernel<T, EPI_GLU>), grid, block, 0, st, p); break;
    }
    CUM_CHECK_LAUNCH();
  }
  return CUM_OK;
}

template <typename T>
static int launch_gemm_ring(const GemmParams &p, int epi, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    const int NB = (p.N + 255) / 256, MB = (p.M + 127) / 128;
    dim3 grid(8 * NB * ((MB + 7) / 8)), block(512);
    switch (epi) {
      case EPI_BIAS: hipLaunchKernelGGL((gemm_nt_ring_kernel<T, EPI_BIAS>), grid, block, 0, st, p); break;
      case EPI_RELU: hipLaunchKernelGGL((gemm_nt_ring_kernel<T, EPI_RELU>), grid, block, 0, st, p); break;
      case EPI_MASK: hipLaunchKernelGGL((gemm_nt_ring_kernel<T, EPI_MASK>), grid, block, 0, st, p); break;
      case EPI_GLU_BWD: hipLaunchKernelGGL((gemm_nt_ring_kernel<T, EPI_GLU_BWD>), grid, block, 0, st, p); break;
      default: hipLaunchKernelGGL((gemm_nt_ring_kernel<T, EPI_GLU>), grid, block, 0, st, p); break;
    }
    CUM_CHECK_LAUNCH();
  }
  return CUM_OK;
}

template <typename T, int BM, int BN>
static int launch_gemm_tile(const GemmParams &p, int epi, hipStream_t st) {
  const int NB = (p.N + BN - 1) / BN, MB = (p.M + BM - 1) / BM;
  dim3 grid(8 * NB * ((MB + 7) / 8)), block(BM * BN / 64);
  switch (epi) {
    case EPI_BIAS: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_BIAS, BM, BN>), grid, block, 0, st, p); break;
    case EPI_RELU: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_RELU, BM, BN>), grid, block, 0, st, p); break;
    case EPI_MASK: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_MASK, BM, BN>), grid, block, 0, st, p); break;
    case EPI_GLU_BWD: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_GLU_BWD, BM, BN>), grid, block, 0, st, p); break;
    default: hipLaunchKernelGGL((gemm_nt_kernel<T, EPI_GLU, BM, BN>), grid, block, 0, st, p); break;
  }
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

// Tile choice of cum_gemm_nt for one problem (esz: element size).  64 = the small-M kernel (64 x 64 tiles, K split over
// the four waves), 128 = 128 x 128, 256 = 256 x 128 (f32), 384 = 128 x 256 through the three-stage ring (16-bit, few tiles),
// 512 = 256 x 256 with the two wave groups in ping-pong (16-bit).
// cum_gemm_nt_tile() reports it, so tests can tell which kernel a shape is verified on.
static int choose_tile(const GemmParams &p, int esz) {
  const int64_t mb256 = (p.M + 255) / 256;
  const int64_t tiles_256x256 = mb256 * ((p.N + 255) / 256), tiles_256x128 = mb256 * ((p.N + 127) / 128);
  // AB build: CUM_NT_TILE=64|128|256|512 pins the tile (split-K 64x64 / 128x128 / 256x128 / 256x256)
  int tile = (int)cum_knob("CUM_NT_TILE", 0);
  // few tiles and a K axis worth splitting: the small-M kernel (64x64 tiles, K split over the four waves)
  const int64_t tiles_128 = ((p.M + 127) / 128) * ((p.N + 127) / 128);
  const int bk = esz == 2 ? 64 : 32;
  // (allow_split_k == 2: the caller asks for it whatever the tile count -- the narrow Mamba projections, N <= 256 with
  //  K = 2048, where 128-wide tiles waste half of their columns and 78 row tiles do not fill the chip)
  if ((tile == 64 || (!tile && (p.allow_split_k == 2 || (p.allow_split_k && tiles_128 <= 64)))) && p.K >= 4 * bk) return 64;
  if (tile == 64) tile = 128;
  if (!tile) {
    // 256x256 (one workgroup per CU) once it fills the chip and N wastes little of the 256-wide tile; K >= 256 so the
    // saved weight traffic matters (the outer layers are bound by their activation traffic, where the tile shape is
    // irrelevant) ... and enough work per byte for one workgroup per CU to pay off: at N K / (N + K) < 256 (the 256 /
    // 512-channel layers with 320 512 rows) four 128 x 128 workgroups per CU are 5-18 % faster (same-box per-call table)
    if (esz == 2 && p.K >= 256 && tiles_256x256 >= 224 && p.N % 256 == 0 &&
        (int64_t)p.N * p.K >= 256 * (int64_t)(p.N + p.K)) tile = 512;
    // f32 (the parity path): 256-row tiles while they still give every CU >= 2 workgroups per XCD-round.  16-bit types
    // never take this tile: the 128 x 128 kernel routes its epilogue through LDS, which the outer, HBM-bound layers gain
    // more from than from the taller tile.
    else if (esz == 4 && p.K >= 256 && tiles_256x128 >= 1024) tile = 256;
    // few tiles, long K (M ~ 10 000 with N = 512 / 768): 128 x 256 tiles through the three-stage ring, one resident round
    // (from 40 tiles up: below that -- batch-1 inference, M ~ 600 -- twice as many 128 x 128 workgroups measured 1-2 % ahead)
    else if (esz == 2 && p.N % 256 == 0 && p.K >= 512 && ((p.M + 127) / 128) * (int64_t)(p.N / 256) <= cum_knob("CUM_NT_RING", 256) &&
             ((p.M + 127) / 128) * (int64_t)(p.N / 256) >= 40)
      tile = 384;
    else tile = 128;
  }
  if (esz == 2) {
#ifndef CUM_AB
    if (tile == 256) tile = 128;
#endif
    return tile;
  }
  return tile == 512 ? 256 : tile;
}

template <typename T>
static int launch_gemm(const GemmParams &p, int epi, hipStream_t st) {
  const int tile = choose_tile(p, (int)sizeof(T));
  if (tile == 64) return launch_gemm_splitk<T>(p, epi, st);
  if constexpr (sizeof(T) == 2) {
    if (tile == 512) {
#ifdef CUM_AB
      if (cum_knob("CUM_NT9", 1) == 0) return launch_gemm_nt8<T>(p, epi, st);
#endif
      return launch_gemm_nt9<T>(p, epi, st);
    }
    if (tile == 384) return launch_gemm_ring<T>(p, epi, st);
#ifdef CUM_AB
    if (tile == 256) return launch_gemm_tile<T, 256, 128>(p, epi, st);
#endif
    return launch_gemm_tile<T, 128, 128>(p, epi, st);
  } else {
    if (tile == 256) return launch_gemm_tile<T, 256, 128>(p, epi, st);
    return launch_gemm_tile<T, 128, 128>(p, epi, st);
  }
}

}  // namespace cum

using namespace cum;

extern "C" int cum_gemm_nt(const cum_gemm_desc *d, const void *A, const void *W, const float *bias, const void *res,
                           void *out, void *aux, const void *aux2, void *stream) {
  CUM_REQUIRE(d && A && W && out, "gemm: null argument");
  CUM_REQUIRE(dtype_ok(d->dtype), "gemm: dtype must be CUM_F32, CUM_BF16 or CUM_F16");
  CUM_REQUIRE(d->epilogue >= 0 && d->epilogue <= 4, "gemm: bad epilogue");
  CUM_REQUIRE(d->epilogue != EPI_MASK || res, "gemm: the MASK epilogue needs the gating activation in res");
  CUM_REQUIRE(d->epilogue != EPI_GLU_BWD || (aux && d->zero_head == 0 && d->zero_tail == 0),
              "gemm: the GLU_BWD epilogue needs Z in aux and writes no framing rows");
  const int bk = is16(d->dtype) ? 64 : 32;
  const int epc = is16(d->dtype) ? 8 : 4;
  CUM_REQUIRE(d->M >= 0 && d->N > 0 && d->K > 0 && d->K % bk == 0, "gemm: K must be a positive multiple of the K tile");
  CUM_REQUIRE(d->N % (d->epilogue == 2 ? 32 : 16) == 0, "gemm: N must be a multiple of 16 (32 for GLU)");
  CUM_REQUIRE(d->lda % epc == 0 && d->ldw % epc == 0, "gemm: lda/ldw must keep rows 16-byte aligned");
  CUM_REQUIRE(d->ldc % 4 == 0 && d->ldr % 4 == 0 && d->ldz % 4 == 0 && d->n_store % 4 == 0, "gemm: ldc/ldr/ldz/n_store must be multiples of 4");
  CUM_REQUIRE(d->pitch > 0 && d->valid >= 0 && d->zero_head >= 0 && d->zero_tail >= 0, "gemm: bad pitch/valid/zero ranges");
  CUM_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)bias & 15) == 0,
              "gemm: A, W and bias must be 16-byte aligned");
  if (d->M == 0) return CUM_OK;
  GemmParams p{};
  CUM_REQUIRE(!d->gate_only || d->epilogue == EPI_GLU || (d->epilogue == EPI_GLU_BWD && aux2 && d->ldy % 4 == 0),
              "gemm: gate_only applies to the GLU epilogues; GLU_BWD then needs the saved output in aux2");
  p.A = A; p.W = W; p.bias = bias; p.res = res; p.out = out; p.aux = aux; p.aux2 = aux2;
  p.lda = d->lda; p.ldw = d->ldw; p.ldc = d->ldc; p.ldr = d->ldr; p.ldz = d->ldz; p.ldy = d->ldy;
  p.gate_only = d->gate_only;
  p.mask_bits = d->mask_bits;
  p.allow_split_k = d->allow_split_k;
  CUM_REQUIRE(!d->mask_bits || (d->epilogue == EPI_RELU && aux) || d->epilogue == EPI_MASK,
              "gemm: mask_bits applies to RELU (aux = sign array) and MASK (res = sign array)");
  p.M = d->M; p.N = d->N; p.K = d->K; p.pitch = d->pitch; p.valid = d->valid; p.n_store = d->n_store;
  p.zero_head = d->zero_head; p.zero_tail = d->zero_tail;
  p.rows_epilogue = (int)cum_knob("CUM_NT8_ROWS", 1);      // AB build: 0 = the generic GLU-backward epilogue
  if (d->dtype == CUM_BF16) return launch_gemm<__bf16>(p, d->epilogue, (hipStream_t)stream);
  if (d->dtype == CUM_F16) return launch_gemm<f16>(p, d->epilogue, (hipStream_t)stream);
  return launch_gemm<float>(p, d->epilogue, (hipStream_t)stream);
}

extern "C" int cum_gemm_nt_tile(const cum_gemm_desc *d) {
  CUM_REQUIRE(d && dtype_ok(d->dtype), "gemm_nt_tile: bad argument");
  GemmParams p{};
  p.M = d->M; p.N = d->N; p.K = d->K; p.allow_split_k = d->allow_split_k;
  return choose_tile(p, is16(d->dtype) ? 2 : 4);
}

extern "C" int cum_glu_bwd_gate(int32_t dtype, int64_t M, int32_t n_groups, int32_t n_out, const void *Bg, int64_t ldb,
                                const void *Y, int64_t ldy, const void *dOut, int64_t ldo, void *dZ, int64_t ldz,
                                void *stream) {
  CUM_REQUIRE(Bg && Y && dOut && dZ && n_groups > 0 && M >= 0, "glu_bwd_gate: bad argument");
  CUM_REQUIRE(ldb % 4 == 0 && ldy % 4 == 0 && ldo % 4 == 0 && ldz % 4 == 0, "glu_bwd_gate: strides must be multiples of 4");
  if (M == 0) return CUM_OK;
  const int64_t total = M * n_groups * 4;
  int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (dtype == CUM_BF16)
    hipLaunchKernelGGL(glu_bwd_gate_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)Bg,
                       (const __bf16 *)Y, (const __bf16 *)dOut, (__bf16 *)dZ, M, n_groups, ldb, ldy, ldo, ldz, n_out);
  else if (dtype == CUM_F16)
    hipLaunchKernelGGL(glu_bwd_gate_kernel<f16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16 *)Bg,
                       (const f16 *)Y, (const f16 *)dOut, (f16 *)dZ, M, n_groups, ldb, ldy, ldo, ldz, n_out);
  else
    hipLaunchKernelGGL(glu_bwd_gate_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float *)Bg,
                       (const float *)Y, (const float *)dOut, (float *)dZ, M, n_groups, ldb, ldy, ldo, ldz, n_out);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

extern "C" int cum_glu_bwd(int32_t dtype, int64_t M, int32_t n_groups, int32_t n_out, const void *Z, int64_t ldz,
                           const void *dOut, int64_t ldo, void *dZ, void *stream) {
  CUM_REQUIRE(Z && dOut && dZ && n_groups > 0 && M >= 0, "glu_bwd: bad argument");
  CUM_REQUIRE(ldz % 4 == 0 && ldo % 4 == 0, "glu_bwd: strides must be multiples of 4");
  if (M == 0) return CUM_OK;
  const int64_t total = M * n_groups * 4;
  int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (dtype == CUM_BF16)
    hipLaunchKernelGGL(glu_bwd_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)Z,
                       (const __bf16 *)dOut, (__bf16 *)dZ, M, n_groups, ldz, ldo, n_out);
  else if (dtype == CUM_F16)
    hipLaunchKernelGGL(glu_bwd_kernel<f16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16 *)Z,
                       (const f16 *)dOut, (f16 *)dZ, M, n_groups, ldz, ldo, n_out);
  else
    hipLaunchKernelGGL(glu_bwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float *)Z,
                       (const float *)dOut, (float *)dZ, M, n_groups, ldz, ldo, n_out);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

extern "C" int cum_relu_bwd(int32_t dtype, int64_t M, int32_t n_cols, const void *Y, int64_t ldy, const void *dOut,
                            int64_t ldo, void *dZ, int64_t ldz, int64_t zero_head, int64_t zero_tail, void *stream) {
  CUM_REQUIRE(Y && dOut && dZ && n_cols > 0 && n_cols % 4 == 0 && M >= 0, "relu_bwd: bad argument");
  CUM_REQUIRE(ldy % 4 == 0 && ldo % 4 == 0 && ldz % 4 == 0, "relu_bwd: strides must be multiples of 4");
  if (M == 0) return CUM_OK;
  const int64_t total = M * (n_cols / 4);
  int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (dtype == CUM_BF16)
    hipLaunchKernelGGL(relu_bwd_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)Y,
                       (const __bf16 *)dOut, (__bf16 *)dZ, M, n_cols / 4, ldy, ldo, ldz, zero_head, zero_tail);
  else if (dtype == CUM_F16)
    hipLaunchKernelGGL(relu_bwd_kernel<f16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16 *)Y,
                       (const f16 *)dOut, (f16 *)dZ, M, n_cols / 4, ldy, ldo, ldz, zero_head, zero_tail);
  else
    hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float *)Y,
                       (const float *)dOut, (float *)dZ, M, n_cols / 4, ldy, ldo, ldz, zero_head, zero_tail);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

extern "C" int64_t cum_colsum_workspace_elems(int64_t M, int32_t n_cols) {
  const int64_t parts = (M + 1023) / 1024;
  return parts * n_cols;
}

extern "C" int cum_colsum(int32_t dtype, int64_t M, int32_t n_cols, const void *X, int64_t ld, float *out,
                          float *workspace, void *stream) {
  CUM_REQUIRE(X && out && workspace && n_cols > 0 && M >= 0, "colsum: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    (void)hipMemsetAsync(out, 0, sizeof(float) * n_cols, st);
    return CUM_OK;
  }
  const int parts = (int)((M + 1023) / 1024);
  dim3 grid((n_cols + 63) / 64, parts), block(64);
  if (dtype == CUM_BF16)
    hipLaunchKernelGGL(colsum_stage1<__bf16>, grid, block, 0, st, (const __bf16 *)X, M, n_cols, ld, 1024, workspace);
  else if (dtype == CUM_F16)
    hipLaunchKernelGGL(colsum_stage1<f16>, grid, block, 0, st, (const f16 *)X, M, n_cols, ld, 1024, workspace);
  else
    hipLaunchKernelGGL(colsum_stage1<float>, grid, block, 0, st, (const float *)X, M, n_cols, ld, 1024, workspace);
  CUM_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_stage2, dim3((n_cols + 63) / 64), dim3(64), 0, st, workspace, parts, n_cols, out);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}
