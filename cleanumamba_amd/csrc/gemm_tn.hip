// Weight-gradient GEMM (split over the row axis) with fused bias gradient, for gfx950.
//
//   dW[n][k] = sum_m dZ[m][n] * X[m*ldx + k]        db[n] = sum_m dZ[m][n]
//
// i.e. the backward-weights pass of every encoder / decoder layer (autograd of the
// reference's nn.Conv1d / nn.ConvTranspose1d, src/network/CleanUMamba.py:108-130 under
// src/training/train.py:282-285).  X rows may overlap exactly as in gemm.hip, so the
// k=4/s=2 conv and transposed-conv weight gradients are single launches.
//
// The reduction runs over m (B*T, up to 1.3 M) while the output is small (<= 1536 x 3072),
// so the m axis is split across workgroups (grid.z); every split writes an f32 slab and a
// second kernel adds the slabs in a fixed order (deterministic, no float atomics).
// Both operands are stored m-major, i.e. "transposed" for the MFMA, whose lanes want 8
// consecutive reduction indices: bf16 tiles are staged row-major in LDS and read with
// ds_read_b64_tr_b16 (the CDNA4 transposing LDS read); f32 tiles map directly onto
// v_mfma_f32_16x16x4_f32 (one reduction index per lane).  X is the MFMA "A" operand so a
// lane ends with 4 consecutive k of one output row n: 16-byte slab stores.
// Column sums of dZ (the bias gradient) are accumulated from the registers that stage
// dZ, for free, by the workgroups of the first k tile.
#include "common.h"

namespace cum {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct TnParams {
  const void *dZ, *X;
  float *slab;       // [S][Np][Kp]
  float *bslab;      // [S][Np] or null
  int64_t ldz, ldx;
  int64_t M;
  int N, K;          // valid columns of dZ / of an X row (multiples of 4)
  int Np, Kp;        // slab dims (multiples of 128)
  int rows_per_split;
};

constexpr int TN_T = 128;  // output tile (n and k)

template <typename T>
struct TnCfg;
template <>
struct TnCfg<__bf16> {
  static constexpr int EPC = 8, BMK = 32;
};
template <>
struct TnCfg<float> {
  static constexpr int EPC = 4, BMK = 16;
};

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_tn_kernel(const TnParams p) {
  constexpr int EPC = TnCfg<T>::EPC, BMK = TnCfg<T>::BMK;
  constexpr int CPR = TN_T / EPC;          // 16-byte chunks per tile row: 16 (bf16) / 32 (f32)
  constexpr int NCH = BMK * CPR / 256;     // chunks per thread per operand: 2 (bf16: 512) / 2 (f32: 512)
  constexpr int RS = TN_T * sizeof(T) + 16;  // LDS row stride in bytes (+16 B pad breaks the power-of-two stride)
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][BMK * RS];  // [stage][0: dZ, 1: X]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, r = lane & 15;
  const int n0 = blockIdx.x * TN_T, k0 = blockIdx.y * TN_T, sp = blockIdx.z;
  const int64_t m_begin = (int64_t)sp * p.rows_per_split;
  int64_t m_end = m_begin + p.rows_per_split;
  m_end = m_end < p.M ? m_end : p.M;
  const T *dZ = static_cast<const T *>(p.dZ);
  const T *X = static_cast<const T *>(p.X);
  const bool do_bias = p.bslab != nullptr && blockIdx.y == 0;

  // staging: chunk c = tid + 256*i -> row c / CPR, column chunk c % CPR (the same for every i)
  const int cc = tid % CPR;
  int zcol = n0 + cc * EPC, xcol = k0 + cc * EPC;
  const bool zok = zcol < p.N, xok = xcol < p.K;
  zcol = zok ? zcol : 0;
  xcol = xok ? xcol : 0;
  uint4 rz[NCH], rx[NCH];
  float bsum[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) bsum[e] = 0.f;

  auto gload = [&](int64_t mb) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int row = (tid + 256 * i) / CPR;
      const int64_t m = mb + row;
      const bool ok = m < m_end;
      const int64_t mc = ok ? m : m_end - 1;
      uint4 vz = *reinterpret_cast<const uint4 *>(dZ + mc * p.ldz + zcol);
      uint4 vx = *reinterpret_cast<const uint4 *>(X + mc * p.ldx + xcol);
      if (!(ok && zok)) vz = make_uint4(0, 0, 0, 0);
      if (!(ok && xok)) vx = make_uint4(0, 0, 0, 0);
      rz[i] = vz;
      rx[i] = vx;
    }
  };
  auto lstore = [&](int st) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int row = (tid + 256 * i) / CPR;
      *reinterpret_cast<uint4 *>(&lds[st][0][row * RS + cc * 16]) = rz[i];
      *reinterpret_cast<uint4 *>(&lds[st][1][row * RS + cc * 16]) = rx[i];
      if (do_bias) {
        if constexpr (sizeof(T) == 2) {
          const bf16x8 v = __builtin_bit_cast(bf16x8, rz[i]);
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum[e] += (float)v[e];
        } else {
          bsum[0] += __builtin_bit_cast(float, rz[i].x);
          bsum[1] += __builtin_bit_cast(float, rz[i].y);
          bsum[2] += __builtin_bit_cast(float, rz[i].z);
          bsum[3] += __builtin_bit_cast(float, rz[i].w);
        }
      }
    }
  };

  f32x4 acc[4][4];  // [ki][ni]: D[i = k][j = n]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int64_t nsteps = (m_end - m_begin + BMK - 1) / BMK;
  if (nsteps > 0) {
    gload(m_begin);
    lstore(0);
  }
  __syncthreads();
  for (int64_t s = 0; s < nsteps; ++s) {
    const int st = s & 1;
    if (s + 1 < nsteps) gload(m_begin + (s + 1) * BMK);
    if constexpr (sizeof(T) == 2) {
      // transposing reads: the 16 lanes of group g fetch a [4 rows][16 cols] block; lane (q, p) = (r>>2, r&3)
      // addresses row q, cols 4p..4p+3 and receives column r of the 4 rows
      const int q = r >> 2, pp = r & 3;
      bf16x8 xf[4], zf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int xc = wk * 64 + i * 16 + 4 * pp, zc = wn * 64 + i * 16 + 4 * pp;
        bf16x4 lo, hi;
        typedef __attribute__((address_space(3))) bf16x4 *lp;
        lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(&lds[st][1][(8 * g + q) * RS + xc * 2]));
        hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(&lds[st][1][(8 * g + 4 + q) * RS + xc * 2]));
        xf[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(&lds[st][0][(8 * g + q) * RS + zc * 2]));
        hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(&lds[st][0][(8 * g + 4 + q) * RS + zc * 2]));
        zf[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int ki = 0; ki < 4; ++ki)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[ki], zf[ni], acc[ki][ni], 0, 0, 0);
    } else {
#pragma unroll
      for (int ss = 0; ss < BMK / 4; ++ss) {
        float xf[4], zf[4];
        const int row = 4 * ss + g;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xf[i] = *reinterpret_cast<const float *>(&lds[st][1][row * RS + (wk * 64 + i * 16 + r) * 4]);
          zf[i] = *reinterpret_cast<const float *>(&lds[st][0][row * RS + (wn * 64 + i * 16 + r) * 4]);
        }
#pragma unroll
        for (int ki = 0; ki < 4; ++ki)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[ki], zf[ni], acc[ki][ni], 0, 0, 0);
      }
    }
    if (s + 1 < nsteps) lstore(st ^ 1);
    __syncthreads();
  }

  // ---- slab store: lane holds D[k = kb + 4g + j][n = nb + r]
  float *slab = p.slab + (int64_t)sp * p.Np * p.Kp;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = n0 + wn * 64 + ni * 16 + r;
#pragma unroll
    for (int ki = 0; ki < 4; ++ki) {
      const int k = k0 + wk * 64 + ki * 16 + 4 * g;
      *reinterpret_cast<float4 *>(slab + (int64_t)n * p.Kp + k) =
          make_float4(acc[ki][ni][0], acc[ki][ni][1], acc[ki][ni][2], acc[ki][ni][3]);
    }
  }
  // ---- bias gradient: threads with the same column chunk (tid % CPR) hold partial sums
  if (do_bias) {
    __syncthreads();
    float *red = reinterpret_cast<float *>(&lds[0][0][0]);  // [256 / CPR][128]
    const int tr = tid / CPR;
#pragma unroll
    for (int e = 0; e < EPC; ++e) red[tr * TN_T + cc * EPC + e] = bsum[e];
    __syncthreads();
    if (tid < TN_T) {
      float s = 0.f;
      for (int j = 0; j < 256 / CPR; ++j) s += red[j * TN_T + tid];
      p.bslab[(int64_t)sp * p.Np + n0 + tid] = s;
    }
  }
}

// out[n][k] = sum_s slab[s][n][k] for n < N, k < K; bias likewise.
__global__ void tn_reduce_kernel(const float *__restrict__ slab, const float *__restrict__ bslab, int S, int Np, int Kp,
                                 int N, int K, float *__restrict__ out, int64_t ldo, float *__restrict__ bout) {
  const int64_t total = (int64_t)N * (K / 4);
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < total) {
    const int n = i / (K / 4), k = (i % (K / 4)) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sp = 0; sp < S; ++sp) {
      const float4 v = *reinterpret_cast<const float4 *>(slab + ((int64_t)sp * Np + n) * Kp + k);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4 *>(out + (int64_t)n * ldo + k) = s;
  }
  if (bout && i < N) {
    float s = 0.f;
    for (int sp = 0; sp < S; ++sp) s += bslab[(int64_t)sp * Np + i];
    bout[i] = s;
  }
}

}  // namespace cum

using namespace cum;

static void tn_plan(int64_t M, int32_t N, int32_t K, int32_t dtype, int *Np, int *Kp, int *S, int *rps) {
  *Np = (N + TN_T - 1) / TN_T * TN_T;
  *Kp = (K + TN_T - 1) / TN_T * TN_T;
  const int tiles = (*Np / TN_T) * (*Kp / TN_T);
  const int bmk = dtype == CUM_BF16 ? 32 : 16;
  int64_t want = (1024 + tiles - 1) / tiles;            // ~4 workgroups per CU in total
  const int64_t max_s = (M + 8 * bmk - 1) / (8 * bmk);  // at least 8 steps per split
  if (want > max_s) want = max_s;
  if (want < 1) want = 1;
  int64_t rows = (M + want - 1) / want;
  rows = (rows + bmk - 1) / bmk * bmk;
  *rps = (int)rows;
  *S = (int)((M + rows - 1) / rows);
  if (*S < 1) *S = 1;
}

extern "C" int64_t cum_gemm_tn_workspace_elems(int32_t dtype, int64_t M, int32_t N, int32_t K) {
  int Np, Kp, S, rps;
  tn_plan(M, N, K, dtype, &Np, &Kp, &S, &rps);
  return (int64_t)S * Np * Kp + (int64_t)S * Np;
}

extern "C" int cum_gemm_tn(int32_t dtype, int64_t M, int32_t N, int32_t K, const void *dZ, int64_t ldz, const void *X,
                           int64_t ldx, float *dW, int64_t ldw, float *db, float *workspace, void *stream) {
  CUM_REQUIRE(dtype == CUM_F32 || dtype == CUM_BF16, "gemm_tn: dtype must be CUM_F32 or CUM_BF16");
  CUM_REQUIRE(dZ && X && dW && workspace && M >= 0 && N > 0 && K > 0, "gemm_tn: bad argument");
  const int epc = dtype == CUM_BF16 ? 8 : 4;
  CUM_REQUIRE(N % epc == 0 && K % epc == 0 && ldz % epc == 0 && ldx % epc == 0 && ldw % 4 == 0,
              "gemm_tn: N, K and strides must keep 16-byte alignment");
  CUM_REQUIRE(((uintptr_t)dZ & 15) == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)dW & 15) == 0,
              "gemm_tn: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    (void)hipMemset2DAsync(dW, sizeof(float) * ldw, 0, sizeof(float) * K, N, st);
    if (db) (void)hipMemsetAsync(db, 0, sizeof(float) * N, st);
    return CUM_OK;
  }
  int Np, Kp, S, rps;
  tn_plan(M, N, K, dtype, &Np, &Kp, &S, &rps);
  TnParams p{};
  p.dZ = dZ; p.X = X; p.ldz = ldz; p.ldx = ldx; p.M = M; p.N = N; p.K = K; p.Np = Np; p.Kp = Kp;
  p.rows_per_split = rps;
  p.slab = workspace;
  p.bslab = db ? workspace + (int64_t)S * Np * Kp : nullptr;
  dim3 grid(Np / TN_T, Kp / TN_T, S), block(256);
  if (dtype == CUM_BF16)
    hipLaunchKernelGGL(gemm_tn_kernel<__bf16>, grid, block, 0, st, p);
  else
    hipLaunchKernelGGL(gemm_tn_kernel<float>, grid, block, 0, st, p);
  CUM_CHECK_LAUNCH();
  const int64_t total = (int64_t)N * (K / 4);
  const int64_t threads = total > N ? total : N;
  hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, p.slab, p.bslab, S, Np,
                     Kp, N, K, dW, ldw, db);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}
