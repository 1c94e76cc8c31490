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
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct TnParams {
  const void *dZ, *X;
  float *slab;       // [S][Np][Kp]
  float *bslab;      // [S][Np] or null
  int64_t ldz, ldx;
  int64_t M;
  int N, K;          // valid columns of dZ / of an X row (multiples of 4)
  int Np, Kp;        // slab dims (multiples of 128)
  int rows_per_split, nsplit;
#ifdef CUM_AB
  int skip_store;    // timing experiment (CUM_TN_NOSTORE=1, tools/tn_intercept.py): the slabs are not written
#endif
};

constexpr int TN_T = 128;  // output tile (n and k)

template <typename T>
struct TnCfg;
template <>
struct TnCfg<__bf16> {
  static constexpr int EPC = 8, BMK = 64;   // 64 rows per step = two 32-deep MFMA reductions
  // 16-byte chunk swizzle (in chunks): consecutive rows, and rows 8 apart, land in different 32-byte slots so the
  // 4-row x 32-byte blocks fetched by ds_read_b64_tr_b16 do not collide; XOR with an even number keeps each
  // 32-byte block (two chunks) together.
  static __device__ __forceinline__ int swz(int row) { return 2 * ((row & 3) | (((row >> 3) & 1) << 2)); }
};
template <>
struct TnCfg<f16> : TnCfg<__bf16> {};
template <>
struct TnCfg<float> {
  static constexpr int EPC = 4, BMK = 32;
  static __device__ __forceinline__ int swz(int row) { return 4 * (row & 7); }  // 64-byte shifts
};

template <typename T>
// (f32, the parity path, needs a few registers more than the 168 of three waves per SIMD: two there instead of spills)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 4 ? 2 : 3, 3))) void gemm_tn_kernel(const TnParams p) {
  constexpr int EPC = TnCfg<T>::EPC, BMK = TnCfg<T>::BMK;
  constexpr int CPR = TN_T / EPC;          // 16-byte chunks per tile row: 16 (bf16) / 32 (f32)
  constexpr int NCH = BMK * CPR / 256;     // chunks per thread per operand: 4
  __shared__ uint4 lds[2][BMK * CPR];      // [0: dZ, 1: X][row * CPR + physical chunk]; 32 KB, single-buffered

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = uniform(wave);
  const int wk = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, r = lane & 15;
  // XCD-aware order (ids b and b+8 share an XCD / L2): all (n, k) tiles of one row split run back to back on one
  // XCD, so its dZ and X rows come from HBM once and are re-read from that L2 by the other tiles.
  const int ntn = p.Np / TN_T, ntk = p.Kp / TN_T, tiles = ntn * ntk;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int sp = (local / tiles) * 8 + xcd;
  if (sp >= p.nsplit) return;
  const int tile = local % tiles;
  const int n0 = (tile % ntn) * TN_T, k0 = (tile / ntn) * TN_T;
  const int64_t m_begin = (int64_t)sp * p.rows_per_split;
  int64_t m_end = m_begin + p.rows_per_split;
  m_end = m_end < p.M ? m_end : p.M;
  const T *dZ = static_cast<const T *>(p.dZ);
  const T *X = static_cast<const T *>(p.X);
  const bool do_bias = p.bslab != nullptr && k0 == 0;

  // HBM -> LDS directly: the thread's i-th DMA lands at linear chunk position i*256 + tid = (row, physical chunk);
  // it fetches the LOGICAL chunk (physical ^ swz(row)) so that reads can use the swizzled address.
  typedef __attribute__((address_space(3))) void *lds_ptr;
  typedef const __attribute__((address_space(1))) void *glb_ptr;
  int lrow[NCH], zoff[NCH], xoff[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int pos = i * 256 + tid;
    lrow[i] = pos / CPR;
    const int clog = (pos % CPR) ^ TnCfg<T>::swz(lrow[i]);
    const int zc = n0 + clog * EPC, xc = k0 + clog * EPC;
    zoff[i] = zc < p.N ? zc : 0;   // columns past the edge fetch valid memory; their products land in slab
    xoff[i] = xc < p.K ? xc : 0;   // columns that are never read
  }
  // column sums for the bias gradient: this thread owns logical chunk (tid % CPR) of rows tid / CPR + 256/CPR * i
  const int bc = tid % CPR;
  float bsum[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) bsum[e] = 0.f;

  f32x4 acc[4][4];  // [ki][ni]: D[i = k][j = n]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int64_t nsteps = (m_end - m_begin + BMK - 1) / BMK;
  for (int64_t s = 0; s < nsteps; ++s) {
    const int64_t mb = m_begin + s * BMK;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int64_t m = mb + lrow[i];
      m = m < m_end ? m : m_end - 1;
      __builtin_amdgcn_global_load_lds((glb_ptr)(dZ + m * p.ldz + zoff[i]), (lds_ptr)(&lds[0][i * 256 + wave_u * 64]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr)(X + m * p.ldx + xoff[i]), (lds_ptr)(&lds[1][i * 256 + wave_u * 64]), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (mb + BMK > m_end) {  // ragged last step: rows past the end must contribute nothing
#pragma unroll
      for (int i = 0; i < NCH; ++i)
        if (mb + lrow[i] >= m_end) {
          lds[0][i * 256 + tid] = make_uint4(0, 0, 0, 0);
          lds[1][i * 256 + tid] = make_uint4(0, 0, 0, 0);
        }
    }
    __syncthreads();
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int row = tid / CPR + (256 / CPR) * i;
        const uint4 v = lds[0][row * CPR + (bc ^ TnCfg<T>::swz(row))];
        if constexpr (__is_same(T, f16)) {
          const f16x8 h = __builtin_bit_cast(f16x8, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum[e] += (float)h[e];
        } else if constexpr (sizeof(T) == 2) {
          const bf16x8 h = __builtin_bit_cast(bf16x8, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum[e] += (float)h[e];
        } else {
          bsum[0] += __builtin_bit_cast(float, v.x);
          bsum[1] += __builtin_bit_cast(float, v.y);
          bsum[2] += __builtin_bit_cast(float, v.z);
          bsum[3] += __builtin_bit_cast(float, v.w);
        }
      }
    }
    const unsigned char *lz = reinterpret_cast<const unsigned char *>(&lds[0][0]);
    const unsigned char *lx = reinterpret_cast<const unsigned char *>(&lds[1][0]);
    if constexpr (sizeof(T) == 2) {
      // transposing reads: the 16 lanes of group g fetch a [4 rows][16 cols] block; lane (q, pp) = (r>>2, r&3)
      // addresses row q, cols 4pp..4pp+3 and receives column r of the 4 rows
      const int q = r >> 2, pp = r & 3;
      typedef __attribute__((address_space(3))) bf16x4 *lp;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 xf[4], zf[4];
        const int r0 = 32 * kk + 8 * g + q, r1 = r0 + 4;
        const int s0 = TnCfg<T>::swz(r0), s1 = TnCfg<T>::swz(r1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int xch = (wk * 64 + i * 16) / 8, zch = (wn * 64 + i * 16) / 8;   // even chunk index of the 16-col block
          bf16x4 lo, hi;
          lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(lx + (r0 * CPR + (xch ^ s0)) * 16 + pp * 8));
          hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(lx + (r1 * CPR + (xch ^ s1)) * 16 + pp * 8));
          xf[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(lz + (r0 * CPR + (zch ^ s0)) * 16 + pp * 8));
          hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp)(lz + (r1 * CPR + (zch ^ s1)) * 16 + pp * 8));
          zf[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int ki = 0; ki < 4; ++ki)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            // the transposing read moves 16-bit lanes; only the MFMA interprets them
            if constexpr (__is_same(T, f16))
              acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xf[ki]),
                                                                   __builtin_bit_cast(f16x8, zf[ni]), acc[ki][ni], 0, 0, 0);
            else
              acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[ki], zf[ni], acc[ki][ni], 0, 0, 0);
          }
      }
    } else {
#pragma unroll
      for (int ss = 0; ss < BMK / 4; ++ss) {
        float xf[4], zf[4];
        const int row = 4 * ss + g;
        const int sw = TnCfg<T>::swz(row);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int xcol = wk * 64 + i * 16 + r, zcol = wn * 64 + i * 16 + r;
          xf[i] = *reinterpret_cast<const float *>(lx + (row * CPR + ((xcol >> 2) ^ sw)) * 16 + (xcol & 3) * 4);
          zf[i] = *reinterpret_cast<const float *>(lz + (row * CPR + ((zcol >> 2) ^ sw)) * 16 + (zcol & 3) * 4);
        }
#pragma unroll
        for (int ki = 0; ki < 4; ++ki)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[ki], zf[ni], acc[ki][ni], 0, 0, 0);
      }
    }
    __syncthreads();  // all reads done before the next step's DMA overwrites the tiles
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
    float *red = reinterpret_cast<float *>(&lds[0][0]);  // [256 / CPR][128]
    const int tr = tid / CPR;
#pragma unroll
    for (int e = 0; e < EPC; ++e) red[tr * TN_T + bc * EPC + e] = bsum[e];
    __syncthreads();
    if (tid < TN_T) {
      float sum = 0.f;
      for (int j = 0; j < 256 / CPR; ++j) sum += red[j * TN_T + tid];
      p.bslab[(int64_t)sp * p.Np + n0 + tid] = sum;
    }
  }
}

// ---------------------------------------------------------------- whole 128 x 256 / 256 x 128 result, streamed (16-bit)
// The two widest layers' weight gradients (enc1 / dec6 at E8: N x K = 128 x 256 and 256 x 128 over M = 641 024 rows) are pure
// streaming reductions: 0.33 - 0.49 GB of operands, 42 GFLOP, a result of 32 K floats.  gemm_tn_kernel runs them as two
// 128 x 128 tiles per row split on single-buffered LDS (issue 32 KB, wait for all of it, compute, barrier): three
// workgroups per CU each spend most of a step waiting for their one stage -- 2.7 - 3.5 TB/s.  Here ONE workgroup of eight
// waves owns the whole result for its row split (every operand byte crosses L2 -> LDS once), and the 64-row steps go
// through a ring of three 48 KB LDS stages filled by LDS-DMA two steps ahead: 96 KB per CU in flight behind one counted
// vmcnt and ONE barrier per step (the stage refilled at step s is the one every wave finished reading before it arrived at
// step s's barrier).  Same fragment reads, swizzle, MFMA order inside a tile and slab layout as gemm_tn_kernel.
template <typename T, int NT, int KT>
__global__ __launch_bounds__(512) void gemm_tn_stream_kernel(const TnParams p) {
  static_assert(sizeof(T) == 2 && NT * KT == 128 * 256, "16-bit operands, a 32 K-element result");
  constexpr int EPC = 8, BMK = 64, NST = 3;
  constexpr int CZ = NT / EPC, CX = KT / EPC;          // 16-byte chunks per stage row: 16 / 32
  constexpr int ZCH = BMK * CZ, XCH = BMK * CX;        // chunks per stage and operand
  constexpr int NZ = ZCH / 512, NX = XCH / 512;        // LDS-DMA instructions per thread and step: 2 + 4 or 4 + 2
  constexpr int STG = ZCH + XCH;                       // 3 072 chunks = 48 KB
  constexpr int WN = NT / 64;
  __shared__ uint4 lds[NST * STG];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wn = wave % WN, wk = wave / WN;
  const int g = lane >> 4, r = lane & 15;
  const int sp = blockIdx.x;
  const int64_t m_begin = (int64_t)sp * p.rows_per_split;
  int64_t m_end = m_begin + p.rows_per_split;
  m_end = m_end < p.M ? m_end : p.M;
  const int rows = (int)(m_end - m_begin);
  const T *dZ = static_cast<const T *>(p.dZ) + m_begin * p.ldz;
  const T *X = static_cast<const T *>(p.X) + m_begin * p.ldx;
  const int ldz = (int)p.ldz, ldx = (int)p.ldx;        // a split's extent fits 32 bits (checked by the launcher)

  typedef __attribute__((address_space(3))) void *lds_ptr;
  typedef const __attribute__((address_space(1))) void *glb_ptr;
  int zrow[NZ], zoff[NZ], xrow[NX], xoff[NX];
#pragma unroll
  for (int i = 0; i < NZ; ++i) {
    const int pos = i * 512 + tid;
    zrow[i] = pos / CZ;
    const int c = ((pos % CZ) ^ TnCfg<T>::swz(zrow[i])) * EPC;
    zoff[i] = c < p.N ? c : 0;                         // columns past the edge: valid memory, products never read
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int pos = i * 512 + tid;
    xrow[i] = pos / CX;
    const int c = ((pos % CX) ^ TnCfg<T>::swz(xrow[i])) * EPC;
    xoff[i] = c < p.K ? c : 0;
  }
  auto issue = [&](int s, int buf) {
    uint4 *st = lds + buf * STG;
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      int m = s * BMK + zrow[i];
      m = m < rows ? m : rows - 1;
      __builtin_amdgcn_global_load_lds((glb_ptr)(dZ + (m * ldz + zoff[i])), (lds_ptr)(&st[i * 512 + wave * 64]), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      int m = s * BMK + xrow[i];
      m = m < rows ? m : rows - 1;
      __builtin_amdgcn_global_load_lds((glb_ptr)(X + (m * ldx + xoff[i])), (lds_ptr)(&st[ZCH + i * 512 + wave * 64]), 16, 0, 0);
    }
  };

  constexpr int BROWS = 512 / CZ;                      // rows of a stage one pass of the workgroup's bias sums covers
  const int bc = tid % CZ;
  float bsum[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) bsum[e] = 0.f;
  const bool do_bias = p.bslab != nullptr;

  f32x4 acc[4][4];                                     // [ki][ni]: D[i = k][j = n]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Fragment reads (ds_read_b64_tr_b16, as in gemm_tn_kernel): the 16 lanes of group g fetch a [4 rows][16 cols] block, lane
  // (q, pp) = (r >> 2, r & 3) addresses row 8 g + q (+ 4: second half, + 32: second MFMA), columns 4 pp .. 4 pp + 3.  Rows
  // r0 and r0 + 4 share a swizzle, so one address per 16-column block and immediates for the rest.  Every LDS read of the
  // loop is inline asm: a compiler-visible read behind an LDS-DMA gets an s_waitcnt vmcnt(0) in front of it (and
  // __syncthreads() one in front of the barrier), which would drain the two steps in flight at every step.
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)lds;
  const int q = r >> 2, pp = r & 3;
  const int r0 = 8 * g + q, s0 = TnCfg<T>::swz(r0);
  unsigned ax[4], az[4], ab[BMK / BROWS];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ax[i] = (unsigned)((ZCH + r0 * CX + ((wk * 8 + 2 * i) ^ s0)) * 16 + pp * 8);
    az[i] = (unsigned)((r0 * CZ + ((wn * 8 + 2 * i) ^ s0)) * 16 + pp * 8);
  }
#pragma unroll
  for (int i = 0; i < BMK / BROWS; ++i) {
    const int row = tid / CZ + BROWS * i;
    ab[i] = (unsigned)((row * CZ + (bc ^ TnCfg<T>::swz(row))) * 16);
  }
#define CUM_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")

  const int nsteps = (rows + BMK - 1) / BMK;
  issue(0, 0);
  if (nsteps > 1) issue(1, 1);
  int buf = 0;
  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NZ + NX) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((s + 1) * BMK > rows) {                        // ragged last step: rows past the end contribute nothing
      uint4 *st = lds + buf * STG;
#pragma unroll
      for (int i = 0; i < NZ; ++i)
        if (s * BMK + zrow[i] >= rows) st[i * 512 + tid] = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NX; ++i)
        if (s * BMK + xrow[i] >= rows) st[ZCH + i * 512 + tid] = make_uint4(0, 0, 0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");            // step s is visible; every wave is done with step s - 1
    if (s + 2 < nsteps) issue(s + 2, buf == 0 ? 2 : buf - 1);
    const unsigned sb = lds0 + (unsigned)(buf * STG * 16);
    u32x2 xl[2][4], xh[2][4], zl[2][4], zh[2][4];
    u32x4 bv[BMK / BROWS];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      CUM_TR(xl[0][i], sb + ax[i], 0);
      CUM_TR(xh[0][i], sb + ax[i], 4 * CX * 16);
      CUM_TR(zl[0][i], sb + az[i], 0);
      CUM_TR(zh[0][i], sb + az[i], 4 * CZ * 16);
    }
#pragma unroll
    for (int i = 0; i < BMK / BROWS; ++i)            // (unconditional: a branch here would put copies in front of the waits)
      asm volatile("ds_read_b128 %0, %1" : "=v"(bv[i]) : "v"(sb + ab[i]) : "memory");
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      // the wait takes the read results themselves as operands: nothing the compiler emits may touch them before it
#define CUM_TR_WAIT(n)                                                                                              \
  asm volatile("s_waitcnt lgkmcnt(%16)"                                                                            \
               : "+v"(xl[kk][0]), "+v"(xl[kk][1]), "+v"(xl[kk][2]), "+v"(xl[kk][3]), "+v"(xh[kk][0]), "+v"(xh[kk][1]), \
                 "+v"(xh[kk][2]), "+v"(xh[kk][3]), "+v"(zl[kk][0]), "+v"(zl[kk][1]), "+v"(zl[kk][2]), "+v"(zl[kk][3]), \
                 "+v"(zh[kk][0]), "+v"(zh[kk][1]), "+v"(zh[kk][2]), "+v"(zh[kk][3])                                    \
               : "n"(n) : "memory")
      __builtin_amdgcn_sched_barrier(0);               // (the first batch of MFMAs stays in front of the second wait)
      if (kk == 0) CUM_TR_WAIT(BMK / BROWS);           // (lgkmcnt counts to 15: the second batch is issued behind this wait)
      else CUM_TR_WAIT(0);
#undef CUM_TR_WAIT
      if (kk == 0) {                                   // the second batch's reads fly under the first batch's MFMAs
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          CUM_TR(xl[1][i], sb + ax[i], 32 * CX * 16);
          CUM_TR(xh[1][i], sb + ax[i], 36 * CX * 16);
          CUM_TR(zl[1][i], sb + az[i], 32 * CZ * 16);
          CUM_TR(zh[1][i], sb + az[i], 36 * CZ * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      u32x4 xf[4], zf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xf[i] = __builtin_shufflevector(xl[kk][i], xh[kk][i], 0, 1, 2, 3);
        zf[i] = __builtin_shufflevector(zl[kk][i], zh[kk][i], 0, 1, 2, 3);
      }
#pragma unroll
      for (int ki = 0; ki < 4; ++ki)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          if constexpr (__is_same(T, f16))
            acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, xf[ki]),
                                                                 __builtin_bit_cast(f16x8, zf[ni]), acc[ki][ni], 0, 0, 0);
          else
            acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xf[ki]),
                                                                  __builtin_bit_cast(bf16x8, zf[ni]), acc[ki][ni], 0, 0, 0);
        }
      if (kk == 0) {                                   // the dZ rows' column sums ride in the first MFMA batch's shadow
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (BMK / BROWS == 2)
          asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(bv[0]), "+v"(bv[1]) : : "memory");
        else
          asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[BMK / BROWS - 1]) : : "memory");
#pragma unroll
        for (int i = 0; i < BMK / BROWS; ++i) {
          if constexpr (__is_same(T, f16)) {
            const f16x8 h = __builtin_bit_cast(f16x8, bv[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum[e] += (float)h[e];
          } else {
            const bf16x8 h = __builtin_bit_cast(bf16x8, bv[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum[e] += (float)h[e];
          }
        }
      }
    }
    buf = buf == NST - 1 ? 0 : buf + 1;
  }
#undef CUM_TR

  // ---- slab store: lane holds D[k = kb + 4g + j][n = nb + r]
  float *slab = p.slab + (int64_t)sp * p.Np * p.Kp;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = wn * 64 + ni * 16 + r;
#pragma unroll
    for (int ki = 0; ki < 4; ++ki) {
      const int k = wk * 64 + ki * 16 + 4 * g;
      *reinterpret_cast<float4 *>(slab + (int64_t)n * p.Kp + k) =
          make_float4(acc[ki][ni][0], acc[ki][ni][1], acc[ki][ni][2], acc[ki][ni][3]);
    }
  }
  if (do_bias) {                                       // threads with the same column chunk hold partial sums
    __syncthreads();                                   // the last step's reads are done: the stages are free
    float *red = reinterpret_cast<float *>(lds);       // [512 / CZ][NT]
    const int tr = tid / CZ;
#pragma unroll
    for (int e = 0; e < EPC; ++e) red[tr * NT + bc * EPC + e] = bsum[e];
    __syncthreads();
    if (tid < NT) {
      float sum = 0.f;
      for (int j = 0; j < 512 / CZ; ++j) sum += red[j * NT + tid];
      p.bslab[(int64_t)sp * p.Np + tid] = sum;
    }
  }
}

// ---------------------------------------------------------------- 256 x 256 output tile, 8 waves (16-bit types)
// Same pipeline as gemm_nt8_kernel (gemm.hip): a reduction step of 64 rows is four 16 KB UNITS -- X columns 0-127 /
// 128-255 of the tile (X0, X1) and dZ columns 0-127 / 128-255 (Z0, Z1), each [64 rows][16 chunks] in the swizzled
// layout of gemm_tn_kernel -- two steps of units = 128 KB of LDS, one workgroup per CU.  Wave (wk, wn) owns k columns
// [128 wk, +128) x n columns [64 wn, +64): it reads unit X_wk whole at the start of the step (32 transposing reads, kept
// in registers) and its half of Z_(wn >> 1) in two parts, so the X units are free after barrier B2 and the Z units after
// B3, and step s + 2's units are DMA'd into them under step s's 64 MFMAs per wave; the top of step s + 1 waits
// `vmcnt(8)`: step s + 1 has landed, step s + 2's eight DMAs stay in flight.  Fragment reads are inline asm for the
// reason given in gemm.hip (the compiler would drain vmcnt before every LDS read of a DMA target).
// The ragged step of a split (rows not a multiple of 64) is its FIRST one: its out-of-range rows are fetched from a
// clamped row and zeroed in LDS in the prologue, where registers are plentiful; every later step is full.
// Bias gradient: db[n] = sum_m dZ[m][n] is one more MFMA per dZ fragment with a 0/1 selector as the other operand
// (selector rows 4 ni .. 4 ni + 3 are ones: the four fragments of a wave accumulate into the four row groups of one
// 16 x 16 result), by the waves with wk == 0 of the first k tile.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#ifdef CUM_AB   // gemm_tn8_kernel: the predecessor of gemm_tn9_kernel (same pipeline, all waves in one phase), CUM_TN9=0
template <typename T>
__global__ __launch_bounds__(512) void gemm_tn8_kernel(const TnParams p) {
  static_assert(sizeof(T) == 2, "16-bit element types only");
  constexpr int UNIT = 64 * 16;                      // 16-byte chunks of one unit
  __shared__ uint4 lds_all[2 * 4 * UNIT];            // [step parity][X0, X1, Z0, Z1]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wk = wave >> 2, wn = wave & 3;
  const int g = lane >> 4, r = lane & 15;
  // One resident round: work item w = xcd * (slots per XCD) + slot, split = w / tiles, tile = w % tiles (n fastest):
  // the workgroups of one XCD hold consecutive tiles of one or two splits, so their rows meet in that XCD's L2.
  const int ntn = p.Np / 256, ntk = p.Kp / 256, tiles = ntn * ntk;
  const int w = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  if (w >= tiles * p.nsplit) return;
  const int sp = w / tiles, tile = w % tiles;
  const int n0 = (tile % ntn) * 256, k0 = (tile / ntn) * 256;
  const int64_t m_begin = (int64_t)sp * p.rows_per_split;
  int64_t m_end = m_begin + p.rows_per_split;
  m_end = m_end < p.M ? m_end : p.M;
  const int len = (int)(m_end - m_begin);
  const int nk = (len + 63) / 64;
  const int rem = len - 64 * (nk - 1);               // rows of step 0 (1 .. 64)
  const T *dZ = static_cast<const T *>(p.dZ);
  const T *X = static_cast<const T *>(p.X);
  const bool bias_wave = p.bslab != nullptr && k0 == 0 && wk == 0;

  typedef __attribute__((address_space(3))) void *lds_ptr;
  typedef const __attribute__((address_space(1))) void *glb_ptr;
  // DMA sources: unit u, instruction it fills linear chunk it * 512 + tid of the unit = (row, physical chunk).  The
  // address is a wave-uniform base (SGPRs: tile corner + step) plus a small per-thread byte offset (row * ld + chunk),
  // so a step's eight DMAs need four offset registers and no vector address arithmetic.
  const int64_t stepx = 128 * p.ldx, stepz = 128 * p.ldz;                 // bytes per 64-row step
  const char *xb = reinterpret_cast<const char *>(X + k0) + m_begin * p.ldx * 2;
  const char *zb = reinterpret_cast<const char *>(dZ + n0) + m_begin * p.ldz * 2;
  unsigned xo[2], zo[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int pos = it * 512 + tid;
    const int row = pos >> 4;
    const int clog = (pos & 15) ^ TnCfg<T>::swz(row);
    const int rowc = row < rem ? row : rem - 1;                            // step 0: clamped rows (zeroed below)
    xo[it] = (unsigned)(row * (int)p.ldx * 2 + clog * 16);
    zo[it] = (unsigned)(row * (int)p.ldz * 2 + clog * 16);
    const unsigned x0 = (unsigned)(rowc * (int)p.ldx * 2 + clog * 16), z0 = (unsigned)(rowc * (int)p.ldz * 2 + clog * 16);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      __builtin_amdgcn_global_load_lds((glb_ptr)((u < 2 ? xb + x0 : zb + z0) + 256 * (u & 1)),
                                       (lds_ptr)(&lds_all[u * UNIT + it * 512 + wave * 64]), 16, 0, 0);
  }
  // step s >= 1 covers rows m_begin + rem + 64 (s - 1) ..: base of step s = b1 + s * step
  const char *xb1 = xb + (int64_t)(rem - 64) * p.ldx * 2, *zb1 = zb + (int64_t)(rem - 64) * p.ldz * 2;
#define CUM_STAGE(u, s, par)                                                                                    \
  do {                                                                                                          \
    const char *ub = ((u) < 2 ? xb1 + (s) * stepx : zb1 + (s) * stepz) + 256 * ((u) & 1);                        \
    _Pragma("unroll") for (int it = 0; it < 2; ++it)                                                           \
      __builtin_amdgcn_global_load_lds((glb_ptr)(ub + ((u) < 2 ? xo[it] : zo[it])),                            \
                                       (lds_ptr)(&lds_all[((par) * 4 + (u)) * UNIT + it * 512 + wave * 64]), 16, 0, 0); \
  } while (0)
  if (nk > 1) {
#pragma unroll
    for (int u = 0; u < 4; ++u) CUM_STAGE(u, 1, 1);
  }
  if (rem < 64) {                                    // rows past the end of the split must contribute nothing
    if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int pos = it * 512 + tid;
      if ((pos >> 4) >= rem) {                       // this thread's own DMA filled the chunk: no barrier needed
#pragma unroll
        for (int u = 0; u < 4; ++u) lds_all[u * UNIT + pos] = make_uint4(0, 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  f32x4 acc[2][4][4];                                // [k half][ni][ki]: k = 128 wk + 64 h + 16 ki + 4 g + j, n = 64 wn + 16 ni + r
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bacc = f32x4{0.f, 0.f, 0.f, 0.f};

  // Transposing reads (see gemm_tn_kernel): lane (g, q = r >> 2, pp = r & 3) addresses row 32 ks + 8 g + q (+ 4 for the
  // upper half of the fragment), columns 16 blk + 4 pp ..; the row swizzle term t = q | (g & 1) << 2 is the same for both
  // halves and both ks, so a block's four reads share one address register and differ by immediate offsets.
  const int q = r >> 2, pp = r & 3, t = q | ((g & 1) << 2);
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)lds_all;
  const unsigned lane_base = lds0 + (unsigned)((8 * g + q) * 256 + 8 * pp);
  unsigned ax[8], az[4];
#pragma unroll
  for (int b = 0; b < 8; ++b) ax[b] = lane_base + (unsigned)(wk * UNIT * 16 + 32 * (b ^ t));
#pragma unroll
  for (int b = 0; b < 4; ++b) az[b] = lane_base + (unsigned)((2 + (wn >> 1)) * UNIT * 16 + 32 * ((4 * (wn & 1) + b) ^ t));
  // bias selector for fragment ni: lanes whose MFMA row r lies in [4 ni, 4 ni + 4) hold ones
  const unsigned one2 = __is_same(T, f16) ? 0x3C003C00u : 0x3F803F80u;

#define CUM_TR(dst, reg, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #off : "={" reg "}"(dst) : "v"(addr) : "memory")
#define CUM_MFMA(a, b, c)                                                                                      \
  do {                                                                                                         \
    if constexpr (__is_same(T, f16))                                                                           \
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0); \
    else                                                                                                       \
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0); \
  } while (0)
#define CUM_HALFQ(h, nlo, ks)                                                                                  \
  do {                                                                                                         \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                           \
      _Pragma("unroll") for (int ki = 0; ki < 4; ++ki)                                                         \
        CUM_MFMA(xf[ks][4 * (h) + ki], zf[ks][ni], acc[h][(nlo) + ni][ki]);                                    \
  } while (0)
#define CUM_BIAS(nlo)                                                                                          \
  do {                                                                                                         \
    if (bias_wave) {                                                                                           \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {                                                       \
        const unsigned sv = q == (nlo) + ni ? one2 : 0u;                                                       \
        const u32x4 sel = u32x4{sv, sv, sv, sv};                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) CUM_MFMA(sel, zf[ks][ni], bacc);                      \
      }                                                                                                        \
    }                                                                                                          \
  } while (0)

  // The two halves of a fragment come from two 64-bit transposing reads but must sit in one aligned 128-bit register
  // tuple for the MFMA, and an asm operand cannot name half of a tuple: the fragments are pinned to physical registers
  // (x (ks, blk): v[176 + 32 ks + 4 blk ..+3], dZ (ks, ni): v[240 + 8 ks + 4 ni ..+3]).  A read defines its half, the wait
  // statement consumes both halves and defines the whole fragment in the same registers: no copies (checked in the ISA:
  // a v_mov of a half issued before the wait would read a register the LDS has not written yet).
  for (int s = 0; s < nk; ++s) {
    const int par = s & 1;
    const bool more = s + 2 < nk;
    if (s + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // step s landed; step s + 1 stays in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");                              // B1
    u32x2 xl[2][8], xh[2][8], zl[2][2], zh[2][2];
    u32x4 xf[2][8], zf[2][2];
    // ---- phase 1: reads in the order the MFMAs consume them
    CUM_TR(zl[0][0], "v[240:241]", az[0], 0); CUM_TR(zh[0][0], "v[242:243]", az[0], 1024);  CUM_TR(zl[0][1], "v[244:245]", az[1], 0); CUM_TR(zh[0][1], "v[246:247]", az[1], 1024);
    CUM_TR(xl[0][0], "v[176:177]", ax[0], 0); CUM_TR(xh[0][0], "v[178:179]", ax[0], 1024);  CUM_TR(xl[0][1], "v[180:181]", ax[1], 0); CUM_TR(xh[0][1], "v[182:183]", ax[1], 1024);
    CUM_TR(xl[0][2], "v[184:185]", ax[2], 0); CUM_TR(xh[0][2], "v[186:187]", ax[2], 1024);  CUM_TR(xl[0][3], "v[188:189]", ax[3], 0); CUM_TR(xh[0][3], "v[190:191]", ax[3], 1024);
    CUM_TR(zl[1][0], "v[248:249]", az[0], 8192); CUM_TR(zh[1][0], "v[250:251]", az[0], 9216);  CUM_TR(zl[1][1], "v[252:253]", az[1], 8192); CUM_TR(zh[1][1], "v[254:255]", az[1], 9216);
    CUM_TR(xl[1][0], "v[208:209]", ax[0], 8192); CUM_TR(xh[1][0], "v[210:211]", ax[0], 9216);  CUM_TR(xl[1][1], "v[212:213]", ax[1], 8192); CUM_TR(xh[1][1], "v[214:215]", ax[1], 9216);
    CUM_TR(xl[1][2], "v[216:217]", ax[2], 8192); CUM_TR(xh[1][2], "v[218:219]", ax[2], 9216);  CUM_TR(xl[1][3], "v[220:221]", ax[3], 8192); CUM_TR(xh[1][3], "v[222:223]", ax[3], 9216);
    asm volatile("s_waitcnt lgkmcnt(12)"
                 : "={v[240:243]}"(zf[0][0]), "={v[244:247]}"(zf[0][1]), "={v[176:179]}"(xf[0][0]), "={v[180:183]}"(xf[0][1]), "={v[184:187]}"(xf[0][2]), "={v[188:191]}"(xf[0][3])
                 : "{v[240:241]}"(zl[0][0]), "{v[242:243]}"(zh[0][0]),
                   "{v[244:245]}"(zl[0][1]), "{v[246:247]}"(zh[0][1]),
                   "{v[176:177]}"(xl[0][0]), "{v[178:179]}"(xh[0][0]),
                   "{v[180:181]}"(xl[0][1]), "{v[182:183]}"(xh[0][1]),
                   "{v[184:185]}"(xl[0][2]), "{v[186:187]}"(xh[0][2]),
                   "{v[188:189]}"(xl[0][3]), "{v[190:191]}"(xh[0][3]) : "memory");
    __builtin_amdgcn_s_setprio(1);
    CUM_HALFQ(0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    CUM_TR(xl[0][4], "v[192:193]", ax[4], 0); CUM_TR(xh[0][4], "v[194:195]", ax[4], 1024);  CUM_TR(xl[0][5], "v[196:197]", ax[5], 0); CUM_TR(xh[0][5], "v[198:199]", ax[5], 1024);
    CUM_TR(xl[0][6], "v[200:201]", ax[6], 0); CUM_TR(xh[0][6], "v[202:203]", ax[6], 1024);  CUM_TR(xl[0][7], "v[204:205]", ax[7], 0); CUM_TR(xh[0][7], "v[206:207]", ax[7], 1024);
    asm volatile("s_waitcnt lgkmcnt(8)"
                 : "={v[248:251]}"(zf[1][0]), "={v[252:255]}"(zf[1][1]), "={v[208:211]}"(xf[1][0]), "={v[212:215]}"(xf[1][1]), "={v[216:219]}"(xf[1][2]), "={v[220:223]}"(xf[1][3])
                 : "{v[248:249]}"(zl[1][0]), "{v[250:251]}"(zh[1][0]),
                   "{v[252:253]}"(zl[1][1]), "{v[254:255]}"(zh[1][1]),
                   "{v[208:209]}"(xl[1][0]), "{v[210:211]}"(xh[1][0]),
                   "{v[212:213]}"(xl[1][1]), "{v[214:215]}"(xh[1][1]),
                   "{v[216:217]}"(xl[1][2]), "{v[218:219]}"(xh[1][2]),
                   "{v[220:221]}"(xl[1][3]), "{v[222:223]}"(xh[1][3]) : "memory");
    CUM_HALFQ(0, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    CUM_TR(xl[1][4], "v[224:225]", ax[4], 8192); CUM_TR(xh[1][4], "v[226:227]", ax[4], 9216);  CUM_TR(xl[1][5], "v[228:229]", ax[5], 8192); CUM_TR(xh[1][5], "v[230:231]", ax[5], 9216);
    CUM_TR(xl[1][6], "v[232:233]", ax[6], 8192); CUM_TR(xh[1][6], "v[234:235]", ax[6], 9216);  CUM_TR(xl[1][7], "v[236:237]", ax[7], 8192); CUM_TR(xh[1][7], "v[238:239]", ax[7], 9216);
    CUM_BIAS(0);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "={v[192:195]}"(xf[0][4]), "={v[196:199]}"(xf[0][5]), "={v[200:203]}"(xf[0][6]), "={v[204:207]}"(xf[0][7]), "={v[224:227]}"(xf[1][4]), "={v[228:231]}"(xf[1][5]), "={v[232:235]}"(xf[1][6]), "={v[236:239]}"(xf[1][7])
                 : "{v[192:193]}"(xl[0][4]), "{v[194:195]}"(xh[0][4]),
                   "{v[196:197]}"(xl[0][5]), "{v[198:199]}"(xh[0][5]),
                   "{v[200:201]}"(xl[0][6]), "{v[202:203]}"(xh[0][6]),
                   "{v[204:205]}"(xl[0][7]), "{v[206:207]}"(xh[0][7]),
                   "{v[224:225]}"(xl[1][4]), "{v[226:227]}"(xh[1][4]),
                   "{v[228:229]}"(xl[1][5]), "{v[230:231]}"(xh[1][5]),
                   "{v[232:233]}"(xl[1][6]), "{v[234:235]}"(xh[1][6]),
                   "{v[236:237]}"(xl[1][7]), "{v[238:239]}"(xh[1][7]) : "memory");
    asm volatile("s_barrier" ::: "memory");                              // B2: the X units of this parity are free
    if (more) {
      CUM_STAGE(0, s + 2, par);
      CUM_STAGE(1, s + 2, par);
    }
    // ---- phase 2
    __builtin_amdgcn_s_setprio(1);
    CUM_HALFQ(1, 0, 0);
    CUM_HALFQ(1, 0, 1);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 3: dZ fragments of n columns 32-63 (same registers as columns 0-31)
    CUM_TR(zl[0][0], "v[240:241]", az[2], 0); CUM_TR(zh[0][0], "v[242:243]", az[2], 1024);  CUM_TR(zl[0][1], "v[244:245]", az[3], 0); CUM_TR(zh[0][1], "v[246:247]", az[3], 1024);
    CUM_TR(zl[1][0], "v[248:249]", az[2], 8192); CUM_TR(zh[1][0], "v[250:251]", az[2], 9216);  CUM_TR(zl[1][1], "v[252:253]", az[3], 8192); CUM_TR(zh[1][1], "v[254:255]", az[3], 9216);
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "={v[240:243]}"(zf[0][0]), "={v[244:247]}"(zf[0][1]), "={v[248:251]}"(zf[1][0]), "={v[252:255]}"(zf[1][1])
                 : "{v[240:241]}"(zl[0][0]), "{v[242:243]}"(zh[0][0]),
                   "{v[244:245]}"(zl[0][1]), "{v[246:247]}"(zh[0][1]),
                   "{v[248:249]}"(zl[1][0]), "{v[250:251]}"(zh[1][0]),
                   "{v[252:253]}"(zl[1][1]), "{v[254:255]}"(zh[1][1]) : "memory");
    asm volatile("s_barrier" ::: "memory");                              // B3: the Z units of this parity are free
    if (more) {
      CUM_STAGE(2, s + 2, par);
      CUM_STAGE(3, s + 2, par);
    }
    __builtin_amdgcn_s_setprio(1);
    CUM_HALFQ(1, 2, 0);
    CUM_HALFQ(1, 2, 1);
    // ---- phase 4
    CUM_HALFQ(0, 2, 0);
    CUM_HALFQ(0, 2, 1);
    CUM_BIAS(2);
    __builtin_amdgcn_s_setprio(0);
    const unsigned flip = par ? 0u - 65536u : 65536u;                   // the other parity's units
#pragma unroll
    for (int b = 0; b < 8; ++b) ax[b] += flip;
#pragma unroll
    for (int b = 0; b < 4; ++b) az[b] += flip;
  }
#undef CUM_TR
#undef CUM_MFMA
#undef CUM_HALFQ
#undef CUM_BIAS
#undef CUM_STAGE

  // ---- slab store: lane holds D[k = kb + 4g + j][n = nb + r]
  float *slab = p.slab + (int64_t)sp * p.Np * p.Kp;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = n0 + wn * 64 + ni * 16 + r;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ki = 0; ki < 4; ++ki) {
        const int k = k0 + wk * 128 + h * 64 + ki * 16 + 4 * g;
        *reinterpret_cast<float4 *>(slab + (int64_t)n * p.Kp + k) =
            make_float4(acc[h][ni][ki][0], acc[h][ni][ki][1], acc[h][ni][ki][2], acc[h][ni][ki][3]);
      }
  }
  // selector rows 4 ni .. 4 ni + 3 (held by lane group g = ni) carry the column sums of fragment ni
  if (bias_wave) p.bslab[(int64_t)sp * p.Np + n0 + wn * 64 + 16 * g + r] = bacc[0];
}

#endif  // CUM_AB

template <typename T>
__global__ __launch_bounds__(512) void gemm_tn9_kernel(const TnParams p) {
  static_assert(sizeof(T) == 2, "16-bit element types only");
  constexpr int UNIT = 64 * 16;                      // 16-byte chunks of one unit
  __shared__ uint4 lds_all[2 * 4 * UNIT];            // [step parity][X0, X1, Z0, Z1]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wk = wave >> 2, wn = wave & 3;
  const int g = lane >> 4, r = lane & 15;
  // One resident round: work item w = xcd * (slots per XCD) + slot, split = w / tiles, tile = w % tiles (n fastest):
  // the workgroups of one XCD hold consecutive tiles of one or two splits, so their rows meet in that XCD's L2.
  const int ntn = p.Np / 256, ntk = p.Kp / 256, tiles = ntn * ntk;
  const int w = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  if (w >= tiles * p.nsplit) return;
  const int sp = w / tiles, tile = w % tiles;
  const int n0 = (tile % ntn) * 256, k0 = (tile / ntn) * 256;
  const int64_t m_begin = (int64_t)sp * p.rows_per_split;
  int64_t m_end = m_begin + p.rows_per_split;
  m_end = m_end < p.M ? m_end : p.M;
  const int len = (int)(m_end - m_begin);
  const int nk = (len + 63) / 64;
  const int rem = len - 64 * (nk - 1);               // rows of step 0 (1 .. 64)
  const T *dZ = static_cast<const T *>(p.dZ);
  const T *X = static_cast<const T *>(p.X);
  const bool bias_wave = p.bslab != nullptr && k0 == 0 && wk == 0;

  typedef __attribute__((address_space(3))) void *lds_ptr;
  typedef const __attribute__((address_space(1))) void *glb_ptr;
  // DMA sources: unit u, instruction it fills linear chunk it * 512 + tid of the unit = (row, physical chunk).  The
  // address is a wave-uniform base (SGPRs: tile corner + step) plus a small per-thread byte offset (row * ld + chunk),
  // so a step's eight DMAs need four offset registers and no vector address arithmetic.
  const int64_t stepx = 128 * p.ldx, stepz = 128 * p.ldz;                 // bytes per 64-row step
  const char *xb = reinterpret_cast<const char *>(X + k0) + m_begin * p.ldx * 2;
  const char *zb = reinterpret_cast<const char *>(dZ + n0) + m_begin * p.ldz * 2;
  unsigned xo[2], zo[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int pos = it * 512 + tid;
    const int row = pos >> 4;
    const int clog = (pos & 15) ^ TnCfg<T>::swz(row);
    const int rowc = row < rem ? row : rem - 1;                            // step 0: clamped rows (zeroed below)
    xo[it] = (unsigned)(row * (int)p.ldx * 2 + clog * 16);
    zo[it] = (unsigned)(row * (int)p.ldz * 2 + clog * 16);
    const unsigned x0 = (unsigned)(rowc * (int)p.ldx * 2 + clog * 16), z0 = (unsigned)(rowc * (int)p.ldz * 2 + clog * 16);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      __builtin_amdgcn_global_load_lds((glb_ptr)((u < 2 ? xb + x0 : zb + z0) + 256 * (u & 1)),
                                       (lds_ptr)(&lds_all[u * UNIT + it * 512 + wave * 64]), 16, 0, 0);
  }
  // step s >= 1 covers rows m_begin + rem + 64 (s - 1) ..: base of step s = b1 + s * step
  const char *xb1 = xb + (int64_t)(rem - 64) * p.ldx * 2, *zb1 = zb + (int64_t)(rem - 64) * p.ldz * 2;
#define CUM_STAGE(u, s, par)                                                                                    \
  do {                                                                                                          \
    const char *ub = ((u) < 2 ? xb1 + (s) * stepx : zb1 + (s) * stepz) + 256 * ((u) & 1);                        \
    _Pragma("unroll") for (int it = 0; it < 2; ++it)                                                           \
      __builtin_amdgcn_global_load_lds((glb_ptr)(ub + ((u) < 2 ? xo[it] : zo[it])),                            \
                                       (lds_ptr)(&lds_all[((par) * 4 + (u)) * UNIT + it * 512 + wave * 64]), 16, 0, 0); \
  } while (0)
  if (nk > 1) {
#pragma unroll
    for (int u = 0; u < 4; ++u) CUM_STAGE(u, 1, 1);
  }
  if (rem < 64) {                                    // rows past the end of the split must contribute nothing
    if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int pos = it * 512 + tid;
      if ((pos >> 4) >= rem) {                       // this thread's own DMA filled the chunk: no barrier needed
#pragma unroll
        for (int u = 0; u < 4; ++u) lds_all[u * UNIT + pos] = make_uint4(0, 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  f32x4 acc[2][4][4];                                // [k half][ni][ki]: k = 128 wk + 64 h + 16 ki + 4 g + j, n = 64 wn + 16 ni + r
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bacc = f32x4{0.f, 0.f, 0.f, 0.f};

  // Transposing reads (see gemm_tn_kernel): lane (g, q = r >> 2, pp = r & 3) addresses row 32 ks + 8 g + q (+ 4 for the
  // upper half of the fragment), columns 16 blk + 4 pp ..; the row swizzle term t = q | (g & 1) << 2 is the same for both
  // halves and both ks, so a block's four reads share one address register and differ by immediate offsets.
  const int q = r >> 2, pp = r & 3, t = q | ((g & 1) << 2);
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)lds_all;
  const unsigned lane_base = lds0 + (unsigned)((8 * g + q) * 256 + 8 * pp);
  unsigned ax[8], az[4];
#pragma unroll
  for (int b = 0; b < 8; ++b) ax[b] = lane_base + (unsigned)(wk * UNIT * 16 + 32 * (b ^ t));
#pragma unroll
  for (int b = 0; b < 4; ++b) az[b] = lane_base + (unsigned)((2 + (wn >> 1)) * UNIT * 16 + 32 * ((4 * (wn & 1) + b) ^ t));
  // bias selector for fragment ni: lanes whose MFMA row r lies in [4 ni, 4 ni + 4) hold ones
  const unsigned one2 = __is_same(T, f16) ? 0x3C003C00u : 0x3F803F80u;

#define CUM_TR(dst, reg, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #off : "={" reg "}"(dst) : "v"(addr) : "memory")
#define CUM_MFMA(a, b, c)                                                                                      \
  do {                                                                                                         \
    if constexpr (__is_same(T, f16))                                                                           \
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0); \
    else                                                                                                       \
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0); \
  } while (0)
#define CUM_HALFQ(h, nlo, ks)                                                                                  \
  do {                                                                                                         \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                           \
      _Pragma("unroll") for (int ki = 0; ki < 4; ++ki)                                                         \
        CUM_MFMA(xf[ks][4 * (h) + ki], zf[ks][ni], acc[h][(nlo) + ni][ki]);                                    \
  } while (0)
#define CUM_BIAS(nlo)                                                                                          \
  do {                                                                                                         \
    if (bias_wave) {                                                                                           \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) {                                                       \
        const unsigned sv = q == (nlo) + ni ? one2 : 0u;                                                       \
        const u32x4 sel = u32x4{sv, sv, sv, sv};                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) CUM_MFMA(sel, zf[ks][ni], bacc);                      \
      }                                                                                                        \
    }                                                                                                          \
  } while (0)

  // Fragments are pinned to physical registers as in gemm_tn8_kernel (x (ks, blk): v[176 + 32 ks + 4 blk ..+3], dZ (ks, ni):
  // v[240 + 8 ks + 4 ni ..+3]).  Schedule: gemm_nt9_kernel's ping-pong (gemm.hip) -- eight slots per reduction step, load
  // and compute alternating, waves 4-7 (wk = 1, the X1 unit) one slot behind waves 0-3 (wk = 0, X0):
  //   L1: X columns 0-63 of the wave's half + dZ columns 0-31 (24 reads)   C1: (k 0-63,  n 0-31) + bias MFMAs
  //   L2: X columns 64-127 (16 reads)                                      C2: (k 64-127, n 0-31)
  //   L3: dZ columns 32-63 (8 reads)                                       C3: (k 64-127, n 32-63)
  //   L4: -                                                                C4: (k 0-63,  n 32-63) + bias MFMAs
  // LDS-DMA of step s + 2: group 0: X0 in L3, X1 in L4, Z0 + Z1 in the next step's L1; group 1: X0 + X1 in L3, Z0 + Z1 in L4.
#define CUM_BAR()                               \
  do {                                          \
    __builtin_amdgcn_sched_barrier(0);          \
    asm volatile("s_barrier" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);          \
  } while (0)
#define CUM_QUADT(h, nlo)                \
  do {                                   \
    __builtin_amdgcn_s_setprio(1);       \
    CUM_HALFQ(h, nlo, 0);                \
    CUM_HALFQ(h, nlo, 1);                \
    __builtin_amdgcn_s_setprio(0);       \
  } while (0)
  if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // step 0 landed (and zeroed where ragged); step 1 in flight
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CUM_BAR();
  if (wk != 0) CUM_BAR();                                                // group 1 runs one slot behind
  for (int s = 0; s < nk; ++s) {
    const int par = s & 1;
    const bool more1 = s + 1 < nk, more2 = s + 2 < nk;
    u32x2 xl[2][8], xh[2][8], zl[2][2], zh[2][2];
    u32x4 xf[2][8], zf[2][2];
    // ---- L1
    CUM_TR(zl[0][0], "v[240:241]", az[0], 0); CUM_TR(zh[0][0], "v[242:243]", az[0], 1024);  CUM_TR(zl[0][1], "v[244:245]", az[1], 0); CUM_TR(zh[0][1], "v[246:247]", az[1], 1024);
    CUM_TR(xl[0][0], "v[176:177]", ax[0], 0); CUM_TR(xh[0][0], "v[178:179]", ax[0], 1024);  CUM_TR(xl[0][1], "v[180:181]", ax[1], 0); CUM_TR(xh[0][1], "v[182:183]", ax[1], 1024);
    CUM_TR(xl[0][2], "v[184:185]", ax[2], 0); CUM_TR(xh[0][2], "v[186:187]", ax[2], 1024);  CUM_TR(xl[0][3], "v[188:189]", ax[3], 0); CUM_TR(xh[0][3], "v[190:191]", ax[3], 1024);
    CUM_TR(zl[1][0], "v[248:249]", az[0], 8192); CUM_TR(zh[1][0], "v[250:251]", az[0], 9216);  CUM_TR(zl[1][1], "v[252:253]", az[1], 8192); CUM_TR(zh[1][1], "v[254:255]", az[1], 9216);
    CUM_TR(xl[1][0], "v[208:209]", ax[0], 8192); CUM_TR(xh[1][0], "v[210:211]", ax[0], 9216);  CUM_TR(xl[1][1], "v[212:213]", ax[1], 8192); CUM_TR(xh[1][1], "v[214:215]", ax[1], 9216);
    CUM_TR(xl[1][2], "v[216:217]", ax[2], 8192); CUM_TR(xh[1][2], "v[218:219]", ax[2], 9216);  CUM_TR(xl[1][3], "v[220:221]", ax[3], 8192); CUM_TR(xh[1][3], "v[222:223]", ax[3], 9216);
    if (wk == 0 && s >= 1 && more1) {
      CUM_STAGE(2, s + 1, par ^ 1);
      CUM_STAGE(3, s + 1, par ^ 1);
    }
    CUM_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "={v[240:243]}"(zf[0][0]), "={v[244:247]}"(zf[0][1]), "={v[248:251]}"(zf[1][0]), "={v[252:255]}"(zf[1][1]),
                   "={v[176:179]}"(xf[0][0]), "={v[180:183]}"(xf[0][1]), "={v[184:187]}"(xf[0][2]), "={v[188:191]}"(xf[0][3]),
                   "={v[208:211]}"(xf[1][0]), "={v[212:215]}"(xf[1][1]), "={v[216:219]}"(xf[1][2]), "={v[220:223]}"(xf[1][3])
                 : "{v[240:241]}"(zl[0][0]), "{v[242:243]}"(zh[0][0]), "{v[244:245]}"(zl[0][1]), "{v[246:247]}"(zh[0][1]),
                   "{v[248:249]}"(zl[1][0]), "{v[250:251]}"(zh[1][0]), "{v[252:253]}"(zl[1][1]), "{v[254:255]}"(zh[1][1]),
                   "{v[176:177]}"(xl[0][0]), "{v[178:179]}"(xh[0][0]), "{v[180:181]}"(xl[0][1]), "{v[182:183]}"(xh[0][1]),
                   "{v[184:185]}"(xl[0][2]), "{v[186:187]}"(xh[0][2]), "{v[188:189]}"(xl[0][3]), "{v[190:191]}"(xh[0][3]),
                   "{v[208:209]}"(xl[1][0]), "{v[210:211]}"(xh[1][0]), "{v[212:213]}"(xl[1][1]), "{v[214:215]}"(xh[1][1]),
                   "{v[216:217]}"(xl[1][2]), "{v[218:219]}"(xh[1][2]), "{v[220:221]}"(xl[1][3]), "{v[222:223]}"(xh[1][3]) : "memory");
    CUM_QUADT(0, 0);                                                       // C1
    CUM_BIAS(0);
    CUM_BAR();
    // ---- L2
    CUM_TR(xl[0][4], "v[192:193]", ax[4], 0); CUM_TR(xh[0][4], "v[194:195]", ax[4], 1024);  CUM_TR(xl[0][5], "v[196:197]", ax[5], 0); CUM_TR(xh[0][5], "v[198:199]", ax[5], 1024);
    CUM_TR(xl[0][6], "v[200:201]", ax[6], 0); CUM_TR(xh[0][6], "v[202:203]", ax[6], 1024);  CUM_TR(xl[0][7], "v[204:205]", ax[7], 0); CUM_TR(xh[0][7], "v[206:207]", ax[7], 1024);
    CUM_TR(xl[1][4], "v[224:225]", ax[4], 8192); CUM_TR(xh[1][4], "v[226:227]", ax[4], 9216);  CUM_TR(xl[1][5], "v[228:229]", ax[5], 8192); CUM_TR(xh[1][5], "v[230:231]", ax[5], 9216);
    CUM_TR(xl[1][6], "v[232:233]", ax[6], 8192); CUM_TR(xh[1][6], "v[234:235]", ax[6], 9216);  CUM_TR(xl[1][7], "v[236:237]", ax[7], 8192); CUM_TR(xh[1][7], "v[238:239]", ax[7], 9216);
    CUM_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "={v[192:195]}"(xf[0][4]), "={v[196:199]}"(xf[0][5]), "={v[200:203]}"(xf[0][6]), "={v[204:207]}"(xf[0][7]), "={v[224:227]}"(xf[1][4]), "={v[228:231]}"(xf[1][5]), "={v[232:235]}"(xf[1][6]), "={v[236:239]}"(xf[1][7])
                 : "{v[192:193]}"(xl[0][4]), "{v[194:195]}"(xh[0][4]),
                   "{v[196:197]}"(xl[0][5]), "{v[198:199]}"(xh[0][5]),
                   "{v[200:201]}"(xl[0][6]), "{v[202:203]}"(xh[0][6]),
                   "{v[204:205]}"(xl[0][7]), "{v[206:207]}"(xh[0][7]),
                   "{v[224:225]}"(xl[1][4]), "{v[226:227]}"(xh[1][4]),
                   "{v[228:229]}"(xl[1][5]), "{v[230:231]}"(xh[1][5]),
                   "{v[232:233]}"(xl[1][6]), "{v[234:235]}"(xh[1][6]),
                   "{v[236:237]}"(xl[1][7]), "{v[238:239]}"(xh[1][7]) : "memory");
    CUM_QUADT(1, 0);                                                       // C2
    CUM_BAR();
    // ---- L3: dZ fragments of n columns 32-63 (same registers as columns 0-31); the X units are free: step s + 2
    CUM_TR(zl[0][0], "v[240:241]", az[2], 0); CUM_TR(zh[0][0], "v[242:243]", az[2], 1024);  CUM_TR(zl[0][1], "v[244:245]", az[3], 0); CUM_TR(zh[0][1], "v[246:247]", az[3], 1024);
    CUM_TR(zl[1][0], "v[248:249]", az[2], 8192); CUM_TR(zh[1][0], "v[250:251]", az[2], 9216);  CUM_TR(zl[1][1], "v[252:253]", az[3], 8192); CUM_TR(zh[1][1], "v[254:255]", az[3], 9216);
    if (more2) {
      CUM_STAGE(0, s + 2, par);
      if (wk != 0) CUM_STAGE(1, s + 2, par);
    }
    CUM_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "={v[240:243]}"(zf[0][0]), "={v[244:247]}"(zf[0][1]), "={v[248:251]}"(zf[1][0]), "={v[252:255]}"(zf[1][1])
                 : "{v[240:241]}"(zl[0][0]), "{v[242:243]}"(zh[0][0]),
                   "{v[244:245]}"(zl[0][1]), "{v[246:247]}"(zh[0][1]),
                   "{v[248:249]}"(zl[1][0]), "{v[250:251]}"(zh[1][0]),
                   "{v[252:253]}"(zl[1][1]), "{v[254:255]}"(zh[1][1]) : "memory");
    CUM_QUADT(1, 2);                                                       // C3
    CUM_BAR();
    // ---- L4
    if (wk == 0) {
      if (more2) CUM_STAGE(1, s + 2, par);
    } else {
      if (more1) {
        if (more2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (more2) {
        CUM_STAGE(2, s + 2, par);
        CUM_STAGE(3, s + 2, par);
      }
    }
    CUM_BAR();
    CUM_QUADT(0, 2);                                                       // C4
    CUM_BIAS(2);
    if (wk == 0 && more1) {
      if (more2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    CUM_BAR();
    const unsigned flip = par ? 0u - 65536u : 65536u;                   // the other parity's units
#pragma unroll
    for (int b = 0; b < 8; ++b) ax[b] += flip;
#pragma unroll
    for (int b = 0; b < 4; ++b) az[b] += flip;
  }
  if (wk == 0) CUM_BAR();                                                // group 1's last slot
#undef CUM_BAR
#undef CUM_QUADT
#undef CUM_TR
#undef CUM_MFMA
#undef CUM_HALFQ
#undef CUM_BIAS
#undef CUM_STAGE

#ifdef CUM_AB
  if (p.skip_store && acc[0][0][0][0] != 12345.f) return;
#endif
  // ---- slab store: lane holds D[k = kb + 4g + j][n = nb + r]
  float *slab = p.slab + (int64_t)sp * p.Np * p.Kp;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = n0 + wn * 64 + ni * 16 + r;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ki = 0; ki < 4; ++ki) {
        const int k = k0 + wk * 128 + h * 64 + ki * 16 + 4 * g;
        *reinterpret_cast<float4 *>(slab + (int64_t)n * p.Kp + k) =
            make_float4(acc[h][ni][ki][0], acc[h][ni][ki][1], acc[h][ni][ki][2], acc[h][ni][ki][3]);
      }
  }
  // selector rows 4 ni .. 4 ni + 3 (held by lane group g = ni) carry the column sums of fragment ni
  if (bias_wave) p.bslab[(int64_t)sp * p.Np + n0 + wn * 64 + 16 * g + r] = bacc[0];
}

// Slab reduction, parallel over outputs AND over slabs, fixed summation order (deterministic).
//   in : [S][rows][ld_in] f32        out: [gridDim.y][rows][ld_out] partial sums of S / gridDim.y slabs each
// A workgroup = 64 float4 outputs x 4 slab lanes; lanes are combined through LDS.  One launch carries two jobs (the
// weight-gradient slabs and the bias-gradient slabs): workgroups [0, nb0) serve job 0, the rest job 1.
struct ReduceJob {
  const float *in;
  float *out;
  int64_t in_slab, out_slab, ld_out;
  int ld_in, rows, cols4;
};

__global__ __launch_bounds__(256) void tn_reduce_kernel(const ReduceJob j0, const ReduceJob j1, int nb0, int S) {
  __shared__ float4 red[4][64];
  const bool first = (int)blockIdx.x < nb0;
  const ReduceJob &j = first ? j0 : j1;
  const int bx = first ? blockIdx.x : blockIdx.x - nb0;
  const int v = bx * 64 + (threadIdx.x & 63);
  const int sl = threadIdx.x >> 6;
  const int chunks = gridDim.y, ch = blockIdx.y;
  const int per = (S + chunks - 1) / chunks;
  const int s0 = ch * per;
  int s1 = s0 + per;
  s1 = s1 < S ? s1 : S;
  const int64_t total = (int64_t)j.rows * j.cols4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  int n = 0, k = 0;
  if (v < total) {
    n = v / j.cols4;
    k = (v % j.cols4) * 4;
    const float *base = j.in + (int64_t)n * j.ld_in + k;
    int s = s0 + sl;
    for (; s + 4 < s1; s += 8) {  // two independent chains keep two loads in flight
      const float4 x = *reinterpret_cast<const float4 *>(base + (int64_t)s * j.in_slab);
      const float4 y = *reinterpret_cast<const float4 *>(base + (int64_t)(s + 4) * j.in_slab);
      a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
      b.x += y.x; b.y += y.y; b.z += y.z; b.w += y.w;
    }
    for (; s < s1; s += 4) {
      const float4 x = *reinterpret_cast<const float4 *>(base + (int64_t)s * j.in_slab);
      a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
    }
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  red[sl][threadIdx.x & 63] = a;
  __syncthreads();
  if (sl == 0 && v < total) {
    float4 r = red[0][threadIdx.x];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      const float4 t = red[q][threadIdx.x];
      r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
    }
    *reinterpret_cast<float4 *>(j.out + (int64_t)ch * j.out_slab + (int64_t)n * j.ld_out + k) = r;
  }
}

// The same reduction writing its result where the PARAMETER lives (cum_gemm_tn_scatter): workgroups walk the float4 of
// the destination matrices (a parameter's gradient seen as rows x cols, row-major, contiguous) and gather their four
// sources from the GEMM-layout result through two small tables -- source position rowoff[r] + coloff[c] in the virtual
// [N][K] result, positions N K + n = the bias gradient -- so that the f32 result never makes the round trip through an
// arena in GEMM layout and a second, un-packing launch.  A conv weight (H, C, 4) reads its four taps as four 256-byte
// runs per wave and writes whole 1-KiB rows.  Fixed summation order (slab lanes through LDS, as above): deterministic.
struct ScatterJob {
  const float *in, *bin;     // weight slabs [S][rows_in][ld_in], bias slabs [S][ldb]
  float *dst;
  const int *ro, *co;
  int64_t in_slab, bin_slab;
  int ld_in, rows, cols4, K, NK, fold;
};

__global__ __launch_bounds__(256) void tn_reduce_scatter_kernel(const ScatterJob j0, const ScatterJob j1, int nb0, int S) {
  __shared__ float4 red[4][64];
  const bool first = (int)blockIdx.x < nb0;
  const ScatterJob &j = first ? j0 : j1;
  const int bx = first ? blockIdx.x : blockIdx.x - nb0;
  const int v = bx * 64 + (threadIdx.x & 63);
  const int sl = threadIdx.x >> 6;
  const int64_t total = (int64_t)j.rows * j.cols4;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  int r = 0, c = 0;
  if (v < total) {
    r = v / j.cols4;
    c = (v % j.cols4) * 4;
    const int base = j.ro[r];
    const float *src[4];
    int64_t stride[4];
    int fold[4];
    bool any_fold = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = base + j.co[c + i];
      fold[i] = 0;
      if (q < j.NK) {
        const int n = q / j.K;
        src[i] = j.in + (int64_t)n * j.ld_in + (q - n * j.K);
        stride[i] = j.in_slab;
      } else {
        src[i] = j.bin + (q - j.NK);
        stride[i] = j.bin_slab;
        fold[i] = j.fold;
        any_fold = any_fold || j.fold > 0;
      }
    }
    for (int s = sl; s < S; s += 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] += src[i][(int64_t)s * stride[i]];
      if (any_fold) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (fold[i] > 0) a[i] += src[i][(int64_t)s * stride[i] + fold[i]];
      }
    }
  }
  red[sl][threadIdx.x & 63] = make_float4(a[0], a[1], a[2], a[3]);
  __syncthreads();
  if (sl == 0 && v < total) {
    float4 rr = red[0][threadIdx.x];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      const float4 t = red[q][threadIdx.x];
      rr.x += t.x; rr.y += t.y; rr.z += t.z; rr.w += t.w;
    }
    *reinterpret_cast<float4 *>(j.dst + ((int64_t)r * j.cols4 * 4 + c)) = rr;
  }
}

}  // namespace cum

using namespace cum;

static int tn8_enabled() { return (int)cum_knob("CUM_TN8", 1); }      // AB build: 0 = the 128 x 128 kernel everywhere

// 256 x 256 tiles (gemm_tn8_kernel): 16-bit types, N and K multiples of 256.  One workgroup per CU, so the split count is
// the largest that keeps tiles x splits within ONE resident round of 256 workgroups.
static bool tn_use8(int64_t M, int32_t N, int32_t K, int32_t dtype) {
  return is16(dtype) && N % 256 == 0 && K % 256 == 0 && M >= 256 && tn8_enabled();
}

// The streaming kernel (gemm_tn_stream_kernel): 16-bit types, a padded result of exactly 128 x 256 or 256 x 128, and enough
// rows that 256 splits (one workgroup per CU, one resident round) still run >= 8 steps each.
static bool tn_use_stream(int64_t M, int32_t N, int32_t K, int32_t dtype) {
  if (!is16(dtype) || tn_use8(M, N, K, dtype) || cum_knob("CUM_TN_STREAM", 1) == 0) return false;
  const int np = (N + TN_T - 1) / TN_T, kp = (K + TN_T - 1) / TN_T;
  return np * kp == 2 && M >= 256 * 64 * 8;
}

static void tn_plan(int64_t M, int32_t N, int32_t K, int32_t dtype, int *Np, int *Kp, int *S, int *rps) {
  const int force = (int)cum_knob("CUM_TN_SPLITS", 0);             // AB build: pins the split count
  if (tn_use_stream(M, N, K, dtype)) {
    *Np = (N + TN_T - 1) / TN_T * TN_T;
    *Kp = (K + TN_T - 1) / TN_T * TN_T;
    int64_t rows = (M + 255) / 256;
    if (force > 0) rows = (M + force - 1) / force;
    rows = (rows + 63) / 64 * 64;
    *rps = (int)rows;
    *S = (int)((M + rows - 1) / rows);
    return;
  }
  if (tn_use8(M, N, K, dtype)) {
    *Np = N;
    *Kp = K;
    const int tiles = (N / 256) * (K / 256);
    int64_t want = tiles >= 256 ? 1 : 256 / tiles;
    if (force > 0) want = force;
    const int64_t max_s = (M + 255) / 256;              // at least 4 steps per split
    if (want > max_s) want = max_s;
    int64_t rows = (M + want - 1) / want;
    rows = (rows + 63) / 64 * 64;
    *rps = (int)rows;
    *S = (int)((M + rows - 1) / rows);
    return;
  }
  *Np = (N + TN_T - 1) / TN_T * TN_T;
  *Kp = (K + TN_T - 1) / TN_T * TN_T;
  const int tiles = (*Np / TN_T) * (*Kp / TN_T);
  const int bmk = is16(dtype) ? 32 : 16;
  // Splits are dealt to the 8 XCDs round-robin and an XCD holds 96 workgroups at a time (32 CUs x 3).  One split
  // with >= 96 tiles already fills its XCD: 8 splits.  Smaller tile counts take as many splits per XCD as fit into
  // ONE resident round (32 tiles -> 3 per XCD = 24 splits): a second, partly filled round costs a full round's
  // time (measured on the E8 shapes: -18...-27 % against "about 1024 workgroups"), and every extra split adds a
  // slab of Np x Kp floats to write and reduce.  CUM_TN_SPLITS pins the count for experiments.
  const int per_xcd = tiles >= 96 ? 1 : 96 / tiles;
  int64_t want = 8 * per_xcd;
  if (force > 0) want = force;
  const int64_t max_s = (M + 8 * bmk - 1) / (8 * bmk);  // at least 8 steps per split
  if (want > max_s) want = max_s;
  if (want < 1) want = 1;
  int64_t rows = (M + want - 1) / want;
  rows = (rows + bmk - 1) / bmk * bmk;
  *rps = (int)rows;
  *S = (int)((M + rows - 1) / rows);
  if (*S < 1) *S = 1;
}

static int reduce_chunks(int S) { return S >= 64 ? 16 : 1; }

extern "C" int64_t cum_gemm_tn_workspace_elems(int32_t dtype, int64_t M, int32_t N, int32_t K) {
  int Np, Kp, S, rps;
  tn_plan(M, N, K, dtype, &Np, &Kp, &S, &rps);
  const int64_t C = reduce_chunks(S);
  return (int64_t)S * Np * Kp + (int64_t)S * Np + C * (int64_t)N * K + C * (int64_t)Np;
}

extern "C" int cum_gemm_tn_tile(int32_t dtype, int64_t M, int32_t N, int32_t K) {
  return tn_use8(M, N, K, dtype) ? 256 : tn_use_stream(M, N, K, dtype) ? 384 : TN_T;
}

static int gemm_tn_impl(int32_t dtype, int64_t M, int32_t N, int32_t K, const void *dZ, int64_t ldz, const void *X,
                        int64_t ldx, float *dW, int64_t ldw, float *db, const cum_tn_scatter *sj, int nsj, float *workspace,
                        void *stream) {
  CUM_REQUIRE(dtype_ok(dtype), "gemm_tn: dtype must be CUM_F32, CUM_BF16 or CUM_F16");
  CUM_REQUIRE(dZ && X && (dW || sj) && workspace && M >= 0 && N > 0 && K > 0, "gemm_tn: bad argument");
  const int epc = is16(dtype) ? 8 : 4;
  CUM_REQUIRE(N % epc == 0 && K % epc == 0 && ldz % epc == 0 && ldx % epc == 0 && ldw % 4 == 0,
              "gemm_tn: N, K and strides must keep 16-byte alignment");
  CUM_REQUIRE(((uintptr_t)dZ & 15) == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)dW & 15) == 0,
              "gemm_tn: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  bool want_bias = db != nullptr;
  if (sj) {
    CUM_REQUIRE(nsj >= 1 && nsj <= 2, "gemm_tn_scatter: one or two destinations");
    for (int i = 0; i < nsj; ++i) {
      CUM_REQUIRE(sj[i].dst && sj[i].rowoff && sj[i].coloff && sj[i].rows > 0 && sj[i].cols > 0 && sj[i].cols % 4 == 0 &&
                      ((uintptr_t)sj[i].dst & 15) == 0,
                  "gemm_tn_scatter: destinations are 16-byte aligned matrices with a multiple of 4 columns");
      want_bias = want_bias || sj[i].reads_bias;
    }
  }
  if (M == 0) {
    if (sj) {
      for (int i = 0; i < nsj; ++i) (void)hipMemsetAsync(sj[i].dst, 0, sizeof(float) * (size_t)sj[i].rows * sj[i].cols, st);
      return CUM_OK;
    }
    (void)hipMemset2DAsync(dW, sizeof(float) * ldw, 0, sizeof(float) * K, N, st);
    if (db) (void)hipMemsetAsync(db, 0, sizeof(float) * N, st);
    return CUM_OK;
  }
  int Np, Kp, S, rps;
  tn_plan(M, N, K, dtype, &Np, &Kp, &S, &rps);
  TnParams p{};
  p.dZ = dZ; p.X = X; p.ldz = ldz; p.ldx = ldx; p.M = M; p.N = N; p.K = K; p.Np = Np; p.Kp = Kp;
  p.rows_per_split = rps;
  p.nsplit = S;
  p.slab = workspace;
  p.bslab = want_bias ? workspace + (int64_t)S * Np * Kp : nullptr;
#ifdef CUM_AB
  p.skip_store = (int)cum_knob("CUM_TN_NOSTORE", 0);
#endif
  dim3 grid(8 * (Np / TN_T) * (Kp / TN_T) * ((S + 7) / 8)), block(256);
  if (tn_use8(M, N, K, dtype)) {
    const int items = (N / 256) * (K / 256) * S;
    const dim3 grid8(8 * ((items + 7) / 8)), block8(512);
#ifdef CUM_AB
    if (cum_knob("CUM_TN9", 1) == 0) {
      if (dtype == CUM_BF16) hipLaunchKernelGGL(gemm_tn8_kernel<__bf16>, grid8, block8, 0, st, p);
      else hipLaunchKernelGGL(gemm_tn8_kernel<f16>, grid8, block8, 0, st, p);
    } else
#endif
    if (dtype == CUM_BF16)
      hipLaunchKernelGGL(gemm_tn9_kernel<__bf16>, grid8, block8, 0, st, p);
    else
      hipLaunchKernelGGL(gemm_tn9_kernel<f16>, grid8, block8, 0, st, p);
  } else if (tn_use_stream(M, N, K, dtype)) {
    CUM_REQUIRE((int64_t)rps * (ldz > ldx ? ldz : ldx) + 512 < (int64_t)1 << 31, "gemm_tn: a row split exceeds 2^31 elements");
    const dim3 gs(S), bs(512);
    if (Np == 128) {
      if (dtype == CUM_BF16) hipLaunchKernelGGL((gemm_tn_stream_kernel<__bf16, 128, 256>), gs, bs, 0, st, p);
      else hipLaunchKernelGGL((gemm_tn_stream_kernel<f16, 128, 256>), gs, bs, 0, st, p);
    } else {
      if (dtype == CUM_BF16) hipLaunchKernelGGL((gemm_tn_stream_kernel<__bf16, 256, 128>), gs, bs, 0, st, p);
      else hipLaunchKernelGGL((gemm_tn_stream_kernel<f16, 256, 128>), gs, bs, 0, st, p);
    }
  } else if (dtype == CUM_BF16)
    hipLaunchKernelGGL(gemm_tn_kernel<__bf16>, grid, block, 0, st, p);
  else if (dtype == CUM_F16)
    hipLaunchKernelGGL(gemm_tn_kernel<f16>, grid, block, 0, st, p);
  else
    hipLaunchKernelGGL(gemm_tn_kernel<float>, grid, block, 0, st, p);
  CUM_CHECK_LAUNCH();
#ifdef CUM_AB
  if (cum_knob("CUM_TN_NOREDUCE", 0)) return CUM_OK;      // timing experiment: the slabs are not combined
#endif
  // combine the slabs: one pass for few slabs, two passes (16 partial sums, then those) for many
  const int C = reduce_chunks(S);
  float *part = workspace + (int64_t)S * Np * Kp + (int64_t)S * Np;
  float *bpart = part + (int64_t)C * N * K;
  const int cols4 = K / 4, bc4 = N / 4;
  const int gx = (int)(((int64_t)N * cols4 + 63) / 64), bx = want_bias ? (bc4 + 63) / 64 : 0;
  // the last pass in scatter form: destination matrices instead of dW / db (one launch, up to two destinations)
  auto scatter_pass = [&](const float *win, int64_t wslab, int ldin, const float *bin, int64_t bslab, int Sx) {
    ScatterJob js[2];
    int nb[2] = {0, 0};
    for (int i = 0; i < 2; ++i) {
      const cum_tn_scatter &q = sj[i < nsj ? i : 0];
      js[i] = ScatterJob{win, bin, q.dst, q.rowoff, q.coloff, wslab, bslab, ldin, q.rows, q.cols / 4, K, N * K, q.bias_fold};
      nb[i] = i < nsj ? (int)(((int64_t)q.rows * (q.cols / 4) + 63) / 64) : 0;
    }
    hipLaunchKernelGGL(tn_reduce_scatter_kernel, dim3(nb[0] + nb[1], 1), dim3(256), 0, st, js[0], js[1], nb[0], Sx);
  };
  // job 0: dW slabs [S][Np][Kp]; job 1: bias slabs [S][Np] seen as one row of Np / 4 float4
  if (C == 1) {
    if (sj) {
      scatter_pass(p.slab, (int64_t)Np * Kp, Kp, p.bslab, (int64_t)Np, S);
    } else {
      const ReduceJob w{p.slab, dW, (int64_t)Np * Kp, 0, ldw, Kp, N, cols4};
      const ReduceJob bj{p.bslab, db, (int64_t)Np, 0, N, Np, 1, bc4};
      hipLaunchKernelGGL(tn_reduce_kernel, dim3(gx + bx, 1), dim3(256), 0, st, w, bj, gx, S);
    }
  } else {
    const ReduceJob w1{p.slab, part, (int64_t)Np * Kp, (int64_t)N * K, (int64_t)K, Kp, N, cols4};
    const ReduceJob b1{p.bslab, bpart, (int64_t)Np, (int64_t)Np, (int64_t)Np, Np, 1, bc4};
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(gx + bx, C), dim3(256), 0, st, w1, b1, gx, S);
    if (sj) {
      scatter_pass(part, (int64_t)N * K, K, bpart, (int64_t)Np, C);
    } else {
      const ReduceJob w2{part, dW, (int64_t)N * K, 0, ldw, K, N, cols4};
      const ReduceJob b2{bpart, db, (int64_t)Np, 0, N, Np, 1, bc4};
      hipLaunchKernelGGL(tn_reduce_kernel, dim3(gx + bx, 1), dim3(256), 0, st, w2, b2, gx, C);
    }
  }
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

extern "C" int cum_gemm_tn(int32_t dtype, int64_t M, int32_t N, int32_t K, const void *dZ, int64_t ldz, const void *X,
                           int64_t ldx, float *dW, int64_t ldw, float *db, float *workspace, void *stream) {
  CUM_REQUIRE(dW, "gemm_tn: bad argument");
  return gemm_tn_impl(dtype, M, N, K, dZ, ldz, X, ldx, dW, ldw, db, nullptr, 0, workspace, stream);
}

extern "C" int cum_gemm_tn_scatter(int32_t dtype, int64_t M, int32_t N, int32_t K, const void *dZ, int64_t ldz, const void *X,
                                   int64_t ldx, const cum_tn_scatter *jobs, int32_t njobs, float *workspace, void *stream) {
  CUM_REQUIRE(jobs, "gemm_tn_scatter: bad argument");
  return gemm_tn_impl(dtype, M, N, K, dZ, ldz, X, ldx, nullptr, 0, nullptr, jobs, njobs, workspace, stream);
}
