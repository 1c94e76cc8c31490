// Selective scan forward + single-step update for gfx950.
//
// Replaces selective_scan_cuda.fwd of mamba-ssm 1.2.2 (reached from the reference via
// create_block -> Mamba.forward, src/network/CleanUMamba.py:172-189, 289-290) and the
// Triton selective_state_update used by Mamba.step (CleanUMamba.py:451-454).
// Semantics: SURVEY.md Appendix A.2.  Mapping: scan_common.h.
//
// At d_state = 64 the kernel is bound by v_exp_f32 issue (one per state update), not
// by HBM -- see DESIGN.md "scan roofline".
#include <stdlib.h>
#include "scan_common.h"

namespace cum {

template <int NW, bool FAST, typename TIO>
__global__ __launch_bounds__(NW * 64) void scan_fwd_kernel(const ScanParams p) {
  constexpr int K = (TB + NW - 1) / NW;  // (t, d) rows per thread in phases A / C
  __shared__ float s_dt[TB][64];
  __shared__ float s_du[TB][64];
  __shared__ float s_y[NW][TB][64];

  const int lane = threadIdx.x & 63;
  const int w = uniform(threadIdx.x >> 6);
  const int b = blockIdx.y;
  const int d = blockIdx.x * 64 + lane;
  const int N = p.s.dstate, L = p.s.len, Dm = p.s.dim;
  const bool dok = d < Dm;
  const int dc = dok ? d : Dm - 1;
  const int n0 = w * NS;
  const int nvalid = (N - n0) < NS ? (N - n0) : NS;

  float Ap[NS], x[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int jj = j < nvalid ? j : nvalid - 1;
    const float a = p.A[(int64_t)dc * N + n0 + jj] * kLog2e;
    Ap[j] = (j < nvalid) ? a : 0.f;
    x[j] = 0.f;
  }
  const float Dd = p.D ? p.D[dc] : 0.f;
  const float bias = p.bias ? p.bias[dc] : 0.f;
  const TIO *up = static_cast<const TIO *>(p.u) + b * p.s.u_sb + dc * p.s.u_sd;
  const TIO *dtp = static_cast<const TIO *>(p.delta) + b * p.s.dt_sb + dc * p.s.dt_sd;
  const bool has_z = p.z != nullptr;
  const TIO *zp = has_z ? static_cast<const TIO *>(p.z) + b * p.s.z_sb + dc * p.s.z_sd : up;  // !has_z: valid dummy address
  TIO *op = static_cast<TIO *>(p.out) + b * p.s.o_sb + dc * p.s.o_sd;
  const float *Bw = p.Bm + b * p.s.B_sb + n0 * p.s.B_sn;
  const float *Cw = p.Cm + b * p.s.C_sb + n0 * p.s.C_sn;
  // within-batch offsets fit 32 bits (checked on the host)
  const int u_sl = (int)p.s.u_sl, dt_sl = (int)p.s.dt_sl, z_sl = has_z ? (int)p.s.z_sl : (int)p.s.u_sl;
  const int o_sl = (int)p.s.o_sl;
  const int B_sl = (int)p.s.B_sl, C_sl = (int)p.s.C_sl, B_sn = (int)p.s.B_sn, C_sn = (int)p.s.C_sn;
  const int softplus = p.s.delta_softplus;

  float ru[K], rdt[K], rz[K];
  auto load_rows = [&](int t0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      int t = t0 + w + k * NW;
      t = t < L ? t : L - 1;  // clamped address; the value is masked in phase A
      ru[k] = (float)up[t * u_sl];
      rdt[k] = (float)dtp[t * dt_sl];
      rz[k] = (float)zp[t * z_sl];
    }
  };
  load_rows(0);

  const int nchunks = p.nchunks;
  for (int c = 0; c < nchunks; ++c) {
    const int t0 = c * TB;
    float eu[K], ez[K];
    // ---- phase A: delta' = softplus(delta + bias), du = delta' * u  -> LDS
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int tl = w + k * NW;
      if (tl < TB) {
        const bool ok = dok && (t0 + tl) < L;
        float dtv = rdt[k] + bias;
        if (softplus) dtv = softplus20(dtv);
        dtv = ok ? dtv : 0.f;
        s_dt[tl][lane] = dtv;
        s_du[tl][lane] = ok ? dtv * ru[k] : 0.f;
      }
      eu[k] = ru[k];
      ez[k] = rz[k];
    }
    if (c + 1 < nchunks) load_rows(t0 + TB);  // prefetch the next chunk's rows
    if (p.ckpt && dok) ckpt_store(p.ckpt, ckpt_slot(b, nchunks, c, 0, NW, w, Dm, d), x);
    __syncthreads();
    // ---- phase B: 16 sequential steps.  Operands (s_load for B/C, ds_read for delta'/du) are fetched TWO steps
    //      at a time, one pair ahead of their use: scalar loads return out of order, so any wait on them is a
    //      wait for all of them -- batching by pairs gives each wait two full steps of compute to hide behind.
    struct PairOps {
      float bv[2][NS], cv[2][NS];
      float dt[2], du[2];
    };
    const float *bp = Bw + t0 * B_sl, *cp = Cw + t0 * C_sl;  // t0 < L always
    auto fetch_pair = [&](int tl, PairOps &o) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        load_bc<FAST>(bp, B_sn, nvalid, o.bv[h]);
        load_bc<FAST>(cp, C_sn, nvalid, o.cv[h]);
        o.dt[h] = s_dt[tl + h][lane];
        o.du[h] = s_du[tl + h][lane];
        const int inc = (t0 + tl + h + 1 < L) ? 1 : 0;  // address clamps at the last valid row
        bp = opaque(bp + inc * B_sl);
        cp = opaque(cp + inc * C_sl);
      }
    };
    PairOps cur, nxt;
    fetch_pair(0, cur);
#pragma unroll
    for (int tp = 0; tp < TB; tp += 2) {
      if (tp + 2 < TB) fetch_pair(tp + 2, nxt);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float dt = cur.dt[h], du = cur.du[h];
        float y = 0.f;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          const float a = __builtin_amdgcn_exp2f(dt * Ap[j]);
          x[j] = fmaf(a, x[j], du * cur.bv[h][j]);
          y = fmaf(cur.cv[h][j], x[j], y);
        }
        s_y[w][tp + h][lane] = y;
        if (tp + h == SUB - 1 && p.ckpt && dok)   // state entering the second half of the chunk
          ckpt_store(p.ckpt, ckpt_slot(b, nchunks, c, 1, NW, w, Dm, d), x);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (tp + 2 < TB) cur = nxt;
    }
    __syncthreads();
    // ---- phase C: sum partial y over the state slices, skip term, gate, store
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int tl = w + k * NW;
      const int t = t0 + tl;
      if (tl < TB && t < L && dok) {
        float y = Dd * eu[k];
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) y += s_y[ww][tl][lane];
        if (has_z) {
          const float zv = ez[k];
          y *= zv * sigmoidf_(zv);
        }
        op[t * o_sl] = (TIO)y;
      }
    }
    // no third barrier: the next phase A writes s_dt/s_du (read only in phase B, which
    // ended at the barrier above) and the next phase B writes s_y after the next
    // phase-A barrier, which every wave reaches only after this phase C.
  }
  if (p.last_state && dok) {
    float *ls = p.last_state + ((int64_t)b * Dm + d) * N + n0;
#pragma unroll
    for (int j = 0; j < NS; ++j)
      if (j < nvalid) ls[j] = x[j];
  }
}

// Variant with B_t / C_t staged through LDS instead of scalar loads: phase A loads the chunk's [16][N] B and C
// tiles with coalesced vector loads (any strides, padding masked to zero), phase B reads each wave's 8-float slice
// with wave-uniform ds_read_b128 (LDS broadcast), one step ahead.  LDS returns in order, so waits are counted and
// nothing in the step loop waits on the scalar cache.
template <int NW, typename TIO>
__global__ __launch_bounds__(NW * 64) void scan_fwd_lds_kernel(const ScanParams p) {
  constexpr int K = (TB + NW - 1) / NW;
  constexpr int NT = NW * 64;
  constexpr int NP = NW * NS;                    // padded state count
  constexpr int BCK = (TB * NP + NT - 1) / NT;   // B (and C) elements per thread per chunk
  __shared__ float s_dt[TB][64];
  __shared__ float s_du[TB][64];
  __shared__ float s_y[NW][TB][64];
  __shared__ __attribute__((aligned(16))) float s_B[TB][NP];
  __shared__ __attribute__((aligned(16))) float s_C[TB][NP];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = uniform(tid >> 6);
  const int b = blockIdx.y;
  const int d = blockIdx.x * 64 + lane;
  const int N = p.s.dstate, L = p.s.len, Dm = p.s.dim;
  const bool dok = d < Dm;
  const int dc = dok ? d : Dm - 1;
  const int n0 = w * NS;
  const int nvalid = (N - n0) < NS ? (N - n0) : NS;

  f2 Ap[NS / 2], x[NS / 2];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int jj = j < nvalid ? j : nvalid - 1;
    const float a = p.A[(int64_t)dc * N + n0 + jj] * kLog2e;
    Ap[j / 2][j % 2] = (j < nvalid) ? a : 0.f;
    x[j / 2][j % 2] = 0.f;
  }
  const float Dd = p.D ? p.D[dc] : 0.f;
  const float bias = p.bias ? p.bias[dc] : 0.f;
  const TIO *up = static_cast<const TIO *>(p.u) + b * p.s.u_sb + dc * p.s.u_sd;
  const TIO *dtp = static_cast<const TIO *>(p.delta) + b * p.s.dt_sb + dc * p.s.dt_sd;
  const bool has_z = p.z != nullptr;
  const TIO *zp = has_z ? static_cast<const TIO *>(p.z) + b * p.s.z_sb + dc * p.s.z_sd : up;
  TIO *op = static_cast<TIO *>(p.out) + b * p.s.o_sb + dc * p.s.o_sd;
  TIO *yp = p.ypre ? static_cast<TIO *>(p.ypre) + b * p.s.o_sb + dc * p.s.o_sd : nullptr;   // y before the gate, kept for the backward
  const float *Bb = p.Bm + b * p.s.B_sb, *Cb = p.Cm + b * p.s.C_sb;
  const int u_sl = (int)p.s.u_sl, dt_sl = (int)p.s.dt_sl, z_sl = has_z ? (int)p.s.z_sl : (int)p.s.u_sl;
  const int o_sl = (int)p.s.o_sl;
  const int B_sl = (int)p.s.B_sl, C_sl = (int)p.s.C_sl, B_sn = (int)p.s.B_sn, C_sn = (int)p.s.C_sn;
  const int softplus = p.s.delta_softplus;

  float ru[K], rdt[K], rz[K], rb[BCK], rc[BCK];
  auto load_rows = [&](int t0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      int t = t0 + w + k * NW;
      t = t < L ? t : L - 1;
      ru[k] = (float)up[t * u_sl];
      rdt[k] = (float)dtp[t * dt_sl];
      rz[k] = (float)zp[t * z_sl];
    }
#pragma unroll
    for (int k = 0; k < BCK; ++k) {
      const int e = tid + k * NT;
      const int tl = e / NP, n = e % NP;
      int t = t0 + tl;
      t = t < L ? t : L - 1;
      const int nc = n < N ? n : N - 1;
      const float bvv = Bb[t * B_sl + nc * B_sn], cvv = Cb[t * C_sl + nc * C_sn];
      rb[k] = n < N ? bvv : 0.f;
      rc[k] = n < N ? cvv : 0.f;
    }
  };
  load_rows(0);

  const int nchunks = p.nchunks;
  for (int c = 0; c < nchunks; ++c) {
    const int t0 = c * TB;
    float eu[K], ez[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int tl = w + k * NW;
      if (tl < TB) {
        const bool ok = dok && (t0 + tl) < L;
        float dtv = rdt[k] + bias;
        if (softplus) dtv = softplus20(dtv);
        dtv = ok ? dtv : 0.f;
        s_dt[tl][lane] = dtv;
        s_du[tl][lane] = ok ? dtv * ru[k] : 0.f;
      }
      eu[k] = ru[k];
      ez[k] = rz[k];
    }
#pragma unroll
    for (int k = 0; k < BCK; ++k) {
      const int e = tid + k * NT;
      if (e < TB * NP) {
        (&s_B[0][0])[e] = rb[k];
        (&s_C[0][0])[e] = rc[k];
      }
    }
    if (c + 1 < nchunks) load_rows(t0 + TB);
    if (p.ckpt && dok) ckpt_store(p.ckpt, ckpt_slot(b, nchunks, c, 0, NW, w, Dm, d), x);
    __syncthreads();
    float4 b0 = *reinterpret_cast<const float4 *>(&s_B[0][n0]), b1 = *reinterpret_cast<const float4 *>(&s_B[0][n0 + 4]);
    float4 c0 = *reinterpret_cast<const float4 *>(&s_C[0][n0]), c1 = *reinterpret_cast<const float4 *>(&s_C[0][n0 + 4]);
    float dt = s_dt[0][lane], du = s_du[0][lane];
#pragma unroll
    for (int tl = 0; tl < TB; ++tl) {
      float4 nb0 = b0, nb1 = b1, nc0 = c0, nc1 = c1;
      float ndt = 0.f, ndu = 0.f;
      if (tl + 1 < TB) {
        nb0 = *reinterpret_cast<const float4 *>(&s_B[tl + 1][n0]);
        nb1 = *reinterpret_cast<const float4 *>(&s_B[tl + 1][n0 + 4]);
        nc0 = *reinterpret_cast<const float4 *>(&s_C[tl + 1][n0]);
        nc1 = *reinterpret_cast<const float4 *>(&s_C[tl + 1][n0 + 4]);
        ndt = s_dt[tl + 1][lane];
        ndu = s_du[tl + 1][lane];
      }
      // pairs of states -> v_pk_mul_f32 / v_pk_fma_f32 (two state updates per issue slot); even and odd states
      // are summed apart and joined once per step
      const f2 bv[NS / 2] = {f2{b0.x, b0.y}, f2{b0.z, b0.w}, f2{b1.x, b1.y}, f2{b1.z, b1.w}};
      const f2 cv[NS / 2] = {f2{c0.x, c0.y}, f2{c0.z, c0.w}, f2{c1.x, c1.y}, f2{c1.z, c1.w}};
      f2 y = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NS / 2; ++j) {
        const f2 e = dt * Ap[j];
        f2 a;
        a.x = __builtin_amdgcn_exp2f(e.x);
        a.y = __builtin_amdgcn_exp2f(e.y);
        x[j] = a * x[j] + du * bv[j];
        y = cv[j] * x[j] + y;
      }
      s_y[w][tl][lane] = y.x + y.y;
      if (tl == SUB - 1 && p.ckpt && dok)   // state entering the second half of the chunk
        ckpt_store(p.ckpt, ckpt_slot(b, nchunks, c, 1, NW, w, Dm, d), x);
      __builtin_amdgcn_sched_barrier(0);
      b0 = nb0; b1 = nb1; c0 = nc0; c1 = nc1; dt = ndt; du = ndu;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int tl = w + k * NW;
      const int t = t0 + tl;
      if (tl < TB && t < L && dok) {
        float y = Dd * eu[k];
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) y += s_y[ww][tl][lane];
        if (yp) yp[t * o_sl] = (TIO)y;
        if (has_z) {
          const float zv = ez[k];
          y *= zv * sigmoidf_(zv);
        }
        op[t * o_sl] = (TIO)y;
      }
    }
  }
  if (p.last_state && dok) {
    float *ls = p.last_state + ((int64_t)b * Dm + d) * N + n0;
#pragma unroll
    for (int j = 0; j < NS; ++j)
      if (j < nvalid) ls[j] = x[j / 2][j % 2];
  }
}

// d_state <= 16 (the 442K model, every pruned checkpoint): a wave holds ALL states of its 64 channels, so nothing is
// shared between waves -- no LDS exchange of partial sums, no workgroup barrier, one wave per workgroup.  The grid
// then only has batch * dim / 64 waves (half the chip's SIMDs at B = 16, D = 2048), so the kernel is built to run
// alone on its SIMD: the (t, d) rows of the NEXT block of PB steps are already in flight in registers while the
// current block is computed (PB * ~260 issue cycles ~ 2 us of cover), and the block's B_t / C_t rows are staged
// through a wave-private LDS tile (written and read by the same wave: s_waitcnt, no barrier) and fetched back as
// broadcast ds_read_b128.  Checkpoints are written in the layout of the NW-wave kernels, so the backward is shared.
// PB = 16: a whole chunk of rows in flight (the wave that runs alone on its SIMD); PB = 8: half a chunk -- 48 registers
// fewer (164 -> <= 128 at d_state 8), i.e. FOUR waves per SIMD instead of three, for grids that bring more than three
// waves per SIMD anyway (batch 128 at D = 2048: 4096 waves = one resident round instead of a 3 + 1 split).
template <int NW, typename TIO, int PB>
__global__ __launch_bounds__(64) void scan_fwd_small_kernel(const ScanParams p) {
  constexpr int NPD = NW * NS;           // padded state count (8 or 16)
  constexpr int NP2 = NPD / 2;
  constexpr int BCE = PB * 2 * NPD / 64; // B and C elements per lane and block
  constexpr int NBLK = TB / PB;          // blocks per 16-step chunk: checkpoints fall on chunk starts and middles
  static_assert(PB == TB || PB == SUB, "a block is a chunk or one of its halves");
  static_assert(BCE >= 1, "the B / C tile of a block gives every lane at least one element");
  __shared__ __attribute__((aligned(16))) float s_bc[2][PB][2 * NPD];

  const int lane = threadIdx.x;
  const int b = blockIdx.y;
  const int d = blockIdx.x * 64 + lane;
  const int N = p.s.dstate, L = p.s.len, Dm = p.s.dim;
  const bool dok = d < Dm;
  const int dc = dok ? d : Dm - 1;

  f2 Ap[NP2], x[NP2];
#pragma unroll
  for (int j = 0; j < NPD; ++j) {
    const int jj = j < N ? j : N - 1;
    const float a = p.A[(int64_t)dc * N + jj] * kLog2e;
    Ap[j / 2][j % 2] = j < N ? a : 0.f;
    x[j / 2][j % 2] = 0.f;
  }
  const float Dd = p.D ? p.D[dc] : 0.f;
  const float bias = p.bias ? p.bias[dc] : 0.f;
  const TIO *up = static_cast<const TIO *>(p.u) + b * p.s.u_sb + dc * p.s.u_sd;
  const TIO *dtp = static_cast<const TIO *>(p.delta) + b * p.s.dt_sb + dc * p.s.dt_sd;
  const bool has_z = p.z != nullptr;
  const TIO *zp = has_z ? static_cast<const TIO *>(p.z) + b * p.s.z_sb + dc * p.s.z_sd : up;
  TIO *op = static_cast<TIO *>(p.out) + b * p.s.o_sb + dc * p.s.o_sd;
  const float *Bb = p.Bm + b * p.s.B_sb, *Cb = p.Cm + b * p.s.C_sb;
  const int u_sl = (int)p.s.u_sl, dt_sl = (int)p.s.dt_sl, z_sl = has_z ? (int)p.s.z_sl : (int)p.s.u_sl;
  const int o_sl = (int)p.s.o_sl;
  const int B_sl = (int)p.s.B_sl, C_sl = (int)p.s.C_sl, B_sn = (int)p.s.B_sn, C_sn = (int)p.s.C_sn;
  const int softplus = p.s.delta_softplus;
  const int nchunks = p.nchunks;

  float ru[PB], rdt[PB], rz[PB], rbc[BCE];
  auto load_block = [&](int t0) {
#pragma unroll
    for (int k = 0; k < PB; ++k) {
      int t = t0 + k;
      t = t < L ? t : L - 1;               // clamped address; out-of-range steps are never stored
      ru[k] = (float)up[t * u_sl];
      rdt[k] = (float)dtp[t * dt_sl];
      rz[k] = (float)zp[t * z_sl];
    }
#pragma unroll
    for (int k = 0; k < BCE; ++k) {
      const int e = lane + 64 * k;         // (step, column) of the [PB][B | C] tile
      const int tl = e / (2 * NPD), j = e % (2 * NPD);
      int t = t0 + tl;
      t = t < L ? t : L - 1;
      const bool isC = j >= NPD;
      const int n = isC ? j - NPD : j;
      const int nc = n < N ? n : N - 1;
      const float v = isC ? Cb[t * C_sl + nc * C_sn] : Bb[t * B_sl + nc * B_sn];
      rbc[k] = n < N ? v : 0.f;
    }
  };
  load_block(0);

  const int nblocks = nchunks * NBLK;
  for (int blk = 0; blk < nblocks; ++blk) {
    const int t0 = blk * PB;
    const int c = blk / NBLK;
    float (*tile)[2 * NPD] = s_bc[blk & 1];
    float cu[PB], cdt[PB], cz[PB];
#pragma unroll
    for (int k = 0; k < PB; ++k) {
      cu[k] = ru[k]; cdt[k] = rdt[k]; cz[k] = rz[k];
    }
#pragma unroll
    for (int k = 0; k < BCE; ++k) (&tile[0][0])[lane + 64 * k] = rbc[k];
    if (blk + 1 < nblocks) load_block(t0 + PB);      // the next block's rows travel while this one is computed
    // (same wave wrote the tile: LDS operations of one wave complete in order, the reads below see the writes)
#pragma unroll
    for (int k = 0; k < PB; ++k) {
      if ((k == 0 || (PB == TB && k == SUB)) && p.ckpt && dok) {
        const int half = PB == TB ? (k == 0 ? 0 : 1) : (blk % NBLK);
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const f2 (&xs)[NS / 2] = *reinterpret_cast<const f2 (*)[NS / 2]>(&x[w * (NS / 2)]);
          ckpt_store(p.ckpt, ckpt_slot(b, nchunks, c, half, NW, w, Dm, d), xs);
        }
      }
      float dtv = cdt[k] + bias;
      if (softplus) dtv = softplus20(dtv);
      dtv = (t0 + k < L) ? dtv : 0.f;                // steps past the end leave the state alone (a = 1, b = 0)
      const float uv = cu[k];
      const float du = dtv * uv;
      f2 y = {0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NPD / 4; ++q) {            // four states (two pairs) per 16-byte broadcast read
        const float4 bq = *reinterpret_cast<const float4 *>(&tile[k][4 * q]);
        const float4 cq = *reinterpret_cast<const float4 *>(&tile[k][NPD + 4 * q]);
        const f2 bv[2] = {f2{bq.x, bq.y}, f2{bq.z, bq.w}}, cv[2] = {f2{cq.x, cq.y}, f2{cq.z, cq.w}};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = 2 * q + h;
          const f2 e = dtv * Ap[j];
          f2 a;
          a.x = __builtin_amdgcn_exp2f(e.x);
          a.y = __builtin_amdgcn_exp2f(e.y);
          x[j] = a * x[j] + du * bv[h];
          y = cv[h] * x[j] + y;
        }
      }
      float yv = y.x + y.y + Dd * uv;
      if (has_z) {
        const float zv = cz[k];
        yv *= zv * sigmoidf_(zv);
      }
      if (dok && t0 + k < L) op[(t0 + k) * o_sl] = (TIO)yv;
    }
  }
  if (p.last_state && dok) {
    float *ls = p.last_state + ((int64_t)b * Dm + d) * N;
#pragma unroll
    for (int j = 0; j < NPD; ++j)
      if (j < N) ls[j] = x[j / 2][j % 2];
  }
}

// Wave-specialised form of the same small-d_state scan.  With all states of a channel in one wave, the work that
// depends on (t, d) only -- softplus (exp, log, rcp), the SiLU gate (exp, rcp), delta*u, D*u, the loads and their
// address arithmetic: ~155 issue cycles per step -- costs MORE than the eight state updates it feeds (~140), and the grid
// (batch * dim / 64 workgroups) leaves half of the chip's SIMDs idle at B = 16, D = 2048.  So a workgroup is one
// PRODUCER wave + NW CONSUMER waves (8 states each) on different SIMDs of a CU: the producer prepares block p + 1 --
// rows already in flight in its registers, softplus / gate / skip term into an LDS slot, B_t / C_t tile staged --
// while the consumers walk block p's recurrence and write the output themselves.  One workgroup barrier per 16-step
// block; both roles cost about the same per step, so the step time roughly halves.
template <int NW, typename TIO>
__global__ __launch_bounds__((NW + 1) * 64) void scan_fwd_ws_kernel(const ScanParams p) {
  constexpr int PB = TB;
  constexpr int NPD = NW * NS;
  constexpr int BCE = PB * 2 * NPD / 64;
  __shared__ float s_dt[2][PB][64], s_du[2][PB][64], s_sk[2][PB][64], s_gt[2][PB][64];
  __shared__ float s_y0[NW == 2 ? 2 : 1][NW == 2 ? PB : 1][64];   // NW == 2: partial sums of consumer 0
  __shared__ __attribute__((aligned(16))) float s_bc[2][PB][2 * NPD];

  const int lane = threadIdx.x & 63;
  const int w = uniform(threadIdx.x >> 6);
  const bool producer = w == NW;
  const int b = blockIdx.y;
  const int d = blockIdx.x * 64 + lane;
  const int N = p.s.dstate, L = p.s.len, Dm = p.s.dim;
  const bool dok = d < Dm;
  const int dc = dok ? d : Dm - 1;       // lanes past the last channel walk a valid one; only their stores are masked
  const int nchunks = p.nchunks;
  const bool has_z = p.z != nullptr;

  if (producer) {
    const float Dd = p.D ? p.D[dc] : 0.f;
    const float bias = p.bias ? p.bias[dc] : 0.f;
    // wave-uniform row bases + one per-lane 32-bit offset: loads take the (SGPR base, VGPR offset) form and the per-row
    // address arithmetic stays on the scalar unit
    const TIO *ub = static_cast<const TIO *>(p.u) + b * p.s.u_sb;
    const TIO *db = static_cast<const TIO *>(p.delta) + b * p.s.dt_sb;
    const TIO *zb = has_z ? static_cast<const TIO *>(p.z) + b * p.s.z_sb : ub;
    const int u_lo = dc * (int)p.s.u_sd, d_lo = dc * (int)p.s.dt_sd, z_lo = has_z ? dc * (int)p.s.z_sd : u_lo;
    const float *Bb = p.Bm + b * p.s.B_sb, *Cb = p.Cm + b * p.s.C_sb;
    const int u_sl = (int)p.s.u_sl, dt_sl = (int)p.s.dt_sl, z_sl = has_z ? (int)p.s.z_sl : (int)p.s.u_sl;
    const int B_sl = (int)p.s.B_sl, C_sl = (int)p.s.C_sl, B_sn = (int)p.s.B_sn, C_sn = (int)p.s.C_sn;
    const int softplus = p.s.delta_softplus;
    float ru[PB], rdt[PB], rz[PB], rbc[BCE];
    auto load_block = [&](int t0) {
      const bool FULL = t0 + PB <= L;      // wave-uniform: the clamps below are scalar selects
      const TIO *u0 = ub + (int64_t)t0 * u_sl, *d0 = db + (int64_t)t0 * dt_sl, *z0 = zb + (int64_t)t0 * z_sl;
#pragma unroll
      for (int k = 0; k < PB; ++k) {
        const int kk = FULL ? k : ((t0 + k < L) ? k : L - 1 - t0);     // clamped row; masked in prepare()
        ru[k] = (float)u0[kk * u_sl + u_lo];
        rdt[k] = (float)d0[kk * dt_sl + d_lo];
        rz[k] = (float)z0[kk * z_sl + z_lo];
      }
#pragma unroll
      for (int k = 0; k < BCE; ++k) {                  // this lane's elements of the [PB][B | C] tile
        const int e = lane + 64 * k;
        const int tl = e / (2 * NPD), j = e % (2 * NPD);
        int t = t0 + tl;
        if (!FULL) t = t < L ? t : L - 1;
        const bool isC = j >= NPD;
        const int n = isC ? j - NPD : j;
        const int nc = n < N ? n : N - 1;
        const float v = isC ? Cb[t * C_sl + nc * C_sn] : Bb[t * B_sl + nc * B_sn];
        rbc[k] = n < N ? v : 0.f;
      }
    };
    auto prepare = [&](int c) {          // registers hold block c's rows -> LDS slot c & 1
      const int slot = c & 1, t0 = c * PB;
      const bool FULL = t0 + PB <= L;
#pragma unroll
      for (int k = 0; k < PB; ++k) {
        float v = rdt[k] + bias;
        if (softplus) v = softplus20(v);
        if (!FULL) v = (t0 + k < L) ? v : 0.f;       // steps past the end leave the state alone
        s_dt[slot][k][lane] = v;
        s_du[slot][k][lane] = v * ru[k];
        s_sk[slot][k][lane] = Dd * ru[k];
        const float zv = rz[k];
        s_gt[slot][k][lane] = has_z ? zv * sigmoidf_(zv) : 1.f;
      }
#pragma unroll
      for (int k = 0; k < BCE; ++k) (&s_bc[slot][0][0])[lane + 64 * k] = rbc[k];
    };
    load_block(0);
    prepare(0);
    if (nchunks > 1) load_block(PB);
    for (int c = 0; c < nchunks; ++c) {
      __syncthreads();                   // block c is complete in LDS; the consumers are done with slot (c + 1) & 1
      if (c + 1 < nchunks) {
        prepare(c + 1);
        if (c + 2 < nchunks) load_block((c + 2) * PB);
      }
    }
    return;
  }

  // ---------------------------------------------------------------- consumers
  const int n0 = w * NS;
  const int nvalid = (N - n0) < NS ? (N - n0) : NS;
  f2 Ap[NS / 2], x[NS / 2];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int jj = j < nvalid ? j : nvalid - 1;
    const float a = p.A[(int64_t)dc * N + n0 + jj] * kLog2e;
    Ap[j / 2][j % 2] = (j < nvalid) ? a : 0.f;
    x[j / 2][j % 2] = 0.f;
  }
  TIO *ob = static_cast<TIO *>(p.out) + b * p.s.o_sb;
  const int o_lo = dc * (int)p.s.o_sd;
  const int o_sl = (int)p.s.o_sl;
  float yk[PB], skk[PB], gtk[PB];        // NW == 2, last consumer: its partial sums and the epilogue operands of
                                         // the previous block (the slot they came from is rewritten meanwhile)
  auto epilogue_prev = [&](int c) {      // NW == 2: outputs of block c (consumer 0's partial sums are in LDS now)
    TIO *o0 = ob + (int64_t)c * PB * o_sl;
#pragma unroll
    for (int k = 0; k < PB; ++k) {
      const float yv = (yk[k] + s_y0[NW == 2 ? (c & 1) : 0][NW == 2 ? k : 0][lane] + skk[k]) * gtk[k];
      if (dok && c * PB + k < L) o0[k * o_sl + o_lo] = (TIO)yv;
    }
  };
  auto block = [&](int c) {
    const int slot = c & 1, t0 = c * PB;
    const bool FULL = t0 + PB <= L;
    TIO *o0 = ob + (int64_t)t0 * o_sl;
    float4 b0 = *reinterpret_cast<const float4 *>(&s_bc[slot][0][n0]), b1 = *reinterpret_cast<const float4 *>(&s_bc[slot][0][n0 + 4]);
    float4 c0 = *reinterpret_cast<const float4 *>(&s_bc[slot][0][NPD + n0]), c1 = *reinterpret_cast<const float4 *>(&s_bc[slot][0][NPD + n0 + 4]);
    float dt = s_dt[slot][0][lane], du = s_du[slot][0][lane];
#pragma unroll
    for (int k = 0; k < PB; ++k) {
      float4 nb0 = b0, nb1 = b1, nc0 = c0, nc1 = c1;
      float ndt = 0.f, ndu = 0.f;
      if (k + 1 < PB) {
        nb0 = *reinterpret_cast<const float4 *>(&s_bc[slot][k + 1][n0]);
        nb1 = *reinterpret_cast<const float4 *>(&s_bc[slot][k + 1][n0 + 4]);
        nc0 = *reinterpret_cast<const float4 *>(&s_bc[slot][k + 1][NPD + n0]);
        nc1 = *reinterpret_cast<const float4 *>(&s_bc[slot][k + 1][NPD + n0 + 4]);
        ndt = s_dt[slot][k + 1][lane];
        ndu = s_du[slot][k + 1][lane];
      }
      if ((k == 0 || k == SUB) && p.ckpt && dok) ckpt_store(p.ckpt, ckpt_slot(b, nchunks, c, k == 0 ? 0 : 1, NW, w, Dm, d), x);
      const f2 bv[NS / 2] = {f2{b0.x, b0.y}, f2{b0.z, b0.w}, f2{b1.x, b1.y}, f2{b1.z, b1.w}};
      const f2 cv[NS / 2] = {f2{c0.x, c0.y}, f2{c0.z, c0.w}, f2{c1.x, c1.y}, f2{c1.z, c1.w}};
      f2 y = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NS / 2; ++j) {
        const f2 e = dt * Ap[j];
        f2 a;
        a.x = __builtin_amdgcn_exp2f(e.x);
        a.y = __builtin_amdgcn_exp2f(e.y);
        x[j] = a * x[j] + du * bv[j];
        y = cv[j] * x[j] + y;
      }
      const float ys = y.x + y.y;
      if constexpr (NW == 1) {
        const float yv = (ys + s_sk[slot][k][lane]) * s_gt[slot][k][lane];
        if (dok && (FULL || t0 + k < L)) o0[k * o_sl + o_lo] = (TIO)yv;
      } else {
        if (w == 0) {
          s_y0[NW == 2 ? slot : 0][NW == 2 ? k : 0][lane] = ys;
        } else {
          yk[k] = ys;
          skk[k] = s_sk[slot][k][lane];
          gtk[k] = s_gt[slot][k][lane];
        }
      }
      b0 = nb0; b1 = nb1; c0 = nc0; c1 = nc1; dt = ndt; du = ndu;
    }
  };
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();
    if constexpr (NW == 2) {
      if (w == 1 && c > 0) epilogue_prev(c - 1);
    }
    block(c);
  }
  if constexpr (NW == 2) {
    __syncthreads();                     // consumer 0's partial sums of the last block
    if (w == 1) epilogue_prev(nchunks - 1);
  }
  if (p.last_state && dok) {
    float *ls = p.last_state + ((int64_t)b * Dm + d) * N + n0;
#pragma unroll
    for (int j = 0; j < NS; ++j)
      if (j < nvalid) ls[j] = x[j / 2][j % 2];
  }
}

#ifdef CUM_AB
// EXPERIMENT, AB builds only (measured and rejected, round 3): the role split of scan_fwd_ws_kernel / scan_bwd_ws_kernel at
// d_state > 16.  Result on MI355X, E8 shape (B = 16, D = 2048, N = 64, L = 624): 0.405 ms against 0.272 ms for
// scan_fwd_lds_kernel.  At N = 64 the grid already puts 16 recurrence waves on every CU (4 per SIMD): the SIMDs' issue
// slots are the bound, and moving the per-(t, d) work to helper waves does not remove it from them -- it only adds two
// waves per workgroup that land 3 : 2 on the SIMDs (10 waves over 4 SIMDs), and the barrier waits for the fuller ones.
// The specialised form pays only where SIMDs would otherwise idle (d_state <= 16 at small batch).
// d_state > 16 (E6 / E8: 64), the kernel north_star names.  scan_fwd_lds_kernel above makes every wave do three jobs per
// 16-step chunk -- prepare two of the chunk's rows (softplus, delta u: phase A), walk its 8 states through the 16 steps
// (phase B), sum the 8 per-wave partial results of two rows, gate and store them (phase C) -- with two workgroup barriers;
// it measured 0.56 of the update loop's issue roof (bench.py scan rows).  Here the roles of scan_fwd_ws_kernel /
// scan_bwd_ws_kernel are split over waves: NW CONSUMER waves do nothing but the recurrence (per step 6 LDS reads, 24
// packed / transcendental ops, one partial-sum write), a LOADER wave owns the way in (rows of u / delta / z and the B_t / C_t
// tile one unit ahead in registers; softplus, delta u, D u, silu(z) once per (t, d) into LDS) and a FINISHER wave the way
// out, one unit behind (sum of the NW partial sums + D u, gate, store).  Unit = the 8 steps between two saved states; one
// barrier per unit.  In interval i the loader writes operand slot (i + 1) & 1 / finisher-operand slot (i + 1) % 3, the
// consumers read slot i & 1 and write partial-sum slot i & 1, the finisher reads partial sums (i - 1) & 1 and its
// operands (i - 1) % 3.
template <int NW, typename TIO>
__global__ __launch_bounds__((NW + 2) * 64) void scan_fwd_ws3_kernel(const ScanParams p) {
  constexpr int NP = NW * NS;                    // padded state count
  constexpr int BCE = SUB * 2 * NP / 64;         // B / C tile elements per loader lane and unit
  __shared__ __attribute__((aligned(8))) float2 s_op[2][SUB][64];     // {dt, dt * u}; 0 for masked steps / lanes
  __shared__ __attribute__((aligned(8))) float2 s_fin[3][SUB][64];    // {D u, silu gate}
  __shared__ float s_y[2][NW][SUB][64];                               // per-wave sum_n C x
  __shared__ __attribute__((aligned(16))) float s_bc[2][SUB][2 * NP];

  const int lane = threadIdx.x & 63;
  const int w = uniform(threadIdx.x >> 6);
  const int b = blockIdx.y;
  const int d = blockIdx.x * 64 + lane;
  const int N = p.s.dstate, L = p.s.len, Dm = p.s.dim;
  const bool dok = d < Dm;
  const int dc = dok ? d : Dm - 1;
  const int nu = (L + SUB - 1) / SUB;            // units
  const int nchunks = p.nchunks;
  const bool has_z = p.z != nullptr;

  if (w == NW) {
    // ------------------------------------------------------------------------------------------------ loader
    const float Dd = p.D ? p.D[dc] : 0.f;
    const float bias = p.bias ? p.bias[dc] : 0.f;
    const TIO *ub = static_cast<const TIO *>(p.u) + b * p.s.u_sb;
    const TIO *db = static_cast<const TIO *>(p.delta) + b * p.s.dt_sb;
    const TIO *zb = has_z ? static_cast<const TIO *>(p.z) + b * p.s.z_sb : ub;
    const int u_lo = dc * (int)p.s.u_sd, d_lo = dc * (int)p.s.dt_sd, z_lo = has_z ? dc * (int)p.s.z_sd : u_lo;
    const int u_sl = (int)p.s.u_sl, dt_sl = (int)p.s.dt_sl, z_sl = has_z ? (int)p.s.z_sl : (int)p.s.u_sl;
    const float *Bb = p.Bm + b * p.s.B_sb, *Cb = p.Cm + b * p.s.C_sb;
    const int B_sl = (int)p.s.B_sl, C_sl = (int)p.s.C_sl, B_sn = (int)p.s.B_sn, C_sn = (int)p.s.C_sn;
    const int softplus = p.s.delta_softplus;
    float ru[SUB], rdt[SUB], rz[SUB], rbc[BCE];
    auto load_rows = [&](int h) {
      const int t0 = h * SUB;
      const bool full = t0 + SUB <= L;            // wave-uniform: the clamps below are scalar selects
      const TIO *u0 = ub + (int64_t)t0 * u_sl, *d0 = db + (int64_t)t0 * dt_sl, *z0 = zb + (int64_t)t0 * z_sl;
#pragma unroll
      for (int k = 0; k < SUB; ++k) {
        const int kk = full ? k : ((t0 + k < L) ? k : L - 1 - t0);
        ru[k] = (float)u0[kk * u_sl + u_lo];
        rdt[k] = (float)d0[kk * dt_sl + d_lo];
        rz[k] = (float)z0[kk * z_sl + z_lo];
      }
#pragma unroll
      for (int k = 0; k < BCE; ++k) {
        const int e = lane + 64 * k;
        const int tl = e / (2 * NP), j = e % (2 * NP);
        int t = t0 + tl;
        if (!full) t = t < L ? t : L - 1;
        const bool isC = j >= NP;
        const int n = isC ? j - NP : j;
        const int nc = n < N ? n : N - 1;
        const float v = isC ? Cb[t * C_sl + nc * C_sn] : Bb[t * B_sl + nc * B_sn];
        rbc[k] = n < N ? v : 0.f;
      }
    };
    auto prepare = [&](int h, int i) {            // the rows in registers are unit h's -> slots of interval i
      const int t0 = h * SUB;
#pragma unroll
      for (int k = 0; k < SUB; ++k) {
        const bool ok = dok && t0 + k < L;
        float v = rdt[k] + bias;
        if (softplus) v = softplus20(v);
        v = ok ? v : 0.f;                         // steps past the end leave the state alone (a = 1, b = 0)
        const float zv = rz[k];
        s_op[i & 1][k][lane] = make_float2(v, v * ru[k]);
        s_fin[i % 3][k][lane] = make_float2(Dd * ru[k], has_z ? zv * sigmoidf_(zv) : 1.f);
      }
#pragma unroll
      for (int k = 0; k < BCE; ++k) (&s_bc[i & 1][0][0])[lane + 64 * k] = rbc[k];
    };
    load_rows(0);
    prepare(0, 0);
    if (nu > 1) load_rows(1);
    for (int i = 0; i < nu + 1; ++i) {
      __syncthreads();
      if (i + 1 < nu) {
        prepare(i + 1, i + 1);
        if (i + 2 < nu) load_rows(i + 2);
      }
    }
    return;
  }

  if (w == NW + 1) {
    // ------------------------------------------------------------------------------------------------ finisher
    TIO *ob = static_cast<TIO *>(p.out) + b * p.s.o_sb;
    const int o_lo = dc * (int)p.s.o_sd, o_sl = (int)p.s.o_sl;
    auto finish = [&](int h) {                    // unit h was walked in interval h
      const int t0 = h * SUB, ps = h & 1, fs = h % 3;
      TIO *o0 = ob + (int64_t)t0 * o_sl;
#pragma unroll
      for (int k = 0; k < SUB; ++k) {
        const float2 f = s_fin[fs][k][lane];
        float y = f.x;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) y += s_y[ps][ww][k][lane];
        if (dok && t0 + k < L) o0[k * o_sl + o_lo] = (TIO)(y * f.y);
      }
    };
    for (int i = 0; i < nu + 1; ++i) {
      __syncthreads();
      if (i > 0) finish(i - 1);
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- consumers
  const int n0 = w * NS;
  const int nvalid = (N - n0) < NS ? (N - n0) : NS;
  f2 Ap[NS / 2], x[NS / 2];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int jj = j < nvalid ? j : nvalid - 1;
    const float a = p.A[(int64_t)dc * N + n0 + jj] * kLog2e;
    Ap[j / 2][j % 2] = (j < nvalid) ? a : 0.f;
    x[j / 2][j % 2] = 0.f;
  }
  for (int i = 0; i < nu + 1; ++i) {
    __syncthreads();
    if (i >= nu) break;                           // (the drain interval belongs to the finisher)
    const int slot = i & 1;
    if (p.ckpt && dok) ckpt_store(p.ckpt, ckpt_slot(b, nchunks, i >> 1, i & 1, NW, w, Dm, d), x);
    float4 b0 = *reinterpret_cast<const float4 *>(&s_bc[slot][0][n0]), b1 = *reinterpret_cast<const float4 *>(&s_bc[slot][0][n0 + 4]);
    float4 c0 = *reinterpret_cast<const float4 *>(&s_bc[slot][0][NP + n0]), c1 = *reinterpret_cast<const float4 *>(&s_bc[slot][0][NP + n0 + 4]);
    float2 op = s_op[slot][0][lane];
#pragma unroll
    for (int k = 0; k < SUB; ++k) {
      float4 nb0 = b0, nb1 = b1, nc0 = c0, nc1 = c1;
      float2 nop = op;
      if (k + 1 < SUB) {
        nb0 = *reinterpret_cast<const float4 *>(&s_bc[slot][k + 1][n0]);
        nb1 = *reinterpret_cast<const float4 *>(&s_bc[slot][k + 1][n0 + 4]);
        nc0 = *reinterpret_cast<const float4 *>(&s_bc[slot][k + 1][NP + n0]);
        nc1 = *reinterpret_cast<const float4 *>(&s_bc[slot][k + 1][NP + n0 + 4]);
        nop = s_op[slot][k + 1][lane];
      }
      const f2 bv[NS / 2] = {f2{b0.x, b0.y}, f2{b0.z, b0.w}, f2{b1.x, b1.y}, f2{b1.z, b1.w}};
      const f2 cv[NS / 2] = {f2{c0.x, c0.y}, f2{c0.z, c0.w}, f2{c1.x, c1.y}, f2{c1.z, c1.w}};
      const float dt = op.x, du = op.y;
      f2 y = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NS / 2; ++j) {
        const f2 e = dt * Ap[j];
        f2 a;
        a.x = __builtin_amdgcn_exp2f(e.x);
        a.y = __builtin_amdgcn_exp2f(e.y);
        x[j] = a * x[j] + du * bv[j];
        y = cv[j] * x[j] + y;
      }
      s_y[slot][w][k][lane] = y.x + y.y;
      __builtin_amdgcn_sched_barrier(0);
      b0 = nb0; b1 = nb1; c0 = nc0; c1 = nc1; op = nop;
    }
  }
  if (p.last_state && dok) {
    float *ls = p.last_state + ((int64_t)b * Dm + d) * N + n0;
#pragma unroll
    for (int j = 0; j < NS; ++j)
      if (j < nvalid) ls[j] = x[j / 2][j % 2];
  }
}

#endif  // CUM_AB

// selective_state_update: one thread per (stream b, channel d); state row of N floats.
__global__ void state_update_kernel(int batch, int dim, int N, float *__restrict__ state, const float *__restrict__ x,
                                    const float *__restrict__ dt, const float *__restrict__ A,
                                    const float *__restrict__ Bv, int64_t B_sb, const float *__restrict__ Cv,
                                    int64_t C_sb, const float *__restrict__ D, const float *__restrict__ z,
                                    const float *__restrict__ dt_bias, int softplus, float *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)batch * dim) return;
  const int b = i / dim, d = i % dim;
  float dtv = dt[i] + (dt_bias ? dt_bias[d] : 0.f);
  if (softplus) dtv = softplus20(dtv);
  const float xv = x[i];
  const float du = dtv * xv;
  float *st = state + i * N;
  const float *Ar = A + (int64_t)d * N;
  const float *Br = Bv + b * B_sb, *Cr = Cv + b * C_sb;
  float y = 0.f;
  for (int n = 0; n < N; ++n) {
    const float a = __builtin_amdgcn_exp2f(dtv * (Ar[n] * kLog2e));
    const float s = fmaf(a, st[n], du * Br[n]);
    st[n] = s;
    y = fmaf(Cr[n], s, y);
  }
  if (D) y = fmaf(D[d], xv, y);
  if (z) {
    const float zv = z[i];
    y *= zv * sigmoidf_(zv);
  }
  out[i] = y;
}

template <int NW, typename TIO>
static int launch_fwd_io(const ScanParams &p, hipStream_t st) {
  dim3 grid((p.s.dim + 63) / 64, p.s.batch), block(NW * 64);
  if constexpr (NW <= 2) {
    // d_state <= 16.  Measured on MI355X (D = 2048, L = 2499, f32 I/O; tools/bench_scan.py): the wave-specialised form
    // (producer wave + NW consumer waves per workgroup) wins while the grid cannot give every SIMD a wave of its own --
    // up to 1024 workgroups at d_state <= 8, 512 at d_state <= 16; beyond that one wave per workgroup holding all states.
    // AB build: CUM_SCAN_SMALL=0 the NW-wave kernel below, 2 always the one-wave kernel; CUM_SCAN_WS_MAX the threshold.
    const int v = (int)cum_knob("CUM_SCAN_SMALL", 1);
    const int64_t ws_max = cum_knob("CUM_SCAN_WS_MAX", NW == 1 ? 1024 : 512);
    if (v == 2 || (v == 1 && (int64_t)grid.x * grid.y > ws_max)) {
      // more waves per SIMD (1024 SIMDs) than the whole-chunk form fits (3 at d_state <= 8, 2 at <= 16): the half-chunk
      // form, which fits one more
      const int64_t half_min = cum_knob("CUM_SCAN_SMALL_HALF_MIN", NW == 1 ? 3072 : 2048);
      if ((int64_t)grid.x * grid.y > half_min)
        hipLaunchKernelGGL((scan_fwd_small_kernel<NW, TIO, SUB>), grid, dim3(64), 0, st, p);
      else
        hipLaunchKernelGGL((scan_fwd_small_kernel<NW, TIO, TB>), grid, dim3(64), 0, st, p);
      CUM_CHECK_LAUNCH();
      return CUM_OK;
    }
#ifdef CUM_AB
    if (v == 1)
#endif
    {
      hipLaunchKernelGGL((scan_fwd_ws_kernel<NW, TIO>), grid, dim3((NW + 1) * 64), 0, st, p);
      CUM_CHECK_LAUNCH();
      return CUM_OK;
    }
  }
#ifndef CUM_AB
  if constexpr (NW > 2)     // (the NW-wave kernels are not even instantiated for d_state <= 16 outside an AB build)
#endif
  {
#ifdef CUM_AB   // CUM_SCAN_FWD_LDS=0: B_t / C_t through the constant address space (s_load) instead of the LDS tile
  if (cum_knob("CUM_SCAN_FWD_LDS", 1) == 0) {
    if (p.s.B_sn == 1 && p.s.C_sn == 1 && p.s.dstate == NS * NW)
      hipLaunchKernelGGL((scan_fwd_kernel<NW, true, TIO>), grid, block, 0, st, p);
    else
      hipLaunchKernelGGL((scan_fwd_kernel<NW, false, TIO>), grid, block, 0, st, p);
    CUM_CHECK_LAUNCH();
    return CUM_OK;
  }
#endif
#ifdef CUM_AB   // CUM_SCAN_FWD_WS3=1: loader / consumer / finisher waves (scan_fwd_ws3_kernel: measured 49 % slower at N = 64)
    if (cum_knob("CUM_SCAN_FWD_WS3", 0) == 1) {
      hipLaunchKernelGGL((scan_fwd_ws3_kernel<NW, TIO>), grid, dim3((NW + 2) * 64), 0, st, p);
      CUM_CHECK_LAUNCH();
      return CUM_OK;
    }
#endif
    hipLaunchKernelGGL((scan_fwd_lds_kernel<NW, TIO>), grid, block, 0, st, p);
    CUM_CHECK_LAUNCH();
  }
  return CUM_OK;
}

template <int NW>
static int launch_fwd(const ScanParams &p, hipStream_t st) {
  if (p.s.io_dtype == CUM_BF16) return launch_fwd_io<NW, __bf16>(p, st);
  if (p.s.io_dtype == CUM_F16) return launch_fwd_io<NW, f16>(p, st);
  return launch_fwd_io<NW, float>(p, st);
}

int scan_check_shape(const cum_scan_shape *s) {
  CUM_REQUIRE(s != nullptr, "scan: null shape");
  CUM_REQUIRE(s->batch >= 0 && s->dim >= 1 && s->len >= 0, "scan: bad batch/dim/len");
  CUM_REQUIRE(s->dstate >= 1 && s->dstate <= 64, "scan: d_state must be in [1, 64]");
  CUM_REQUIRE(dtype_ok(s->io_dtype), "scan: io_dtype must be CUM_F32, CUM_BF16 or CUM_F16");
  const int64_t lim = 2147483647LL;
  const int64_t Lm = s->len > 0 ? s->len - 1 : 0;
  CUM_REQUIRE(s->u_sl >= 0 && s->dt_sl >= 0 && s->z_sl >= 0 && s->o_sl >= 0 && s->B_sl >= 0 && s->C_sl >= 0 &&
                  s->B_sn >= 0 && s->C_sn >= 0,
              "scan: negative strides are not supported");
  CUM_REQUIRE(Lm * s->u_sl < lim && Lm * s->dt_sl < lim && Lm * s->z_sl < lim && Lm * s->o_sl < lim &&
                  Lm * s->B_sl < lim && Lm * s->C_sl < lim && (int64_t)s->dstate * s->B_sn < lim &&
                  (int64_t)s->dstate * s->C_sn < lim,
              "scan: per-batch time offsets must fit in 31 bits");
  return CUM_OK;
}

}  // namespace cum

using namespace cum;

extern "C" int cum_scan_chunk(void) { return TB; }

extern "C" int64_t cum_scan_ckpt_elems(int32_t batch, int32_t dim, int32_t dstate, int32_t len) {
  const int64_t nchunks = (len + TB - 1) / TB;
  const int64_t nw = (dstate + NS - 1) / NS;
  return 2 * (int64_t)batch * nchunks * nw * dim * NS;   // the state entering each 8-step half of every chunk
}

extern "C" int64_t cum_scan_fwd_workspace_elems(int32_t batch, int32_t dim, int32_t dstate, int32_t len) {
  if (batch <= 0 || dim <= 0 || dstate <= 0 || dstate > 64 || len <= 0) return 0;
  int nseg, sc;
  scan_seg_plan(batch, dim, dstate, len, &nseg, &sc);
  return nseg > 1 ? scan_seg_carry_elems(batch, dim, dstate, nseg) : 0;
}

extern "C" int cum_selective_scan_fwd(const cum_scan_shape *s, const void *u, const void *delta, const float *A,
                                      const float *Bm, const float *Cm, const float *D, const void *z,
                                      const float *delta_bias, void *out, float *last_state, float *ckpt,
                                      void *stream) {
  return cum_selective_scan_fwd_ws(s, u, delta, A, Bm, Cm, D, z, delta_bias, out, nullptr, last_state, ckpt, nullptr, stream);
}

// 1 if the forward this shape takes (with / without a workspace offered) can keep y before the gate for the backward:
// the sequential kernel for d_state > 16 (the E6 / E8 bottleneck); the backward of the other shapes rebuilds it.
extern "C" int32_t cum_scan_fwd_keeps_y(int32_t batch, int32_t dim, int32_t dstate, int32_t len, int32_t with_workspace) {
  if (batch <= 0 || dim <= 0 || len <= 0 || dstate <= 2 * NS) return 0;
#ifdef CUM_AB   // (the A/B forward variants do not write it)
  if (cum_knob("CUM_SCAN_FWD_LDS", 1) == 0 || cum_knob("CUM_SCAN_FWD_WS3", 0) == 1) return 0;
#endif
  if (with_workspace) {
    int nseg = 1, segc = 0;
    scan_seg_plan(batch, dim, dstate, len, &nseg, &segc);
    if (nseg > 1) return 0;
  }
  return 1;
}

extern "C" int cum_selective_scan_fwd_ws(const cum_scan_shape *s, const void *u, const void *delta, const float *A,
                                         const float *Bm, const float *Cm, const float *D, const void *z,
                                         const float *delta_bias, void *out, void *y_pre, float *last_state, float *ckpt,
                                         float *workspace, void *stream) {
  if (int rc = scan_check_shape(s)) return rc;
  if (s->batch == 0) return CUM_OK;
  hipStream_t st = (hipStream_t)stream;
  if (s->len == 0) {  // empty sequences carry null data pointers
    if (last_state)
      (void)hipMemsetAsync(last_state, 0, sizeof(float) * (size_t)s->batch * s->dim * s->dstate, st);
    return CUM_OK;
  }
  CUM_REQUIRE(u && delta && A && Bm && Cm && out, "scan_fwd: null tensor");
  ScanParams p{};
  p.s = *s;
  p.u = u; p.delta = delta; p.A = A; p.Bm = Bm; p.Cm = Cm; p.D = D; p.z = z; p.bias = delta_bias;
  p.out = out; p.last_state = last_state; p.ckpt = ckpt;
  p.ypre = y_pre;
  CUM_REQUIRE(!y_pre || cum_scan_fwd_keeps_y(s->batch, s->dim, s->dstate, s->len, workspace != nullptr),
              "scan_fwd: y_pre is kept only where cum_scan_fwd_keeps_y says so");
  p.nchunks = (s->len + TB - 1) / TB;
  p.ngroups = (s->dim + 63) / 64;
  if (workspace) {          // the caller offers the segmented path its workspace: taken when the plan splits the sequence
    scan_seg_plan(s->batch, s->dim, s->dstate, s->len, &p.nseg, &p.seg_chunks);
    if (p.nseg > 1) {
      p.carry = workspace;
      return launch_fwd_segmented(p, st);
    }
  }
  switch ((s->dstate + NS - 1) / NS) {
    case 1: return launch_fwd<1>(p, st);
    case 2: return launch_fwd<2>(p, st);
    case 3: return launch_fwd<3>(p, st);
    case 4: return launch_fwd<4>(p, st);
    case 5: return launch_fwd<5>(p, st);
    case 6: return launch_fwd<6>(p, st);
    case 7: return launch_fwd<7>(p, st);
    default: return launch_fwd<8>(p, st);
  }
}

extern "C" int cum_selective_state_update(int32_t batch, int32_t dim, int32_t dstate, float *state, const float *x,
                                          const float *dt, const float *A, const float *Bv, int64_t B_sb,
                                          const float *Cv, int64_t C_sb, const float *D, const float *z,
                                          const float *dt_bias, int32_t dt_softplus, float *out, void *stream) {
  CUM_REQUIRE(batch >= 0 && dim >= 1 && dstate >= 1, "state_update: bad sizes");
  CUM_REQUIRE(state && x && dt && A && Bv && Cv && out, "state_update: null tensor");
  if (batch == 0) return CUM_OK;
  const int64_t total = (int64_t)batch * dim;
  hipLaunchKernelGGL(state_update_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, (hipStream_t)stream, batch,
                     dim, dstate, state, x, dt, A, Bv, B_sb, Cv, C_sb, D, z, dt_bias, dt_softplus, out);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}
