// Time-parallel (segmented, chunk-carry) selective scan forward for grids that do not fill the chip.
//
// Same op as scan_fwd.hip (selective_scan_cuda.fwd of mamba-ssm 1.2.2, reached from the reference through create_block ->
// Mamba.forward, src/network/CleanUMamba.py:172-189, 289-290; semantics SURVEY.md Appendix A.2).  The kernels there walk
// time sequentially with lane <-> channel and wave <-> 8 states: batch * ceil(dim / 64) * ceil(d_state / 8) waves in
// all.  File denoising at batch 1 (src/examples/denoise.py: 256 waves at E8 on 1 024 SIMDs), the 442K model and the
// pruned checkpoints (d_inner 8...136, d_state 8...16) leave most of the chip idle and the launch takes one wave's walk of
// all L steps.  The recurrence x_t = a_t x_{t-1} + b_t is linear, so time splits into S segments whose effects compose
// with the associative operator (a, b) o (a', b') = (a' a, a' b + b'):
//
//   pass 1  (scan_seg_kernel<.., 1>, grid x S): every segment is walked from a ZERO state: x_end^0[d, n], and
//           sum_t delta'_t[d].  The segment's decay needs no product over time: a_t = exp(delta'_t A), so
//           prod_t a_t = exp(A sum_t delta'_t).
//   carry   (prologue of pass 2): per (b, d, n), sequentially over the earlier segments: X_0 = 0,
//           X_{s+1} = exp2(A log2e * sum delta'_s) X_s + x_end^0_s  -- the true state entering every segment.
//   pass 2  (scan_seg_kernel<.., 2>, grid x S): every segment re-walked from X_s with outputs, the z gate, the saved
//           states of the backward (same checkpoint layout: scan_bwd*.hip are unchanged) and last_state.
//
// Twice the state updates (pass 1 carries no C_t, no y, no gate: ~0.6 of a pass-2 step), S times the waves: chosen by
// scan_seg_plan() when the sequential grid brings fewer than two waves per SIMD.  Bit-reproducible (no atomics); against
// the sequential kernels the segment decay exp2(A' sum delta') replaces a product of per-step exp2 -- equal up to f32
// rounding (parity tests: the same goldens and odd shapes through both paths).
#include "scan_common.h"

namespace cum {

template <int NW, typename TIO, int PASS>
__global__ __launch_bounds__(NW * 64) void scan_seg_kernel(const ScanParams p) {
  constexpr int K = (TB + NW - 1) / NW;
  constexpr int NT = NW * 64;
  constexpr int NP = NW * NS;                    // padded state count
  constexpr int BCK = (TB * NP + NT - 1) / NT;   // B (and C) elements per thread per chunk
  constexpr bool OUT = PASS == 2;
  __shared__ float s_dt[TB][64];
  __shared__ float s_du[TB][64];
  __shared__ float s_y[OUT ? NW : 1][OUT ? TB : 1][64];
  __shared__ __attribute__((aligned(16))) float s_B[TB][NP];
  __shared__ __attribute__((aligned(16))) float s_C[OUT ? TB : 1][NP];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = uniform(tid >> 6);
  const int b = blockIdx.y, seg = blockIdx.z;
  const int d = blockIdx.x * 64 + lane;
  const int N = p.s.dstate, L = p.s.len, Dm = p.s.dim;
  const bool dok = d < Dm;
  const int dc = dok ? d : Dm - 1;
  const int n0 = w * NS;
  const int nvalid = (N - n0) < NS ? (N - n0) : NS;
  const int nchunks = p.nchunks, nseg = p.nseg;
  const int c_lo = seg * p.seg_chunks;
  const int c_hi = (c_lo + p.seg_chunks) < nchunks ? (c_lo + p.seg_chunks) : nchunks;
  float *xsum = p.carry + (int64_t)p.s.batch * nseg * NW * Dm * NS;      // [(b, seg, d)]

  f2 Ap[NS / 2], x[NS / 2];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int jj = j < nvalid ? j : nvalid - 1;
    const float a = p.A[(int64_t)dc * N + n0 + jj] * kLog2e;
    Ap[j / 2][j % 2] = (j < nvalid) ? a : 0.f;
    x[j / 2][j % 2] = 0.f;
  }
  if constexpr (OUT) {
    // The true entering state: compose the (decay, end state) pairs of all earlier segments, in order.  seg - 1 dependent
    // exp2 + fma per state on loads that do not depend on each other (a separate carry launch between the passes cost
    // more than these few L2 reads: its own prologue, a launch boundary, and a second trip of the states through memory).
#pragma unroll 4
    for (int sp = 0; sp < seg; ++sp) {
      f2 e[NS / 2];
      ckpt_load(p.carry, carry_slot(b, nseg, sp, NW, w, Dm, dc), e);
      const float ds = xsum[((int64_t)b * nseg + sp) * Dm + dc];
#pragma unroll
      for (int j = 0; j < NS / 2; ++j) {
        const f2 t = ds * Ap[j];
        f2 a;
        a.x = __builtin_amdgcn_exp2f(t.x);
        a.y = __builtin_amdgcn_exp2f(t.y);
        x[j] = a * x[j] + e[j];
      }
    }
  }
  const float Dd = p.D ? p.D[dc] : 0.f;
  const float bias = p.bias ? p.bias[dc] : 0.f;
  const TIO *up = static_cast<const TIO *>(p.u) + b * p.s.u_sb + dc * p.s.u_sd;
  const TIO *dtp = static_cast<const TIO *>(p.delta) + b * p.s.dt_sb + dc * p.s.dt_sd;
  const bool has_z = OUT && p.z != nullptr;
  const TIO *zp = has_z ? static_cast<const TIO *>(p.z) + b * p.s.z_sb + dc * p.s.z_sd : up;
  TIO *op = static_cast<TIO *>(p.out) + b * p.s.o_sb + dc * p.s.o_sd;
  const float *Bb = p.Bm + b * p.s.B_sb, *Cb = p.Cm + b * p.s.C_sb;
  const int u_sl = (int)p.s.u_sl, dt_sl = (int)p.s.dt_sl, z_sl = has_z ? (int)p.s.z_sl : (int)p.s.u_sl;
  const int o_sl = (int)p.s.o_sl;
  const int B_sl = (int)p.s.B_sl, C_sl = (int)p.s.C_sl, B_sn = (int)p.s.B_sn, C_sn = (int)p.s.C_sn;
  const int softplus = p.s.delta_softplus;

  float ru[K], rdt[K], rz[OUT ? K : 1], rb[BCK], rc[OUT ? BCK : 1];
  auto load_rows = [&](int t0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      int t = t0 + w + k * NW;
      t = t < L ? t : L - 1;
      ru[k] = (float)up[t * u_sl];
      rdt[k] = (float)dtp[t * dt_sl];
      if constexpr (OUT) rz[k] = (float)zp[t * z_sl];
    }
#pragma unroll
    for (int k = 0; k < BCK; ++k) {
      const int e = tid + k * NT;
      const int tl = e / NP, n = e % NP;
      int t = t0 + tl;
      t = t < L ? t : L - 1;
      const int nc = n < N ? n : N - 1;
      const float bvv = Bb[t * B_sl + nc * B_sn];
      rb[k] = n < N ? bvv : 0.f;
      if constexpr (OUT) {
        const float cvv = Cb[t * C_sl + nc * C_sn];
        rc[k] = n < N ? cvv : 0.f;
      }
    }
  };
  load_rows(c_lo * TB);

  float dsum = 0.f;                              // pass 1: sum of delta' over the segment (every wave holds the same)
  for (int c = c_lo; c < c_hi; ++c) {
    const int t0 = c * TB;
    float eu[OUT ? K : 1], ez[OUT ? K : 1];
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
      if constexpr (OUT) {
        eu[k] = ru[k];
        ez[k] = rz[k];
      }
    }
#pragma unroll
    for (int k = 0; k < BCK; ++k) {
      const int e = tid + k * NT;
      if (e < TB * NP) {
        (&s_B[0][0])[e] = rb[k];
        if constexpr (OUT) (&s_C[0][0])[e] = rc[k];
      }
    }
    if (c + 1 < c_hi) load_rows(t0 + TB);
    if (OUT && p.ckpt && dok) ckpt_store(p.ckpt, ckpt_slot(b, nchunks, c, 0, NW, w, Dm, d), x);
    __syncthreads();
    float4 b0 = *reinterpret_cast<const float4 *>(&s_B[0][n0]), b1 = *reinterpret_cast<const float4 *>(&s_B[0][n0 + 4]);
    float4 c0 = b0, c1 = b1;
    if constexpr (OUT) {
      c0 = *reinterpret_cast<const float4 *>(&s_C[0][n0]);
      c1 = *reinterpret_cast<const float4 *>(&s_C[0][n0 + 4]);
    }
    float dt = s_dt[0][lane], du = s_du[0][lane];
#pragma unroll
    for (int tl = 0; tl < TB; ++tl) {
      float4 nb0 = b0, nb1 = b1, nc0 = c0, nc1 = c1;
      float ndt = 0.f, ndu = 0.f;
      if (tl + 1 < TB) {
        nb0 = *reinterpret_cast<const float4 *>(&s_B[tl + 1][n0]);
        nb1 = *reinterpret_cast<const float4 *>(&s_B[tl + 1][n0 + 4]);
        if constexpr (OUT) {
          nc0 = *reinterpret_cast<const float4 *>(&s_C[tl + 1][n0]);
          nc1 = *reinterpret_cast<const float4 *>(&s_C[tl + 1][n0 + 4]);
        }
        ndt = s_dt[tl + 1][lane];
        ndu = s_du[tl + 1][lane];
      }
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
        if constexpr (OUT) y = cv[j] * x[j] + y;
      }
      if constexpr (OUT) {
        s_y[w][tl][lane] = y.x + y.y;
        if (tl == SUB - 1 && p.ckpt && dok)   // state entering the second half of the chunk
          ckpt_store(p.ckpt, ckpt_slot(b, nchunks, c, 1, NW, w, Dm, d), x);
      } else {
        dsum += dt;
      }
      __builtin_amdgcn_sched_barrier(0);
      b0 = nb0; b1 = nb1; c0 = nc0; c1 = nc1; dt = ndt; du = ndu;
    }
    __syncthreads();
    if constexpr (OUT) {
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
    }
  }
  if constexpr (!OUT) {
    if (dok) {
      ckpt_store(p.carry, carry_slot(b, nseg, seg, NW, w, Dm, d), x);
      if (w == 0) xsum[((int64_t)b * nseg + seg) * Dm + d] = dsum;
    }
  } else {
    if (p.last_state && dok && seg == nseg - 1) {
      float *ls = p.last_state + ((int64_t)b * Dm + d) * N + n0;
#pragma unroll
      for (int j = 0; j < NS; ++j)
        if (j < nvalid) ls[j] = x[j / 2][j % 2];
    }
  }
}

// Segment plan.  waves = what the sequential kernels launch.  Target (segment-count sweep on MI355X, same box,
// gpurun_out/r04_scan_tp_sweep.txt): d_state > 16 -- ONE 8-wave workgroup per CU (2 048 waves: batch-1 E8, 32 groups ->
// 8 segments = 40.9 us against 45.8 us at 13 and 50.2 us at 20; the kernel is issue-bound from two waves per SIMD, more
// co-resident workgroups only lengthen every chunk); d_state <= 16 -- small workgroups, about four waves per SIMD
// (4 096: the 442K model 25.8 us at 20 segments against 35.3 us at 8).  At least two 16-step chunks per segment;
// segmented only when that gives >= 3 segments.
void scan_seg_plan(int batch, int dim, int dstate, int len, int *nseg, int *seg_chunks) {
  const int64_t NW = (dstate + NS - 1) / NS, groups = (dim + 63) / 64;
  const int64_t waves = (int64_t)batch * groups * NW;
  const int nchunks = (len + TB - 1) / TB;
  *nseg = 1;
  *seg_chunks = nchunks;
  const int64_t force = cum_knob("CUM_SCAN_SEGMENTS", -1);       // AB build: 0 = never, n > 1 = that many segments
  if (force == 0 || waves <= 0 || nchunks < 6) return;
  // Measured on MI355X (bench.py scan rows, same box): at d_state > 16 the sequential kernel runs 8 waves per 64 channels and
  // the segmented form pays from < 1 wave per SIMD (batch <= 3 at D = 2048); at d_state <= 16 the sequential kernels are
  // the wave-specialised ones (producer + consumer waves: already two waves per SIMD at 512 "waves" here) and the
  // segmented form, which uses the generic chunk structure, only wins while the grid is <= 256 waves (B = 16, D = 2048,
  // N = 16: 0.56 ms sequential against 0.64 ms segmented -- not taken).
  if (force < 0 && waves >= (NW > 2 ? 1024 : 257)) return;
  const int64_t target = NW > 2 ? 2048 : 4096;
  int64_t want = force > 1 ? force : (target + waves - 1) / waves;
  int sc = (int)((nchunks + want - 1) / want);
  if (sc < 2) sc = 2;
  const int S = (nchunks + sc - 1) / sc;
  if (S < 3) return;
  *nseg = S;
  *seg_chunks = sc;
}

// The backward's plan (scan_bwd_small.hip, d_state <= 16): its sequential grid is batch * ceil(dim / 64) workgroups of
// NW + 2 waves that each walk all halves with one barrier per half; segmented while that grid leaves most of the chip idle,
// about 2 560 workgroups in all, >= 2 chunks per segment, >= 3 segments.  Twice the reverse-walk work (pass 1 is the bare
// recurrence).  Measured on MI355X (tools/bench_scan_tp_bwd.py, same box, graph-replay timing): 16-32 workgroups 5.7-6.2 x
// the sequential kernel, 64 3.6 x, 128 1.5-2.0 x, 256 0.9-1.1 x (three workgroups per CU by LDS: the chip is as busy as the
// wave-specialised design gets) -> taken up to 192.
void scan_seg_plan_bwd(int batch, int dim, int dstate, int len, int *nseg, int *seg_chunks) {
  const int64_t groups = (int64_t)batch * ((dim + 63) / 64);
  const int nchunks = (len + TB - 1) / TB;
  *nseg = 1;
  *seg_chunks = nchunks;
  const int64_t force = cum_knob("CUM_SCAN_BWD_SEGMENTS", -1);   // AB build: 0 = never, n > 1 = that many segments
  if (dstate > 2 * NS || force == 0 || groups <= 0 || nchunks < 6) return;
  if (force < 0 && groups > 192) return;
  int64_t want = force > 1 ? force : (2560 + groups - 1) / groups;
  int sc = (int)((nchunks + want - 1) / want);
  if (sc < 2) sc = 2;
  const int S = (nchunks + sc - 1) / sc;
  if (S < 3) return;
  *nseg = S;
  *seg_chunks = sc;
}

int64_t scan_seg_carry_elems(int batch, int dim, int dstate, int nseg) {
  const int64_t NW = (dstate + NS - 1) / NS;
  return (int64_t)batch * nseg * NW * dim * NS + (int64_t)batch * nseg * dim;
}

template <int NW, typename TIO>
static int launch_seg_io(const ScanParams &p, hipStream_t st) {
  dim3 grid(p.ngroups, p.s.batch, p.nseg), grid1(p.ngroups, p.s.batch, p.nseg - 1), block(NW * 64);
  hipLaunchKernelGGL((scan_seg_kernel<NW, TIO, 1>), grid1, block, 0, st, p);
  CUM_CHECK_LAUNCH();
  hipLaunchKernelGGL((scan_seg_kernel<NW, TIO, 2>), grid, block, 0, st, p);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

template <int NW>
static int launch_seg_nw(const ScanParams &p, hipStream_t st) {
  if (p.s.io_dtype == CUM_BF16) return launch_seg_io<NW, __bf16>(p, st);
  if (p.s.io_dtype == CUM_F16) return launch_seg_io<NW, f16>(p, st);
  return launch_seg_io<NW, float>(p, st);
}

int launch_fwd_segmented(const ScanParams &p, hipStream_t st) {
  switch ((p.s.dstate + NS - 1) / NS) {
    case 1: return launch_seg_nw<1>(p, st);
    case 2: return launch_seg_nw<2>(p, st);
    case 3: return launch_seg_nw<3>(p, st);
    case 4: return launch_seg_nw<4>(p, st);
    case 5: return launch_seg_nw<5>(p, st);
    case 6: return launch_seg_nw<6>(p, st);
    case 7: return launch_seg_nw<7>(p, st);
    default: return launch_seg_nw<8>(p, st);
  }
}

}  // namespace cum
