// Selective scan backward for gfx950.
//
// Replaces selective_scan_cuda.bwd of mamba-ssm 1.2.2 (autograd of the op the reference
// reaches through Mamba.forward, src/network/CleanUMamba.py:289-290 under
// scaler.scale(loss).backward(), src/training/train.py:282-285).
// Gradient formulas: SURVEY.md Appendix A.3.  Mapping: scan_common.h.
//
// The forward saved the state entering every 8-step half of every 16-step chunk.  Chunks are
// walked in reverse; inside a chunk each wave recomputes the states of one half into VGPRs
// (8 x NS registers, 7 forward steps from the saved state) and walks that half backwards; the
// decay factors of five of those steps come back from LDS, the other three are recomputed:
// 7/8 recomputed forward steps + 1 reverse step per time step, 1.25 v_exp_f32 per
// state element).  Sums over the channel axis (dB, dC) are reduced inside the wave with
// v_permlane32_swap / v_permlane16_swap + DPP row adds; sums over workgroups (dB, dC)
// and over the batch (dA, dD, dbias) go through fp32 slabs and a deterministic finalize
// kernel -- no float atomics, bit-reproducible run to run.
#include "scan_reduce.h"

namespace cum {

// -DCUM_SCAN_PROBE (tools/scan_phase_probe.py): wave 0 of every workgroup sums s_memtime deltas per phase of a chunk; the
// totals leave through the delta-bias slab (the op's ddelta_bias output then holds, per channel group, the cycles of phase
// i in channel i).  Not compiled into the shipped library.
#ifdef CUM_SCAN_PROBE
#define PROBE(i)                                                   \
  do {                                                             \
    __builtin_amdgcn_sched_barrier(0);                             \
    const unsigned long long t__ = __builtin_amdgcn_s_memtime();   \
    ph[i] += (float)(t__ - tprev);                                 \
    tprev = t__;                                                   \
    __builtin_amdgcn_sched_barrier(0);                             \
  } while (0)
#else
#define PROBE(i) do { } while (0)
#endif

constexpr int NA = 5;         // reverse steps per 8-step half that take their decay factors from LDS (80 KB at NW = 8)
// BC = 0: B_t / C_t through scalar loads, generic strides; 1: scalar loads, unit stride, all NS states valid;
// 2: the chunk's B / C tiles staged in LDS by the whole workgroup (one coalesced load per chunk, broadcast
// ds_read_b128 per step): no SGPR pressure -- the scalar variants keep 256 B/C values per chunk in flight and spend
// ~20 % of their VALU instructions moving spilled SGPRs through VGPR lanes.
// FULL: dstate == NW * NS known at compile time (every wave owns NS valid states, slab rows are 8-byte aligned pairs).
// YIN: the forward kept y before the gate (ScanParams::ypre_in): the reverse step then neither rebuilds sum_n C x_t (one
// packed fma per state pair, one add and one LDS store per step) nor does phase C sum it over the waves.
template <int NW, int BC, typename TIO, bool FULL, bool YIN = false>
__global__ __launch_bounds__(NW * 64) void scan_bwd_kernel(const ScanParams p) {
  constexpr bool FAST = BC == 1;
  constexpr bool LDSBC = BC == 2;
  constexpr int K = (TB + NW - 1) / NW;
  constexpr int NT = NW * 64;
  constexpr int NP = NW * NS;
  constexpr int BCK = LDSBC ? (TB * NP + NT - 1) / NT : 1;
  __shared__ __attribute__((aligned(16))) float s_B[LDSBC ? TB : 1][LDSBC ? NP : 4];
  __shared__ __attribute__((aligned(16))) float s_C[LDSBC ? TB : 1][LDSBC ? NP : 4];
  __shared__ __attribute__((aligned(16))) float4 s_op[TB][64];   // per (t, d): {delta', delta' u, dy, -}: one 16-byte read per step
  // per wave, step slot and channel: {sum_n g * A' (-> ddelta), sum_n dx * B (-> ddelta, du)}: one 8-byte store per step
  __shared__ __attribute__((aligned(8))) float2 s_p12[NW][SUB][64];
  __shared__ float s_y[YIN ? 1 : NW][YIN ? 1 : SUB][64];   // sum_n C * x_t  (-> dz)
  // decay factors a_t = exp2(dt * A') of the first NA steps of the half being processed: written by the recomputed
  // forward steps, read back by the reverse steps instead of a second v_exp_f32 (each lane reads what it wrote)
  __shared__ __attribute__((aligned(16))) float4 s_a[NA][2][NT];

  const int tid = threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int w = uniform(threadIdx.x >> 6);
  const int b = blockIdx.y;
  const int g = blockIdx.x;
  const int d = g * 64 + lane;
  const int N = FULL ? NW * NS : p.s.dstate, L = p.s.len, Dm = p.s.dim;
  const bool dok = d < Dm;
  const int dc = dok ? d : Dm - 1;
  const int n0 = w * NS;
  const int nvalid = FULL ? NS : ((N - n0) < NS ? (N - n0) : NS);
  const int nchunks = p.nchunks;

  f2 Ap[NP2], dAacc[NP2], dxc[NP2];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int jj = j < nvalid ? j : nvalid - 1;
    const float a = p.A[(int64_t)dc * N + n0 + jj] * kLog2e;
    Ap[j / 2][j % 2] = (j < nvalid) ? a : 0.f;
    dAacc[j / 2][j % 2] = 0.f;
    dxc[j / 2][j % 2] = 0.f;
  }
  const float Dd = p.D ? p.D[dc] : 0.f;
  const float bias = p.bias ? p.bias[dc] : 0.f;
  const TIO *up = static_cast<const TIO *>(p.u) + b * p.s.u_sb + dc * p.s.u_sd;
  const TIO *dtp = static_cast<const TIO *>(p.delta) + b * p.s.dt_sb + dc * p.s.dt_sd;
  const bool has_z = p.z != nullptr;
  const TIO *zp = has_z ? static_cast<const TIO *>(p.z) + b * p.s.z_sb + dc * p.s.z_sd : up;
  const TIO *dop = static_cast<const TIO *>(p.dout) + b * p.s.o_sb + dc * p.s.o_sd;
  const TIO *yip = YIN ? static_cast<const TIO *>(p.ypre_in) + b * p.s.o_sb + dc * p.s.o_sd : dop;
  TIO *dup = static_cast<TIO *>(p.du) + b * p.gs.du_sb + dc * p.gs.du_sd;
  TIO *ddtp = static_cast<TIO *>(p.ddelta) + b * p.gs.dd_sb + dc * p.gs.dd_sd;
  TIO *dzp = has_z ? static_cast<TIO *>(p.dz) + b * p.gs.dz_sb + dc * p.gs.dz_sd : nullptr;
  const int du_sl = (int)p.gs.du_sl, dd_sl = (int)p.gs.dd_sl, dz_sl = (int)p.gs.dz_sl;
  const float *Bw = p.Bm + b * p.s.B_sb + n0 * p.s.B_sn;
  const float *Cw = p.Cm + b * p.s.C_sb + n0 * p.s.C_sn;
  float *wsB = p.ws_dB + ((int64_t)b * p.ngroups + g) * L * N + n0;
  float *wsC = p.ws_dC + ((int64_t)b * p.ngroups + g) * L * N + n0;
  const int u_sl = (int)p.s.u_sl, dt_sl = (int)p.s.dt_sl, z_sl = has_z ? (int)p.s.z_sl : (int)p.s.u_sl;
  const int o_sl = (int)p.s.o_sl;
  const int B_sl = (int)p.s.B_sl, C_sl = (int)p.s.C_sl, B_sn = (int)p.s.B_sn, C_sn = (int)p.s.C_sn;
  const int softplus = p.s.delta_softplus;

  // dB / dC slab stores: the first lane of every quad owns one total (wave_reduce_scatter8x2q): quads 0 / 1 of row q the
  // dB sums of states 2q / 2q + 1 of the wave's slice, quads 2 / 3 the dC sums
  const unsigned qoff = 2u * (lane >> 4) + ((lane >> 2) & 1);
  const bool st_on = (lane & 3) == 0 && (int)qoff < nvalid;

  float accD = 0.f, accBias = 0.f;

  // raw (t, d) rows of the chunk about to be processed; fetched one chunk ahead
  float ru[K], rdl[K], rz[K], rdo[K], ry[K], rb[BCK], rc[BCK];
  const float *Bb = p.Bm + b * p.s.B_sb, *Cb = p.Cm + b * p.s.C_sb;
  auto load_rows = [&](int c) {
    const int t0 = c * TB, tlast = L - 1 - t0;
    if constexpr (LDSBC) {
#pragma unroll
      for (int k = 0; k < BCK; ++k) {
        const int e = tid + k * NT;
        const int tl = e / NP, n = e % NP;
        const int t = t0 + (tl <= tlast ? tl : tlast);
        const int nc = n < N ? n : N - 1;
        const float bvv = Bb[t * B_sl + nc * B_sn], cvv = Cb[t * C_sl + nc * C_sn];
        rb[k] = n < N ? bvv : 0.f;
        rc[k] = n < N ? cvv : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int tl = w + k * NW;
      const int tc = t0 + (tl <= tlast ? tl : tlast);  // clamped address, masked value
      ru[k] = (float)up[tc * u_sl];
      rdl[k] = (float)dtp[tc * dt_sl];
      rz[k] = (float)zp[tc * z_sl];
      rdo[k] = (float)dop[tc * o_sl];
      if constexpr (YIN) ry[k] = (float)yip[tc * o_sl];
    }
  };
  load_rows(nchunks - 1);
#ifdef CUM_SCAN_PROBE
  float ph[12] = {};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif

  for (int c = nchunks - 1; c >= 0; --c) {
    const int t0 = c * TB;
    const int tlast = L - 1 - t0;  // last valid local step of this chunk (>= 0)
    // per-lane slab (dB or dC) + this chunk's first row + per-lane column: the per-step row offset is then an immediate
    float *const cBC = ((lane & 8) ? wsC : wsB) + (int64_t)t0 * N + qoff;
    // state entering the chunk: needed by both halves, requested now so that its latency hides behind phase A
    f2 x0[NP2], x8[NP2];   // x8: state entering the second half (local step 8), read only if the chunk reaches it
    ckpt_load(p.ckpt_in, ckpt_slot(b, nchunks, c, 0, NW, w, Dm, dc), x0);
    ckpt_load(p.ckpt_in, ckpt_slot(b, nchunks, c, tlast >= SUB ? 1 : 0, NW, w, Dm, dc), x8);
    float eu[K], ez[K], edo[K], edt[K], esg[K], ey[K];
    // ---- phase A: per-(t, d) quantities, once, into LDS
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int tl = w + k * NW;
      const bool ok = dok && tl < TB && tl <= tlast;
      const float uv = ru[k], dv = rdl[k], zv = rz[k], dov = rdo[k];
      const float pre = dv + bias;
      float dtv = pre, sg = 1.f;
      if (softplus) {
        dtv = softplus20(pre);
        sg = pre <= 20.f ? sigmoidf_(pre) : 1.f;
      }
      dtv = ok ? dtv : 0.f;
      float dy = ok ? dov : 0.f;
      if (has_z) dy *= zv * sigmoidf_(zv);
      if (tl < TB) s_op[tl][lane] = make_float4(dtv, ok ? dtv * uv : 0.f, ok ? dy : 0.f, 0.f);
      eu[k] = uv; ez[k] = zv; edo[k] = dov; edt[k] = dtv; esg[k] = sg;
      if constexpr (YIN) ey[k] = ry[k];
    }
    if constexpr (LDSBC) {
#pragma unroll
      for (int k = 0; k < BCK; ++k) {
        const int e = tid + k * NT;
        if (e < TB * NP) {
          (&s_B[0][0])[e] = rb[k];
          (&s_C[0][0])[e] = rc[k];
        }
      }
    }
    if (c > 0) load_rows(c - 1);
    PROBE(0);
    __syncthreads();
    PROBE(1);

    // this wave's slices of the B / C tiles: addresses kept in vector registers (left to itself the compiler re-creates the
    // wave-uniform address from a scalar before every step's reads)
    typedef const __attribute__((address_space(3))) float *lds_cfp;
    lds_cfp lB = (lds_cfp)&s_B[0][LDSBC ? n0 : 0], lC = (lds_cfp)&s_C[0][LDSBC ? n0 : 0];
    asm volatile("" : "+v"(lB), "+v"(lC));
    f2 xs[SUB][NP2];   // states before each step of the half being processed
    // Operands of one time step: B_t / C_t slices (SGPRs via s_load) and the per-(t, d) values from LDS.  They are
    // fetched one step ahead of their use so that neither the scalar-load nor the LDS latency is exposed.
    struct StepOps {
      f2 bv[NP2], cv[NP2];
      f2 a[NP2];        // decay factors, only for steps whose slot is < NA
      float dt, du, dy;
    };
    auto fetch = [&](int tl, StepOps &o, int aslot = -1) {
      const int tc = tl <= tlast ? tl : tlast;
      if (aslot >= 0 && aslot < NA) {
        const float4 a0 = s_a[aslot][0][tid], a1 = s_a[aslot][1][tid];
        o.a[0] = f2{a0.x, a0.y}; o.a[1] = f2{a0.z, a0.w}; o.a[2] = f2{a1.x, a1.y}; o.a[3] = f2{a1.z, a1.w};
      }
      if constexpr (LDSBC) {
        // (row tl itself, not the clamped tc: rows past the clip's end hold the last row's values and meet zero dt / du / dy,
        //  and a compile-time tl makes these addresses immediates)
        typedef float f4v __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(3))) f4v *lds_c4p;
        const f4v b0 = *(lds_c4p)(lB + tl * NP), b1 = *(lds_c4p)(lB + tl * NP + 4);
        const f4v c0 = *(lds_c4p)(lC + tl * NP), c1 = *(lds_c4p)(lC + tl * NP + 4);
        o.bv[0] = f2{b0.x, b0.y}; o.bv[1] = f2{b0.z, b0.w}; o.bv[2] = f2{b1.x, b1.y}; o.bv[3] = f2{b1.z, b1.w};
        o.cv[0] = f2{c0.x, c0.y}; o.cv[1] = f2{c0.z, c0.w}; o.cv[2] = f2{c1.x, c1.y}; o.cv[3] = f2{c1.z, c1.w};
      } else {
        float bs[NS], cs[NS];
        load_bc<FAST>(opaque(Bw + (t0 + tc) * B_sl), B_sn, nvalid, bs);
        load_bc<FAST>(opaque(Cw + (t0 + tc) * C_sl), C_sn, nvalid, cs);
#pragma unroll
        for (int j = 0; j < NP2; ++j) {
          o.bv[j] = f2{bs[2 * j], bs[2 * j + 1]};
          o.cv[j] = f2{cs[2 * j], cs[2 * j + 1]};
        }
      }
      const float4 op = s_op[tl][lane];
      o.dt = op.x; o.du = op.y; o.dy = op.z;
    };
    // one recomputed forward step (state only)
    auto fwd_step = [&](f2 (&x)[NP2], const StepOps &o, int aslot = -1) {
      f2 a[NP2];
#pragma unroll
      for (int j = 0; j < NP2; ++j) {
        a[j] = exp2_2(o.dt * Ap[j]);
        x[j] = a[j] * x[j] + o.du * o.bv[j];
      }
      if (aslot >= 0 && aslot < NA) {
        s_a[aslot][0][tid] = make_float4(a[0].x, a[0].y, a[1].x, a[1].y);
        s_a[aslot][1][tid] = make_float4(a[2].x, a[2].y, a[3].x, a[3].y);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // one reverse step; xp = state before step tl; slot = tl % SUB
    auto rev_step = [&](const f2 (&xp)[NP2], int tl, int slot, const StepOps &o) {
      const float dt = o.dt, du = o.du, dy = o.dy;
      f2 p1, p2, yp = {0.f, 0.f};   // even / odd states summed apart, joined below
      float ra[8], rb[8];   // the dB / dC contributions of this step, as the reduction takes them (scan_reduce.h)
#pragma unroll
      for (int j = 0; j < NP2; ++j) {
        const f2 a = slot < NA ? o.a[j] : exp2_2(dt * Ap[j]);
        // the state after this step is the saved state before the next one (recomputed only for the half's last step)
        const f2 xt = slot + 1 < SUB ? xs[slot + 1][j] : a * xp[j] + du * o.bv[j];
        const f2 dx = o.cv[j] * dy + dxc[j];
        if constexpr (!YIN) yp = o.cv[j] * xt + yp;
        // (scalar multiplies on purpose: sixteen free-standing registers for the exchanges instead of eight register
        //  copies out of packed results)
        float *const qB = j < 2 ? ra : rb, *const qC = j < 2 ? ra + 4 : rb + 4;
        qB[2 * (j & 1)] = dx.x * du;
        qB[2 * (j & 1) + 1] = dx.y * du;
        qC[2 * (j & 1)] = dy * xt.x;
        qC[2 * (j & 1) + 1] = dy * xt.y;
        dxc[j] = a * dx;
        const f2 gg = dxc[j] * xp[j];
        dAacc[j] = gg * dt + dAacc[j];
        if (j == 0) {            // (the first pair starts the sums: no zeroed accumulators to set up every step)
          p1 = gg * Ap[j];
          p2 = dx * o.bv[j];
        } else {
          p1 = gg * Ap[j] + p1;
          p2 = dx * o.bv[j] + p2;
        }
      }
      {
        float *const q = reinterpret_cast<float *>(&s_p12[w][slot][lane]);
        q[0] = p1.x + p1.y;
        q[1] = p2.x + p2.y;
      }
      if constexpr (!YIN) s_y[w][slot][lane] = yp.x + yp.y;
      const float tot = wave_reduce_scatter8x2q_s(ra, rb);
      if (tl <= tlast && st_on) cBC[tl * N] = tot;
      __builtin_amdgcn_sched_barrier(0);
    };
    // phase C for one half: combine the per-wave partial sums, write du / ddelta / dz
    auto finish_half = [&](int half) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int tl = w + k * NW;
        const int slot = tl - half * SUB;
        if (tl < TB && slot >= 0 && slot < SUB && tl <= tlast && dok) {
          const int t = t0 + tl;
          float q1 = 0.f, q2 = 0.f, y = YIN ? ey[k] : Dd * eu[k];
#pragma unroll
          for (int ww = 0; ww < NW; ++ww) {
            const float2 q = s_p12[ww][slot][lane];
            q1 += q.x;
            q2 += q.y;
            if constexpr (!YIN) y += s_y[ww][slot][lane];
          }
          const float zv = ez[k], dov = edo[k];
          float dy = dov;
          if (has_z) {
            const float sz = sigmoidf_(zv);
            dy = dov * zv * sz;
            dzp[t * dz_sl] = (TIO)(dov * y * sz * (1.f + zv * (1.f - sz)));
          }
          const float ddt = kLn2 * q1 + eu[k] * q2;  // d loss / d delta'
          const float dpre = ddt * esg[k];
          dup[t * du_sl] = (TIO)fmaf(dy, Dd, edt[k] * q2);
          ddtp[t * dd_sl] = (TIO)dpre;
          accD = fmaf(dy, eu[k], accD);
          accBias += dpre;
        }
      }
    };

    StepOps cur, nxt;
    // ---- second half (local steps 8..15), only if the chunk reaches it
    if (tlast >= SUB) {
      f2 x[NP2];
#pragma unroll
      for (int j = 0; j < NP2; ++j) x[j] = x8[j];
      fetch(SUB, cur);
#pragma unroll
      for (int s = 0; s < SUB; ++s) {          // steps 8..14, keeping the state before every step
#pragma unroll
        for (int j = 0; j < NP2; ++j) xs[s][j] = x[j];
        if (s + 1 < SUB) {
          fetch(SUB + s + 1, nxt);
          fwd_step(x, cur, s);
          cur = nxt;
        }
      }
      PROBE(2);
#pragma unroll
      for (int s = SUB - 1; s >= 0; --s) {     // reverse 15..8 (cur holds step 15's operands)
        if (s > 0) fetch(SUB + s - 1, nxt, s - 1);
        rev_step(xs[s], SUB + s, s, cur);
        if (s > 0) cur = nxt;
      }
      PROBE(3);
      __syncthreads();
      PROBE(4);
      finish_half(1);
      PROBE(5);
      __syncthreads();  // partial-sum buffers are reused by the first half
      PROBE(6);
    }
    // ---- first half (local steps 0..7)
    {
      f2 x[NP2];
#pragma unroll
      for (int j = 0; j < NP2; ++j) x[j] = x0[j];
      fetch(0, cur);
#pragma unroll
      for (int s = 0; s < SUB; ++s) {
#pragma unroll
        for (int j = 0; j < NP2; ++j) xs[s][j] = x[j];
        if (s + 1 < SUB) {
          fetch(s + 1, nxt);
          fwd_step(x, cur, s);
          cur = nxt;
        }
      }
      PROBE(7);
#pragma unroll
      for (int s = SUB - 1; s >= 0; --s) {
        if (s > 0) fetch(s - 1, nxt, s - 1);
        rev_step(xs[s], s, s, cur);
        if (s > 0) cur = nxt;
      }
      PROBE(8);
      __syncthreads();
      PROBE(9);
      finish_half(0);
      PROBE(10);
    }
    // The next chunk's phase A writes only s_op / s_B / s_C (their readers finished before
    // the last barrier) and its first rev_step runs after that phase's barrier, which
    // every wave reaches only after this finish_half.
  }
  // ---- per-(b, d, n) dA and per-(b, d) dD / dbias slabs
  if (dok) {
    float *wa = p.ws_dA + ((int64_t)b * Dm + d) * N + n0;
#pragma unroll
    for (int j = 0; j < NS; ++j)
      if (j < nvalid) wa[j] = dAacc[j / 2][j % 2];
  }
  __syncthreads();
  s_p12[w][0][lane] = make_float2(accD, accBias);
  __syncthreads();
  if (w == 0 && dok) {
    float a = 0.f, c2 = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) {
      const float2 q = s_p12[ww][0][lane];
      a += q.x;
      c2 += q.y;
    }
    p.ws_dD[(int64_t)b * Dm + d] = a;
    p.ws_dbias[(int64_t)b * Dm + d] = c2;
#ifdef CUM_SCAN_PROBE
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) v = lane == i ? ph[i] : v;
    p.ws_dbias[(int64_t)b * Dm + d] = v;
#endif
  }
}

// Deterministic slab reductions: dA, dD, dbias over batch; dB, dC over channel groups.  VEC: a thread owns four
// consecutive dB / dC values (L * N a multiple of 4, 16-byte aligned buffers) and keeps eight 16-byte loads in flight --
// the dB / dC slabs are the bulk of what this kernel reads (B * G * L * N * 8 bytes: 163 MB at the E8 bottleneck).
// dA, dD, dbias over MANY slab rows (time-parallel form: batch x segments rows, 3 840 at the pruned block with 256 clips):
// one wave per output element, lanes over the rows (four loads in flight each), fixed-order lane sums + xor tree ->
// bit-reproducible.  (One thread per output walking the rows one dependent load at a time, as the kernel below does for
// the <= 64 rows of the sequential form, took 3.8 ms there.)
__global__ __launch_bounds__(256) void scan_bwd_finalize_rows_kernel(const ScanParams p, float *dA, float *dD,
                                                                     float *dbias) {
  const int64_t N = p.s.dstate, Dm = p.s.dim, nA = Dm * N, n_out = nA + 2 * Dm;
  const int64_t Bs = (int64_t)p.s.batch * (p.nseg > 1 ? p.nseg : 1);
  const int lane = threadIdx.x & 63;
  for (int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); o < n_out; o += (int64_t)gridDim.x * 4) {
    const float *src;
    int64_t stride;
    float *dst;
    if (o < nA) {
      src = p.ws_dA + o, stride = nA, dst = dA + o;
    } else if (o < nA + Dm) {
      src = p.ws_dD + (o - nA), stride = Dm, dst = dD ? dD + (o - nA) : nullptr;
    } else {
      src = p.ws_dbias + (o - nA - Dm), stride = Dm, dst = dbias ? dbias + (o - nA - Dm) : nullptr;
    }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int64_t b = lane;
    for (; b + 192 < Bs; b += 256) {
      a0 += src[b * stride];
      a1 += src[(b + 64) * stride];
      a2 += src[(b + 128) * stride];
      a3 += src[(b + 192) * stride];
    }
    for (; b < Bs; b += 64) a0 += src[b * stride];
    float v = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0 && dst) *dst = v;
  }
}

template <bool VEC>
__global__ void scan_bwd_finalize_kernel(const ScanParams p, float *dA, float *dD, float *dbias, float *dB,
                                         float *dC, int rows_done) {
  constexpr int V = VEC ? 4 : 1;
  const int64_t N = p.s.dstate, L = p.s.len, Dm = p.s.dim, Bn = p.s.batch, G = p.ngroups;
  const int64_t Bs = Bn * (p.nseg > 1 ? p.nseg : 1);     // rows of the dA / dD / dbias slabs: (batch, segment)
  const int64_t nA = Dm * N, nBC = Bn * L * N, LN = L * N;
  const int64_t total = nA + 2 * Dm + 2 * nBC / V;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nA + 2 * Dm && rows_done) {
      // (summed by scan_bwd_finalize_rows_kernel)
    } else if (i < nA) {
      float s = 0.f;
      for (int64_t b = 0; b < Bs; ++b) s += p.ws_dA[b * nA + i];
      dA[i] = s;
    } else if (i < nA + Dm) {
      const int64_t d = i - nA;
      float s = 0.f;
      for (int64_t b = 0; b < Bs; ++b) s += p.ws_dD[b * Dm + d];
      if (dD) dD[d] = s;
    } else if (i < nA + 2 * Dm) {
      const int64_t d = i - nA - Dm;
      float s = 0.f;
      for (int64_t b = 0; b < Bs; ++b) s += p.ws_dbias[b * Dm + d];
      if (dbias) dbias[d] = s;
    } else {
      int64_t r = (i - nA - 2 * Dm) * V;
      const bool isC = r >= nBC;
      if (isC) r -= nBC;
      const int64_t b = r / LN, tn = r % LN;
      const float *ws = (isC ? p.ws_dC : p.ws_dB) + b * G * LN + tn;
      if constexpr (VEC) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int64_t gg = 0;
        for (; gg + 8 <= G; gg += 8) {
          float4 v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4 *>(ws + (gg + k) * LN);
#pragma unroll
          for (int k = 0; k < 8; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }   // (same order as the scalar form)
        }
        for (; gg < G; ++gg) {
          const float4 v = *reinterpret_cast<const float4 *>(ws + gg * LN);
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4 *>((isC ? dC : dB) + r) = s;
      } else {
        float s = 0.f;
        for (int64_t gg = 0; gg < G; ++gg) s += ws[gg * LN];
        (isC ? dC : dB)[r] = s;
      }
    }
  }
}


template <int NW, typename TIO>
static int launch_bwd_io(const ScanParams &p, hipStream_t st) {
  dim3 grid((p.s.dim + 63) / 64, p.s.batch), block(NW * 64);
#ifdef CUM_AB   // CUM_SCAN_BWD_LDS=0: B_t / C_t through scalar loads instead of the LDS tile (-6.5 % per launch)
  if (cum_knob("CUM_SCAN_BWD_LDS", 1) == 0) {
    if (p.s.B_sn == 1 && p.s.C_sn == 1 && p.s.dstate == NS * NW)
      hipLaunchKernelGGL((scan_bwd_kernel<NW, 1, TIO, false>), grid, block, 0, st, p);
    else
      hipLaunchKernelGGL((scan_bwd_kernel<NW, 0, TIO, false>), grid, block, 0, st, p);
    CUM_CHECK_LAUNCH();
    return CUM_OK;
  }
#endif
  if (p.s.dstate == NS * NW && p.ypre_in && p.z)     // (y only enters dz; shapes with a ragged last wave rebuild it)
    hipLaunchKernelGGL((scan_bwd_kernel<NW, 2, TIO, true, true>), grid, block, 0, st, p);
  else if (p.s.dstate == NS * NW)
    hipLaunchKernelGGL((scan_bwd_kernel<NW, 2, TIO, true>), grid, block, 0, st, p);
  else
    hipLaunchKernelGGL((scan_bwd_kernel<NW, 2, TIO, false>), grid, block, 0, st, p);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

template <int NW>
static int launch_bwd(const ScanParams &p, hipStream_t st) {
  if (p.s.io_dtype == CUM_BF16) return launch_bwd_io<NW, __bf16>(p, st);
  if (p.s.io_dtype == CUM_F16) return launch_bwd_io<NW, f16>(p, st);
  return launch_bwd_io<NW, float>(p, st);
}

}  // namespace cum

using namespace cum;

extern "C" int64_t cum_scan_bwd_workspace_elems(int32_t batch, int32_t dim, int32_t dstate, int32_t len) {
  const int64_t G = (dim + 63) / 64;
  return (int64_t)batch * dim * dstate + 2 * (int64_t)batch * dim + 2 * (int64_t)batch * G * len * dstate;
}

// time-parallel backward (cum_selective_scan_bwd_tp): 0 = "the plan keeps this shape on the sequential kernels"
extern "C" int64_t cum_scan_bwd_tp_workspace_elems(int32_t batch, int32_t dim, int32_t dstate, int32_t len) {
  if (batch <= 0 || dim <= 0 || dstate <= 0 || len <= 0) return 0;
  int nseg = 1, sc = 0;
  scan_seg_plan_bwd(batch, dim, dstate, len, &nseg, &sc);
  if (nseg <= 1) return 0;
  const int64_t G = (dim + 63) / 64;
  return (int64_t)nseg * ((int64_t)batch * dim * dstate + 2 * (int64_t)batch * dim) + 2 * (int64_t)batch * G * len * dstate +
         scan_seg_carry_elems(batch, dim, dstate, nseg);
}

static int scan_bwd_impl(const cum_scan_shape *s, const cum_scan_grad_strides *gs, const void *u, const void *delta,
                         const float *A, const float *Bm, const float *Cm, const float *D, const void *z,
                         const float *delta_bias, const void *dout, const void *y_pre, const float *ckpt, void *du,
                         void *ddelta, float *dA, float *dB, float *dC, float *dD, void *dz, float *ddelta_bias,
                         float *workspace, void *stream, bool time_parallel) {
  if (int rc = scan_check_shape(s)) return rc;
  CUM_REQUIRE(gs && dA, "scan_bwd: null tensor");
  {
    const int64_t lim = 2147483647LL, Lm = s->len > 0 ? s->len - 1 : 0;
    CUM_REQUIRE(gs->du_sl >= 0 && gs->dd_sl >= 0 && gs->dz_sl >= 0 && Lm * gs->du_sl < lim && Lm * gs->dd_sl < lim &&
                    Lm * gs->dz_sl < lim,
                "scan_bwd: gradient time strides must be non-negative and fit in 31 bits");
  }
  CUM_REQUIRE((z == nullptr) == (dz == nullptr), "scan_bwd: z and dz must be given together");
  hipStream_t st = (hipStream_t)stream;
  if (s->batch == 0 || s->len == 0) {
    (void)hipMemsetAsync(dA, 0, sizeof(float) * (size_t)s->dim * s->dstate, st);
    if (dD) (void)hipMemsetAsync(dD, 0, sizeof(float) * s->dim, st);
    if (ddelta_bias) (void)hipMemsetAsync(ddelta_bias, 0, sizeof(float) * s->dim, st);
    return CUM_OK;
  }
  CUM_REQUIRE(u && delta && A && Bm && Cm && dout && du && ddelta && dB && dC, "scan_bwd: null tensor");
  CUM_REQUIRE(ckpt && workspace, "scan_bwd: ckpt and workspace are required");
  ScanParams p{};
  p.s = *s;
  p.gs = *gs;
  p.u = u; p.delta = delta; p.A = A; p.Bm = Bm; p.Cm = Cm; p.D = D; p.z = z; p.bias = delta_bias;
  p.dout = dout; p.ypre_in = y_pre; p.ckpt_in = ckpt; p.du = du; p.ddelta = ddelta; p.dz = dz;
  p.nchunks = (s->len + TB - 1) / TB;
  p.ngroups = (s->dim + 63) / 64;
  p.nseg = 1;
  if (time_parallel) scan_seg_plan_bwd(s->batch, s->dim, s->dstate, s->len, &p.nseg, &p.seg_chunks);
  const int64_t nA = (int64_t)p.nseg * s->batch * s->dim * s->dstate, nD = (int64_t)p.nseg * s->batch * s->dim;
  const int64_t nBC = (int64_t)s->batch * p.ngroups * s->len * s->dstate;
  p.ws_dA = workspace;
  p.ws_dD = workspace + nA;
  p.ws_dbias = p.ws_dD + nD;
  p.ws_dB = p.ws_dbias + nD;
  p.ws_dC = p.ws_dB + nBC;
  p.carry = p.ws_dC + nBC;            // (time-parallel form: leaving dx carries and sums of delta' of the segments)
  int rc;
  switch ((s->dstate + NS - 1) / NS) {
    case 1:
    case 2: rc = launch_bwd_small(p, st); break;
    case 3: rc = launch_bwd<3>(p, st); break;
    case 4: rc = launch_bwd<4>(p, st); break;
    case 5: rc = launch_bwd<5>(p, st); break;
    case 6: rc = launch_bwd<6>(p, st); break;
    case 7: rc = launch_bwd<7>(p, st); break;
    default: rc = launch_bwd<8>(p, st); break;
  }
  if (rc) return rc;
  const int64_t LN = (int64_t)s->len * s->dstate;
  const bool vec = LN % 4 == 0 && (((uintptr_t)dB | (uintptr_t)dC | (uintptr_t)p.ws_dB | (uintptr_t)p.ws_dC) & 15) == 0;
  const int64_t total = (int64_t)s->dim * s->dstate + 2 * s->dim + 2 * (int64_t)s->batch * LN / (vec ? 4 : 1);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  const int rows_done = (int64_t)s->batch * p.nseg > 64;
  if (rows_done) {
    const int64_t n_out = (int64_t)s->dim * s->dstate + 2 * s->dim;
    hipLaunchKernelGGL(scan_bwd_finalize_rows_kernel, dim3((unsigned)((n_out + 3) / 4 < 4096 ? (n_out + 3) / 4 : 4096)),
                       dim3(256), 0, st, p, dA, dD, ddelta_bias);
    CUM_CHECK_LAUNCH();
  }
  if (vec)
    hipLaunchKernelGGL(scan_bwd_finalize_kernel<true>, dim3(blocks), dim3(256), 0, st, p, dA, dD, ddelta_bias, dB, dC,
                       rows_done);
  else
    hipLaunchKernelGGL(scan_bwd_finalize_kernel<false>, dim3(blocks), dim3(256), 0, st, p, dA, dD, ddelta_bias, dB, dC,
                       rows_done);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

extern "C" int cum_selective_scan_bwd(const cum_scan_shape *s, const cum_scan_grad_strides *gs, const void *u,
                                      const void *delta, const float *A, const float *Bm, const float *Cm, const float *D,
                                      const void *z, const float *delta_bias, const void *dout, const void *y_pre,
                                      const float *ckpt, void *du, void *ddelta, float *dA, float *dB, float *dC, float *dD,
                                      void *dz, float *ddelta_bias, float *workspace, void *stream) {
  return scan_bwd_impl(s, gs, u, delta, A, Bm, Cm, D, z, delta_bias, dout, y_pre, ckpt, du, ddelta, dA, dB, dC, dD, dz,
                       ddelta_bias, workspace, stream, false);
}

extern "C" int cum_selective_scan_bwd_tp(const cum_scan_shape *s, const cum_scan_grad_strides *gs, const void *u,
                                         const void *delta, const float *A, const float *Bm, const float *Cm,
                                         const float *D, const void *z, const float *delta_bias, const void *dout,
                                         const void *y_pre, const float *ckpt, void *du, void *ddelta, float *dA, float *dB,
                                         float *dC, float *dD, void *dz, float *ddelta_bias, float *workspace,
                                         void *stream) {
  return scan_bwd_impl(s, gs, u, delta, A, Bm, Cm, D, z, delta_bias, dout, y_pre, ckpt, du, ddelta, dA, dB, dC, dD, dz,
                       ddelta_bias, workspace, stream, true);
}
