// Selective scan backward for d_state <= 16 on gfx950 -- the 442K model (N = 16) and every pruned checkpoint the reference
// ships (N = 8 ... 14): checkpoints/experiments/Experiment_CleanU_Mamba.pkl, checkpoints/pruned/*.pkl.
//
// Replaces selective_scan_cuda.bwd of mamba-ssm 1.2.2 at these sizes (same arithmetic as scan_bwd.hip; SURVEY.md
// Appendix A.3).  The NW-wave kernel of scan_bwd.hip gives such shapes workgroups of one or two waves in which every
// wave also does the per-(t, d) work of 16 / NW rows: at N = 8 that is 144 row registers per lane (spilled) and a chip
// that is half empty (batch * dim / 64 waves on 1024 SIMDs): 6.1 ms at B = 16, D = 2048, L = 2499 against 0.48 ms forward.
//
// Wave-specialised design (the forward's scan_fwd_ws_kernel carried over):
//   * a workgroup = NW CONSUMER waves (8 states each; lane <-> channel, 64 channels) + a LOADER wave + a FINISHER wave,
//     on different SIMDs of a CU;
//   * the unit of the pipeline is the 8-step half the forward checkpointed; halves are walked in reverse, one per
//     barrier interval;
//   * the loader owns what depends on (t, d) only, on the way IN: it loads the u / delta / z / dout rows (one half ahead,
//     in registers) and the B_t / C_t tile, computes softplus, its derivative, the SiLU gate and its derivative ONCE per
//     (t, d) into an LDS ring slot one interval before the consumers need it;
//   * a consumer loads the saved state entering the half, recomputes the seven states after it (decay factors kept in
//     registers: 1.0 v_exp_f32 per state element, no LDS parking), walks the half backwards, reduce-scatters the dB / dC
//     contributions over its 64 channels (scan_reduce.h) into the f32 slabs, and leaves three partial sums per (t, d)
//     (sum_n g A', sum_n dx B, sum_n C x) in LDS;
//   * the finisher owns the way OUT, one interval behind: partial sums + the loader's operands -> du / ddelta / dz and
//     the dD / dbias accumulators -- the consumers never touch a transcendental other than the v_exp_f32 of the state
//     update and never store an activation gradient;
//   * ONE workgroup barrier per half.  In interval i the loader writes ring slot (i + 1) % 3, the consumers read slot
//     i % 3 and the finisher slot (i - 1) % 3; partial sums and B / C tiles alternate between two slots.
//   (First version: one producer wave doing both the loader's and the finisher's work -- as many instructions per half
//   as a consumer, 1.13 ms at B = 16, D = 2048, N = 8, L = 2499 with 36 % of the wave time parked at the barrier.)
// Sums over workgroups / batch go through the slabs and scan_bwd_finalize_kernel: no float atomics, bit-reproducible.
#include "scan_reduce.h"

namespace cum {

// PASS 0: the whole backward over the workgroup's range of halves -- all of them (sequential form), or one SEGMENT of
// whole chunks (time-parallel form, blockIdx.z = segment) entered with the dx carry composed from the later segments.
// PASS 1 (time-parallel form only, segments 1 .. nseg - 1): nothing but the reverse recurrence g <- a_t (C_t dy_t + g)
// from a zero carry and the segment's sum of delta' -> p.carry, as scan_seg.hip's pass 1 does for the forward: the
// reverse recurrence is the same linear operator run backwards, a segment's decay is exp2(A' sum delta') again.
template <int NW, typename TIO, int PASS>
__global__ __launch_bounds__((NW + 2) * 64) void scan_bwd_ws_kernel(const ScanParams p) {
  constexpr int NPD = NW * NS;               // padded state count: 8 or 16
  constexpr int BCE = SUB * 2 * NPD / 64;    // B / C tile elements per loader lane and half
  // per-(t, d) operands: written by the loader wave one half ahead, read by the consumers (dt, u, dy) in the half's own
  // interval and by the finisher wave (all five) one interval later -> a ring of three slots
  // s_op: {dt = softplus(delta + bias), du = dt * u, dy = dout * silu(z), u}; dt, du, dy are 0 for masked steps / lanes
  // s_dv: {d softplus, dout * d silu(z)}.  One 16-byte / 8-byte access per lane instead of five 4-byte ones.
  __shared__ __attribute__((aligned(16))) float4 s_op[3][SUB][64];
  __shared__ __attribute__((aligned(8))) float2 s_dv[3][SUB][64];
  // consumers' per-(t, d) partial sums: written in interval i, read by the finisher in interval i + 1 -> two slots
  __shared__ float s_p1[2][NW][SUB][64];     // sum_n g A'
  __shared__ float s_p2[2][NW][SUB][64];     // sum_n dx B
  __shared__ float s_y[2][NW][SUB][64];      // sum_n C x_t
  __shared__ __attribute__((aligned(16))) float s_bc[2][SUB][2 * NPD];

  const int lane = threadIdx.x & 63;
  const int w = uniform(threadIdx.x >> 6);
  const int b = blockIdx.y;
  const int g = blockIdx.x;
  const int d = g * 64 + lane;
  const int N = p.s.dstate, L = p.s.len, Dm = p.s.dim;
  const bool dok = d < Dm;
  const int dc = dok ? d : Dm - 1;           // lanes past the last channel walk a valid one; their values are zeroed
  const int nchunks = p.nchunks, nseg = p.nseg;
  const int seg = nseg > 1 ? (int)blockIdx.z + (PASS == 1 ? 1 : 0) : 0;
  const int nh_all = (L + SUB - 1) / SUB;
  const int h_lo = nseg > 1 ? 2 * seg * p.seg_chunks : 0;
  const int hb = nseg > 1 ? (h_lo + 2 * p.seg_chunks < nh_all ? h_lo + 2 * p.seg_chunks : nh_all) : nh_all;
  const int nh = hb - h_lo;                  // halves of this workgroup, walked hb - 1 ... h_lo: interval i handles hb - 1 - i
  const int64_t slab = (int64_t)b * nseg + seg;                      // row of the dA / dD / dbias slabs
  float *xsum = p.carry + (int64_t)p.s.batch * nseg * NW * Dm * NS;   // [(b, seg, d)] sums of delta' (time-parallel form)
  const bool has_z = p.z != nullptr;

  if (w == NW) {
    // ------------------------------------------------------------------------------------------------ loader
    // rows of u / delta / z / dout and the B / C tile, one half ahead in registers; softplus, its derivative, the SiLU
    // gate and its derivative once per (t, d) into the ring
    const float bias = p.bias ? p.bias[dc] : 0.f;
    // wave-uniform row bases + one per-lane 32-bit offset: the per-row address arithmetic stays on the scalar unit
    const TIO *ub = static_cast<const TIO *>(p.u) + b * p.s.u_sb;
    const TIO *db = static_cast<const TIO *>(p.delta) + b * p.s.dt_sb;
    const TIO *zb = has_z ? static_cast<const TIO *>(p.z) + b * p.s.z_sb : ub;
    const TIO *ob = static_cast<const TIO *>(p.dout) + b * p.s.o_sb;
    const int u_lo = dc * (int)p.s.u_sd, d_lo = dc * (int)p.s.dt_sd, z_lo = has_z ? dc * (int)p.s.z_sd : u_lo;
    const int o_lo = dc * (int)p.s.o_sd;
    const int u_sl = (int)p.s.u_sl, dt_sl = (int)p.s.dt_sl, z_sl = has_z ? (int)p.s.z_sl : (int)p.s.u_sl;
    const int o_sl = (int)p.s.o_sl;
    const float *Bb = p.Bm + b * p.s.B_sb, *Cb = p.Cm + b * p.s.C_sb;
    const int B_sl = (int)p.s.B_sl, C_sl = (int)p.s.C_sl, B_sn = (int)p.s.B_sn, C_sn = (int)p.s.C_sn;
    const int softplus = p.s.delta_softplus;

    float ru[SUB], rdl[SUB], rz[SUB], rdo[SUB], rbc[BCE];
    auto load_rows = [&](int h) {
      const int t0 = h * SUB;
      const bool full = t0 + SUB <= L;       // wave-uniform: the clamps below are scalar selects
      const TIO *u0 = ub + (int64_t)t0 * u_sl, *d0 = db + (int64_t)t0 * dt_sl, *z0 = zb + (int64_t)t0 * z_sl;
      const TIO *o0 = ob + (int64_t)t0 * o_sl;
#pragma unroll
      for (int k = 0; k < SUB; ++k) {
        const int kk = full ? k : (t0 + k < L ? k : L - 1 - t0);     // clamped row, masked value
        ru[k] = (float)u0[kk * u_sl + u_lo];
        rdl[k] = (float)d0[kk * dt_sl + d_lo];
        rz[k] = (float)z0[kk * z_sl + z_lo];
        rdo[k] = (float)o0[kk * o_sl + o_lo];
      }
#pragma unroll
      for (int k = 0; k < BCE; ++k) {        // this lane's elements of the [SUB][B | C] tile
        const int e = lane + 64 * k;
        const int tl = e / (2 * NPD), j = e % (2 * NPD);
        int t = t0 + tl;
        if (!full) t = t < L ? t : L - 1;
        const bool isC = j >= NPD;
        const int n = isC ? j - NPD : j;
        const int nc = n < N ? n : N - 1;
        const float v = isC ? Cb[t * C_sl + nc * C_sn] : Bb[t * B_sl + nc * B_sn];
        rbc[k] = n < N ? v : 0.f;            // states past d_state: B = C = 0 (and A' = 0) -> they stay zero everywhere
      }
    };
    auto prepare = [&](int h, int i) {       // the rows in registers are half h's -> ring slot i % 3, tile slot i & 1
      const int t0 = h * SUB, slot = i % 3;
#pragma unroll
      for (int k = 0; k < SUB; ++k) {
        const bool ok = dok && t0 + k < L;
        const float uv = ru[k], zv = rz[k], dov = rdo[k];
        const float pre = rdl[k] + bias;
        float dtv = pre, sg = 1.f;
        if (softplus) {
          dtv = softplus20(pre);
          sg = pre <= 20.f ? sigmoidf_(pre) : 1.f;
        }
        float dy = dov, gz = 0.f;
        if (has_z) {
          const float sz = sigmoidf_(zv);
          dy = dov * zv * sz;
          gz = dov * sz * (1.f + zv * (1.f - sz));
        }
        dtv = ok ? dtv : 0.f;
        s_op[slot][k][lane] = make_float4(dtv, dtv * uv, ok ? dy : 0.f, uv);
        s_dv[slot][k][lane] = make_float2(sg, gz);
      }
#pragma unroll
      for (int k = 0; k < BCE; ++k) (&s_bc[i & 1][0][0])[lane + 64 * k] = rbc[k];
    };
    load_rows(hb - 1);
    prepare(hb - 1, 0);
    if (nh > 1) load_rows(hb - 2);
    for (int i = 0; i < nh; ++i) {
      __syncthreads();                       // interval i: the consumers walk half hb - 1 - i
      if (i + 1 < nh) {
        prepare(hb - 2 - i, i + 1);
        if (i + 2 < nh) load_rows(hb - 3 - i);
      }
    }
    __syncthreads();
    return;
  }

  if (w == NW + 1) {
    // ------------------------------------------------------------------------------------------------ finisher
    // one interval behind the consumers: their per-(t, d) partial sums -> du / ddelta / dz, dD / dbias accumulators
    const float Dd = p.D ? p.D[dc] : 0.f;
    TIO *dub = static_cast<TIO *>(p.du) + b * p.gs.du_sb;
    TIO *ddb = static_cast<TIO *>(p.ddelta) + b * p.gs.dd_sb;
    TIO *dzb = has_z ? static_cast<TIO *>(p.dz) + b * p.gs.dz_sb : nullptr;
    const int du_lo = dc * (int)p.gs.du_sd, dd_lo = dc * (int)p.gs.dd_sd, dz_lo = has_z ? dc * (int)p.gs.dz_sd : 0;
    const int du_sl = (int)p.gs.du_sl, dd_sl = (int)p.gs.dd_sl, dz_sl = (int)p.gs.dz_sl;
    float accD = 0.f, accBias = 0.f;
    auto epilogue = [&](int h, int i) {      // half h was walked in interval i
      const int t0 = h * SUB, slot = i % 3, ps = i & 1;
      TIO *du0 = dub + (int64_t)t0 * du_sl, *dd0 = ddb + (int64_t)t0 * dd_sl;
      TIO *dz0 = has_z ? dzb + (int64_t)t0 * dz_sl : nullptr;
#pragma unroll
      for (int k = 0; k < SUB; ++k) {
        float q1 = 0.f, q2 = 0.f, y = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
          q1 += s_p1[ps][ww][k][lane];
          q2 += s_p2[ps][ww][k][lane];
          y += s_y[ps][ww][k][lane];
        }
        const float4 op = s_op[slot][k][lane];
        const float2 dv = s_dv[slot][k][lane];
        const float uv = op.w, dtv = op.x, dy = op.z, sg = dv.x, gz = dv.y;
        if (dok && t0 + k < L) {
          y = fmaf(Dd, uv, y);
          if (has_z) dz0[k * dz_sl + dz_lo] = (TIO)(gz * y);
          const float ddt = kLn2 * q1 + uv * q2;  // d loss / d delta'
          const float dpre = ddt * sg;
          du0[k * du_sl + du_lo] = (TIO)fmaf(dy, Dd, dtv * q2);
          dd0[k * dd_sl + dd_lo] = (TIO)dpre;
          accD = fmaf(dy, uv, accD);
          accBias += dpre;
        }
      }
    };
    for (int i = 0; i < nh; ++i) {
      __syncthreads();
      if (PASS == 0 && i > 0) epilogue(hb - i, i - 1);
    }
    __syncthreads();
    if constexpr (PASS == 0) {
      epilogue(h_lo, nh - 1);
      if (dok) {
        p.ws_dD[slab * Dm + d] = accD;
        p.ws_dbias[slab * Dm + d] = accBias;
      }
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- consumers
  const int n0 = w * NS;
  const int nvalid = (N - n0) < NS ? (N - n0) : NS;
  f2 Ap[NP2], dAacc[NP2], dxc[NP2];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int jj = j < nvalid ? j : nvalid - 1;
    const float a = p.A[(int64_t)dc * N + n0 + jj] * kLog2e;
    Ap[j / 2][j % 2] = (j < nvalid) ? a : 0.f;
    dAacc[j / 2][j % 2] = 0.f;
    dxc[j / 2][j % 2] = 0.f;
  }
  float *wsB = p.ws_dB + ((int64_t)b * p.ngroups + g) * L * N + n0;
  float *wsC = p.ws_dC + ((int64_t)b * p.ngroups + g) * L * N + n0;
  // dB / dC slab stores: the first lane of every quad owns one total (wave_reduce_scatter8x2q): quads 0 / 1 of row q the
  // dB sums of states 2q / 2q + 1 of the wave's slice, quads 2 / 3 the dC sums
  const unsigned qoff = 2u * (lane >> 4) + ((lane >> 2) & 1);
  const bool st_on = (lane & 3) == 0 && (int)qoff < nvalid;
  float *wsBC = ((lane & 8) ? wsC : wsB) + qoff;

  auto ck = [&](int h) { return ckpt_slot(b, nchunks, h >> 1, h & 1, NW, w, Dm, dc); };
  if constexpr (PASS == 1) {
    // ---- time-parallel pass 1: the dx recurrence of this segment from a zero carry, and its sum of delta'
    float dsum = 0.f;
    for (int i = 0; i < nh; ++i) {
      const int slot = i % 3, ps = i & 1;
      __syncthreads();
      const float (*tile)[2 * NPD] = s_bc[ps];
#pragma unroll
      for (int k = SUB - 1; k >= 0; --k) {
        const float4 o4 = s_op[slot][k][lane];
        const float dt = o4.x, dy = o4.z;
        const float4 c0 = *reinterpret_cast<const float4 *>(&tile[k][NPD + n0]), c1 = *reinterpret_cast<const float4 *>(&tile[k][NPD + n0 + 4]);
        const f2 cv[NP2] = {f2{c0.x, c0.y}, f2{c0.z, c0.w}, f2{c1.x, c1.y}, f2{c1.z, c1.w}};
#pragma unroll
        for (int j = 0; j < NP2; ++j) dxc[j] = exp2_2(dt * Ap[j]) * (cv[j] * dy + dxc[j]);
        dsum += dt;
      }
    }
    __syncthreads();
    if (dok) {
      ckpt_store(p.carry, carry_slot(b, nseg, seg, NW, w, Dm, d), dxc);
      if (w == 0) xsum[slab * Dm + d] = dsum;
    }
    return;
  }
  if (nseg > 1) {
    // the carry entering this segment from the future: compose the (decay, g leaving from zero) pairs of all later
    // segments, last first -- dependent fma chains on loads that do not depend on each other
#pragma unroll 4
    for (int sp = nseg - 1; sp > seg; --sp) {
      f2 e[NP2];
      ckpt_load(p.carry, carry_slot(b, nseg, sp, NW, w, Dm, dc), e);
      const float ds = xsum[((int64_t)b * nseg + sp) * Dm + dc];
#pragma unroll
      for (int j = 0; j < NP2; ++j) dxc[j] = exp2_2(ds * Ap[j]) * dxc[j] + e[j];
    }
  }
  f2 xn[NP2];                                // state entering the next half to be processed, requested one half ahead
  ckpt_load(p.ckpt_in, ck(hb - 1), xn);

  for (int i = 0; i < nh; ++i) {
    const int h = hb - 1 - i, slot = i % 3, ps = i & 1, t0 = h * SUB;
    f2 x[NP2];
#pragma unroll
    for (int j = 0; j < NP2; ++j) x[j] = xn[j];
    __syncthreads();                         // the producer has finished this half's slot
    if (h > h_lo) ckpt_load(p.ckpt_in, ck(h - 1), xn);
    const float (*tile)[2 * NPD] = s_bc[ps];
    f2 xs[SUB][NP2];                         // state before each step
    f2 as[SUB - 1][NP2];                     // decay factors of steps 0 .. 6 (step 7's is formed in the reverse walk)
    // ---- recompute the states of the half
#pragma unroll
    for (int k = 0; k < SUB; ++k) {
#pragma unroll
      for (int j = 0; j < NP2; ++j) xs[k][j] = x[j];
      if (k + 1 < SUB) {
        const float2 o2 = *reinterpret_cast<const float2 *>(&s_op[slot][k][lane]);
        const float dt = o2.x, du = o2.y;
        const float4 b0 = *reinterpret_cast<const float4 *>(&tile[k][n0]), b1 = *reinterpret_cast<const float4 *>(&tile[k][n0 + 4]);
        const f2 bv[NP2] = {f2{b0.x, b0.y}, f2{b0.z, b0.w}, f2{b1.x, b1.y}, f2{b1.z, b1.w}};
#pragma unroll
        for (int j = 0; j < NP2; ++j) {
          as[k][j] = exp2_2(dt * Ap[j]);
          x[j] = as[k][j] * x[j] + du * bv[j];
        }
      }
    }
    // ---- walk it backwards
#pragma unroll
    for (int k = SUB - 1; k >= 0; --k) {
      const float4 o4 = s_op[slot][k][lane];
      const float dt = o4.x, du = o4.y, dy = o4.z;
      const float4 b0 = *reinterpret_cast<const float4 *>(&tile[k][n0]), b1 = *reinterpret_cast<const float4 *>(&tile[k][n0 + 4]);
      const float4 c0 = *reinterpret_cast<const float4 *>(&tile[k][NPD + n0]), c1 = *reinterpret_cast<const float4 *>(&tile[k][NPD + n0 + 4]);
      const f2 bv[NP2] = {f2{b0.x, b0.y}, f2{b0.z, b0.w}, f2{b1.x, b1.y}, f2{b1.z, b1.w}};
      const f2 cv[NP2] = {f2{c0.x, c0.y}, f2{c0.z, c0.w}, f2{c1.x, c1.y}, f2{c1.z, c1.w}};
      f2 p1 = {0.f, 0.f}, p2 = {0.f, 0.f}, yp = {0.f, 0.f};   // even / odd states summed apart, joined below
      f2 dBp[NP2], dCp[NP2];
#pragma unroll
      for (int j = 0; j < NP2; ++j) {
        const f2 a = k + 1 < SUB ? as[k < SUB - 1 ? k : 0][j] : exp2_2(dt * Ap[j]);
        // the state after this step is the saved state before the next one (recomputed only for the half's last step)
        const f2 xt = k + 1 < SUB ? xs[k + 1 < SUB ? k + 1 : 0][j] : a * xs[k][j] + du * bv[j];
        const f2 dx = cv[j] * dy + dxc[j];
        yp = cv[j] * xt + yp;
        dCp[j] = dy * xt;
        dBp[j] = dx * du;
        dxc[j] = a * dx;
        const f2 gg = dxc[j] * xs[k][j];
        dAacc[j] = gg * dt + dAacc[j];
        p1 = gg * Ap[j] + p1;
        p2 = dx * bv[j] + p2;
      }
      s_p1[ps][w][k][lane] = p1.x + p1.y;
      s_p2[ps][w][k][lane] = p2.x + p2.y;
      s_y[ps][w][k][lane] = yp.x + yp.y;
      const float tot = wave_reduce_scatter8x2q(dBp, dCp);
      if (t0 + k < L && st_on) wsBC[(int64_t)(t0 + k) * N] = tot;
    }
  }
  __syncthreads();                           // lets the finisher close the last half
  if (dok) {
    float *wa = p.ws_dA + (slab * Dm + d) * N + n0;
#pragma unroll
    for (int j = 0; j < NS; ++j)
      if (j < nvalid) wa[j] = dAacc[j / 2][j % 2];
  }
}

template <int NW, typename TIO>
static int launch_small_io(const ScanParams &p, hipStream_t st) {
  const dim3 block((NW + 2) * 64);
  if (p.nseg > 1) {       // time-parallel: pass 1 over segments 1 .. nseg - 1, then every segment with its composed carry
    hipLaunchKernelGGL((scan_bwd_ws_kernel<NW, TIO, 1>), dim3((p.s.dim + 63) / 64, p.s.batch, p.nseg - 1), block, 0, st, p);
    CUM_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL((scan_bwd_ws_kernel<NW, TIO, 0>), dim3((p.s.dim + 63) / 64, p.s.batch, p.nseg > 1 ? p.nseg : 1), block,
                     0, st, p);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}

template <int NW>
static int launch_small(const ScanParams &p, hipStream_t st) {
  if (p.s.io_dtype == CUM_BF16) return launch_small_io<NW, __bf16>(p, st);
  if (p.s.io_dtype == CUM_F16) return launch_small_io<NW, f16>(p, st);
  return launch_small_io<NW, float>(p, st);
}

int launch_bwd_small(const ScanParams &p, hipStream_t st) {
  return p.s.dstate <= NS ? launch_small<1>(p, st) : launch_small<2>(p, st);
}

}  // namespace cum
