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

// ---- Lane <-> state-slot permutation (the "xor scatter").
// The per-step sums over the 64 channels of a wave (dB_t[n], dC_t[n]: 16 values per lane) are a reduce-scatter over the
// lanes.  With every lane holding state n in the SAME register, one exchange level costs two instructions per eliminated
// register (v_permlane*_swap + v_add, or two bank-masked DPP adds): 32 cross-lane instructions + 5 hazard pads per step,
// 30 % of the kernel's instruction stream (profiles/r04_scan_bwd_experiments.md).  Here lane l keeps state  k ^ h(l)  in
// register slot k, with  h(l) = b0 ^ 2 b1 ^ 7 b2  (b_i = bit i of l).  Then the partner lane of a DPP exchange holds the
// state this lane keeps in the NEIGHBOUR slot, and "my slot k + partner's slot k ^ m" is ONE v_add_f32_dpp that
// eliminates one register:
//   level 1: lanes l, l ^ 1 (quad_perm [1,0,3,2]), slots k, k ^ 1      8 -> 4 registers per array, 4 instructions
//   level 2: lanes l, l ^ 2 (quad_perm [2,3,0,1]), slots k, k ^ 2      4 -> 2, 2 instructions
//   level 3: lanes l, l ^ 7 (row_half_mirror),     slots k, k ^ 4      2 -> 1, 1 instruction  (h(l ^ 7) = h(l) ^ 4)
//   level 4: lanes l, l ^ 8 (row_ror:8), bank-masked: lanes 0-7 of a row keep the dB sum, lanes 8-15 the dC sum
// = 16 DPP adds per step (and no pad: the two odd hazard slots carry the step's two pair sums).  The sum over the four
// 16-lane rows is deferred: the 8 per-step registers of a half are reduce-scattered over the rows with 4 + 2 swaps, so a
// half ends with two registers in which EVERY lane holds one finished total:
//   lane l:  array = bit 3 of l (0: dB, 1: dC),  state = h(l & 7),  step = 4 * (register) + 3 - (2 * (r & 1) + (r >> 1)), r = l >> 4
// and leaves through two 64-lane stores per half instead of eight 16-lane ones.
// What the permutation costs: B_t / C_t must reach a lane in ITS slot order.  Bit 2 of h swaps the two 16-byte halves of
// the wave's 8-state slice (an address bit); bits 0, 1 permute inside a float4, so the half-chunk's B / C tiles are staged
// in LDS in four variants v (position q of variant v holds state q ^ v), each row rotated by 16 v floats so that the
// eight distinct float4 a wave reads per step fall on eight different bank groups.  Checkpoints, A and dA are permuted
// by their addresses (4-byte accesses).  tools/xor_scatter_model.py is the numpy model of the lane algebra.
__device__ __forceinline__ int lane_hmask(int lane) {
  const int b2 = (lane >> 2) & 1;
  return ((lane ^ b2) & 1) | ((((lane >> 1) ^ b2) & 1) << 1) | (b2 << 2);
}

// One reverse step's reduction.  b[k] = dx * (delta' u) of slot k (formed by the caller with scalar multiplies: eight
// free-standing registers -- halves of packed results handed to an asm as read-write operands cost a register copy per
// pair), xt[k] = x_t of slot k, read only: the dC products dy * x_t are formed INSIDE the block, each into a register the
// dB chain has just finished with (b1, b3, b5, b7 after level 1, b2, b6 after level 2), so that the sixteen products
// never exist together (eight registers less at the kernel's 256-register peak) and serve as the hazard cover of the
// other chain (a VALU result needs two wait states before a DPP read).  On return X = row-level totals (level 4 above),
// q0 = p1x + p1y, q1 = p2x + p2y (the step's two pair sums: two more hazard slots).
__device__ __forceinline__ void xor_reduce16(float (&b)[8], const float (&xt)[8], float dy, float p1x, float p1y, float p2x,
                                             float p2y, float &X, float &q0, float &q1) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %3, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %5, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %7, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32 %1, %19, %11\n\t"                                                     // c0 -> b1
      "v_mul_f32 %3, %19, %12\n\t"                                                     // c1 -> b3
      "v_mul_f32 %5, %19, %13\n\t"                                                     // c2 -> b5
      "v_mul_f32 %7, %19, %14\n\t"                                                     // c3 -> b7
      "v_add_f32_dpp %0, %2, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %6, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32 %2, %19, %15\n\t"                                                     // c4 -> b2
      "v_mul_f32 %6, %19, %16\n\t"                                                     // c5 -> b6
      "v_add_f32_dpp %1, %3, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"    // c0 += c1
      "v_add_f32_dpp %5, %7, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"    // c2 += c3
      "v_mul_f32 %3, %19, %17\n\t"                                                     // c6 -> b3
      "v_mul_f32 %7, %19, %18\n\t"                                                     // c7 -> b7
      "v_add_f32_dpp %2, %6, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"    // c4 += c5
      "v_add_f32_dpp %1, %5, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"    // c01 += c23
      "v_add_f32_dpp %3, %7, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"    // c6 += c7
      "v_add_f32 %9, %20, %21\n\t"
      "v_add_f32_dpp %0, %4, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"        // dB: level 3
      "v_add_f32_dpp %2, %3, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"    // c45 += c67
      "v_add_f32 %10, %22, %23\n\t"
      "v_add_f32_dpp %8, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"              // level 4, dB lanes
      "v_add_f32_dpp %1, %2, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"        // dC: level 3
      "s_nop 1\n\t"
      "v_add_f32_dpp %8, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc"                   // level 4, dC lanes
      : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "=&v"(X),
        "=&v"(q0), "=&v"(q1)
      : "v"(xt[0]), "v"(xt[1]), "v"(xt[2]), "v"(xt[3]), "v"(xt[4]), "v"(xt[5]), "v"(xt[6]), "v"(xt[7]), "v"(dy), "v"(p1x),
        "v"(p1y), "v"(p2x), "v"(p2y));
}

// a + b after a half / row exchange: lanes 0-31 (even rows) end with the two-half (two-row) sums of a, the others with b's
__device__ __forceinline__ float swap32_add(float a, float b) {
  swap32(a, b);
  return a + b;
}
__device__ __forceinline__ float swap16_add(float a, float b) {
  swap16(a, b);
  return a + b;
}

// FULL: dstate == NW * NS known at compile time (every wave owns NS valid states).
// YIN: the forward kept y before the gate (ScanParams::ypre_in): the reverse step then neither rebuilds sum_n C x_t (one
// packed fma per state pair, one add and one LDS store per step) nor does phase C sum it over the waves.
template <int NW, typename TIO, bool FULL, bool YIN = false>
__global__ __launch_bounds__(NW * 64) void scan_bwd_kernel(const ScanParams p) {
  constexpr int K = (TB + NW - 1) / NW;
  constexpr int NT = NW * 64;
  constexpr int NP = NW * NS;
  static_assert(NT == SUB * NP, "one B / C element per thread and half");
  // reverse steps per 8-step half that take their decay factors from LDS (16 KB each at NW = 8); the variant that also
  // keeps the per-wave partial y in LDS gives one up to stay inside 160 KB
  constexpr int NA = (!YIN && NW == 8) ? 4 : 5;
  // the current half's B / C tiles: [B | C][variant][row][64 floats, rotated by 16 * variant]
  __shared__ __attribute__((aligned(16))) float s_bc[2][4][SUB][64];
  __shared__ __attribute__((aligned(16))) float4 s_op[TB][64];   // per (t, d): {delta', delta' u, dy, u}: one 16-byte read per step
  // per (t, d), for phase C only: {dout * d silu(z) / dz (-> dz = that * y), d softplus (-> ddelta), y before the gate, -}.
  // (Phase A's values used to wait in 12 registers per lane across both walks; at 256 registers a spilled value costs a
  //  scratch reload + s_waitcnt vmcnt(0), which also waits for every row and checkpoint load in flight.)
  __shared__ __attribute__((aligned(16))) float4 s_fin[TB][64];
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
  const int hm = lane_hmask(lane);       // slot k of this lane holds state n0 + (k ^ hm)
#ifdef CUM_BWD_PRIO   // experiment: static priority for the second-dispatched half of the workgroup (MI355X_MICROARCH.md, item 4)
  if (NW == 8 && w >= 4) __builtin_amdgcn_s_setprio(1);
#endif

  f2 Ap[NP2], dAacc[NP2], dxc[NP2];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int j = k ^ hm;
    const int jj = j < nvalid ? j : nvalid - 1;
    const float a = scan_A(p, (int64_t)dc * N + n0 + jj) * kLog2e;
    Ap[k / 2][k % 2] = (j < nvalid) ? a : 0.f;
    dAacc[k / 2][k % 2] = 0.f;
    dxc[k / 2][k % 2] = 0.f;
  }
  const float Dd = p.D ? p.D[dc] : 0.f;
  const float bias = p.bias ? p.bias[dc] : 0.f;
  // wave-uniform bases (scalar registers) + one 32-bit element offset per lane and tensor (scan_check_shape bounds them)
  const TIO *up = static_cast<const TIO *>(p.u) + b * p.s.u_sb;
  const TIO *dtp = static_cast<const TIO *>(p.delta) + b * p.s.dt_sb;
  const bool has_z = p.z != nullptr;
  const TIO *zp = has_z ? static_cast<const TIO *>(p.z) + b * p.s.z_sb : up;
  const TIO *dop = static_cast<const TIO *>(p.dout) + b * p.s.o_sb;
  const TIO *yip = YIN ? static_cast<const TIO *>(p.ypre_in) + b * p.s.o_sb : dop;
  TIO *dup = static_cast<TIO *>(p.du) + b * p.gs.du_sb;
  TIO *ddtp = static_cast<TIO *>(p.ddelta) + b * p.gs.dd_sb;
  TIO *dzp = has_z ? static_cast<TIO *>(p.dz) + b * p.gs.dz_sb : nullptr;
  const int u_o = dc * (int)p.s.u_sd, dt_o = dc * (int)p.s.dt_sd, z_o = has_z ? dc * (int)p.s.z_sd : u_o;
  const int o_o = dc * (int)p.s.o_sd;
  const int du_o = dc * (int)p.gs.du_sd, dd_o = dc * (int)p.gs.dd_sd, dz_o = dc * (int)p.gs.dz_sd;
  const int du_sl = (int)p.gs.du_sl, dd_sl = (int)p.gs.dd_sl, dz_sl = (int)p.gs.dz_sl;
  const int u_sl = (int)p.s.u_sl, dt_sl = (int)p.s.dt_sl, z_sl = has_z ? (int)p.s.z_sl : (int)p.s.u_sl;
  const int o_sl = (int)p.s.o_sl;
  const int B_sl = (int)p.s.B_sl, C_sl = (int)p.s.C_sl, B_sn = (int)p.s.B_sn, C_sn = (int)p.s.C_sn;
  const int softplus = p.s.delta_softplus & kScanSoftplus;

  // dB / dC slab stores (see the header): this lane's array, state and step offset inside a group of four steps
  const int st_state = lane_hmask(lane & 7);
  const int st_row = lane >> 4;
  const int st_step = 3 - (2 * (st_row & 1) + (st_row >> 1));
  const int st_lim = st_state < nvalid ? st_step : (1 << 24);      // (a slot without a state never passes the step test)
  // wave-uniform slab bases (scalar registers) + one 32-bit element offset per lane; which slab a lane stores to is
  // selected where it stores (two v_cndmask per store, two stores per half) instead of living in a 64-bit register pair
  // per lane across the walks
  float *const sB = p.ws_dB + ((int64_t)b * p.ngroups + g) * L * N + n0;
  float *const sC = p.ws_dC + ((int64_t)b * p.ngroups + g) * L * N + n0;
  const int st_o = st_state + st_step * N;
  int lane8 = lane & 8;

  float accD = 0.f, accBias = 0.f;

  // this thread's element of a half's B / C tiles: row tid / NP of the half, state tid % NP
  const int bc_row = tid / NP, bc_n = tid % NP;
  const bool bc_ok = bc_n < N;
  const float *Bb = p.Bm + b * p.s.B_sb, *Cb = p.Cm + b * p.s.C_sb;
  const int B_o = (bc_ok ? bc_n : N - 1) * B_sn, C_o = (bc_ok ? bc_n : N - 1) * C_sn;
  auto stage_bc = [&](float bv, float cv) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int q = ((bc_n ^ v) + 16 * v) & 63;
      s_bc[0][v][bc_row][q] = bv;
      s_bc[1][v][bc_row][q] = cv;
    }
  };
  // this lane's two float4 of a row: variant hm & 3, halves swapped by bit 2 of hm
  typedef const __attribute__((address_space(3))) float *lds_cfp;
  lds_cfp lB0 = (lds_cfp)&s_bc[0][hm & 3][0][((n0 + 16 * (hm & 3)) & 63) + 4 * (hm >> 2)];
  lds_cfp lB1 = (lds_cfp)&s_bc[0][hm & 3][0][((n0 + 16 * (hm & 3)) & 63) + 4 * (1 - (hm >> 2))];
  asm volatile("" : "+v"(lB0), "+v"(lB1));   // (kept in vector registers; the per-step offsets are immediates)
  constexpr int kCoff = 4 * SUB * 64;          // s_bc[1] - s_bc[0] in floats

  // raw (t, d) rows of the NEXT chunk to be processed and its two B / C elements, kept as loaded (a conversion beside the
  // load makes the wave wait for the row right where it asks for it; so does packing two 16-bit loads into one register).
  // They are requested in front of the first half's walk -- not a whole chunk ahead in phase A: the second half's walk
  // is the kernel's register peak (both entering states are still live there), and rows requested after it are dead
  // again (phase A turns them into LDS records) before the next one
  TIO ru[K], rdl[K], rz[K], rdo[K], ry[K];
  float rb[2], rc[2];
  auto load_rows = [&](int c) {
    const int t0 = c * TB, tlast = L - 1 - t0;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int tl = hf * SUB + bc_row;
      const int t = t0 + (tl <= tlast ? tl : tlast);
      const float bvv = Bb[B_o + t * B_sl], cvv = Cb[C_o + t * C_sl];
      rb[hf] = bc_ok ? bvv : 0.f;
      rc[hf] = bc_ok ? cvv : 0.f;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int tl = w + k * NW;
      const int tc = t0 + (tl <= tlast ? tl : tlast);  // clamped address, masked value
      ru[k] = up[u_o + tc * u_sl];
      rdl[k] = dtp[dt_o + tc * dt_sl];
      rz[k] = zp[z_o + tc * z_sl];
      rdo[k] = dop[o_o + tc * o_sl];
      if constexpr (YIN) ry[k] = yip[o_o + tc * o_sl];
    }
  };
  // checkpoint of this wave's slice, straight into slot order: slot k = row k ^ hm of the (b, c, h, w, g) block, column
  // pi(lane) (scan_common.h, wide layout) -- eight 4-byte loads, each 8 whole 32-byte sectors per wave; a wave-uniform
  // base + one 32-bit offset per lane, the row selected by ONE xor (row and column bits do not overlap)
  const int ck_o = hm * 64 + ckpt_pi(lane);
  auto ckpt_load_slots = [&](int c, int h, f2 (&x)[NP2]) {
    const float *q = p.ckpt_in + ckpt_wide_block(b, nchunks, c, h, NW, w, Dm, g);
#pragma unroll
    for (int k = 0; k < NS; ++k) x[k / 2][k % 2] = q[ck_o ^ (k * 64)];
  };
  load_rows(nchunks - 1);
#ifdef CUM_SCAN_PROBE
  float ph[12] = {};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif

  for (int c = nchunks - 1; c >= 0; --c) {
    const int t0 = c * TB;
    const int tlast = L - 1 - t0;  // last valid local step of this chunk (>= 0)
    // state entering the chunk: needed by both halves, requested now so that its latency hides behind phase A
    f2 x0[NP2], x8[NP2];   // x8: state entering the second half (local step 8), read only if the chunk reaches it
    // (requesting the first half's state only behind the second half's walk would free eight registers there; measured
    //  at compile time it costs the allocator more than it frees: 52 spilled registers against 9)
    const bool two = tlast >= SUB;
    ckpt_load_slots(c, 0, x0);
    ckpt_load_slots(c, two ? 1 : 0, x8);
    // ---- phase A: per-(t, d) quantities, once, into LDS
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int tl = w + k * NW;
      const bool ok = dok && tl < TB && tl <= tlast;
      const float uv = (float)ru[k], dv = (float)rdl[k], zv = (float)rz[k], dov = (float)rdo[k];
      const float pre = dv + bias;
      float dtv = pre, sg = 1.f;
      if (softplus) {
        dtv = softplus20(pre);
        sg = pre <= 20.f ? sigmoidf_(pre) : 1.f;
      }
      dtv = ok ? dtv : 0.f;
      float dy = ok ? dov : 0.f, dzf = 0.f;
      if (has_z) {
        const float sz = sigmoidf_(zv);
        dy *= zv * sz;
        dzf = dov * sz * (1.f + zv * (1.f - sz));
      }
      if (tl < TB) {
        s_op[tl][lane] = make_float4(dtv, ok ? dtv * uv : 0.f, ok ? dy : 0.f, uv);
        s_fin[tl][lane] = make_float4(dzf, sg, YIN ? (float)ry[k] : 0.f, 0.f);
      }
    }
    // B / C tiles of the half walked first: the second one if the chunk reaches it (rows past the clip's end hold the last
    // row's values and meet zero dt / du / dy); the first half's element waits in two registers
    stage_bc(two ? rb[1] : rb[0], two ? rc[1] : rc[0]);
    const float hb0 = rb[0], hc0 = rc[0];
    PROBE(0);
    __syncthreads();
    PROBE(1);

    f2 xs[SUB][NP2];   // states before each step of the half being processed
    // Operands of one time step: B_t / C_t in slot order and the per-(t, d) values, all from LDS.  They are
    // fetched one step ahead of their use so that the LDS latency is not exposed.
    struct StepOps {
      f2 bv[NP2], cv[NP2];
      float dt, du, dy;
    };
    auto fetch = [&](int tl, StepOps &o) {
      typedef float f4v __attribute__((ext_vector_type(4)));
      typedef const __attribute__((address_space(3))) f4v *lds_c4p;
      const int r = (tl % SUB) * 64;     // row of the half's tile (compile-time: an immediate offset)
      const f4v b0 = *(lds_c4p)(lB0 + r), b1 = *(lds_c4p)(lB1 + r);
      const f4v c0 = *(lds_c4p)(lB0 + r + kCoff), c1 = *(lds_c4p)(lB1 + r + kCoff);
      o.bv[0] = f2{b0.x, b0.y}; o.bv[1] = f2{b0.z, b0.w}; o.bv[2] = f2{b1.x, b1.y}; o.bv[3] = f2{b1.z, b1.w};
      o.cv[0] = f2{c0.x, c0.y}; o.cv[1] = f2{c0.z, c0.w}; o.cv[2] = f2{c1.x, c1.y}; o.cv[3] = f2{c1.z, c1.w};
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
    // one reverse step; xp = state before step tl; slot = tl % SUB; returns the step's row-level dB / dC totals
    auto rev_step = [&](const f2 (&xp)[NP2], int slot, const StepOps &o) -> float {
      const float dt = o.dt, du = o.du, dy = o.dy;
      // decay factors: parked in LDS by the recomputed step (slots < NA), requested at the top of their OWN step -- a step
      // ahead they would hold eight more registers across the previous step's peak -- and first used behind dx and the
      // sixteen products
      f2 ap[NP2];
      if (slot < NA) {
        const float4 a0 = s_a[slot][0][tid], a1 = s_a[slot][1][tid];
        ap[0] = f2{a0.x, a0.y}; ap[1] = f2{a0.z, a0.w}; ap[2] = f2{a1.x, a1.y}; ap[3] = f2{a1.z, a1.w};
      }
      f2 p1, p2, yp = {0.f, 0.f};   // even / odd states summed apart, joined in the reduction's hazard slots
      float rdx[NS], rxt[NS];
#pragma unroll
      for (int j = 0; j < NP2; ++j) {
        const f2 a = slot < NA ? ap[j] : exp2_2(dt * Ap[j]);
        // the state after this step is the saved state before the next one (recomputed only for the half's last step)
        const f2 xt = slot + 1 < SUB ? xs[slot + 1][j] : a * xp[j] + du * o.bv[j];
        const f2 dx = o.cv[j] * dy + dxc[j];
        if constexpr (!YIN) yp = o.cv[j] * xt + yp;
        rdx[2 * j] = dx.x * du; rdx[2 * j + 1] = dx.y * du;      // (scalar multiplies on purpose: see xor_reduce16)
        rxt[2 * j] = xt.x; rxt[2 * j + 1] = xt.y;
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
      float X, q0, q1;
      xor_reduce16(rdx, rxt, dy, p1.x, p1.y, p2.x, p2.y, X, q0, q1);
      s_p12[w][slot][lane] = make_float2(q0, q1);
      if constexpr (!YIN) s_y[w][slot][lane] = yp.x + yp.y;
      __builtin_amdgcn_sched_barrier(0);
      return X;
    };
    // the reverse walk of one half (cur holds the operands of its last step): after every second step the two steps'
    // totals meet across the lane halves, after every fourth across the rows -> one finished register per four steps
    auto rev_walk = [&](int base, StepOps &cur, StepOps &nxt) {
      float Xo = 0.f, Yo = 0.f;
#pragma unroll
      for (int s = SUB - 1; s >= 0; --s) {
        if (s > 0) fetch(base + s - 1, nxt);
        const float X = rev_step(xs[s], s, cur);
        if (s > 0) cur = nxt;
        if (s & 1) {
          Xo = X;
        } else {
          const float Y = swap32_add(Xo, X);       // lanes 0-31: step s + 1, lanes 32-63: step s
          if (s & 2) {
            Yo = Y;
          } else {
            const float Z = swap16_add(Yo, Y);     // rows 0..3: steps s + 3, s + 1, s + 2, s
            asm volatile("" : "+v"(lane8));        // (keeps the select below at the store)
            float *const slab = lane8 ? sC : sB;
            if (base + s + st_lim <= tlast) slab[st_o + (t0 + base + s) * N] = Z;
          }
        }
      }
    };
    // phase C for one half: combine the per-wave partial sums, write du / ddelta / dz
    auto finish_half = [&](int half) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int tl = w + k * NW;
        const int slot = tl - half * SUB;
        if (tl < TB && slot >= 0 && slot < SUB && tl <= tlast && dok) {
          const int t = t0 + tl;
          const float4 op = s_op[tl][lane], fin = s_fin[tl][lane];   // {delta', -, dy, u}, {dout dsilu, dsoftplus, y, -}
          float q1 = 0.f, q2 = 0.f, y = YIN ? fin.z : Dd * op.w;
#pragma unroll
          for (int ww = 0; ww < NW; ++ww) {
            const float2 q = s_p12[ww][slot][lane];
            q1 += q.x;
            q2 += q.y;
            if constexpr (!YIN) y += s_y[ww][slot][lane];
          }
          if (has_z) dzp[dz_o + t * dz_sl] = (TIO)(fin.x * y);
          const float ddt = kLn2 * q1 + op.w * q2;  // d loss / d delta'
          const float dpre = ddt * fin.y;
          dup[du_o + t * du_sl] = (TIO)fmaf(op.z, Dd, op.x * q2);
          ddtp[dd_o + t * dd_sl] = (TIO)dpre;
          accD = fmaf(op.z, op.w, accD);
          accBias += dpre;
        }
      }
    };

    StepOps cur, nxt;
    // ---- second half (local steps 8..15), only if the chunk reaches it
    if (two) {
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
      rev_walk(SUB, cur, nxt);                 // reverse 15..8 (cur holds step 15's operands)
      PROBE(3);
      __syncthreads();
      PROBE(4);
      stage_bc(hb0, hc0);                      // the first half's B / C tiles (every reader of the second's is past the barrier)
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
      // the next chunk's rows (see the declaration of ru); behind the recompute, whose first step waited for this chunk's
      // entering state: a request in front of it would put these loads under that wait
      if (c > 0) load_rows(c - 1);
      rev_walk(0, cur, nxt);
      PROBE(8);
      __syncthreads();
      PROBE(9);
      finish_half(0);
      PROBE(10);
    }
    // The next chunk's phase A writes only s_op / s_bc (their readers finished before
    // the last barrier) and its first rev_step runs after that phase's barrier, which
    // every wave reaches only after this finish_half.
  }
  // ---- per-(b, d, n) dA and per-(b, d) dD / dbias slabs
  if (dok) {
    float *wa = p.ws_dA + ((int64_t)b * Dm + d) * N + n0;
#pragma unroll
    for (int k = 0; k < NS; ++k)
      if ((k ^ hm) < nvalid) wa[k ^ hm] = dAacc[k / 2][k % 2];
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
    if (lane == 0 && dst) *dst = (o < nA && (p.s.delta_softplus & kScanAIsLog)) ? v * scan_A(p, o) : v;   // dA_log = dA * A
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
      dA[i] = (p.s.delta_softplus & kScanAIsLog) ? s * scan_A(p, i) : s;      // A given as A_log: dA_log = dA * A
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
  if (p.s.dstate == NS * NW && p.ypre_in && p.z)     // (y only enters dz; shapes with a ragged last wave rebuild it)
    hipLaunchKernelGGL((scan_bwd_kernel<NW, TIO, true, true>), grid, block, 0, st, p);
  else if (p.s.dstate == NS * NW)
    hipLaunchKernelGGL((scan_bwd_kernel<NW, TIO, true>), grid, block, 0, st, p);
  else
    hipLaunchKernelGGL((scan_bwd_kernel<NW, TIO, false>), grid, block, 0, st, p);
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
  if (s->dstate > 2 * NS) {
    // scan_bwd_kernel addresses a clip's u / delta / z / dout / du / ddelta / dz through a wave-uniform base + ONE 32-bit
    // byte offset per lane: the clip's extent (channel and time strides) has to fit
    const int64_t lim = (1LL << 29), Lm = s->len > 0 ? s->len - 1 : 0, Dl = s->dim - 1;
    CUM_REQUIRE(s->u_sd >= 0 && s->dt_sd >= 0 && s->z_sd >= 0 && s->o_sd >= 0 && gs->du_sd >= 0 && gs->dd_sd >= 0 &&
                    gs->dz_sd >= 0,
                "scan_bwd: negative channel strides are not supported");
    CUM_REQUIRE(Dl * s->u_sd + Lm * s->u_sl < lim && Dl * s->dt_sd + Lm * s->dt_sl < lim &&
                    Dl * s->z_sd + Lm * s->z_sl < lim && Dl * s->o_sd + Lm * s->o_sl < lim &&
                    Dl * gs->du_sd + Lm * gs->du_sl < lim && Dl * gs->dd_sd + Lm * gs->dd_sl < lim &&
                    Dl * gs->dz_sd + Lm * gs->dz_sl < lim && (int64_t)s->dim * NS < lim,
                "scan_bwd: one clip's tensors must span fewer than 2^29 elements (d_state > 16)");
  }
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
#ifndef CUM_ONLY8   // (compile-time experiments on the d_state-64 kernel alone)
    case 3: rc = launch_bwd<3>(p, st); break;
    case 4: rc = launch_bwd<4>(p, st); break;
    case 5: rc = launch_bwd<5>(p, st); break;
    case 6: rc = launch_bwd<6>(p, st); break;
    case 7: rc = launch_bwd<7>(p, st); break;
#endif
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
