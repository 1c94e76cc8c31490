// The streaming hop as ONE launch ("persistent kernels ... one launch per frame").
//
// Replaces, for models whose weights are small enough to be re-read from L2 by every stream (every shipped pruned
// checkpoint and the 442K model), the whole of CleanUMamba.feed / _denoise_frame
// (src/network/CleanUMamba.py:370-490): running input std, the E incremental encoder layers, tsfm_conv1, the Mamba
// blocks' single-token step (Block.forward + Mamba.step), norm_f, tsfm_conv2, the E decoder layers with their
// overlap-add tails and skip connections, and the output scaling -- 68 launches per hop on the per-layer path.
//
// A 512-thread workgroup owns ONE stream for the whole launch (and for `n_hops` consecutive hops of it: streams are
// the parallel axis, time is walked inside the kernel).  Activations of a hop never leave the CU: LDS regions for
// layer input / hidden / transposed-conv taps / the bottleneck's vectors.  Stream state lives in one f32 block per
// stream in HBM:
//   * ring_i  [3 n_i][ld_i]  outputs of encoder layer i, n_i = hop >> (i + 1) new rows per hop.  The window the
//     reference keeps per layer (3 n_i - 2 rows) is a delay line: a hop appends n_i rows, the decoder's skip reads the
//     n_i oldest ones -- written two hops earlier.  Rows sit at (absolute row) mod 3 n_i: nothing is ever shifted.
//   * tail_j  [2][cq_j]      overhang of decoder layer j's transposed conv (bias excluded), as the per-layer path keeps it
//   * conv_state / ssm_state of every Mamba block, the running input std, its frame count, the ring phase.
//
// The kernel is an INTERPRETER of a short op list the host compiles from the model (cleanumamba_amd/network/hopplan.py:
// 65 ops for an E8 model with 3 blocks; read by scalar loads): every op names its LDS places and its
// offsets into the weight blob / the state block, ends with a workgroup barrier, and is one of: input std, first
// encoder conv (1 input channel: VALU), matrix product (below; an encoder layer's output product also appends its rows
// to the layer's ring and fetches the two carry rows), add + LayerNorm, Mamba conv step, Mamba state update, decoder
// overlap-add.  The products themselves are compiled further, into per-wave STAGE LISTS (see hop_gemm).  Why a table and not straight-line code: the first version inlined a
// specialised matrix product per layer (115 KB of code walked once per hop: instruction-fetch bound, 370 us per hop);
// the second called one shared body per layer (arguments through scratch memory, generic pointers -> flat loads and
// vmcnt(0) in front of every MFMA: 10 k cycles of fixed cost per product).  Here each product body exists once, its
// operands sit in scalar registers and LDS / global address spaces are known to the compiler.
//
// Matrix products run on v_mfma_f32_16x16x4_f32 (exact f32: a k-ordered fma chain) with the WEIGHTS as the A operand,
// packed by the host in fragment order ([tile][16-deep k chunk][lane][4]: one coalesced 1 KiB load per wave and chunk;
// a wave has the loads of two stages of up to 4 chunks in flight), and the activations as the B operand straight out of LDS
// (ds_read_b128); a lane then holds 4 consecutive output channels of one row, so GLU pairs (the two 16-row tiles of a
// pair share their B fragments) and bias / ReLU / sigmoid epilogues are lane-local.  All k extents are multiples of 16
// and the packed weights are zero-padded, so padding columns come out as exact zeros and need no guards.
#include "common.h"

namespace cum {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kHopThreads = 512, kHopWaves = 8;
constexpr int kHopMaxOps = 160, kHopOpInts = 24, kHopHdrInts = 16, kHopMaxStages = 8192;
constexpr int kHopMagic = 0x486f7033;      // "Hop3"

enum {
  kOpEnd = 0,      // -
  kOpStd = 1,      // -, std_off
  kOpEnc0 = 2,     // n, ld_h, w1, b1, x (LDS), h (LDS), ph
  kOpGemm = 3,     // HopG (18 ints), nacc, ks, kcs, mt (16-row tiles per item: 1, 2 or 4), ring (state offset | -1); its stages: wtab / stages
  kOpRing = 4,     // n, ldo, ring, src (LDS), po, carry (LDS)   (not emitted any more: merged into the product in front)
  kOpLn = 5,       // hs, res, out (LDS), w, b, eps bits, dm, dmp, has_res
  kOpConvStep = 6, // di, dip, W, conv_state, conv_w, conv_b, xz (LDS), x (LDS)
  kOpSsm = 7,      // di, dip, N, ssm_state, A, D, dt, x, Bv, Cv, z, y (LDS)
  kOpOverlap = 8,  // L, ldo, cq, cout, y (LDS), ldy, b2, tail, skip_ring, skip_ld, skip_n, relu, last, dst (LDS), po
};

// int32 header + ops; the host fills it as a flat int32 array (cum_stream_hop_plan_ints()).
struct HopPlan {
  int32_t magic, n_ops, frame_len, hop_len, lds_floats, phase_off, ops_lds, n_stages, pad[kHopHdrInts - 8];
  int32_t ops[kHopMaxOps * kHopOpInts];
  int32_t wtab[kHopMaxOps * kHopWaves];      // per op and wave: first stage | stage count << 16
  int32_t stages[kHopMaxStages * 4];         // {weight offset, operand offset, nb | first << 3 | last << 4, out}
};

// -DCUM_HOP_PROBE (tools/hop_phase_probe.py): s_memtime stamp of workgroup 0 after every op of its last hop
#ifdef CUM_HOP_PROBE
__device__ unsigned long long hop_stamps[kHopMaxOps + 1];
#define HOP_STAMP(i)                                                                                            \
  do {                                                                                                          \
    if (threadIdx.x == 0 && blockIdx.x == 0 && hop == n_hops - 1) hop_stamps[(i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#ifndef CUM_HOP_PROBE_PC
#define CUM_HOP_PROBE_PC 68
#endif
__device__ int hop_probe_on;
#define HOP_FINE(i)                                                                                       \
  do {                                                                                                    \
    if (threadIdx.x == 0 && blockIdx.x == 0 && hop_probe_on) hop_stamps[120 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define HOP_STAMP(i) \
  do {               \
  } while (0)
#define HOP_FINE(i) \
  do {              \
  } while (0)
#endif

extern __shared__ __attribute__((aligned(16))) float hop_lds[];
typedef const __attribute__((address_space(1))) f4 *hop_gf4;      // global (not flat) 16-byte loads
typedef const __attribute__((address_space(4))) int *hop_cint;    // constant: uniform addresses become scalar loads

__device__ __forceinline__ f4 hop_ld4(const float *p) { return *reinterpret_cast<const f4 *>(p); }
__device__ __forceinline__ void hop_st4(float *p, f4 v) { *reinterpret_cast<f4 *>(p) = v; }
__device__ __forceinline__ f4 hop_gld4(const float *p) { return *(hop_gf4)p; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding GLOBAL access
// (s_waitcnt vmcnt(0)), i.e. parks each of a hop's ~75 ops behind its non-temporal state stores.  Ops hand their results
// to each other through LDS; the stream state in HBM is re-read either by the thread that wrote it or one hop later,
// behind the full barrier that ends a hop (the running std, read by all and written by one, keeps a full barrier too).
__device__ __forceinline__ void hop_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float hop_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over the workgroup; every thread gets it.  `red` holds kHopWaves floats.
__device__ __forceinline__ float hop_block_sum(float v, float *red, int wave, int lane) {
  v = hop_wave_sum(v);
  hop_barrier();
  if (lane == 0) red[wave] = v;
  hop_barrier();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < kHopWaves; ++w) t += red[w];
  return t;
}

// One matrix product.  LDS places are float offsets into hop_lds, weight / bias places float offsets into the weight
// blob (-1: none).
//   Out[n][m] = sum_k W[n][k] X[m][k] for m < M, n < 16 * ntg (per accumulator set).  W: packed tiles
//   [ntg * NACC][kcn][64 lanes][4]; X: LDS, row m at x + m * xs; k-chunk kc (16 floats) of a row sits at
//   (kc / kpr) * seg + (kc % kpr) * 16 (a strided conv reads its 4 input rows as 4 segments of kpr chunks; seg = the row
//   pitch).  Row pitches are 4 (mod 8) floats so that the 16 rows of a ds_read_b128 fall on different banks.
// Epilogue of a quad of 4 consecutive output channels n0..n0+3 of row m (n0 < nlimit):
//   v = acc + bias;  GLU (NACC = 2): v *= sigmoid(acc2 + bias2)  |  ReLU  |  softplus;  v += add (LDS);
//   dst[(m + row_off) * pitch + n0] = v
// Work items (tile group, group of MT 16-row tiles, k slice) are dealt round-robin to the waves; when the tiles alone
// would leave waves idle the k range is split (ks slices), the partial accumulators go through the scratch and are
// summed in slice order (bit-reproducible) by all threads -- a hop's deep layers are few-row products whose time is the
// latency of streaming their weights from L2.
enum { kActNone = 0, kActRelu = 1, kActSoftplus = 2 };
struct HopG {
  int w, x, scratch, ntg, kcn, xs, kpr, seg, M, scratch_floats;
  int dst, bias, bias2, add, pitch, row_off, act, nlimit;
  // an encoder layer's output also goes to its ring in the stream state (rows (rbase + m) mod n3, 16 ntg floats each)
  float *ringp;
  int rbase, n3;
};

__device__ __forceinline__ int hop_wrap(int a, int n) {      // a mod n for 0 <= a < 3 n
  a -= a >= n ? n : 0;
  return a - (a >= n ? n : 0);
}

template <int NACC>
__device__ __forceinline__ void hop_epi(const HopG &g, int n0, int m, f4 v0, f4 v1, f4 b0, f4 b1) {
  if (n0 >= g.nlimit) return;
  v0 += b0;
  if (NACC > 1) {
    v1 += b1;
    v0 *= f4{sigmoidf_(v1[0]), sigmoidf_(v1[1]), sigmoidf_(v1[2]), sigmoidf_(v1[3])};
  } else if (g.act == kActRelu) {
    v0 = f4{fmaxf(v0[0], 0.f), fmaxf(v0[1], 0.f), fmaxf(v0[2], 0.f), fmaxf(v0[3], 0.f)};
  } else if (g.act == kActSoftplus) {
    v0 = f4{softplus20(v0[0]), softplus20(v0[1]), softplus20(v0[2]), softplus20(v0[3])};
  }
  if (g.add >= 0) v0 += hop_ld4(hop_lds + g.add + n0);
  hop_st4(hop_lds + g.dst + (m + g.row_off) * g.pitch + n0, v0);
  if (g.ringp) __builtin_nontemporal_store(v0, reinterpret_cast<f4 *>(g.ringp + hop_wrap(g.rbase + m, g.n3) * (16 * g.ntg) + n0));
}

// floor(a / d) for small non-negative a (< 2^15) and d (<= 2^10) without the ~40-instruction integer division: inv = 1 / d
__device__ __forceinline__ int hop_fdiv(int a, float inv) { return (int)(((float)a + 0.5f) * inv); }

// A wave's work on a product is a flat list of STAGES the HOST compiled (hopplan.py::_stages): up to 4 consecutive
// 16-deep k chunks of one work item (tile group, group of MT 16-row tiles, k slice) that are consecutive in the weight
// blob AND in the LDS operand (a stage never straddles two input rows of a strided conv), with the blob offset of its
// first fragment, the LDS offset of its first operand chunk, its chunk count, whether it opens an item (accumulators
// start at zero) or closes one (epilogue, or the partial sums of a k slice into the scratch) and where the result goes.
// The wave fetches ITS list for the op with one 16-byte load per lane (lane i = stage i, <= 64 stages) -- requested one
// op ahead -- and reads a stage's four words with v_readlane; the fragment loads of stage s + 1 are issued before the
// MFMAs of stage s (two register sets).  History: the cursor that walked items and chunks in the kernel cost ~380
// instructions per op and wave, and that path length -- not the weights' latency, not the MFMAs of 16-row tiles that
// hold 1-8 real rows -- was the deep layers' time.
typedef int i4 __attribute__((ext_vector_type(4)));
struct HopStage {
  int woff, xoff, meta, out;
};
__device__ __forceinline__ HopStage hop_stage(const i4 &v, int i) {
  HopStage t;
  t.woff = __builtin_amdgcn_readlane(v[0], i);
  t.xoff = __builtin_amdgcn_readlane(v[1], i);
  t.meta = __builtin_amdgcn_readlane(v[2], i);
  t.out = __builtin_amdgcn_readlane(v[3], i);
  return t;
}

template <int NACC, int MT>
__device__ __forceinline__ void hop_gemm(const float *__restrict__ wb, const HopG &g, int ks, const i4 &dsc, int count,
                                         int tid, int lane) {
  constexpr int D = 4;
  const int lr = lane & 15, lg = lane >> 4;
  const int ntg = g.ntg, M = g.M;
  const int mgs = (M + 16 * MT - 1) / (16 * MT);
  const int base = ntg * mgs;
  constexpr int blk = NACC * MT * 256;
  const float *xlane = hop_lds + lr * g.xs + 4 * lg;
  const hop_gf4 wlane = (hop_gf4)wb + lane;
  const int w2 = g.kcn * 64;                       // second accumulator's fragments, in 16-byte units
  const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
  f4 acc[NACC][MT];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[a][mt] = zero;

  // fragment (and, for a closing stage of an unsplit product, bias) loads of a stage
  auto load = [&](f4 (&wv)[D][NACC], f4 &b0, f4 &b1, const HopStage &t) {
    const hop_gf4 wp = wlane + (t.woff >> 2);
    const int nb = t.meta & 7;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (d < nb) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) wv[d][a] = wp[a * w2 + d * 64];
      }
    }
    if ((t.meta & 16) && ks == 1) {
      const int n0 = (t.out & 0xffff) + 4 * lg;
      if (g.bias >= 0) b0 = hop_gld4(wb + g.bias + n0);
      if (NACC > 1) b1 = hop_gld4(wb + g.bias2 + n0);
    }
  };
  auto compute = [&](const f4 (&wv)[D][NACC], const f4 &b0, const f4 &b1, const HopStage &t) {
    const int nb = t.meta & 7;
    if (t.meta & 8) {
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[a][mt] = zero;
    }
    const float *xp = xlane + t.xoff;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (d < nb) {
        f4 xv[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xv[mt] = hop_ld4(xp + mt * 16 * g.xs + d * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int a = 0; a < NACC; ++a)
              acc[a][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[d][a][j], xv[mt][j], acc[a][mt], 0, 0, 0);
      }
    }
    if (t.meta & 16) {              // last stage of the item
      if (ks == 1) {
        const int n0 = t.out & 0xffff, m0 = t.out >> 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int m = m0 + mt * 16 + lr;
          if (m < M) hop_epi<NACC>(g, n0 + 4 * lg, m, acc[0][mt], acc[NACC - 1][mt], b0, b1);
        }
      } else {
        float *P = hop_lds + t.out;
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) hop_st4(P + ((a * MT + mt) * 16 + lr) * 16 + 4 * lg, acc[a][mt]);
      }
    }
  };
  if (count > 0) {
    f4 wA[D][NACC], wB[D][NACC], bA0 = zero, bA1 = zero, bB0 = zero, bB1 = zero;
    HopStage tA = hop_stage(dsc, 0), tB;
    HOP_FINE(0);
    load(wA, bA0, bA1, tA);
    HOP_FINE(1);
    int fine = 2;
    for (int i = 0;;) {
      if (i + 1 < count) {
        tB = hop_stage(dsc, i + 1);
        load(wB, bB0, bB1, tB);
      }
      HOP_FINE(fine);
      ++fine;
      compute(wA, bA0, bA1, tA);
      HOP_FINE(fine);
      ++fine;
      if (i + 1 >= count) break;
      if (i + 2 < count) {
        tA = hop_stage(dsc, i + 2);
        load(wA, bA0, bA1, tA);
      }
      HOP_FINE(fine);
      ++fine;
      compute(wB, bB0, bB1, tB);
      HOP_FINE(fine);
      ++fine;
      i += 2;
      if (i >= count) break;
    }
    HOP_FINE(fine);
  }
  if (ks > 1) {
    hop_barrier();
    HOP_FINE(30);
    const float inv_ntg = 1.f / (float)ntg;
    for (int u = tid; u < base * MT * 64; u += kHopThreads) {
      const int q = u & 3, ml = (u >> 2) % (MT * 16), b = u / (MT * 64);
      const int bm = hop_fdiv(b, inv_ntg), tg = b - bm * ntg, m = bm * (16 * MT) + ml;
      const int n0 = tg * 16 + 4 * q;
      const f4 bq = hop_gld4(wb + max(g.bias, 0) + n0);
      const f4 b0 = g.bias >= 0 ? bq : zero;
      f4 b1 = zero;
      if (NACC > 1) b1 = hop_gld4(wb + g.bias2 + n0);
      if (m < M) {
        f4 v0 = zero, v1 = zero;
        for (int sl = 0; sl < ks; ++sl) {
          const float *P = hop_lds + g.scratch + (sl * base + b) * blk + ml * 16 + 4 * q;
          v0 += hop_ld4(P);
          if (NACC > 1) v1 += hop_ld4(P + MT * 256);
        }
        hop_epi<NACC>(g, n0, m, v0, v1, b0, b1);
      }
    }
  }
}

__global__ __launch_bounds__(kHopThreads) void stream_hop_kernel(const HopPlan *__restrict__ plan,
                                                                  const float *__restrict__ w, float *state,
                                                                  int64_t state_stride, const float *__restrict__ in,
                                                                  int64_t in_stride, float *__restrict__ out,
                                                                  int64_t out_stride, int n_hops) {
  float *lds = hop_lds;
  __shared__ float red[kHopWaves];
  const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
  float *st = state + (int64_t)blockIdx.x * state_stride;
  const int n_ops = plan->n_ops, frame_len = plan->frame_len, hop_len = plan->hop_len, phase_off = plan->phase_off;
  // ops are read with SCALAR loads from the plan (constant address space: s_load_dwordx8 / x16 straight into scalar
  // registers, served by the scalar cache all workgroups share) -- through LDS they cost 24 v_readfirstlane per op
  const hop_cint ops = (hop_cint)(uintptr_t)plan->ops;
  const hop_cint wtab = (hop_cint)(uintptr_t)plan->wtab;
  const __attribute__((address_space(1))) i4 *stages = (const __attribute__((address_space(1))) i4 *)plan->stages;
  // this wave's stage list of the op about to run (lane i = stage i), requested while the op in front of it ran
  i4 dsc = i4{0, 0, 0, 0};
  int dcount = 0;
  {
    const int wt = wtab[wave];
    dcount = wt >> 16;
    dsc = stages[(wt & 0xffff) + min(lane, max(dcount - 1, 0))];
  }

  for (int hop = 0; hop < n_hops; ++hop) {
    const float *frame = in + (int64_t)blockIdx.x * in_stride + (int64_t)hop * hop_len;
    float *o = out + (int64_t)blockIdx.x * out_stride + (int64_t)hop * hop_len;
    const int phase = (int)st[phase_off];
    float stdv = 1.f;
    HOP_STAMP(0);

    for (int pc = 0; pc < n_ops; ++pc) {
#ifdef CUM_HOP_PROBE
      if (tid == 0 && blockIdx.x == 0) hop_probe_on = (pc == CUM_HOP_PROBE_PC && hop == n_hops - 1);   // (thread 0 reads it)
#endif
      HOP_FINE(34);
      int f[kHopOpInts];
      {
        const hop_cint src = ops + pc * kHopOpInts;
#pragma unroll
        for (int i = 0; i < kHopOpInts; ++i) f[i] = src[i];
      }
      // the next op's stage list (the next hop's first op behind the last): in flight while this op runs
      i4 ndsc;
      int ncount;
      // (no branch around the request -- other ops have an empty list and fetch stage 0: behind a branch the compiler
      //  cannot count the loads in flight and drains them all, this one included, in front of the op's first stage)
      {
        const int wt = wtab[(pc + 1 == n_ops ? 0 : pc + 1) * kHopWaves + wave];
        ncount = wt >> 16;
        ndsc = stages[(wt & 0xffff) + min(lane, max(ncount - 1, 0))];
      }
      HOP_FINE(35);
      switch (f[0]) {
        case kOpStd: {
          // running mean of the per-frame std (src/network/CleanUMamba.py:399-401), unbiased as torch.std
          const int std_off = f[2];
          float part = 0.f;
          for (int t = tid; t < frame_len; t += kHopThreads) part += frame[t];
          const float mean = hop_block_sum(part, red, wave, lane) / frame_len;
          part = 0.f;
          for (int t = tid; t < frame_len; t += kHopThreads) {
            const float d = frame[t] - mean;
            part += d * d;
          }
          const float fs = sqrtf(hop_block_sum(part, red, wave, lane) / (frame_len - 1)) + 1e-3f;
          const float cnt = st[std_off + 1] + 1.f;
          stdv = fs / cnt + (1.f - 1.f / cnt) * st[std_off];
          __syncthreads();                       // everybody has read the old values
          if (tid == 0) {
            st[std_off] = stdv;
            st[std_off + 1] = cnt;
          }
        } break;
        case kOpEnc0: {
          // first encoder conv: 1 input channel, 4 taps, ReLU
          const int n = f[1], ld_h = f[2], ph = f[7];
          const float *w1 = w + f[3], *b1 = w + f[4];
          float *X = lds + f[5], *H = lds + f[6];
          const int t_in = 2 * n + 2;
          for (int t = tid; t < t_in; t += kHopThreads) X[t] = frame[frame_len - t_in + t] / stdv;
          __syncthreads();
          const float inv_h = 1.f / (float)ld_h;
          for (int idx = tid; idx < n * ld_h; idx += kHopThreads) {
            const int m = hop_fdiv(idx, inv_h), c = idx - m * ld_h;
            float v = b1[c];
#pragma unroll
            for (int k = 0; k < 4; ++k) v = fmaf(w1[c * 4 + k], X[2 * m + k], v);
            H[m * ph + c] = fmaxf(v, 0.f);
          }
        } break;
        case kOpGemm: {
          HopG g;
          g.w = f[1], g.x = f[2], g.scratch = f[3], g.ntg = f[4], g.kcn = f[5], g.xs = f[6], g.kpr = f[7], g.seg = f[8];
          g.M = f[9], g.scratch_floats = f[10], g.dst = f[11], g.bias = f[12], g.bias2 = f[13], g.add = f[14];
          g.pitch = f[15], g.row_off = f[16], g.act = f[17], g.nlimit = f[18];
          // f[23] >= 0: the product is an encoder layer's output (rows 2.. of the next layer's input): its n = M new rows
          // also go to the layer's ring, and the two ring rows in front of them (written a hop ago) become rows 0, 1 --
          // requested here, stored behind the product.  State traffic is non-temporal: 32 streams per XCD move more
          // bytes per hop than their L2 holds, and what has to stay there is the weight blob every one of them re-reads.
          g.ringp = nullptr, g.rbase = 0, g.n3 = 1;
          f4 cv = f4{0.f, 0.f, 0.f, 0.f};
          const int lq = 4 * g.ntg;
          if (f[23] >= 0) {
            g.ringp = st + f[23];
            g.n3 = 3 * g.M;
            g.rbase = g.n3 - 2 + phase * g.M;
            if (tid < 2 * lq) {
              const int q = tid >= lq, c = (tid - q * lq) * 4;
              cv = hop_ld4(g.ringp + hop_wrap(g.rbase - 2 + q + g.n3, g.n3) * (4 * lq) + c);
            }
          }
          if (f[19] == 1) {
            if (f[22] == 1)
              hop_gemm<1, 1>(w, g, f[20], dsc, dcount, tid, lane);
            else if (f[22] == 2)
              hop_gemm<1, 2>(w, g, f[20], dsc, dcount, tid, lane);
            else
              hop_gemm<1, 4>(w, g, f[20], dsc, dcount, tid, lane);
          } else {
            if (f[22] == 1)
              hop_gemm<2, 1>(w, g, f[20], dsc, dcount, tid, lane);
            else if (f[22] == 2)
              hop_gemm<2, 2>(w, g, f[20], dsc, dcount, tid, lane);
            else
              hop_gemm<2, 4>(w, g, f[20], dsc, dcount, tid, lane);
          }
          if (f[23] >= 0 && tid < 2 * lq) {
            const int q = tid >= lq, c = (tid - q * lq) * 4;
            hop_st4(lds + g.dst + q * g.pitch + c, cv);
          }
        } break;
        case kOpRing: {
          // the n new rows -> ring; the two rows in front of them -> rows 0, 1 of the next layer's input.  State
          // traffic is non-temporal: 32 streams per XCD move more bytes per hop than their L2 holds, and what has to
          // stay there is the weight blob every one of them re-reads.
          const int n = f[1], ldo = f[2], po = f[5], lq = ldo >> 2;
          float *ring = st + f[3];
          const float *src = lds + f[4];
          float *carry = lds + f[6];
          const int n3 = 3 * n, base = n3 - 2 + phase * n;
          const float inv_q = 1.f / (float)lq;
          for (int idx = tid; idx < n * lq; idx += kHopThreads) {
            const int r = hop_fdiv(idx, inv_q), c = (idx - r * lq) * 4;
            __builtin_nontemporal_store(hop_ld4(src + r * po + c), reinterpret_cast<f4 *>(ring + hop_wrap(base + r, n3) * ldo + c));
          }
          for (int idx = tid; idx < 2 * lq; idx += kHopThreads) {
            const int q = idx >= lq, c = (idx - q * lq) * 4;
            hop_st4(carry + q * po + c, hop_ld4(ring + hop_wrap(base - 2 + q + n3, n3) * ldo + c));
          }
        } break;
        case kOpLn: {
          // res <- hs (+ res), out <- LayerNorm(res), zero padding up to dmp
          const float *hs = lds + f[1];
          float *res = lds + f[2], *outv = lds + f[3];
          const float *lw = w + f[4], *lb = w + f[5];
          const float eps = __builtin_bit_cast(float, f[6]);
          const int dm = f[7], dmp = f[8];
          const bool has_res = f[9] != 0;
          float part = 0.f;
          for (int k = tid; k < dm; k += kHopThreads) {
            const float r = hs[k] + (has_res ? res[k] : 0.f);
            res[k] = r;
            part += r;
          }
          const float mean = hop_block_sum(part, red, wave, lane) / dm;
          part = 0.f;
          for (int k = tid; k < dm; k += kHopThreads) {
            const float d = res[k] - mean;
            part += d * d;
          }
          const float rstd = rsqrtf(hop_block_sum(part, red, wave, lane) / dm + eps);
          for (int k = tid; k < dmp; k += kHopThreads) outv[k] = k < dm ? (res[k] - mean) * rstd * lw[k] + lb[k] : 0.f;
        } break;
        case kOpConvStep: {
          // causal conv update + SiLU: the state rolls left, the new sample enters on the right
          const int di = f[1], dip = f[2], W = f[3];
          float *cstate = st + f[4];
          const float *cw = w + f[5], *cb = f[6] >= 0 ? w + f[6] : nullptr;
          const float *xz = lds + f[7];
          float *xo = lds + f[8];
          for (int d = tid; d < dip; d += kHopThreads) {
            float val = 0.f;
            if (d < di) {
              float *cs = cstate + d * W;
              float acc = cb ? cb[d] : 0.f;
              for (int kk = 0; kk < W; ++kk) {
                const float v = (kk + 1 < W) ? cs[kk + 1] : xz[d];
                cs[kk] = v;
                acc = fmaf(cw[d * W + kk], v, acc);
              }
              val = acc * sigmoidf_(acc);
            }
            xo[d] = val;
          }
        } break;
        case kOpSsm: {
          // state update, D skip, silu(z) gate
          const int di = f[1], dip = f[2], N = f[3];
          float *sstate = st + f[4];
          const float *A = w + f[5], *Dv = f[6] >= 0 ? w + f[6] : nullptr;
          const float *dtv = lds + f[7], *xv = lds + f[8], *Bv = lds + f[9], *Cv = lds + f[10], *zv = lds + f[11];
          float *yv = lds + f[12];
          for (int d = tid; d < dip; d += kHopThreads) {
            float val = 0.f;
            if (d < di) {
              const float dt = dtv[d], x = xv[d];
              float *ss = sstate + d * N;
              float acc = 0.f;
              if ((N & 3) == 0) {                     // (rows of 16-byte quads: four loads in flight, not one float at a time)
                for (int n = 0; n < N; n += 4) {
                  const f4 a4 = hop_gld4(A + d * N + n), s4 = hop_ld4(ss + n);
                  f4 v4;
#pragma unroll
                  for (int j = 0; j < 4; ++j) {
                    const float a = __builtin_amdgcn_exp2f(dt * a4[j] * kLog2e);
                    v4[j] = fmaf(a, s4[j], dt * Bv[n + j] * x);
                    acc = fmaf(Cv[n + j], v4[j], acc);
                  }
                  hop_st4(ss + n, v4);
                }
              } else {
                for (int n = 0; n < N; ++n) {
                  const float a = __builtin_amdgcn_exp2f(dt * A[d * N + n] * kLog2e);
                  const float v = fmaf(a, ss[n], dt * Bv[n] * x);
                  ss[n] = v;
                  acc = fmaf(Cv[n], v, acc);
                }
              }
              const float z = zv[d];
              val = (acc + (Dv ? Dv[d] : 0.f) * x) * (z * sigmoidf_(z));
            }
            yv[d] = val;
          }
        } break;
        case kOpOverlap: {
          // overlap-add, bias, ReLU, skip; rows 2L, 2L + 1 become the tail of the next hop.  Quads of channels: the
          // padding channels below cq come out as exact zeros by themselves (zero weight rows, bias, tail, skip)
          const int L = f[1], ldo = f[2], cq = f[3], ldy = f[6], po = f[15], lq = ldo >> 2;
          const float *Y = lds + f[5], *b2 = w + f[7];
          float *tail = st + f[8];
          const float *skip = f[9] >= 0 ? st + f[9] + ((phase + 1) % 3) * f[11] * f[10] : nullptr;
          const int skip_ld = f[10];
          const bool relu = f[12] != 0, last = f[13] != 0;
          float *dst = lds + f[14];
          const float inv_q = 1.f / (float)lq;
          const int total = 2 * L * lq;
          // the global operands (skip rows, written two hops ago; bias) of up to four passes are requested before the
          // first one is used: a pass per round trip to HBM otherwise
          constexpr int kAhead = 4;
          for (int i0 = tid; i0 < total; i0 += kAhead * kHopThreads) {
            f4 sk[kAhead], bq[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
              const int idx = i0 + u * kHopThreads;
              const int r = hop_fdiv(min(idx, total - 1), inv_q), c = (min(idx, total - 1) - r * lq) * 4;
              sk[u] = f4{0.f, 0.f, 0.f, 0.f};
              bq[u] = sk[u];
              if (idx < total && c < cq) {
                bq[u] = hop_gld4(b2 + c);
                if (skip) sk[u] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(skip + r * skip_ld + c));
              }
            }
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
              const int idx = i0 + u * kHopThreads;
              if (idx >= total) break;
              const int r = hop_fdiv(idx, inv_q), c = (idx - r * lq) * 4;
              f4 v = f4{0.f, 0.f, 0.f, 0.f};
              if (c < cq) {
                const int t = r >> 1, q = r & 1;
                v = hop_ld4(Y + t * ldy + q * cq + c) + bq[u];
                if (t > 0) {
                  v += hop_ld4(Y + (t - 1) * ldy + (q + 2) * cq + c);
                } else {
                  v += hop_ld4(tail + q * cq + c);
                  hop_st4(tail + q * cq + c, hop_ld4(Y + (L - 1) * ldy + (q + 2) * cq + c));
                }
                if (relu) v = f4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                v += sk[u];
              }
              if (!last)
                hop_st4(dst + r * po + c, v);
              else if (c == 0)
                o[r] = v[0] * stdv;
            }
          }
        } break;
        default:
          break;
      }
      HOP_FINE(36);
      hop_barrier();
      HOP_FINE(37);
      HOP_STAMP(pc + 1);
      dsc = ndsc;
      dcount = ncount;
    }
    if (tid == 0) st[phase_off] = (float)((phase + 1) % 3);
    __syncthreads();
  }
}

}  // namespace cum

using namespace cum;

#ifdef CUM_HOP_PROBE
extern "C" int cum_stream_hop_probe_read(unsigned long long *host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(hop_stamps),
                                  sizeof(unsigned long long) * (n < kHopMaxOps + 1 ? n : kHopMaxOps + 1));
}
#endif

extern "C" int cum_stream_hop_plan_ints(void) { return (int)(sizeof(HopPlan) / sizeof(int32_t)); }

extern "C" int cum_stream_hop_max_lds_bytes(void) { return 160 * 1024 - (int)(kHopWaves * sizeof(float)) - 256; }

extern "C" int cum_stream_hop(const void *plan, const float *weights, float *state, int64_t state_stride,
                              int32_t streams, const float *in, int64_t in_stride, float *out, int64_t out_stride,
                              int32_t n_hops, int32_t lds_bytes, void *stream) {
  CUM_REQUIRE(streams >= 0 && n_hops >= 0 && state_stride > 0, "stream_hop: bad shape");
  CUM_REQUIRE(lds_bytes > 0 && lds_bytes <= cum_stream_hop_max_lds_bytes() && lds_bytes % 16 == 0,
              "stream_hop: LDS size outside the kernel's limit (cum_stream_hop_max_lds_bytes)");
  if (streams == 0 || n_hops == 0) return CUM_OK;
  CUM_REQUIRE(plan && weights && state && in && out, "stream_hop: null pointer");
  CUM_REQUIRE(((uintptr_t)weights & 15) == 0 && ((uintptr_t)state & 15) == 0, "stream_hop: weights / state must be 16-byte aligned");
  hipError_t e = hipFuncSetAttribute((const void *)stream_hop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) {
    cum_set_error(hipGetErrorString(e));
    return CUM_ELAUNCH;
  }
  hipLaunchKernelGGL(stream_hop_kernel, dim3(streams), dim3(kHopThreads), lds_bytes, (hipStream_t)stream,
                     (const HopPlan *)plan, weights, state, state_stride, in, in_stride, out, out_stride, n_hops);
  CUM_CHECK_LAUNCH();
  return CUM_OK;
}
