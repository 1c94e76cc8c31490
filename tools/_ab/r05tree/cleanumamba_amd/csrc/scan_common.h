// Shared pieces of the selective-scan kernels (scan_fwd.hip, scan_bwd.hip).
//
// MI355X mapping (not the upstream time-parallel BlockScan design):
//   * lane  <-> channel d (64 channels per workgroup: coalesced 256-B rows in the
//     channel-contiguous layout the in_proj GEMM produces),
//   * wave  <-> slice of NS = 8 states; the NW = ceil(N/8) waves of a workgroup cover
//     all states of its 64 channels (N = 64 -> 8 waves, 512 threads),
//   * time is walked sequentially; x_t lives in VGPRs for the whole sequence,
//   * B_t / C_t are wave-uniform -> read through the constant address space so they
//     arrive by s_load into SGPRs (no LDS traffic, no VGPRs),
//   * per 16-step chunk the workgroup computes softplus / silu once per (t, d) into
//     LDS (phase A), scans (phase B), and sums the per-wave partial results over the
//     state slices through LDS (phase C).
#pragma once
#include "common.h"

namespace cum {

constexpr int TB = 16;  // steps per chunk == checkpoint interval
constexpr int SUB = 8;  // steps whose states are held in registers in backward
constexpr int NS = 8;   // states per wave (ckpt_store / ckpt_load assume 8)

// Two states of one lane side by side: the scan arithmetic is written on pairs so that it maps onto the packed
// f32 VALU ops (v_pk_mul_f32 / v_pk_fma_f32: two results per lane per issue slot).
typedef float f2 __attribute__((ext_vector_type(2)));

struct ScanParams {
  cum_scan_shape s;
  cum_scan_grad_strides gs;
  // u, delta, z, out, dout, du, ddelta, dz hold elements of s.io_dtype (f32 or bf16); everything else is f32
  const void *u, *delta, *z;
  const float *A, *Bm, *Cm, *D, *bias;
  void *out;
  void *ypre;            // forward, optional: y before the gate (out's element type and strides), for the backward
  const void *ypre_in;   // backward, optional: the forward's ypre (dout's strides); else y is rebuilt
  float *last_state, *ckpt;
  const void *dout;
  const float *ckpt_in;
  void *du, *ddelta, *dz;
  float *ws_dA, *ws_dD, *ws_dbias, *ws_dB, *ws_dC;
  int nchunks, ngroups;
  // segmented (time-parallel) forward, scan_seg.hip: nseg segments of seg_chunks chunks; carry = its workspace
  int nseg, seg_chunks;
  float *carry;
};

// Checkpoint buffer: the NS states a lane (channel d) of wave w holds, entering half h of chunk c of clip b, as
// 32 contiguous bytes -- [(b, c, h, w, d)][NS]: two 16-byte accesses per lane, 2 KB contiguous per wave.
__device__ __forceinline__ int64_t ckpt_slot(int b, int nchunks, int c, int h, int NW, int w, int Dm, int d) {
  return (((((int64_t)b * nchunks + c) * 2 + h) * NW + w) * Dm + d) * NS;
}
__device__ __forceinline__ void ckpt_store(float *ck, int64_t slot, const float (&x)[NS]) {
  float4 *q = reinterpret_cast<float4 *>(ck + slot);
  q[0] = make_float4(x[0], x[1], x[2], x[3]);
  q[1] = make_float4(x[4], x[5], x[6], x[7]);
}
__device__ __forceinline__ void ckpt_store(float *ck, int64_t slot, const f2 (&x)[NS / 2]) {
  float4 *q = reinterpret_cast<float4 *>(ck + slot);
  q[0] = make_float4(x[0].x, x[0].y, x[1].x, x[1].y);
  q[1] = make_float4(x[2].x, x[2].y, x[3].x, x[3].y);
}
__device__ __forceinline__ void ckpt_load(const float *ck, int64_t slot, f2 (&x)[NS / 2]) {
  const float4 *q = reinterpret_cast<const float4 *>(ck + slot);
  const float4 a = q[0], c = q[1];
  x[0] = f2{a.x, a.y}; x[1] = f2{a.z, a.w}; x[2] = f2{c.x, c.y}; x[3] = f2{c.z, c.w};
}

// carry buffer of the time-parallel forms (scan_seg.hip forward, scan_bwd_small.hip backward): [(b, seg, w, d)][NS] f32
// (a segment's end state / leaving dx carry from a zero start), then the per-segment sums of delta' [(b, seg, d)]
__device__ __forceinline__ int64_t carry_slot(int b, int nseg, int seg, int NW, int w, int Dm, int d) {
  return ((((int64_t)b * nseg + seg) * NW + w) * Dm + d) * NS;
}

typedef const float __attribute__((address_space(4))) *cfp;

// B_t or C_t slice of this wave (wave-uniform address -> s_load_dwordx8).
template <bool FAST>
__device__ __forceinline__ void load_bc(const float *base, int sn, int nvalid, float (&v)[NS]) {
  cfp bp = (cfp)base;
  if constexpr (FAST) {
#pragma unroll
    for (int j = 0; j < NS; ++j) v[j] = bp[j];
  } else {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int jj = j < nvalid ? j : nvalid - 1;
      const float t = bp[jj * sn];
      v[j] = j < nvalid ? t : 0.f;
    }
  }
}

int scan_check_shape(const cum_scan_shape *s);
// segment plan of the time-parallel backward (d_state <= 16 only: scan_bwd_small.hip)
void scan_seg_plan_bwd(int batch, int dim, int dstate, int len, int *nseg, int *seg_chunks);
// time-parallel forward for grids that do not fill the chip (scan_seg.hip): plan (nseg == 1: sequential kernels),
// workspace size in f32 elements, launcher
void scan_seg_plan(int batch, int dim, int dstate, int len, int *nseg, int *seg_chunks);
int64_t scan_seg_carry_elems(int batch, int dim, int dstate, int nseg);
int launch_fwd_segmented(const ScanParams &p, hipStream_t st);
// d_state <= 16: wave-specialised backward (scan_bwd_small.hip); the caller runs scan_bwd_finalize_kernel afterwards
int launch_bwd_small(const ScanParams &p, hipStream_t st);

}  // namespace cum
