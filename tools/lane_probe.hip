// Probe of the cross-lane primitives used by scan_bwd (run on the GPU box; prints lane maps).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../cleanumamba_amd/csrc/common.h"
extern "C" void cum_set_error(const char*) {}
using namespace cum;
__device__ __forceinline__ void rs8(const float (&v)[8], float (&r)[2], float* dbg, int lane) {
  float h[4];
  for (int i = 0; i < 4; ++i) { float a = v[i], b = v[4 + i]; swap32(a, b); h[i] = a + b; }
  float q[2];
  for (int i = 0; i < 2; ++i) { float a = h[i], b = h[2 + i]; swap16(a, b); q[i] = a + b; }
  dbg[lane] = h[0]; dbg[64 + lane] = h[2]; dbg[128 + lane] = q[0]; dbg[192 + lane] = q[1];
  r[0] = row16_allsum(q[0]);
  r[1] = row16_allsum(q[1]);
}
__global__ void k(float* out, int mode) {
  int lane = threadIdx.x;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = (mode == 0) ? (lane == 0 ? (float)(j + 1) : 0.f) : (float)(j + 1);
  float r[2];
  rs8(v, r, out + 256, lane);
  out[lane] = r[0]; out[64 + lane] = r[1];
  out[128 + lane] = row16_allsum(lane == 0 ? 1.f : 0.f);
  out[192 + lane] = row16_allsum(1.f);
}
int main() {
  float* d; hipMalloc(&d, 512 * 4);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"r0", "r1", "allsum(lane0=1)", "allsum(1)", "h0", "h2", "q0", "q1"};
    for (int r = 0; r < 8; ++r) { printf("mode%d %s:", mode, names[r]); for (int i = 0; i < 64; ++i) printf(" %g", h[r * 64 + i]); printf("\n"); }
  }
  return 0;
}
