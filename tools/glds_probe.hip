// Probe: does global_load_lds_dwordx4 reach LDS addresses >= 64 KiB (M0 width)?  Build: hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void *lds_ptr;
typedef const __attribute__((address_space(1))) void *glb_ptr;
__global__ __launch_bounds__(64) void probe(const uint4 *src, uint4 *out, int off_chunks) {
  __shared__ uint4 lds[8192 + 64];  // 128 KiB + 1 KiB
  for (int i = threadIdx.x; i < 8192 + 64; i += 64) lds[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  const int oc = __builtin_amdgcn_readfirstlane(off_chunks);
  __builtin_amdgcn_global_load_lds((glb_ptr)(src + threadIdx.x), (lds_ptr)(&lds[oc]), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[threadIdx.x] = lds[oc + threadIdx.x];                 // where it should be
  out[64 + threadIdx.x] = lds[(oc & 4095) + threadIdx.x];   // where a 16-bit M0 would put it
}
int main() {
  uint4 h[64], *d, *o, r[128];
  for (int i = 0; i < 64; ++i) h[i] = make_uint4(1000 + i, 2, 3, 4);
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int off : {0, 2048, 4095, 4096, 6000, 8192}) {
    probe<<<1, 64>>>(d, o, off);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("off %5d chunks (%6d B): at target lane0=%u lane63=%u | at (off mod 64K) lane0=%u\n", off, off * 16, r[0].x, r[63].x, r[64].x);
  }
  return 0;
}
