// Helpers shared by the fused outer layers (enc0.hip: first encoder layer, dec7.hip: last decoder layer), gfx950.
#pragma once
#include "common.h"

namespace cum {

typedef __attribute__((ext_vector_type(8))) __bf16 e0_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 e0_f16x8;
typedef __attribute__((ext_vector_type(4))) float e0_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned e0_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned e0_u32x2;

template <typename T>
__device__ __forceinline__ e0_f32x4 e0_mfma(e0_u32x4 a, e0_u32x4 b, e0_f32x4 c) {
  if constexpr (__is_same(T, f16))
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(e0_f16x8, a), __builtin_bit_cast(e0_f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(e0_bf16x8, a), __builtin_bit_cast(e0_bf16x8, b), c, 0, 0, 0);
}

// Row m0 + row is a real row of its clip (not one of the zero rows between clips, not past the end).  t0 = m0 mod pitch
// is formed once per step on 32-bit values: a 64-bit `m % pitch` per lane is a ~100-instruction software division, which
// made the first version of these kernels compute-bound at a third of their memory rate.
__device__ __forceinline__ bool e0_row_ok(unsigned t0, int row, int64_t m0, const int64_t M, unsigned pitch, unsigned valid) {
  unsigned tt = t0 + (unsigned)row;
  tt = tt >= pitch ? tt - pitch : tt;                // t0 < pitch, row < 32 <= pitch: one conditional subtraction
  return m0 + row < M && tt < valid;
}

// `v` where ok, else 0 -- as a select on an already computed value: left to itself the compiler branches around the
// transcendental-heavy expressions that are only used under `ok` (four exec-mask branches per 16-row tile).
__device__ __forceinline__ float e0_keep(bool ok, float v) {
  asm volatile("" : "+v"(v));
  return ok ? v : 0.f;
}

template <typename T>
__device__ __forceinline__ unsigned e0_pack2(float a, float b) {
  typedef T V2 __attribute__((ext_vector_type(2)));
  V2 v = {(T)a, (T)b};
  return __builtin_bit_cast(unsigned, v);
}

// ds_read_b64_tr_b16 pair: rows 8 g .. 8 g + 7 of one 16-column block of a row-major 16-bit tile (row stride STRIDE bytes),
// as the 8-deep K fragment of an MFMA operand.  a0: this lane's address (row 8 g + q, columns 4 pp .. of the block).
template <int STRIDE>
__device__ __forceinline__ e0_u32x4 e0_tr_read(const unsigned char *a0) {
  e0_u32x2 lo, hi;
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)a0;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(4 * STRIDE) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi) : : "memory");
  return e0_u32x4{lo.x, lo.y, hi.x, hi.y};
}

}  // namespace cum
