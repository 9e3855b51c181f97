// fp32 products on the bf16 matrix pipe (shared by gemm.hip and dense_x3.hip).
// gfx950 runs f32 MFMA at 1/16 of the bf16 rate.  Every operand is split three ways, x = s0 + s1 + s2 with s0 = bf16(x),
// s1 = bf16(x - s0), s2 = bf16(x - s0 - s1) (24 mantissa bits in all), and a product keeps the six cross terms down to
// 2^-24 of the result (a0 b0 + a0 b1 + a1 b0 + a0 b2 + a1 b1 + a2 b0), each a v_mfma_f32_32x32x16_bf16 accumulating in
// fp32 (a bf16 x bf16 product is exact in fp32): six bf16 MFMAs of K = 16 replace eight f32 MFMAs of K = 2 at 3/8 of the
// cycles, fp32-accurate.
#pragma once
#include "common.h"

namespace mp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3_bf16(const float (&x)[8], bf16x8& s0, bf16x8& s1, bf16x8& s2) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 b0 = (__bf16)x[i];
    const float r1 = x[i] - (float)b0;
    const __bf16 b1 = (__bf16)r1;
    s0[i] = b0; s1[i] = b1; s2[i] = (__bf16)(r1 - (float)b1);
  }
}

// smallest terms first, so that the accumulator meets them before the leading product
__device__ __forceinline__ void mfma6(f32x16& acc, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

}  // namespace mp
