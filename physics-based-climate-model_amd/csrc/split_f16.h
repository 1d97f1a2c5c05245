// fp32 -> two fp16 pieces (hi, lo; 22 mantissa bits in total) for the "fp16x3" matrix-core kernels:
//   a*b ~= hi*hi + hi*lo + lo*hi, fp32 accumulate  (dropped lo*lo and the pieces' own residuals: < 2^-21 |a*b|)
// Half the MFMA work and two thirds of the LDS traffic of bf16x6 (split_bf16.h).  fp16 has 5 exponent bits, so the
// operand is multiplied by a power of two first (exact) that places the tensor's largest magnitude near 2^14; elements
// more than ~2^-28 below the maximum lose relative (not absolute) precision, which an L2-norm criterion does not see.
#pragma once
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2h_t __attribute__((ext_vector_type(2)));

// Split two (pre-scaled) floats into two packed fp16 pairs (element 0 in the low half).  The residual v - float(hi) is
// exact in fp32 (hi keeps 11 significant bits of a 24-bit value).
__device__ __forceinline__ void split2_pair_f16(float v0, float v1, unsigned& ph, unsigned& pl) {
  const f32x2h_t v = {v0, v1};
  const f16x2_t h = __builtin_convertvector(v, f16x2_t);
  ph = __builtin_bit_cast(unsigned, h);
  const f32x2h_t hf = __builtin_convertvector(h, f32x2h_t);
  const f32x2h_t r = v - hf;
  pl = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
}
