// fp32 -> three bf16 pieces (hi, mid, lo; 24 mantissa bits in total) for the "bf16x6" matrix-core kernels
// (conv3x3_split.hip, wgrad3x3_split.hip): a*b ~= hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid, fp32 accumulate.
#pragma once
#include "common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// Split two floats into three packed bf16 pairs (element 0 in the low half).  The vector convert lowers to
// v_cvt_pk_bf16_f32 (round-to-nearest-even); each residual v - float(piece) is exact in fp32.
__device__ __forceinline__ void split3_pair(float v0, float v1, unsigned& ph, unsigned& pm, unsigned& pl) {
  f32x2_t v = {v0, v1};
  ph = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
  f32x2_t hf = {__uint_as_float(ph << 16), __uint_as_float(ph & 0xffff0000u)};
  const f32x2_t r1 = v - hf;
  pm = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_t));
  f32x2_t mf = {__uint_as_float(pm << 16), __uint_as_float(pm & 0xffff0000u)};
  const f32x2_t r2 = r1 - mf;
  pl = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2_t));
}

// The six leading piece products, smallest terms first.  a[], b[] = {hi, mid, lo} fragments.
__device__ __forceinline__ f32x16 mfma_bf16x6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
  return acc;
}
