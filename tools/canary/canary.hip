// Probe-only kernels (not part of the product library): a "canary" workgroup that parks known values in its VGPRs, its LDS
// and behind its barriers for a while and then checks them -- which per-CU resource, if any, does a co-resident kernel
// disturb?  Built by tools/coresidency_probe.py into tools/canary/libcanary.so.
#include <hip/hip_runtime.h>

namespace {

__device__ __forceinline__ unsigned mix(unsigned a, unsigned b) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
  h ^= h >> 15;
  return h * 0xC2B2AE3Du ^ (h >> 13);
}

// report[0] register mismatches, [1] LDS mismatches, [2] barrier-phase mismatches, [3] global-load mismatches,
// [4] workgroups that ran
template <int NREG>
__global__ __launch_bounds__(1024) void canary_kernel(unsigned* report, const unsigned* src, int nsrc, int spins,
                                                      int lds_words) {
  extern __shared__ unsigned sh[];
  const int tid = threadIdx.x, wave = tid >> 6, nw = blockDim.x >> 6;
  unsigned* phase = sh;               // [16]
  unsigned* fill = sh + 16;           // [lds_words]
  unsigned r[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    r[i] = mix(blockIdx.x * 1024u + tid, i);
    asm volatile("" : "+v"(r[i]));
  }
  for (int i = tid; i < lds_words; i += blockDim.x) fill[i] = mix(blockIdx.x, i);
  if (tid < 16) phase[tid] = 0;
  __syncthreads();
  unsigned bad_bar = 0, bad_ld = 0;
  for (int s = 1; s <= spins; ++s) {
    if ((tid & 63) == 0) phase[wave] = s;
    __syncthreads();
    for (int w = 0; w < nw; ++w) bad_bar += phase[w] != (unsigned)s;
    // a few global loads of known content (the pattern src[i] = mix(i, 12345))
    const int i = (int)(mix(tid + s * 977u, blockIdx.x) % (unsigned)(nsrc - 1)) & ~1;
    const uint2 v = *reinterpret_cast<const uint2*>(src + i);
    bad_ld += (v.x != mix(i, 12345u)) + (v.y != mix(i + 1, 12345u));
#pragma unroll
    for (int k = 0; k < NREG; ++k) asm volatile("" : "+v"(r[k]));
    __builtin_amdgcn_s_sleep(20);
    __syncthreads();
  }
  unsigned bad_reg = 0, bad_lds = 0;
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    asm volatile("" : "+v"(r[i]));
    bad_reg += r[i] != mix(blockIdx.x * 1024u + tid, i);
  }
  for (int i = tid; i < lds_words; i += blockDim.x) bad_lds += fill[i] != mix(blockIdx.x, i);
  if (bad_reg) atomicAdd(report + 0, bad_reg);
  if (bad_lds) atomicAdd(report + 1, bad_lds);
  if (bad_bar) atomicAdd(report + 2, bad_bar);
  if (bad_ld) atomicAdd(report + 3, bad_ld);
  if (tid == 0) atomicAdd(report + 4, 1u);
}

__global__ void fill_src(unsigned* src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) src[i] = mix(i, 12345u);
}

}  // namespace

extern "C" {

int canary_fill(unsigned* src, int n, void* stream) {
  fill_src<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(src, n);
  return (int)hipGetLastError();
}

// nreg in {24, 28, 56}: parked registers per thread (the kernel's VGPR count is a little more)
int canary_launch(unsigned* report, const unsigned* src, int nsrc, int blocks, int threads, int nreg, int spins,
                  int lds_words, void* stream) {
  const size_t lds = (size_t)(16 + lds_words) * 4;
  hipStream_t st = (hipStream_t)stream;
  if (nreg == 24) canary_kernel<24><<<blocks, threads, lds, st>>>(report, src, nsrc, spins, lds_words);
  else if (nreg == 56) canary_kernel<56><<<blocks, threads, lds, st>>>(report, src, nsrc, spins, lds_words);
  else canary_kernel<28><<<blocks, threads, lds, st>>>(report, src, nsrc, spins, lds_words);
  return (int)hipGetLastError();
}

}  // extern "C"

// ---- packed-FP32 canary: one instruction form, executed ITERS times on fixed operands and compared with the same arithmetic
// done by scalar v_mul / v_add / v_fma (tools/coresidency_probe.py traced a co-residency mismatch of cm_block_tail_bwd to
// the lanes 48-63 of a wave executing v_pk_mul_f32 with an op_sel swizzle).
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

// report[0..3]: mismatching results per 16-lane row of the wave; [4]: workgroups run; [5]: iterations with any mismatch
template <int FORM>
__global__ __launch_bounds__(1024) void pk_canary_kernel(unsigned* report, int iters, int lds_words) {
  extern __shared__ unsigned sh[];
  const int tid = threadIdx.x;
  for (int i = tid; i < lds_words; i += blockDim.x) sh[i] = i;       // (same LDS footprint as the tail kernel)
  __syncthreads();
  f2 a, b, c;
  a.x = 1.f + (mix(tid, 1) & 0xffff) * (1.f / 65536.f);
  a.y = 1.f + (mix(tid, 2) & 0xffff) * (1.f / 65536.f);
  b.x = 1.f + (mix(tid, 3) & 0xffff) * (1.f / 65536.f);
  b.y = 1.f + (mix(tid, 4) & 0xffff) * (1.f / 65536.f);
  c.x = 1.f + (mix(tid, 5) & 0xffff) * (1.f / 65536.f);
  c.y = 1.f + (mix(tid, 6) & 0xffff) * (1.f / 65536.f);
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
  f2 want;
  if (FORM == 0) { want.x = __fmul_rn(a.x, b.y); want.y = __fmul_rn(a.y, b.x); }
  if (FORM == 1) { want.x = __fmul_rn(a.y, b.x); want.y = __fmul_rn(a.y, b.y); }
  if (FORM == 2) { want.x = __fsub_rn(a.y, b.x); want.y = __fsub_rn(a.x, b.y); }
  if (FORM == 3) { want.x = __fmaf_rn(a.y, b.x, c.x); want.y = __fmaf_rn(a.y, b.y, c.y); }
  if (FORM == 4) { want.x = __fmul_rn(a.x, b.x); want.y = __fmul_rn(a.y, b.y); }
  if (FORM == 5) { want.x = __fmaf_rn(a.x, b.x, c.x); want.y = __fmaf_rn(a.x, b.y, c.y); }
  if (FORM == 6) { want.x = __fmul_rn(a.x, b.y); want.y = __fmul_rn(a.y, b.x); }
  if (FORM == 13) { want.x = __fmul_rn(a.y, b.x); want.y = __fmul_rn(a.x, b.y); }
  if (FORM == 14) { want.x = __fadd_rn(a.x, b.y); want.y = __fadd_rn(a.y, b.x); }
  if (FORM == 15) { want.x = __fmaf_rn(a.x, b.y, c.x); want.y = __fmaf_rn(a.y, b.x, c.y); }
  if (FORM == 16) { want.x = __fmul_rn(a.y, b.y); want.y = __fmul_rn(a.x, b.x); }
  if (FORM == 17) { want.x = a.y; want.y = b.x; }          // v_pk_mov_b32 op_sel:[1,0]: lo <- src0.hi, hi <- src1.lo
  if (FORM == 18) { want.x = a.x; want.y = b.y; }          // v_pk_mov_b32 op_sel:[0,1]: lo <- src0.lo, hi <- src1.hi
  if (FORM == 19) { want.x = a.y; want.y = b.y; }          // v_pk_mov_b32 op_sel:[1,1]
  if (FORM == 20) { want.x = a.x; want.y = b.x; }          // v_pk_mov_b32 op_sel:[0,0]
  // (the references of the swapped forms must not themselves be compiled into a packed instruction: pin them)
  asm volatile("" : "+v"(want.x));
  asm volatile("" : "+v"(want.y));
  asm volatile("" : "+v"(want));
  unsigned park[56];          // (keeps the kernel's register allocation at the tail kernel's size: 69 VGPRs)
#pragma unroll
  for (int i = 0; i < 56; ++i) {
    park[i] = mix(tid, 100 + i);
    asm volatile("" : "+v"(park[i]));
  }
  unsigned bad = 0;
  for (int s = 0; s < iters; ++s) {
#pragma unroll
    for (int i = 0; i < 56; ++i) asm volatile("" : "+v"(park[i]));
    f2 r;
    if (FORM == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    if (FORM == 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    if (FORM == 2)
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    if (FORM == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    if (FORM == 4) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    if (FORM == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    if (FORM == 13) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    if (FORM == 14) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    if (FORM == 15) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    if (FORM == 16) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
    if (FORM == 17) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    if (FORM == 18) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    if (FORM == 19) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    if (FORM == 20) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(r) : "v"(a), "v"(b));
    if (FORM == 6) {
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r.x) : "v"(a.x), "v"(b.y));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r.y) : "v"(a.y), "v"(b.x));
    }
    if (FORM >= 7 && FORM <= 12) {
      unsigned ua = __float_as_uint(a.x), ub = __float_as_uint(b.y), ur = 0, ur2 = 0;
      if (FORM == 7) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(ur) : "v"(ua), "v"(ub));
      if (FORM == 8) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(ur) : "v"(ua), "v"(ub));
      if (FORM == 9) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(ur) : "v"(ua));
      if (FORM == 10) asm volatile("v_rcp_f32 %0, %1" : "=v"(ur) : "v"(ua));
      if (FORM == 11) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(*(unsigned long long*)&r) : "v"(ua), "v"(ub) : "vcc");
      if (FORM == 12) asm volatile("v_lshrrev_b32 %0, 13, %1\n v_xor_b32 %0, %0, %2" : "=&v"(ur) : "v"(ua), "v"(ub));
      if (FORM != 11) { r.x = __uint_as_float(ur); r.y = __uint_as_float(ur2); }
      if (s == 0) want = r;         // (the first iteration's result is the reference; a wrong reference shows as ~ITERS mismatches)
    }
    asm volatile("s_nop 1" ::: "memory");
    bad += (__float_as_uint(r.x) != __float_as_uint(want.x)) + (__float_as_uint(r.y) != __float_as_uint(want.y));
  }
#pragma unroll
  for (int i = 0; i < 56; ++i) bad += park[i] != mix(tid, 100 + i) ? 1000000u : 0u;
  // is threadIdx.x itself (v0, written by the wave launcher) still what it must be?  lane id from the hardware counter,
  // wave number from lane 0's copy
  const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int tid_now = threadIdx.x;
  int tid_chk = tid_now;
  asm volatile("" : "+v"(tid_chk));
  const int must = (__builtin_amdgcn_readfirstlane(tid_chk) & ~63) + lane;
  if (tid_chk != must) {
    atomicAdd(report + 6, 1u);                       // lanes whose threadIdx.x changed
    atomicMax(report + 7, (unsigned)tid_chk);        // largest foreign value seen
    atomicAdd(report + 8 + (lane >> 4), 1u);         // by 16-lane row (hardware lane id)
    report[12 + (lane & 15)] = (unsigned)tid_chk;    // one sample of the foreign values
    report[28 + (lane & 15)] = (unsigned)must;
  }
  if (bad) {
    atomicAdd(report + (lane >> 4), bad);
    atomicAdd(report + 5, 1u);
  }
  if (tid == 0) atomicAdd(report + 4, 1u);
}

}  // namespace

extern "C" int pk_canary_launch(unsigned* report, int form, int blocks, int threads, int iters, int lds_words, void* stream) {
  const size_t lds = (size_t)lds_words * 4;
  hipStream_t st = (hipStream_t)stream;
  switch (form) {
    case 0: pk_canary_kernel<0><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 1: pk_canary_kernel<1><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 2: pk_canary_kernel<2><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 3: pk_canary_kernel<3><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 4: pk_canary_kernel<4><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 5: pk_canary_kernel<5><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 6: pk_canary_kernel<6><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 7: pk_canary_kernel<7><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 8: pk_canary_kernel<8><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 9: pk_canary_kernel<9><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 10: pk_canary_kernel<10><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 11: pk_canary_kernel<11><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 12: pk_canary_kernel<12><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 13: pk_canary_kernel<13><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 14: pk_canary_kernel<14><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 15: pk_canary_kernel<15><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 16: pk_canary_kernel<16><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 17: pk_canary_kernel<17><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 18: pk_canary_kernel<18><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    case 19: pk_canary_kernel<19><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
    default: pk_canary_kernel<20><<<blocks, threads, lds, st>>>(report, iters, lds_words); break;
  }
  return (int)hipGetLastError();
}

// ---- synthetic aggressors: one suspect instruction class each, in a loop (which of them disturbs the canary?)
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void aggressor_kernel(float* sink, int iters) {
  const int tid = threadIdx.x;
  float v = 1.f + (tid & 63) * 0.001f;
  int x = tid * 3 + 1;
  f16v acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  h8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.001f * (tid + i)); hb[i] = (_Float16)(0.002f * (tid - i)); }
  for (int s = 0; s < iters; ++s) {
    if (MODE == 0) {          // row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3), as wave_sum / wave_max_nonneg use them
      x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, true));
      x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, true));
    }
    if (MODE == 1) {          // row_shr only
      x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true));
      x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true));
    }
    if (MODE == 2) {          // the f16 matrix instruction of the fp16x3 kernels
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
    }
    if (MODE == 3) {          // readlane / readfirstlane
      x += __builtin_amdgcn_readlane(x, 63) & 1;
      x += __builtin_amdgcn_readfirstlane(x) & 1;
    }
    if (MODE == 4) {          // row_bcast:15 alone
      x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, true));
    }
    if (MODE == 5) {          // row_bcast:31 alone
      x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, true));
    }
    if (MODE == 6) {          // row_bcast:15 with all rows enabled and no bound_ctrl
      x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x142, 0xf, 0xf, false));
    }
    x = x * 3 + 1;
    v = v * 1.0001f + 0.5f;
  }
  float t = v + (float)x;
  for (int i = 0; i < 16; ++i) t += acc[i];
  if (t == 12345.678f) sink[tid] = t;      // (never true: keeps the loop alive)
}

}  // namespace

extern "C" int aggressor_launch(float* sink, int mode, int blocks, int iters, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case 0: aggressor_kernel<0><<<blocks, 256, 0, st>>>(sink, iters); break;
    case 1: aggressor_kernel<1><<<blocks, 256, 0, st>>>(sink, iters); break;
    case 2: aggressor_kernel<2><<<blocks, 256, 0, st>>>(sink, iters); break;
    case 3: aggressor_kernel<3><<<blocks, 256, 0, st>>>(sink, iters); break;
    case 4: aggressor_kernel<4><<<blocks, 256, 0, st>>>(sink, iters); break;
    case 5: aggressor_kernel<5><<<blocks, 256, 0, st>>>(sink, iters); break;
    default: aggressor_kernel<6><<<blocks, 256, 0, st>>>(sink, iters); break;
  }
  return (int)hipGetLastError();
}
