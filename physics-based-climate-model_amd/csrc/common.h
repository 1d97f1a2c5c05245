// Shared device/host helpers for the gfx950 climate-emulator kernels.
// All tensors are fp32, NCHW with channel stride == H*W.  Every kernel takes an explicit per-sample stride (in
// elements) next to each pointer, so time slices of [B,T,...] buffers (pointer + t*CHW, stride T*CHW) and
// channel-offset slices of concat buffers need no copies.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CM_WAVE 64

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blocks of up to 1024 threads; result valid in every thread. `red` needs >= 17 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
// accurate variants (expf / tanhf) are used where the 1e-4 parity budget matters
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.f / (1.f + expf(-x)); }

__host__ __device__ static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

#define CM_CHECK_LAUNCH()                      \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return (int)e__;    \
  } while (0)
