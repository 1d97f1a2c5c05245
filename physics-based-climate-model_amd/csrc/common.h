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

// Sum over the 64 lanes, returned wave-uniform.  Six DPP adds (row shifts 1/2/4/8 inside the 16-lane rows, then
// row_bcast:15 / row_bcast:31 across rows: lane 63 ends with the total) + one readlane, instead of six ds_bpermute round
// trips through the LDS crossbar (__shfl_xor) -- the reductions of the small per-sample kernels are latency chains of
// exactly these.  Fixed summation order.
__device__ __forceinline__ float wave_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));    // row_shr:1
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));    // row_shr:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));    // row_shr:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));    // row_shr:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, true));    // row_bcast:15
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, true));    // row_bcast:31
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Maximum of NON-NEGATIVE floats over the wave (integer compare of the bit patterns), returned wave-uniform.  Six DPP
// v_max steps (row shifts 1/2/4/8 inside the 16-lane rows, then row_bcast:15 / row_bcast:31 across rows) instead of six
// dependent ds_bpermute round trips (~400 cycles of LDS-crossbar latency) that __shfl_xor costs.  NaN bit patterns
// compare above +inf, so a NaN input yields a NaN-or-inf-class exponent downstream (what the callers want).
__device__ __forceinline__ float wave_max_nonneg(float v) {
  int x = __float_as_int(v);
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true));    // row_shr:1 (lanes shifted in: 0)
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true));    // row_shr:2
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true));    // row_shr:4
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true));    // row_shr:8  -> lane 15 of a row = row max
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, true));    // row_bcast:15 into rows 1 and 3
  x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, true));    // row_bcast:31 into rows 2 and 3
  return __int_as_float(__builtin_amdgcn_readlane(x, 63));
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

// Counter-based dropout (cnn_transformer; nn.Dropout / the attention-probability dropout of nn.MultiheadAttention,
// reference src/cnn_transformer.py:26-33).  Nothing is stored: forward and backward REGENERATE the same decision from
// (seed, step counter, site id, element index) with a 32-bit avalanche hash ("lowbias32").  rng = {seed, counter} lives in
// device memory so that a replayed hipGraph draws fresh masks every step (cm_rng_advance).  The stream differs from
// torch's Philox stream by construction -- masks are statistically, not bitwise, the reference's.
__device__ __forceinline__ unsigned cm_hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
struct DropSite {
  unsigned key, thresh;   // drop when hash(idx ^ key) < thresh
  float scale;            // 1 / (1 - p); 0 thresh = dropout off
};
__device__ __forceinline__ DropSite cm_drop_site(const unsigned* rng, unsigned site, float p) {
  DropSite d;
  d.thresh = 0u; d.scale = 1.f; d.key = 0u;
  if (rng != nullptr && p > 0.f) {
    d.key = cm_hash32(rng[0] ^ cm_hash32(rng[1] * 0x9e3779b9u + site));
    d.thresh = p >= 1.f ? 0xffffffffu : (unsigned)(p * 4294967296.0f);
    d.scale = p >= 1.f ? 0.f : 1.f / (1.f - p);
  }
  return d;
}
// multiplier of element idx: 0 (dropped) or 1 / (1 - p)
__device__ __forceinline__ float cm_drop_mul(const DropSite& d, unsigned idx) {
  return (d.thresh != 0u && cm_hash32(idx ^ d.key) < d.thresh) ? 0.f : d.scale;
}

__host__ __device__ static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

#define CM_CHECK_LAUNCH()                      \
  do {                                         \
    hipError_t e__ = hipGetLastError();        \
    if (e__ != hipSuccess) return (int)e__;    \
  } while (0)
// the job that owns this block: the descriptors' first blocks ascend (record ndesc holds the total), so it is the number of
// first blocks <= blockIdx.x minus one -- counted 64 records at a time across the lanes (one load latency; the former
// `while (descs[d + 1].first <= block) ++d` was up to ndesc DEPENDENT loads in every thread of both pack launches)
__device__ __forceinline__ int cm_job_of_block(const long long* __restrict__ descs, int ndesc) {
  int d = -1;
  for (int j0 = 0; j0 < ndesc; j0 += 64) {
    const int j = j0 + (threadIdx.x & 63);
    const bool le = j < ndesc && descs[(long long)j * 8 + 7] <= (long long)blockIdx.x;
    d += __popcll(__ballot(le));
  }
  return d;
}


