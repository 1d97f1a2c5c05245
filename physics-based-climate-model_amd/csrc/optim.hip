// Fused multi-tensor Adam over the flat parameter / gradient / moment buffers, plus small buffer utilities.
//
// Reference: optim.Adam(self.parameters(), lr, weight_decay) (main_final.py:742-746), torch defaults
// betas=(0.9, 0.999), eps=1e-8, coupled L2 weight decay; one step per batch (Lightning automatic optimisation).
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// `grad_scale` multiplies the gradient first (1/world_size after the RCCL sum all-reduce).
#include "common.h"
#include "../../include/climate_hip.h"

namespace {

// state[0] = step count (as int bits), state[1] = lr/(1-b1^t), state[2] = sqrt(1-b2^t): advanced on the device so
// that a captured hipGraph replays with the right bias corrections.
__global__ void adam_advance_kernel(float* __restrict__ state, float lr, float b1, float b2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int t = __float_as_int(state[0]) + 1;
    state[0] = __int_as_float(t);
    const double bc1 = 1.0 - pow((double)b1, (double)t);
    const double bc2 = 1.0 - pow((double)b2, (double)t);
    state[1] = (float)((double)lr / bc1);
    state[2] = (float)sqrt(bc2);
  }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float step_size, float bc2_sqrt,
                            const float* __restrict__ state, float b1, float b2, float eps, float wd,
                            float grad_scale) {
  if (state) {
    step_size = state[1];
    bc2_sqrt = state[2];
  }
  const long long n4 = n / 4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long t0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  auto upd = [&](float& pv, float gv, float& mv, float& vv) {
    gv = gv * grad_scale + wd * pv;
    mv = b1 * mv + (1.f - b1) * gv;
    vv = b2 * vv + (1.f - b2) * gv * gv;
    pv -= step_size * (mv / (sqrtf(vv) / bc2_sqrt + eps));
  };
  for (long long i = t0; i < n4; i += stride) {
    float4 pv = p4[i], mv = m4[i], vv = v4[i];
    const float4 gv = g4[i];
    upd(pv.x, gv.x, mv.x, vv.x);
    upd(pv.y, gv.y, mv.y, vv.y);
    upd(pv.z, gv.z, mv.z, vv.z);
    upd(pv.w, gv.w, mv.w, vv.w);
    p4[i] = pv; m4[i] = mv; v4[i] = vv;
  }
  for (long long i = n4 * 4 + t0; i < n; i += stride) upd(p[i], g[i], m[i], v[i]);
}

// Zero-fill as a kernel, not hipMemsetAsync: a captured hipMemsetAsync node was observed (ROCm 7.0 runtime bundled
// with torch, MI355X) to clear memory beyond its range on hipGraph replay; a kernel node replays exactly.
__global__ void zero_kernel(uint32_t* __restrict__ p, size_t words, unsigned char* __restrict__ tail, int tail_bytes) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t t0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t w4 = words / 4;
  uint4* p4 = reinterpret_cast<uint4*>(p);
  const bool aligned = ((size_t)p & 15) == 0;
  if (aligned) {
    for (size_t i = t0; i < w4; i += stride) p4[i] = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = w4 * 4 + t0; i < words; i += stride) p[i] = 0u;
  } else {
    for (size_t i = t0; i < words; i += stride) p[i] = 0u;
  }
  if (t0 < (size_t)tail_bytes) tail[t0] = 0;
}

__global__ void scale_kernel(float* __restrict__ x, long long n, float s) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    x[i] *= s;
}

}  // namespace

extern "C" {

int cm_version(void) { return 3; }   // 3: gate backward derives (umax, cnt) itself; 2: bf16x6 + first-layer entry points
const char* cm_arch(void) { return "gfx950"; }

int cm_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int step, float grad_scale, cm_stream stream) {
  if (n <= 0 || step < 1) return -22;
  if ((((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) != 0) return -22;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  adam_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, step_size, bc2_sqrt, nullptr, beta1, beta2,
                                                             eps, weight_decay, grad_scale);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, float* state, float lr, float beta1,
                     float beta2, float eps, float weight_decay, float grad_scale, cm_stream stream) {
  if (n <= 0 || !state) return -22;
  if ((((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) != 0) return -22;
  adam_advance_kernel<<<1, 64, 0, (hipStream_t)stream>>>(state, lr, beta1, beta2);
  CM_CHECK_LAUNCH();
  long long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  adam_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, 0.f, 1.f, state, beta1, beta2, eps,
                                                             weight_decay, grad_scale);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_scale(float* x, long long n, float s, cm_stream stream) {
  if (n <= 0) return -22;
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  scale_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(x, n, s);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_zero(void* p, size_t bytes, cm_stream stream) {
  if (bytes == 0) return 0;
  if ((size_t)p & 3) return -22;
  const size_t words = bytes / 4;
  const int tail = (int)(bytes % 4);
  size_t blocks = (words / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  zero_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>((uint32_t*)p, words, (unsigned char*)p + words * 4, tail);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
