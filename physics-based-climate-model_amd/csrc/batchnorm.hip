// BatchNorm2d (+ residual add, + ReLU), Dropout2d plane scaling and the 1x1-in-3x3 weight embedding of SimpleCNN.
//
// Reference: ResidualBlock / SimpleCNN (src/models.py:44-123): nn.BatchNorm2d(c) in train mode normalises with the
// statistics of the batch -- mean and BIASED variance per channel over (N, H, W) -- and moves its running buffers by
// momentum 0.1 towards the batch mean and the UNBIASED variance; in eval mode it normalises with the running buffers.
// `out += skip(identity); relu(out)` (src/models.py:72-73) is folded into the second BatchNorm's launch, the ReLU after a
// BatchNorm (src/models.py:66,93,115) into that BatchNorm's.  nn.Dropout2d (src/models.py:109) zeroes whole (sample,
// channel) planes and scales the rest by 1/(1-p): a per-plane multiplier.
//
// One workgroup per channel: the channel's N planes of H*W floats are read for the statistics (mean first, then the sum
// of squares about it) and once more (from L2) by the apply pass.
#include "common.h"
#include "../../include/climate_hip.h"

namespace {

constexpr int BN_THREADS = 512;

// x[n][c][hw]: element i of channel c's N*HW values lives at ((i / HW) * C + c) * HW + i % HW; vectorised when HW % 4 == 0
template <bool VEC, typename F>
__device__ __forceinline__ void for_channel(int c, int C, int N, int HW, F&& f) {
  const int tid = threadIdx.x;
  if constexpr (VEC) {
    const int hv = HW >> 2;
    for (int i = tid; i < N * hv; i += BN_THREADS) {
      const int n = i / hv, q = i - n * hv;
      f(((long long)n * C + c) * HW + 4 * q);
    }
  } else {
    for (int i = tid; i < N * HW; i += BN_THREADS) {
      const int n = i / HW, q = i - n * HW;
      f(((long long)n * C + c) * HW + q);
    }
  }
}

template <bool VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_fwd_train_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta,
                                                                   const float* __restrict__ resid,
                                                                   float* __restrict__ y, float* __restrict__ save,
                                                                   float* __restrict__ rmean, float* __restrict__ rvar,
                                                                   float momentum, float eps, int relu, int N, int C,
                                                                   int HW) {
  __shared__ float red[32];
  const int c = blockIdx.x;
  const float cnt = (float)N * (float)HW;
  // pass 1: mean (sums shifted by the first element only to keep them small)
  const float pivot = x[(long long)c * HW];
  float s1 = 0.f;
  for_channel<VEC>(c, C, N, HW, [&](long long o) {
    if constexpr (VEC) {
      const float4 v = *reinterpret_cast<const float4*>(x + o);
      s1 += ((v.x - pivot) + (v.y - pivot)) + ((v.z - pivot) + (v.w - pivot));
    } else {
      s1 += x[o] - pivot;
    }
  });
  const float mean = pivot + block_sum(s1, red) / cnt;
  // pass 2: sum of squares about the mean
  float s2 = 0.f;
  for_channel<VEC>(c, C, N, HW, [&](long long o) {
    if constexpr (VEC) {
      const float4 v = *reinterpret_cast<const float4*>(x + o);
      const float a = v.x - mean, b = v.y - mean, d = v.z - mean, e = v.w - mean;
      s2 += (a * a + b * b) + (d * d + e * e);
    } else {
      const float a = x[o] - mean;
      s2 += a * a;
    }
  });
  const float var = block_sum(s2, red) / cnt;            // biased: what normalises
  const float rstd = rsqrtf(var + eps);
  if (threadIdx.x == 0) {
    save[2 * c] = mean;
    save[2 * c + 1] = rstd;
    if (rmean) {
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
      const float unb = cnt > 1.f ? var * cnt / (cnt - 1.f) : var;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
    }
  }
  const float ga = gamma[c] * rstd, be = beta[c] - mean * rstd * gamma[c];
  for_channel<VEC>(c, C, N, HW, [&](long long o) {
    if constexpr (VEC) {
      const float4 v = *reinterpret_cast<const float4*>(x + o);
      float4 r = make_float4(v.x * ga + be, v.y * ga + be, v.z * ga + be, v.w * ga + be);
      if (resid) {
        const float4 q = *reinterpret_cast<const float4*>(resid + o);
        r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
      }
      if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
      *reinterpret_cast<float4*>(y + o) = r;
    } else {
      float r = x[o] * ga + be;
      if (resid) r += resid[o];
      y[o] = relu ? fmaxf(r, 0.f) : r;
    }
  });
}

// eval mode: y = relu?((x - running_mean) / sqrt(running_var + eps) * gamma + beta + resid); also writes save = {mean, rstd}
__global__ void bn_fwd_eval_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, const float* __restrict__ rmean,
                                   const float* __restrict__ rvar, const float* __restrict__ resid,
                                   float* __restrict__ y, float* __restrict__ save, float eps, int relu, int C, int HW,
                                   long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % C);
    const float rstd = rsqrtf(rvar[c] + eps);
    float r = (x[i] - rmean[c]) * rstd * gamma[c] + beta[c];
    if (resid) r += resid[i];
    y[i] = relu ? fmaxf(r, 0.f) : r;
    if (save && i < C) {
      save[2 * i] = rmean[i];
      save[2 * i + 1] = rsqrtf(rvar[i] + eps);
    }
  }
}

// backward: g = dy * [y > 0] (relu) ; dbeta += sum g ; dgamma += sum g * xhat ;
//   train: dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)) ; eval (batch_stats = 0): dx = gamma * rstd * g ;
//   dres (nullable) = g: the gradient of the residual input added before the ReLU
template <bool VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ dy,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ save, float* __restrict__ dx,
                                                             float* __restrict__ dres, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int relu, int batch_stats,
                                                             int N, int C, int HW) {
  __shared__ float red[32];
  const int c = blockIdx.x;
  const float cnt = (float)N * (float)HW;
  const float mean = save[2 * c], rstd = save[2 * c + 1];
  float sg = 0.f, sgx = 0.f;
  for_channel<VEC>(c, C, N, HW, [&](long long o) {
    if constexpr (VEC) {
      const float4 xv = *reinterpret_cast<const float4*>(x + o);
      float4 g = *reinterpret_cast<const float4*>(dy + o);
      if (relu) {
        const float4 yv = *reinterpret_cast<const float4*>(y + o);
        g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
      }
      sg += (g.x + g.y) + (g.z + g.w);
      sgx += (g.x * ((xv.x - mean) * rstd) + g.y * ((xv.y - mean) * rstd)) +
             (g.z * ((xv.z - mean) * rstd) + g.w * ((xv.w - mean) * rstd));
    } else {
      float g = dy[o];
      if (relu) g = y[o] > 0.f ? g : 0.f;
      sg += g;
      sgx += g * ((x[o] - mean) * rstd);
    }
  });
  sg = block_sum(sg, red);
  sgx = block_sum(sgx, red);
  if (threadIdx.x == 0) {
    dbeta[c] += sg;
    dgamma[c] += sgx;
  }
  const float k = gamma[c] * rstd;
  const float m1 = batch_stats ? sg / cnt : 0.f, m2 = batch_stats ? sgx / cnt : 0.f;
  for_channel<VEC>(c, C, N, HW, [&](long long o) {
    if constexpr (VEC) {
      const float4 xv = *reinterpret_cast<const float4*>(x + o);
      float4 g = *reinterpret_cast<const float4*>(dy + o);
      if (relu) {
        const float4 yv = *reinterpret_cast<const float4*>(y + o);
        g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
      }
      if (dres) *reinterpret_cast<float4*>(dres + o) = g;
      float4 r;
      r.x = k * (g.x - m1 - ((xv.x - mean) * rstd) * m2);
      r.y = k * (g.y - m1 - ((xv.y - mean) * rstd) * m2);
      r.z = k * (g.z - m1 - ((xv.z - mean) * rstd) * m2);
      r.w = k * (g.w - m1 - ((xv.w - mean) * rstd) * m2);
      *reinterpret_cast<float4*>(dx + o) = r;
    } else {
      float g = dy[o];
      if (relu) g = y[o] > 0.f ? g : 0.f;
      if (dres) dres[o] = g;
      dx[o] = k * (g - m1 - ((x[o] - mean) * rstd) * m2);
    }
  });
}

__global__ void scale_planes_kernel(const float* __restrict__ x, const float* __restrict__ m, float* __restrict__ out,
                                    int HW, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    out[i] = x[i] * m[i / HW];
}

// w3[co][ci][3][3] = 0 except the centre tap = w1[co][ci]
__global__ void embed_center_kernel(const float* __restrict__ w1, float* __restrict__ w3, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n * 9; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i / 9;
    w3[i] = (i - e * 9) == 4 ? w1[e] : 0.f;
  }
}

// dw1[co][ci] += G[co][4][ci]  (centre tap of a tap-major weight-gradient staging tensor G[co][9][ctot])
__global__ void extract_center_kernel(const float* __restrict__ g, float* __restrict__ dw1, int cout, int ctot) {
  const long long n = (long long)cout * ctot;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long o = i / ctot, c = i - o * ctot;
    dw1[i] += g[(o * 9 + 4) * ctot + c];
  }
}

inline int bn_grid(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int cm_bn_fwd(const float* x, const float* gamma, const float* beta, const float* resid, float* y, float* save,
              float* running_mean, float* running_var, float momentum, float eps, int relu, int training, int n, int c,
              int hw, cm_stream stream) {
  if (n <= 0 || c <= 0 || hw <= 0 || !x || !y || !save) return -22;
  hipStream_t st = (hipStream_t)stream;
  if (training) {
    const bool vec = hw % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)resid) & 15) == 0;
    if (vec)
      bn_fwd_train_kernel<true><<<c, BN_THREADS, 0, st>>>(x, gamma, beta, resid, y, save, running_mean, running_var,
                                                          momentum, eps, relu, n, c, hw);
    else
      bn_fwd_train_kernel<false><<<c, BN_THREADS, 0, st>>>(x, gamma, beta, resid, y, save, running_mean, running_var,
                                                           momentum, eps, relu, n, c, hw);
  } else {
    if (!running_mean || !running_var) return -22;
    const long long total = (long long)n * c * hw;
    bn_fwd_eval_kernel<<<bn_grid(total), 256, 0, st>>>(x, gamma, beta, running_mean, running_var, resid, y, save, eps,
                                                       relu, c, hw, total);
  }
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_bn_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* save, float* dx,
              float* dres, float* dgamma, float* dbeta, int relu, int training, int n, int c, int hw,
              cm_stream stream) {
  if (n <= 0 || c <= 0 || hw <= 0 || !x || !dy || !dx || !save || (relu && !y)) return -22;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = hw % 4 == 0 &&
                   (((uintptr_t)x | (uintptr_t)y | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)dres) & 15) == 0;
  if (vec)
    bn_bwd_kernel<true><<<c, BN_THREADS, 0, st>>>(x, y, dy, gamma, save, dx, dres, dgamma, dbeta, relu, training, n, c, hw);
  else
    bn_bwd_kernel<false><<<c, BN_THREADS, 0, st>>>(x, y, dy, gamma, save, dx, dres, dgamma, dbeta, relu, training, n, c, hw);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_scale_planes(const float* x, const float* mult, float* out, long long planes, int hw, cm_stream stream) {
  if (planes <= 0 || hw <= 0) return -22;
  const long long total = planes * hw;
  scale_planes_kernel<<<bn_grid(total), 256, 0, (hipStream_t)stream>>>(x, mult, out, hw, total);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_embed_center_tap(const float* w1, float* w3, int cout, int cin, cm_stream stream) {
  if (cout <= 0 || cin <= 0) return -22;
  const long long n = (long long)cout * cin;
  embed_center_kernel<<<bn_grid(n * 9), 256, 0, (hipStream_t)stream>>>(w1, w3, n);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_extract_center_tap(const float* g, float* dw1, int cout, int ctot, cm_stream stream) {
  if (cout <= 0 || ctot <= 0) return -22;
  extract_center_kernel<<<bn_grid((long long)cout * ctot), 256, 0, (hipStream_t)stream>>>(g, dw1, cout, ctot);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
