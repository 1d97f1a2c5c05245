// Output head (1x1 conv + bias), MSE loss, and their backward.
//
// Reference: self.head = nn.Conv2d(base, out_ch, 1) (src/unet_convlstm_attention.py:56,104); nn.MSELoss()
// (main_final.py:544,559): loss = mean((pred - y)^2) over B*out_ch*H*W; d pred = 2 (pred - y) / N.
#include "common.h"
#include "../../include/climate_hip.h"

namespace {

constexpr int MAXOC = 8;

__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, long long sx,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        float* __restrict__ pred, int C, int OC, int HW) {
  const int n = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  float acc[MAXOC];
#pragma unroll
  for (int o = 0; o < MAXOC; ++o) acc[o] = (o < OC) ? b[o] : 0.f;
  const float* xp = x + (long long)n * sx + p;
  for (int c = 0; c < C; ++c) {
    const float xv = xp[(long long)c * HW];
#pragma unroll
    for (int o = 0; o < MAXOC; ++o)
      if (o < OC) acc[o] += w[o * C + c] * xv;
  }
#pragma unroll
  for (int o = 0; o < MAXOC; ++o)
    if (o < OC) pred[((long long)n * OC + o) * HW + p] = acc[o];
}

__global__ void zero1_kernel(float* p) { *p = 0.f; }

// loss += sum (pred-y)^2 / total ; dpred = 2 (pred - y) / total
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred, const float* __restrict__ y,
                                                   float* __restrict__ loss, float* __restrict__ dpred,
                                                   long long total) {
  __shared__ float red[32];
  const float inv = 1.f / (float)total;
  float a = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const float d = pred[i] - y[i];
    a += d * d;
    if (dpred) dpred[i] = 2.f * d * inv;
  }
  a = block_sum(a, red);
  if (threadIdx.x == 0) unsafeAtomicAdd(loss, a * inv);
}

// dx[n,c,p] = sum_o W[o][c] dpred[n,o,p]; dW[o][c] += sum_{n,p} dpred x; db[o] += sum dpred
// A workgroup walks `ppb` pixels of one sample; per 16-channel chunk every thread keeps its dW partials in registers
// over all its pixels and the cross-lane reduction happens once per (o, c), not once per pixel tile.
constexpr int HB_CC = 16;
template <int OCT>   // OCT >= OC: compile-time bound of the output-channel loops (register arrays sized by it)
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dpred, const float* __restrict__ x,
                                                        long long sx, const float* __restrict__ w,
                                                        float* __restrict__ dx, long long sdx,
                                                        float* __restrict__ dw, float* __restrict__ db, int C,
                                                        int OC, int HW, int ppb) {
  extern __shared__ float sh[];  // [OC*C] dW partial + [OC] db partial + [OC*C] weights
  const int tid = threadIdx.x, lane = tid & 63;
  const int n = blockIdx.y;
  const int p_begin = blockIdx.x * ppb, p_end = min(HW, p_begin + ppb);
  float* wsh = sh + OC * C + OC;
  for (int i = tid; i < OC * C + OC; i += 256) sh[i] = 0.f;
  for (int i = tid; i < OC * C; i += 256) wsh[i] = w[i];
  __syncthreads();
  float dbp[OCT];
#pragma unroll
  for (int o = 0; o < OCT; ++o) dbp[o] = 0.f;
  for (int c0 = 0; c0 < C; c0 += HB_CC) {
    float dwp[OCT][HB_CC];
#pragma unroll
    for (int o = 0; o < OCT; ++o)
#pragma unroll
      for (int j = 0; j < HB_CC; ++j) dwp[o][j] = 0.f;
    for (int p = p_begin + tid; p < p_end; p += 256) {
      float dp[OCT];
#pragma unroll
      for (int o = 0; o < OCT; ++o) dp[o] = (o < OC) ? dpred[((long long)n * OC + o) * HW + p] : 0.f;
      if (c0 == 0) {
#pragma unroll
        for (int o = 0; o < OCT; ++o) dbp[o] += dp[o];
      }
      float xv[HB_CC];
#pragma unroll
      for (int j = 0; j < HB_CC; ++j) {
        const int c = min(c0 + j, C - 1);
        xv[j] = x[(long long)n * sx + (long long)c * HW + p];
      }
#pragma unroll
      for (int j = 0; j < HB_CC; ++j) {
        const int c = c0 + j;
        if (c < C) {
          float g = 0.f;
#pragma unroll
          for (int o = 0; o < OCT; ++o) {
            if (o < OC) {
              g += wsh[o * C + c] * dp[o];
              dwp[o][j] += dp[o] * xv[j];
            }
          }
          dx[(long long)n * sdx + (long long)c * HW + p] = g;
        }
      }
    }
#pragma unroll
    for (int o = 0; o < OCT; ++o) {
      if (o < OC) {
#pragma unroll
        for (int j = 0; j < HB_CC; ++j) {
          if (c0 + j < C) {
            const float sred = wave_sum(dwp[o][j]);
            if (lane == 0) atomicAdd(&sh[o * C + c0 + j], sred);
          }
        }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < OCT; ++o) {
    if (o < OC) {
      const float sred = wave_sum(dbp[o]);
      if (lane == 0) atomicAdd(&sh[OC * C + o], sred);
    }
  }
  __syncthreads();
  for (int i = tid; i < OC * C; i += 256) unsafeAtomicAdd(dw + i, sh[i]);
  for (int i = tid; i < OC; i += 256) unsafeAtomicAdd(db + i, sh[OC * C + i]);
}

// Output head, MSE loss and the head's backward in ONE pass over the last decoder activation (fused training step
// only): per pixel pred = b + W x (same summation order as head_fwd_kernel), d = pred - y, loss += d^2 / total,
// dpred = 2 d / total, then dx / dW / db exactly as head_bwd_kernel.  x is read from HBM once (the second, chunked
// walk over the channels hits the cache) and pred / dpred never exist in memory unless `pred_out` asks for them.
template <int OCT>
__global__ __launch_bounds__(256) void head_mse_bwd_kernel(const float* __restrict__ x, long long sx,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            const float* __restrict__ y, float* __restrict__ pred_out,
                                                            float* __restrict__ loss, float* __restrict__ dx,
                                                            long long sdx, float* __restrict__ dw,
                                                            float* __restrict__ db, int C, int OC, int HW, int ppb,
                                                            float inv_total) {
  extern __shared__ float sh[];  // [OC*C] dW partial + [OC] db partial + [OC*C] weights + [OC] bias
  __shared__ float red[32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int n = blockIdx.y;
  const int p_begin = blockIdx.x * ppb, p_end = min(HW, p_begin + ppb);
  float* wsh = sh + OC * C + OC;
  float* bsh = wsh + OC * C;
  for (int i = tid; i < OC * C + OC; i += 256) sh[i] = 0.f;
  for (int i = tid; i < OC * C; i += 256) wsh[i] = w[i];
  for (int i = tid; i < OC; i += 256) bsh[i] = b[i];
  __syncthreads();
  float dbp[OCT], lsum = 0.f;
  // One pixel per thread (the host launches ppb = 256): pred, the loss term and dpred are computed ONCE, then the
  // channels are walked in chunks of HB_CC whose dW partials live in registers (the chunked walk re-reads x from the
  // cache).  (The first version recomputed pred inside every chunk: C / HB_CC + 1 passes over the channels.)
  const int p = p_begin + tid;
  const bool live = p < p_end;
  const float* xp = x + (long long)n * sx + (live ? p : p_begin);
  float dp[OCT];
  {
    float acc[OCT];
#pragma unroll
    for (int o = 0; o < OCT; ++o) acc[o] = (o < OC) ? bsh[o] : 0.f;
    if (live) {
      for (int c = 0; c < C; ++c) {
        const float xv = xp[(long long)c * HW];
#pragma unroll
        for (int o = 0; o < OCT; ++o)
          if (o < OC) acc[o] += wsh[o * C + c] * xv;
      }
    }
#pragma unroll
    for (int o = 0; o < OCT; ++o) {
      float d = 0.f;
      if (o < OC && live) {
        d = acc[o] - y[((long long)n * OC + o) * HW + p];
        lsum += d * d;
        if (pred_out) pred_out[((long long)n * OC + o) * HW + p] = acc[o];
      }
      dp[o] = 2.f * d * inv_total;
      dbp[o] = dp[o];
    }
  }
  for (int c0 = 0; c0 < C; c0 += HB_CC) {
    float dwp[OCT][HB_CC];
#pragma unroll
    for (int j = 0; j < HB_CC; ++j) {
      const int c = c0 + j;
      float xv = 0.f;
      if (c < C && live) {
        xv = xp[(long long)c * HW];
        float g = 0.f;
#pragma unroll
        for (int o = 0; o < OCT; ++o)
          if (o < OC) g += wsh[o * C + c] * dp[o];
        dx[(long long)n * sdx + (long long)c * HW + p] = g;
      }
#pragma unroll
      for (int o = 0; o < OCT; ++o) dwp[o][j] = dp[o] * xv;
    }
#pragma unroll
    for (int o = 0; o < OCT; ++o) {
      if (o < OC) {
#pragma unroll
        for (int j = 0; j < HB_CC; ++j) {
          if (c0 + j < C) {
            const float sred = wave_sum(dwp[o][j]);
            if (lane == 0) atomicAdd(&sh[o * C + c0 + j], sred);
          }
        }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < OCT; ++o) {
    if (o < OC) {
      const float sred = wave_sum(dbp[o]);
      if (lane == 0) atomicAdd(&sh[OC * C + o], sred);
    }
  }
  lsum = block_sum(lsum, red);
  __syncthreads();
  for (int i = tid; i < OC * C; i += 256) unsafeAtomicAdd(dw + i, sh[i]);
  for (int i = tid; i < OC; i += 256) unsafeAtomicAdd(db + i, sh[OC * C + i]);
  if (tid == 0) unsafeAtomicAdd(loss, lsum * inv_total);
}

}  // namespace

extern "C" {

int cm_head_fwd(const float* x, long long sx, const float* w, const float* b, float* pred, int n, int c, int oc,
                int hw, cm_stream stream) {
  if (n <= 0 || c <= 0 || oc <= 0 || oc > MAXOC || hw <= 0) return -22;
  head_fwd_kernel<<<dim3(cdiv(hw, 256), n), 256, 0, (hipStream_t)stream>>>(x, sx, w, b, pred, c, oc, hw);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_mse_loss(const float* pred, const float* y, float* loss, float* dpred, long long total, cm_stream stream) {
  if (total <= 0) return -22;
  zero1_kernel<<<1, 1, 0, (hipStream_t)stream>>>(loss);   // a kernel node, not a memset node (see cm_zero)
  CM_CHECK_LAUNCH();
  long long blocks = (total + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  mse_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(pred, y, loss, dpred, total);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_head_bwd(const float* dpred, const float* x, long long sx, const float* w, float* dx, long long sdx, float* dw,
                float* db, int n, int c, int oc, int hw, cm_stream stream) {
  if (n <= 0 || c <= 0 || oc <= 0 || oc > MAXOC || hw <= 0) return -22;
  int parts = 1;
  while (n * parts < 256 && hw / (parts * 2) >= 256) parts *= 2;   // >= 256 pixels per workgroup
  const int ppb = cdiv(hw, parts);
  const dim3 grid(cdiv(hw, ppb), n);
  const size_t lds = (size_t)(2 * oc * c + oc) * sizeof(float);
  if (oc <= 2)
    head_bwd_kernel<2><<<grid, 256, lds, (hipStream_t)stream>>>(dpred, x, sx, w, dx, sdx, dw, db, c, oc, hw, ppb);
  else if (oc <= 4)
    head_bwd_kernel<4><<<grid, 256, lds, (hipStream_t)stream>>>(dpred, x, sx, w, dx, sdx, dw, db, c, oc, hw, ppb);
  else
    head_bwd_kernel<MAXOC><<<grid, 256, lds, (hipStream_t)stream>>>(dpred, x, sx, w, dx, sdx, dw, db, c, oc, hw, ppb);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_head_mse_bwd(const float* x, long long sx, const float* w, const float* b, const float* y, float* pred_out,
                    float* loss, float* dx, long long sdx, float* dw, float* db, int n, int c, int oc, int hw,
                    cm_stream stream) {
  if (n <= 0 || c <= 0 || oc <= 0 || oc > MAXOC || hw <= 0) return -22;
  const int ppb = 256;                                   // one pixel per thread: pred is computed once per pixel
  const dim3 grid(cdiv(hw, ppb), n);
  const size_t lds = (size_t)(2 * oc * c + 2 * oc) * sizeof(float);
  const float inv_total = 1.f / ((float)n * (float)oc * (float)hw);
#define CM_HMB(T)                                                                                                  \
  head_mse_bwd_kernel<T><<<grid, 256, lds, (hipStream_t)stream>>>(x, sx, w, b, y, pred_out, loss, dx, sdx, dw, db, \
                                                                  c, oc, hw, ppb, inv_total)
  if (oc <= 2) CM_HMB(2);
  else if (oc <= 4) CM_HMB(4);
  else CM_HMB(MAXOC);
#undef CM_HMB
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
