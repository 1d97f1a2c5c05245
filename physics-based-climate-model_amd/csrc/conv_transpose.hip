// ConvTranspose2d(kernel 2, stride 2) of the decoder's Up block: forward, data gradient, weight gradient.
//
// Reference: nn.ConvTranspose2d(c_in, c_out, 2, stride=2) (src/unet.py:63,67); weight layout [C_in][C_out][2][2].
//   y[n,o,2y+dy,2x+dx] = b[o] + sum_c x[n,c,y,x] * W[c,o,dy,dx]
// Non-overlapping: each output pixel has exactly one source pixel, so this is four 1x1 GEMMs with K = C_in.  The forward
// lives in conv_transpose_mfma.hip (MFMA); this file holds the data gradient (VALU, LDS-staged weights: measured faster
// than the MFMA form at these small sizes, whose K = 4*C_out loop is latency bound) and the weight gradient.
#include "common.h"
#include "../../include/climate_hip.h"

namespace {

constexpr int OT = 8;    // input channels per thread (bwd data)
constexpr int KCH = 64;  // reduction-channel chunk whose weight slab is staged in LDS

// dx[n,c,y,x] = sum_{o,k} dy[n,o,2y+ky,2x+kx] * W[c,o,k];  thread <-> input pixel, OT input channels per thread;
// weight slab w[c0..c0+OT)[o chunk][4] staged in LDS as [o][OT][4].
__global__ __launch_bounds__(256) void convT_bwd_data_kernel(const float* __restrict__ dy, long long sdy,
                                                              const float* __restrict__ w, float* __restrict__ dx,
                                                              long long sdx, int Ci, int Co, int H, int W) {
  __shared__ __attribute__((aligned(16))) float wsh[KCH][OT * 4];
  const int HW = H * W, Wo = 2 * W;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y, c0 = blockIdx.z * OT;
  const bool live = p < HW;
  const int pp = live ? p : 0;
  const int yy = pp / W, xx = pp % W;
  float acc[OT];
#pragma unroll
  for (int j = 0; j < OT; ++j) acc[j] = 0.f;
  const float* dp = dy + (long long)n * sdy + (long long)(2 * yy) * Wo + 2 * xx;
  for (int o0 = 0; o0 < Co; o0 += KCH) {
    const int ko = min(KCH, Co - o0);
    __syncthreads();
    for (int i = threadIdx.x; i < ko * OT * 4; i += blockDim.x) {
      const int o = i / (OT * 4), r = i % (OT * 4), j = r / 4, k = r % 4;
      wsh[o][r] = (c0 + j < Ci) ? w[((long long)(c0 + j) * Co + o0 + o) * 4 + k] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int o = 0; o < ko; ++o) {
      const float2 d0 = *reinterpret_cast<const float2*>(dp + (long long)(o0 + o) * 4 * HW);
      const float2 d1 = *reinterpret_cast<const float2*>(dp + (long long)(o0 + o) * 4 * HW + Wo);
      const float4* wr = reinterpret_cast<const float4*>(&wsh[o][0]);
#pragma unroll
      for (int j = 0; j < OT; ++j) {
        const float4 wv = wr[j];
        acc[j] += d0.x * wv.x + d0.y * wv.y + d1.x * wv.z + d1.y * wv.w;
      }
    }
  }
  if (!live) return;
#pragma unroll
  for (int j = 0; j < OT; ++j)
    if (c0 + j < Ci) dx[(long long)n * sdx + (long long)(c0 + j) * HW + p] = acc[j];
}

// dW[c,o,k] += sum_{n,p} x[n,c,p] * dy[n,o,pos_k(p)].  Workgroup = 16 input channels x 16 output channels (one
// thread per (c,o), 4 accumulators), pixel tiles of 64 staged in LDS, pixel range split over blockIdx.z.
constexpr int WT = 16, PT = 64;
__global__ __launch_bounds__(256) void convT_bwd_weight_kernel(const float* __restrict__ x, long long sx,
                                                                const float* __restrict__ dy, long long sdy,
                                                                float* __restrict__ dw, int Ci, int Co, int N, int H,
                                                                int W, int tiles_per_block) {
  __shared__ float xs[WT][PT + 1];
  __shared__ float ds[WT][PT * 4 + 4];
  const int HW = H * W, Wo = 2 * W;
  const int tid = threadIdx.x, cl = tid / WT, ol = tid % WT;
  const int c0 = blockIdx.x * WT, o0 = blockIdx.y * WT;
  const int tiles_per_sample = cdiv(HW, PT);
  const int total_tiles = tiles_per_sample * N;
  const int t0 = blockIdx.z * tiles_per_block, t1 = min(total_tiles, t0 + tiles_per_block);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int t = t0; t < t1; ++t) {
    const int n = t / tiles_per_sample, p0 = (t % tiles_per_sample) * PT;
    __syncthreads();
    for (int i = tid; i < WT * PT; i += 256) {
      const int c = i / PT, pp = i % PT;
      const int p = p0 + pp;
      xs[c][pp] = (c0 + c < Ci && p < HW) ? x[(long long)n * sx + (long long)(c0 + c) * HW + p] : 0.f;
    }
    for (int i = tid; i < WT * PT * 2; i += 256) {  // float2 granules: (o, pixel, row)
      const int o = i / (PT * 2), r = i % (PT * 2), pp = r / 2, ky = r % 2;
      const int p = p0 + pp;
      float2 v = make_float2(0.f, 0.f);
      if (o0 + o < Co && p < HW) {
        const int yy = p / W, xx = p % W;
        v = *reinterpret_cast<const float2*>(dy + (long long)n * sdy + (long long)(o0 + o) * 4 * HW +
                                             (long long)(2 * yy + ky) * Wo + 2 * xx);
      }
      ds[o][pp * 4 + ky * 2] = v.x;
      ds[o][pp * 4 + ky * 2 + 1] = v.y;
    }
    __syncthreads();
#pragma unroll 8
    for (int pp = 0; pp < PT; ++pp) {
      const float xv = xs[cl][pp];
      a0 += xv * ds[ol][pp * 4];
      a1 += xv * ds[ol][pp * 4 + 1];
      a2 += xv * ds[ol][pp * 4 + 2];
      a3 += xv * ds[ol][pp * 4 + 3];
    }
  }
  if (c0 + cl < Ci && o0 + ol < Co && t0 < t1) {
    float* g = dw + ((long long)(c0 + cl) * Co + o0 + ol) * 4;
    unsafeAtomicAdd(g, a0);
    unsafeAtomicAdd(g + 1, a1);
    unsafeAtomicAdd(g + 2, a2);
    unsafeAtomicAdd(g + 3, a3);
  }
}

}  // namespace

extern "C" {

int cm_convT2x2_bwd_data(const float* dy, long long sdy, const float* w, float* dx, long long sdx, int n, int ci,
                         int co, int h, int w_, cm_stream stream) {
  if (n <= 0 || ci <= 0 || co <= 0 || h <= 0 || w_ <= 0) return -22;
  const int hw = h * w_, bs = hw >= 256 ? 256 : 64;
  convT_bwd_data_kernel<<<dim3(cdiv(hw, bs), n, cdiv(ci, OT)), bs, 0, (hipStream_t)stream>>>(dy, sdy, w, dx, sdx, ci,
                                                                                             co, h, w_);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_convT2x2_bwd_weight(const float* x, long long sx, const float* dy, long long sdy, float* dw, int n, int ci,
                           int co, int h, int w_, cm_stream stream) {
  if (n <= 0 || ci <= 0 || co <= 0 || h <= 0 || w_ <= 0) return -22;
  const int total_tiles = cdiv(h * w_, PT) * n;
  const int gx = cdiv(ci, WT), gy = cdiv(co, WT);
  int splits = cdiv(1024, gx * gy);
  if (splits > total_tiles) splits = total_tiles;
  if (splits < 1) splits = 1;
  const int tpb = cdiv(total_tiles, splits);
  convT_bwd_weight_kernel<<<dim3(gx, gy, cdiv(total_tiles, tpb)), 256, 0, (hipStream_t)stream>>>(x, sx, dy, sdy, dw,
                                                                                                 ci, co, n, h, w_, tpb);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
