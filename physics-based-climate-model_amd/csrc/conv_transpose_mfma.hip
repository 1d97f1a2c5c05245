// ConvTranspose2d(kernel 2, stride 2) of the decoder's Up block on the fp32 matrix cores: forward, data gradient and
// weight gradient.
//
// Reference: nn.ConvTranspose2d(c_in, c_out, 2, stride=2) (src/unet.py:63,67); weight [C_in][C_out][2][2].
// Non-overlapping taps: every output pixel has exactly one source pixel, so all three are plain GEMMs over the
// flattened (sample, input pixel) index p, with (o,k) addressing output pixel (2y + k/2, 2x + k%2) of channel o:
//   forward : Y[(o,k), p] = b[o] + sum_c  W[c][(o,k)] * X[c, p]              rows (o,k) = 4*C_out, K = C_in
//   data    : dX[c, p]    =        sum_(o,k) W[c][(o,k)] * dY[(o,k), p]      rows c = C_in,      K = 4*C_out
//   weight  : dW[c][(o,k)] =       sum_p  X[c, p] * dY[(o,k), p]             rows c, cols (o,k), K = N*H*W
//
// These GEMMs are tiny (0.45 GFLOP per decoder level at the benchmark shape) and their cost is launch width and load
// latency, not arithmetic.  forward/data: a workgroup owns a 32-row x 64-pixel tile and its four waves SPLIT THE
// REDUCTION four ways (operands straight from global memory in MFMA fragment order, all loads of a wave in flight at
// once), then combine through LDS.  weight: 64x64 tiles, pixel chunks transposed through LDS, reduction split over
// blockIdx.z with float atomics on 128-byte segments.
#include "common.h"
#include "../../include/climate_hip.h"

namespace {

constexpr int UNR = 16;   // k-steps (of 2) whose fragment loads are issued before their MFMAs

template <bool BWD>
__global__ __launch_bounds__(256) void convT_mfma_kernel(const float* __restrict__ src, long long ssrc,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ dst, long long sdst, int N, int Ci, int Co,
                                                          int H, int W) {
  __shared__ float red[4][2][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int HW = H * W, Wo = 2 * W, M4 = 4 * Co;
  const int rows = BWD ? Ci : M4;          // GEMM M
  const int K = BWD ? M4 : Ci;             // GEMM K
  const long long NP = (long long)N * HW;
  const int mt = blockIdx.y;
  const long long pt0 = (long long)blockIdx.x * 2;     // first of this workgroup's two 32-pixel tiles
  const int kw = ((K + 7) / 8) * 2;                    // this wave's share of the reduction (even)
  const int kbeg = wave * kw, kend = min(K, kbeg + kw);

  // A fragment addressing: forward A[i][kk] = w[kk*M4 + i] (i contiguous); backward A[i][kk] = w[i*M4 + kk]
  const int ai = mt * 32 + l31;
  const bool aok = ai < rows;
  const long long abase = BWD ? (long long)(aok ? ai : 0) * M4 : (aok ? ai : 0);
  const long long astep = BWD ? 1 : M4;

  long long bbase[2];
  bool bok[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const long long gp = (pt0 + t) * 32 + l31;
    bok[t] = gp < NP;
    const long long g2 = bok[t] ? gp : 0;
    const int n = (int)(g2 / HW), p = (int)(g2 % HW);
    if (!BWD) {
      bbase[t] = (long long)n * ssrc + p;                                  // X[n, c, p]: + c*HW
    } else {
      const int yy = p / W, xx = p % W;
      bbase[t] = (long long)n * ssrc + (long long)(2 * yy) * Wo + 2 * xx;  // dY[n, o, 2y+ky, 2x+kx]: + o*4HW + ky*Wo + kx
    }
  }

  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  for (int k0 = kbeg; k0 < kend; k0 += 2 * UNR) {
    float av[UNR], bv[2][UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int kk = k0 + 2 * u + half;
      const bool kok = kk < kend;
      const int kc = kok ? kk : 0;
      const float a = w[abase + (long long)kc * astep];
      av[u] = (kok && aok) ? a : 0.f;
      long long boff;
      if (!BWD) boff = (long long)kc * HW;
      else boff = (long long)(kc >> 2) * 4 * HW + ((kc >> 1) & 1) * Wo + (kc & 1);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float b = src[bbase[t] + boff];
        bv[t][u] = (kok && bok[t]) ? b : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[t][u], acc[t], 0, 0, 0);
  }

  // combine the four partial tiles: wave q finishes accumulator registers 4q..4q+3 of both pixel tiles
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][t][r][lane] = acc[t][r];
  __syncthreads();
  const int g4 = wave;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      v[q] = (red[0][t][4 * g4 + q][lane] + red[1][t][4 * g4 + q][lane]) +
             (red[2][t][4 * g4 + q][lane] + red[3][t][4 * g4 + q][lane]);
    if (!bok[t]) continue;
    const long long gp = (pt0 + t) * 32 + l31;
    const int n = (int)(gp / HW), p = (int)(gp % HW);
    // D[i][j]: lane holds column j = l31 (pixel), rows (r&3) + 8*(r>>2) + 4*half
    if (!BWD) {
      const int yy = p / W, xx = p % W;
      const int o = mt * 8 + 2 * g4 + half;      // rows 4*o .. 4*o+3 are this output channel's four taps
      if (o < Co) {
        const float bb = bias ? bias[o] : 0.f;
        float* yp = dst + (long long)n * sdst + (long long)o * 4 * HW + (long long)(2 * yy) * Wo + 2 * xx;
        *reinterpret_cast<float2*>(yp) = make_float2(v[0] + bb, v[1] + bb);
        *reinterpret_cast<float2*>(yp + Wo) = make_float2(v[2] + bb, v[3] + bb);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = mt * 32 + q + 8 * g4 + 4 * half;
        if (c < Ci) dst[(long long)n * sdst + (long long)c * HW + p] = v[q];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- weight gradient
// Workgroup tile: 64 input channels x 64 (o,k) columns (= 16 output channels), waves 2x2, each a 32x32 MFMA tile.
// Pixel chunks of 32 are loaded coalesced along the pixel index and transposed through LDS (odd pitch).
constexpr int WPX = 32;
__global__ __launch_bounds__(256) void convT_wgrad_mfma_kernel(const float* __restrict__ x, long long sx,
                                                                const float* __restrict__ dy, long long sdy,
                                                                float* __restrict__ dw, int N, int Ci, int Co, int H,
                                                                int W, int chunks_per_block) {
  __shared__ float Xs[64][WPX + 1];
  __shared__ float Ds[64][WPX + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int HW = H * W, Wo = 2 * W, M4 = 4 * Co;
  const long long NP = (long long)N * HW;
  const int c0 = blockIdx.x * 64, o0 = blockIdx.y * 16;
  const long long nchunks = (NP + WPX - 1) / WPX;
  const long long ch0 = (long long)blockIdx.z * chunks_per_block;
  const long long ch1 = min(nchunks, ch0 + chunks_per_block);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // staging roles: X: 8 values per thread (pixel = tid%32, channels tid/32 + 8q); dY: 4 float2 per thread
  const int spx = tid & 31, sgrp = tid >> 5;
  float xr[8];
  float2 dr[4];
  auto load = [&](long long ch) {
    const long long gp = ch * WPX + spx;
    const bool pok = gp < NP;
    const long long g2 = pok ? gp : 0;
    const int n = (int)(g2 / HW), p = (int)(g2 % HW);
    const int yy = p / W, xx = p % W;
    const float* xp = x + (long long)n * sx + p;
    const float* dp = dy + (long long)n * sdy + (long long)(2 * yy) * Wo + 2 * xx;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = c0 + sgrp + 8 * q;
      const bool ok = pok && c < Ci;
      const float v = xp[ok ? (long long)c * HW : 0];
      xr[q] = ok ? v : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int oo = (sgrp + 8 * q) >> 1, ky = (sgrp + 8 * q) & 1;      // 16 channels x 2 rows
      const bool ok = pok && o0 + oo < Co;
      const float2 v = *reinterpret_cast<const float2*>(dp + (ok ? (long long)(o0 + oo) * 4 * HW + ky * Wo : 0));
      dr[q] = ok ? v : make_float2(0.f, 0.f);
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int q = 0; q < 8; ++q) Xs[sgrp + 8 * q][spx] = xr[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int oo = (sgrp + 8 * q) >> 1, ky = (sgrp + 8 * q) & 1;
      Ds[oo * 4 + ky * 2][spx] = dr[q].x;
      Ds[oo * 4 + ky * 2 + 1][spx] = dr[q].y;
    }
  };

  if (ch0 < ch1) load(ch0);
  for (long long ch = ch0; ch < ch1; ++ch) {
    __syncthreads();
    store();
    __syncthreads();
    if (ch + 1 < ch1) load(ch + 1);
#pragma unroll
    for (int k2 = 0; k2 < WPX / 2; ++k2) {
      const float a = Xs[wm * 32 + l31][2 * k2 + half];
      const float b = Ds[wn * 32 + l31][2 * k2 + half];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  if (ch0 >= ch1) return;
  // D[i][j]: lane holds column j = l31, rows (r&3) + 8*(r>>2) + 4*half; dW row c is contiguous along j
  const int j = o0 * 4 + wn * 32 + l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = c0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (c < Ci && j < M4) unsafeAtomicAdd(dw + (long long)c * M4 + j, acc[r]);
  }
}

}  // namespace

extern "C" {

int cm_convT2x2_fwd(const float* x, long long sx, const float* w, const float* b, float* y, long long sy, int n,
                    int ci, int co, int h, int w_, cm_stream stream) {
  if (n <= 0 || ci <= 0 || co <= 0 || h <= 0 || w_ <= 0 || (sy & 1)) return -22;
  const long long np = (long long)n * h * w_;
  dim3 grid((unsigned)((np + 63) / 64), (unsigned)((4 * co + 31) / 32));
  convT_mfma_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(x, sx, w, b, y, sy, n, ci, co, h, w_);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_convT2x2_bwd_data(const float* dy, long long sdy, const float* w, float* dx, long long sdx, int n, int ci,
                         int co, int h, int w_, cm_stream stream) {
  if (n <= 0 || ci <= 0 || co <= 0 || h <= 0 || w_ <= 0 || (sdy & 1)) return -22;
  const long long np = (long long)n * h * w_;
  dim3 grid((unsigned)((np + 63) / 64), (unsigned)((ci + 31) / 32));
  convT_mfma_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(dy, sdy, w, nullptr, dx, sdx, n, ci, co, h, w_);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_convT2x2_bwd_weight(const float* x, long long sx, const float* dy, long long sdy, float* dw, int n, int ci,
                           int co, int h, int w_, cm_stream stream) {
  if (n <= 0 || ci <= 0 || co <= 0 || h <= 0 || w_ <= 0 || (sdy & 1)) return -22;
  const long long nchunks = ((long long)n * h * w_ + WPX - 1) / WPX;
  const int gx = cdiv(ci, 64), gy = cdiv(co, 16);
  long long splits = cdiv(1024, gx * gy);
  if (splits > nchunks) splits = nchunks;
  if (splits < 1) splits = 1;
  const int cpb = (int)((nchunks + splits - 1) / splits);
  convT_wgrad_mfma_kernel<<<dim3(gx, gy, (unsigned)((nchunks + cpb - 1) / cpb)), 256, 0, (hipStream_t)stream>>>(
      x, sx, dy, sdy, dw, n, ci, co, h, w_, cpb);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
