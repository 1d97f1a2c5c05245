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
// latency, not arithmetic.  forward/data: a workgroup owns a 32-row x 32-pixel tile and its eight waves SPLIT THE
// REDUCTION eight ways (operands straight from global memory in MFMA fragment order, all loads of a wave in flight at
// once), then combine through LDS.  weight: 64x64 tiles, pixel chunks transposed through LDS, reduction split over
// blockIdx.z with float atomics on 128-byte segments.
#include "common.h"
#include "../../include/climate_hip.h"

namespace {

constexpr int UNR = 16;   // k-steps (of 2) per round: all fragment loads of a round are issued before its MFMAs
constexpr int NWV = 8;    // waves per workgroup = reduction split

template <bool BWD>
__global__ __launch_bounds__(NWV * 64) void convT_mfma_kernel(const float* __restrict__ src, long long ssrc,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ dst,
                                                               long long sdst, int N, int Ci, int Co, int H, int W) {
  // backward: the weight rows are contiguous along the REDUCTION index, so A fragments (lanes = rows) straight from
  // global memory would touch 32 cache lines per load; each wave stages its [32 rows][2*UNR] slab through LDS with
  // coalesced loads instead.  The slabs are dead when the partial tiles are combined, so `red` aliases them.
  constexpr int SLAB = 32 * (2 * UNR + 1);
  constexpr int RED = NWV * 16 * 64;
  __shared__ float smem[(BWD && NWV * SLAB > RED) ? NWV * SLAB : RED];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int HW = H * W, Wo = 2 * W, M4 = 4 * Co;
  const int rows = BWD ? Ci : M4;          // GEMM M
  const int K = BWD ? M4 : Ci;             // GEMM K
  const long long NP = (long long)N * HW;
  const int mt = blockIdx.y;
  const int kw = ((K + 2 * NWV - 1) / (2 * NWV)) * 2;  // this wave's share of the reduction (even)
  const int kbeg = wave * kw, kend = min(K, kbeg + kw);

  // forward A[i][kk] = w[kk*M4 + i] (i contiguous)
  const int ai = mt * 32 + l31;
  const bool aok = ai < rows;

  const long long gp = (long long)blockIdx.x * 32 + l31;
  const bool bok = gp < NP;
  const int pn = (int)((bok ? gp : 0) / HW), pp = (int)((bok ? gp : 0) % HW);
  const int py = pp / W, px = pp % W;
  // X[n, c, p]: + c*HW          dY[n, o, 2y+ky, 2x+kx]: + o*4HW + ky*Wo + kx
  const long long bbase = BWD ? (long long)pn * ssrc + (long long)(2 * py) * Wo + 2 * px : (long long)pn * ssrc + pp;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  float* slab = smem + wave * SLAB;
  for (int k0 = kbeg; k0 < kend; k0 += 2 * UNR) {
    if (BWD) {
      __builtin_amdgcn_wave_barrier();
      // 2*UNR = 32 reduction indices x 32 rows: lane -> (row parity, k), 16 loads of two rows each
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int r = 2 * q + half, row = mt * 32 + r, kk = k0 + l31;
        const bool ok = row < rows && kk < kend;
        const float v = w[ok ? (long long)row * M4 + kk : 0];
        slab[r * (2 * UNR + 1) + l31] = ok ? v : 0.f;
      }
      __builtin_amdgcn_wave_barrier();
    }
    float av[UNR], bv[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int kk = k0 + 2 * u + half;
      const bool kok = kk < kend;
      const int kc = kok ? kk : 0;
      if (BWD) {
        av[u] = slab[l31 * (2 * UNR + 1) + 2 * u + half];
      } else {
        const float a = w[(long long)kc * M4 + (aok ? ai : 0)];
        av[u] = (kok && aok) ? a : 0.f;
      }
      long long boff;
      if (!BWD) boff = (long long)kc * HW;
      else boff = (long long)(kc >> 2) * 4 * HW + ((kc >> 1) & 1) * Wo + (kc & 1);
      const float b = src[bbase + boff];
      bv[u] = (kok && bok) ? b : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
  }

  // combine the NWV partial tiles: wave q finishes accumulator registers 2q, 2q+1
  __syncthreads();
  float(*red)[16][64] = reinterpret_cast<float(*)[16][64]>(smem);
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  float v[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < NWV; ++k) t += red[k][2 * wave + q][lane];
    v[q] = t;
  }
  if (!bok) return;
  // D[i][j]: lane holds column j = l31 (pixel), rows (r&3) + 8*(r>>2) + 4*half
  if (!BWD) {
    const int g4 = wave >> 1, ky = wave & 1;     // registers 4*g4 + 2*ky + {0,1}: output row 2y+ky of channel o
    const int o = mt * 8 + 2 * g4 + half;
    if (o < Co) {
      const float bb = bias ? bias[o] : 0.f;
      float* yp = dst + (long long)pn * sdst + (long long)o * 4 * HW + (long long)(2 * py + ky) * Wo + 2 * px;
      *reinterpret_cast<float2*>(yp) = make_float2(v[0] + bb, v[1] + bb);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = 2 * wave + q;
      const int c = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (c < Ci) dst[(long long)pn * sdst + (long long)c * HW + pp] = v[q];
    }
  }
}

// ---------------------------------------------------------------------------------------------- weight gradient
// Workgroup tile: 64 input channels x 64 (o,k) columns (= 16 output channels), waves 2x2, each a 32x32 MFMA tile.
// Pixel chunks of 64 are loaded coalesced along the pixel index and transposed through LDS (odd pitch).
constexpr int WPX = 64;
__global__ __launch_bounds__(256) void convT_wgrad_mfma_kernel(const float* __restrict__ x, long long sx,
                                                                const float* __restrict__ dy, long long sdy,
                                                                float* __restrict__ dw, float* __restrict__ db, int N,
                                                                int Ci, int Co, int H, int W, int chunks_per_block) {
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
  float bsum = 0.f;

  // staging roles: X: 16 values per thread (pixel = tid%64, channels tid/64 + 4q); dY: 8 float2 per thread
  const int spx = tid & 63, sgrp = tid >> 6;
  float xr[16];
  float2 dr[8];
  auto load = [&](long long ch) {
    const long long gp = ch * WPX + spx;
    const bool pok = gp < NP;
    const long long g2 = pok ? gp : 0;
    const int n = (int)(g2 / HW), p = (int)(g2 % HW);
    const int yy = p / W, xx = p % W;
    const float* xp = x + (long long)n * sx + p;
    const float* dp = dy + (long long)n * sdy + (long long)(2 * yy) * Wo + 2 * xx;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = c0 + sgrp + 4 * q;
      const bool ok = pok && c < Ci;
      const float v = xp[ok ? (long long)c * HW : 0];
      xr[q] = ok ? v : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int oo = (sgrp + 4 * q) >> 1, ky = (sgrp + 4 * q) & 1;      // 16 channels x 2 rows
      const bool ok = pok && o0 + oo < Co;
      const float2 v = *reinterpret_cast<const float2*>(dp + (ok ? (long long)(o0 + oo) * 4 * HW + ky * Wo : 0));
      dr[q] = ok ? v : make_float2(0.f, 0.f);
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int q = 0; q < 16; ++q) Xs[sgrp + 4 * q][spx] = xr[q];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int oo = (sgrp + 4 * q) >> 1, ky = (sgrp + 4 * q) & 1;
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
    // bias gradient db[o] = sum dY: the input-channel tile 0 workgroups own it (their dY tile is already in LDS)
    if (db && blockIdx.x == 0 && tid < 64) {
      float t = 0.f;
#pragma unroll 16
      for (int k = 0; k < WPX; ++k) t += Ds[tid][k];   // row pitch WPX+1: conflict free
      bsum += t;
    }
  }
  if (ch0 >= ch1) return;
  if (db && blockIdx.x == 0 && tid < 64) {
    // the four taps of an output channel sit in four adjacent lanes
    bsum += __shfl_xor(bsum, 1, 64);
    bsum += __shfl_xor(bsum, 2, 64);
    const int o = o0 + (tid >> 2);
    if ((tid & 3) == 0 && o < Co) unsafeAtomicAdd(db + o, bsum);
  }
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
  dim3 grid((unsigned)((np + 31) / 32), (unsigned)((4 * co + 31) / 32));
  convT_mfma_kernel<false><<<grid, NWV * 64, 0, (hipStream_t)stream>>>(x, sx, w, b, y, sy, n, ci, co, h, w_);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_convT2x2_bwd_data(const float* dy, long long sdy, const float* w, float* dx, long long sdx, int n, int ci,
                         int co, int h, int w_, cm_stream stream) {
  if (n <= 0 || ci <= 0 || co <= 0 || h <= 0 || w_ <= 0 || (sdy & 1)) return -22;
  const long long np = (long long)n * h * w_;
  dim3 grid((unsigned)((np + 31) / 32), (unsigned)((ci + 31) / 32));
  convT_mfma_kernel<true><<<grid, NWV * 64, 0, (hipStream_t)stream>>>(dy, sdy, w, nullptr, dx, sdx, n, ci, co, h, w_);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_convT2x2_bwd_weight(const float* x, long long sx, const float* dy, long long sdy, float* dw, float* db, int n,
                           int ci, int co, int h, int w_, cm_stream stream) {
  if (n <= 0 || ci <= 0 || co <= 0 || h <= 0 || w_ <= 0 || (sdy & 1)) return -22;
  const long long nchunks = ((long long)n * h * w_ + WPX - 1) / WPX;
  const int gx = cdiv(ci, 64), gy = cdiv(co, 16);
  long long splits = cdiv(512, gx * gy);   // float atomics per workgroup (4096) vs serial chunks per workgroup
  if (splits > nchunks) splits = nchunks;
  if (splits < 1) splits = 1;
  const int cpb = (int)((nchunks + splits - 1) / splits);
  convT_wgrad_mfma_kernel<<<dim3(gx, gy, (unsigned)((nchunks + cpb - 1) / cpb)), 256, 0, (hipStream_t)stream>>>(
      x, sx, dy, sdy, dw, db, n, ci, co, h, w_, cpb);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
