// ConvTranspose2d(2, stride 2) forward and data gradient as per-pixel GEMMs on the fp32 matrix cores, operands read
// straight from global memory in MFMA fragment order (no LDS): both operands are contiguous along the MFMA row /
// column index, so every fragment load is a coalesced 128-byte segment per half-wave.
//
// Reference: nn.ConvTranspose2d(c_in, c_out, 2, stride=2) (src/unet.py:63,67); weight [C_in][C_out][2][2].
//   forward : Y[(o,k), p] = b[o] + sum_c  W[c][(o,k)] * X[c, p]              rows (o,k) = 4*C_out, K = C_in
//   backward: dX[c, p]    =        sum_(o,k) W[c][(o,k)] * dY[(o,k), p]      rows c = C_in,      K = 4*C_out
// where column p runs over the flattened (sample, input pixel) index and (o,k) addresses output pixel
// (2y + k/2, 2x + k%2) of channel o.  One wave owns a 32-row x 64-column tile (two accumulators sharing the A fragment).
#include "common.h"
#include "../../include/climate_hip.h"

namespace {

constexpr int UNR = 8;   // k-steps (of 2) whose fragment loads are issued before their MFMAs

template <bool BWD>
__global__ __launch_bounds__(256) void convT_mfma_kernel(const float* __restrict__ src, long long ssrc,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ dst, long long sdst, int N, int Ci, int Co,
                                                          int H, int W) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int HW = H * W, Wo = 2 * W, M4 = 4 * Co;
  const int rows = BWD ? Ci : M4;          // GEMM M
  const int K = BWD ? M4 : Ci;             // GEMM K
  const long long NP = (long long)N * HW;
  const int mt = blockIdx.y;
  const long long pt0 = ((long long)blockIdx.x * 4 + wave) * 2;     // first of this wave's two 32-pixel tiles

  // A fragment addressing: forward A[i][kk] = w[kk*M4 + i] (i contiguous); backward A[i][kk] = w[i*M4 + kk]
  const int ai = mt * 32 + l31;
  const bool aok = ai < rows;
  const long long abase = BWD ? (long long)(aok ? ai : 0) * M4 : (aok ? ai : 0);
  const long long astep = BWD ? 1 : M4;

  // B fragment addressing per pixel tile
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

  // backward: the weight rows are contiguous along the REDUCTION index, so the A fragments (lanes = rows) would be
  // strided in global memory; stage a [32 rows][KCHUNK] slab in LDS (coalesced loads, odd pitch) and read from there.
  constexpr int KCHUNK = 256;
  __shared__ float Wl[BWD ? 32 * (KCHUNK + 1) : 1];

  for (int k0 = 0; k0 < K; k0 += 2 * UNR) {
    if (BWD && (k0 % KCHUNK) == 0) {
      __syncthreads();
      for (int e = threadIdx.x; e < 32 * KCHUNK; e += 256) {
        const int r = e / KCHUNK, kk = e % KCHUNK;
        const int row = mt * 32 + r;
        Wl[r * (KCHUNK + 1) + kk] = (row < rows && k0 + kk < K) ? w[(long long)row * M4 + k0 + kk] : 0.f;
      }
      __syncthreads();
    }
    float av[UNR], bv[2][UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int kk = k0 + 2 * u + half;
      const bool kok = kk < K;
      const int kc = kok ? kk : 0;
      float a;
      if (BWD) a = Wl[l31 * (KCHUNK + 1) + (kc % KCHUNK)];
      else a = w[abase + (long long)kc * astep];
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

  // D[i][j]: lane holds column j = l31 (pixel), rows (r&3) + 8*(r>>2) + 4*half
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (!bok[t]) continue;
    const long long gp = (pt0 + t) * 32 + l31;
    const int n = (int)(gp / HW), p = (int)(gp % HW);
    if (!BWD) {
      const int yy = p / W, xx = p % W;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int o = mt * 8 + 2 * g4 + half;      // rows 4*o .. 4*o+3 are this output channel's four taps
        if (o < Co) {
          const float bb = bias ? bias[o] : 0.f;
          float* yp = dst + (long long)n * sdst + (long long)o * 4 * HW + (long long)(2 * yy) * Wo + 2 * xx;
          *reinterpret_cast<float2*>(yp) = make_float2(acc[t][4 * g4] + bb, acc[t][4 * g4 + 1] + bb);
          *reinterpret_cast<float2*>(yp + Wo) = make_float2(acc[t][4 * g4 + 2] + bb, acc[t][4 * g4 + 3] + bb);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (c < Ci) dst[(long long)n * sdst + (long long)c * HW + p] = acc[t][r];
      }
    }
  }
}

}  // namespace

extern "C" {

int cm_convT2x2_fwd(const float* x, long long sx, const float* w, const float* b, float* y, long long sy, int n,
                    int ci, int co, int h, int w_, cm_stream stream) {
  if (n <= 0 || ci <= 0 || co <= 0 || h <= 0 || w_ <= 0 || (sy & 1)) return -22;
  const long long np = (long long)n * h * w_;
  dim3 grid((unsigned)((np + 255) / 256), (unsigned)((4 * co + 31) / 32));
  convT_mfma_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(x, sx, w, b, y, sy, n, ci, co, h, w_);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
