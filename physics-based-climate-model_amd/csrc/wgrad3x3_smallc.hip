// Weight gradient of a 3x3 / pad 1 convolution with VERY FEW input channels (C_in * 9 <= 64): the network's first
// layer, 5 forcing variables -> base channels (reference: nn.Conv2d(in_ch, base, 3, padding=1), src/unet.py:36 as
// instantiated at src/unet_convlstm_attention.py:35).
//
// The general kernels tile 32 input channels per MFMA column block, so C_in = 5 wastes 84 % of their work (measured
// 105-124 us for 1.9 GFLOP).  Here the GEMM columns are the (input channel, tap) PAIRS -- 45 of 64 columns used:
//     D[co][(ci,tap)] += sum_{n,y,x} dY[n,co,y,x] * X[n,ci,y+dy,x+dx]
// rows = 32 output channels (A = dY), reduction = pixels, two at a time on v_mfma_f32_32x32x2_f32 (exact fp32; the
// layer is bound by reading dY once, not by arithmetic).  A workgroup stages a band of R rows of one sample: the dY
// tile [32][R*W] and the zero-padded input tile [C_in][R+2][W+2]; a lane's B operand is its column's (ci, tap) base
// offset plus the pixel offset, so the nine shifts are plain LDS addressing.  The four waves split the band's pixel
// pairs, workgroups loop over (sample, band) units and keep their accumulators, and each workgroup stores ONE
// 32 x 64 partial tile; cm_wgrad3x3_smallc's second launch folds the partials into the tap-major staging tensor
// (thousands of workgroups adding into the same 1440 addresses would serialise in the L2, cf. conv7_bwd).
#include "common.h"
#include "../../include/climate_hip.h"

namespace {

constexpr int SC_MAXPX = 320;   // pixels of one band (R * W)

struct ScArgs {
  const float* x;
  long long sx;
  const float* dy;
  long long sdy;
  float* part;     // [gridDim.y][gridDim.x][32][64] partial tiles
  int N, H, W, Cin, Cout, R, nbands, nunits;
};

__global__ __launch_bounds__(256) void wgrad3x3_smallc_kernel(ScArgs a) {
  extern __shared__ float sh[];
  const int W = a.W, H = a.H, HW = H * W, R = a.R, Cin = a.Cin;
  const int PW = W + 2, XROWS = R + 2, XPL = XROWS * PW;      // padded input plane
  const int NPX = R * W, DPITCH = NPX | 1;                     // odd pitch: 32 channels hit 32 banks
  float* Dl = sh;                                              // [32][DPITCH]
  float* Xl = sh + 32 * DPITCH;                                // [Cin][XROWS][PW] + one zero pad row
  const int xtot = Cin * XPL;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int co0 = blockIdx.y * 32;

  // column -> (ci, tap) base offset inside Xl; unused columns read the zero row behind the tile
  int jbase[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int j = t * 32 + l31;
    const int ci = j / 9, tap = j % 9;
    jbase[t] = j < Cin * 9 ? (ci * XROWS + tap / 3) * PW + tap % 3 : xtot;
  }
  for (int i = tid; i < PW + 2; i += 256) Xl[xtot + i] = 0.f;

  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int npairs = NPX / 2;                                  // W even (host check) -> a pair never wraps a row
  const int kpw = (npairs + 3) / 4;
  const int k0 = wave * kpw, k1 = min(npairs, k0 + kpw);

  for (int u = blockIdx.x; u < a.nunits; u += gridDim.x) {
    const int n = u / a.nbands, y0 = (u % a.nbands) * R;
    __syncthreads();                                           // previous unit's MFMA phase is done with the tiles
    // ---- dY tile: float4 along the contiguous band (R*W floats per channel; W % 4 == 0 checked on the host) ----
    const int rows = min(R, H - y0), npx = rows * W;
    for (int i = tid; i < 32 * (NPX / 4); i += 256) {
      const int c = i / (NPX / 4), q = i % (NPX / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (co0 + c < a.Cout && q * 4 < npx)
        v = *reinterpret_cast<const float4*>(a.dy + (long long)n * a.sdy + (long long)(co0 + c) * HW + y0 * W + q * 4);
      float* d = Dl + c * DPITCH + q * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    // ---- input tile with halo, zero padded ----
    for (int i = tid; i < xtot; i += 256) {
      const int ci = i / XPL, r = (i % XPL) / PW, cpos = i % PW;
      const int yy = y0 - 1 + r, xx = cpos - 1;
      float v = 0.f;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = a.x[(long long)n * a.sx + (long long)ci * HW + yy * W + xx];
      Xl[i] = v;
    }
    __syncthreads();
    // ---- MFMA phase: this wave's share of the pixel pairs ----
    const float* ap = Dl + l31 * DPITCH + half;
    const float* bp0 = Xl + jbase[0] + (jbase[0] < xtot ? half : 0);
    const float* bp1 = Xl + jbase[1] + (jbase[1] < xtot ? half : 0);
    const int m0 = jbase[0] < xtot ? 1 : 0, m1 = jbase[1] < xtot ? 1 : 0;   // unused columns stay on the zero row
    int row = (2 * k0) / W, col = (2 * k0) % W;                // pixel of the pair's first element, kept incrementally
    for (int k = k0; k < k1; k += 4) {
      float av[4], b0[4], b1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {                            // all operand reads of four k-steps, then their MFMAs
        const bool ok = k + q < k1;
        const int p = ok ? 2 * (k + q) : 2 * k0;
        const int poff = ok ? row * PW + col : ((2 * k0) / W) * PW + (2 * k0) % W;
        av[q] = ok ? ap[p] : 0.f;
        b0[q] = bp0[m0 * poff];
        b1[q] = bp1[m1 * poff];
        col += 2;
        if (col >= W) { col -= W; ++row; }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], b0[q], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], b1[q], acc[1], 0, 0, 0);
      }
    }
  }

  // ---- combine the four waves through LDS, store one partial tile per workgroup ----
  __syncthreads();
  float* red = sh;                                             // [4][2][16][64]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave * 2 + t) * 16 + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  // D[i][j]: lane holds column j = l31, rows (r&3) + 8*(r>>2) + 4*half.  Wave w finishes registers 4w..4w+3.
  float* out = a.part + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * (32 * 64);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 4 * wave + q;
      float v = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 4; ++w2) v += red[((w2 * 2 + t) * 16 + r) * 64 + lane];
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      out[row * 64 + t * 32 + l31] = v;
    }
}

// g[(co*9 + tap)*Ctot + c_off + ci] += sum_blk part[cot][blk][co % 32][ci*9 + tap]
__global__ __launch_bounds__(256) void wgrad3x3_smallc_fold_kernel(const float* __restrict__ part, int nblk,
                                                                   float* __restrict__ g, int Cin, int Cout, int Ctot,
                                                                   int c_off) {
  __shared__ float redf[4][64];
  const int cot = blockIdx.y, row = blockIdx.x;                 // one (cout tile, output channel) per workgroup
  const int j = threadIdx.x & 63, sl = threadIdx.x >> 6;       // 64 columns x 4 slices of the partial tiles
  const float* p = part + ((long long)cot * nblk) * (32 * 64) + row * 64 + j;
  // blockIdx.z takes a contiguous share of the partial tiles (a single workgroup walking all of them is latency bound)
  const int per = (nblk + gridDim.z - 1) / gridDim.z, b0 = blockIdx.z * per, b1 = min(nblk, b0 + per);
  float a0 = 0.f, a1 = 0.f;
  int b = b0 + sl;
  for (; b + 4 < b1; b += 8) {
    a0 += p[(long long)b * (32 * 64)];
    a1 += p[(long long)(b + 4) * (32 * 64)];
  }
  if (b < b1) a0 += p[(long long)b * (32 * 64)];
  a0 += a1;
  redf[sl][j] = a0;
  __syncthreads();
  const int co = cot * 32 + row;
  if (sl == 0 && j < Cin * 9 && co < Cout) {
    const float t = (redf[0][j] + redf[1][j]) + (redf[2][j] + redf[3][j]);
    const int ci = j / 9, tap = j % 9;
    unsafeAtomicAdd(g + ((long long)co * 9 + tap) * Ctot + c_off + ci, t);
  }
}

int sc_rows(int w) { return max(1, SC_MAXPX / w); }

}  // namespace

extern "C" {

/* number of partial-tile floats cm_wgrad3x3_smallc needs as scratch (no initialisation required) */
long long cm_wgrad3x3_smallc_scratch_elems(int n, int h, int w, int cout) {
  if (n <= 0 || h <= 0 || w <= 0 || cout <= 0) return -22;
  const int nunits = n * cdiv(h, sc_rows(w));
  const int nblk = nunits < 768 ? nunits : 768;
  return (long long)cdiv(cout, 32) * nblk * 32 * 64;
}

int cm_wgrad3x3_smallc(const float* x, long long sx, int cin, const float* dy, long long sdy, float* g, int ctot,
                       int c_off, int n, int h, int w, int cout, float* scratch, cm_stream stream) {
  if (n <= 0 || h <= 0 || w <= 0 || cout <= 0 || cin <= 0 || c_off < 0 || c_off + cin > ctot || !scratch) return -22;
  if (cin * 9 > 64 || (w & 3) || w > SC_MAXPX || (sdy & 3) || ((uintptr_t)dy & 15))
    return -22;                                                 // few channels, float4 rows, one row fits a band
  ScArgs a;
  a.x = x; a.sx = sx; a.dy = dy; a.sdy = sdy; a.part = scratch;
  a.N = n; a.H = h; a.W = w; a.Cin = cin; a.Cout = cout;
  a.R = sc_rows(w);
  a.nbands = cdiv(h, a.R);
  a.nunits = n * a.nbands;
  const int nblk = a.nunits < 768 ? a.nunits : 768;            // three resident workgroups per CU
  const int npx = a.R * w, dpitch = npx | 1;
  const size_t tiles = (size_t)32 * dpitch + (size_t)cin * (a.R + 2) * (w + 2) + (w + 4);
  const size_t lds = (tiles > 4 * 2 * 16 * 64 ? tiles : 4 * 2 * 16 * 64) * sizeof(float);
  if (lds > 64 * 1024) return -22;
  const dim3 grid(nblk, cdiv(cout, 32));
  wgrad3x3_smallc_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(a);
  CM_CHECK_LAUNCH();
  wgrad3x3_smallc_fold_kernel<<<dim3(32, cdiv(cout, 32), nblk >= 64 ? 16 : 1), 256, 0, (hipStream_t)stream>>>(
      scratch, nblk, g, cin, cout, ctot, c_off);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
