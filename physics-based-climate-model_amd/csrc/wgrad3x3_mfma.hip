// Weight gradient of the 3x3 / pad 1 convolution on the fp32 matrix cores.
//
//   dW[o][c][tap] = sum_{n,y,x} dY[n,o,y,x] * X[n,c,y+dy-1,x+dx-1]
//
// i.e. the `convolution_backward` weight branch behind nn.Conv2d (reference: src/unet.py:36,38 and
// src/convlstm.py:9 under loss.backward(), main_final.py:556-561).  GEMM view: rows = output channels (A operand =
// dY), columns = input channels of one tap (B operand = shifted X), reduction = pixels.  A workgroup of 3 waves owns
// a [32*MO couts] x [32 input channels] x [9 taps] block of dW and a share of the pixels: wave w handles the three
// taps of kernel row dy = w.  Per pixel tile it stages X [32][S][TH+2][TW+2] and dY [32*MO][S*TH*TW] in LDS and
// walks the pixels two at a time (one 32x32x2 MFMA k-step).  Partial blocks are accumulated with fp32 atomics into
// a tap-major staging buffer G[o][tap][c] (32 consecutive channels = one 128-byte segment per half-wave, the access
// shape MI355X float atomics run at full rate on), which cm_wgrad3x3_unpack then transposes into the parameter layout.
#include "common.h"
#include "../../include/climate_hip.h"

namespace {

struct WgArgs {
  const float* x0;
  const float* x1;
  long long sx0, sx1;
  int C0, C1;
  const float* dy;
  long long sdy;
  float* g;       // [Cout][9][Ctot]
  int Ctot, c_off;  // channel count / offset of this conv's input range inside the full weight
  int N, H, W, Cout;
  int tiles_x, tiles_y, units, units_per_block;
};

template <int TH, int TW, int S, int MO, bool DUAL>
__global__ __launch_bounds__(192) void wgrad3x3_mfma_kernel(WgArgs a) {
  constexpr int THREADS = 192;
  constexpr int PITCH = TW + 2;
  constexpr int SS = (TH + 2) * PITCH;
  constexpr int CS0 = S * SS;
  constexpr int CS = (CS0 % 2 == 0) ? CS0 + 1 : CS0;  // odd channel pitch: 32 channels hit 32 banks
  constexpr int PIX = S * TH * TW;
  constexpr int PP = (PIX % 2 == 0) ? PIX + 1 : PIX;  // odd cout pitch
  constexpr int XN = 32 * CS0;                        // elements staged (pitch CS in LDS)
  constexpr int DN = 32 * MO * PIX;
  constexpr int NLX = (XN + THREADS - 1) / THREADS;
  constexpr int NLD = (DN + THREADS - 1) / THREADS;
  static_assert((TH * TW) % 2 == 0, "pixel pairs must not straddle samples");

  __shared__ float Xl[32 * CS];
  __shared__ float Dl[32 * MO * PP];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;  // wave == kernel row dy
  const int l31 = lane & 31, half = lane >> 5;
  const int HW = a.H * a.W;
  const int co0 = blockIdx.y * 32 * MO;
  const int ci0 = blockIdx.z * 32;
  const int Cin = a.C0 + a.C1;

  const float* const a_x0 = a.x0;
  const float* const a_x1 = a.x1;
  const float* const a_dy = a.dy;
  const long long sx0 = a.sx0, sx1 = a.sx1, sdy = a.sdy;
  const int C0 = a.C0, N = a.N, H = a.H, W = a.W, Cout = a.Cout;
  const int tiles_x = a.tiles_x, tiles_y = a.tiles_y;

  f32x16 acc[MO][3];
#pragma unroll
  for (int m = 0; m < MO; ++m)
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][d][r] = 0.f;

  // B operand lane bases: channel l31, kernel row = wave; second k-slice (half) is the next pixel: +1, or +3 when
  // the pair wraps to the next tile row (only possible for odd TW).
  const int bN = l31 * CS + wave * PITCH + half;
  const int bW = l31 * CS + wave * PITCH + half * (PITCH - TW + 1);
  const int aB = l31 * PP + half;

  const int u_begin = blockIdx.x * a.units_per_block;
  const int u_end = min(a.units, u_begin + a.units_per_block);

  for (int u = u_begin; u < u_end; ++u) {
    int t = u;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int g = t / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH, n0 = g * S;

    __syncthreads();  // previous unit's MFMA phase has finished reading LDS
    // ---- stage X: 32 channels x S samples x haloed tile (zero outside image / batch / channel range) ----
#pragma unroll 4
    for (int i = 0; i < NLX; ++i) {
      const int e = tid + i * THREADS;
      const int c = e / CS0, r1 = e % CS0;
      const int s = r1 / SS, r2 = r1 % SS;
      const int row = r2 / PITCH, col = r2 % PITCH;
      const int gy = y0 - 1 + row, gx = x0 - 1 + col;
      const int ch = ci0 + c;
      const bool ok = (e < XN) && ch < Cin && (n0 + s < N) && gy >= 0 && gy < H && gx >= 0 && gx < W;
      float v = 0.f;
      if (ok) {
        if (!DUAL || ch < C0)
          v = a_x0[(long long)(n0 + s) * sx0 + (long long)ch * HW + gy * W + gx];
        else
          v = a_x1[(long long)(n0 + s) * sx1 + (long long)(ch - C0) * HW + gy * W + gx];
      }
      if (e < XN) Xl[c * CS + r1] = v;
    }
    // ---- stage dY: 32*MO couts x PIX pixels (zero outside the image / batch) ----
#pragma unroll 4
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + i * THREADS;
      const int o = e / PIX, q = e % PIX;
      const int s = q / (TH * TW), rem = q % (TH * TW);
      const int py = rem / TW, px = rem % TW;
      const int gy = y0 + py, gx = x0 + px, co = co0 + o;
      const bool ok = (e < DN) && co < Cout && (n0 + s < N) && gy < H && gx < W;
      const float v = ok ? a_dy[(long long)(n0 + s) * sdy + (long long)co * HW + gy * W + gx] : 0.f;
      if (e < DN) Dl[o * PP + q] = v;
    }
    __syncthreads();

#pragma unroll
    for (int kp = 0; kp < PIX / 2; ++kp) {
      constexpr int TT = TH * TW;
      const int q0 = 2 * kp;
      const int f = (q0 / TT) * SS + ((q0 % TT) / TW) * PITCH + (q0 % TW);  // compile-time after unrolling
      const bool wrap = (q0 % TW) == TW - 1;
      const int bb = (wrap ? bW : bN) + f;
      float av[MO], bv[3];
#pragma unroll
      for (int m = 0; m < MO; ++m) av[m] = Dl[aB + m * 32 * PP + q0];
#pragma unroll
      for (int d = 0; d < 3; ++d) bv[d] = Xl[bb + d];
#pragma unroll
      for (int m = 0; m < MO; ++m)
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[m][d] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[d], acc[m][d], 0, 0, 0);
    }
  }

  // ---- accumulate: D[i = cout][j = channel]; lane holds channel ci0 + l31 ----
  const int ch = ci0 + l31;
  if (ch < Cin && u_begin < u_end) {
#pragma unroll
    for (int m = 0; m < MO; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < Cout) {
#pragma unroll
          for (int d = 0; d < 3; ++d)
            unsafeAtomicAdd(a.g + ((long long)co * 9 + wave * 3 + d) * a.Ctot + a.c_off + ch, acc[m][d][r]);
        }
      }
  }
}

// G[o][tap][c] -> dW[o][c][tap]   (optionally scaled)
__global__ void wgrad_unpack_kernel(const float* __restrict__ g, float* __restrict__ dw, int cout, int ctot,
                                    float scale) {
  const long long total = (long long)cout * ctot * 9;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % 9);
    const long long oc = i / 9;
    const int c = (int)(oc % ctot);
    const int o = (int)(oc / ctot);
    dw[i] = scale * g[((long long)o * 9 + tap) * ctot + c];
  }
}

struct WgCfg {
  int th, tw, s, mo;
};
constexpr WgCfg kWg[] = {
    {8, 24, 1, 1},   // 0: 192 px
    {8, 12, 1, 2},   // 1:  96 px, 64 couts
    {4, 18, 1, 2},   // 2:  72 px, 64 couts
    {6, 18, 1, 2},   // 3: 108 px, 64 couts
    {6, 9, 2, 2},    // 4: 2 samples x 54 px, 64 couts
    {8, 16, 1, 1},   // 5: generic 128 px
    {4, 36, 1, 1},   // 6: 144 px
};
constexpr int kNumWg = sizeof(kWg) / sizeof(kWg[0]);

template <int I, bool DUAL>
int launch_wg(const WgArgs& a0, hipStream_t st) {
  constexpr WgCfg c = kWg[I];
  WgArgs a = a0;
  a.tiles_x = cdiv(a.W, c.tw);
  a.tiles_y = cdiv(a.H, c.th);
  a.units = a.tiles_x * a.tiles_y * cdiv(a.N, c.s);
  const int gy = cdiv(a.Cout, 32 * c.mo), gz = cdiv(a.C0 + a.C1, 32);
  // aim for ~4 resident-block rounds worth of workgroups, but keep at least 4 units per block to amortise atomics
  int p = cdiv(2048, gy * gz);
  if (p < 1) p = 1;
  int upb = cdiv(a.units, p);
  if (upb < 4) upb = a.units < 4 ? a.units : 4;
  a.units_per_block = upb;
  dim3 grid(cdiv(a.units, upb), gy, gz);
  wgrad3x3_mfma_kernel<c.th, c.tw, c.s, c.mo, DUAL><<<grid, 192, 0, st>>>(a);
  CM_CHECK_LAUNCH();
  return 0;
}

template <bool DUAL>
int dispatch_wg(int cfg, const WgArgs& a, hipStream_t st) {
  switch (cfg) {
    case 0: return launch_wg<0, DUAL>(a, st);
    case 1: return launch_wg<1, DUAL>(a, st);
    case 2: return launch_wg<2, DUAL>(a, st);
    case 3: return launch_wg<3, DUAL>(a, st);
    case 4: return launch_wg<4, DUAL>(a, st);
    case 5: return launch_wg<5, DUAL>(a, st);
    case 6: return launch_wg<6, DUAL>(a, st);
    default: return -22;
  }
}

}  // namespace

extern "C" {

int cm_wgrad3x3_num_configs(void) { return kNumWg; }

int cm_wgrad3x3_pick_config(int n, int h, int w, int cout) {
  (void)n;
  int best = 5;
  double bc = 1e300;
  for (int i = 0; i < kNumWg; ++i) {
    const WgCfg& c = kWg[i];
    if (cout <= 32 && c.mo > 1) continue;
    const double cover = (double)cdiv(w, c.tw) * c.tw * cdiv(h, c.th) * c.th * ((c.s > 1) ? 1.0 : 1.0);
    const double waste = cover / ((double)h * w);
    const double halo = (double)(c.th + 2) * (c.tw + 2) / (c.th * c.tw);
    const double stage = (32.0 * halo + 32.0 * c.mo) / (32.0 * c.mo * 9.0);  // staged floats per MFMA column
    const double cost = waste * (1.0 + 6.0 * stage) * (c.s * c.th * c.tw < 96 ? 1.15 : 1.0);
    if (cost < bc) {
      bc = cost;
      best = i;
    }
  }
  return best;
}

int cm_wgrad3x3(const float* x0, long long sx0, int c0, const float* x1, long long sx1, int c1, const float* dy,
                long long sdy, float* g, int ctot, int c_off, int n, int h, int w, int cout, int config,
                cm_stream stream) {
  if (n <= 0 || h <= 0 || w <= 0 || cout <= 0 || c0 <= 0 || c1 < 0 || c_off < 0 || c_off + c0 + c1 > ctot) return -22;
  WgArgs a;
  a.x0 = x0; a.x1 = x1; a.sx0 = sx0; a.sx1 = sx1; a.C0 = c0; a.C1 = c1;
  a.dy = dy; a.sdy = sdy; a.g = g; a.Ctot = ctot; a.c_off = c_off;
  a.N = n; a.H = h; a.W = w; a.Cout = cout;
  a.tiles_x = a.tiles_y = a.units = a.units_per_block = 0;
  if (config < 0) config = cm_wgrad3x3_pick_config(n, h, w, cout);
  return c1 > 0 ? dispatch_wg<true>(config, a, (hipStream_t)stream) : dispatch_wg<false>(config, a, (hipStream_t)stream);
}

int cm_wgrad3x3_unpack(const float* g, float* dw, int cout, int ctot, float scale, cm_stream stream) {
  const long long total = (long long)cout * ctot * 9;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  wgrad_unpack_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(g, dw, cout, ctot, scale);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
