// Forward 3x3 / pad 1 convolution with VERY FEW input channels (C_in * 9 <= 64): the network's first layer
// (nn.Conv2d(in_ch, base, 3, padding=1), src/unet.py:36 as instantiated at src/unet_convlstm_attention.py:35).
//
// The general kernels pad the reduction to 8 or 16 channels per tap; with 5 input channels that is 38-69 % padding
// and the layer really is bound by writing its output.  Here the reduction index is the (input channel, tap) pair
// itself -- K = C_in * 9 (45), two per v_mfma_f32_32x32x2_f32 step (exact fp32):
//     Y[co][p] = b[co] + sum_k W[co][k] * X[ci(k)][y + dy(k)][x + dx(k)]
// rows = 32 output channels with the A fragments (the UNPACKED weight rows, contiguous in k) held in registers for
// the whole kernel, columns = 32 pixels whose B operand is one LDS read at (pixel offset + per-step tap offset).
// A workgroup stages a zero-padded band of R rows of one sample and its four waves take the band's 32-pixel tiles;
// stores are 128-byte rows of one output channel.
#include "common.h"
#include "../../include/climate_hip.h"

namespace {

constexpr int FC_MAXPX = 640;   // pixels of one band (R * W)

struct FcArgs {
  const float* x;
  long long sx;
  const float* w;      // [Cout][Cin][3][3], unpacked
  const float* bias;   // nullable
  float* out;
  long long sto;
  int N, H, W, Cin, Cout, R, nbands, nunits;
};

template <int KS2>   // k-steps of two: ceil(Cin * 9 / 2) <= KS2
__global__ __launch_bounds__(256, 3) void conv3x3_smallc_kernel(FcArgs a) {   // (256 threads, >= 3 waves per EU)
  extern __shared__ float sh[];
  const int W = a.W, H = a.H, HW = H * W, R = a.R, Cin = a.Cin, K = Cin * 9;
  const int PW = W + 2, XROWS = R + 2, XPL = XROWS * PW;
  const int xtot = Cin * XPL;
  float* Xl = sh;                                   // [Cin][XROWS][PW] + a zero row for the padded k indices
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int co0 = blockIdx.y * 32;

  // A fragments: row = output channel co0 + l31, k = 2*s + half -- straight from the parameter tensor
  float av[KS2];
#pragma unroll
  for (int s = 0; s < KS2; ++s) {
    const int k = 2 * s + half;
    const bool ok = k < K && co0 + l31 < a.Cout;
    av[s] = ok ? a.w[(long long)(co0 + l31) * K + k] : 0.f;
  }
  // tap offset of reduction index k inside the padded tile: (ci, dy, dx) are compile-time per step and parity
  auto koff_of = [&](int k) { return k < K ? (k / 9) * XPL + ((k % 9) / 3) * PW + (k % 9) % 3 : -1; };
  for (int i = tid; i < FC_MAXPX + 2 * PW + 4; i += 256) Xl[xtot + i] = 0.f;
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    bv[r] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
  }

  const int NPX = R * W, ntiles = (NPX + 31) / 32;
  for (int u = blockIdx.x; u < a.nunits; u += gridDim.x) {
    const int n = u / a.nbands, y0 = (u % a.nbands) * R;
    const int npx = min(R, H - y0) * W;
    __syncthreads();
    for (int i = tid; i < xtot; i += 256) {
      const int ci = i / XPL, r = (i % XPL) / PW, cpos = i % PW;
      const int yy = y0 - 1 + r, xx = cpos - 1;
      float v = 0.f;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = a.x[(long long)n * a.sx + (long long)ci * HW + yy * W + xx];
      Xl[i] = v;
    }
    __syncthreads();
    for (int t = wave; t < ntiles; t += 4) {
      const int p = t * 32 + l31;                   // this lane's pixel (column of the MFMA tile)
      const bool live = p < npx;
      const int pp = live ? p : 0;
      const int poff = (pp / W) * PW + pp % W;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < KS2; ++s) {
        const int ke = koff_of(2 * s), ko = koff_of(2 * s + 1);
        const int kk = half ? ko : ke;
        const float b = Xl[kk >= 0 ? kk + poff : xtot];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b, acc, 0, 0, 0);
      }
      if (live) {
        float* op = a.out + (long long)n * a.sto + (long long)y0 * W + p;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (co < a.Cout) op[(long long)co * HW] = acc[r] + bv[r];
        }
      }
    }
  }
}

int fc_rows(int w) { return max(1, FC_MAXPX / w); }

}  // namespace

extern "C" {

int cm_conv3x3_smallc(const float* x, long long sx, int cin, const float* w, const float* bias, float* out,
                      long long st_out, int n, int h, int w_, int cout, cm_stream stream) {
  if (n <= 0 || h <= 0 || w_ <= 0 || cout <= 0 || cin <= 0 || cin * 9 > 64 || w_ > FC_MAXPX) return -22;
  FcArgs a;
  a.x = x; a.sx = sx; a.w = w; a.bias = bias; a.out = out; a.sto = st_out;
  a.N = n; a.H = h; a.W = w_; a.Cin = cin; a.Cout = cout;
  a.R = fc_rows(w_);
  a.nbands = cdiv(h, a.R);
  a.nunits = n * a.nbands;
  const int nblk = a.nunits < 768 ? a.nunits : 768;            // three resident workgroups per CU, several units each
  const size_t lds = ((size_t)cin * (a.R + 2) * (w_ + 2) + FC_MAXPX + 2 * (w_ + 2) + 4) * sizeof(float);
  if (lds > 64 * 1024) return -22;
  const dim3 grid(nblk, cdiv(cout, 32));
  if (cin * 9 <= 46)
    conv3x3_smallc_kernel<23><<<grid, 256, lds, (hipStream_t)stream>>>(a);
  else
    conv3x3_smallc_kernel<32><<<grid, 256, lds, (hipStream_t)stream>>>(a);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
