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
#include <stdlib.h>
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

// Wave roles: wave = (group, kernel row dy); group = (mo, ks): mo selects the 32-cout sub-block, ks the share of the
// unit's pixel pairs.  Every wave owns 3 accumulators (the three taps of its kernel row).
template <int TH, int TW, int S, int MO, int KS, bool DUAL>
__global__ __launch_bounds__(192 * MO * KS, 2) void wgrad3x3_mfma_kernel(WgArgs a) {
  constexpr int THREADS = 192 * MO * KS;
  constexpr int PITCH = TW + 2;
  constexpr int SS = (TH + 2) * PITCH;
  constexpr int CS0 = S * SS;
  constexpr int CS = (CS0 % 2 == 0) ? CS0 + 1 : CS0;  // odd channel pitch: 32 channels hit 32 banks
  constexpr int PIX = S * TH * TW;
  constexpr int PP = (PIX % 2 == 0) ? PIX + 1 : PIX;  // odd cout pitch
  constexpr int RSTEP = (TW % 2) ? 2 : 1;               // tile rows per MFMA-loop iteration
  constexpr int NPAIR = RSTEP * TW / 2;                 // pixel pairs (MFMA k-steps) per iteration
  constexpr int ITERS_PER_SAMPLE = TH / RSTEP;
  constexpr int ITERS = S * ITERS_PER_SAMPLE;
  static_assert(TH % RSTEP == 0, "odd tile widths need an even tile height");

  __shared__ float Xl[32 * CS];
  __shared__ float Dl[32 * MO * PP];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int dyrow = wave % 3, grp = wave / 3;
  const int mo = grp % MO, ks = grp / MO;
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

  f32x16 acc[3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

  // B operand lane bases: channel l31, kernel row dyrow; second k-slice (half) is the next pixel: +1, or +3 when
  // the pair wraps to the next tile row (only possible for odd TW).
  const int bN = l31 * CS + dyrow * PITCH + half;
  const int bW = l31 * CS + dyrow * PITCH + half * (PITCH - TW + 1);
  const int aB = (mo * 32 + l31) * PP + half;

  const int u_begin = blockIdx.x * a.units_per_block;
  const int u_end = min(a.units, u_begin + a.units_per_block);

  // Register prefetch: the global loads of unit u+1 are in flight while unit u is in its MFMA phase.
  // Staging is ROW based: a work item is 4 consecutive columns of one (channel, sample, tile-row); its index is
  // decomposed once per item (not per element) and the 4 elements are loaded through a buffer descriptor whose range
  // check returns 0 for out-of-image / out-of-batch / out-of-channel lanes (offset 0xFFFFFFFF), so there are no
  // validity masks and nothing consumes the loaded values before the LDS store of the next iteration.
  constexpr int XTPR = (PITCH + 3) / 4;                 // items per haloed X row
  constexpr int XROWS = 32 * S * (TH + 2);
  constexpr int XITEMS = XROWS * XTPR;
  constexpr int NIX = (XITEMS + THREADS - 1) / THREADS;
  constexpr int DTPR = (TW + 3) / 4;                    // items per dY row
  constexpr int DROWS = 32 * MO * S * TH;
  constexpr int DITEMS = DROWS * DTPR;
  constexpr int NID = (DITEMS + THREADS - 1) / THREADS;
  float xr[NIX][4], dr[NID][4];
  auto load_unit = [&](int u) {
    int t = u;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int g = t / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH, n0 = g * S;
    const float* xb0 = a_x0 + (long long)n0 * sx0;
    const float* xb1 = DUAL ? a_x1 + (long long)n0 * sx1 : a_x0;
    const float* db = a_dy + (long long)n0 * sdy + (long long)co0 * HW;
    // descriptors over [base, base + 2 GiB): every valid element of this unit lies inside, 0xFFFFFFFF lies outside
    const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb0), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb1), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(db), 0, 0x7FFFFFFF, 0x00020000);
    int tq = tid;
    asm volatile("" : "+v"(tq));   // opaque per call: keeps the per-item index math out of loop-invariant registers
#pragma unroll
    for (int it = 0; it < NIX; ++it) {
      const int item = tq + it * THREADS;
      const int row = item / XTPR, col0 = (item % XTPR) * 4;
      const int c = row / (S * (TH + 2)), rem = row % (S * (TH + 2));
      const int s = rem / (TH + 2), r = rem % (TH + 2);
      const int gy = y0 - 1 + r, gx0 = x0 - 1 + col0;
      const int ch = ci0 + c;
      const bool rowok = (item < XITEMS) && ch < Cin && (n0 + s < N) && gy >= 0 && gy < H;
      // virtual concat: channels [0, C0) come from x0, [C0, Cin) from x1; the other source's lane is out of range (-> 0)
      const bool in1 = DUAL && ch >= C0;
      const int off0 = (s * (int)sx0 + ch * HW + gy * W + gx0) * 4;
      const int off1 = (s * (int)sx1 + (ch - C0) * HW + gy * W + gx0) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = rowok && (col0 + j < PITCH) && (gx0 + j >= 0) && (gx0 + j < W);
        int v = __builtin_amdgcn_raw_buffer_load_b32(rx0, (ok && !in1) ? off0 + 4 * j : -1, 0, 0);
        if (DUAL) v |= __builtin_amdgcn_raw_buffer_load_b32(rx1, (ok && in1) ? off1 + 4 * j : -1, 0, 0);
        xr[it][j] = __int_as_float(v);
      }
    }
#pragma unroll
    for (int it = 0; it < NID; ++it) {
      const int item = tq + it * THREADS;
      const int row = item / DTPR, col0 = (item % DTPR) * 4;
      const int o = row / (S * TH), rem = row % (S * TH);
      const int s = rem / TH, r = rem % TH;
      const int gy = y0 + r, gx0 = x0 + col0;
      const bool rowok = (item < DITEMS) && co0 + o < Cout && (n0 + s < N) && gy < H;
      const int off = (s * (int)sdy + o * HW + gy * W + gx0) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = rowok && (col0 + j < TW) && (gx0 + j < W);
        dr[it][j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, ok ? off + 4 * j : -1, 0, 0));
      }
    }
  };

  if (u_begin < u_end) load_unit(u_begin);
  for (int u = u_begin; u < u_end; ++u) {
    __syncthreads();  // previous unit's MFMA phase has finished reading LDS
    int ts = tid;
    asm volatile("" : "+v"(ts));
#pragma unroll
    for (int it = 0; it < NIX; ++it) {
      const int item = ts + it * THREADS;
      const int row = item / XTPR, col0 = (item % XTPR) * 4;
      const int c = row / (S * (TH + 2)), rem = row % (S * (TH + 2));
      const int base = c * CS + rem * PITCH + col0;
      if (item < XITEMS) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (col0 + j < PITCH) Xl[base + j] = xr[it][j];
      }
    }
#pragma unroll
    for (int it = 0; it < NID; ++it) {
      const int item = ts + it * THREADS;
      const int row = item / DTPR, col0 = (item % DTPR) * 4;
      const int o = row / (S * TH), rem = row % (S * TH);
      const int base = o * PP + rem * TW + col0;
      if (item < DITEMS) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (col0 + j < TW) Dl[base + j] = dr[it][j];
      }
    }
    __syncthreads();
    if (u + 1 < u_end) load_unit(u + 1);
    __builtin_amdgcn_sched_barrier(0);

    // MFMA phase: a runtime loop over row groups (RSTEP tile rows each; 2 for odd TW so that pixel pairs tile the
    // group exactly), the pairs inside a group unrolled with compile-time LDS offsets.  The KS wave groups take
    // alternate row groups.
#pragma unroll 1
    for (int it = ks; it < ITERS; it += KS) {
      const int sidx = it / ITERS_PER_SAMPLE, r = (it % ITERS_PER_SAMPLE) * RSTEP;
      const int xo = sidx * SS + r * PITCH;
      const int dofs = sidx * (TH * TW) + r * TW;
      const float* dlp = Dl + aB + dofs;
      const float* xn = Xl + bN + xo;
      const float* xw = Xl + bW + xo;
#pragma unroll
      for (int j = 0; j < NPAIR; ++j) {
        const int q0 = 2 * j;
        const int f = (q0 / TW) * PITCH + (q0 % TW);
        const bool wrap = (q0 % TW) == TW - 1;
        const float av = dlp[q0];
        float bv[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) bv[d] = wrap ? xw[f + d] : xn[f + d];
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[d], acc[d], 0, 0, 0);
      }
    }
  }

  // ---- accumulate: D[i = cout][j = channel]; lane holds channel ci0 + l31 ----
  const int ch = ci0 + l31;
  if (ch < Cin && u_begin < u_end) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + mo * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co < Cout) {
#pragma unroll
        for (int d = 0; d < 3; ++d)
          unsafeAtomicAdd(a.g + ((long long)co * 9 + dyrow * 3 + d) * a.Ctot + a.c_off + ch, acc[d][r]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Double-buffered variant: one 12-wave workgroup per CU.  Two LDS stages let the LDS stores of unit u+1 and the global
// loads of unit u+2 overlap the MFMA phase of unit u with ONE barrier per unit; the KS wave groups (which split a
// unit's pixel pairs) are summed through LDS at the end so each output element costs one atomic per workgroup.
template <int TH, int TW, int S, int MO, int KS, bool DUAL>
__global__ __launch_bounds__(192 * MO * KS) void wgrad3x3_db_kernel(WgArgs a) {
  constexpr int THREADS = 192 * MO * KS;
  constexpr int PITCH = TW + 2;
  constexpr int SS = (TH + 2) * PITCH;
  constexpr int CS0 = S * SS;
  constexpr int CS = (CS0 % 2 == 0) ? CS0 + 1 : CS0;
  constexpr int PIX = S * TH * TW;
  constexpr int PP = (PIX % 2 == 0) ? PIX + 1 : PIX;
  constexpr int RSTEP = (TW % 2) ? 2 : 1;
  constexpr int NPAIR = RSTEP * TW / 2;
  constexpr int ITERS_PER_SAMPLE = TH / RSTEP;
  constexpr int ITERS = S * ITERS_PER_SAMPLE;
  constexpr int XSZ = 32 * CS, DSZ = 32 * MO * PP, STAGE = XSZ + DSZ;
  static_assert(TH % RSTEP == 0, "odd tile widths need an even tile height");
  static_assert((KS & (KS - 1)) == 0, "KS must be a power of two");
  // (the launcher sizes the dynamic LDS as max(2 * STAGE, (KS/2) * MO * 3 * 3072) floats for the final reduction)

  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int dyrow = wave % 3, grp = wave / 3;
  const int mo = grp % MO, ks = grp / MO;
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

  f32x16 acc[3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

  const int bN = l31 * CS + dyrow * PITCH + half;
  const int bW = l31 * CS + dyrow * PITCH + half * (PITCH - TW + 1);
  const int aB = XSZ + (mo * 32 + l31) * PP + half;

  const int u_begin = blockIdx.x * a.units_per_block;
  const int u_end = min(a.units, u_begin + a.units_per_block);
  if (u_begin >= u_end) return;

  constexpr int XTPR = (PITCH + 3) / 4;
  constexpr int XROWS = 32 * S * (TH + 2);
  constexpr int XITEMS = XROWS * XTPR;
  constexpr int NIX = (XITEMS + THREADS - 1) / THREADS;
  constexpr int DTPR = (TW + 3) / 4;
  constexpr int DROWS = 32 * MO * S * TH;
  constexpr int DITEMS = DROWS * DTPR;
  constexpr int NID = (DITEMS + THREADS - 1) / THREADS;
  float xr[NIX][4], dr[NID][4];

  auto load_unit = [&](int u) {
    int t = u;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    const int g = t / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH, n0 = g * S;
    const float* xb0 = a_x0 + (long long)n0 * sx0;
    const float* xb1 = DUAL ? a_x1 + (long long)n0 * sx1 : a_x0;
    const float* db = a_dy + (long long)n0 * sdy + (long long)co0 * HW;
    const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb0), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb1), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(db), 0, 0x7FFFFFFF, 0x00020000);
    int tq = tid;
    asm volatile("" : "+v"(tq));
#pragma unroll
    for (int it = 0; it < NIX; ++it) {
      const int item = tq + it * THREADS;
      const int row = item / XTPR, col0 = (item % XTPR) * 4;
      const int c = row / (S * (TH + 2)), rem = row % (S * (TH + 2));
      const int s = rem / (TH + 2), r = rem % (TH + 2);
      const int gy = y0 - 1 + r, gx0 = x0 - 1 + col0;
      const int ch = ci0 + c;
      const bool rowok = (item < XITEMS) && ch < Cin && (n0 + s < N) && gy >= 0 && gy < H;
      const bool in1 = DUAL && ch >= C0;
      const int off0 = (s * (int)sx0 + ch * HW + gy * W + gx0) * 4;
      const int off1 = (s * (int)sx1 + (ch - C0) * HW + gy * W + gx0) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = rowok && (col0 + j < PITCH) && (gx0 + j >= 0) && (gx0 + j < W);
        int v = __builtin_amdgcn_raw_buffer_load_b32(rx0, (ok && !in1) ? off0 + 4 * j : -1, 0, 0);
        if (DUAL) v |= __builtin_amdgcn_raw_buffer_load_b32(rx1, (ok && in1) ? off1 + 4 * j : -1, 0, 0);
        xr[it][j] = __int_as_float(v);
      }
    }
#pragma unroll
    for (int it = 0; it < NID; ++it) {
      const int item = tq + it * THREADS;
      const int row = item / DTPR, col0 = (item % DTPR) * 4;
      const int o = row / (S * TH), rem = row % (S * TH);
      const int s = rem / TH, r = rem % TH;
      const int gy = y0 + r, gx0 = x0 + col0;
      const bool rowok = (item < DITEMS) && co0 + o < Cout && (n0 + s < N) && gy < H;
      const int off = (s * (int)sdy + o * HW + gy * W + gx0) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = rowok && (col0 + j < TW) && (gx0 + j < W);
        dr[it][j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, ok ? off + 4 * j : -1, 0, 0));
      }
    }
  };
  auto store_unit = [&](float* stage) {
    int ts = tid;
    asm volatile("" : "+v"(ts));
#pragma unroll
    for (int it = 0; it < NIX; ++it) {
      const int item = ts + it * THREADS;
      const int row = item / XTPR, col0 = (item % XTPR) * 4;
      const int c = row / (S * (TH + 2)), rem = row % (S * (TH + 2));
      const int base = c * CS + rem * PITCH + col0;
      if (item < XITEMS) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (col0 + j < PITCH) stage[base + j] = xr[it][j];
      }
    }
#pragma unroll
    for (int it = 0; it < NID; ++it) {
      const int item = ts + it * THREADS;
      const int row = item / DTPR, col0 = (item % DTPR) * 4;
      const int o = row / (S * TH), rem = row % (S * TH);
      const int base = XSZ + o * PP + rem * TW + col0;
      if (item < DITEMS) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (col0 + j < TW) stage[base + j] = dr[it][j];
      }
    }
  };

  load_unit(u_begin);
  store_unit(lds);
  if (u_begin + 1 < u_end) load_unit(u_begin + 1);
  __syncthreads();
  for (int u = u_begin; u < u_end; ++u) {
    float* cur = lds + ((u - u_begin) & 1) * STAGE;
    float* nxt = lds + (((u - u_begin) & 1) ^ 1) * STAGE;
    if (u + 1 < u_end) store_unit(nxt);        // nxt was last read in the MFMA phase of unit u-1 (before the barrier)
    if (u + 2 < u_end) load_unit(u + 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int it = ks; it < ITERS; it += KS) {
      const int sidx = it / ITERS_PER_SAMPLE, r = (it % ITERS_PER_SAMPLE) * RSTEP;
      const int xo = sidx * SS + r * PITCH;
      const int dofs = sidx * (TH * TW) + r * TW;
      const float* dlp = cur + aB + dofs;
      const float* xn = cur + bN + xo;
      const float* xw = cur + bW + xo;
#pragma unroll
      for (int j = 0; j < NPAIR; ++j) {
        const int q0 = 2 * j;
        const int f = (q0 / TW) * PITCH + (q0 % TW);
        const bool wrap = (q0 % TW) == TW - 1;
        const float av = dlp[q0];
        float bv[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) bv[d] = wrap ? xw[f + d] : xn[f + d];
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[d], acc[d], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- sum the KS wave groups through LDS (tree), then one atomic per output element and workgroup ----
  if (KS > 1) {
    const int slot = mo * 3 + dyrow;                       // wave position inside its group
#pragma unroll
    for (int stride = KS / 2; stride >= 1; stride >>= 1) {
      if (ks >= stride && ks < 2 * stride) {
        float* dst = lds + ((ks - stride) * (MO * 3) + slot) * 3072 + lane;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[(d * 16 + r) * 64] = acc[d][r];
      }
      __syncthreads();
      if (ks < stride) {
        const float* src = lds + (ks * (MO * 3) + slot) * 3072 + lane;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[d][r] += src[(d * 16 + r) * 64];
      }
      __syncthreads();
    }
  }
  const int ch = ci0 + l31;
  if (ks == 0 && ch < Cin) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + mo * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co < Cout) {
#pragma unroll
        for (int d = 0; d < 3; ++d)
          unsafeAtomicAdd(a.g + ((long long)co * 9 + dyrow * 3 + d) * a.Ctot + a.c_off + ch, acc[d][r]);
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

// Batched unpack: descs = (ndesc + 1) records of 8 int64 {g ptr, dw ptr, cout, ctot, -, -, -, first block}.
__global__ void wgrad_unpack_batch_kernel(const long long* __restrict__ descs, int ndesc, float scale) {
  const int d = cm_job_of_block(descs, ndesc);
  const long long* r = descs + d * 8;
  const float* g = reinterpret_cast<const float*>(r[0]);
  float* dw = reinterpret_cast<float*>(r[1]);
  const int cout = (int)r[2], ctot = (int)r[3];
  const int b0 = (int)r[7], nb = (int)descs[(d + 1) * 8 + 7] - b0;
  // one thread per (o, c): nine coalesced tap-major reads, nine consecutive writes
  const long long pairs = (long long)cout * ctot;
  for (long long i = (long long)(blockIdx.x - b0) * blockDim.x + threadIdx.x; i < pairs; i += (long long)nb * blockDim.x) {
    const int c = (int)(i % ctot);
    const long long o = i / ctot;
    float v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) v[t] = g[(o * 9 + t) * ctot + c];
#pragma unroll
    for (int t = 0; t < 9; ++t) dw[i * 9 + t] = scale * v[t];
  }
}

struct WgCfg {
  int th, tw, s, mo, ks, db;   // db = 1: double-buffered 12-wave kernel
};
constexpr WgCfg kWg[] = {
    {8, 24, 1, 1, 2, 0},   // 0: 192 px, 32 couts, pixel pairs split over 2 wave groups
    {8, 12, 1, 2, 1, 0},   // 1:  96 px, 64 couts
    {4, 18, 1, 2, 1, 0},   // 2:  72 px, 64 couts
    {6, 18, 1, 2, 1, 0},   // 3: 108 px, 64 couts
    {6, 9, 2, 2, 1, 0},    // 4: 2 samples x 54 px, 64 couts
    {8, 16, 1, 1, 2, 0},   // 5: generic 128 px, 32 couts
    {4, 36, 1, 1, 2, 0},   // 6: 144 px, 32 couts
    {8, 16, 1, 2, 1, 0},   // 7: generic 128 px, 64 couts
    {6, 9, 4, 2, 2, 0},    // 8: 4 samples x 54 px, 64 couts, 12 waves
    {12, 18, 1, 2, 2, 0},  // 9: 216 px, 64 couts, 12 waves
    {8, 24, 1, 1, 1, 0},   // 10: 192 px, 32 couts, 3 waves
    {8, 16, 1, 1, 1, 0},   // 11: 128 px, 32 couts, 3 waves
    {6, 9, 2, 2, 2, 1},    // 12: db, 2 samples x 54 px, 64 couts
    {6, 18, 1, 2, 2, 1},   // 13: db, 108 px, 64 couts
    {8, 12, 1, 2, 2, 1},   // 14: db, 96 px, 64 couts
    {8, 16, 1, 1, 4, 1},   // 15: db, 128 px, 32 couts
    {8, 24, 1, 1, 4, 1},   // 16: db, 192 px, 32 couts
    {4, 36, 1, 1, 4, 1},   // 17: db, 144 px, 32 couts
    {6, 9, 2, 1, 4, 1},    // 18: db, 2 samples x 54 px, 32 couts
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
  // Size the grid to the number of resident workgroup slots (256 CUs x occupancy): with `rounds4`/4 rounds of
  // equally loaded workgroups there is no partially filled last round.  a0.units_per_block carries rounds4 (0 = 4).
  constexpr int PITCHc = c.tw + 2, CS0c = c.s * (c.th + 2) * PITCHc, CSc = (CS0c % 2 == 0) ? CS0c + 1 : CS0c;
  constexpr int PIXc = c.s * c.th * c.tw, PPc = (PIXc % 2 == 0) ? PIXc + 1 : PIXc;
  constexpr size_t db_stage = (size_t)2 * (32 * CSc + 32 * c.mo * PPc), db_red = (size_t)(c.ks / 2) * c.mo * 3 * 3072;
  constexpr size_t db_lds = (db_stage > db_red ? db_stage : db_red) * sizeof(float);
  static int occ = 0;
  if (occ == 0) {
    int nb = 0;
    if (c.db) {
      if (hipFuncSetAttribute((const void*)wgrad3x3_db_kernel<c.th, c.tw, c.s, c.mo, c.ks, DUAL>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)db_lds) != hipSuccess)
        return -22;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, wgrad3x3_db_kernel<c.th, c.tw, c.s, c.mo, c.ks, DUAL>,
                                                       192 * c.mo * c.ks, db_lds) != hipSuccess || nb < 1)
        nb = 1;
    } else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, wgrad3x3_mfma_kernel<c.th, c.tw, c.s, c.mo, c.ks, DUAL>,
                                                            192 * c.mo * c.ks, 0) != hipSuccess || nb < 1) {
      nb = 1;
    }
    occ = nb;
  }
  const int rounds4 = a0.units_per_block > 0 ? a0.units_per_block : 4;
  long long target = (long long)256 * occ * rounds4 / 4;        // workgroups wanted in total
  int p = (int)(target / (gy * gz));                            // pixel splits per output tile
  if (p < 1) p = 1;
  int upb = cdiv(a.units, p);
  if (upb < 1) upb = 1;
  a.units_per_block = upb;
  dim3 grid(cdiv(a.units, upb), gy, gz);
  if (c.db)
    wgrad3x3_db_kernel<c.th, c.tw, c.s, c.mo, c.ks, DUAL><<<grid, 192 * c.mo * c.ks, db_lds, st>>>(a);
  else
    wgrad3x3_mfma_kernel<c.th, c.tw, c.s, c.mo, c.ks, DUAL><<<grid, 192 * c.mo * c.ks, 0, st>>>(a);
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
    case 7: return launch_wg<7, DUAL>(a, st);
    case 8: return launch_wg<8, DUAL>(a, st);
    case 9: return launch_wg<9, DUAL>(a, st);
    case 10: return launch_wg<10, DUAL>(a, st);
    case 11: return launch_wg<11, DUAL>(a, st);
    case 12: return launch_wg<12, DUAL>(a, st);
    case 13: return launch_wg<13, DUAL>(a, st);
    case 14: return launch_wg<14, DUAL>(a, st);
    case 15: return launch_wg<15, DUAL>(a, st);
    case 16: return launch_wg<16, DUAL>(a, st);
    case 17: return launch_wg<17, DUAL>(a, st);
    case 18: return launch_wg<18, DUAL>(a, st);
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
  a.tiles_x = a.tiles_y = a.units = 0;
  a.units_per_block = 0;

  if (config < 0) config = cm_wgrad3x3_pick_config(n, h, w, cout);
  if ((config >> 8) > 0) a.units_per_block = config >> 8;   // bits 8.. = grid size in quarter rounds of resident slots
  config &= 0xff;
  return c1 > 0 ? dispatch_wg<true>(config, a, (hipStream_t)stream) : dispatch_wg<false>(config, a, (hipStream_t)stream);
}

int cm_wgrad3x3_unpack_batch(const void* descs_dev, int ndesc, int total_blocks, float scale, cm_stream stream) {
  if (ndesc <= 0 || total_blocks <= 0) return -22;
  wgrad_unpack_batch_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>((const long long*)descs_dev, ndesc, scale);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_wgrad3x3_unpack(const float* g, float* dw, int cout, int ctot, float scale, cm_stream stream) {
  const long long total = (long long)cout * ctot * 9;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  wgrad_unpack_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(g, dw, cout, ctot, scale);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
