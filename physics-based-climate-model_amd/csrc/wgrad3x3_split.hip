// Weight gradient of the 3x3 / pad 1 convolutions on the bf16 matrix cores with fp32-equivalent accuracy ("bf16x6",
// see split_bf16.h and conv3x3_split.hip for the arithmetic and its error analysis).
//
// Same role and output format as wgrad3x3_mfma.hip (reference: autograd of nn.Conv2d at src/unet.py:36,38 and
// src/convlstm.py:9): G[co][tap][ci] += sum_{n,y,x} dY[n,co,y,x] * X[n,ci,y+dy,x+dx], tap-major staging, float atomics.
//
// GEMM orientation: rows = output channels (A = dY), columns = input channels (B = X), reduction = (sample, pixel).
// v_mfma_f32_32x32x16_bf16 wants 8 CONSECUTIVE reduction indices per lane in one 16-byte register group.  Taking 8
// consecutive pixels would make the nine tap shifts misaligned LDS reads, so the 8 indices are 8 SAMPLES of the same
// pixel: an LDS record is (channel, pixel) x 8 samples x bf16 = 16 bytes, a tap shift is a record offset, every
// fragment read is one aligned ds_read_b128.  The 16-deep k-step pairs two adjacent pixels (lane half 0/1).
//
// A workgroup owns one 32*MO x 32 output tile and walks "units" = (8-sample group, column segment of TW pixels, row
// band): per row it stages one new X row (TW+2 columns incl. halo) into a 3-row ring and one dY row, converting to
// three bf16 pieces on the way (8 coalesced dword loads -> 3 records per (channel, pixel)).  Waves = 3 kernel rows dy
// x KS shares of the row's pixel pairs; a wave holds the 3 dx accumulators of its kernel row for MO row tiles.
// Rows/taps outside the image are skipped (wave uniform), not zero padded.  Global loads of the next row are issued
// before the MFMAs of the current one; the pipeline does not drain between units.
#include "common.h"
#include "split_bf16.h"
#include "split_f16.h"
#include "../../include/climate_hip.h"

namespace {

struct WsArgs {
  const float* x0;
  const float* x1;
  long long sx0, sx1;
  int C0, C1;
  const float* dy;
  long long sdy;
  float* g;         // [Cout][9][Ctot]
  int Ctot, c_off;  // channel count / offset of this conv's input range inside the full weight
  int N, H, W, Cout;
  int RB, ngroups, nsegs, nunits;   // rows per band; units = groups x segments x bands
  const unsigned* bex;              // fp16x3: biased exponent of max|x| per sample [N] (0: the sample is all zeros)
  const unsigned* bey;              // ... of max|dy| per sample
};

constexpr int cdiv_c(int a, int b) { return (a + b - 1) / b; }

template <int TW, int MO, int KS, int NP = 3>
struct WsGeom {
  static constexpr int NPAIR = (TW + 1) / 2;
  static constexpr int XCOLS = 2 * NPAIR + 2;   // dY positions 0..2*NPAIR-1 (zero beyond TW); X column = position + dx + 1
  static constexpr int XP = XCOLS | 1;          // odd record pitch: 32 channels hit distinct bank groups
  static constexpr int DP = (2 * NPAIR) | 1;
  static constexpr int BCO = 32 * MO;
  static constexpr int XSLOT = NP * 32 * XP;    // records of one ring slot (NP pieces)
  static constexpr int STAGE = 3 * XSLOT + NP * BCO * DP;
  static constexpr int RED = (KS > 1) ? MO * 3 * 3 * 16 * 64 / 4 : 0;   // one wave set of accumulators, in records
  static constexpr size_t LDS = (size_t)(STAGE > RED ? STAGE : RED) * 16;
};

// LEAN form (KS = 2, MO = 1): no conversion in the MFMA shadow and no fragment ping-pong, which brings the kernel under
// 168 registers so that TWO 6-wave workgroups share a CU (3 waves per SIMD instead of 1.5; the second
// __launch_bounds__ argument is hip-clang's MIN WAVES PER EU).
constexpr bool ws_lean(int mo, int ks) { return mo == 1 && ks == 2; }

// NP = 3: bf16x6; NP = 2: fp16x3 (split_f16.h).  fp16x3 scaling is PER SAMPLE and product balanced.  The reduction runs
// over (sample, pixel) and an LDS record holds 8 samples of one pixel, so samples of very different magnitude meet in
// one accumulator: a left-padded all-zero frame (GroupNorm's rstd = 316 per layer puts its dY 2^28 above a real
// frame's, while its own contribution cancels) next to real frames.  One scale per record group would flush the real
// frames' dY to zero.  Instead every sample n gets its own pair of exact powers of two (a_n for x, b_n for dy) with
// a_n * b_n = P THE SAME FOR ALL SAMPLES, so the accumulators carry one constant scale and no rescaling ever happens:
// with ex_n, ey_n the exponents of the sample's max|x|, max|dy| (published by the forward / data-gradient conv that
// read the same tensors, or by cm_sample_exponents) and E = max_n(ex_n + ey_n), the deficit d_n = E - ex_n - ey_n of a
// sample's largest product is split between its operands: a_n = 2^(140 - ex_n - floor(d_n/2)), b_n = 2^(140 - ey_n -
// ceil(d_n/2)).  Every scaled value stays below 2^14, P = 2^(280 - E), and a sample keeps its 22 bits as long as
// d_n < ~34 (beyond that its products are below fp32 resolution of the sum anyway).
template <int TW, int MO, int KS, bool DUAL, int NP>
__global__ __launch_bounds__(192 * KS, ws_lean(MO, KS) ? 3 : 1) void wgrad3x3_split_kernel(WsArgs a) {
  using G = WsGeom<TW, MO, KS, NP>;
  constexpr int THREADS = 192 * KS;
  constexpr int NPAIR = G::NPAIR, XP = G::XP, DP = G::DP, BCO = G::BCO, XSLOT = G::XSLOT;

  extern __shared__ u32x4 lds[];
  u32x4* const Xl = lds;                // [3 slots][NP pieces][32][XP]
  u32x4* const Dl = lds + 3 * XSLOT;    // [NP pieces][BCO][DP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int dy = wave % 3 - 1, ks = wave / 3;
  const int H = a.H, W = a.W, HW = H * W, N = a.N;

  // this workgroup's input-channel tile (second pointer of a virtual concat when DUAL)
  // blockIdx.x = reduction share: consecutive workgroup ids go to consecutive XCDs, so XCD k works on the sample
  // groups g == k (mod 8) for ALL output tiles and its 4 MiB L2 sees each X / dY row 1x from the fabric, not tiles x.
  const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * BCO;
  const float* xin = a.x0;
  long long sx = a.sx0;
  int cbase = ci0, cx = a.C0;
  if (DUAL && ci0 >= a.C0) {
    xin = a.x1; sx = a.sx1; cbase = ci0 - a.C0; cx = a.C1;
  }
  const float* const dyp = a.dy;
  const long long sdy = a.sdy;

  // zero the whole stage once: pad columns / positions are never written again and must not hold NaN patterns
  for (int i = tid; i < G::STAGE; i += THREADS) lds[i] = u32x4{0u, 0u, 0u, 0u};

  // ---- staging items: (channel, 4 adjacent columns) x 8 samples = 8 dwordx4 loads -> 4 records x 3 pieces.
  // (dword loads of one record each made the texture addresser, not the matrix cores, the bottleneck: a wave's 64
  // lanes touched ~8 cache lines per instruction.)  Items 0..NIX-1 are X quads, NIX.. are dY quads.
  constexpr int NQX = cdiv_c(TW + 2, 4), NQD = cdiv_c(TW, 4);
  constexpr int NIX = 32 * NQX, NI = NIX + BCO * NQD;
  constexpr int NIT = cdiv_c(NI, THREADS);       // items per thread
  constexpr int NUNIT = 16 * NIT;                // conversion units (one split3_pair each) per thread and iteration
  // converting in the MFMA shadow keeps 48 result registers per item alive across the barrier: only where they fit
  constexpr bool LEAN = ws_lean(MO, KS);
  constexpr bool SHADOW = NIT == 1 && MO == 1 && THREADS <= 384 && !LEAN;
  bool it_x[NIT];
  int it_ch[NIT], it_c0[NIT], it_rec[NIT];       // channel inside the tile, first column, first LDS record
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int i = tid + k * THREADS;
    it_x[k] = i < NIX;
    const int r = it_x[k] ? i : i - NIX;
    const int nq = it_x[k] ? NQX : NQD;
    it_ch[k] = r / nq;
    it_c0[k] = (r % nq) * 4;
    it_rec[k] = it_ch[k] * (it_x[k] ? XP : DP) + it_c0[k];
    const bool chok = i < NI && (it_x[k] ? cbase + it_ch[k] < cx : co0 + it_ch[k] < a.Cout);
    if (!chok) it_ch[k] = -1;
  }

  f32x16 acc[MO][3];
#pragma unroll
  for (int m = 0; m < MO; ++m)
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][d][r] = 0.f;

  // ---- unit state (wave uniform) ----
  int u = blockIdx.x;
  bool have = u < a.nunits;
  int n0 = 0, x0 = 0, y0 = 0, y1 = 0, xs = 0, xe = 0, nval = 0;
  int it_off[NIT];        // element offset of the quad inside (sample, row 0); may be -1 at the left image edge
  unsigned it_msk[NIT];   // bit e: element e of the quad is a real pixel of a real channel
  // fp16x3: E = max over ALL samples of (ex + ey) (wave uniform), then the unit's eight scale pairs
  int emax = 0;
  float sxs[8], sys[8];
  if constexpr (NP == 2) {
    float m = 0.f;
    for (int n = lane; n < N; n += 64) {
      const unsigned ex = a.bex[n], ey = a.bey[n];
      m = fmaxf(m, (ex && ey) ? (float)(ex + ey) : 0.f);
    }
    emax = (int)wave_max_nonneg(m);
  }
  auto decode = [&]() {
    const int g = u % a.ngroups, rest = u / a.ngroups;
    const int seg = rest % a.nsegs, band = rest / a.nsegs;
    n0 = g * 8;
    nval = min(8, N - n0);
    x0 = seg * TW;
    y0 = band * a.RB;
    y1 = min(H, y0 + a.RB);
    xs = max(0, y0 - 1);
    xe = min(H - 1, y1);
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int xq = x0 + it_c0[k] - (it_x[k] ? 1 : 0);
      const int lim = it_x[k] ? TW + 2 : TW;
      unsigned m = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (it_ch[k] >= 0 && xq + e >= 0 && xq + e < W && it_c0[k] + e < lim) m |= 1u << e;
      it_msk[k] = m;
      it_off[k] = ((it_x[k] ? cbase : co0) + max(it_ch[k], 0)) * HW + xq;
    }
    if constexpr (NP == 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int n = __builtin_amdgcn_readfirstlane(n0 + min(j, nval - 1));
        const int ex = (int)a.bex[n], ey = (int)a.bey[n];
        const int d = emax - (ex + ey);
        const int fx = 267 - ex - (d >> 1), fy = 267 - ey - ((d + 1) >> 1);
        const bool ok = ex > 0 && ey > 0 && j < nval && fx >= 1 && fx <= 254 && fy >= 1 && fy <= 254;
        sxs[j] = ok ? __uint_as_float((unsigned)fx << 23) : 0.f;
        sys[j] = ok ? __uint_as_float((unsigned)fy << 23) : 0.f;
      }
    }
  };
  if (have) decode();
  int t = 0;
  int crow = -1;     // image row whose dY sits in Dl: computed in the next iteration

  f32x4 lr[NIT][8];        // prefetched quads: [item][sample]
  u32x4 cv[NIT][3][4];     // converted records: [item][piece][element]
  __syncthreads();

  while (true) {
    // ---- issue this iteration's global loads: X row xs+t (ring), dY row xs+t-1 ----
    int xrow = -1, drow = -1;
    if (have) {
      if (xs + t <= xe) xrow = xs + t;
      const int d = xs + t - 1;
      if (d >= y0 && d < y1) drow = d;
      {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          const int row = it_x[k] ? xrow : drow;
          if (row < 0 || it_msk[k] == 0) continue;
          const float* base = it_x[k] ? xin : dyp;
          const long long ss = it_x[k] ? sx : sdy;
          const int cend = (it_x[k] ? cx : a.Cout) * HW;           // end of one sample's channel block
          const int lo = it_off[k] + row * W;
          const float* p = base + (long long)n0 * ss;
          if (lo >= 0 && lo + 4 <= cend) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
              const f4u v = *reinterpret_cast<const f4u*>(p + (long long)min(j, nval - 1) * ss + lo);
              lr[k][j] = f32x4{v.x, v.y, v.z, v.w};
            }
          } else {   // first / last quad of a sample's block: element-wise, clamped (masked elements are zeroed later)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float* q = p + (long long)min(j, nval - 1) * ss;
#pragma unroll
              for (int e = 0; e < 4; ++e) lr[k][j][e] = q[min(max(lo + e, 0), cend - 1)];
            }
          }
        }
      }
    }

    // conversion unit c of this thread: item c/16, element (c/4)%4, sample pair c%4
    auto convert_unit = [&](int c) {
      const int k = c / 16, e = (c / 4) % 4, q = c % 4;
      const bool ok0 = ((it_msk[k] >> e) & 1) && 2 * q < nval, ok1 = ((it_msk[k] >> e) & 1) && 2 * q + 1 < nval;
      unsigned a_, b_, c_ = 0;
      if constexpr (NP == 3) {
        split3_pair(ok0 ? lr[k][2 * q][e] : 0.f, ok1 ? lr[k][2 * q + 1][e] : 0.f, a_, b_, c_);
      } else {
        const float s0 = it_x[k] ? sxs[2 * q] : sys[2 * q], s1 = it_x[k] ? sxs[2 * q + 1] : sys[2 * q + 1];
        split2_pair_f16(ok0 ? lr[k][2 * q][e] * s0 : 0.f, ok1 ? lr[k][2 * q + 1][e] * s1 : 0.f, a_, b_);
      }
      cv[k][0][e][q] = a_;
      cv[k][1][e][q] = b_;
      cv[k][2][e][q] = c_;
    };

    // ---- MFMA phase: row crow of the previous iteration's dY against ring rows crow+dy ----
    // Steps = (pixel pair, dx); the fragments of step s+1 are read (ping-pong registers) before the MFMAs of step s.
    // The fp32 -> 3 x bf16 conversion of the rows just requested runs in the shadow of the second half of the steps
    // (a few VALU instructions per MFMA group), so that after the barrier only the LDS writes remain.
    bool converted = false;
    if (crow >= 0) {
      const int yy = crow + dy;
      if (yy >= 0 && yy < H) {
        constexpr int JN = cdiv_c(NPAIR, KS);
        constexpr int NSTEP = JN * 3, S0 = NSTEP / 2, UPS = cdiv_c(NUNIT, NSTEP - S0);   // units per late step
        const int jb = ks * JN;
        const u32x4* xb = Xl + ((yy + 1) % 3) * XSLOT + l31 * XP + half + 1;
        const u32x4* db = Dl + l31 * DP + half;
        u32x4 af[2][MO][NP], bf[2][NP];
        auto load_a = [&](int buf, int jj) {
          const int j = min(jb + jj, NPAIR - 1);
#pragma unroll
          for (int m = 0; m < MO; ++m)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) af[buf][m][pc] = db[(pc * BCO + m * 32) * DP + 2 * j];
        };
        auto load_b = [&](int buf, int jj, int d) {
          const int j = min(jb + jj, NPAIR - 1);
#pragma unroll
          for (int pc = 0; pc < NP; ++pc) bf[buf][pc] = xb[pc * 32 * XP + 2 * j + d - 1];
        };
        auto mma = [&](const u32x4 (&A)[NP], const u32x4 (&B)[NP], f32x16 c) {
          if constexpr (NP == 3) {
            const bf16x8 a3[3] = {__builtin_bit_cast(bf16x8, A[0]), __builtin_bit_cast(bf16x8, A[1]),
                                  __builtin_bit_cast(bf16x8, A[2])};
            const bf16x8 b3[3] = {__builtin_bit_cast(bf16x8, B[0]), __builtin_bit_cast(bf16x8, B[1]),
                                  __builtin_bit_cast(bf16x8, B[2])};
            return mfma_bf16x6(a3, b3, c);
          } else {
            const f16x8 a0 = __builtin_bit_cast(f16x8, A[0]), a1 = __builtin_bit_cast(f16x8, A[1]);
            const f16x8 b0 = __builtin_bit_cast(f16x8, B[0]), b1 = __builtin_bit_cast(f16x8, B[1]);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c, 0, 0, 0);
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
          }
        };
        if (LEAN) {
#pragma unroll
          for (int s = 0; s < NSTEP; ++s) {
            const int jj = s / 3, d = s % 3;
            if (d == 0) load_a(0, jj);
            load_b(0, jj, d);
            if (NPAIR % KS == 0 || jb + jj < NPAIR) {
#pragma unroll
              for (int m = 0; m < MO; ++m) acc[m][d] = mma(af[0][m], bf[0], acc[m][d]);
            }
          }
        } else {
        load_a(0, 0);
          load_b(0, 0, 0);
  #pragma unroll
          for (int s = 0; s < NSTEP; ++s) {
            const int jj = s / 3, d = s % 3;
            if (s + 1 < NSTEP) {
              const int nj = (s + 1) / 3, nd = (s + 1) % 3;
              if (nd == 0) load_a(nj & 1, nj);
              load_b((s + 1) & 1, nj, nd);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the reads of step s+1 ahead of the MFMAs of step s
            if (KS == 1 || NPAIR % KS == 0 || jb + jj < NPAIR) {
  #pragma unroll
              for (int m = 0; m < MO; ++m) acc[m][d] = mma(af[jj & 1][m], bf[s & 1], acc[m][d]);
            }
            if (SHADOW && s >= S0) {
  #pragma unroll
              for (int c = (s - S0) * UPS; c < (s - S0 + 1) * UPS && c < NUNIT; ++c) convert_unit(c);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        converted = true;
      }
    }
    if (!have) break;
    if (SHADOW && !converted) {
#pragma unroll
      for (int c = 0; c < NUNIT; ++c) convert_unit(c);
    }

    // ---- store the converted rows ----
    __syncthreads();
    {
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        if (!SHADOW) {
#pragma unroll
          for (int c = 16 * k; c < 16 * k + 16; ++c) convert_unit(c);
        }
        const int row = it_x[k] ? xrow : drow;
        if (row < 0 || tid + k * THREADS >= NI) continue;
        u32x4* dst = it_x[k] ? Xl + ((row + 1) % 3) * XSLOT : Dl;
        const int pstride = it_x[k] ? 32 * XP : BCO * DP;
        const int lim = it_x[k] ? TW + 2 : TW;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (it_c0[k] + e < lim) {
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) dst[pc * pstride + it_rec[k] + e] = cv[k][pc][e];
          }
        }
      }
    }
    __syncthreads();
    crow = drow;
    // ---- advance (unit, t) ----
    ++t;
    if (t > y1 - xs) {     // iterations 0 .. y1-xs: the last one stages dY row y1-1 (and X row y1 if it exists)
      u += gridDim.x;
      have = u < a.nunits;
      t = 0;
      if (have) decode();
    }
  }

  if constexpr (NP == 2) {                 // undo P = 2^(280 - E) (two exact steps: the exponent may exceed fp32's range)
    const int e = emax - 280, e1 = e >> 1, e2 = e - e1;
    const float fx = emax > 0 ? __uint_as_float((unsigned)min(max(127 + e1, 1), 254) << 23) : 0.f;
    const float fy = emax > 0 ? __uint_as_float((unsigned)min(max(127 + e2, 1), 254) << 23) : 0.f;
#pragma unroll
    for (int m = 0; m < MO; ++m)
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][d][r] = (acc[m][d][r] * fx) * fy;
  }
  // ---- combine the KS shares of every kernel row through LDS, then one atomic per output element ----
  if (KS > 1) {
    float* red = reinterpret_cast<float*>(lds);
    const int slot = (dy + 1) * MO * 3 * 16 * 64;
    for (int k = 1; k < KS; ++k) {
      __syncthreads();
      if (ks == k) {
#pragma unroll
        for (int m = 0; m < MO; ++m)
#pragma unroll
          for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[slot + ((m * 3 + d) * 16 + r) * 64 + lane] = acc[m][d][r];
      }
      __syncthreads();
      if (ks == 0) {
#pragma unroll
        for (int m = 0; m < MO; ++m)
#pragma unroll
          for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][d][r] += red[slot + ((m * 3 + d) * 16 + r) * 64 + lane];
      }
    }
    if (ks != 0) return;
  }
  // D[i][j]: lane holds column j = l31 (input channel), rows (r&3) + 8*(r>>2) + 4*half (output channel)
  const int ci = ci0 + l31;
  if (ci >= a.C0 + a.C1) return;
#pragma unroll
  for (int m = 0; m < MO; ++m)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int tap = (dy + 1) * 3 + d;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < a.Cout) unsafeAtomicAdd(a.g + ((long long)co * 9 + tap) * a.Ctot + a.c_off + ci, acc[m][d][r]);
      }
    }
}

// (A producer / consumer form -- three MFMA-only kernel-row waves fed by two dedicated staging waves through a four-slot
// ring -- was built and measured in round 1: correct, but 15-20 % SLOWER (256->256 @6x9: 112 vs 93 us).  Its ablation
// showed why: the staging side alone takes 105 us on two waves; the fp32 -> 3 x bf16 conversion is VALU work of the
// same order as the MFMA time and needs all four SIMDs, so concentrating it on dedicated waves starves it.)

__global__ __launch_bounds__(256) void sample_exponents_kernel(const float* __restrict__ x, long long stride,
                                                                long long len, unsigned* __restrict__ be,
                                                                long long be_stride) {
  const float* xs = x + (long long)blockIdx.y * stride;
  float m = 0.f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < len; i += 256LL * gridDim.x) {
    const float v = fabsf(xs[i]);
    m = (v > m || v != v) ? v : m;          // (a NaN sticks)
  }
  unsigned bits = __float_as_uint(m) >> 23 & 0xffu;
  // wave reduction on the exponent (an integer in [0, 255], exact as a float)
  const unsigned wm = (unsigned)wave_max_nonneg((float)bits);
  if ((threadIdx.x & 63) == 0 && wm) atomicMax(be + (long long)blockIdx.y * be_stride, wm);
}

struct WsCfg {
  int tw, mo, ks;
};
constexpr WsCfg kWs[] = {
    {9, 1, 1},  {9, 2, 1},  {9, 1, 2},  {9, 2, 2},    // 0-3   W = 9, 18, 36, 72
    {8, 1, 1},  {8, 2, 1},  {8, 1, 2},  {8, 2, 2},    // 4-7   W = 8k
    {6, 1, 1},  {6, 2, 1},                            // 8-9   small tiles, 3 workgroups per CU
    {12, 1, 2}, {12, 2, 2},                           // 10-11
    {18, 1, 3}, {12, 1, 1},                           // 12-13
};
constexpr int kNumWs = sizeof(kWs) / sizeof(kWs[0]);

template <int I, bool DUAL, int NP>
int launch_ws(WsArgs a, int rounds4, hipStream_t st) {
  constexpr WsCfg c = kWs[I];
  using G = WsGeom<c.tw, c.mo, c.ks, NP>;
  constexpr size_t LDSB = G::LDS;
  constexpr int NTHR = 192 * c.ks;
  auto kern = wgrad3x3_split_kernel<c.tw, c.mo, c.ks, DUAL, NP>;
  static int occ = 0;
  if (occ == 0) {
    int nb = 0;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSB) != hipSuccess)
      return -22;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NTHR, LDSB) != hipSuccess || nb < 1) nb = 1;
    occ = nb;
  }
  const int tiles_ci = cdiv(a.C0 + a.C1, 32), tiles_co = cdiv(a.Cout, 32 * c.mo);
  a.ngroups = cdiv(a.N, 8);
  a.nsegs = cdiv(a.W, c.tw);
  // workgroups wanted per output tile, then row bands so that every workgroup gets >= 2 units where possible
  long long target = (long long)256 * occ * rounds4 / 4 / ((long long)tiles_ci * tiles_co);
  if (target < 1) target = 1;
  const int full = a.ngroups * a.nsegs;
  int nbands = cdiv(2 * target, full);
  const int maxb = a.H >= 8 ? a.H / 4 : 1;
  if (nbands > maxb) nbands = maxb;
  if (nbands < 1) nbands = 1;
  a.RB = cdiv(a.H, nbands);
  nbands = cdiv(a.H, a.RB);
  a.nunits = full * nbands;
  int z = (int)(target < a.nunits ? target : a.nunits);
  if (z >= 8) z = z / 8 * 8;                  // whole XCD rounds
  z = cdiv(a.nunits, cdiv(a.nunits, z));      // equal unit counts per workgroup (up to one)
  kern<<<dim3(z, tiles_ci, tiles_co), NTHR, LDSB, st>>>(a);
  CM_CHECK_LAUNCH();
  return 0;
}

template <bool DUAL, int NP, int I = 0>
int dispatch_ws(int cfg, const WsArgs& a, int rounds4, hipStream_t st) {
  if constexpr (I < kNumWs) {
    if (cfg == I) return launch_ws<I, DUAL, NP>(a, rounds4, st);
    return dispatch_ws<DUAL, NP, I + 1>(cfg, a, rounds4, st);
  } else {
    return -22;
  }
}

}  // namespace

extern "C" {

int cm_wgrad3x3_split_num_configs(void) { return kNumWs; }

int cm_wgrad3x3_split(const float* x0, long long sx0, int c0, const float* x1, long long sx1, int c1, const float* dy,
                      long long sdy, float* g, int ctot, int c_off, int n, int h, int w, int cout, int config,
                      cm_stream stream) {
  if (n <= 0 || h <= 0 || w <= 0 || cout <= 0 || c0 <= 0 || c1 < 0 || c_off < 0 || c_off + c0 + c1 > ctot || config < 0)
    return -22;
  if (c1 > 0 && (c0 % 32) != 0) return -22;   // a 32-channel tile must not straddle the two inputs
  WsArgs a;
  a.x0 = x0; a.x1 = x1; a.sx0 = sx0; a.sx1 = sx1; a.C0 = c0; a.C1 = c1;
  a.dy = dy; a.sdy = sdy; a.g = g; a.Ctot = ctot; a.c_off = c_off;
  a.N = n; a.H = h; a.W = w; a.Cout = cout;
  a.RB = a.ngroups = a.nsegs = a.nunits = 0;
  a.bex = a.bey = nullptr;
  const int rounds4 = (config >> 8) > 0 ? (config >> 8) : 4;   // bits 8.. = grid size in quarter rounds of resident slots
  config &= 0xff;
  return c1 > 0 ? dispatch_ws<true, 3>(config, a, rounds4, (hipStream_t)stream)
                : dispatch_ws<false, 3>(config, a, rounds4, (hipStream_t)stream);
}

/* fp16x3 form of cm_wgrad3x3_split (same configurations and staging format; csrc/split_f16.h): two fp16 pieces per
 * operand, three products.  be_x / be_y [n]: biased exponent (float bits >> 23) of max|x| / max|dy| of every sample, 0
 * for an all-zero sample -- as published by cm_conv3x3_h3 (sample_be) for the tensors it read, or by
 * cm_sample_exponents.  They need not be tight: any value >= the true exponent is safe (it only costs precision). */
int cm_wgrad3x3_h3(const float* x0, long long sx0, int c0, const float* x1, long long sx1, int c1, const float* dy,
                   long long sdy, const unsigned* be_x, const unsigned* be_y, float* g, int ctot, int c_off, int n,
                   int h, int w, int cout, int config, cm_stream stream) {
  if (n <= 0 || h <= 0 || w <= 0 || cout <= 0 || c0 <= 0 || c1 < 0 || c_off < 0 || c_off + c0 + c1 > ctot ||
      config < 0 || !be_x || !be_y)
    return -22;
  if (c1 > 0 && (c0 % 32) != 0) return -22;
  WsArgs a;
  a.x0 = x0; a.x1 = x1; a.sx0 = sx0; a.sx1 = sx1; a.C0 = c0; a.C1 = c1;
  a.dy = dy; a.sdy = sdy; a.g = g; a.Ctot = ctot; a.c_off = c_off;
  a.N = n; a.H = h; a.W = w; a.Cout = cout;
  a.RB = a.ngroups = a.nsegs = a.nunits = 0;
  a.bex = be_x; a.bey = be_y;
  const int rounds4 = (config >> 8) > 0 ? (config >> 8) : 4;
  config &= 0xff;
  return c1 > 0 ? dispatch_ws<true, 2>(config, a, rounds4, (hipStream_t)stream)
                : dispatch_ws<false, 2>(config, a, rounds4, (hipStream_t)stream);
}

/* be[i * be_stride] = max(be[...], biased exponent of max |x[i * stride + 0 .. len)|) for i < n (0 stays 0 for an
 * all-zero sample; NaN / inf give 255).  `be` must have been zeroed (or hold a previous partial result). */
int cm_sample_exponents(const float* x, long long stride, int n, long long len, unsigned* be, long long be_stride,
                        cm_stream stream) {
  if (n <= 0 || len <= 0 || !x || !be) return -22;
  const int per = (int)((len + 256LL * 16 - 1) / (256LL * 16));
  const int blocks = per < 64 ? (per < 1 ? 1 : per) : 64;
  sample_exponents_kernel<<<dim3(blocks, n), 256, 0, (hipStream_t)stream>>>(x, stride, len, be, be_stride);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
