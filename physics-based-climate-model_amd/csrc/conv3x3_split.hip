// 3x3 / pad 1 convolution on the bf16 matrix cores with fp32-equivalent accuracy ("bf16x6"):
// every fp32 operand is split into three bf16 pieces (hi, mid, lo; 24 mantissa bits in total) and the product is
// accumulated in fp32 from the six leading piece products
//     a*b ~= hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid           (dropped terms < 2^-24 |a*b|)
// with v_mfma_f32_32x32x16_bf16.  gfx950 runs bf16 MFMA at 16x the fp32-MFMA rate, so six bf16 MFMAs per 16-deep
// k-step cost 192 cycles where the fp32 kernel's eight 32x32x2 MFMAs cost 512.  tools/emulate_bf16_split.py shows on
// the CPU that this keeps every gradient of the model within the fp32 noise floor (~5e-6 rel), whereas the 3-term
// variant (bf16x3) sits at 9e-5 -- too close to the 1e-4 parity bound -- and is therefore not offered.
//
// Same role as conv3x3_mfma.hip (reference call sites src/unet.py:36,38, src/convlstm.py:9,13 and their data
// gradients), same GEMM orientation (rows = output channels, columns = pixels, 128-byte coalesced stores), but:
//   * K runs over 16 input channels of one tap per MFMA; a lane's fragment is 8 consecutive channels (16 bytes);
//   * the haloed input tile is converted on the way into LDS: a thread loads 8 channels of one pixel (8 coalesced
//     dword loads), splits them, and writes three 16-byte records Xl[piece][channel-octet][pixel];
//   * weight fragments are pre-split by cm_pack_conv3x3_split into [piece][16-channel step][tap][octet][cout][8];
//     the slab of one 16-channel step ([3][9][2][32*WM] 16-byte records) is staged in LDS next to the input tile and
//     shared by the workgroup's waves (reading the fragments straight from L2 in every wave cost ~60 % of a CU's L2
//     bandwidth and 24-48 VGPRs of ping-pong registers).
#include "common.h"
#include "split_bf16.h"
#include "split_f16.h"
#include "../../include/climate_hip.h"

namespace {

constexpr int SKC = 16;   // input channels per LDS stage = one 16-deep MFMA k-step per tap

struct SplitArgs {
  const float* in0;
  const float* in1;
  long long st0, st1;
  int C0, C1;
  const u32x4* wps;    // split weights: [3][nsteps][9][2][CoutP] records of 8 bf16
  const float* bias;
  const float* resid;
  float* out;
  long long sto;
  int N, H, W, Cout, CoutP, nsteps, tiles_x, tiles_y;   // nsteps = 16-channel k-steps (= LDS stages)
  int ksplit;  // > 1: blockIdx.z owns a share of the k-steps and accumulates into a pre-zeroed output with atomics
  int prezeroed;   // the caller already zeroed `out` (one fill for several launches): skip the internal zero launch
  const float* winv;      // fp16x3: device scalar, 1 / (power-of-two scale the packed weights carry)
  unsigned* be_out;       // fp16x3, optional: per-sample biased exponent of max|input| (atomic max), see cm_conv3x3_h3
  long long be_stride;
  long long zstride;      // > 0 ("partial slices"): reduction share blockIdx.z STORES its partial sums into slice z of a
                          // [ksplit][N] stack (out + z * zstride) -- no zero fill, no atomics; the consumer adds the slices
  float* gn_part;         // fp16x3, S == 1, no reduction split, optional: GroupNorm(8) partial statistics of the OUTPUT,
  int gn_slots, gn_cpg;   // [N][8][gn_slots][3] = {count, mean, sum of squares about that mean} per (tile, co sub-block)
};

// Workgroups with one 32x32 tile per wave sit 2 registers above the 3-waves-per-SIMD allocation (170 of 168): ask for
// three waves per SIMD (hip-clang's second __launch_bounds__ argument is MIN WAVES PER EU) so that the allocator trims
// them; LDS allows three 4-wave workgroups per CU (<= 53 KB each).
constexpr int split_min_waves(int waves, int npt, int wm) { return (npt * wm == 1 && waves == 4) ? 3 : 1; }

// NP = 3: bf16x6 (three bf16 pieces, six products); NP = 2: fp16x3 (two fp16 pieces, three products; split_f16.h)
// GN: also write GroupNorm partial statistics of the output tile (fp16x3, one sample per workgroup; a separate
// instantiation so that the plain kernels keep their register allocation)
template <int TH, int TW, int S, int WAVES, int NPT, int WM, bool DUAL, int NP, bool GN = false>
__global__ __launch_bounds__(WAVES * 64, split_min_waves(WAVES, NPT, WM)) void conv3x3_split_kernel(SplitArgs a) {
  static_assert(!GN || (NP == 2 && S == 1), "epilogue statistics: fp16x3, one sample per workgroup");
  constexpr int THREADS = WAVES * 64;
  constexpr int PITCH = TW + 2;
  constexpr int SS = (TH + 2) * PITCH;
  constexpr int PH = S * SS;                     // haloed pixels per stage
  constexpr int ITEMS = PH * 2;                  // (pixel, channel octet) records per 16-channel stage
  constexpr int NI = (ITEMS + THREADS - 1) / THREADS;
  constexpr int BCO = 32 * WM;
  constexpr int WREC = NP * 9 * 2 * BCO;         // weight records (16 B) per stage
  constexpr int NWR = (WREC + THREADS - 1) / THREADS;
  constexpr int PIX = S * TH * TW;
  static_assert(WAVES * NPT * 32 >= PIX, "block does not cover its pixel set");

  __shared__ u32x4 Xl[NP * 2 * PH];              // [piece][octet][pixel]
  __shared__ u32x4 Wl[WREC];                     // [piece][tap][octet][cout]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;

  const float* const a_in0 = a.in0;
  const float* const a_in1 = a.in1;
  const long long a_st0 = a.st0, a_st1 = a.st1;
  const int a_C0 = a.C0, a_C1 = a.C1, a_CoutP = a.CoutP, nsteps = a.nsteps;
  const u32x4* const a_wps = a.wps;

  int bx = blockIdx.x;
  const int tx = bx % a.tiles_x;
  bx /= a.tiles_x;
  const int ty = bx % a.tiles_y;
  const int g = bx / a.tiles_y;
  const int x0 = tx * TW, y0 = ty * TH, n0 = g * S;
  const int co0 = blockIdx.y * BCO;
  const int HW = a.H * a.W;

  // ---- stage-invariant staging offsets: item -> (octet, pixel) ----
  int goff0[NI];
  int goff1[DUAL ? NI : 1];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int e = tid + i * THREADS;
    const int pix = e % PH;
    const int s = pix / SS, r2 = pix % SS;
    const int row = r2 / PITCH, col = r2 % PITCH;
    const int gy = y0 - 1 + row, gx = x0 - 1 + col;
    const bool ok = (e < ITEMS) && (n0 + s < a.N) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    const int sp = gy * a.W + gx;
    goff0[i] = ok ? (int)(s * a.st0) + sp : -1;
    if (DUAL) goff1[i] = ok ? (int)(s * a.st1) + sp : -1;
  }
  // Global loads go through raw buffer instructions: a 32-bit per-lane BYTE offset that never changes over the K loop
  // (sample, pixel, channel octet) + a wave-uniform scalar offset per (stage, channel) -- no 64-bit address arithmetic
  // and no select per load; masked records (halo outside the image, samples past N) carry an offset beyond the
  // descriptor's range, for which the hardware returns 0 without touching memory.  (The compiler turned the previous
  // `src[ok ? off : 0]` form into ~4 64-bit VALU instructions and a branch per 4-byte load: 6 VALU per MFMA.)
  constexpr unsigned OOB = 0x80000000u;
  unsigned voff0[NI], voff1[DUAL ? NI : 1], wvoff[NWR];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int oct = (tid + i * THREADS) / PH;
    voff0[i] = goff0[i] >= 0 ? (unsigned)(goff0[i] + oct * 8 * HW) * 4u : OOB;
    if (DUAL) voff1[i] = goff1[i] >= 0 ? (unsigned)(goff1[i] + oct * 8 * HW) * 4u : OOB;
  }
  const __amdgpu_buffer_rsrc_t rs0 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_in0 + (long long)n0 * a_st0), 0, (int)OOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(DUAL ? a_in1 + (long long)n0 * a_st1 : a_in0), 0, (int)OOB, 0x00020000);

  float xr[NI][8];
  u32x4 wr[NWR];
  int cvalid_pending = 0;
  const long long piece_stride = (long long)nsteps * 9 * 2 * a_CoutP;   // records per piece in the packed weights

  // weight slab of one step: record r = ((piece*9 + tap)*2 + half)*BCO + col; per-lane part of its byte offset
#pragma unroll
  for (int i = 0; i < NWR; ++i) {
    const int r = min(tid + i * THREADS, WREC - 1);
    const int col = r % BCO, t2 = r / BCO;              // t2 = (piece*9 + tap)*2 + half
    const int pc = t2 / 18, th = t2 % 18;
    wvoff[i] = (unsigned)(pc * piece_stride + (long long)th * a_CoutP + col) * 16u;
  }
  const __amdgpu_buffer_rsrc_t rsw =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(a_wps + co0), 0, (int)OOB, 0x00020000);

  auto load_chunk = [&](int chunk) {
    const int ch0 = chunk * SKC;
    const bool second = DUAL && ch0 >= a_C0;
    const int cbase = second ? ch0 - a_C0 : ch0;                 // first channel of the stage inside its tensor
    const int cvalid = (second ? a_C0 + a_C1 : a_C0) - ch0;      // channels of this stage that exist (<= 16 matter)
    cvalid_pending = cvalid;
    const __amdgpu_buffer_rsrc_t rs = second ? rs1 : rs0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const unsigned vo = second ? voff1[DUAL ? i : 0] : voff0[i];
      if (cvalid >= SKC) {                                       // (wave uniform) every channel of the stage exists
#pragma unroll
        for (int j = 0; j < 8; ++j)
          xr[i][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo, (cbase + j) * HW * 4, 0));
      } else {                                                   // last stage of a channel count that is no multiple of 16
        const int oct = (tid + i * THREADS) / PH;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          xr[i][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, oct * 8 + j < cvalid ? vo : OOB,
                                                                           (cbase + j) * HW * 4, 0));
      }
    }
    const int wso = chunk * 9 * 2 * a_CoutP * 16;
#pragma unroll
    for (int i = 0; i < NWR; ++i) wr[i] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff[i], wso, 0);
  };
  auto store_chunk = [&](const float (&xsc_i)[NI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int e = tid + i * THREADS;
      const int oct = e / PH, pix = e % PH;
      u32x4 ph, pm, pl;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned a_, b_, c_ = 0;           // (masked records / channels were loaded as zeros: buffer range check)
        if constexpr (NP == 3)
          split3_pair(xr[i][2 * q], xr[i][2 * q + 1], a_, b_, c_);
        else
          split2_pair_f16(xr[i][2 * q] * xsc_i[i], xr[i][2 * q + 1] * xsc_i[i], a_, b_);
        ph[q] = a_; pm[q] = b_; pl[q] = c_;
      }
      if (e < ITEMS) {
        Xl[(0 * 2 + oct) * PH + pix] = ph;
        Xl[(1 * 2 + oct) * PH + pix] = pm;
        if constexpr (NP == 3) Xl[(2 * 2 + oct) * PH + pix] = pl;
      }
    }
#pragma unroll
    for (int i = 0; i < NWR; ++i) {
      const int r = tid + i * THREADS;
      if (r < WREC) Wl[r] = wr[i];
    }
  };

  // ---- per-lane pixel bookkeeping ----
  int xbase[NPT];
  long long obase[NPT];
  bool pvalid[NPT];
#pragma unroll
  for (int p = 0; p < NPT; ++p) {
    const int q = (wave * NPT + p) * 32 + l31;
    const bool inq = q < PIX;
    const int qq = inq ? q : 0;
    const int s = qq / (TH * TW), rem = qq % (TH * TW);
    const int py = rem / TW, px = rem % TW;
    xbase[p] = half * PH + s * SS + py * PITCH + px;      // octet = half -> +half*PH records
    const int n = n0 + s, gy = y0 + py, gx = x0 + px;
    pvalid[p] = inq && n < a.N && gy < a.H && gx < a.W;
    obase[p] = (long long)n * a.sto + (long long)gy * a.W + gx;
  }

  f32x16 acc[WM][NPT];
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int p = 0; p < NPT; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][p][r] = 0.f;

  const int wlane = half * BCO + l31;

  const int cps = (nsteps + a.ksplit - 1) / a.ksplit;
  const int c_begin = blockIdx.z * cps, c_end = min(nsteps, c_begin + cps);
  if (c_begin >= c_end) return;
  // fp16x3: every staged (sample, 16-channel stage) slab is scaled by a power of two chosen from the RUNNING maximum of
  // |x| over everything this workgroup has staged of THAT SAMPLE so far (exact: no history, no calibration, any input
  // range; samples of very different magnitude in one workgroup -- a left-padded frame next to a real one -- do not
  // share a scale).  A GEMM column is a pixel of one sample and K runs over that sample's channels and taps, so a
  // per-sample scale is constant along K; it lives per LANE on the accumulator side.  Each wave posts the per-sample
  // maxima of the registers it just loaded (LDS atomic max) before the barrier that ends a stage; after it every thread
  // combines the posts.  A scale can only shrink; when it does, the accumulators (which carry it) follow.
  __shared__ unsigned smax[2][S];
  unsigned be_run[S];                     // biased exponent of each sample's running maximum (0: only zeros so far)
  unsigned be_col[NPT];                   // ... of the sample this lane's GEMM column p belongs to
  int s_col[NPT], s_item[NI];
#pragma unroll
  for (int s_ = 0; s_ < S; ++s_) be_run[s_] = 0;
#pragma unroll
  for (int p = 0; p < NPT; ++p) {
    const int q = min((wave * NPT + p) * 32 + l31, PIX - 1);
    s_col[p] = q / (TH * TW);
    be_col[p] = 0;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) s_item[i] = ((tid + i * THREADS) % PH) / SS;
  if constexpr (NP == 2) {
    if (tid < 2 * S) smax[tid / S][tid % S] = 0u;
    __syncthreads();
  }
  auto post_max = [&](int buf) {
    float mi[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float m = 0.f;                         // (masked records / channels hold zeros)
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(xr[i][j]));
      mi[i] = m;
    }
    if constexpr (S == 1) {
      float m = mi[0];
#pragma unroll
      for (int i = 1; i < NI; ++i) m = fmaxf(m, mi[i]);
      m = wave_max_nonneg(m);
      if (lane == 0) atomicMax(&smax[buf][0], __float_as_uint(m));
    } else {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        // one item's 64 lanes cover 64 consecutive records: at most two samples (a sample has >= 88 haloed pixels).
        // The second one is NOT always first + 1: where the item straddles the boundary between the two channel octets
        // the pixel index wraps from the last sample back to sample 0 -- so its id is read from a lane that has it.
        const int first = __builtin_amdgcn_readfirstlane(s_item[i]);
        const unsigned long long othermask = __ballot(s_item[i] != first);
        const float ma = wave_max_nonneg(s_item[i] == first ? mi[i] : 0.f);
        const float mb = wave_max_nonneg(s_item[i] != first ? mi[i] : 0.f);
        const int other = othermask ? __builtin_amdgcn_readlane(s_item[i], (int)__builtin_ctzll(othermask)) : first;
        if (lane == 0) {
          atomicMax(&smax[buf][first], __float_as_uint(ma));
          if (othermask) atomicMax(&smax[buf][other], __float_as_uint(mb));
        }
      }
    }
  };
  // scale 2^(140 - be): the running maximum lands in [2^13, 2^14); its inverse 2^(be - 140)
  auto scale_of = [&](unsigned be) { return __uint_as_float((267u - max(be, 13u)) << 23); };
  static_assert(S == 1 || (TH + 2) * (TW + 2) >= 64, "an item must not span more than two samples");
  load_chunk(c_begin);
  if constexpr (NP == 2) {
    post_max(0);
    __syncthreads();
  }
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    float xsc_i[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) xsc_i[i] = 1.f;
    const int pbuf = (chunk - c_begin) & 1;
    if constexpr (NP == 2) {
      bool changed = false;
#pragma unroll
      for (int s_ = 0; s_ < S; ++s_) {
        const unsigned be = max(be_run[s_], (smax[pbuf][s_] >> 23) & 0xffu);  // (NaN / inf: be = 255, propagates)
        changed |= be != be_run[s_];
        be_run[s_] = be;
      }
      if (changed) {                      // (workgroup uniform) accumulators hold sums at the old scales: rescale them
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
          unsigned be = be_run[0];
#pragma unroll
          for (int s_ = 1; s_ < S; ++s_) be = s_col[p] == s_ ? be_run[s_] : be;
          const int d = (int)be - (int)be_col[p];
          const float f = (be_col[p] == 0 || d > 126) ? 0.f : __uint_as_float((unsigned)(127 - d) << 23);
          be_col[p] = be;
#pragma unroll
          for (int m_ = 0; m_ < WM; ++m_)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m_][p][r] *= f;       // (be_col was 0: nothing but zeros accumulated)
        }
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        unsigned be = be_run[0];
#pragma unroll
        for (int s_ = 1; s_ < S; ++s_) be = s_item[i] == s_ ? be_run[s_] : be;
        xsc_i[i] = scale_of(be);
      }
    }
    store_chunk(xsc_i);
    __syncthreads();
    if constexpr (NP == 2) {
      if (tid < S) smax[pbuf][tid] = 0u;          // read by everyone before this barrier; re-posted two barriers on
    }
    if (chunk + 1 < c_end) load_chunk(chunk + 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if constexpr (NP == 3) {
        bf16x8 bf[3][NPT];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
          for (int p = 0; p < NPT; ++p)
            bf[pc][p] = __builtin_bit_cast(bf16x8, Xl[(pc * 2) * PH + xbase[p] + (tap / 3) * PITCH + (tap % 3)]);
#pragma unroll
        for (int m = 0; m < WM; ++m) {
          bf16x8 af[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc)
            af[pc] = __builtin_bit_cast(bf16x8, Wl[(pc * 9 + tap) * 2 * BCO + wlane + m * 32]);
#pragma unroll
          for (int p = 0; p < NPT; ++p) {
            // smallest terms first
            acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1][p], acc[m][p], 0, 0, 0);
            acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0][p], acc[m][p], 0, 0, 0);
            acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2][p], acc[m][p], 0, 0, 0);
            acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0][p], acc[m][p], 0, 0, 0);
            acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1][p], acc[m][p], 0, 0, 0);
            acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0][p], acc[m][p], 0, 0, 0);
          }
        }
      } else {
        f16x8 bf[2][NPT];
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
#pragma unroll
          for (int p = 0; p < NPT; ++p)
            bf[pc][p] = __builtin_bit_cast(f16x8, Xl[(pc * 2) * PH + xbase[p] + (tap / 3) * PITCH + (tap % 3)]);
#pragma unroll
        for (int m = 0; m < WM; ++m) {
          f16x8 af[2];
#pragma unroll
          for (int pc = 0; pc < 2; ++pc)
            af[pc] = __builtin_bit_cast(f16x8, Wl[(pc * 9 + tap) * 2 * BCO + wlane + m * 32]);
#pragma unroll
          for (int p = 0; p < NPT; ++p) {
            acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], bf[0][p], acc[m][p], 0, 0, 0);
            acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[1][p], acc[m][p], 0, 0, 0);
            acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[0][p], acc[m][p], 0, 0, 0);
          }
        }
      }
    }
    if constexpr (NP == 2) {
      if (chunk + 1 < c_end) post_max(((chunk - c_begin) & 1) ^ 1);
    }
    __syncthreads();
  }

  // ---- epilogue (same accumulator map as the fp32 kernel) ----
  if constexpr (NP == 2) {
    // publish what this workgroup knows about its samples' magnitudes (the fp16x3 weight gradient scales per sample)
    if (a.be_out != nullptr && tid < S && n0 + tid < a.N) {
      unsigned be = be_run[0];
#pragma unroll
      for (int s_ = 1; s_ < S; ++s_) be = tid == s_ ? be_run[s_] : be;
      if (be) atomicMax(a.be_out + (long long)(n0 + tid) * a.be_stride, be);
    }
  }
  if constexpr (NP == 2) {
    // undo the operand scales: 2^(be - 140) for the input (0 if nothing but zeros was staged), *winv for the weights
    const float winv = a.winv[0];
#pragma unroll
    for (int p = 0; p < NPT; ++p) {
      const float xinv = be_col[p] <= 13u ? 0.f : __uint_as_float((be_col[p] - 13u) << 23);
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][p][r] = (acc[m][p][r] * xinv) * winv;
    }
  }
  if (a.bias && blockIdx.z == 0) {
#pragma unroll
    for (int m = 0; m < WM; ++m) {
      float bv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        bv[r] = a.bias[co < a.Cout ? co : 0];
      }
#pragma unroll
      for (int p = 0; p < NPT; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][p][r] += bv[r];
    }
  }
  if (a.resid && blockIdx.z == 0) {
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const bool ok = pvalid[p] && co < a.Cout;
          rv[r] = a.resid[ok ? obase[p] + (long long)co * HW : 0];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][p][r] += rv[r];
      }
  }
  if constexpr (GN) {
    // GroupNorm statistics of this workgroup's output tile, from the accumulators (reference: nn.GroupNorm(8, c) right
    // after the conv, src/unet.py:36-39): per group a LOCAL two-pass -- mean of the tile's part of the group, then the sum
    // of squares about that mean -- stored as {count, mean, M2}; the consumer (cm_gn_silu_fwd_stats) merges the parts of a
    // (sample, group) with the parallel-variance formula, so no pivot has to be agreed on between workgroups.
    // A lane's registers r = 4k + j (j = 0..3) are 4 consecutive channels co0 + 32 m + 8 k + 4 half + j: for groups of
    // >= 4 channels they belong to ONE group, so a lane keeps one partial per (m, k) and the 32 lanes of a half are added
    // by DPP; the few per-wave results go through LDS atomics.
    {
      // per-wave slots, written by lane 0 only and added in wave order: deterministic (the amax / MaxPool decisions
      // downstream depend on the statistics bit for bit, and two runs must take the same ones)
      __shared__ float gw[2][WAVES][16], gcw[WAVES][16], gsum[2][16], gcnt[16];
      const int cpg = a.gn_cpg;
      const int ngl = max(1, BCO / cpg);               // groups (or one part of a group) in this co block
      if (lane < 16) { gw[0][wave][lane] = 0.f; gw[1][wave][lane] = 0.f; gcw[wave][lane] = 0.f; }
      auto group_of = [&](int m, int k, int hf) { return min((m * 32 + 8 * k + 4 * hf) / cpg, ngl - 1); };
      auto lane_reduce = [&](float v, float& lo, float& hi) {
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, true));
        lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
        hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
      };
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float mean = pass == 0 ? 0.f : gsum[0][group_of(m, k, half)];
            float v = 0.f, c = 0.f;
#pragma unroll
            for (int p = 0; p < NPT; ++p)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const bool ok = pvalid[p] && (co0 + m * 32 + 8 * k + 4 * half + j) < a.Cout;
                const float d = acc[m][p][4 * k + j] - mean;
                v += ok ? (pass == 0 ? d : d * d) : 0.f;
                c += ok ? 1.f : 0.f;
              }
            float lo, hi, clo = 0.f, chi = 0.f;
            lane_reduce(v, lo, hi);
            if (pass == 0) lane_reduce(c, clo, chi);
            if (lane == 0) {                           // (one lane, program order: no atomics needed)
              gw[pass][wave][group_of(m, k, 0)] += lo;
              gw[pass][wave][group_of(m, k, 1)] += hi;
              if (pass == 0) {
                gcw[wave][group_of(m, k, 0)] += clo;
                gcw[wave][group_of(m, k, 1)] += chi;
              }
            }
          }
        __syncthreads();
        if (tid < ngl) {
          float t = 0.f, cn = 0.f;
#pragma unroll
          for (int w2 = 0; w2 < WAVES; ++w2) { t += gw[pass][w2][tid]; cn += gcw[w2][tid]; }
          if (pass == 0) {
            gcnt[tid] = cn;
            gsum[0][tid] = cn > 0.f ? t / cn : 0.f;    // local mean
          } else {
            gsum[1][tid] = t;
          }
        }
        __syncthreads();
      }
      if (tid < ngl && n0 < a.N) {
        const int sb = max(1, cpg / BCO);                           // co blocks per group (groups wider than the block)
        const int g = (co0 + tid * cpg) / cpg;
        const int slot = (ty * a.tiles_x + tx) * sb + (sb > 1 ? (int)(blockIdx.y % sb) : 0);
        if (g < 8) {
          float* dst = a.gn_part + (((long long)n0 * 8 + g) * a.gn_slots + slot) * 3;
          dst[0] = gcnt[tid];
          dst[1] = gsum[0][tid];
          dst[2] = gsum[1][tid];
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < WM; ++m)
#pragma unroll
    for (int p = 0; p < NPT; ++p) {
      if (pvalid[p]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (co < a.Cout) {
            if (a.ksplit > 1 && a.zstride == 0)
              unsafeAtomicAdd(a.out + obase[p] + (long long)co * HW, acc[m][p][r]);
            else
              a.out[obase[p] + (long long)co * HW + (long long)blockIdx.z * a.zstride] = acc[m][p][r];
          }
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------ packer
// descs as cm_pack_conv3x3_batch: {w ptr, wps ptr, cout, cin_total, c_off, cin, dgrad, first block}
// output record (piece, T = step*9 + tap, octet-half h, col) = 8 bf16 pieces of k-channels step*16 + h*8 + j
// Per-block maximum of |w| over the WHOLE weight tensor of the block's job (a superset of the slice the job packs: the
// scale only has to be of the right order, and whole tensors stream as contiguous 16-byte loads): partial[blockIdx.x].
__global__ void weight_amax_batch_kernel(const long long* __restrict__ descs, int ndesc, float* __restrict__ partial) {
  __shared__ float red[4];
  const int d = cm_job_of_block(descs, ndesc);
  const long long* r = descs + d * 8;
  const float* w = reinterpret_cast<const float*>(r[0]);
  const long long total = r[2] * r[3] * 9;                 // cout * cin_total * 9 floats
  const int b0 = (int)r[7], nb = (int)descs[(d + 1) * 8 + 7] - b0;
  const long long t0 = (long long)(blockIdx.x - b0) * blockDim.x + threadIdx.x, stride = (long long)nb * blockDim.x;
  float m = 0.f;
  if ((r[0] & 15) == 0) {
    const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
    for (long long i = t0; i < total / 4; i += stride) {
      const f32x4 v = w4[i];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (long long i = (total / 4) * 4 + t0; i < total; i += stride) m = fmaxf(m, fabsf(w[i]));
  } else {
    for (long long i = t0; i < total; i += stride) m = fmaxf(m, fabsf(w[i]));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

template <int NP>
__global__ void pack_split_batch_kernel(const long long* __restrict__ descs, int ndesc, const float* __restrict__ partial,
                                        float* __restrict__ winv_out) {
  const int d = cm_job_of_block(descs, ndesc);
  const long long* r = descs + d * 8;
  const float* w = reinterpret_cast<const float*>(r[0]);
  u32x4* wps = reinterpret_cast<u32x4*>(r[1]);
  const int cout = (int)r[2], cin_total = (int)r[3], c_off = (int)r[4], cin = (int)r[5], dgrad = (int)r[6];
  const int b0 = (int)r[7], nb = (int)descs[(d + 1) * 8 + 7] - b0;
  const int kch = dgrad ? cout : cin, ocs = dgrad ? cin : cout;
  const int nsteps = (kch + SKC - 1) / SKC, colsP = ((ocs + 31) / 32) * 32;
  const long long recs = (long long)nsteps * 9 * 2 * colsP;      // records per piece
  float wscale = 1.f;
  if constexpr (NP == 2) {              // largest weight -> [2^13, 2^14); the conv's epilogue multiplies by winv
    __shared__ float red[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) m = fmaxf(m, partial[b0 + i]);   // this job's block maxima
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const unsigned be = max((__float_as_uint(m) >> 23) & 0xffu, 13u);
    wscale = __uint_as_float((267u - be) << 23);
    if (blockIdx.x == b0 && threadIdx.x == 0) winv_out[d] = be <= 13u ? 0.f : __uint_as_float((be - 13u) << 23);
  }
  for (long long i = (long long)(blockIdx.x - b0) * blockDim.x + threadIdx.x; i < recs; i += (long long)nb * blockDim.x) {
    const int col = (int)(i % colsP);
    long long t = i / colsP;
    const int h = (int)(t % 2);
    t /= 2;
    const int tap = (int)(t % 9), step = (int)(t / 9);
    float vv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kc = step * 16 + h * 8 + j;
      float v = 0.f;
      if (!dgrad) {
        if (kc < cin && col < cout) v = w[((long long)col * cin_total + c_off + kc) * 9 + tap];
      } else {
        if (kc < cout && col < cin) v = w[((long long)kc * cin_total + c_off + col) * 9 + (8 - tap)];
      }
      vv[j] = v;
    }
    u32x4 ph, pm, pl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned a_, b_, c_ = 0;
      if constexpr (NP == 3) split3_pair(vv[2 * q], vv[2 * q + 1], a_, b_, c_);
      else split2_pair_f16(vv[2 * q] * wscale, vv[2 * q + 1] * wscale, a_, b_);
      ph[q] = a_; pm[q] = b_; pl[q] = c_;
    }
    wps[i] = ph;
    wps[recs + i] = pm;
    if constexpr (NP == 3) wps[2 * recs + i] = pl;
  }
}

struct SCfg {
  int th, tw, s, waves, npt, wm;
};
constexpr SCfg kS[] = {
    {8, 16, 1, 4, 1, 1},   // 0: 128 px x 32 co
    {8, 16, 1, 2, 2, 1},   // 1: 128 px x 32 co, 2 waves x 2 tiles
    {8, 16, 1, 2, 2, 2},   // 2: 128 px x 64 co
    {12, 18, 1, 4, 2, 1},  // 3: 216 px x 32 co
    {12, 18, 1, 4, 2, 2},  // 4: 216 px x 64 co
    {6, 9, 2, 4, 1, 1},    // 5: 2 x 54 px x 32 co
    {6, 9, 2, 2, 2, 2},    // 6: 2 x 54 px x 64 co
    {6, 9, 4, 4, 2, 1},    // 7: 4 x 54 px x 32 co
    {6, 9, 4, 4, 2, 2},    // 8: 4 x 54 px x 64 co
    {8, 24, 1, 3, 2, 1},   // 9: 192 px x 32 co
    {8, 24, 1, 3, 2, 2},   // 10: 192 px x 64 co
    {16, 24, 1, 4, 3, 1},  // 11: 384 px x 32 co
    {8, 16, 1, 4, 1, 2},   // 12: 128 px x 64 co
    {16, 16, 1, 8, 1, 1},  // 13: 256 px x 32 co, 8 waves
    {16, 16, 1, 8, 1, 2},  // 14: 256 px x 64 co, 8 waves
    {12, 18, 1, 8, 1, 2},  // 15: 216 px x 64 co, 8 waves
    {6, 9, 4, 8, 1, 2},    // 16: 4 x 54 px x 64 co, 8 waves
    {6, 9, 4, 8, 1, 1},    // 17: 4 x 54 px x 32 co, 8 waves
    {12, 18, 1, 8, 1, 1},  // 18: 216 px x 32 co, 8 waves
    {16, 24, 1, 6, 2, 2},  // 19: 384 px x 64 co, 6 waves
    // Sample groups of 3 and 6: with N = B*T = 192 = 3*64 frames the workgroup count becomes a multiple of the 256 CUs
    // (S = 4 gives 48 groups -> 0.75 or 1.5 rounds of resident slots: a quarter of the chip idles in the last round).
    {6, 9, 3, 3, 2, 1},    // 20: 3 x 54 px x 32 co, 3 waves
    {6, 9, 3, 3, 2, 2},    // 21: 3 x 54 px x 64 co, 3 waves
    {6, 9, 6, 6, 2, 1},    // 22: 6 x 54 px x 32 co, 6 waves
    {6, 9, 6, 6, 2, 2},    // 23: 6 x 54 px x 64 co, 6 waves
    {6, 18, 3, 6, 2, 1},   // 24: 3 x 108 px x 32 co (half images of the 12x18 level), 6 waves
    {6, 18, 3, 6, 2, 2},   // 25: 3 x 108 px x 64 co
    {12, 18, 3, 7, 3, 1},  // 26: 3 x 216 px x 32 co, 7 waves
    {6, 18, 1, 4, 1, 1},   // 27: 108 px x 32 co (half images of the 12x18 level), three workgroups per CU
};
constexpr int kNumS = sizeof(kS) / sizeof(kS[0]);

__global__ void zero_out_split_kernel(float* __restrict__ out, long long sto, int n, long long per) {
  const long long total = (long long)n * per;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    out[(i / per) * sto + i % per] = 0.f;
}

template <int I, bool DUAL, int NP>
int launch_s(const SplitArgs& a0, hipStream_t st) {
  constexpr SCfg c = kS[I];
  SplitArgs a = a0;
  a.tiles_x = cdiv(a.W, c.tw);
  a.tiles_y = cdiv(a.H, c.th);
  // partial slices: the caller reads exactly ksplit slices, each of which must own >= 1 k-step
  if (a.zstride > 0 && (a.ksplit < 2 || a.ksplit > a.nsteps || a.nsteps % a.ksplit != 0)) return -22;
  if (a.ksplit > a.nsteps) a.ksplit = a.nsteps;
  if (a.ksplit < 1) a.ksplit = 1;
  if (a.ksplit > 1 && !a.prezeroed && a.zstride == 0) {
    const long long per = (long long)a.Cout * a.H * a.W;
    const long long zb = ((long long)a.N * per + 255) / 256;
    zero_out_split_kernel<<<(int)(zb > 2048 ? 2048 : zb), 256, 0, st>>>(a.out, a.sto, a.N, per);
  }
  if (a.gn_part && (c.s != 1 || a.ksplit > 1 || a.zstride > 0 ||
                    a.gn_slots != a.tiles_x * a.tiles_y * (a.gn_cpg > 32 * c.wm ? a.gn_cpg / (32 * c.wm) : 1)))
    return -22;          // (the caller sized the partial-statistics buffer with cm_conv3x3_h3_gn_slots for this config)
  dim3 grid(a.tiles_x * a.tiles_y * cdiv(a.N, c.s), cdiv(a.Cout, 32 * c.wm), a.ksplit);
  if constexpr (NP == 2 && c.s == 1) {
    if (a.gn_part) {
      conv3x3_split_kernel<c.th, c.tw, c.s, c.waves, c.npt, c.wm, DUAL, NP, true><<<grid, c.waves * 64, 0, st>>>(a);
      CM_CHECK_LAUNCH();
      return 0;
    }
  }
  conv3x3_split_kernel<c.th, c.tw, c.s, c.waves, c.npt, c.wm, DUAL, NP><<<grid, c.waves * 64, 0, st>>>(a);
  CM_CHECK_LAUNCH();
  return 0;
}

template <bool DUAL, int NP>
int dispatch_s(int cfg, const SplitArgs& a, hipStream_t st) {
  switch (cfg) {
    case 0: return launch_s<0, DUAL, NP>(a, st);
    case 1: return launch_s<1, DUAL, NP>(a, st);
    case 2: return launch_s<2, DUAL, NP>(a, st);
    case 3: return launch_s<3, DUAL, NP>(a, st);
    case 4: return launch_s<4, DUAL, NP>(a, st);
    case 5: return launch_s<5, DUAL, NP>(a, st);
    case 6: return launch_s<6, DUAL, NP>(a, st);
    case 7: return launch_s<7, DUAL, NP>(a, st);
    case 8: return launch_s<8, DUAL, NP>(a, st);
    case 9: return launch_s<9, DUAL, NP>(a, st);
    case 10: return launch_s<10, DUAL, NP>(a, st);
    case 11: return launch_s<11, DUAL, NP>(a, st);
    case 12: return launch_s<12, DUAL, NP>(a, st);
    case 13: return launch_s<13, DUAL, NP>(a, st);
    case 14: return launch_s<14, DUAL, NP>(a, st);
    case 15: return launch_s<15, DUAL, NP>(a, st);
    case 16: return launch_s<16, DUAL, NP>(a, st);
    case 17: return launch_s<17, DUAL, NP>(a, st);
    case 18: return launch_s<18, DUAL, NP>(a, st);
    case 19: return launch_s<19, DUAL, NP>(a, st);
    case 20: return launch_s<20, DUAL, NP>(a, st);
    case 21: return launch_s<21, DUAL, NP>(a, st);
    case 22: return launch_s<22, DUAL, NP>(a, st);
    case 23: return launch_s<23, DUAL, NP>(a, st);
    case 24: return launch_s<24, DUAL, NP>(a, st);
    case 25: return launch_s<25, DUAL, NP>(a, st);
    case 26: return launch_s<26, DUAL, NP>(a, st);
    case 27: return launch_s<27, DUAL, NP>(a, st);
    default: return -22;
  }
}

// the kernels address a workgroup's samples (at most 7) and the packed weights with 32-bit byte offsets
static bool offsets_fit_32bit(const SplitArgs& a) {
  const long long hw4 = (long long)a.H * a.W * 4;
  const long long lim = 0x7fffffffLL;
  if (7 * a.st0 * 4 + (a.C0 + 16) * hw4 >= lim) return false;
  if (a.C1 > 0 && 7 * a.st1 * 4 + (a.C1 + 16) * hw4 >= lim) return false;
  return 3LL * a.nsteps * 9 * 2 * a.CoutP * 16 < lim;
}

}  // namespace

extern "C" {

int cm_conv3x3_split_num_configs(void) { return kNumS; }

long long cm_conv3x3_split_packed_bytes(int k_channels, int out_channels) {
  const long long nsteps = (long long)((k_channels + SKC - 1) / SKC);
  const long long colsP = (long long)((out_channels + 31) / 32) * 32;
  return 3 * nsteps * 9 * 2 * colsP * 16;
}

int cm_pack_conv3x3_split_batch(const void* descs_dev, int ndesc, int total_blocks, cm_stream stream) {
  if (ndesc <= 0 || total_blocks <= 0) return -22;
  pack_split_batch_kernel<3><<<total_blocks, 256, 0, (hipStream_t)stream>>>((const long long*)descs_dev, ndesc, nullptr,
                                                                            nullptr);
  CM_CHECK_LAUNCH();
  return 0;
}

/* fp16x3 operand form: same descriptor records as cm_pack_conv3x3_split_batch; every job's weight slice is scaled by
 * a power of two (largest |w| of its tensor -> [2^13, 2^14)) and split into two fp16 pieces.  scratch: ndesc +
 * total_blocks floats: [0, ndesc) = per-job inverse scales (pass &scratch[job] to cm_conv3x3_h3), the rest workspace. */
int cm_pack_conv3x3_h3_batch(const void* descs_dev, int ndesc, int total_blocks, float* scratch, cm_stream stream) {
  if (ndesc <= 0 || total_blocks <= 0 || !scratch) return -22;
  hipStream_t st = (hipStream_t)stream;
  weight_amax_batch_kernel<<<total_blocks, 256, 0, st>>>((const long long*)descs_dev, ndesc, scratch + ndesc);
  pack_split_batch_kernel<2><<<total_blocks, 256, 0, st>>>((const long long*)descs_dev, ndesc, scratch + ndesc, scratch);
  CM_CHECK_LAUNCH();
  return 0;
}

long long cm_conv3x3_h3_packed_bytes(int k_channels, int out_channels) {
  return cm_conv3x3_split_packed_bytes(k_channels, out_channels) / 3 * 2;
}

int cm_conv3x3_split(const float* in0, long long st0, int c0, const float* in1, long long st1, int c1,
                     const void* wps, const float* bias, const float* resid, long long st_resid, float* out,
                     long long st_out, int n, int h, int w, int cout, int config, cm_stream stream) {
  if (n <= 0 || h <= 0 || w <= 0 || cout <= 0 || c0 <= 0 || c1 < 0 || config < 0) return -22;
  if (c1 > 0 && (c0 % SKC) != 0) return -22;   // a 16-channel stage must not straddle the two inputs
  if (resid && st_resid != st_out) return -22;
  SplitArgs a;
  a.in0 = in0; a.in1 = in1; a.st0 = st0; a.st1 = st1; a.C0 = c0; a.C1 = c1;
  a.wps = (const u32x4*)wps; a.bias = bias; a.resid = resid; a.out = out; a.sto = st_out;
  a.N = n; a.H = h; a.W = w; a.Cout = cout;
  a.CoutP = ((cout + 31) / 32) * 32;
  a.nsteps = (c0 + c1 + SKC - 1) / SKC;
  a.prezeroed = (config >> 30) & 1;          // bit 30: `out` is already zero (caller batches the fills)
  config &= ~(1 << 30);
  a.ksplit = config >> 8;                    // bits 8.. = K split over blockIdx.z (0/1 = none)
  config &= 0xff;
  if (a.ksplit > 1 && resid == out) return -22;   // the in-place residual would be zeroed
  a.tiles_x = a.tiles_y = 0;
  a.winv = nullptr;
  a.be_out = nullptr; a.be_stride = 0; a.zstride = 0;
  a.gn_part = nullptr; a.gn_slots = 0; a.gn_cpg = 0;
  if (!offsets_fit_32bit(a)) return -22;
  return c1 > 0 ? dispatch_s<true, 3>(config, a, (hipStream_t)stream) : dispatch_s<false, 3>(config, a, (hipStream_t)stream);
}

/* cm_conv3x3_split on two fp16 pieces per operand and three products ("fp16x3", csrc/split_f16.h): same contract and
 * config encoding; wps / wscale_inv from cm_pack_conv3x3_h3_batch.  Input scaling is internal (running maximum).
 * sample_be (optional, [n] entries be_stride apart, zeroed by the caller): the kernel raises entry i to the biased exponent
 * of max |input of sample i| over both input tensors (atomic max over workgroups) -- the per-sample magnitudes the
 * fp16x3 weight gradient (cm_wgrad3x3_h3) needs, obtained for free from the launch that reads the same tensor.
 * config bit 29 ("partial slices", needs a reduction split k >= 2 that divides the 16-channel k-steps, no resid): `out` is
 * a [k][n] stack of slices st_out apart; reduction share z STORES its partial sums into slice z (bias in slice 0) -- no
 * zero fill, no atomics, a fixed summation order; the consumer adds the slices (cm_lstm_gates_fwd_parts / _bwd_parts). */
int cm_conv3x3_h3(const float* in0, long long st0, int c0, const float* in1, long long st1, int c1, const void* wps,
                  const float* wscale_inv, const float* bias, const float* resid, long long st_resid, float* out,
                  long long st_out, unsigned* sample_be, long long be_stride, int n, int h, int w, int cout, int config,
                  cm_stream stream) {
  if (n <= 0 || h <= 0 || w <= 0 || cout <= 0 || c0 <= 0 || c1 < 0 || config < 0 || !wscale_inv) return -22;
  if (c1 > 0 && (c0 % SKC) != 0) return -22;
  if (resid && st_resid != st_out) return -22;
  SplitArgs a;
  a.in0 = in0; a.in1 = in1; a.st0 = st0; a.st1 = st1; a.C0 = c0; a.C1 = c1;
  a.wps = (const u32x4*)wps; a.bias = bias; a.resid = resid; a.out = out; a.sto = st_out;
  a.N = n; a.H = h; a.W = w; a.Cout = cout;
  a.CoutP = ((cout + 31) / 32) * 32;
  a.nsteps = (c0 + c1 + SKC - 1) / SKC;
  a.prezeroed = (config >> 30) & 1;
  config &= ~(1 << 30);
  const bool slices = (config >> 29) & 1;    // bit 29: partial slices, `out` = [ksplit][n] stack, st_out apart
  config &= ~(1 << 29);
  a.ksplit = config >> 8;
  config &= 0xff;
  if (a.ksplit > 1 && resid == out) return -22;
  if (slices && resid) return -22;
  a.zstride = slices ? (long long)n * st_out : 0;
  a.tiles_x = a.tiles_y = 0;
  a.winv = wscale_inv;
  a.be_out = sample_be; a.be_stride = be_stride;
  a.gn_part = nullptr; a.gn_slots = 0; a.gn_cpg = 0;
  if (!offsets_fit_32bit(a)) return -22;
  return c1 > 0 ? dispatch_s<true, 2>(config, a, (hipStream_t)stream) : dispatch_s<false, 2>(config, a, (hipStream_t)stream);
}

/* GroupNorm(8) partial statistics from the conv epilogue (src/unet.py:36-39: every ConvBlock conv is followed by
 * nn.GroupNorm(8, cout)): number of {count, mean, M2} records per (sample, group) that cm_conv3x3_h3_gn writes with this
 * configuration, or 0 when the configuration cannot produce them (sample groups, a reduction split, partial slices, cout
 * not 8 groups of a power-of-two >= 4 channels). */
int cm_conv3x3_h3_gn_slots(int config, int h, int w, int cout) {
  if (config < 0 || h <= 0 || w <= 0 || cout <= 0 || cout % 8) return 0;
  if ((config >> 29) & 1) return 0;
  config &= ~(1 << 30);
  if ((config >> 8) > 1) return 0;
  const int ci = config & 0xff;
  if (ci >= kNumS) return 0;
  const SCfg c = kS[ci];
  const int cpg = cout / 8;
  if (c.s != 1 || cpg < 4 || (cpg & (cpg - 1))) return 0;
  const int sb = cpg > 32 * c.wm ? cpg / (32 * c.wm) : 1;
  return cdiv(w, c.tw) * cdiv(h, c.th) * sb;
}

/* cm_conv3x3_h3 that also writes those partial statistics of its output: gn_part [n][8][gn_slots][3] floats,
 * gn_slots = cm_conv3x3_h3_gn_slots(config, h, w, cout) > 0.  Consumer: cm_gn_silu_fwd_stats. */
int cm_conv3x3_h3_gn(const float* in0, long long st0, int c0, const float* in1, long long st1, int c1, const void* wps,
                     const float* wscale_inv, const float* bias, const float* resid, long long st_resid, float* out,
                     long long st_out, unsigned* sample_be, long long be_stride, float* gn_part, int gn_slots, int n, int h,
                     int w, int cout, int config, cm_stream stream) {
  if (n <= 0 || h <= 0 || w <= 0 || cout <= 0 || c0 <= 0 || c1 < 0 || config < 0 || !wscale_inv) return -22;
  if (c1 > 0 && (c0 % SKC) != 0) return -22;
  if (resid && st_resid != st_out) return -22;
  if (!gn_part || gn_slots <= 0 || gn_slots != cm_conv3x3_h3_gn_slots(config, h, w, cout)) return -22;
  SplitArgs a;
  a.in0 = in0; a.in1 = in1; a.st0 = st0; a.st1 = st1; a.C0 = c0; a.C1 = c1;
  a.wps = (const u32x4*)wps; a.bias = bias; a.resid = resid; a.out = out; a.sto = st_out;
  a.N = n; a.H = h; a.W = w; a.Cout = cout;
  a.CoutP = ((cout + 31) / 32) * 32;
  a.nsteps = (c0 + c1 + SKC - 1) / SKC;
  a.prezeroed = 0;
  a.ksplit = 1;
  config &= 0xff;
  a.zstride = 0;
  a.tiles_x = a.tiles_y = 0;
  a.winv = wscale_inv;
  a.be_out = sample_be; a.be_stride = be_stride;
  a.gn_part = gn_part; a.gn_slots = gn_slots; a.gn_cpg = cout / 8;
  if (!offsets_fit_32bit(a)) return -22;
  return c1 > 0 ? dispatch_s<true, 2>(config, a, (hipStream_t)stream) : dispatch_s<false, 2>(config, a, (hipStream_t)stream);
}

}  // extern "C"
