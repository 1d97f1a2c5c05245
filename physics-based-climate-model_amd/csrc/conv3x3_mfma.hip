// 3x3 / pad 1 / stride 1 convolution as an implicit GEMM on the gfx950 fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain -> same numerics class as the CPU reference).
//
// Replaces, on the hot path of the reference:
//   nn.Conv2d(c_in, c_out, 3, padding=1)            src/unet.py:36,38      (ConvBlock.body.0 / body.3)
//   nn.Conv2d(c_in + c_hid, 4*c_hid, 3, padding=1)  src/convlstm.py:9,13   (gate pre-activations; x-part and h-part
//                                                                          are issued as two calls, see engine)
//   torch.cat([x, skip], 1) feeding ConvBlock       src/unet.py:68         (virtual concat: two input pointers)
// and their data-gradients (same kernel, weights packed flipped+transposed by cm_pack_conv3x3_dgrad).
//
// GEMM orientation: MFMA rows (A operand) = output channels, MFMA columns (B operand) = pixels, so that one
// accumulator register of a wave holds 32 consecutive pixels of one output channel -> 128-byte coalesced stores.
// K is ordered [channel pair][tap][channel parity] so that the two k-slices of one 32x32x2 MFMA are the same tap of
// two adjacent channels: the second k-slice is a constant LDS offset (one channel plane) folded into the lane base.
//
// A workgroup owns S samples x (TH x TW) pixels x (32*WM) output channels.  Per K-chunk of KC input channels it
// stages the haloed input tile [KC][S][TH+2][TW+2] and the weight slab [KC*9][32*WM] in LDS; global loads for
// chunk k+1 are issued before the MFMA phase of chunk k and written to LDS after it (register prefetch).
#include "common.h"
#include "../../include/climate_hip.h"

namespace {

constexpr int KCH_PACK = 16;  // packed K rows are padded to a multiple of the largest K-chunk any tile configuration uses

struct Conv3Args {
  const float* in0;
  const float* in1;
  long long st0, st1;  // sample strides (elements)
  int C0, C1;          // channels taken from in0 / in1 (virtual concat)
  const float* wp;     // packed weights [CinP*9][CoutP]
  const float* bias;   // [Cout] or null
  const float* resid;  // same addressing as out, or null
  long long str;
  float* out;
  long long sto;
  int N, H, W, Cout, CoutP, nchunks, tiles_x, tiles_y;
  int ksplit;  // > 1: blockIdx.z owns a share of the K-chunks and accumulates into a pre-zeroed output with atomics
};

template <int TH, int TW, int S, int WAVES, int NPT, int WM, int KC, bool DUAL>
__global__ __launch_bounds__(WAVES * 64) void conv3x3_mfma_kernel(Conv3Args a) {
  constexpr int THREADS = WAVES * 64;
  constexpr int PITCH = TW + 2;
  constexpr int SS = (TH + 2) * PITCH;
  constexpr int CS = S * SS;
  constexpr int BCO = 32 * WM;
  constexpr int XN = KC * CS;
  constexpr int WN4 = KC * 9 * BCO / 4;
  constexpr int NLX = (XN + THREADS - 1) / THREADS;
  constexpr int NLW = (WN4 + THREADS - 1) / THREADS;
  constexpr int PIX = S * TH * TW;
  static_assert(KC % 2 == 0, "KC must be even");
  static_assert(WAVES * NPT * 32 >= PIX, "block does not cover its pixel set");

  __shared__ __attribute__((aligned(16))) float Wl[KC * 9 * BCO];
  __shared__ __attribute__((aligned(16))) float Xl[XN];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;

  // scalar copies: the kernarg struct must never have its address taken (that would force a scratch copy)
  const float* const a_in0 = a.in0;
  const float* const a_in1 = a.in1;
  const long long a_st0 = a.st0, a_st1 = a.st1;
  const int a_C0 = a.C0, a_C1 = a.C1, a_CoutP = a.CoutP;
  const float* const a_wp = a.wp;

  int bx = blockIdx.x;
  const int tx = bx % a.tiles_x;
  bx /= a.tiles_x;
  const int ty = bx % a.tiles_y;
  const int g = bx / a.tiles_y;
  const int x0 = tx * TW, y0 = ty * TH, n0 = g * S;
  const int co0 = blockIdx.y * BCO;
  const int HW = a.H * a.W;

  // ---- chunk-invariant staging offsets (relative to sample n0, first channel of the chunk) ----
  int goff0[NLX];
  int goff1[DUAL ? NLX : 1];
#pragma unroll
  for (int i = 0; i < NLX; ++i) {
    const int e = tid + i * THREADS;
    const int c = e / CS, r1 = e % CS;
    const int s = r1 / SS, r2 = r1 % SS;
    const int row = r2 / PITCH, col = r2 % PITCH;
    const int gy = y0 - 1 + row, gx = x0 - 1 + col;
    const bool ok = (e < XN) && (n0 + s < a.N) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    const int sp = c * HW + gy * a.W + gx;
    goff0[i] = ok ? (int)(s * a.st0) + sp : -1;
    if (DUAL) goff1[i] = ok ? (int)(s * a.st1) + sp : -1;
  }

  float xr[NLX];
  f32x4 wr[NLW];
  int cvalid_pending = 0;   // channel validity bound of the chunk currently held in xr (applied at LDS-store time)

  auto load_chunk = [&](int chunk) {
    const int ch0 = chunk * KC;
    const float* src;
    int cvalid;
    bool second = false;
    if (!DUAL || ch0 < a_C0) {
      src = a_in0 + (long long)n0 * a_st0 + (long long)ch0 * HW;
      cvalid = a_C0 - ch0;
    } else {
      src = a_in1 + (long long)n0 * a_st1 + (long long)(ch0 - a_C0) * HW;
      cvalid = a_C0 + a_C1 - ch0;
      second = true;
    }
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int e = tid + i * THREADS;
      const int c = e / CS;
      const int off = (DUAL && second) ? goff1[i] : goff0[i];
      // unconditional load from a clamped (always valid) address; the zero-select happens at LDS-store time so that
      // nothing consumes the loaded value before the MFMA phase (a select here would force s_waitcnt vmcnt(0))
      const bool ok = off >= 0 && c < cvalid;
      xr[i] = src[ok ? off : 0];
    }
    cvalid_pending = cvalid;
    const float* wsrc = a_wp + (long long)ch0 * 9 * a_CoutP + co0;
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
      const int f = min(tid + i * THREADS, WN4 - 1);
      const int r = f / (BCO / 4), q4 = f % (BCO / 4);
      wr[i] = *reinterpret_cast<const f32x4*>(wsrc + r * a_CoutP + 4 * q4);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int e = tid + i * THREADS;
      const bool ok = goff0[i] >= 0 && (e / CS) < cvalid_pending;   // goff0/goff1 share their validity
      if (e < XN) Xl[e] = ok ? xr[i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
      const int f = tid + i * THREADS;
      if (f < WN4) reinterpret_cast<f32x4*>(Wl)[f] = wr[i];
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
    xbase[p] = half * CS + s * SS + py * PITCH + px;
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

  const int wl_lane = half * BCO + l31;

  const int cps = (a.nchunks + a.ksplit - 1) / a.ksplit;
  const int chunk_begin = blockIdx.z * cps;
  const int chunk_end = min(a.nchunks, chunk_begin + cps);
  if (chunk_begin >= chunk_end) return;   // uniform for the whole workgroup
  load_chunk(chunk_begin);
  for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
    store_chunk();
    __syncthreads();
    if (chunk + 1 < chunk_end) load_chunk(chunk + 1);
#pragma unroll
    for (int cp = 0; cp < KC / 2; ++cp) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        float av[WM], bv[NPT];
#pragma unroll
        for (int m = 0; m < WM; ++m) av[m] = Wl[wl_lane + (cp * 9 + tap) * 2 * BCO + m * 32];
#pragma unroll
        for (int p = 0; p < NPT; ++p) bv[p] = Xl[xbase[p] + cp * 2 * CS + (tap / 3) * PITCH + (tap % 3)];
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
          for (int p = 0; p < NPT; ++p)
            acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[p], acc[m][p], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: D[i = cout][j = pixel]; lane holds pixel j = l31, rows (r&3) + 8*(r>>2) + 4*half ----
  // Bias and residual are folded into the accumulators in their own (wave-uniform) blocks first, so that their loads
  // are issued back to back and the store loop below contains no load: on CDNA4 vmcnt also counts stores, and a
  // load->use inside the store loop would make every store wait for the previous one.
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
          rv[r] = a.resid[ok ? obase[p] + (long long)co * HW : 0];   // same addressing as out (host-checked)
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][p][r] += rv[r];
      }
  }
#pragma unroll
  for (int m = 0; m < WM; ++m) {
#pragma unroll
    for (int p = 0; p < NPT; ++p) {
      if (pvalid[p]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (co < a.Cout) {
            if (a.ksplit > 1)
              unsafeAtomicAdd(a.out + obase[p] + (long long)co * HW, acc[m][p][r]);
            else
              a.out[obase[p] + (long long)co * HW] = acc[m][p][r];
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight packers
// fwd:   wp[((c/2)*9 + tap)*2 + (c&1)][o] = w[o][c_off + c][tap]
// dgrad: wp[((o/2)*9 + tap)*2 + (o&1)][c] = w[o][c_off + c][8 - tap]        (rows run over the FORWARD's outputs)
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ wp, int cout, int cin_total,
                                    int c_off, int cin, int rowsP /* padded k-channels */, int colsP, int dgrad) {
  const long long total = (long long)rowsP * 9 * colsP;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(i % colsP);
    const int row = (int)(i / colsP);
    const int par = row & 1, rt = row >> 1;
    const int tap = rt % 9, kc = (rt / 9) * 2 + par;
    float v = 0.f;
    if (!dgrad) {
      if (kc < cin && col < cout) v = w[((long long)col * cin_total + c_off + kc) * 9 + tap];
    } else {
      if (kc < cout && col < cin) v = w[((long long)kc * cin_total + c_off + col) * 9 + (8 - tap)];
    }
    wp[i] = v;
  }
}

__global__ void zero_out_kernel(float* __restrict__ out, long long sto, int n, long long per) {
  const long long total = (long long)n * per;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    out[(i / per) * sto + (i % per)] = 0.f;
}

// Batched packer: one launch re-packs every 3x3 weight of the model.  descs = (ndesc + 1) records of 8 int64:
// {w ptr, wp ptr, cout, cin_total, c_off, cin, dgrad, first block}; the last record only carries the total block count.
__global__ void pack_batch_kernel(const long long* __restrict__ descs, int ndesc) {
  const int d = cm_job_of_block(descs, ndesc);
  const long long* r = descs + d * 8;
  const float* w = reinterpret_cast<const float*>(r[0]);
  float* wp = reinterpret_cast<float*>(r[1]);
  const int cout = (int)r[2], cin_total = (int)r[3], c_off = (int)r[4], cin = (int)r[5], dgrad = (int)r[6];
  const int b0 = (int)r[7], nb = (int)descs[(d + 1) * 8 + 7] - b0;
  const int kch = dgrad ? cout : cin, ocs = dgrad ? cin : cout;
  const int rowsP = ((kch + KCH_PACK - 1) / KCH_PACK) * KCH_PACK, colsP = ((ocs + 31) / 32) * 32;
  const long long total = (long long)rowsP * 9 * colsP;
  for (long long i = (long long)(blockIdx.x - b0) * blockDim.x + threadIdx.x; i < total; i += (long long)nb * blockDim.x) {
    const int col = (int)(i % colsP);
    const int row = (int)(i / colsP);
    const int par = row & 1, rt = row >> 1;
    const int tap = rt % 9, kc = (rt / 9) * 2 + par;
    float v = 0.f;
    if (!dgrad) {
      if (kc < cin && col < cout) v = w[((long long)col * cin_total + c_off + kc) * 9 + tap];
    } else {
      if (kc < cout && col < cin) v = w[((long long)kc * cin_total + c_off + col) * 9 + (8 - tap)];
    }
    wp[i] = v;
  }
}

struct TileCfg {
  int th, tw, s, waves, npt, wm, kc;
};
constexpr int KCH = KCH_PACK;
constexpr TileCfg kCfgs[] = {
    {8, 24, 1, 3, 2, 1, 8},   // 0: 192 px x 32 co
    {8, 24, 1, 3, 2, 2, 8},   // 1: 192 px x 64 co
    {16, 24, 1, 4, 3, 1, 8},  // 2: 384 px x 32 co
    {8, 36, 1, 3, 3, 2, 8},   // 3: 288 px x 64 co
    {8, 12, 1, 3, 1, 2, 8},   // 4:  96 px x 64 co
    {4, 18, 4, 3, 3, 2, 8},   // 5: 4 samples x 72 px x 64 co
    {12, 18, 1, 4, 2, 2, 8},  // 6: 216 px (7 of 8 tiles) x 64 co
    {6, 9, 2, 4, 1, 1, 8},    // 7: 2 samples x 54 px x 32 co
    {6, 9, 2, 4, 1, 2, 8},    // 8: 2 samples x 54 px x 64 co
    {6, 9, 4, 4, 2, 2, 8},    // 9: 4 samples x 54 px x 64 co
    {8, 16, 1, 4, 1, 1, 8},   // 10: generic 128 px x 32 co
    {8, 16, 1, 4, 1, 2, 8},   // 11: generic 128 px x 64 co
    {12, 18, 1, 4, 2, 1, 8},  // 12: 216 px x 32 co
    {6, 9, 2, 4, 1, 1, 16},   // 13: as 7, 16-channel K chunks
    {8, 16, 1, 4, 1, 1, 16},  // 14: as 10, 16-channel K chunks
    {12, 18, 1, 4, 2, 1, 16}, // 15: as 12, 16-channel K chunks
    {6, 9, 2, 4, 1, 2, 16},   // 16: as 8, 16-channel K chunks
    {6, 9, 4, 4, 2, 1, 8},    // 17: 4 samples x 54 px x 32 co
    {6, 9, 4, 4, 2, 1, 16},   // 18: same, 16-channel K chunks
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

template <int I, bool DUAL>
int launch_cfg(const Conv3Args& a0, hipStream_t st) {
  constexpr TileCfg c = kCfgs[I];
  Conv3Args a = a0;
  a.tiles_x = cdiv(a.W, c.tw);
  a.tiles_y = cdiv(a.H, c.th);
  {
    const int nch = (a.C0 + a.C1 + c.kc - 1) / c.kc;
    if (a.ksplit > nch) a.ksplit = nch;
    if (a.ksplit < 1) a.ksplit = 1;
  }
  if (a.ksplit > 1) {
    const long long per = (long long)a.Cout * a.H * a.W;
    long long zb = ((long long)a.N * per + 255) / 256;
    zero_out_kernel<<<(int)(zb > 2048 ? 2048 : zb), 256, 0, st>>>(a.out, a.sto, a.N, per);
  }
  dim3 grid(a.tiles_x * a.tiles_y * cdiv(a.N, c.s), cdiv(a.Cout, 32 * c.wm), a.ksplit);
  a.nchunks = (a.C0 + a.C1 + c.kc - 1) / c.kc;
  if (a.C1 > 0 && (a.C0 % c.kc) != 0) return -22;  // a K-chunk must not straddle the two inputs
  conv3x3_mfma_kernel<c.th, c.tw, c.s, c.waves, c.npt, c.wm, c.kc, DUAL><<<grid, c.waves * 64, 0, st>>>(a);
  CM_CHECK_LAUNCH();
  return 0;
}

template <bool DUAL>
int dispatch(int cfg, const Conv3Args& a, hipStream_t st) {
  switch (cfg) {
    case 0: return launch_cfg<0, DUAL>(a, st);
    case 1: return launch_cfg<1, DUAL>(a, st);
    case 2: return launch_cfg<2, DUAL>(a, st);
    case 3: return launch_cfg<3, DUAL>(a, st);
    case 4: return launch_cfg<4, DUAL>(a, st);
    case 5: return launch_cfg<5, DUAL>(a, st);
    case 6: return launch_cfg<6, DUAL>(a, st);
    case 7: return launch_cfg<7, DUAL>(a, st);
    case 8: return launch_cfg<8, DUAL>(a, st);
    case 9: return launch_cfg<9, DUAL>(a, st);
    case 10: return launch_cfg<10, DUAL>(a, st);
    case 11: return launch_cfg<11, DUAL>(a, st);
    case 12: return launch_cfg<12, DUAL>(a, st);
    case 13: return launch_cfg<13, DUAL>(a, st);
    case 14: return launch_cfg<14, DUAL>(a, st);
    case 15: return launch_cfg<15, DUAL>(a, st);
    case 16: return launch_cfg<16, DUAL>(a, st);
    case 17: return launch_cfg<17, DUAL>(a, st);
    case 18: return launch_cfg<18, DUAL>(a, st);
    default: return -22;
  }
}

// Modelled cost (arbitrary units ~ MFMA issue slots on the busiest CU) used to pick a tile configuration.
double cfg_cost(const TileCfg& c, int N, int H, int W, int Cout) {
  const double blocks = (double)cdiv(W, c.tw) * cdiv(H, c.th) * cdiv(N, c.s) * cdiv(Cout, 32 * c.wm);
  const double work = (double)c.waves * c.npt * c.wm;                 // 32x32 tiles per block (incl. masked ones)
  const double lds_read = 1.0 + 0.35 * (double)(c.wm + c.npt) / (c.wm * c.npt);  // operand reads per MFMA
  const double halo = (double)(c.th + 2) * (c.tw + 2) / (c.th * c.tw);
  const double per_block = work * lds_read / c.waves + 0.25 * halo * c.s * c.th * c.tw / 100.0;
  const double slots = 256.0 * 2.0;                                   // ~2 resident blocks per CU
  const double rounds = ceil(blocks / slots);
  return rounds * per_block;
}

}  // namespace

extern "C" {

int cm_conv3x3_num_configs(void) { return kNumCfgs; }

int cm_conv3x3_pick_config(int n, int h, int w, int cout) {
  int best = 10;
  double bc = 1e300;
  for (int i = 0; i < kNumCfgs; ++i) {
    if (cout <= 32 && kCfgs[i].wm > 1) continue;
    const double c = cfg_cost(kCfgs[i], n, h, w, cout);
    if (c < bc) {
      bc = c;
      best = i;
    }
  }
  return best;
}

long long cm_conv3x3_packed_elems(int k_channels, int out_channels) {
  const long long rowsP = (long long)((k_channels + KCH - 1) / KCH) * KCH;
  const long long colsP = (long long)((out_channels + 31) / 32) * 32;
  return rowsP * 9 * colsP;
}

int cm_pack_conv3x3(const float* w, int cout, int cin_total, int c_off, int cin, int dgrad, float* wp,
                    cm_stream stream) {
  const int kch = dgrad ? cout : cin, ocs = dgrad ? cin : cout;
  const int rowsP = ((kch + KCH - 1) / KCH) * KCH, colsP = ((ocs + 31) / 32) * 32;
  const long long total = (long long)rowsP * 9 * colsP;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  pack_conv3x3_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(w, wp, cout, cin_total, c_off, cin, rowsP, colsP,
                                                               dgrad);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_pack_conv3x3_batch(const void* descs_dev, int ndesc, int total_blocks, cm_stream stream) {
  if (ndesc <= 0 || total_blocks <= 0) return -22;
  pack_batch_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>((const long long*)descs_dev, ndesc);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_conv3x3(const float* in0, long long st0, int c0, const float* in1, long long st1, int c1, const float* wp,
               const float* bias, const float* resid, long long st_resid, float* out, long long st_out, int n, int h,
               int w, int cout, int config, cm_stream stream) {
  if (n <= 0 || h <= 0 || w <= 0 || cout <= 0 || c0 <= 0 || c1 < 0) return -22;
  if (c1 > 0 && (c0 % 8) != 0) return -22;  // a K-chunk must not straddle the two inputs
  if (resid && st_resid != st_out) return -22;
  Conv3Args a;
  a.in0 = in0; a.in1 = in1; a.st0 = st0; a.st1 = st1; a.C0 = c0; a.C1 = c1;
  a.wp = wp; a.bias = bias; a.resid = resid; a.str = st_resid; a.out = out; a.sto = st_out;
  a.N = n; a.H = h; a.W = w; a.Cout = cout;
  a.CoutP = ((cout + 31) / 32) * 32;
  a.nchunks = 0;   // set per tile configuration
  a.tiles_x = a.tiles_y = 0;
  if (config < 0) config = cm_conv3x3_pick_config(n, h, w, cout);
  a.ksplit = config >> 8;                    // bits 8.. = K split over blockIdx.z (0/1 = none)
  config &= 0xff;
  if (a.ksplit > 1 && resid == out) return -22;   // the in-place residual would be zeroed
  return c1 > 0 ? dispatch<true>(config, a, (hipStream_t)stream) : dispatch<false>(config, a, (hipStream_t)stream);
}

}  // extern "C"
