// One ConvLSTM time step as ONE launch: recurrent projection W_h * h_{t-1} (3x3 conv on the f16 matrix cores, fp16x3) +
// x-projection residual + sigmoid / tanh + Hadamard state update.
//
// Reference: ConvLSTMCell.forward (src/convlstm.py:11-19): gates = conv(cat[x, h]); i, f, o, g = chunk(4);
// c' = sigmoid(f) c + sigmoid(i) tanh(g); h' = sigmoid(o) tanh(c').  The x-part of the gate convolution has no recurrence
// and is computed for all T steps by one cm_conv3x3_h3 launch (engine.convlstm_fwd); this kernel is the strictly serial
// rest of the cell -- north_star's "ConvLSTM gate fusion (4 x conv + sigmoid/tanh + Hadamard state update)".  It replaces
// two launches per step (the K-split projection storing partial slices + cm_lstm_gates_fwd_parts adding them).
//
// Work decomposition (output stationary, full reduction inside the workgroup, so the gate epilogue can be fused):
//   workgroup = (block of 8 hidden channels = 32 gate rows i|f|o|g x 8, group of S samples);
//   GEMM: rows = the 32 gate rows, columns = the group's S*h*w pixels in tiles of 32, K = Ch * 9;
//   the K dimension is split over the NW waves by input channel (wave w owns 16*NSTG channels): a wave loads ITS weight
//   fragments straight from the packed operand (cm_pack_conv3x3_h3_batch layout) into registers ONCE -- 72*NSTG VGPRs,
//   no LDS, no re-reads -- and runs all column tiles against them; h_{t-1} of the group is staged once, split into two
//   fp16 pieces (|h| < 1: fixed scale 2^13), zero haloed, in LDS; the NW partial accumulators of a tile are added through
//   LDS by the tile's owner wave, which then holds all four gates of 4 hidden channels x 32 pixels per lane: the
//   nonlinearities and the state update happen in registers and i, f, o, g, c', h' are written once.
#include "common.h"
#include "split_f16.h"
#include "../../include/climate_hip.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct LstmStep {
  const float* hprev; long long sh;     // h_{t-1} [B, Ch, hw]
  const u32x4* wph;                     // packed fp16x3 operand of W_h: [2][Ch/16][9][2][CoutP] records of 8 fp16
  const float* winv;                    // device scalar: 1 / weight scale
  float* gates; long long sg;           // [B, 4 Ch, hw]: x-projection (+ bias) in, activations i, f, o, g out
  const float* c_prev; long long scp;   // [B, Ch, hw]
  float* c_out; long long sco;
  float* h_out; long long sho;
  int B, Ch, H, W, S, CoutP;
};

template <int NW, int NSTG, int NT>
__global__ __launch_bounds__(NW * 64, 1) void lstm_step_fwd_kernel(LstmStep a) {
  extern __shared__ u32x4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int H = a.H, W = a.W, HW = H * W, Ch = a.Ch, S = a.S;
  const int PITCH = W + 2, SS = (H + 2) * PITCH, PH = S * SS;
  const int NOCT = Ch >> 3, nsteps = Ch >> 4;
  const int hb = blockIdx.x, b0 = blockIdx.y * S;

  // ---- this wave's weight fragments: stages [wave*NSTG, +NSTG), all 9 taps, both pieces ----
  f16x8 af[NSTG][9][2];
  {
    const long long piece_stride = (long long)nsteps * 9 * 2 * a.CoutP;
    const int grow = (l31 >> 3) * Ch + hb * 8 + (l31 & 7);          // gate row of MFMA row l31: q * Ch + hidden channel
#pragma unroll
    for (int st = 0; st < NSTG; ++st)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
          const long long rec = pc * piece_stride + ((long long)((wave * NSTG + st) * 9 + tap) * 2 + half) * a.CoutP + grow;
          af[st][tap][pc] = __builtin_bit_cast(f16x8, a.wph[rec]);
        }
  }

  // ---- stage h_{t-1} of the group: Xl[piece][octet][haloed pixel], zero halo, scale 2^13 ----
  for (int e = tid; e < NOCT * PH; e += NW * 64) {
    const int oct = e / PH, pix = e - oct * PH;
    const int s = pix / SS, r2 = pix - s * SS;
    const int row = r2 / PITCH, col = r2 - row * PITCH;
    const int y = row - 1, x = col - 1;
    const bool ok = (b0 + s < a.B) && y >= 0 && y < H && x >= 0 && x < W;
    u32x4 ph = {0u, 0u, 0u, 0u}, pl = {0u, 0u, 0u, 0u};
    if (ok) {
      const float* src = a.hprev + (long long)(b0 + s) * a.sh + (long long)(oct * 8) * HW + y * W + x;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(long long)j * HW] * 8192.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned hi, lo;
        split2_pair_f16(v[2 * q], v[2 * q + 1], hi, lo);
        ph[q] = hi; pl[q] = lo;
      }
    }
    lds[oct * PH + pix] = ph;
    lds[(NOCT + oct) * PH + pix] = pl;
  }

  // per-lane column bookkeeping: column = tile*32 + l31 -> (sample s, pixel p); invalid columns read sample 0, pixel 0
  int cbase[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = t * 32 + l31;
    const bool ok = col < S * HW;
    const int s = ok ? col / HW : 0, p = ok ? col - s * HW : 0;
    const int y = p / W, x = p - y * W;
    cbase[t] = s * SS + y * PITCH + x;          // top-left of the 3x3 window in haloed coordinates
  }
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  __syncthreads();

#pragma unroll
  for (int st = 0; st < NSTG; ++st) {
    const int oct = (wave * NSTG + st) * 2 + half;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int toff = (tap / 3) * PITCH + (tap % 3);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f16x8 bh = __builtin_bit_cast(f16x8, lds[oct * PH + cbase[t] + toff]);
        const f16x8 bl = __builtin_bit_cast(f16x8, lds[(NOCT + oct) * PH + cbase[t] + toff]);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[st][tap][1], bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[st][tap][0], bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[st][tap][0], bh, acc[t], 0, 0, 0);
      }
    }
  }

  // ---- add the NW partial sums of every tile through LDS (h tile is dead) ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds);            // [NW][NT][16][64]
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave * NT + t) * 16 + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  const float scale = a.winv[0] * (1.f / 8192.f);
  for (int t = wave; t < NT; t += NW) {                  // tile t is finished by wave t % NW
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float sum = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) sum += red[((w2 * NT + t) * 16 + r) * 64 + lane];
      v[r] = sum * scale;
    }
    const int col = t * 32 + l31;
    const int s = col / HW, p = col - s * HW;
    if (col < S * HW && b0 + s < a.B) {
      const int b = b0 + s;
      // accumulator register r of this lane = gate q = r >> 2 of hidden channel hb*8 + 4*half + (r & 3)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int ch = hb * 8 + 4 * half + jj;
        float* gp = a.gates + (long long)b * a.sg + (long long)ch * HW + p;
        const long long per = (long long)Ch * HW;
        const float pi = gp[0] + v[jj], pf = gp[per] + v[4 + jj], po = gp[2 * per] + v[8 + jj], pg = gp[3 * per] + v[12 + jj];
        const float i = sigmoid_acc(pi), f = sigmoid_acc(pf), o = sigmoid_acc(po), g = tanhf(pg);
        const float cp = a.c_prev ? a.c_prev[(long long)b * a.scp + (long long)ch * HW + p] : 0.f;
        const float cn = f * cp + i * g;
        gp[0] = i; gp[per] = f; gp[2 * per] = o; gp[3 * per] = g;
        a.c_out[(long long)b * a.sco + (long long)ch * HW + p] = cn;
        a.h_out[(long long)b * a.sho + (long long)ch * HW + p] = o * tanhf(cn);
      }
    }
  }
}

// ================================================================================================ backward step
// BPTT step t < T-1 as one launch: dh_t = dh_ext + conv3x3(dA_{t+1}, W_h^T flipped) (the recurrent data gradient, fp16x3
// on the matrix cores), then the gate backward of step t (src/convlstm.py:14-19 under autograd):
//   tc = tanh(c_t); dc_t = dc_carry + dh_t o (1 - tc^2); dA_t = [dc_t g i(1-i), dc_t c_{t-1} f(1-f), dh_t tc o(1-o),
//   dc_t i (1-g^2)]; dc_carry = dc_t f.
// Replaces cm_conv3x3_h3 (partial slices) + cm_lstm_gates_bwd_parts.  Workgroup = (32 hidden channels, one sample), 8
// waves; the reduction over the 4 Ch gate channels x 9 taps is split over the waves, each streaming ITS weight fragments
// from the packed operand into registers in PHASES of two 16-channel stages (144 VGPRs); dA_{t+1} of the sample is staged
// once, split into two fp16 pieces with a per-sample power-of-two scale (its magnitudes are unbounded), WITHOUT a halo: a
// tap outside the image reads a zero record.
struct LstmStepBwd {
  const float* dA_next; long long sdn;  // d(pre-activations) of step t+1 [B, 4 Ch, hw]
  const u32x4* wpd;                     // packed fp16x3 data-gradient operand of W_h (k = 4 Ch gate channels, outputs = Ch)
  const float* winv;
  float* gates; long long sg;           // step t: activations in, d(pre-activations) out
  const float* c_prev; long long scp;   // c_{t-1} (nullable: t = 0)
  const float* c_cur; long long scc;    // c_t
  const float* dh_ext; long long sde;   // external gradient wrt h_t (nullable)
  float* dc;                            // [B, Ch, hw] carried dL/dc
  int B, Ch, H, W;
};

template <int PH>     // phases of two stages per wave: 4 Ch = 8 waves * PH * 32 gate channels
__global__ __launch_bounds__(512, 1) void lstm_step_bwd_kernel(LstmStepBwd a) {
  extern __shared__ u32x4 lds[];
  __shared__ unsigned smax;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int H = a.H, W = a.W, HW = H * W, Ch = a.Ch, G = 4 * Ch;
  const int NOCT = G >> 3, nsteps = G >> 4;
  const int hb = blockIdx.x, b = blockIdx.y;
  const int ZREC = 2 * NOCT * HW;                 // index of the all-zero record
  if (tid == 0) smax = 0u;
  if (tid == 1) lds[ZREC] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();

  // ---- this wave's weight fragments of phase 0: issued FIRST so that they fly while dA is staged ----
  const long long piece_stride = (long long)nsteps * 9 * 2 * Ch;      // (outputs = Ch, a multiple of 32)
  const int arow = hb * 32 + l31;
  f16x8 af[2][9][2];
  auto load_phase = [&](int ph_) {
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
          const int step = (wave * PH + ph_) * 2 + st;
          af[st][tap][pc] = __builtin_bit_cast(f16x8, a.wpd[pc * piece_stride + ((long long)(step * 9 + tap) * 2 + half) * Ch + arow]);
        }
  };
  load_phase(0);

  // ---- stage dA_{t+1}[b]: one read into registers, the sample's maximum, then scale + split into LDS ----
  const float* src = a.dA_next + (long long)b * a.sdn;
  constexpr int MAXIT = 8;                       // records per thread: (4 Ch / 8) * hw <= 8 * 512 (host check)
  float v[MAXIT][8];
  {
    float m = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int e = tid + it * 512;
      const bool ok = e < NOCT * HW;
      const int oct = ok ? e / HW : 0, p = ok ? e - oct * HW : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[it][j] = ok ? src[(long long)(oct * 8 + j) * HW + p] : 0.f;
        m = fmaxf(m, fabsf(v[it][j]));
      }
    }
    m = wave_max_nonneg(m);
    if (lane == 0) atomicMax(&smax, __float_as_uint(m));
  }
  __syncthreads();
  const unsigned be = max((smax >> 23) & 0xffu, 13u);
  const float sc = __uint_as_float((267u - be) << 23);                       // largest |dA| -> [2^13, 2^14)
  const float inv = (smax == 0u) ? 0.f : __uint_as_float((be - 13u) << 23) * a.winv[0];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int e = tid + it * 512;
    if (e < NOCT * HW) {
      const int oct = e / HW, p = e - oct * HW;
      u32x4 ph, pl;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned hi, lo;
        split2_pair_f16(v[it][2 * q] * sc, v[it][2 * q + 1] * sc, hi, lo);
        ph[q] = hi; pl[q] = lo;
      }
      lds[oct * HW + p] = ph;
      lds[(NOCT + oct) * HW + p] = pl;
    }
  }
  // per-lane tap table: record offset (inside one octet plane) of the input pixel of every tap, or -1 (outside the image)
  int toff[2][9];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int col = t * 32 + l31;
    const int y = col / W, x = col - y * W;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
      toff[t][tap] = (col < HW && yy >= 0 && yy < H && xx >= 0 && xx < W) ? yy * W + xx : -1;
    }
  }
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  __syncthreads();

#pragma unroll
  for (int ph_ = 0; ph_ < PH; ++ph_) {
    if (ph_ > 0) load_phase(ph_);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the whole phase's fragments were issued back to back above
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int oct = ((wave * PH + ph_) * 2 + st) * 2 + half;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int o = toff[t][tap];
          const f16x8 bh = __builtin_bit_cast(f16x8, lds[o >= 0 ? oct * HW + o : ZREC]);
          const f16x8 bl = __builtin_bit_cast(f16x8, lds[o >= 0 ? (NOCT + oct) * HW + o : ZREC]);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[st][tap][1], bh, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[st][tap][0], bl, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[st][tap][0], bh, acc[t], 0, 0, 0);
        }
    }
  }

  // ---- reduce the 8 partial accumulators through LDS; wave w finishes registers [(w >> 1) * 4, +4) of tile w & 1 ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds);          // [8][2][16][64]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave * 2 + t) * 16 + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  const int t = wave & 1, r0 = (wave >> 1) * 4;
  const int col = t * 32 + l31;
  if (col < HW) {
    const long long per = (long long)Ch * HW;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int r = r0 + jj;
      float dh = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 8; ++w2) dh += red[((w2 * 2 + t) * 16 + r) * 64 + lane];
      dh *= inv;
      const int ch = hb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const long long e = (long long)ch * HW + col;
      if (a.dh_ext) dh += a.dh_ext[(long long)b * a.sde + e];
      float* gp = a.gates + (long long)b * a.sg + e;
      const float i = gp[0], f = gp[per], o = gp[2 * per], g = gp[3 * per];
      const float tc = tanhf(a.c_cur[(long long)b * a.scc + e]);
      const float cp = a.c_prev ? a.c_prev[(long long)b * a.scp + e] : 0.f;
      const long long di = (long long)b * per + e;
      const float dct = a.dc[di] + dh * o * (1.f - tc * tc);
      const float d_o = dh * tc;
      gp[0] = dct * g * i * (1.f - i);
      gp[per] = dct * cp * f * (1.f - f);
      gp[2 * per] = d_o * o * (1.f - o);
      gp[3 * per] = dct * i * (1.f - g * g);
      a.dc[di] = dct * f;
    }
  }
}

struct StepGeo {
  int nw, nstg, nt, s;
  size_t lds;
};

// Ch = NW * 16 * NSTG with (NW, NSTG) in {(4,1), (4,2), (8,2)}; S samples per workgroup such that S*h*w <= 128 columns
static bool step_geometry(int b, int ch, int h, int w, StepGeo* g) {
  if (ch == 64) { g->nw = 4; g->nstg = 1; }
  else if (ch == 128) { g->nw = 4; g->nstg = 2; }
  else if (ch == 256) { g->nw = 8; g->nstg = 2; }
  else return false;
  const int hw = h * w;
  if (hw > 64) return false;
  // one sample per workgroup unless that leaves the chip with fewer workgroups than a quarter of its CUs
  g->s = 1;
  g->nt = (g->s * hw + 31) / 32;
  const size_t xl = (size_t)2 * (ch / 8) * g->s * (h + 2) * (w + 2) * 16;
  const size_t rd = (size_t)g->nw * g->nt * 16 * 64 * 4;
  g->lds = xl > rd ? xl : rd;
  return g->lds <= 150 * 1024;
}

}  // namespace

extern "C" {

int cm_lstm_step_supported(int b, int ch, int h, int w) {
  StepGeo g;
  return b > 0 && step_geometry(b, ch, h, w, &g) ? 1 : 0;
}

int cm_lstm_step_fwd(const float* hprev, long long sh, const void* wph, const float* wscale_inv, float* gates,
                     long long sg, const float* c_prev, long long scp, float* c_out, long long sco, float* h_out,
                     long long sho, int b, int ch, int h, int w, cm_stream stream) {
  StepGeo g;
  if (b <= 0 || !hprev || !wph || !wscale_inv || !gates || !c_out || !h_out || !step_geometry(b, ch, h, w, &g)) return -22;
  LstmStep a;
  a.hprev = hprev; a.sh = sh; a.wph = (const u32x4*)wph; a.winv = wscale_inv; a.gates = gates; a.sg = sg;
  a.c_prev = c_prev; a.scp = scp; a.c_out = c_out; a.sco = sco; a.h_out = h_out; a.sho = sho;
  a.B = b; a.Ch = ch; a.H = h; a.W = w; a.S = g.s; a.CoutP = ((4 * ch + 31) / 32) * 32;
  const dim3 grid(ch / 8, (b + g.s - 1) / g.s);
  hipStream_t st = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)lstm_step_fwd_kernel<4, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    hipFuncSetAttribute((const void*)lstm_step_fwd_kernel<4, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    hipFuncSetAttribute((const void*)lstm_step_fwd_kernel<4, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    hipFuncSetAttribute((const void*)lstm_step_fwd_kernel<4, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    hipFuncSetAttribute((const void*)lstm_step_fwd_kernel<8, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    hipFuncSetAttribute((const void*)lstm_step_fwd_kernel<8, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    (void)hipGetLastError();
    attr = true;
  }
#define CM_STEP(NWV, NSV, NTV) \
  lstm_step_fwd_kernel<NWV, NSV, NTV><<<grid, NWV * 64, g.lds, st>>>(a)
  if (g.nw == 4 && g.nstg == 1) { if (g.nt == 1) CM_STEP(4, 1, 1); else CM_STEP(4, 1, 2); }
  else if (g.nw == 4 && g.nstg == 2) { if (g.nt == 1) CM_STEP(4, 2, 1); else CM_STEP(4, 2, 2); }
  else { if (g.nt == 1) CM_STEP(8, 2, 1); else CM_STEP(8, 2, 2); }
#undef CM_STEP
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_lstm_step_bwd_supported(int b, int ch, int h, int w) {
  // 4 ch = 8 waves x PH phases x 32 gate channels (PH = 1: ch 64, PH = 2: ch 128); the sample's dA fits LDS un-haloed
  if (b <= 0 || (ch != 64 && ch != 128) || h * w > 64 || h * w < 1 || (4 * ch / 8) * h * w > 8 * 512) return 0;
  const size_t xl = ((size_t)2 * (4 * ch / 8) * h * w + 1) * 16, rd = (size_t)8 * 2 * 16 * 64 * 4;
  return (xl > rd ? xl : rd) <= 150 * 1024;
}

int cm_lstm_step_bwd(const float* dA_next, long long sdn, const void* wpd, const float* wscale_inv, float* gates,
                     long long sg, const float* c_prev, long long scp, const float* c_cur, long long scc,
                     const float* dh_ext, long long sde, float* dc, int b, int ch, int h, int w, cm_stream stream) {
  if (!cm_lstm_step_bwd_supported(b, ch, h, w) || !dA_next || !wpd || !wscale_inv || !gates || !c_cur || !dc) return -22;
  LstmStepBwd a;
  a.dA_next = dA_next; a.sdn = sdn; a.wpd = (const u32x4*)wpd; a.winv = wscale_inv; a.gates = gates; a.sg = sg;
  a.c_prev = c_prev; a.scp = scp; a.c_cur = c_cur; a.scc = scc; a.dh_ext = dh_ext; a.sde = sde; a.dc = dc;
  a.B = b; a.Ch = ch; a.H = h; a.W = w;
  const size_t xl = ((size_t)2 * (4 * ch / 8) * h * w + 1) * 16, rd = (size_t)8 * 2 * 16 * 64 * 4;
  const size_t lds = xl > rd ? xl : rd;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)lstm_step_bwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    hipFuncSetAttribute((const void*)lstm_step_bwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    (void)hipGetLastError();
    attr = true;
  }
  const dim3 grid(ch / 32, b);
  if (ch == 64) lstm_step_bwd_kernel<1><<<grid, 512, lds, (hipStream_t)stream>>>(a);
  else lstm_step_bwd_kernel<2><<<grid, 512, lds, (hipStream_t)stream>>>(a);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
