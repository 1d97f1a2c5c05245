// Multi-head self-attention core on the f16 matrix cores with fp32-equivalent accuracy ("fp16x3", split_f16.h) for
// head_dim 32 and up to 224 tokens (BASELINE.json configs[3]: 216 tokens, 8 heads of 32; reference
// src/cnn_transformer.py:26-33 -> nn.MultiheadAttention inside nn.TransformerEncoderLayer).
//
// Everything is computed TRANSPOSED: a score tile is S^T = K Q^T (rows = keys, columns = queries), so in the MFMA
// accumulator layout (lane = column, 16 rows per lane) a lane owns ONE query and the softmax reductions over the keys are
// in-lane loops plus a single exchange between the two half-waves -- no cross-lane reduction tree.  The probabilities
// never leave the registers: P^T in accumulator layout becomes the B operand of O^T = V^T P^T after four
// v_permlane32_swap per 16 keys (which turn "rows r, r+4 of both half-waves" into "8 consecutive keys per lane") and the
// fp16 split.  Nothing of size S x S is written: the forward keeps the row statistics (max, sum) and the backward
// recomputes the probabilities from Q and K (flash-attention style), once per orientation:
//   attn_mfma_fwd     O = softmax(Q K^T / sqrt(d)) V, stats                      workgroup = (128 queries, head, sample)
//   attn_mfma_bwd_q   dQ = dS K, D = rowsum(dO o O)           (S^T orientation)  workgroup = (128 queries, head, sample)
//   attn_mfma_bwd_kv  dV = P_drop^T dO, dK = dS^T Q           (S orientation)    workgroup = (128 keys, head, sample)
// with dS = P o (dP - D) / sqrt(d), dP = (dO V^T) o dropout mask.  Dropout (common.h) multiplies P after the softmax;
// its mask is a function of the element index (b, h, q, key) and is regenerated wherever P is recomputed.
//
// Operand scaling: Q/sqrt(d), K, V, dO are scaled per (sample, head) by exact powers of two from their own maxima (the
// whole head is staged at once, so the maximum is known before anything is converted); P (<= 1/(1-p)) by 2^12; dS by the
// maximum of the workgroup's whole dS block.  LDS holds 16-byte records of 8 fp16: [row][octet of d] for Q, K, dO
// (a fragment = one record), [d][octet of keys] for the transposed operands V^T, K^T, Q^T, dO^T.
#include "common.h"
#include "split_f16.h"
#include "../../include/climate_hip.h"

namespace {

typedef unsigned int au32x4 __attribute__((ext_vector_type(4)));

constexpr int AD = 32;            // head dim
constexpr int AMAXT = 7;          // key / query tiles of 32 at most (S <= 224: the key-side backward needs 152 KB of LDS)
constexpr int AWQ = 128;          // rows (queries or keys) owned by a workgroup: 4 waves x 32
constexpr int RP = 4;             // records per row of a [row][4 octets] operand
// record index of (row, octet): the octet is XOR-swizzled with bits 2..3 of the row so that the 16 rows a b128 read
// touches per phase land in 16 different 4-bank groups (a plain pitch of 4 records would be a 4-way conflict)
__device__ __forceinline__ int rrec(int row, int oct) { return row * RP + (oct ^ ((row >> 2) & 3)); }

__device__ __forceinline__ float block_max3(float v, float* red, int slot) {
  v = wave_max_nonneg(v);
  if ((threadIdx.x & 63) == 0) red[slot * 8 + (threadIdx.x >> 6)] = v;      // (up to 8 waves per workgroup)
  return v;
}
template <int NW = 4>
__device__ __forceinline__ float red_max(const float* red, int slot) {
  float m = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) m = fmaxf(m, red[slot * 8 + w]);
  return m;
}

// scale 2^(140 - be) puts a maximum with biased exponent be into [2^13, 2^14); returns the scale, *inv its inverse
__device__ __forceinline__ float pow2_scale(float mx, float& inv) {
  const unsigned be = (__float_as_uint(mx) >> 23) & 0xffu;
  if (be == 0u) { inv = 0.f; return 0.f; }
  const unsigned b = be < 13u ? 13u : be;
  inv = __uint_as_float((b - 13u) << 23);
  return __uint_as_float((267u - b) << 23);
}

// Stage a [rows x 32] fp32 operand (row r at src + r * ld) as fp16 pieces: row-major records dst[piece][row][oct] (pitch
// RP) and / or transposed records dstT[piece][d][row octet] (pitch tp).  rows beyond `valid` are zero.  `pre` multiplies
// before the power-of-two scale `sc`.
template <int NT = 256>
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, long long ld, int valid, int rows, float pre,
                                           float sc, au32x4* dst, int dst_rows, au32x4* dstT, int tp) {
  for (int it = threadIdx.x; it < rows * 4; it += NT) {
    const int r = it >> 2, oct = it & 3;
    float v[8];
    if (r < valid) {
      const float4 a = *reinterpret_cast<const float4*>(src + (long long)r * ld + oct * 8);
      const float4 b = *reinterpret_cast<const float4*>(src + (long long)r * ld + oct * 8 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    au32x4 ph, pl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned a_, b_;
      split2_pair_f16(v[2 * q] * pre * sc, v[2 * q + 1] * pre * sc, a_, b_);
      ph[q] = a_; pl[q] = b_;
    }
    if (dst) {
      dst[0 * dst_rows * RP + rrec(r, oct)] = ph;
      dst[1 * dst_rows * RP + rrec(r, oct)] = pl;
    }
    if (dstT) {
      // element (d = oct*8 + j) of row r goes to record (d, r >> 3), half-word r & 7
      _Float16* th = reinterpret_cast<_Float16*>(dstT);
      const int tsz = AD * tp * 8;            // half-words per piece
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned hw = (j & 1) ? (ph[j >> 1] >> 16) : (ph[j >> 1] & 0xffffu);
        const unsigned lw = (j & 1) ? (pl[j >> 1] >> 16) : (pl[j >> 1] & 0xffffu);
        const int e = ((oct * 8 + j) * tp + (r >> 3)) * 8 + (r & 7);
        reinterpret_cast<unsigned short*>(th)[e] = (unsigned short)hw;
        reinterpret_cast<unsigned short*>(th)[tsz + e] = (unsigned short)lw;
      }
    }
  }
}

// max |x| of a [rows x 32] operand (for the scale), one thread-local partial
template <int NT = 256>
__device__ __forceinline__ float rows_absmax(const float* __restrict__ src, long long ld, int valid) {
  float m = 0.f;
  for (int it = threadIdx.x; it < valid * 8; it += NT) {
    const float4 a = *reinterpret_cast<const float4*>(src + (long long)(it >> 3) * ld + (it & 7) * 4);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
  }
  return m;
}

__device__ __forceinline__ f32x16 mma3(const au32x4 (&A)[2], const au32x4 (&B)[2], f32x16 c) {
  const f16x8 a0 = __builtin_bit_cast(f16x8, A[0]), a1 = __builtin_bit_cast(f16x8, A[1]);
  const f16x8 b0 = __builtin_bit_cast(f16x8, B[0]), b1 = __builtin_bit_cast(f16x8, B[1]);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c, 0, 0, 0);
}

// Accumulator tile (rows (r&3) + 8 (r>>2) + 4 half, column = lane & 31) -> B operand fragments of its two 16-row k-steps
// (lane: column, 8 consecutive rows 16 ks + 8 half ..), scaled by sc and split into fp16 pieces.
__device__ __forceinline__ void acc_to_b(const f32x16& t, float sc, au32x4 (&b0)[2], au32x4 (&b1)[2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned lo = __float_as_uint(t[8 * ks + i]), hi = __float_as_uint(t[8 * ks + 4 + i]);
      const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
      x[i] = __uint_as_float(r[0]);
      x[4 + i] = __uint_as_float(r[1]);
    }
    au32x4 ph, pl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned a_, b_;
      split2_pair_f16(x[2 * q] * sc, x[2 * q + 1] * sc, a_, b_);
      ph[q] = a_; pl[q] = b_;
    }
    if (ks == 0) { b0[0] = ph; b0[1] = pl; } else { b1[0] = ph; b1[1] = pl; }
  }
}

struct AttnArgs {
  const float* qkv;     // [B*S][3E]
  float* O;             // fwd: out [B*S][E]
  float* stats;         // [B][H][S][2] = {row max of the scaled scores, row sum of exp}
  const float* dO;      // bwd: [B*S][E]
  float* dqkv;          // bwd: [B*S][3E]
  float* Dq;            // bwd: [B][H][S] rowsum(P o dP), written by bwd_q, read by bwd_kv
  int S, E, H, NT;      // NT = ceil(S / 32)
  float scale;          // 1 / sqrt(d)
  const unsigned* rng;
  unsigned site;
  float drop_p;
};

// ------------------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void attn_mfma_fwd_kernel(AttnArgs a) {
  extern __shared__ au32x4 lds[];
  const int S = a.S, E = a.E, H = a.H, NT = a.NT;
  const int KP = NT * 32;                 // padded key count
  const int TP = (KP / 8) | 1;            // record pitch of the transposed operand (odd)
  au32x4* Kr = lds;                       // [2][KP][RP]
  au32x4* Vt = Kr + 2 * KP * RP;          // [2][32][TP]
  au32x4* Qr = Vt + 2 * AD * TP;          // [2][AWQ][RP]
  __shared__ float red[24];
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AWQ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const float* base = a.qkv + (long long)b * S * 3 * E + h * AD;
  const long long ld = 3LL * E;
  const int nq = min(AWQ, S - q0);

  // maxima -> scales (identical in both query blocks of a head: K and V are staged whole)
  const float mq = rows_absmax(base + (long long)q0 * ld, ld, nq) * a.scale;
  const float mk = rows_absmax(base + E, ld, S), mv = rows_absmax(base + 2 * E, ld, S);
  block_max3(mq, red, 0); block_max3(mk, red, 1); block_max3(mv, red, 2);
  for (int i = tid; i < 2 * KP * RP + 2 * AD * TP; i += 256) lds[i] = au32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  float iq, ik, iv;
  const float sq = pow2_scale(red_max(red, 0), iq);
  const float sk = pow2_scale(red_max(red, 1), ik);
  const float sv = pow2_scale(red_max(red, 2), iv);
  stage_rows(base + (long long)q0 * ld, ld, nq, AWQ, a.scale, sq, Qr, AWQ, nullptr, 0);
  stage_rows(base + E, ld, S, KP, 1.f, sk, Kr, KP, nullptr, 0);
  stage_rows(base + 2 * E, ld, S, KP, 1.f, sv, nullptr, 0, Vt, TP);
  __syncthreads();

  // ---- scores S^T = K Q^T: this wave's 32 queries against every key tile ----
  au32x4 bq[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) bq[ks][pc] = Qr[pc * AWQ * RP + rrec(wave * 32 + l31, 2 * ks + half)];
  f32x16 acc[AMAXT];
  const float is = iq * ik;
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < AMAXT; ++kt) {
    if (kt < NT) {
      f32x16 c;
#pragma unroll
      for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        au32x4 ak[2];
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) ak[pc] = Kr[pc * KP * RP + rrec(kt * 32 + l31, 2 * ks + half)];
        c = mma3(ak, bq[ks], c);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        c[r] = key < S ? c[r] * is : -INFINITY;
        mx = fmaxf(mx, c[r]);
      }
      acc[kt] = c;
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < AMAXT; ++kt)
    if (kt < NT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = expf(acc[kt][r] - mx);      // (-inf - finite -> 0)
        acc[kt][r] = e;
        sum += e;
      }
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
  const int qi = q0 + wave * 32 + l31;
  const long long row = ((long long)b * H + h) * S + qi;
  if (qi < S && half == 0) {
    a.stats[2 * row] = mx;
    a.stats[2 * row + 1] = sum;
  }
  // ---- O^T = V^T P^T ----
  const DropSite drop = cm_drop_site(a.rng, a.site, a.drop_p);
  constexpr float SP = 4096.f;            // 2^12: P / (1 - p) <= 2 for p <= 0.5 ... stays far below fp16's range
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
  for (int kt = 0; kt < AMAXT; ++kt)
    if (kt < NT) {
      f32x16 p = acc[kt];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = p[r] * inv;
        if (drop.thresh) v *= cm_drop_mul(drop, (unsigned)(row * S + key));
        p[r] = v;
      }
      au32x4 b0[2], b1[2];
      acc_to_b(p, SP, b0, b1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        au32x4 av[2];
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) av[pc] = Vt[(pc * AD + l31) * TP + kt * 4 + 2 * ks + half];
        o = mma3(av, ks == 0 ? b0 : b1, o);
      }
    }
  if (qi < S) {
    const float io = iv * (1.f / SP);
    float* op = a.O + ((long long)b * S + qi) * E + h * AD;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 v = make_float4(o[4 * g] * io, o[4 * g + 1] * io, o[4 * g + 2] * io, o[4 * g + 3] * io);
      *reinterpret_cast<float4*>(op + 8 * g + 4 * half) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------- backward, query side
// S^T orientation (lane = query).  Recomputes P^T, dP^T = V dO^T (masked), D = sum_keys P dP, dS^T = P (dP - D) / sqrt(d),
// then dQ^T = K^T dS^T.  Writes dQ into dqkv and D into Dq.
template <int NW>      // waves per workgroup = 32-query tiles it owns (8: the whole 216-token head in one workgroup, 2 waves per SIMD)
__global__ __launch_bounds__(NW * 64) void attn_mfma_bwd_q_kernel(AttnArgs a) {
  constexpr int AWQ = NW * 32;
  extern __shared__ au32x4 lds[];
  const int S = a.S, E = a.E, H = a.H, NT = a.NT;
  const int KP = NT * 32, TP = (KP / 8) | 1;
  au32x4* Kr = lds;                       // [2][KP][RP]   K rows       (A of the scores)
  au32x4* Vr = Kr + 2 * KP * RP;          // [2][KP][RP]   V rows       (A of dP^T = V dO^T)
  au32x4* Kt = Vr + 2 * KP * RP;          // [2][32][TP]   K transposed (A of dQ^T = K^T dS^T)
  au32x4* Qr = Kt + 2 * AD * TP;          // [2][AWQ][RP]  Q rows       (B of the scores)
  au32x4* Gr = Qr + 2 * AWQ * RP;         // [2][AWQ][RP]  dO rows      (B of dP^T)
  __shared__ float red[32];
  __shared__ float dsmax[8];
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AWQ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const float* base = a.qkv + (long long)b * S * 3 * E + h * AD;
  const float* gbase = a.dO + (long long)b * S * E + h * AD;
  const long long ld = 3LL * E;
  const int nq = min(AWQ, S - q0);

  block_max3(rows_absmax<NW * 64>(base + (long long)q0 * ld, ld, nq) * a.scale, red, 0);
  block_max3(rows_absmax<NW * 64>(base + E, ld, S), red, 1);
  block_max3(rows_absmax<NW * 64>(base + 2 * E, ld, S), red, 2);
  block_max3(rows_absmax<NW * 64>(gbase + (long long)q0 * E, E, nq), red, 3);
  for (int i = tid; i < 4 * KP * RP + 2 * AD * TP + 4 * AWQ * RP; i += NW * 64) lds[i] = au32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  float iq, ik, iv, ig;
  const float sq = pow2_scale(red_max<NW>(red, 0), iq);
  const float sk = pow2_scale(red_max<NW>(red, 1), ik);
  const float sv = pow2_scale(red_max<NW>(red, 2), iv);
  const float sg = pow2_scale(red_max<NW>(red, 3), ig);
  stage_rows<NW * 64>(base + (long long)q0 * ld, ld, nq, AWQ, a.scale, sq, Qr, AWQ, nullptr, 0);
  stage_rows<NW * 64>(base + E, ld, S, KP, 1.f, sk, Kr, KP, Kt, TP);
  stage_rows<NW * 64>(base + 2 * E, ld, S, KP, 1.f, sv, Vr, KP, nullptr, 0);
  stage_rows<NW * 64>(gbase + (long long)q0 * E, E, nq, AWQ, 1.f, sg, Gr, AWQ, nullptr, 0);
  __syncthreads();

  au32x4 bq[2][2], bg[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      bq[ks][pc] = Qr[pc * AWQ * RP + rrec(wave * 32 + l31, 2 * ks + half)];
      bg[ks][pc] = Gr[pc * AWQ * RP + rrec(wave * 32 + l31, 2 * ks + half)];
    }
  const int qi = q0 + wave * 32 + l31;
  const bool qlive = qi < S;
  const long long row = ((long long)b * H + h) * S + (qlive ? qi : 0);
  const float mrow = a.stats[2 * row], linv = 1.f / a.stats[2 * row + 1];
  const DropSite drop = cm_drop_site(a.rng, a.site, a.drop_p);
  const float is = iq * ik, ip = iv * ig;
  // D = sum_keys P_drop dP = dO . O (the forward's output already contains the dropped probabilities): 16 elements per
  // half-wave, one exchange
  float dsum = 0.f;
  if (qlive) {
    const float* orow = a.O + ((long long)b * S + qi) * E + h * AD + 16 * half;
    const float* grow = a.dO + ((long long)b * S + qi) * E + h * AD + 16 * half;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 x = *reinterpret_cast<const float4*>(orow + 4 * i), y = *reinterpret_cast<const float4*>(grow + 4 * i);
      dsum += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    }
  }
  dsum += __shfl_xor(dsum, 32, 64);
  if (qlive && half == 0) a.Dq[row] = dsum;
  // One (key tile) step: P^T and the masked dP^T of this wave's 32 queries (12 MFMAs).
  auto tile = [&](int kt, f32x16& p, f32x16& dp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { p[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      au32x4 ak[2], av[2];
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        ak[pc] = Kr[pc * KP * RP + rrec(kt * 32 + l31, 2 * ks + half)];
        av[pc] = Vr[pc * KP * RP + rrec(kt * 32 + l31, 2 * ks + half)];
      }
      p = mma3(ak, bq[ks], p);
      dp = mma3(av, bg[ks], dp);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      p[r] = key < S ? expf(p[r] * is - mrow) * linv : 0.f;
      float g = dp[r] * ip;
      if (drop.thresh) g *= cm_drop_mul(drop, (unsigned)(row * S + key));
      dp[r] = g;
    }
  };
  f32x16 ds[AMAXT];
  // dS^T tiles (all kept in registers: they share one scale, their exact maximum)
  float dmx = 0.f;
#pragma unroll
  for (int kt = 0; kt < AMAXT; ++kt)
    if (kt < NT) {
      f32x16 p, dp;
      tile(kt, p, dp);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = p[r] * (dp[r] - dsum) * a.scale;
        p[r] = v;
        dmx = fmaxf(dmx, fabsf(v));
      }
      ds[kt] = p;
    }
  dmx = wave_max_nonneg(dmx);
  if (lane == 0) dsmax[wave] = dmx;
  __syncthreads();
  float ids;
  float dall = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) dall = fmaxf(dall, dsmax[w]);
  const float sds = pow2_scale(dall, ids);
  // dQ^T = K^T dS^T (rows = d, columns = queries)
  f32x16 dq;
#pragma unroll
  for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
  for (int kt = 0; kt < AMAXT; ++kt)
    if (kt < NT) {
      au32x4 b0[2], b1[2];
      acc_to_b(ds[kt], sds, b0, b1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        au32x4 at[2];
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) at[pc] = Kt[(pc * AD + l31) * TP + kt * 4 + 2 * ks + half];
        dq = mma3(at, ks == 0 ? b0 : b1, dq);
      }
    }
  if (qlive) {
    const float io = ik * ids;
    float* op = a.dqkv + ((long long)b * S + qi) * 3 * E + h * AD;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(op + 8 * g + 4 * half) =
          make_float4(dq[4 * g] * io, dq[4 * g + 1] * io, dq[4 * g + 2] * io, dq[4 * g + 3] * io);
  }
}

// --------------------------------------------------------------------------------------------- backward, key side
// S orientation (lane = key): S = Q K^T tiles (rows = queries).  dV^T = dO^T P_drop, dK^T = Q^T dS.  The row quantities
// (max, 1/sum, D) belong to the ROWS here, i.e. 16 different queries per lane: read from LDS tables.
// 8 waves: wave w owns key tile w & 3 and the query tiles of parity w >> 2 (two waves per SIMD hide the latency of the
// exp / hash / split arithmetic between the MFMAs -- with 152 KB of LDS only one workgroup fits a CU); the two partial
// results of a key tile are combined through LDS at the end.
__global__ __launch_bounds__(512) void attn_mfma_bwd_kv_kernel(AttnArgs a) {
  extern __shared__ au32x4 lds[];
  const int S = a.S, E = a.E, H = a.H, NT = a.NT;
  const int QP = NT * 32, TP = (QP / 8) | 1;
  au32x4* Qr = lds;                       // [2][QP][RP]   Q rows (scaled by 1/sqrt(d))   (A of the scores)
  au32x4* Gr = Qr + 2 * QP * RP;          // [2][QP][RP]   dO rows                        (A of dP = dO V^T)
  au32x4* Qt = Gr + 2 * QP * RP;          // [2][32][TP]   Q^T                            (A of dK^T = Q^T dS)
  au32x4* Gt = Qt + 2 * AD * TP;          // [2][32][TP]   dO^T                           (A of dV^T = dO^T P)
  au32x4* Kr = Gt + 2 * AD * TP;          // [2][AWQ][RP]  this workgroup's K rows        (B of the scores)
  au32x4* Vr = Kr + 2 * AWQ * RP;         // [2][AWQ][RP]  this workgroup's V rows        (B of dP)
  float* tab = reinterpret_cast<float*>(Vr + 2 * AWQ * RP);   // [3][QP]: row max, 1 / row sum, D
  __shared__ float red[32];
  const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * AWQ;
  const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int wave = wave8 & 3, par = wave8 >> 2;
  const float* base = a.qkv + (long long)b * S * 3 * E + h * AD;
  const float* gbase = a.dO + (long long)b * S * E + h * AD;
  const long long ld = 3LL * E;
  const int nk = min(AWQ, S - k0);

  block_max3(rows_absmax<512>(base, ld, S) * a.scale, red, 0);
  block_max3(rows_absmax<512>(base + E + (long long)k0 * ld, ld, nk), red, 1);
  block_max3(rows_absmax<512>(base + 2 * E + (long long)k0 * ld, ld, nk), red, 2);
  block_max3(rows_absmax<512>(gbase, E, S), red, 3);
  for (int i = tid; i < 4 * QP * RP + 4 * AD * TP + 4 * AWQ * RP; i += 512) lds[i] = au32x4{0u, 0u, 0u, 0u};
  const long long rbase = ((long long)b * H + h) * S;
  __syncthreads();
  for (int i = tid; i < QP; i += 512) {
    const bool ok = i < S;
    tab[i] = ok ? a.stats[2 * (rbase + i)] : 0.f;
    tab[QP + i] = ok ? 1.f / a.stats[2 * (rbase + i) + 1] : 0.f;
    tab[2 * QP + i] = ok ? a.Dq[rbase + i] : 0.f;
  }
  float iq, ik, iv, ig;
  const float sq = pow2_scale(red_max<8>(red, 0), iq);
  const float sk = pow2_scale(red_max<8>(red, 1), ik);
  const float sv = pow2_scale(red_max<8>(red, 2), iv);
  const float sg = pow2_scale(red_max<8>(red, 3), ig);
  stage_rows<512>(base, ld, S, QP, a.scale, sq, Qr, QP, Qt, TP);
  stage_rows<512>(gbase, E, S, QP, 1.f, sg, Gr, QP, Gt, TP);
  stage_rows<512>(base + E + (long long)k0 * ld, ld, nk, AWQ, 1.f, sk, Kr, AWQ, nullptr, 0);
  stage_rows<512>(base + 2 * E + (long long)k0 * ld, ld, nk, AWQ, 1.f, sv, Vr, AWQ, nullptr, 0);
  __syncthreads();

  au32x4 bk[2][2], bv[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      bk[ks][pc] = Kr[pc * AWQ * RP + rrec(wave * 32 + l31, 2 * ks + half)];
      bv[ks][pc] = Vr[pc * AWQ * RP + rrec(wave * 32 + l31, 2 * ks + half)];
    }
  const int key = k0 + wave * 32 + l31;
  const bool klive = key < S;
  const DropSite drop = cm_drop_site(a.rng, a.site, a.drop_p);
  const float is = iq * ik, ip = ig * iv;
  constexpr float SP = 4096.f;
  // The dS tiles of the whole reduction (all query tiles) must share one scale, and their maximum is not known before the
  // first MFMA.  As in the conv kernels: RUNNING maximum (wave uniform -- a wave owns its keys' accumulators), exact
  // power-of-two scale from it, and when the scale shrinks the dK accumulator follows (16 multiplies).  (An a-priori
  // bound 64 max|dO| max|V| was tried first: 2^15 above the real values with near-uniform attention, which pushed the
  // small fp16 piece into subnormals -- 5e-4 error in a training step.)
  unsigned be_run = 0;
  {
    f32x16 dv, dk;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dv[r] = 0.f; dk[r] = 0.f; }
#pragma unroll 1
    for (int qt = par; qt < NT; qt += 2) {
      f32x16 c, g;
#pragma unroll
      for (int r = 0; r < 16; ++r) { c[r] = 0.f; g[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        au32x4 aq[2], ag[2];
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
          aq[pc] = Qr[pc * QP * RP + rrec(qt * 32 + l31, 2 * ks + half)];
          ag[pc] = Gr[pc * QP * RP + rrec(qt * 32 + l31, 2 * ks + half)];
        }
        c = mma3(aq, bk[ks], c);      // S tile: rows = queries, column = this lane's key
        g = mma3(ag, bv[ks], g);      // dP tile
      }
      f32x16 pd;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float p = (qi < S && klive) ? expf(c[r] * is - tab[qi]) * tab[QP + qi] : 0.f;
        const float m = drop.thresh ? cm_drop_mul(drop, (unsigned)((rbase + qi) * S + key)) : 1.f;
        const float dp = g[r] * ip * m;
        const float v = p * (dp - tab[2 * QP + qi]);      // (Q^T below already carries the 1 / sqrt(d))
        pd[r] = p * m;
        c[r] = v;
      }
      {
        float tm = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) tm = fmaxf(tm, fabsf(c[r]));
        const unsigned be_t = (__float_as_uint(wave_max_nonneg(tm)) >> 23) & 0xffu;
        if (be_t > be_run) {                                   // (wave uniform) the scale 2^(140 - max(be, 13)) shrinks
          const unsigned bn = max(be_t, 13u), bo = max(be_run, 13u);
          if (be_run != 0u && bn > bo) {
            const unsigned d = bn - bo;
            const float f = d > 126u ? 0.f : __uint_as_float((127u - d) << 23);
#pragma unroll
            for (int r = 0; r < 16; ++r) dk[r] *= f;
          }
          be_run = be_t;
        }
        const float sds = __uint_as_float((267u - max(be_run, 13u)) << 23);
        au32x4 p0[2], p1[2], s0[2], s1[2];
        acc_to_b(pd, SP, p0, p1);
        acc_to_b(c, sds, s0, s1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          au32x4 agt[2], aqt[2];
#pragma unroll
          for (int pc = 0; pc < 2; ++pc) {
            agt[pc] = Gt[(pc * AD + l31) * TP + qt * 4 + 2 * ks + half];
            aqt[pc] = Qt[(pc * AD + l31) * TP + qt * 4 + 2 * ks + half];
          }
          dv = mma3(agt, ks == 0 ? p0 : p1, dv);
          dk = mma3(aqt, ks == 0 ? s0 : s1, dk);
        }
      }
    }
    // each wave undoes ITS scales (the two waves of a key tile ran their own running maxima), then the odd-parity wave
    // hands its partial to the even one through LDS (the operand images are dead by now)
    const float ids = be_run <= 13u ? 0.f : __uint_as_float((be_run - 13u) << 23);   // 2^(be - 140)
    const float iov = ig * (1.f / SP), iok = iq * ids;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dv[r] *= iov; dk[r] *= iok; }
    __syncthreads();
    float* comb = reinterpret_cast<float*>(lds) + (size_t)wave * 32 * 64;
    if (par == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { comb[r * 64 + lane] = dv[r]; comb[(16 + r) * 64 + lane] = dk[r]; }
    }
    __syncthreads();
    if (par == 0 && klive) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { dv[r] += comb[r * 64 + lane]; dk[r] += comb[(16 + r) * 64 + lane]; }
      float* vp = a.dqkv + ((long long)b * S + key) * 3 * E + 2 * E + h * AD;
      float* kp = a.dqkv + ((long long)b * S + key) * 3 * E + E + h * AD;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        *reinterpret_cast<float4*>(vp + 8 * g4 + 4 * half) = make_float4(dv[4 * g4], dv[4 * g4 + 1], dv[4 * g4 + 2], dv[4 * g4 + 3]);
        *reinterpret_cast<float4*>(kp + 8 * g4 + 4 * half) = make_float4(dk[4 * g4], dk[4 * g4 + 1], dk[4 * g4 + 2], dk[4 * g4 + 3]);
      }
    }
  }
}

bool mfma_ok(int b, int s, int e, int h) {
  return b > 0 && h > 0 && e % h == 0 && e / h == AD && s > 0 && s <= 32 * AMAXT && (e % 4) == 0;
}

}  // namespace

extern "C" {

int cm_attention_mfma_supported(int b, int s, int e, int h) { return mfma_ok(b, s, e, h) ? 1 : 0; }

int cm_attention_mfma_fwd(const float* qkv, float* stats, float* o, const unsigned* rng, unsigned site, float drop_p,
                          int b, int s, int e, int h, cm_stream stream) {
  if (!mfma_ok(b, s, e, h) || !qkv || !stats || !o || drop_p < 0.f || drop_p > 0.75f) return -22;
  if ((long long)b * h * s * s > 0xffffffffLL) return -22;
  AttnArgs a{};
  a.qkv = qkv; a.O = o; a.stats = stats; a.S = s; a.E = e; a.H = h; a.NT = cdiv(s, 32);
  a.scale = 1.f / sqrtf((float)AD);
  a.rng = drop_p > 0.f ? rng : nullptr; a.site = site; a.drop_p = drop_p;
  const int kp = a.NT * 32, tp = (kp / 8) | 1;
  const size_t lds = (size_t)(2 * kp * RP + 2 * AD * tp + 2 * AWQ * RP) * 16;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)attn_mfma_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) !=
        hipSuccess)
      return -22;
    attr = true;
  }
  attn_mfma_fwd_kernel<<<dim3(cdiv(s, AWQ), h, b), 256, lds, (hipStream_t)stream>>>(a);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_attention_mfma_bwd(const float* qkv, const float* stats, const float* o, const float* d_o, float* dq_rowsum,
                          float* dqkv, const unsigned* rng, unsigned site, float drop_p, int b, int s, int e, int h,
                          cm_stream stream) {
  if (!mfma_ok(b, s, e, h) || !qkv || !stats || !o || !d_o || !dq_rowsum || !dqkv || drop_p < 0.f || drop_p > 0.75f)
    return -22;
  AttnArgs a{};
  a.qkv = qkv; a.stats = const_cast<float*>(stats); a.O = const_cast<float*>(o); a.dO = d_o; a.Dq = dq_rowsum;
  a.dqkv = dqkv;
  a.S = s; a.E = e; a.H = h; a.NT = cdiv(s, 32);
  a.scale = 1.f / sqrtf((float)AD);
  a.rng = drop_p > 0.f ? rng : nullptr; a.site = site; a.drop_p = drop_p;
  const int kp = a.NT * 32, tp = (kp / 8) | 1;
  const int nwq = s > 128 ? 8 : 4;                              // query-side workgroup: 8 waves take a whole 216-token head
  const size_t lds_q = (size_t)(4 * kp * RP + 2 * AD * tp + 4 * nwq * 32 * RP) * 16;
  const size_t lds_kv = (size_t)(4 * kp * RP + 4 * AD * tp + 4 * AWQ * RP) * 16 + (size_t)3 * kp * 4;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)attn_mfma_bwd_q_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) !=
            hipSuccess ||
        hipFuncSetAttribute((const void*)attn_mfma_bwd_q_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) !=
            hipSuccess ||
        hipFuncSetAttribute((const void*)attn_mfma_bwd_kv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) !=
            hipSuccess)
      return -22;
    attr = true;
  }
  hipStream_t st = (hipStream_t)stream;
  if (nwq == 8) attn_mfma_bwd_q_kernel<8><<<dim3(cdiv(s, 256), h, b), 512, lds_q, st>>>(a);
  else attn_mfma_bwd_q_kernel<4><<<dim3(cdiv(s, 128), h, b), 256, lds_q, st>>>(a);
  CM_CHECK_LAUNCH();
  attn_mfma_bwd_kv_kernel<<<dim3(cdiv(s, AWQ), h, b), 512, lds_kv, st>>>(a);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
