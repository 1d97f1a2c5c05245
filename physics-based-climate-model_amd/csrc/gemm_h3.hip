// Dense GEMM on the f16 matrix cores with fp32-equivalent accuracy ("fp16x3", see split_f16.h / conv3x3_split.hip):
//   C[M,N] = mask(relu(op(A)[M,K] * op(B)[K,N] + bias[n]) + residual), fp32 in HBM on both sides.
//
// Used by the cnn_transformer path (BASELINE.json configs[3]; reference src/cnn_transformer.py): the linear layers of
// nn.TransformerEncoderLayer (in_proj, out_proj, linear1, linear2 -- F.linear: y = x W^T + b), their data and weight
// gradients, and the two stride-2 convolutions after im2col.
//
// Operand storage (row-major, leading dimension in elements):
//   TA = false: A is [M][K]  (k contiguous)        TA = true: A is [K][M]  (m contiguous: a transposed view)
//   TB = false: B is [N][K]  (k contiguous: W[out][in], so C = A W^T)      TB = true: B is [K][N] (n contiguous)
// Every fp32 operand tile is split into two fp16 pieces on its way into LDS (records of 8 consecutive k of one row /
// column = one 16-byte MFMA fragment) and multiplied as hi*hi + hi*lo + lo*hi with v_mfma_f32_32x32x16_f16.  Scaling:
// one power of two per operand from the RUNNING maximum of what the workgroup has staged (exact, no calibration); the
// accumulators follow when a scale shrinks.  Workgroup = 128 x 128 output tile, 4 waves as 2 x 2, 2 x 2 MFMA tiles per
// wave, 32-deep K stages; blockIdx.z splits K (atomic accumulation into a zeroed C).
#include <stdlib.h>
#include "common.h"
#include "split_f16.h"
#include "../../include/climate_hip.h"

namespace {

typedef unsigned int gu32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;      // [N] or null
  const float* resid;     // [res_rows][ldr] or null: C += resid[m % res_rows][n]
  const float* mask;      // [M][ldm] or null: C = mask[m][n] > 0 ? C : 0   (ReLU backward through a stored output)
  long long lda, ldb, ldc, ldr, ldm;
  int M, N, K, res_rows, relu, ksplit;
  const unsigned* rng;    // dropout (common.h) applied after bias / ReLU, before the residual; null = off
  unsigned site;
  float drop_p;
  float mask_scale;       // factor on the elements the mask keeps (ReLU backward through a DROPPED activation: 1/(1-p))
  float* a_rowsum;        // [M] or null (transposed A only): += sum over k of op(A)[m][k] -- with A = dY^T the bias gradient
  const gu32x4* Bp;       // packed B (cm_gemm_h3_pack_b_batch): [n tile of 64][32-deep image][piece][k octet][row] records
  const unsigned* bexp;   // ... and the biased exponent of its tensor's max |b| (the scale the packer used)
};

constexpr int GBM = 128, GBN = 128, GBK = 32;

// stage one 128 x 32 operand tile: returns this thread's 16 values (two records of 8 consecutive k)
//   k-contiguous storage: thread -> (row = tid / 2, k-half = tid % 2): four 16-byte loads along k
//   row-contiguous storage (transposed view): thread -> (k octet = tid / 64, rows lane and lane + 64): 16 strided loads,
//   each load instruction covering 64 consecutive rows of one k
template <bool TRANS>
__device__ __forceinline__ void gemm_load_tile(const float* __restrict__ P, long long ld, int rows, int K, int r0, int k0,
                                               int tid, float (&v)[16]) {
  if constexpr (!TRANS) {
    const int row = r0 + (tid >> 1), kk = k0 + (tid & 1) * 16;
    const bool rok = row < rows;
    const float* src = P + (long long)(rok ? row : 0) * ld + kk;
    const bool vec = (ld & 3) == 0 && (((uintptr_t)P) & 15) == 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (vec && rok && kk + 4 * q + 4 <= K) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + 4 * q);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] = (rok && kk + 4 * q + e < K) ? src[4 * q + e] : 0.f;
      }
    }
  } else {
    const int ko = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = r0 + lane + 64 * h;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + ko * 8 + j;
        v[8 * h + j] = (row < rows && k < K) ? P[(long long)k * ld + row] : 0.f;
      }
    }
  }
}

// LDS record index of (row, k octet) inside a [piece][4 octets][128 rows] image
template <bool TRANS>
__device__ __forceinline__ void gemm_store_tile(gu32x4* __restrict__ L, int tid, const float (&v)[16], float sc) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int row, oct;
    if constexpr (!TRANS) { row = tid >> 1; oct = (tid & 1) * 2 + h; }
    else                  { row = (tid & 63) + 64 * h; oct = tid >> 6; }
    gu32x4 ph, pl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned a_, b_;
      split2_pair_f16(v[8 * h + 2 * q] * sc, v[8 * h + 2 * q + 1] * sc, a_, b_);
      ph[q] = a_; pl[q] = b_;
    }
    L[(0 * 4 + oct) * 128 + row] = ph;
    L[(1 * 4 + oct) * 128 + row] = pl;
  }
}

// 64-row operand tile (the B side of the narrow 128 x 64 workgroup tile): one record of 8 consecutive k per thread
//   k-contiguous storage: thread -> (row = tid / 4, k octet = tid % 4): two 16-byte loads
//   row-contiguous storage: thread -> (k octet = tid / 64, row = lane): 8 strided loads of 64 consecutive rows each
template <bool TRANS>
__device__ __forceinline__ void gemm_load_tile64(const float* __restrict__ P, long long ld, int rows, int K, int r0, int k0,
                                                 int tid, float (&v)[16]) {
  if constexpr (!TRANS) {
    const int row = r0 + (tid >> 2), kk = k0 + (tid & 3) * 8;
    const bool rok = row < rows;
    const float* src = P + (long long)(rok ? row : 0) * ld + kk;
    const bool vec = (ld & 3) == 0 && (((uintptr_t)P) & 15) == 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (vec && rok && kk + 4 * q + 4 <= K) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + 4 * q);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] = (rok && kk + 4 * q + e < K) ? src[4 * q + e] : 0.f;
      }
    }
  } else {
    const int ko = tid >> 6, row = r0 + (tid & 63);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + ko * 8 + j;
      v[j] = (row < rows && k < K) ? P[(long long)k * ld + row] : 0.f;
    }
  }
#pragma unroll
  for (int j = 8; j < 16; ++j) v[j] = 0.f;
}

template <bool TRANS, int PITCH = 128>
__device__ __forceinline__ void gemm_store_tile64(gu32x4* __restrict__ L, int tid, const float (&v)[16], float sc) {
  const int row = TRANS ? (tid & 63) : (tid >> 2), oct = TRANS ? (tid >> 6) : (tid & 3);
  gu32x4 ph, pl;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned a_, b_;
    split2_pair_f16(v[2 * q] * sc, v[2 * q + 1] * sc, a_, b_);
    ph[q] = a_; pl[q] = b_;
  }
  L[(0 * 4 + oct) * PITCH + row] = ph;
  L[(1 * 4 + oct) * PITCH + row] = pl;
}

// BN = 128: 2 x 2 waves of 64 x 64;  BN = 64: 4 x 1 waves of 32 x 64 (twice the workgroups for the narrow GEMMs --
// 13824 x 256 x 256 is 216 tiles of 128 x 128 on 256 CUs, i.e. one 4-wave workgroup per CU and nothing to hide the
// staging arithmetic behind).
// BM = BN = 64: 2 x 2 waves of 32 x 32 (four times the workgroups: for the GEMMs whose 128 x 64 tiling still leaves the
// chip under ~4 workgroups per CU).
template <bool TA, bool TB, int BN = 128, int BM = 128, int NHALF = 2, bool PB = false>
__global__ __launch_bounds__(256) void gemm_h3_kernel(GemmArgs a) {
  static_assert(BM == 128 || (BM == 64 && BN == 64), "tile shapes: 128x128, 128x64, 64x64");
  static_assert(!PB || (BM == 64 && BN == 64 && !TB), "packed B: 64 x 64 tiles (its orientation is fixed by the packer)");
  constexpr int TI = (BM == 128 && BN == 128) ? 2 : 1, TJ = BM == 64 ? 1 : 2;
  constexpr int NH = (BM == 64 && BN == 64) ? NHALF : 1;      // 32-deep images per stage (64 x 64 tiles only)
  constexpr int LP = NH * 64 > 128 ? NH * 64 : 128;           // row pitch of the LDS images
  __shared__ gu32x4 Al[2 * 4 * LP], Bl[2 * 4 * LP];           // [piece][k octet][row / column]
  __shared__ unsigned smax[2][2];                       // [stage parity][A, B] posted maxima
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wr_off = BM == 64 ? (wave >> 1) * 32 : (BN == 128 ? (wave >> 1) * 64 : wave * 32);   // this wave's rows ..
  const int wc_off = BM == 64 ? (wave & 1) * 32 : (BN == 128 ? (wave & 1) * 64 : 0);            // .. and columns in the tile
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  // 64 x 64 tiles stage 64 of K per barrier pair: two 32-deep images side by side (rows 0-63 and 64-127 of the LDS arrays,
  // which are sized for the 128-row tiles), one maximum reduction and one rescale check for both
  constexpr int SK = GBK * NH;
  const int nstage = (a.K + SK - 1) / SK;
  const int sps = (nstage + a.ksplit - 1) / a.ksplit;
  const int s_begin = blockIdx.z * sps, s_end = min(nstage, s_begin + sps);
  if (s_begin >= s_end) return;

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto load_a = [&](int k0, float (&v)[16]) {
    if constexpr (BM == 128) gemm_load_tile<TA>(a.A, a.lda, a.M, a.K, m0, k0, tid, v);
    else gemm_load_tile64<TA>(a.A, a.lda, a.M, a.K, m0, k0, tid, v);
  };
  auto load_b = [&](int k0, float (&v)[16]) {
    if constexpr (PB) return;
    else if constexpr (BN == 128) gemm_load_tile<TB>(a.B, a.ldb, a.N, a.K, n0, k0, tid, v);
    else gemm_load_tile64<TB>(a.B, a.ldb, a.N, a.K, n0, k0, tid, v);
  };
  // packed B: image (n tile, 32-deep step) is 512 ready-made records; a thread moves records tid and tid + 256
  gu32x4 pbr[PB ? NH : 1][2];
  const long long nimg = (long long)((a.K + SK - 1) / SK) * NH;              // images per n tile (K padded by the packer)
  auto load_pb = [&](int s) {
    if constexpr (PB) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const gu32x4* src = a.Bp + ((long long)blockIdx.x * nimg + (long long)s * NH + h) * 512;
        pbr[h][0] = src[tid];
        pbr[h][1] = src[tid + 256];
      }
    }
  };
  auto store_pb = [&]() {
    if constexpr (PB) {
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = tid + 256 * i;                   // r = (piece*4 + octet)*64 + row
          Bl[(r >> 6) * LP + h * 64 + (r & 63)] = pbr[h][i];
        }
    }
  };

  if (tid < 4) smax[tid >> 1][tid & 1] = 0u;
  __syncthreads();
  float va[16], vb[16], va2[NH > 1 ? NH - 1 : 1][16], vb2[NH > 1 ? NH - 1 : 1][16];
  float rsum[2] = {0.f, 0.f};
  const bool sum_rows = TA && a.a_rowsum != nullptr && blockIdx.x == 0;     // (the first column of tiles does it once)
  unsigned bea = 0, beb = 0;                 // biased exponents of the running maxima
  const unsigned beb_fixed = PB ? a.bexp[0] : 0u;        // packed B: one scale for the whole tensor, known up front
  auto post = [&](int par) {
    float ma = 0.f, mb = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      ma = fmaxf(ma, fabsf(va[i]));
      if constexpr (!PB) mb = fmaxf(mb, fabsf(vb[i]));
    }
    if constexpr (NH > 1) {
#pragma unroll
      for (int h = 0; h < NH - 1; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          ma = fmaxf(ma, fabsf(va2[h][i]));
          if constexpr (!PB) mb = fmaxf(mb, fabsf(vb2[h][i]));
        }
    }
    ma = wave_max_nonneg(ma);
    if constexpr (!PB) mb = wave_max_nonneg(mb);
    if (lane == 0) {
      atomicMax(&smax[par][0], __float_as_uint(ma));
      if constexpr (!PB) atomicMax(&smax[par][1], __float_as_uint(mb));
    }
  };
  load_a(s_begin * SK, va);
  load_b(s_begin * SK, vb);
  load_pb(s_begin);
  if constexpr (NH > 1) {
#pragma unroll
    for (int h = 0; h < NH - 1; ++h) {
      load_a(s_begin * SK + (h + 1) * GBK, va2[h]);
      load_b(s_begin * SK + (h + 1) * GBK, vb2[h]);
    }
  }
  post(0);
  __syncthreads();
  for (int s = s_begin; s < s_end; ++s) {
    const int par = (s - s_begin) & 1;
    const unsigned na = max(bea, (smax[par][0] >> 23) & 0xffu);
    const unsigned nb = PB ? beb_fixed : max(beb, (smax[par][1] >> 23) & 0xffu);
    if (na != bea || nb != beb) {            // (workgroup uniform) a scale shrank: the accumulators follow
      if (bea != 0 || beb != 0) {
        const int d = (int)(max(na, 13u) - max(bea, 13u)) + (int)(max(nb, 13u) - max(beb, 13u));
        const float f = (bea == 0 || beb == 0 || d > 126) ? 0.f : __uint_as_float((unsigned)(127 - d) << 23);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
      }
      bea = na; beb = nb;
    }
    const float sa = __uint_as_float((267u - max(bea, 13u)) << 23), sb = __uint_as_float((267u - max(beb, 13u)) << 23);
    if constexpr (TA) {                      // row sums of op(A) ride along: this thread's rows are fixed (lane, lane + 64)
      if (sum_rows) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { rsum[0] += va[j]; rsum[1] += va[8 + j]; }
        if constexpr (NH > 1) {
#pragma unroll
          for (int h = 0; h < NH - 1; ++h)
#pragma unroll
            for (int j = 0; j < 8; ++j) rsum[0] += va2[h][j];
        }
      }
    }
    if constexpr (BM == 128) gemm_store_tile<TA>(Al, tid, va, sa);
    else gemm_store_tile64<TA, LP>(Al, tid, va, sa);
    if constexpr (PB) store_pb();
    else if constexpr (BN == 128) gemm_store_tile<TB>(Bl, tid, vb, sb);
    else gemm_store_tile64<TB, LP>(Bl, tid, vb, sb);
    if constexpr (NH > 1) {
#pragma unroll
      for (int h = 0; h < NH - 1; ++h) {
        gemm_store_tile64<TA, LP>(Al + 64 * (h + 1), tid, va2[h], sa);
        if constexpr (!PB) gemm_store_tile64<TB, LP>(Bl + 64 * (h + 1), tid, vb2[h], sb);
      }
    }
    __syncthreads();
    if (tid < 2) smax[par][tid] = 0u;
    if (s + 1 < s_end) {
      load_a((s + 1) * SK, va);
      load_b((s + 1) * SK, vb);
      load_pb(s + 1);
      if constexpr (NH > 1) {
#pragma unroll
        for (int h = 0; h < NH - 1; ++h) {
          load_a((s + 1) * SK + (h + 1) * GBK, va2[h]);
          load_b((s + 1) * SK + (h + 1) * GBK, vb2[h]);
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2 * NH; ++ks) {    // 16-deep MFMA k-steps of the stage; the lane half picks the octet
      const int oct = (ks & 1) * 2 + half, hb = (ks >> 1) * 64;
      f16x8 af[2][TI], bf[2][TJ];            // [piece][tile]
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
        for (int t = 0; t < TI; ++t)
          af[pc][t] = __builtin_bit_cast(f16x8, Al[(pc * 4 + oct) * LP + hb + wr_off + t * 32 + l31]);
#pragma unroll
        for (int t = 0; t < TJ; ++t)
          bf[pc][t] = __builtin_bit_cast(f16x8, Bl[(pc * 4 + oct) * LP + hb + wc_off + t * 32 + l31]);
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
        }
    }
    if (s + 1 < s_end) post(par ^ 1);
    __syncthreads();
  }

  if constexpr (TA) {
    if (sum_rows) {                          // 4 waves hold the 4 k octets of every row: combine through LDS, one atomic per row
      float* red = reinterpret_cast<float*>(Al);          // (free after the loop's last barrier)
      red[wave * 128 + lane] = rsum[0];
      red[wave * 128 + 64 + lane] = rsum[1];
      __syncthreads();
      if (tid < BM && m0 + tid < a.M)
        unsafeAtomicAdd(a.a_rowsum + m0 + tid, (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]));
    }
  }
  // ---- epilogue: undo the scales, bias / residual / ReLU / mask, store (or accumulate for split K) ----
  const float ia = bea <= 13u ? 0.f : __uint_as_float((bea - 13u) << 23);
  const float ib = beb <= 13u ? 0.f : __uint_as_float((beb - 13u) << 23);
  const DropSite drop = cm_drop_site(a.rng, a.site, a.drop_p);
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int n = n0 + wc_off + j * 32 + l31;
      if (n >= a.N) continue;
      const float bv = (a.bias && blockIdx.z == 0) ? a.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr_off + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= a.M) continue;
        float v = (acc[i][j][r] * ia) * ib + bv;
        if (a.ksplit > 1) {
          unsafeAtomicAdd(a.C + (long long)m * a.ldc + n, v);
          continue;
        }
        if (a.relu) v = fmaxf(v, 0.f);                              // (ReLU first: relu(conv) + pos_embedding)
        if (drop.thresh) v *= cm_drop_mul(drop, (unsigned)m * (unsigned)a.N + (unsigned)n);   // x + dropout(sublayer)
        if (a.resid) v += a.resid[(long long)(m % a.res_rows) * a.ldr + n];
        if (a.mask) v = a.mask[(long long)m * a.ldm + n] > 0.f ? v * a.mask_scale : 0.f;
        a.C[(long long)m * a.ldc + n] = v;
      }
    }
}

// ---- packed B operand -------------------------------------------------------------------------------------------------
// A weight matrix is the B operand of the forward (W [out][in]: rows n = out, k = in) and of the data gradient (rows n = in,
// k = out: the same storage read across) GEMMs of every step; splitting it inside every workgroup of every launch was
// half of the kernel's staging arithmetic.  The packer does it once per step and orientation:
//   job table (int64 x 8 per job): {w, out, be, N, K, ld, flags, first_block}; flags bit 0: B[n][k] = w[k * ld + n], bit 1:
//   the exponent `be` is filled by another job of the same tensor (both orientations of a weight share one).
//   pass 1 (amax): be[job] = max over the tensor of the biased exponent of |w| (atomic max of the float bits >> 23);
//   pass 2 (pack): block = (n tile of 64, 32-deep image): 512 records of 8 consecutive k of one row, scaled by
//   2^(140 - max(be, 13)) and split into fp16 pieces exactly as gemm_store_tile64 does; rows >= N and k >= K hold zeros.
__global__ __launch_bounds__(256) void gemm_pack_amax_kernel(const long long* __restrict__ table, int njobs) {
  const int job = blockIdx.y;
  if (job >= njobs) return;
  const long long* t = table + (long long)job * 8;
  if (t[6] & 2) return;                                  // (flag 2: another job of the same tensor fills this exponent)
  const float* w = reinterpret_cast<const float*>(t[0]);
  unsigned* be = reinterpret_cast<unsigned*>(t[2]);
  const long long rows = (t[6] & 1) ? t[4] : t[3], cols = (t[6] & 1) ? t[3] : t[4], ld = t[5];     // storage rows x cols
  float m = 0.f;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < rows * cols; i += 256LL * gridDim.x)
    m = fmaxf(m, fabsf(w[(i / cols) * ld + i % cols]));
  m = wave_max_nonneg(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(be, __float_as_uint(m) >> 23 & 0xffu);
}

__global__ __launch_bounds__(256) void gemm_pack_b_kernel(const long long* __restrict__ table, int njobs) {
  // which job owns this block: first_block ascending, so the job is the number of entries <= blockIdx.x minus one --
  // counted 64 entries at a time across the lanes (one load latency instead of njobs dependent ones)
  int job = -1;
  for (int j0 = 0; j0 < njobs; j0 += 64) {
    const int j = j0 + (threadIdx.x & 63);
    const bool le = j < njobs && table[(long long)j * 8 + 7] <= (long long)blockIdx.x;
    job += __popcll(__ballot(le));
  }
  const long long* t = table + (long long)job * 8;
  const float* w = reinterpret_cast<const float*>(t[0]);
  gu32x4* out = reinterpret_cast<gu32x4*>(t[1]);
  const unsigned be = *reinterpret_cast<const unsigned*>(t[2]);
  const int N = (int)t[3], K = (int)t[4];
  const long long ld = t[5];
  const bool trans = (t[6] & 1) != 0;
  const int blk = (int)(blockIdx.x - t[7]);
  const int nimg = ((K + 63) / 64) * 2;                 // images per n tile (K padded to the 64-deep stage)
  const int ntile = blk / nimg, img = blk % nimg;
  const float sc = __uint_as_float((267u - max(be, 13u)) << 23);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = threadIdx.x + 256 * i;                // r = (piece*4 + octet)*64 + row; this thread makes BOTH pieces of
    if (r >= 256) continue;                             // (octet, row) = r when r < 256
    const int oct = r >> 6, row = r & 63;
    const int n = ntile * 64 + row, k0 = img * 32 + oct * 8;
    float v[8];
    if (!trans && n < N && k0 + 8 <= K && (ld & 3) == 0 && (((uintptr_t)w) & 15) == 0) {    // one row's 8 k: two 16-byte loads
      const f32x4 lo = *reinterpret_cast<const f32x4*>(w + (long long)n * ld + k0);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(w + (long long)n * ld + k0 + 4);
      v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        v[j] = (n < N && k < K) ? (trans ? w[(long long)k * ld + n] : w[(long long)n * ld + k]) : 0.f;
      }
    }
    gu32x4 ph, pl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned a_, b_;
      split2_pair_f16(v[2 * q] * sc, v[2 * q + 1] * sc, a_, b_);
      ph[q] = a_; pl[q] = b_;
    }
    out[(long long)blk * 512 + r] = ph;
    out[(long long)blk * 512 + 256 + r] = pl;
  }
}

}  // namespace

extern "C" {

long long cm_gemm_h3_packed_b_bytes(int n, int k) { return (long long)cdiv(n, 64) * cdiv(k, 64) * 2 * 512 * 16; }

int cm_gemm_h3_pack_b_batch(const long long* table, int njobs, int total_blocks, int amax_blocks, cm_stream stream) {
  if (!table || njobs <= 0 || total_blocks <= 0 || amax_blocks <= 0) return -22;
  hipStream_t st = (hipStream_t)stream;
  gemm_pack_amax_kernel<<<dim3(amax_blocks, njobs), 256, 0, st>>>(table, njobs);
  CM_CHECK_LAUNCH();
  gemm_pack_b_kernel<<<total_blocks, 256, 0, st>>>(table, njobs);
  CM_CHECK_LAUNCH();
  return 0;
}

static int gemm_h3_launch(const float* a, long long lda, int trans_a, const float* b, long long ldb, int trans_b, float* c,
                          long long ldc, const float* bias, const float* resid, long long ldr, int res_rows,
                          const float* mask, long long ldm, float mask_scale, int relu, const unsigned* rng, unsigned site,
                          float drop_p, int m, int n, int k, int ksplit, int tile, float* a_rowsum, cm_stream stream,
                          const void* b_packed = nullptr, const unsigned* b_exp = nullptr);

int cm_gemm_h3(const float* a, long long lda, int trans_a, const float* b, long long ldb, int trans_b, float* c,
               long long ldc, const float* bias, const float* resid, long long ldr, int res_rows, const float* mask,
               long long ldm, float mask_scale, int relu, const unsigned* rng, unsigned site, float drop_p, int m, int n,
               int k, int ksplit, int tile, cm_stream stream) {
  return gemm_h3_launch(a, lda, trans_a, b, ldb, trans_b, c, ldc, bias, resid, ldr, res_rows, mask, ldm, mask_scale, relu,
                        rng, site, drop_p, m, n, k, ksplit, tile, nullptr, stream);
}

int cm_gemm_h3_wgrad(const float* dy, long long ld_dy, const float* x, long long ldx, float* dw, long long ld_dw,
                     float* dbias, int n_out, int k_in, int tokens, int ksplit, int tile, cm_stream stream) {
  return gemm_h3_launch(dy, ld_dy, 1, x, ldx, 1, dw, ld_dw, nullptr, nullptr, 0, 0, nullptr, 0, 1.f, 0, nullptr, 0, 0.f,
                        n_out, k_in, tokens, ksplit, tile, dbias, stream);
}

int cm_gemm_h3_pb(const float* a, long long lda, const void* b_packed, const unsigned* b_exp, float* c, long long ldc,
                  const float* bias, const float* resid, long long ldr, int res_rows, const float* mask, long long ldm,
                  float mask_scale, int relu, const unsigned* rng, unsigned site, float drop_p, int m, int n, int k,
                  cm_stream stream) {
  if (!b_packed || !b_exp) return -22;
  return gemm_h3_launch(a, lda, 0, a, 1, 0, c, ldc, bias, resid, ldr, res_rows, mask, ldm, mask_scale, relu, rng, site, drop_p,
                        m, n, k, 1, 3, nullptr, stream, b_packed, b_exp);
}

static int gemm_h3_launch(const float* a, long long lda, int trans_a, const float* b, long long ldb, int trans_b, float* c,
                          long long ldc, const float* bias, const float* resid, long long ldr, int res_rows,
                          const float* mask, long long ldm, float mask_scale, int relu, const unsigned* rng, unsigned site,
                          float drop_p, int m, int n, int k, int ksplit, int tile, float* a_rowsum, cm_stream stream,
                          const void* b_packed, const unsigned* b_exp) {
  if (m <= 0 || n <= 0 || k <= 0 || !a || !b || !c || lda <= 0 || ldb <= 0 || ldc < n) return -22;
  if (ksplit < 1) ksplit = 1;
  if (ksplit > 1 && (resid || mask || relu || (rng && drop_p > 0.f))) return -22;   // split K accumulates raw sums
  if (drop_p < 0.f || drop_p > 1.f || (long long)m * n > 0xffffffffLL) return -22;
  if (resid && (res_rows <= 0 || ldr < n)) return -22;
  if (mask && ldm < n) return -22;
  GemmArgs g;
  g.A = a; g.B = b; g.C = c; g.bias = bias; g.resid = resid; g.mask = mask;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.ldm = ldm;
  g.M = m; g.N = n; g.K = k; g.res_rows = res_rows > 0 ? res_rows : m; g.relu = relu;
  g.rng = drop_p > 0.f ? rng : nullptr; g.site = site; g.drop_p = drop_p; g.mask_scale = mask_scale;
  g.a_rowsum = a_rowsum;
  g.Bp = reinterpret_cast<const gu32x4*>(b_packed); g.bexp = b_exp;
  if (b_packed) {        // packed B: 64 x 64 tiles, 64-deep stages (the packer's image size), A stored [m][k]
    if (trans_a || ksplit != 1) return -22;
    g.ksplit = 1;
    gemm_h3_kernel<false, false, 64, 64, 2, true><<<dim3(cdiv(n, 64), cdiv(m, 64), 1), 256, 0, (hipStream_t)stream>>>(g);
    CM_CHECK_LAUNCH();
    return 0;
  }
  const int nstage = cdiv(k, GBK);              // (the 64 x 64 kernel counts 64-deep stages itself: empty shares return)
  g.ksplit = ksplit > nstage ? nstage : ksplit;
  hipStream_t st = (hipStream_t)stream;
  // narrow tiles when 128 x 128 tiles would give fewer than ~4 workgroups per CU (CM_GEMM_NARROW: 512 -> 9665, 1024 -> 9765, always -> 9755, never -> 8749 samples/s at config 4)
  static const long long narrow_below = getenv("CM_GEMM_NARROW") ? atoll(getenv("CM_GEMM_NARROW")) : 1024;
  const bool narrow = (long long)cdiv(n, GBN) * cdiv(m, GBM) * g.ksplit < narrow_below;
  // 64 x 64 tiles by default (CM_GEMM_SMALL=<n>: only below n workgroups of 128 x 64; 0 -> 10062, 1024 -> 10293,
  // always -> 10354 samples/s at config 4: at these sizes four times the workgroups beat the better LDS reuse of the
  // larger tiles everywhere)
  static const long long small_below = getenv("CM_GEMM_SMALL") ? atoll(getenv("CM_GEMM_SMALL")) : (1LL << 40);
  if (tile < 0 || tile > 3) return -22;
  const bool want_small = tile == 3 || (tile == 0 && (long long)cdiv(n, 64) * cdiv(m, GBM) * g.ksplit < small_below);
  if (want_small) {      // 64 x 64 tiles
    const dim3 grid(cdiv(n, 64), cdiv(m, 64), g.ksplit);
    // 32-deep images per stage: CM_GEMM_NH = 1 | 2 | 4 (default 2: 64 of K per barrier pair)
    static const int nh = getenv("CM_GEMM_NH") ? atoi(getenv("CM_GEMM_NH")) : 2;
#define CM_G64(NHV)                                                                                    \
    do {                                                                                               \
      if (!trans_a && !trans_b) gemm_h3_kernel<false, false, 64, 64, NHV><<<grid, 256, 0, st>>>(g);     \
      else if (!trans_a && trans_b) gemm_h3_kernel<false, true, 64, 64, NHV><<<grid, 256, 0, st>>>(g);  \
      else if (trans_a && trans_b) gemm_h3_kernel<true, true, 64, 64, NHV><<<grid, 256, 0, st>>>(g);    \
      else gemm_h3_kernel<true, false, 64, 64, NHV><<<grid, 256, 0, st>>>(g);                           \
    } while (0)
    if (nh == 1) CM_G64(1); else if (nh == 4) CM_G64(4); else CM_G64(2);
#undef CM_G64
    CM_CHECK_LAUNCH();
    return 0;
  }
  if (tile == 2 || (tile == 0 && narrow)) {
    const dim3 grid(cdiv(n, 64), cdiv(m, GBM), g.ksplit);
    if (!trans_a && !trans_b) gemm_h3_kernel<false, false, 64><<<grid, 256, 0, st>>>(g);
    else if (!trans_a && trans_b) gemm_h3_kernel<false, true, 64><<<grid, 256, 0, st>>>(g);
    else if (trans_a && trans_b) gemm_h3_kernel<true, true, 64><<<grid, 256, 0, st>>>(g);
    else gemm_h3_kernel<true, false, 64><<<grid, 256, 0, st>>>(g);
    CM_CHECK_LAUNCH();
    return 0;
  }
  const dim3 grid(cdiv(n, GBN), cdiv(m, GBM), g.ksplit);
  if (!trans_a && !trans_b) gemm_h3_kernel<false, false><<<grid, 256, 0, st>>>(g);
  else if (!trans_a && trans_b) gemm_h3_kernel<false, true><<<grid, 256, 0, st>>>(g);
  else if (trans_a && trans_b) gemm_h3_kernel<true, true><<<grid, 256, 0, st>>>(g);
  else gemm_h3_kernel<true, false><<<grid, 256, 0, st>>>(g);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
