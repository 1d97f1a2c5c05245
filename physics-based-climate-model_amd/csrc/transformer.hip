// The non-GEMM pieces of the cnn_transformer path (BASELINE.json configs[3]; reference src/cnn_transformer.py:4-54):
//   * residual + LayerNorm (post-norm nn.TransformerEncoderLayer: x = norm(x + sublayer(x)), eps 1e-5), fwd / bwd;
//   * multi-head self-attention core (softmax(Q K^T / sqrt(d)) V per (sample, head); 216 tokens, head_dim 32 at config
//     4), forward and backward, reading Q, K, V as column slices of the packed in_proj output [tokens, 3E];
//   * im2col / col2im of the two 3x3 stride-2 pad-1 convolutions (src/cnn_transformer.py:9-13), NCHW or token-major in;
//   * token-major <-> NCHW transposes around the decoder, ReLU and its mask, row-group sums (bias / pos-embedding grads).
// The dense contractions themselves run in gemm_h3.hip.  All tensors fp32; "tokens" = [B * S, E] row-major.
#include <stdlib.h>
#include "common.h"
#include "../../include/climate_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------ LayerNorm
// One wave per row; E <= 1024 (up to 16 values per lane).  s = x + r is stored (the backward needs it).
template <int VPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ s, float* __restrict__ y,
                                                     float* __restrict__ stats, int M, int E, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float v[VPL];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < E ? x[(long long)row * E + c] + (r ? r[(long long)row * E + c] : 0.f) : 0.f;
    sum += v[i];
  }
  const float mean = wave_sum(sum) / (float)E;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const float d = (lane + 64 * i < E) ? v[i] - mean : 0.f;
    sq += d * d;
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)E + eps);
  if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = lane + 64 * i;
    if (c < E) {
      if (s) s[(long long)row * E + c] = v[i];
      y[(long long)row * E + c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
    }
  }
}

// ds = rstd * (dy*gamma - mean(dy*gamma) - xhat * mean(dy*gamma*xhat)); dgamma += sum dy*xhat; dbeta += sum dy.
// Workgroup = 4 rows (one per wave) x ROWS_PER_WG row groups; the parameter gradients are combined through LDS and
// added with one atomic per column and workgroup.
// Side outputs (both nullable) for the sublayer that fed this LayerNorm through x + dropout(sublayer(x)): ds_drop = ds times
// the dropout multipliers of (rng, site, p) -- the gradient wrt the sublayer's output -- and dbias += column sums of that
// gradient (the bias gradient of the sublayer's last linear layer): saves a cm_dropout and a cm_rowgroup_sum launch.
template <int VPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ s, const float* __restrict__ stats,
                                                     const float* __restrict__ gamma, const float* __restrict__ dy,
                                                     float* __restrict__ ds, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int M, int E, int rows_per_wg,
                                                     float* __restrict__ ds_drop, float* __restrict__ dbias,
                                                     const unsigned* __restrict__ rng, unsigned site, float drop_p) {
  extern __shared__ float sh[];              // [3][E]
  for (int i = threadIdx.x; i < 3 * E; i += 256) sh[i] = 0.f;
  __syncthreads();
  const DropSite drop = cm_drop_site(rng, site, drop_p);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float ag[VPL], ab[VPL], abias[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) { ag[i] = 0.f; ab[i] = 0.f; abias[i] = 0.f; }
  const int r0 = blockIdx.x * rows_per_wg, rend = min(M, r0 + rows_per_wg);
  // RB rows of a wave are in flight at once (their loads are issued before the first reduction): with one row at a time a
  // wave waited out a full load latency per row, and the launch has too few waves per CU to hide it behind others
  constexpr int RB = VPL <= 4 ? 4 : 2;
  for (int rowb = r0 + wv; rowb < rend; rowb += 4 * RB) {
    float xh[RB][VPL], g[RB][VPL], dv[RB][VPL], mean[RB], rstd[RB], s1[RB], s2[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const int row = rowb + 4 * b;
      const bool rok = row < rend;
      mean[b] = rok ? stats[2 * row] : 0.f;
      rstd[b] = rok ? stats[2 * row + 1] : 0.f;
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        const bool ok = rok && c < E;
        dv[b][i] = ok ? dy[(long long)row * E + c] : 0.f;
        xh[b][i] = ok ? s[(long long)row * E + c] : 0.f;
      }
    }
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      s1[b] = 0.f; s2[b] = 0.f;
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        const bool ok = rowb + 4 * b < rend && c < E;
        xh[b][i] = ok ? (xh[b][i] - mean[b]) * rstd[b] : 0.f;
        g[b][i] = ok ? dv[b][i] * gamma[c] : 0.f;
        s1[b] += g[b][i];
        s2[b] += g[b][i] * xh[b][i];
        ag[i] += dv[b][i] * xh[b][i];
        ab[i] += dv[b][i];
      }
    }
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      s1[b] = wave_sum(s1[b]) / (float)E;
      s2[b] = wave_sum(s2[b]) / (float)E;
    }
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const int row = rowb + 4 * b;
      if (row >= rend) continue;
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < E) {
          const float d = rstd[b] * (g[b][i] - s1[b] - xh[b][i] * s2[b]);
          ds[(long long)row * E + c] = d;
          if (ds_drop || dbias) {
            const float dd = drop.thresh ? d * cm_drop_mul(drop, (unsigned)row * (unsigned)E + (unsigned)c) : d;
            if (ds_drop) ds_drop[(long long)row * E + c] = dd;
            abias[i] += dd;
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = lane + 64 * i;
    if (c < E) {
      atomicAdd(&sh[c], ag[i]);
      atomicAdd(&sh[E + c], ab[i]);
      if (dbias) atomicAdd(&sh[2 * E + c], abias[i]);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < E; c += 256) {
    unsafeAtomicAdd(dgamma + c, sh[c]);
    unsafeAtomicAdd(dbeta + c, sh[E + c]);
    if (dbias) unsafeAtomicAdd(dbias + c, sh[2 * E + c]);
  }
}

// ------------------------------------------------------------------------------------------------ attention
// qkv [B*S, 3E]: q at column h*D, k at E + h*D, v at 2E + h*D.  Workgroup = (query block of 64, head, sample); the
// head's K and V (S x D each) sit in LDS; a query row is handled by 4 adjacent lanes that split the keys (stride 4).
// P [B, H, S, S] (softmax probabilities) is written for the backward.  D <= 32, S <= 256.
constexpr int AQ = 64;       // queries per workgroup
constexpr int AKMAX = 64;    // keys per lane at most (S <= 256)

template <int D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ P,
                                                       float* __restrict__ O, int S, int E, int H, float scale,
                                                       const unsigned* __restrict__ rng, unsigned site, float drop_p) {
  extern __shared__ float sh[];              // K [S][D+1], V [S][D+1]
  float* Ks = sh;
  float* Vs = sh + (size_t)S * (D + 1);
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AQ;
  const long long base = (long long)b * S * 3 * E;
  for (int i = threadIdx.x; i < S * D; i += 256) {
    const int t = i / D, d = i % D;
    Ks[t * (D + 1) + d] = qkv[base + (long long)t * 3 * E + E + h * D + d];
    Vs[t * (D + 1) + d] = qkv[base + (long long)t * 3 * E + 2 * E + h * D + d];
  }
  __syncthreads();
  const int qi = q0 + (threadIdx.x >> 2), part = threadIdx.x & 3;
  const bool live = qi < S;
  float q[D];
#pragma unroll
  for (int d = 0; d < D; ++d) q[d] = live ? qkv[base + (long long)qi * 3 * E + h * D + d] * scale : 0.f;
  float sc[AKMAX];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < AKMAX; ++j) {
    const int key = part + 4 * j;
    float s = -INFINITY;
    if (key < S) {
      s = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) s += q[d] * Ks[key * (D + 1) + d];
    }
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < AKMAX; ++j) {
    sc[j] = (part + 4 * j < S) ? expf(sc[j] - mx) : 0.f;
    sum += sc[j];
  }
  sum += __shfl_xor(sum, 1, 64);
  sum += __shfl_xor(sum, 2, 64);
  const float inv = 1.f / sum;
  float o[D];
#pragma unroll
  for (int d = 0; d < D; ++d) o[d] = 0.f;
  const long long prow_i = (((long long)b * H + h) * S + (live ? qi : 0)) * S;
  float* prow = P + prow_i;
  // dropout on the attention probabilities (nn.MultiheadAttention(dropout=p)): P is stored UNdropped, the mask is a
  // function of the element index and is regenerated by the backward kernels
  const DropSite drop = cm_drop_site(rng, site, drop_p);
#pragma unroll
  for (int j = 0; j < AKMAX; ++j) {
    const int key = part + 4 * j;
    if (key < S) {
      const float p = sc[j] * inv;
      if (live) prow[key] = p;
      const float pd = drop.thresh ? p * cm_drop_mul(drop, (unsigned)(prow_i + key)) : p;
#pragma unroll
      for (int d = 0; d < D; ++d) o[d] += pd * Vs[key * (D + 1) + d];
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d) {
    o[d] += __shfl_xor(o[d], 1, 64);
    o[d] += __shfl_xor(o[d], 2, 64);
  }
  if (live && part == 0) {
#pragma unroll
    for (int d = 0; d < D; ++d) O[((long long)b * S + qi) * E + h * D + d] = o[d];
  }
}

// Backward, query side: dP = dO V^T, dS = P (dP - sum_j P dP) * scale, dQ = dS K.  dS is written over P's twin buffer.
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                         const float* __restrict__ dO, float* __restrict__ dS,
                                                         float* __restrict__ dqkv, int S, int E, int H, float scale,
                                                         const unsigned* __restrict__ rng, unsigned site, float drop_p) {
  extern __shared__ float sh[];
  float* Ks = sh;
  float* Vs = sh + (size_t)S * (D + 1);
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AQ;
  const long long base = (long long)b * S * 3 * E;
  for (int i = threadIdx.x; i < S * D; i += 256) {
    const int t = i / D, d = i % D;
    Ks[t * (D + 1) + d] = qkv[base + (long long)t * 3 * E + E + h * D + d];
    Vs[t * (D + 1) + d] = qkv[base + (long long)t * 3 * E + 2 * E + h * D + d];
  }
  __syncthreads();
  const int qi = q0 + (threadIdx.x >> 2), part = threadIdx.x & 3;
  const bool live = qi < S;
  float go[D];
#pragma unroll
  for (int d = 0; d < D; ++d) go[d] = live ? dO[((long long)b * S + qi) * E + h * D + d] : 0.f;
  const long long prow = (((long long)b * H + h) * S + (live ? qi : 0)) * S;
  float dp[AKMAX], pr[AKMAX];
  float dot = 0.f;
  const DropSite drop = cm_drop_site(rng, site, drop_p);
#pragma unroll
  for (int j = 0; j < AKMAX; ++j) {
    const int key = part + 4 * j;
    float v = 0.f, p = 0.f;
    if (key < S && live) {
      p = P[prow + key];
#pragma unroll
      for (int d = 0; d < D; ++d) v += go[d] * Vs[key * (D + 1) + d];
      if (drop.thresh) v *= cm_drop_mul(drop, (unsigned)(prow + key));     // d(P) = d(dropped P) * mask / (1 - p)
    }
    dp[j] = v; pr[j] = p;
    dot += p * v;
  }
  dot += __shfl_xor(dot, 1, 64);
  dot += __shfl_xor(dot, 2, 64);
  float dq[D];
#pragma unroll
  for (int d = 0; d < D; ++d) dq[d] = 0.f;
#pragma unroll
  for (int j = 0; j < AKMAX; ++j) {
    const int key = part + 4 * j;
    if (key < S && live) {
      const float ds = pr[j] * (dp[j] - dot) * scale;
      dS[prow + key] = ds;
#pragma unroll
      for (int d = 0; d < D; ++d) dq[d] += ds * Ks[key * (D + 1) + d];
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d) {
    dq[d] += __shfl_xor(dq[d], 1, 64);
    dq[d] += __shfl_xor(dq[d], 2, 64);
  }
  if (live && part == 0) {
#pragma unroll
    for (int d = 0; d < D; ++d) dqkv[base + (long long)qi * 3 * E + h * D + d] = dq[d];
  }
}

// Backward, key side: dV[key] = sum_q P[q][key] dO[q], dK[key] = sum_q dS[q][key] Q[q].  Workgroup = (key block of 64,
// head, sample); Q and dO of the head in LDS; a key is handled by 4 adjacent lanes splitting the queries -- lanes of a
// wave read 16 consecutive keys of a P row (coalesced along the row).
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                          const float* __restrict__ dS, const float* __restrict__ dO,
                                                          float* __restrict__ dqkv, int S, int E, int H,
                                                          const unsigned* __restrict__ rng, unsigned site, float drop_p) {
  extern __shared__ float sh[];
  float* Qs = sh;
  float* Gs = sh + (size_t)S * (D + 1);
  const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * AQ;
  const long long base = (long long)b * S * 3 * E;
  for (int i = threadIdx.x; i < S * D; i += 256) {
    const int t = i / D, d = i % D;
    Qs[t * (D + 1) + d] = qkv[base + (long long)t * 3 * E + h * D + d];
    Gs[t * (D + 1) + d] = dO[((long long)b * S + t) * E + h * D + d];
  }
  __syncthreads();
  // lane layout: keys along the fast lane index so that P[q][key .. key+15] is one contiguous run per query part
  const int key = k0 + (threadIdx.x & 63), part = threadIdx.x >> 6;      // 4 waves = 4 query parts
  const bool live = key < S;
  float dv[D], dk[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { dv[d] = 0.f; dk[d] = 0.f; }
  const long long pbase = ((long long)b * H + h) * S * S;
  const DropSite drop = cm_drop_site(rng, site, drop_p);
  for (int qq = part; qq < S; qq += 4) {
    float p = live ? P[pbase + (long long)qq * S + key] : 0.f;
    if (drop.thresh) p *= cm_drop_mul(drop, (unsigned)(pbase + (long long)qq * S + key));   // dV sees the DROPPED P
    const float ds = live ? dS[pbase + (long long)qq * S + key] : 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      dv[d] += p * Gs[qq * (D + 1) + d];
      dk[d] += ds * Qs[qq * (D + 1) + d];
    }
  }
  // combine the 4 query parts (one per wave) through LDS, dV then dK (the Q / dO tiles are dead by now)
  float* red = sh;                           // [4][64][D] floats (host sizes the LDS for max(tiles, this))
  const int l = threadIdx.x & 63;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    __syncthreads();
#pragma unroll
    for (int d = 0; d < D; ++d) red[(part * 64 + l) * D + d] = which == 0 ? dv[d] : dk[d];
    __syncthreads();
    if (part == 0 && live) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const float a = (red[(0 * 64 + l) * D + d] + red[(1 * 64 + l) * D + d]) +
                        (red[(2 * 64 + l) * D + d] + red[(3 * 64 + l) * D + d]);
        dqkv[base + (long long)key * 3 * E + (which == 0 ? 2 * E : E) + h * D + d] = a;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ im2col / col2im
// 3x3, stride 2, pad 1: out (oy, ox) reads in (2 oy - 1 + dy, 2 ox - 1 + dx).  col [B*OH*OW][ldc], column = ci*9 + tap
// (the order of a [Cout][Cin][3][3] weight row), columns >= Cin*9 zero.  Input NCHW (tokens = 0) or token-major [B*H*W][Cin].
__global__ void im2col_s2_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int Cin, int H, int W,
                                 int OH, int OW, int ldc, int tokens, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % ldc);
    const long long m = i / ldc;
    float v = 0.f;
    if (k < Cin * 9) {
      const int ci = k / 9, tap = k % 9;
      const int ox = (int)(m % OW), oy = (int)((m / OW) % OH), b = (int)(m / ((long long)OW * OH));
      const int iy = 2 * oy - 1 + tap / 3, ix = 2 * ox - 1 + tap % 3;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W)
        v = tokens ? x[(((long long)b * H + iy) * W + ix) * Cin + ci] : x[(((long long)b * Cin + ci) * H + iy) * W + ix];
    }
    col[i] = v;
  }
}

// dx (token-major [B*H*W][Cin]) = gather-sum of dcol: input pixel (iy, ix) receives tap (dy, dx) of output
// (oy, ox) = ((iy + 1 - dy) / 2, (ix + 1 - dx) / 2) when those are integers in range.
__global__ void col2im_s2_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B, int Cin, int H, int W,
                                 int OH, int OW, int ldc, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    const long long pix = i / Cin;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
    float acc = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int ty = iy + 1 - dy;
      if (ty < 0 || (ty & 1) || ty / 2 >= OH) continue;
#pragma unroll
      for (int dxx = 0; dxx < 3; ++dxx) {
        const int tx = ix + 1 - dxx;
        if (tx < 0 || (tx & 1) || tx / 2 >= OW) continue;
        acc += dcol[(((long long)b * OH + ty / 2) * OW + tx / 2) * ldc + ci * 9 + dy * 3 + dxx];
      }
    }
    dx[i] = acc;
  }
}

// ------------------------------------------------------------------------------------------------ small elementwise
// tokens [B][S][E] <-> NCHW [B][E][S] (S = H*W): 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
  __shared__ float t[32][33];                // in: [batch][R][C] -> out: [batch][C][R]
  const long long base = (long long)blockIdx.z * R * C;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8)
    if (r0 + k < R && c0 + tx < C) t[k][tx] = in[base + (long long)(r0 + k) * C + c0 + tx];
  __syncthreads();
  for (int k = ty; k < 32; k += 8)
    if (c0 + k < C && r0 + tx < R) out[base + (long long)(c0 + k) * R + r0 + tx] = t[tx][k];
}

__global__ void relu_kernel(float* __restrict__ x, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    x[i] = fmaxf(x[i], 0.f);
}

// out = in * (0 | 1/(1-p)) by the counter-based mask of (rng, site) -- element index = flat index
__global__ void dropout_kernel(const float* __restrict__ in, float* __restrict__ out, long long n,
                               const unsigned* __restrict__ rng, unsigned site, float p) {
  const DropSite drop = cm_drop_site(rng, site, p);
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * gridDim.x)
    out[i] = in[i] * cm_drop_mul(drop, (unsigned)i);
}
__global__ void rng_advance_kernel(unsigned* rng) {
  if (threadIdx.x == 0 && blockIdx.x == 0) rng[1] += 1u;
}

// g = (y > 0) ? g : 0 in place (y = the ReLU's stored OUTPUT)
__global__ void relu_mask_kernel(float* __restrict__ g, const float* __restrict__ y, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    if (!(y[i] > 0.f)) g[i] = 0.f;
}

// out[r % period][c] += sum over rows: x [rows][cols] (period = 1: column sums = bias gradients; period = S: the
// positional-embedding gradient).  One workgroup per (column block of 64, row slice); atomics into out.
__global__ __launch_bounds__(256) void rowgroup_sum_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           long long rows, int cols, int period, int slices) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  const int ph = blockIdx.y % period, sl = blockIdx.y / period;
  float a = 0.f;
  if (c < cols)
    for (long long r = ph + (long long)(sl * 4 + part) * period; r < rows; r += (long long)slices * 4 * period)
      a += x[r * cols + c];
  red[part][threadIdx.x & 63] = a;
  __syncthreads();
  if (part == 0 && c < cols) {
    const int l = threadIdx.x & 63;
    unsafeAtomicAdd(out + (long long)ph * cols + c, (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]));
  }
}

// out[r][c] = x[r][c] + add[r % period][c]
__global__ void add_rowgroup_kernel(const float* __restrict__ x, const float* __restrict__ add, float* __restrict__ out,
                                    long long rows, int cols, int period) {
  const long long total = rows * cols;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols;
    out[i] = x[i] + add[(r % period) * cols + i % cols];
  }
}

inline int grid_for(long long n) {
  long long b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int cm_layernorm_fwd(const float* x, const float* resid, const float* gamma, const float* beta, float* sum_out, float* y,
                     float* stats, int m, int e, float eps, cm_stream stream) {
  if (m <= 0 || e <= 0 || e > 1024 || !x || !y || !stats) return -22;
  hipStream_t st = (hipStream_t)stream;
  const int grid = cdiv(m, 4), vpl = cdiv(e, 64);
#define LNF(V) ln_fwd_kernel<V><<<grid, 256, 0, st>>>(x, resid, gamma, beta, sum_out, y, stats, m, e, eps)
  if (vpl <= 1) LNF(1); else if (vpl <= 2) LNF(2); else if (vpl <= 4) LNF(4); else if (vpl <= 8) LNF(8); else LNF(16);
#undef LNF
  CM_CHECK_LAUNCH();
  return 0;
}

static int ln_bwd_launch(const float* sum_in, const float* stats, const float* gamma, const float* dy, float* ds,
                         float* dgamma, float* dbeta, float* ds_drop, float* dbias, const unsigned* rng, unsigned site,
                         float drop_p, int m, int e, cm_stream stream) {
  if (m <= 0 || e <= 0 || e > 1024 || !sum_in || !stats || !dy || !ds) return -22;
  if (drop_p < 0.f || drop_p > 1.f || (long long)m * e > 0xffffffffLL) return -22;
  if (drop_p == 0.f) rng = nullptr;
  hipStream_t st = (hipStream_t)stream;
  static const int rpw_env = getenv("CM_LN_BWD_ROWS") ? atoi(getenv("CM_LN_BWD_ROWS")) : 32;
  const int rpw = rpw_env >= 4 ? rpw_env / 4 * 4 : 4, grid = cdiv(m, rpw), vpl = cdiv(e, 64);
  const size_t lds = 3 * (size_t)e * sizeof(float);
#define LNB(V) ln_bwd_kernel<V><<<grid, 256, lds, st>>>(sum_in, stats, gamma, dy, ds, dgamma, dbeta, m, e, rpw, ds_drop, dbias, rng, site, drop_p)
  if (vpl <= 1) LNB(1); else if (vpl <= 2) LNB(2); else if (vpl <= 4) LNB(4); else if (vpl <= 8) LNB(8); else LNB(16);
#undef LNB
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_layernorm_bwd(const float* sum_in, const float* stats, const float* gamma, const float* dy, float* ds,
                     float* dgamma, float* dbeta, int m, int e, cm_stream stream) {
  return ln_bwd_launch(sum_in, stats, gamma, dy, ds, dgamma, dbeta, nullptr, nullptr, nullptr, 0, 0.f, m, e, stream);
}

int cm_layernorm_bwd_sublayer(const float* sum_in, const float* stats, const float* gamma, const float* dy, float* ds,
                              float* dgamma, float* dbeta, float* ds_drop, float* dbias, const unsigned* rng,
                              unsigned site, float drop_p, int m, int e, cm_stream stream) {
  if (!ds_drop && !dbias) return -22;
  return ln_bwd_launch(sum_in, stats, gamma, dy, ds, dgamma, dbeta, ds_drop, dbias, rng, site, drop_p, m, e, stream);
}

static bool attn_ok(int b, int s, int e, int h) {
  return b > 0 && s > 0 && s <= 4 * AKMAX && h > 0 && e % h == 0 && (e / h == 8 || e / h == 16 || e / h == 32);
}

int cm_attention_fwd(const float* qkv, float* p, float* o, const unsigned* rng, unsigned site, float drop_p, int b, int s,
                     int e, int h, cm_stream stream) {
  if (!attn_ok(b, s, e, h) || !qkv || !p || !o || drop_p < 0.f || drop_p > 1.f) return -22;
  if ((long long)b * h * s * s > 0xffffffffLL) return -22;     // the dropout index is 32 bits
  if (drop_p == 0.f) rng = nullptr;
  const int d = e / h;
  const size_t lds = 2 * (size_t)s * (d + 1) * sizeof(float);
  const dim3 grid(cdiv(s, AQ), h, b);
  const float scale = 1.f / sqrtf((float)d);
  hipStream_t st = (hipStream_t)stream;
  if (d == 32) attn_fwd_kernel<32><<<grid, 256, lds, st>>>(qkv, p, o, s, e, h, scale, rng, site, drop_p);
  else if (d == 16) attn_fwd_kernel<16><<<grid, 256, lds, st>>>(qkv, p, o, s, e, h, scale, rng, site, drop_p);
  else attn_fwd_kernel<8><<<grid, 256, lds, st>>>(qkv, p, o, s, e, h, scale, rng, site, drop_p);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_attention_bwd(const float* qkv, const float* p, const float* d_o, float* ds_scratch, float* dqkv,
                     const unsigned* rng, unsigned site, float drop_p, int b, int s, int e, int h, cm_stream stream) {
  if (!attn_ok(b, s, e, h) || !qkv || !p || !d_o || !ds_scratch || !dqkv || drop_p < 0.f || drop_p > 1.f) return -22;
  if (drop_p == 0.f) rng = nullptr;
  const int d = e / h;
  size_t lds = 2 * (size_t)s * (d + 1) * sizeof(float);
  const size_t red = 4 * 64 * (size_t)d * sizeof(float);
  const dim3 grid(cdiv(s, AQ), h, b);
  const float scale = 1.f / sqrtf((float)d);
  hipStream_t st = (hipStream_t)stream;
  if (d == 32) attn_bwd_q_kernel<32><<<grid, 256, lds, st>>>(qkv, p, d_o, ds_scratch, dqkv, s, e, h, scale, rng, site, drop_p);
  else if (d == 16) attn_bwd_q_kernel<16><<<grid, 256, lds, st>>>(qkv, p, d_o, ds_scratch, dqkv, s, e, h, scale, rng, site, drop_p);
  else attn_bwd_q_kernel<8><<<grid, 256, lds, st>>>(qkv, p, d_o, ds_scratch, dqkv, s, e, h, scale, rng, site, drop_p);
  CM_CHECK_LAUNCH();
  if (red > lds) lds = red;
  if (d == 32) attn_bwd_kv_kernel<32><<<grid, 256, lds, st>>>(qkv, p, ds_scratch, d_o, dqkv, s, e, h, rng, site, drop_p);
  else if (d == 16) attn_bwd_kv_kernel<16><<<grid, 256, lds, st>>>(qkv, p, ds_scratch, d_o, dqkv, s, e, h, rng, site, drop_p);
  else attn_bwd_kv_kernel<8><<<grid, 256, lds, st>>>(qkv, p, ds_scratch, d_o, dqkv, s, e, h, rng, site, drop_p);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_im2col_s2(const float* x, float* col, int b, int cin, int h, int w, int ldc, int tokens_in, cm_stream stream) {
  if (b <= 0 || cin <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 1) || ldc < cin * 9 || !x || !col) return -22;
  const long long total = (long long)b * (h / 2) * (w / 2) * ldc;
  im2col_s2_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(x, col, b, cin, h, w, h / 2, w / 2, ldc, tokens_in,
                                                                    total);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_col2im_s2(const float* dcol, float* dx_tokens, int b, int cin, int h, int w, int ldc, cm_stream stream) {
  if (b <= 0 || cin <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 1) || ldc < cin * 9 || !dcol || !dx_tokens) return -22;
  const long long total = (long long)b * h * w * cin;
  col2im_s2_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(dcol, dx_tokens, b, cin, h, w, h / 2, w / 2, ldc,
                                                                    total);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_transpose_batched(const float* in, float* out, int batch, int rows, int cols, cm_stream stream) {
  if (batch <= 0 || rows <= 0 || cols <= 0 || !in || !out) return -22;
  transpose_kernel<<<dim3(cdiv(cols, 32), cdiv(rows, 32), batch), 256, 0, (hipStream_t)stream>>>(in, out, rows, cols);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_relu(float* x, long long n, cm_stream stream) {
  if (n <= 0 || !x) return -22;
  relu_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(x, n);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_dropout(const float* in, float* out, long long n, const unsigned* rng, unsigned site, float p, cm_stream stream) {
  if (n <= 0 || n > 0xffffffffLL || !in || !out || !rng || p < 0.f || p > 1.f) return -22;
  dropout_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(in, out, n, rng, site, p);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_rng_advance(unsigned* rng, cm_stream stream) {
  if (!rng) return -22;
  rng_advance_kernel<<<1, 64, 0, (hipStream_t)stream>>>(rng);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_relu_mask(float* g, const float* y, long long n, cm_stream stream) {
  if (n <= 0 || !g || !y) return -22;
  relu_mask_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(g, y, n);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_add_rowgroup(const float* x, const float* add, float* out, long long rows, int cols, int period, cm_stream stream) {
  if (rows <= 0 || cols <= 0 || period <= 0 || !x || !add || !out) return -22;
  add_rowgroup_kernel<<<grid_for(rows * cols), 256, 0, (hipStream_t)stream>>>(x, add, out, rows, cols, period);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_rowgroup_sum(const float* x, float* out, long long rows, int cols, int period, cm_stream stream) {
  if (rows <= 0 || cols <= 0 || period <= 0 || rows % period || !x || !out) return -22;
  long long groups = rows / period;
  static const int max_slices = getenv("CM_COLSUM_SLICES") ? atoi(getenv("CM_COLSUM_SLICES")) : 128;   // (64 -> 128: 519 -> 376 us per step at config 4)
  int slices = (int)(groups / 64 < 1 ? 1 : (groups / 64 > max_slices ? max_slices : groups / 64));
  if (period > 1) slices = (int)(groups / 16 < 1 ? 1 : (groups / 16 > 8 ? 8 : groups / 16));
  rowgroup_sum_kernel<<<dim3(cdiv(cols, 64), period * slices), 256, 0, (hipStream_t)stream>>>(x, out, rows, cols, period,
                                                                                             slices);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
