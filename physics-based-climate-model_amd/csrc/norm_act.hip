// GroupNorm(8) + SiLU, forward and backward, one workgroup per (sample, group).
//
// Reference: nn.GroupNorm(8, c_out), nn.SiLU(inplace=True) in ConvBlock.body (src/unet.py:37,39); backward =
// native_group_norm_backward + silu_backward under loss.backward().  The forward optionally emits the per-(n,c)
// spatial mean of its output, which is exactly SEBlock's AdaptiveAvgPool2d(1) (src/unet.py:10,17), so the
// squeeze costs no extra pass.  The "gated" backward variant rebuilds the upstream gradient of the second
// GroupNorm from the SE / spatial-gate backward maps on the fly (see attention_gates.hip), so the full-size
// d(a2) tensor is never materialised.
//
// All passes are HBM/L2 streaming passes; statistics are fp32 sums about a point close to the mean (a true two-pass in
// the register-resident kernels, a sample-mean pivot in the streaming ones) so they stay within ~1e-7 of torch's CPU
// result.
#include <stdlib.h>
#include "common.h"
#include "se_wgrad.h"
#include "../../include/climate_hip.h"

namespace {

constexpr int GN_THREADS = 256;

template <int LPC>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPC / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sigmoid with the hardware exp / rcp (v_exp_f32, v_rcp_f32): ~1e-7 relative, far inside the 1e-4 parity budget
__device__ __forceinline__ float fast_sigmoid(float u) { return __frcp_rn(1.f + __expf(-u)); }

// The forward value a = SiLU(GroupNorm(v)) with every rounding pinned (explicit round-to-nearest intrinsics, no
// contraction freedom): the gated backward RECOMPUTES it from the pre-norm input instead of reading the stored
// activation back (one of its four large streams), and the amax tie test `a*s == max_c(a*s)` needs the recomputed
// value to be bit-identical to what the forward stored.  tests: test_gn_recompute_is_bit_exact.
__device__ __forceinline__ void gn_affine(float gamma, float beta, float mean, float rstd, float& ga, float& be) {
  ga = __fmul_rn(gamma, rstd);
  be = __fsub_rn(beta, __fmul_rn(__fmul_rn(mean, rstd), gamma));
}
__device__ __forceinline__ float gn_silu_parts(float v, float ga, float be, float& u, float& sg) {
  u = __fmaf_rn(v, ga, be);
  sg = __frcp_rn(__fadd_rn(1.f, __expf(-u)));
  return __fmul_rn(u, sg);
}
__device__ __forceinline__ float gn_silu_value(float v, float ga, float be) {
  float u, sg;
  return gn_silu_parts(v, ga, be, u, sg);
}

// PARTS: x arrives as nparts partial sums (slices zs apart: what a "partial slices" cm_conv3x3_h3 launch stores); the
// statistics pass adds them in slice order, WRITES the sum to xsum (the conv output the backward needs) and the apply pass
// re-reads it from there like the plain form re-reads x.
// GPART: the statistics come as partial {count, mean, M2} records written by the producing convolution's epilogue
// (cm_conv3x3_h3_gn: gpart [N][G][gslots][3]); they are merged with the parallel-variance formula and the statistics pass
// over x is skipped -- x is then read once.
template <bool VEC, int LPC, bool PARTS, bool GPART = false>   // LPC lanes cooperate on one channel (64: a wave, 16: four channels per wave)
__global__ __launch_bounds__(GN_THREADS) void gn_silu_fwd_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta,
                                                                   float* __restrict__ y, float* __restrict__ stats,
                                                                   float* __restrict__ pooled, int C, int HW,
                                                                   int G, float eps, const float* __restrict__ parts,
                                                                   long long zs, int nparts, float* xsum,
                                                                   const float* __restrict__ gpart = nullptr,
                                                                   int gslots = 0) {
  __shared__ float red[32];
  const int n = blockIdx.x / G, g = blockIdx.x % G;
  const int cpg = C / G;
  const long long base = ((long long)n * C + (long long)g * cpg) * HW;
  const int L = cpg * HW;
  const float* xg = PARTS ? xsum + base : x + base;
  float* yg = y + base;
  const int tid = threadIdx.x;
  if constexpr (PARTS) {
    // sum the slices into xsum first (one pass; the statistics below read it back from L2 like the plain form reads x)
    const float* pg = parts + base;
    float* sg = xsum + base;
    if (VEC) {
      for (int i = tid; i < L / 4; i += GN_THREADS) {
        float4 v = reinterpret_cast<const float4*>(pg)[i];
        for (int z = 1; z < nparts; ++z) {
          const float4 u = reinterpret_cast<const float4*>(pg + z * zs)[i];
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        reinterpret_cast<float4*>(sg)[i] = v;
      }
    } else {
      for (int i = tid; i < L; i += GN_THREADS) {
        float v = pg[i];
        for (int z = 1; z < nparts; ++z) v += pg[z * zs + i];
        sg[i] = v;
      }
    }
    __threadfence_block();
    __syncthreads();
  }

  // one statistics pass: sums of (x - pivot) and (x - pivot)^2.  The pivot is the mean of a 256-element sample spread
  // over the whole group, i.e. within ~sigma/16 of the group mean, so E[d^2] - E[d]^2 does not cancel (a single
  // element as pivot -- e.g. the zero-padded corner pixel -- can sit many sigma out, which cost 10x in the error of
  // rstd: tools/noise_probe.py).  The apply pass then re-reads the group from L2.
  float mean, rstd;
  if constexpr (GPART) {
    // merge the producer's partial statistics: n = sum n_i, mean = sum n_i mean_i / n, M2 = sum M2_i + sum n_i (mean_i - mean)^2
    const float* gp = gpart + (long long)blockIdx.x * gslots * 3;
    float cn = 0.f, cm = 0.f;
    for (int i = tid; i < gslots; i += GN_THREADS) { cn += gp[3 * i]; cm += gp[3 * i] * gp[3 * i + 1]; }
    cn = block_sum(cn, red);
    cm = block_sum(cm, red);
    mean = cm / cn;
    float m2 = 0.f;
    for (int i = tid; i < gslots; i += GN_THREADS) {
      const float d = gp[3 * i + 1] - mean;
      m2 += gp[3 * i + 2] + gp[3 * i] * d * d;
    }
    m2 = block_sum(m2, red);
    rstd = rsqrtf(fmaxf(m2 / cn, 0.f) + eps);
  } else {
  const float pivot = block_sum(xg[(long long)tid * L / GN_THREADS], red) * (1.f / GN_THREADS);
  float s1 = 0.f, s2 = 0.f;
  if (VEC) {
    const float4* x4 = reinterpret_cast<const float4*>(xg);
    for (int i = tid; i < L / 4; i += GN_THREADS) {
      const float4 v = x4[i];
      const float a = v.x - pivot, b = v.y - pivot, c = v.z - pivot, d = v.w - pivot;
      s1 += (a + b) + (c + d);
      s2 += (a * a + b * b) + (c * c + d * d);
    }
  } else {
    for (int i = tid; i < L; i += GN_THREADS) {
      const float a = xg[i] - pivot;
      s1 += a;
      s2 += a * a;
    }
  }
  s1 = block_sum(s1, red) / (float)L;
  s2 = block_sum(s2, red) / (float)L;
  mean = pivot + s1;
  const float var = fmaxf(s2 - s1 * s1, 0.f);
  rstd = rsqrtf(var + eps);
  }
  if (tid == 0) {
    stats[2 * blockIdx.x] = mean;
    stats[2 * blockIdx.x + 1] = rstd;
  }

  // apply: LPC lanes per channel so the per-channel mean of the output falls out of a sub-wave reduction
  const int lane = tid % LPC, wave = tid / LPC;
  for (int cl = wave; cl < cpg; cl += GN_THREADS / LPC) {
    const int c = g * cpg + cl;
    float ga, be;
    gn_affine(gamma[c], beta[c], mean, rstd, ga, be);
    const float* xc = xg + (long long)cl * HW;
    float* yc = yg + (long long)cl * HW;
    float ps = 0.f;
    if (VEC) {
      const float4* x4 = reinterpret_cast<const float4*>(xc);
      float4* y4 = reinterpret_cast<float4*>(yc);
      for (int i = lane; i < HW / 4; i += LPC) {
        const float4 v = x4[i];
        float4 o;
        o.x = gn_silu_value(v.x, ga, be);
        o.y = gn_silu_value(v.y, ga, be);
        o.z = gn_silu_value(v.z, ga, be);
        o.w = gn_silu_value(v.w, ga, be);
        y4[i] = o;
        ps += (o.x + o.y) + (o.z + o.w);
      }
    } else {
      for (int i = lane; i < HW; i += LPC) {
        const float o = gn_silu_value(xc[i], ga, be);
        yc[i] = o;
        ps += o;
      }
    }
    if (pooled) {
      ps = group_sum<LPC>(ps);
      if (lane == 0) pooled[(long long)n * C + c] = ps / (float)HW;
    }
  }
}

// Extra inputs of the gated variant (all per ConvBlock, see attention_gates.hip for their producers).
// y = SiLU(GroupNorm(x)) from STORED statistics (mean, rstd per (sample, group)): the recomputation the gated backward
// performs inline, exposed so that its bit-exactness against the forward kernels' output can be tested.
__global__ void gn_silu_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, const float* __restrict__ stats,
                                     float* __restrict__ y, int C, int HW, int G, long long total) {
  const int cpg = C / G;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long nc = i / HW;
    const int c = (int)(nc % C);
    const long long ng = (nc / C) * G + c / cpg;
    float ga, be;
    gn_affine(gamma[c], beta[c], stats[2 * ng], stats[2 * ng + 1], ga, be);
    y[i] = gn_silu_value(x[i], ga, be);
  }
}

struct GateBwd {
  const float* a2;     // [N,C,HW]  stored forward activation -- NOT read: recomputed bit-exactly (gn_silu_value)
  const float* dout;   // [N,C,HW]  gradient wrt the ConvBlock output
  const float* gate;   // [N,HW]    spatial gate (post-sigmoid)
  const float* dmap;   // [N,2,HW]  gradient wrt [mean_c U, max_c U]
  const float* umax;   // [N,HW]    max_c U as re-derived by gate_bwd_reduce (NOT the forward's stored map)
  const float* cnt;    // [N,HW]    number of channels attaining umax (>= 1), same launch
  const float* s;      // [N,C]     SE scale
  const float* dpool;  // [N,C]     gradient wrt the SE squeeze (pooled mean)
  SeWgradArgs se;      // side duty: the SE weight gradients of the preceding cm_se_excite_bwd (se.dsig NULL: none)
};

// Backward, one workgroup per (sample, group), one wave per channel.
//   pass 1: du = upstream * silu'(u) is written to dx (scratch use of the output buffer) while the per-channel and
//           per-group sums are accumulated;  pass 2: dx = rstd * (du*gamma - s1 - xhat*s2) in place (du comes back
//           from L2).  MODE 0: upstream gradient given as a tensor; MODE 1: rebuilt from the gate backward maps.
template <int MODE, int V, int LPC>   // V = 4: float4 path (HW % 4 == 0), V = 1: scalar; LPC lanes per channel
__global__ __launch_bounds__(GN_THREADS) void gn_silu_bwd_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta,
                                                                   const float* __restrict__ stats,
                                                                   const float* __restrict__ dA, long long st_dA,
                                                                   GateBwd gb, float* __restrict__ dx,
                                                                   float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, int C, int HW, int G) {
  __shared__ float acc[2];
  if (MODE == 1 && gb.se.dsig) {     // side duty (saves a launch): 32 SE weights per chunk, chunks strided over the grid
    __shared__ float part[8][33];
    for (int ch = blockIdx.x; ch * 32 < 2 * gb.se.C * gb.se.Cr; ch += gridDim.x) se_wgrad_chunk(gb.se, ch, part);
  }
  const int n = blockIdx.x / G, g = blockIdx.x % G;
  const int cpg = C / G;
  const int tid = threadIdx.x, lane = tid % LPC, wave = tid / LPC;
  const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
  const float inv_hw = 1.f / (float)HW, inv_c = 1.f / (float)C;
  if (tid < 2) acc[tid] = 0.f;
  __syncthreads();

  typedef float vec_t __attribute__((ext_vector_type(V)));
  const int HWV = HW / V;

  for (int cl = wave; cl < cpg; cl += GN_THREADS / LPC) {
    const int c = g * cpg + cl;
    const float ga = gamma[c], be = beta[c];
    const long long nc = (long long)n * C + c;
    const vec_t* xc = reinterpret_cast<const vec_t*>(x + nc * HW);
    vec_t* dxc = reinterpret_cast<vec_t*>(dx + nc * HW);
    const vec_t* dAc = MODE == 0 ? reinterpret_cast<const vec_t*>(dA + (long long)n * st_dA + (long long)c * HW) : nullptr;
    float gaf = 0.f, bef = 0.f;               // forward affine of this channel (recomputation of a2)
    if (MODE == 1) gn_affine(ga, be, mean, rstd, gaf, bef);
    const vec_t* doc = MODE == 1 ? reinterpret_cast<const vec_t*>(gb.dout + nc * HW) : nullptr;
    const vec_t* gtc = MODE == 1 ? reinterpret_cast<const vec_t*>(gb.gate + (long long)n * HW) : nullptr;
    const vec_t* dac = MODE == 1 ? reinterpret_cast<const vec_t*>(gb.dmap + (long long)n * 2 * HW) : nullptr;
    const vec_t* dmc = MODE == 1 ? reinterpret_cast<const vec_t*>(gb.dmap + ((long long)n * 2 + 1) * HW) : nullptr;
    const vec_t* mxc = MODE == 1 ? reinterpret_cast<const vec_t*>(gb.umax + (long long)n * HW) : nullptr;
    const vec_t* ctc = MODE == 1 ? reinterpret_cast<const vec_t*>(gb.cnt + (long long)n * HW) : nullptr;
    const float sc = MODE == 1 ? gb.s[nc] : 0.f;
    const float dpl = MODE == 1 ? gb.dpool[nc] * inv_hw : 0.f;
    float sd = 0.f, sdx = 0.f;
    for (int i = lane; i < HWV; i += LPC) {
      const vec_t xv = xc[i];
      vec_t up, ufv, sgv;
      if (MODE == 0) {
        up = dAc[i];
      } else {
        const vec_t dov = doc[i], gtv = gtc[i], dav = dac[i], dmv = dmc[i], mxv = mxc[i], ctv = ctc[i];
#pragma unroll
        for (int k = 0; k < V; ++k) {
          float uf, sgf;
          const float U = gn_silu_parts(xv[k], gaf, bef, uf, sgf) * sc;   // bit-exact forward product: tie test operand
          float dU = dov[k] * gtv[k] + dav[k] * inv_c;
          if (U == mxv[k]) dU += dmv[k] / ctv[k];
          up[k] = dU * sc + dpl;
          ufv[k] = uf;
          sgv[k] = sgf;
        }
      }
      vec_t duv;
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const float xh = (xv[k] - mean) * rstd;
        float u, sg;
        if (MODE == 1) {           // reuse the forward pre-activation and its sigmoid (one exp per element)
          u = ufv[k];
          sg = sgv[k];
        } else {
          u = xh * ga + be;
          sg = fast_sigmoid(u);
        }
        const float du = up[k] * (sg * (1.f + u * (1.f - sg)));
        duv[k] = du;
        sd += du;
        sdx += du * xh;
      }
      dxc[i] = duv;
    }
    sd = group_sum<LPC>(sd);
    sdx = group_sum<LPC>(sdx);
    if (lane == 0) {
      unsafeAtomicAdd(dbeta + c, sd);
      unsafeAtomicAdd(dgamma + c, sdx);
      atomicAdd(&acc[0], sd * ga);
      atomicAdd(&acc[1], sdx * ga);
    }
  }
  __syncthreads();   // also orders this wave's du stores before its own re-reads below (same lanes, same addresses)
  const float m = 1.f / (float)(cpg * HW);
  const float s1 = acc[0] * m, s2 = acc[1] * m;

  for (int cl = wave; cl < cpg; cl += GN_THREADS / LPC) {
    const int c = g * cpg + cl;
    const float ga = gamma[c];
    const long long nc = (long long)n * C + c;
    const vec_t* xc = reinterpret_cast<const vec_t*>(x + nc * HW);
    vec_t* dxc = reinterpret_cast<vec_t*>(dx + nc * HW);
    for (int i = lane; i < HWV; i += LPC) {
      const vec_t xv = xc[i];
      vec_t dv = dxc[i];
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const float xh = (xv[k] - mean) * rstd;
        dv[k] = rstd * (dv[k] * ga - s1 - xh * s2);
      }
      dxc[i] = dv;
    }
  }
}

// Register-resident variants (float4 path, group small enough): each wave owns whole 64-quad "rows" of channels and keeps
// them in registers between the reduction pass and the apply pass, so every tensor is read from HBM exactly once.
//   forward : read x, write y                      (2 passes instead of 3)
//   backward: read x, upstream; write dx           (3 / 4 passes instead of 6 / 7)
template <int MAXQ, int NT = GN_THREADS>
__global__ __launch_bounds__(NT) void gn_silu_fwd_reg_kernel(const float* __restrict__ x,
                                                                       const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta,
                                                                       float* __restrict__ y, float* __restrict__ stats,
                                                                       float* __restrict__ pooled, int C, int HW, int G,
                                                                       float eps) {
  __shared__ float red[32];
  __shared__ float psum[64];
  const int n = blockIdx.x / G, g = blockIdx.x % G;
  const int cpg = C / G;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = NT / 64;
  const int HWV = HW / 4, RPC = (HWV + 63) / 64, rows = cpg * RPC;
  const long long base = ((long long)n * C + (long long)g * cpg) * HW;
  const float4* xg = reinterpret_cast<const float4*>(x + base);
  float4* yg = reinterpret_cast<float4*>(y + base);
  if (tid < 64) psum[tid] = 0.f;
  // the group is register resident, so the statistics are a true two-pass: the mean first (sums shifted by the first
  // element, only to keep them small), then the sum of squares about that mean (with the usual first-order correction)
  const float pivot = x[base];
  float4 v[MAXQ];
  float s1 = 0.f;
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) {
    const int r = wave + q * NW;
    const int cl = r / RPC, i = (r % RPC) * 64 + lane;
    const bool ok = r < rows && i < HWV;
    v[q] = ok ? xg[(long long)cl * HWV + i] : make_float4(pivot, pivot, pivot, pivot);
    s1 += ((v[q].x - pivot) + (v[q].y - pivot)) + ((v[q].z - pivot) + (v[q].w - pivot));
  }
  const float L = (float)(cpg * HW);
  const float mean = pivot + block_sum(s1, red) / L;
  float s2 = 0.f, s1c = 0.f;
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) {
    const int r = wave + q * NW;
    const bool ok = r < rows && (r % RPC) * 64 + lane < HWV;
    const float a = v[q].x - mean, b = v[q].y - mean, c = v[q].z - mean, d = v[q].w - mean;
    s1c += ok ? (a + b) + (c + d) : 0.f;
    s2 += ok ? (a * a + b * b) + (c * c + d * d) : 0.f;
  }
  s1c = block_sum(s1c, red) / L;
  s2 = block_sum(s2, red) / L;
  const float rstd = rsqrtf(fmaxf(s2 - s1c * s1c, 0.f) + eps);
  if (tid == 0) {
    stats[2 * blockIdx.x] = mean;
    stats[2 * blockIdx.x + 1] = rstd;
  }
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) {
    const int r = wave + q * NW;
    const int cl = r / RPC, i = (r % RPC) * 64 + lane;
    const bool ok = r < rows && i < HWV;
    const int c = g * cpg + (r < rows ? cl : 0);
    float ga, be;
    gn_affine(gamma[c], beta[c], mean, rstd, ga, be);
    float4 o;
    o.x = gn_silu_value(v[q].x, ga, be);
    o.y = gn_silu_value(v[q].y, ga, be);
    o.z = gn_silu_value(v[q].z, ga, be);
    o.w = gn_silu_value(v[q].w, ga, be);
    if (ok) yg[(long long)cl * HWV + i] = o;
    if (pooled) {
      float ps = ok ? (o.x + o.y) + (o.z + o.w) : 0.f;
      ps = wave_sum(ps);
      if (lane == 0 && r < rows) atomicAdd(&psum[cl], ps);
    }
  }
  if (pooled) {
    __syncthreads();
    if (tid < cpg) pooled[(long long)n * C + g * cpg + tid] = psum[tid] / (float)HW;
  }
}

// NT threads: 256, or 512 for the largest groups -- the same group in half as many registers per thread, so that two
// 8-wave workgroups (4 waves per SIMD instead of 2) hide the latency of the transcendental-heavy middle phase.
template <int MODE, int MAXQ, int NT = GN_THREADS>
__global__ __launch_bounds__(NT, (MAXQ > 7 && MODE == 0) ? 3 : 1) void gn_silu_bwd_reg_kernel(const float* __restrict__ x,
                                                                       const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta,
                                                                       const float* __restrict__ stats,
                                                                       const float* __restrict__ dA, long long st_dA,
                                                                       GateBwd gb, float* __restrict__ dx,
                                                                       float* __restrict__ dgamma,
                                                                       float* __restrict__ dbeta, int C, int HW, int G) {
  __shared__ float csd[64], csdx[64], tot[2];
  if (MODE == 1 && gb.se.dsig) {     // side duty (saves a launch): 32 SE weights per chunk, chunks strided over the grid
    __shared__ float part[8][33];
    for (int ch = blockIdx.x; ch * 32 < 2 * gb.se.C * gb.se.Cr; ch += gridDim.x) se_wgrad_chunk(gb.se, ch, part);
  }
  const int n = blockIdx.x / G, g = blockIdx.x % G;
  const int cpg = C / G;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = NT / 64;
  const int HWV = HW / 4, RPC = (HWV + 63) / 64, rows = cpg * RPC;
  const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
  const float inv_hw = 1.f / (float)HW, inv_c = 1.f / (float)C;
  if (tid < 64) { csd[tid] = 0.f; csdx[tid] = 0.f; }
  __syncthreads();
  float4 xh[MAXQ], du[MAXQ];
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) {
    const int r = wave + q * NW;
    const int cl = r < rows ? r / RPC : 0, i = (r % RPC) * 64 + lane;
    const bool ok = r < rows && i < HWV;
    const int c = g * cpg + cl;
    const long long nc = (long long)n * C + c;
    const long long e = nc * HWV + (ok ? i : 0);               // quad index inside [N,C,HW]
    const float ga = gamma[c], be = beta[c];
    const float4 xv = reinterpret_cast<const float4*>(x)[e];
    float4 up, uf = make_float4(0.f, 0.f, 0.f, 0.f), sf = uf;
    if (MODE == 0) {
      up = reinterpret_cast<const float4*>(dA + (long long)n * st_dA + (long long)c * HW)[ok ? i : 0];
    } else {
      const long long m0 = (long long)n * HWV + (ok ? i : 0);
      const float4 dov = reinterpret_cast<const float4*>(gb.dout)[e];
      float gaf, bef;                                  // forward affine of this channel (recomputation of a2)
      gn_affine(ga, be, mean, rstd, gaf, bef);
      const float4 a2v = make_float4(gn_silu_parts(xv.x, gaf, bef, uf.x, sf.x), gn_silu_parts(xv.y, gaf, bef, uf.y, sf.y),
                                     gn_silu_parts(xv.z, gaf, bef, uf.z, sf.z), gn_silu_parts(xv.w, gaf, bef, uf.w, sf.w));
      const float4 gtv = reinterpret_cast<const float4*>(gb.gate)[m0], ctv = reinterpret_cast<const float4*>(gb.cnt)[m0];
      const float4 dav = reinterpret_cast<const float4*>(gb.dmap)[2 * (long long)n * HWV + (ok ? i : 0)];
      const float4 dmv = reinterpret_cast<const float4*>(gb.dmap)[(2 * (long long)n + 1) * HWV + (ok ? i : 0)];
      const float4 mxv = reinterpret_cast<const float4*>(gb.umax)[m0];
      const float sc = gb.s[nc], dpl = gb.dpool[nc] * inv_hw;
      auto one = [&](float a2e, float doe, float gte, float dae, float dme, float mxe, float cte) {
        const float U = a2e * sc;                      // bit-exact forward product: operand of the tie test
        float dU = doe * gte + dae * inv_c;
        if (U == mxe) dU += dme / cte;
        return dU * sc + dpl;
      };
      up.x = one(a2v.x, dov.x, gtv.x, dav.x, dmv.x, mxv.x, ctv.x);
      up.y = one(a2v.y, dov.y, gtv.y, dav.y, dmv.y, mxv.y, ctv.y);
      up.z = one(a2v.z, dov.z, gtv.z, dav.z, dmv.z, mxv.z, ctv.z);
      up.w = one(a2v.w, dov.w, gtv.w, dav.w, dmv.w, mxv.w, ctv.w);
    }
    float sd = 0.f, sdx = 0.f;
    auto elem = [&](float xe, float upe, float ufe, float sfe, float& xho, float& duo) {
      const float h = (xe - mean) * rstd;
      float u, sg;
      if (MODE == 1) {             // reuse the forward pre-activation and its sigmoid (one exp per element)
        u = ufe;
        sg = sfe;
      } else {
        u = h * ga + be;
        sg = fast_sigmoid(u);
      }
      const float d = ok ? upe * (sg * (1.f + u * (1.f - sg))) : 0.f;
      xho = h; duo = d; sd += d; sdx += d * h;
    };
    elem(xv.x, up.x, uf.x, sf.x, xh[q].x, du[q].x);
    elem(xv.y, up.y, uf.y, sf.y, xh[q].y, du[q].y);
    elem(xv.z, up.z, uf.z, sf.z, xh[q].z, du[q].z);
    elem(xv.w, up.w, uf.w, sf.w, xh[q].w, du[q].w);
    sd = wave_sum(sd);
    sdx = wave_sum(sdx);
    if (lane == 0 && r < rows) {
      atomicAdd(&csd[cl], sd);
      atomicAdd(&csdx[cl], sdx);
    }
  }
  __syncthreads();
  if (tid < cpg) {
    const int c = g * cpg + tid;
    unsafeAtomicAdd(dbeta + c, csd[tid]);
    unsafeAtomicAdd(dgamma + c, csdx[tid]);
  }
  if (tid == 0) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < cpg; ++k) {
      const float ga = gamma[g * cpg + k];
      a += csd[k] * ga;
      b += csdx[k] * ga;
    }
    const float m = 1.f / (float)(cpg * HW);
    tot[0] = a * m;
    tot[1] = b * m;
  }
  __syncthreads();
  const float s1 = tot[0], s2 = tot[1];
#pragma unroll
  for (int q = 0; q < MAXQ; ++q) {
    const int r = wave + q * NW;
    const int cl = r < rows ? r / RPC : 0, i = (r % RPC) * 64 + lane;
    const bool ok = r < rows && i < HWV;
    const int c = g * cpg + cl;
    const float ga = gamma[c];
    float4 o;
    o.x = rstd * (du[q].x * ga - s1 - xh[q].x * s2);
    o.y = rstd * (du[q].y * ga - s1 - xh[q].y * s2);
    o.z = rstd * (du[q].z * ga - s1 - xh[q].z * s2);
    o.w = rstd * (du[q].w * ga - s1 - xh[q].w * s2);
    if (ok) reinterpret_cast<float4*>(dx)[((long long)n * C + c) * HWV + i] = o;
  }
}

// number of register slots (64-quad rows per wave) a (cpg, hw) group needs with the float4 register path; 0 = n/a
static inline int gn_reg_slots(int cpg, int hw) {
  if (hw % 4 || cpg > 64) return 0;
  const int rpc = (hw / 4 + 63) / 64;
  return (cpg * rpc + GN_THREADS / 64 - 1) / (GN_THREADS / 64);
}

}  // namespace

// largest group (in 1024-element register slots) the register-resident backward takes; CM_GN_BWD_REG overrides (0: never)
static int gn_bwd_reg_max() {
  static const int v = getenv("CM_GN_BWD_REG") ? atoi(getenv("CM_GN_BWD_REG")) : 14;
  return v;
}

// 512-thread form of the register-resident backward for groups of 8..14 slots (CM_GN_BWD_WIDE=0: the 256-thread form)
static bool gn_bwd_wide() {
  static const bool v = !(getenv("CM_GN_BWD_WIDE") && atoi(getenv("CM_GN_BWD_WIDE")) == 0);
  return v;
}

extern "C" {

int cm_gn_silu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, float* pooled,
                   int n, int c, int hw, int groups, float eps, cm_stream stream) {
  if (n <= 0 || c <= 0 || hw <= 0 || groups <= 0 || c % groups) return -22;
  const bool vec = (hw % 4) == 0;
  const bool narrow = (vec ? hw / 4 : hw) <= 64 && (c / groups) >= 8;   // short rows: 16 lanes per channel
  hipStream_t st = (hipStream_t)stream;
  const int slots = gn_reg_slots(c / groups, hw);
  static const bool reg_fwd = getenv("CM_GN_FWD_REG") != nullptr;   // measured slower than the streaming kernel
  if (reg_fwd && slots > 0 && slots <= 14 && !narrow) {
    if (slots <= 7)
      gn_silu_fwd_reg_kernel<7><<<n * groups, GN_THREADS, 0, st>>>(x, gamma, beta, y, stats, pooled, c, hw, groups, eps);
    else
      gn_silu_fwd_reg_kernel<7, 512><<<n * groups, 512, 0, st>>>(x, gamma, beta, y, stats, pooled, c, hw, groups, eps);
    CM_CHECK_LAUNCH();
    return 0;
  }
#define GN_FWD(V, L) gn_silu_fwd_kernel<V, L, false><<<n * groups, GN_THREADS, 0, st>>>(x, gamma, beta, y, stats, pooled, c, hw, groups, eps, nullptr, 0, 0, nullptr)
  if (vec) { if (narrow) GN_FWD(true, 16); else GN_FWD(true, 64); }
  else     { if (narrow) GN_FWD(false, 16); else GN_FWD(false, 64); }
#undef GN_FWD
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_gn_silu_fwd_stats(const float* x, const float* gpart, int gslots, const float* gamma, const float* beta, float* y,
                         float* stats, float* pooled, int n, int c, int hw, int groups, float eps, cm_stream stream) {
  if (n <= 0 || c <= 0 || hw <= 0 || groups <= 0 || c % groups || !gpart || gslots <= 0) return -22;
  const bool vec = (hw % 4) == 0;
  const bool narrow = (vec ? hw / 4 : hw) <= 64 && (c / groups) >= 8;
  hipStream_t st = (hipStream_t)stream;
#define GN_FWD(V, L) gn_silu_fwd_kernel<V, L, false, true><<<n * groups, GN_THREADS, 0, st>>>(x, gamma, beta, y, stats, pooled, c, hw, groups, eps, nullptr, 0, 0, nullptr, gpart, gslots)
  if (vec) { if (narrow) GN_FWD(true, 16); else GN_FWD(true, 64); }
  else     { if (narrow) GN_FWD(false, 16); else GN_FWD(false, 64); }
#undef GN_FWD
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_gn_silu_fwd_parts(const float* parts, long long zs, int nparts, float* xsum, const float* gamma,
                         const float* beta, float* y, float* stats, float* pooled, int n, int c, int hw, int groups,
                         float eps, cm_stream stream) {
  if (n <= 0 || c <= 0 || hw <= 0 || groups <= 0 || c % groups || nparts < 1 || !parts || !xsum) return -22;
  const bool vec = (hw % 4) == 0 && (zs % 4) == 0;
  const bool narrow = (vec ? hw / 4 : hw) <= 64 && (c / groups) >= 8;
  hipStream_t st = (hipStream_t)stream;
#define GN_FWD(V, L) gn_silu_fwd_kernel<V, L, true><<<n * groups, GN_THREADS, 0, st>>>(nullptr, gamma, beta, y, stats, pooled, c, hw, groups, eps, parts, zs, nparts, xsum)
  if (vec) { if (narrow) GN_FWD(true, 16); else GN_FWD(true, 64); }
  else     { if (narrow) GN_FWD(false, 16); else GN_FWD(false, 64); }
#undef GN_FWD
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_gn_silu_bwd(const float* x, const float* gamma, const float* beta, const float* stats, const float* dA,
                   long long st_dA, float* dx, float* dgamma, float* dbeta, int n, int c, int hw, int groups,
                   cm_stream stream) {
  if (n <= 0 || c <= 0 || hw <= 0 || groups <= 0 || c % groups) return -22;
  GateBwd gb = {};
  const bool vec = (hw % 4) == 0 && (st_dA % 4) == 0;
  const bool narrow = (vec ? hw / 4 : hw) <= 64 && (c / groups) >= 8;
  hipStream_t st = (hipStream_t)stream;
  const int slots = gn_reg_slots(c / groups, hw);
  if (vec && slots > 0 && slots <= gn_bwd_reg_max() && !narrow) {
    if (slots <= 7)
      gn_silu_bwd_reg_kernel<0, 7><<<n * groups, GN_THREADS, 0, st>>>(x, gamma, beta, stats, dA, st_dA, gb, dx, dgamma, dbeta, c, hw, groups);
    else if (gn_bwd_wide())
      gn_silu_bwd_reg_kernel<0, 7, 512><<<n * groups, 512, 0, st>>>(x, gamma, beta, stats, dA, st_dA, gb, dx, dgamma, dbeta, c, hw, groups);
    else
      gn_silu_bwd_reg_kernel<0, 14><<<n * groups, GN_THREADS, 0, st>>>(x, gamma, beta, stats, dA, st_dA, gb, dx, dgamma, dbeta, c, hw, groups);
    CM_CHECK_LAUNCH();
    return 0;
  }
#define GN_BWD(V, L) gn_silu_bwd_kernel<0, V, L><<<n * groups, GN_THREADS, 0, st>>>(x, gamma, beta, stats, dA, st_dA, gb, dx, dgamma, dbeta, c, hw, groups)
  if (vec) { if (narrow) GN_BWD(4, 16); else GN_BWD(4, 64); }
  else     { if (narrow) GN_BWD(1, 16); else GN_BWD(1, 64); }
#undef GN_BWD
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_gn_silu_apply(const float* x, const float* gamma, const float* beta, const float* stats, float* y, int n, int c,
                     int hw, int groups, cm_stream stream) {
  if (n <= 0 || c <= 0 || hw <= 0 || groups <= 0 || c % groups) return -22;
  const long long total = (long long)n * c * hw;
  const long long blocks = (total + 255) / 256;
  gn_silu_apply_kernel<<<(int)(blocks > 4096 ? 4096 : blocks), 256, 0, (hipStream_t)stream>>>(x, gamma, beta, stats, y, c,
                                                                                              hw, groups, total);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_gn_silu_bwd_gated(const float* x, const float* gamma, const float* beta, const float* stats,
                         const float* a2, const float* dout, const float* gate, const float* dmap, const float* umax,
                         const float* cnt, const float* s, const float* dpool, float* dx, float* dgamma,
                         float* dbeta, int n, int c, int hw, int groups, const float* se_dsig, const float* se_dz,
                         const float* se_z, const float* se_pooled, float* se_dw1, float* se_dw2, int se_cr,
                         cm_stream stream) {
  if (n <= 0 || c <= 0 || hw <= 0 || groups <= 0 || c % groups) return -22;
  if (se_dsig && (!se_dz || !se_z || !se_pooled || !se_dw1 || !se_dw2 || se_cr <= 0)) return -22;
  GateBwd gb;
  gb.a2 = a2; gb.dout = dout; gb.gate = gate; gb.dmap = dmap; gb.umax = umax; gb.cnt = cnt; gb.s = s; gb.dpool = dpool;
  gb.se.dsig = se_dsig; gb.se.dz = se_dz; gb.se.z = se_z; gb.se.pooled = se_pooled; gb.se.dw1 = se_dw1; gb.se.dw2 = se_dw2;
  gb.se.N = n; gb.se.C = c; gb.se.Cr = se_cr;
  const bool vec = (hw % 4) == 0;
  const bool narrow = (vec ? hw / 4 : hw) <= 64 && (c / groups) >= 8;
  hipStream_t st = (hipStream_t)stream;
  const int slots = gn_reg_slots(c / groups, hw);
  if (vec && slots > 0 && slots <= gn_bwd_reg_max() && !narrow) {
    if (slots <= 7)
      gn_silu_bwd_reg_kernel<1, 7><<<n * groups, GN_THREADS, 0, st>>>(x, gamma, beta, stats, nullptr, 0, gb, dx, dgamma, dbeta, c, hw, groups);
    else if (gn_bwd_wide())
      gn_silu_bwd_reg_kernel<1, 7, 512><<<n * groups, 512, 0, st>>>(x, gamma, beta, stats, nullptr, 0, gb, dx, dgamma, dbeta, c, hw, groups);
    else
      gn_silu_bwd_reg_kernel<1, 14><<<n * groups, GN_THREADS, 0, st>>>(x, gamma, beta, stats, nullptr, 0, gb, dx, dgamma, dbeta, c, hw, groups);
    CM_CHECK_LAUNCH();
    return 0;
  }
#define GN_BWD(V, L) gn_silu_bwd_kernel<1, V, L><<<n * groups, GN_THREADS, 0, st>>>(x, gamma, beta, stats, nullptr, 0, gb, dx, dgamma, dbeta, c, hw, groups)
  if (vec) { if (narrow) GN_BWD(4, 16); else GN_BWD(4, 64); }
  else     { if (narrow) GN_BWD(1, 16); else GN_BWD(1, 64); }
#undef GN_BWD
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
