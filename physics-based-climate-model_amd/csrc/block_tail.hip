// Sample-resident ConvBlock tail: everything of ConvBlock.forward after its second convolution (reference
// src/unet.py:39,45-47: GroupNorm(8) -> SiLU -> SEBlock -> SpatialGate, plus the encoder's MaxPool2d(2) of the result,
// src/unet_convlstm_attention.py:21,25) as ONE launch, and the matching reductions of the backward as one launch.
//
// Every step of that tail is per-sample: GroupNorm statistics per (sample, group), the SE squeeze per (sample, channel),
// the CBAM channel mean / max per (sample, pixel), a 7x7 convolution over the sample's two maps.  One workgroup of up to
// 1024 threads owns one sample and keeps the conv output y2 [C, H*W] in REGISTERS (<= 64 floats per thread: the H/2, H/4
// and H/8 levels at base 32), so y2 is read from HBM once and `out` written once, where the three-launch form (cm_gn_silu_fwd,
// cm_se_spatial_stats, cm_spatial_apply) reads y2 twice, writes the activation a2 and reads it back twice.  a2 itself is
// never stored: the backward recomputes it bit-exactly from y2 and the stored statistics (gn_silu_value, as the gated
// GroupNorm backward already does).
//
// Thread layout: the sample is cut into CS channel slices of CPT channels; slice k is worked by LPS lanes (a power of two
// >= the number of pixel units), lane l owning VEC pixels of every channel of the slice:
//   BLK2 (H, W even): the unit is a 2x2 pixel block (VEC = 4) -- MaxPool2d(2) is then a maximum inside the thread;
//   linear (VEC = 2): units are pixel pairs in row-major order (the 6x9 level, where W is odd and nothing is pooled).
// Reductions: over pixels (GroupNorm sums, SE squeeze) = fixed-order DPP sums inside 32-lane halves + LDS partials;
// over channels (CBAM maps) = in-thread over the slice's channels, then across slices through LDS in slice order.
// Everything is deterministic (no atomics in the forward).
#include <stdlib.h>
#include "common.h"
#include "../../include/climate_hip.h"

namespace {

// rounding-pinned forward value, identical to norm_act.hip (the amax tie test needs bit-equal recomputation)
__device__ __forceinline__ void bt_affine(float gamma, float beta, float mean, float rstd, float& ga, float& be) {
  ga = __fmul_rn(gamma, rstd);
  be = __fsub_rn(beta, __fmul_rn(__fmul_rn(mean, rstd), gamma));
}
__device__ __forceinline__ float bt_silu_parts(float v, float ga, float be, float& u, float& sg) {
  u = __fmaf_rn(v, ga, be);
  sg = __frcp_rn(__fadd_rn(1.f, __expf(-u)));
  return __fmul_rn(u, sg);
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ float bt_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWMASK, 0xf, true));
}
// sums over the two 32-lane halves of the wave, fixed order: lo = lanes 0-31, hi = lanes 32-63 (both wave-uniform)
__device__ __forceinline__ void half_sums(float v, float& lo, float& hi) {
  v += bt_dpp<0x111, 0xf>(v);   // row_shr:1
  v += bt_dpp<0x112, 0xf>(v);   // row_shr:2
  v += bt_dpp<0x114, 0xf>(v);   // row_shr:4
  v += bt_dpp<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of every 16-lane row holds the row's sum
  v += bt_dpp<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3 -> lanes 31 / 63 hold the halves' sums
  lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
  hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

struct TailFwd {
  const float* y2;      // [N, C, HW] conv output (NULL with parts)
  const float* parts;   // [nparts][N, C, HW] partial slices of it (cm_conv3x3_h3 bit 29), zs apart; summed into ysum
  long long zs;
  int nparts;
  float* ysum;          // [N, C, HW] (parts only)
  const float *gamma, *beta, *w1, *w2, *w7;
  float *stats, *pooled, *z, *s, *fmap, *gate, *out, *mp;
  int C, Cr, H, W, LPS, CS, units;
  float eps;
};

// pixel offsets (inside one channel plane) of the VEC pixels of unit l
template <bool BLK2>
__device__ __forceinline__ void unit_pixels(int l, int W, int (&px)[BLK2 ? 4 : 2]) {
  if constexpr (BLK2) {
    const int wh = W >> 1, yy = l / wh, xx = l - yy * wh;
    px[0] = 2 * yy * W + 2 * xx;
    px[1] = px[0] + 1;
    px[2] = px[0] + W;
    px[3] = px[2] + 1;
  } else {
    px[0] = 2 * l;
    px[1] = 2 * l + 1;
  }
}

// per-channel sums over the slice's pixels: v[j] = this thread's partial for channel j of its slice; the LPS lanes'
// partials are added in a fixed order (32-lane halves by DPP, halves through `red`); afterwards red[hw][j] holds the
// partial of half-wave hw.  Caller syncs.
template <int CPT, typename F>
__device__ __forceinline__ void post_halves(F&& value_of, float* red, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    float lo, hi;
    half_sums(value_of(j), lo, hi);
    if (lane == 0) {
      red[(wave * 2) * CPT + j] = lo;
      red[(wave * 2 + 1) * CPT + j] = hi;
    }
  }
}
// total of channel c (global index) from the half-wave partials: the channel's slice spans LPS/32 half-waves
template <int CPT>
__device__ __forceinline__ float channel_total(const float* red, int c, int LPS) {
  const int slice = c / CPT, j = c - slice * CPT, nh = LPS >> 5;
  float t = 0.f;
  for (int h = 0; h < nh; ++h) t += red[(slice * nh + h) * CPT + j];
  return t;
}

template <int CPT, bool BLK2>
__global__ __launch_bounds__(1024) void block_tail_fwd_kernel(TailFwd a) {
  constexpr int VEC = BLK2 ? 4 : 2;
  extern __shared__ float sh[];
  const int C = a.C, Cr = a.Cr, H = a.H, W = a.W, HW = H * W, LPS = a.LPS, CS = a.CS;
  const int tid = threadIdx.x, n = blockIdx.x;
  const int slice = tid / LPS, l = tid - slice * LPS;
  const bool live = l < a.units;
  const int cpg = C / 8;
  const int PW = W + 6;
  // ---- LDS carve-up ----
  float* red = sh;                              // [32 half-waves][CPT]
  float* gst = red + 32 * CPT;                  // [8][2] mean, rstd
  float* csh = gst + 16;                        // [C] pooled, then s
  float* hsh = csh + C;                         // [Cr]
  float* psum = hsh + ((Cr + 3) & ~3);          // [CS][LPS*VEC]
  float* pmax = psum + CS * LPS * VEC;          // [CS][LPS*VEC]
  float* msh = pmax + CS * LPS * VEC;           // [2][(H+6)][PW] zero-padded maps
  float* gsh = msh + 2 * (H + 6) * PW;          // [HW] gate
  __shared__ float wsh[98];
  for (int i = tid; i < 98; i += blockDim.x) wsh[i] = a.w7[i];
  for (int i = tid; i < 2 * (H + 6) * PW; i += blockDim.x) msh[i] = 0.f;

  int px[VEC];
  unit_pixels<BLK2>(live ? l : 0, W, px);
  const int c0 = slice * CPT;
  const long long sbase = (long long)n * C * HW;     // (uniform; per-lane offsets stay 32-bit: saddr + voffset loads)
  const int ch0 = c0 * HW;

  // ---- load the sample (sum of the partial slices, written back as the conv output the backward reads) ----
  float v[CPT][VEC];
  if (a.parts) {
    const float* pb = a.parts + sbase;
    float* yb = a.ysum + sbase;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int o = ch0 + j * HW;
      float2 r0 = *reinterpret_cast<const float2*>(pb + o + px[0]);
      float2 r1 = BLK2 ? *reinterpret_cast<const float2*>(pb + o + px[BLK2 ? 2 : 0]) : make_float2(0.f, 0.f);
      for (int zz = 1; zz < a.nparts; ++zz) {
        const float* pz = pb + zz * a.zs;
        const float2 q0 = *reinterpret_cast<const float2*>(pz + o + px[0]);
        r0.x += q0.x; r0.y += q0.y;
        if constexpr (BLK2) {
          const float2 q1 = *reinterpret_cast<const float2*>(pz + o + px[2]);
          r1.x += q1.x; r1.y += q1.y;
        }
      }
      v[j][0] = r0.x; v[j][1] = r0.y;
      if constexpr (BLK2) { v[j][2] = r1.x; v[j][3] = r1.y; }
      if (live) {
        *reinterpret_cast<float2*>(yb + o + px[0]) = r0;
        if constexpr (BLK2) *reinterpret_cast<float2*>(yb + o + px[2]) = r1;
      }
    }
  } else {
    const float* yb = a.y2 + sbase;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int o = ch0 + j * HW;
      const float2 r0 = *reinterpret_cast<const float2*>(yb + o + px[0]);
      v[j][0] = r0.x; v[j][1] = r0.y;
      if constexpr (BLK2) {
        const float2 r1 = *reinterpret_cast<const float2*>(yb + o + px[2]);
        v[j][2] = r1.x; v[j][3] = r1.y;
      }
    }
  }

  // ---- GroupNorm statistics, true two-pass on the register-resident sample ----
  post_halves<CPT>([&](int j) {
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) q += v[j][e];
    return live ? q : 0.f;
  }, red, tid);
  __syncthreads();
  if (tid < 8) {
    float t = 0.f;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) t += channel_total<CPT>(red, c, LPS);
    gst[2 * tid] = t / (float)(cpg * HW);
  }
  __syncthreads();
  post_halves<CPT>([&](int j) {
    const float m = gst[2 * ((c0 + j) / cpg)];
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float d = v[j][e] - m;
      q += d * d;
    }
    return live ? q : 0.f;
  }, red, tid);
  __syncthreads();
  if (tid < 8) {
    float t = 0.f;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) t += channel_total<CPT>(red, c, LPS);
    const float rstd = rsqrtf(t / (float)(cpg * HW) + a.eps);
    gst[2 * tid + 1] = rstd;
    a.stats[((long long)n * 8 + tid) * 2] = gst[2 * tid];
    a.stats[((long long)n * 8 + tid) * 2 + 1] = rstd;
  }
  __syncthreads();

  // ---- a2 = SiLU(GroupNorm(y2)) in place; SE squeeze ----
  post_halves<CPT>([&](int j) {
    const int c = c0 + j, g = c / cpg;
    float ga, be;
    bt_affine(a.gamma[c], a.beta[c], gst[2 * g], gst[2 * g + 1], ga, be);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float u, sg;
      v[j][e] = bt_silu_parts(v[j][e], ga, be, u, sg);
      q += v[j][e];
    }
    return live ? q : 0.f;
  }, red, tid);
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    const float p = channel_total<CPT>(red, c, LPS) / (float)HW;
    csh[c] = p;
    a.pooled[(long long)n * C + c] = p;
  }
  __syncthreads();
  // ---- SE excite: z = W1 p, s = sigmoid(W2 relu(z)) ----
  {
    const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    for (int r = wave; r < Cr; r += nw) {
      float acc = 0.f;
      for (int c = lane; c < C; c += 64) acc += a.w1[(long long)r * C + c] * csh[c];
      acc = wave_sum(acc);
      if (lane == 0) {
        a.z[(long long)n * Cr + r] = acc;
        hsh[r] = fmaxf(acc, 0.f);
      }
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    float acc = 0.f;
    for (int r = 0; r < Cr; ++r) acc += a.w2[(long long)c * Cr + r] * hsh[r];
    const float sv = sigmoid_acc(acc);
    csh[c] = sv;                                   // (pooled[c] is dead: only this thread touches entry c here)
    a.s[(long long)n * C + c] = sv;
  }
  __syncthreads();

  // ---- U = a2 * s; channel mean / max per pixel ----
  {
    float su[VEC], mx[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { su[e] = 0.f; mx[e] = -INFINITY; }
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const float sc = csh[c0 + j];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        v[j][e] = __fmul_rn(v[j][e], sc);
        su[e] += v[j][e];
        mx[e] = fmaxf(mx[e], v[j][e]);
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      psum[(slice * LPS + l) * VEC + e] = su[e];
      pmax[(slice * LPS + l) * VEC + e] = mx[e];
    }
  }
  __syncthreads();
  for (int i = tid; i < a.units * VEC; i += blockDim.x) {
    float t = 0.f, m = -INFINITY;
    for (int k = 0; k < CS; ++k) {
      t += psum[k * LPS * VEC + i];
      m = fmaxf(m, pmax[k * LPS * VEC + i]);
    }
    int q[VEC];
    unit_pixels<BLK2>(i / VEC, W, q);
    const int p = q[i % VEC];
    const float avg = t / (float)C;
    a.fmap[((long long)n * 2) * HW + p] = avg;
    a.fmap[((long long)n * 2 + 1) * HW + p] = m;
    const int y = p / W, x = p - y * W;
    msh[(y + 3) * PW + x + 3] = avg;
    msh[(H + 6) * PW + (y + 3) * PW + x + 3] = m;
  }
  __syncthreads();
  // ---- gate = sigmoid(conv7x7([avg, max])), accumulation order ch, dy, dx (as cm_spatial_apply) ----
  for (int p = tid; p < HW; p += blockDim.x) {
    const int y = p / W, x = p - y * W;
    float acc = 0.f;
#pragma unroll 1
    for (int chdy = 0; chdy < 14; ++chdy) {        // (not unrolled: 98 hoisted LDS reads would cost 98 registers)
      const int ch = chdy / 7, dy = chdy - ch * 7;
      const float* mr = msh + ch * (H + 6) * PW + (y + dy) * PW + x;
#pragma unroll
      for (int dx = 0; dx < 7; ++dx) acc += wsh[chdy * 7 + dx] * mr[dx];
    }
    const float g = sigmoid_acc(acc);
    gsh[p] = g;
    a.gate[(long long)n * HW + p] = g;
  }
  __syncthreads();
  // ---- out = U * gate (+ MaxPool2d(2)) ----
  if (live) {
    float g[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) g[e] = gsh[px[e]];
    float* ob = a.out + sbase;
    float* mpb = a.mp ? a.mp + (long long)n * C * (HW >> 2) : nullptr;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int o = ch0 + j * HW;
      float ov[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) ov[e] = __fmul_rn(v[j][e], g[e]);
      *reinterpret_cast<float2*>(ob + o + px[0]) = make_float2(ov[0], ov[1]);
      if constexpr (BLK2) {
        *reinterpret_cast<float2*>(ob + o + px[2]) = make_float2(ov[2], ov[3]);
        if (mpb) mpb[(c0 + j) * (HW >> 2) + l] = fmaxf(fmaxf(ov[0], ov[1]), fmaxf(ov[2], ov[3]));
      }
    }
  }
}

// ================================================================================================ backward
// The reductions of the tail's backward that need the whole sample -- everything between d(out) and the GroupNorm
// backward -- in one launch (the four-launch form: cm_gate_bwd_reduce, cm_conv7_bwd, cm_se_bwd_reduce, cm_se_excite_bwd):
//   U = a2*s (a2 recomputed from y2 and the stored statistics), dgate = sum_c dout*U, (umax, cnt) = channel maximum and
//   its multiplicity, dgpre = dgate*g*(1-g), dmap = conv7^T(dgpre), dW7 += map * dgpre, dU = dout*g + dmapA/C +
//   [U == umax] dmapM/cnt, ds_c = sum_p dU*a2, then the SE excite backward (dsig, dz, dpool).
// Outputs are exactly the maps cm_gn_silu_bwd_gated consumes: dmap [N,2,HW], umax, cnt [N,HW], dpool [N,C], dsig [N,C],
// dz [N,Cr].  a2 stays in registers between the two passes; dout is read twice (second time from L2).
struct TailBwd {
  const float *y2, *stats, *gamma, *beta, *s, *z, *gate, *fmap, *w1, *w2, *w7, *dout;
  float *dmap, *umax, *cnt, *dpool, *dsig, *dz, *dw7;
  int C, Cr, H, W, LPS, CS, units, part;   // part: floats per partial array (>= CS*LPS*VEC, 3*part >= 8*98)
};

template <int CPT, bool BLK2>
__global__ __launch_bounds__(1024) void block_tail_bwd_kernel(TailBwd a) {
  constexpr int VEC = BLK2 ? 4 : 2;
  extern __shared__ float sh[];
  const int C = a.C, Cr = a.Cr, H = a.H, W = a.W, HW = H * W, LPS = a.LPS, CS = a.CS;
  const int tid = threadIdx.x, n = blockIdx.x;
  const int slice = tid / LPS, l = tid - slice * LPS;
  const bool live = l < a.units;
  const int cpg = C / 8;
  const int PW = W + 6, PL = (H + 6) * PW;
  float* red = sh;                              // [32][CPT]
  float* csh = red + 32 * CPT;                  // [C] s
  float* dsh = csh + C;                         // [C] ds -> dsig
  float* zsh = dsh + C;                         // [Cr] dz
  float* p0 = zsh + ((Cr + 3) & ~3);            // [CS][LPS*VEC] dgate partials
  float* p1 = p0 + a.part;                      // max partials
  float* p2 = p1 + a.part;                      // count partials
  float* dpre = p2 + a.part;                    // [(H+6)][PW] dgpre, zero padded
  float* msh = dpre + PL;                       // [2][(H+6)][PW] forward maps, zero padded
  float* dmA = msh + 2 * PL;                    // [HW] dmap avg / C
  float* dmM = dmA + HW;                        // [HW] dmap max / cnt
  float* umx = dmM + HW;                        // [HW]
  float* gsh = umx + HW;                        // [HW] gate
  __shared__ float wsh[98];
  for (int i = tid; i < 98; i += blockDim.x) wsh[i] = a.w7[i];
  for (int i = tid; i < 3 * PL; i += blockDim.x) dpre[i] = 0.f;
  for (int c = tid; c < C; c += blockDim.x) csh[c] = a.s[(long long)n * C + c];
  __syncthreads();
  for (int p = tid; p < HW; p += blockDim.x) {
    const int y = p / W, x = p - y * W;
    gsh[p] = a.gate[(long long)n * HW + p];
    msh[(y + 3) * PW + x + 3] = a.fmap[((long long)n * 2) * HW + p];
    msh[PL + (y + 3) * PW + x + 3] = a.fmap[((long long)n * 2 + 1) * HW + p];
  }

  int px[VEC];
  unit_pixels<BLK2>(live ? l : 0, W, px);
  const int c0 = slice * CPT;
  const long long sbase = (long long)n * C * HW;
  const int ch0 = c0 * HW;
  const float* yb = a.y2 + sbase;
  const float* db = a.dout + sbase;
  const float* stn = a.stats + (long long)n * 16;

  // ---- a2 (registers) from y2 and the stored statistics; pass A over (a2, dout) ----
  float v[CPT][VEC];
  float dg[VEC], mx[VEC], kk[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { dg[e] = 0.f; mx[e] = -INFINITY; kk[e] = 0.f; }
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const int c = c0 + j, g = c / cpg;
    const int o = ch0 + j * HW;
    float ga, be;
    bt_affine(a.gamma[c], a.beta[c], stn[2 * g], stn[2 * g + 1], ga, be);
    float xv[VEC], dv[VEC];
    {
      const float2 r0 = *reinterpret_cast<const float2*>(yb + o + px[0]);
      const float2 d0 = *reinterpret_cast<const float2*>(db + o + px[0]);
      xv[0] = r0.x; xv[1] = r0.y; dv[0] = d0.x; dv[1] = d0.y;
      if constexpr (BLK2) {
        const float2 r1 = *reinterpret_cast<const float2*>(yb + o + px[2]);
        const float2 d1 = *reinterpret_cast<const float2*>(db + o + px[2]);
        xv[2] = r1.x; xv[3] = r1.y; dv[2] = d1.x; dv[3] = d1.y;
      }
    }
    const float sc = csh[c];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float u_, sg_;
      v[j][e] = bt_silu_parts(xv[e], ga, be, u_, sg_);
      const float U = __fmul_rn(v[j][e], sc);
      dg[e] += dv[e] * U;
      const float same = (U == mx[e]) ? 1.f : 0.f;
      kk[e] = (U > mx[e]) ? 1.f : kk[e] + same;
      mx[e] = fmaxf(mx[e], U);
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    p0[(slice * LPS + l) * VEC + e] = dg[e];
    p1[(slice * LPS + l) * VEC + e] = mx[e];
    p2[(slice * LPS + l) * VEC + e] = kk[e];
  }
  __syncthreads();
  for (int i = tid; i < a.units * VEC; i += blockDim.x) {
    float t = 0.f, m = -INFINITY;
    for (int k = 0; k < CS; ++k) {
      t += p0[k * LPS * VEC + i];
      m = fmaxf(m, p1[k * LPS * VEC + i]);
    }
    float cn = 0.f;
    for (int k = 0; k < CS; ++k) cn += (p1[k * LPS * VEC + i] == m) ? p2[k * LPS * VEC + i] : 0.f;
    cn = fmaxf(cn, 1.f);
    int q[VEC];
    unit_pixels<BLK2>(i / VEC, W, q);
    const int p = q[i % VEC];
    const float g = gsh[p];
    const int y = p / W, x = p - y * W;
    dpre[(y + 3) * PW + x + 3] = t * g * (1.f - g);
    umx[p] = m;
    dmM[p] = cn;                                   // (count for now; turned into dmapM / cnt below)
    a.umax[(long long)n * HW + p] = m;
    a.cnt[(long long)n * HW + p] = cn;
  }
  __syncthreads();
  // ---- conv7 backward: dmap (flipped kernel) and dW7 ----
  for (int i = tid; i < 2 * HW; i += blockDim.x) {
    const int ch = i / HW, p = i - ch * HW;
    const int y = p / W, x = p - y * W;
    float acc = 0.f;
#pragma unroll 1
    for (int dy = 0; dy < 7; ++dy) {
      const float* dr = dpre + (y + 6 - dy) * PW + x;
#pragma unroll
      for (int dx = 0; dx < 7; ++dx) acc += wsh[ch * 49 + dy * 7 + dx] * dr[6 - dx];
    }
    a.dmap[((long long)n * 2 + ch) * HW + p] = acc;
    if (ch == 0) dmA[p] = acc / (float)C;
    else dmM[p] = acc / dmM[p];
  }
  __syncthreads();                                 // (dmA / dmM complete; the partial arrays are free)
  // dW7[tap] += sum_p map[ch][p + tap - 3] * dgpre[p]: 98 taps x 8 row shares, combined in a fixed order through p0
  for (int i = tid; i < 8 * 98; i += blockDim.x) {
    const int tap = i % 98, share = i / 98;
    const int ch = tap / 49, dy = (tap % 49) / 7, dx = tap % 7;
    float acc = 0.f;
    for (int y = share; y < H; y += 8) {
      const float* mr = msh + ch * PL + (y + dy) * PW + dx;
      const float* dr = dpre + (y + 3) * PW + 3;
      for (int x = 0; x < W; ++x) acc += mr[x] * dr[x];
    }
    p0[share * 98 + tap] = acc;
  }
  __syncthreads();
  for (int i = tid; i < 98; i += blockDim.x) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += p0[k * 98 + i];
    unsafeAtomicAdd(a.dw7 + i, t);
  }
  // ---- pass B: dU and ds_c = sum_p dU * a2 ----
  {
    float g[VEC], da[VEC], dm[VEC], um[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      g[e] = gsh[px[e]]; da[e] = dmA[px[e]]; dm[e] = dmM[px[e]]; um[e] = umx[px[e]];
    }
    post_halves<CPT>([&](int j) {
      const int o = ch0 + j * HW;
      float dv[VEC];
      const float2 d0 = *reinterpret_cast<const float2*>(db + o + px[0]);
      dv[0] = d0.x; dv[1] = d0.y;
      if constexpr (BLK2) {
        const float2 d1 = *reinterpret_cast<const float2*>(db + o + px[2]);
        dv[2] = d1.x; dv[3] = d1.y;
      }
      const float sc = csh[c0 + j];
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float U = __fmul_rn(v[j][e], sc);
        float dU = dv[e] * g[e] + da[e];
        if (U == um[e]) dU += dm[e];
        q += dU * v[j][e];
      }
      return live ? q : 0.f;
    }, red, tid);
  }
  __syncthreads();
  // ---- SE excite backward: dsig = ds*s*(1-s); dz = (W2^T dsig) [z>0]; dpool = W1^T dz ----
  for (int c = tid; c < C; c += blockDim.x) {
    const float sv = csh[c];
    const float d = channel_total<CPT>(red, c, LPS) * sv * (1.f - sv);
    dsh[c] = d;
    a.dsig[(long long)n * C + c] = d;
  }
  __syncthreads();
  {
    const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    for (int r = wave; r < Cr; r += nw) {
      float acc = 0.f;
      for (int c = lane; c < C; c += 64) acc += a.w2[(long long)c * Cr + r] * dsh[c];
      acc = wave_sum(acc);
      if (lane == 0) {
        const float d = a.z[(long long)n * Cr + r] > 0.f ? acc : 0.f;
        zsh[r] = d;
        a.dz[(long long)n * Cr + r] = d;
      }
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    float acc = 0.f;
    for (int r = 0; r < Cr; ++r) acc += a.w1[(long long)r * C + c] * zsh[r];
    a.dpool[(long long)n * C + c] = acc;
  }
}

// geometry of the sample-resident layout; returns false when the sample does not fit (the caller keeps the multi-launch form)
struct TailGeo {
  int cpt, lps, cs, units, threads;
  bool blk2;
};
static bool tail_geometry(int c, int h, int w, TailGeo* g) {
  if (c <= 0 || h <= 0 || w <= 0 || c % 8) return false;
  g->blk2 = (h % 2 == 0) && (w % 2 == 0);
  const int hw = h * w;
  if (!g->blk2 && hw % 2) return false;
  g->units = g->blk2 ? hw / 4 : hw / 2;
  int lps = 32;
  while (lps < g->units) lps *= 2;
  if (lps > 1024) return false;
  g->lps = lps;
  const int vec = g->blk2 ? 4 : 2;
  // fewest channels per thread that fit 1024 threads
  int cpt = 1;
  while (cpt <= 16 && (c % cpt != 0 || (c / cpt) * lps > 1024)) cpt *= 2;
  if (cpt > 16 || cpt * vec > 64) return false;
  const int cpg = c / 8;
  if (cpt % cpg != 0 && cpg % cpt != 0) return false;
  g->cpt = cpt;
  g->cs = c / cpt;
  g->threads = g->cs * lps;
  if (g->threads % 64 || g->threads < 64) return false;
  return true;
}

static size_t tail_fwd_lds(const TailGeo& g, int c, int cr, int h, int w) {
  const int vec = g.blk2 ? 4 : 2;
  return sizeof(float) * ((size_t)32 * g.cpt + 16 + c + ((cr + 3) & ~3) + 2 * (size_t)g.cs * g.lps * vec +
                          2 * (size_t)(h + 6) * (w + 6) + (size_t)h * w);
}
static int tail_bwd_part(const TailGeo& g) {
  const int part = g.cs * g.lps * (g.blk2 ? 4 : 2);
  return part * 3 < 8 * 98 ? (8 * 98 + 2) / 3 : part;
}
static size_t tail_bwd_lds(const TailGeo& g, int c, int cr, int h, int w) {
  const size_t part = (size_t)tail_bwd_part(g);
  return sizeof(float) * ((size_t)32 * g.cpt + 2 * c + ((cr + 3) & ~3) + 3 * part + 3 * (size_t)(h + 6) * (w + 6) +
                          4 * (size_t)h * w);
}

static void tail_attrs() {      // dynamic LDS above 64 KB needs the attribute (once per process)
  static bool done = false;
  if (done) return;
#define CM_TAIL_ATTR(K)                                                                                   \
  hipFuncSetAttribute((const void*)K<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);   \
  hipFuncSetAttribute((const void*)K<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);   \
  hipFuncSetAttribute((const void*)K<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);   \
  hipFuncSetAttribute((const void*)K<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);   \
  hipFuncSetAttribute((const void*)K<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);  \
  hipFuncSetAttribute((const void*)K<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);  \
  hipFuncSetAttribute((const void*)K<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);  \
  hipFuncSetAttribute((const void*)K<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);  \
  hipFuncSetAttribute((const void*)K<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);  \
  hipFuncSetAttribute((const void*)K<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
  CM_TAIL_ATTR(block_tail_fwd_kernel)
  CM_TAIL_ATTR(block_tail_bwd_kernel)
#undef CM_TAIL_ATTR
  (void)hipGetLastError();      // (an attribute the runtime refuses must not poison the launch check below)
  done = true;
}

}  // namespace

extern "C" {

int cm_block_tail_supported(int c, int cr, int h, int w) {
  TailGeo g;
  if (!tail_geometry(c, h, w, &g) || cr <= 0 || cr > 64) return 0;
  return tail_fwd_lds(g, c, cr, h, w) <= 150 * 1024 && tail_bwd_lds(g, c, cr, h, w) <= 150 * 1024;
}

#define CM_TAIL_DISPATCH(KERNEL, ARGS, LDS)                                                            \
  do {                                                                                                 \
    const dim3 grid(n), block(g.threads);                                                              \
    hipStream_t st = (hipStream_t)stream;                                                              \
    if (g.blk2) {                                                                                      \
      switch (g.cpt) {                                                                                 \
        case 1: KERNEL<1, true><<<grid, block, LDS, st>>>(ARGS); break;                                \
        case 2: KERNEL<2, true><<<grid, block, LDS, st>>>(ARGS); break;                                \
        case 4: KERNEL<4, true><<<grid, block, LDS, st>>>(ARGS); break;                                \
        case 8: KERNEL<8, true><<<grid, block, LDS, st>>>(ARGS); break;                                \
        default: KERNEL<16, true><<<grid, block, LDS, st>>>(ARGS); break;                              \
      }                                                                                                \
    } else {                                                                                           \
      switch (g.cpt) {                                                                                 \
        case 1: KERNEL<1, false><<<grid, block, LDS, st>>>(ARGS); break;                               \
        case 2: KERNEL<2, false><<<grid, block, LDS, st>>>(ARGS); break;                               \
        case 4: KERNEL<4, false><<<grid, block, LDS, st>>>(ARGS); break;                               \
        case 8: KERNEL<8, false><<<grid, block, LDS, st>>>(ARGS); break;                               \
        default: KERNEL<16, false><<<grid, block, LDS, st>>>(ARGS); break;                             \
      }                                                                                                \
    }                                                                                                  \
  } while (0)

int cm_block_tail_fwd(const float* y2, const float* parts, long long zs, int nparts, float* ysum, const float* gamma,
                      const float* beta, const float* w1, const float* w2, const float* w7, float* stats,
                      float* pooled, float* z, float* s, float* fmap, float* gate, float* out, float* mp, int n, int c,
                      int cr, int h, int w, float eps, cm_stream stream) {
  TailGeo g;
  if (n <= 0 || !cm_block_tail_supported(c, cr, h, w) || !tail_geometry(c, h, w, &g)) return -22;
  if ((y2 == nullptr) == (parts == nullptr)) return -22;
  if (parts && (nparts < 1 || !ysum || (zs & 1))) return -22;
  if (mp && !g.blk2) return -22;
  TailFwd a;
  a.y2 = y2; a.parts = parts; a.zs = zs; a.nparts = nparts; a.ysum = ysum;
  a.gamma = gamma; a.beta = beta; a.w1 = w1; a.w2 = w2; a.w7 = w7;
  a.stats = stats; a.pooled = pooled; a.z = z; a.s = s; a.fmap = fmap; a.gate = gate; a.out = out; a.mp = mp;
  a.C = c; a.Cr = cr; a.H = h; a.W = w; a.LPS = g.lps; a.CS = g.cs; a.units = g.units; a.eps = eps;
  const size_t lds = tail_fwd_lds(g, c, cr, h, w);
  tail_attrs();
  CM_TAIL_DISPATCH(block_tail_fwd_kernel, a, lds);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_block_tail_bwd(const float* y2, const float* stats, const float* gamma, const float* beta, const float* s,
                      const float* z, const float* gate, const float* fmap, const float* w1, const float* w2,
                      const float* w7, const float* dout, float* dmap, float* umax, float* cnt, float* dpool,
                      float* dsig, float* dz, float* dw7, int n, int c, int cr, int h, int w, cm_stream stream) {
  TailGeo g;
  if (n <= 0 || !cm_block_tail_supported(c, cr, h, w) || !tail_geometry(c, h, w, &g)) return -22;
  if (!y2 || !stats || !dout || !dmap || !umax || !cnt || !dpool || !dsig || !dz || !dw7) return -22;
  TailBwd a;
  a.y2 = y2; a.stats = stats; a.gamma = gamma; a.beta = beta; a.s = s; a.z = z; a.gate = gate; a.fmap = fmap;
  a.w1 = w1; a.w2 = w2; a.w7 = w7; a.dout = dout;
  a.dmap = dmap; a.umax = umax; a.cnt = cnt; a.dpool = dpool; a.dsig = dsig; a.dz = dz; a.dw7 = dw7;
  a.C = c; a.Cr = cr; a.H = h; a.W = w; a.LPS = g.lps; a.CS = g.cs; a.units = g.units; a.part = tail_bwd_part(g);
  tail_attrs();
  const size_t lds = tail_bwd_lds(g, c, cr, h, w);
  CM_TAIL_DISPATCH(block_tail_bwd_kernel, a, lds);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
