// SE channel gate + CBAM spatial gate of ConvBlock: forward and backward.
//
// Reference: SEBlock (src/unet.py:6-17): x * sigmoid(W2 relu(W1 mean_hw x)), bias-free 1x1 convs, r = 8.
//            SpatialGate (src/unet.py:19-29): x * sigmoid(conv7x7(cat[mean_c x, amax_c x])), pad 3, no bias.
// Notation: a2 = activation entering SE, s = SE scale [N,C], U = a2*s, map = [mean_c U, max_c U], gate = sigmoid(conv7(map)),
// out = U*gate.  The squeeze (mean_hw a2) is produced by the GroupNorm kernel (norm_act.hip).
//
// Backward facts pinned by the golden fixtures: amax splits its gradient EQUALLY among tied channels (so the tie
// count map `cnt` is built from the bit-exact product a2*s), mean_c spreads 1/C.
#include <stdlib.h>
#include "common.h"
#include "se_wgrad.h"
#include "../../include/climate_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------ SE excite
// one workgroup per sample: z = W1 p (Cr), s = sigmoid(W2 relu(z)) (C)
__global__ __launch_bounds__(256) void se_excite_fwd_kernel(const float* __restrict__ pooled,
                                                             const float* __restrict__ w1,
                                                             const float* __restrict__ w2, float* __restrict__ z,
                                                             float* __restrict__ s, int C, int Cr) {
  extern __shared__ float sh[];  // [C] pooled + [Cr] hidden
  float* p = sh;
  float* hsh = sh + C;
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < C; c += blockDim.x) p[c] = pooled[(long long)n * C + c];
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  for (int r = wave; r < Cr; r += nw) {
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a += w1[(long long)r * C + c] * p[c];
    a = wave_sum(a);
    if (lane == 0) {
      z[(long long)n * Cr + r] = a;
      hsh[r] = fmaxf(a, 0.f);
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    float a = 0.f;
    for (int r = 0; r < Cr; ++r) a += w2[(long long)c * Cr + r] * hsh[r];
    s[(long long)n * C + c] = sigmoid_acc(a);
  }
}

// per sample: dsig = ds*s*(1-s); dh = W2^T dsig; dz = dh * 1[z>0]; dpool = W1^T dz
__global__ __launch_bounds__(256) void se_excite_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ s,
                                                             const float* __restrict__ z,
                                                             const float* __restrict__ w1,
                                                             const float* __restrict__ w2, float* __restrict__ dsig,
                                                             float* __restrict__ dz, float* __restrict__ dpool, int C,
                                                             int Cr) {
  extern __shared__ float sh[];  // [C] dsig + [Cr] dz
  float* dsg = sh;
  float* dzs = sh + C;
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < C; c += blockDim.x) {
    const float sv = s[(long long)n * C + c];
    const float v = ds[(long long)n * C + c] * sv * (1.f - sv);
    dsg[c] = v;
    dsig[(long long)n * C + c] = v;
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  for (int r = wave; r < Cr; r += nw) {
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a += w2[(long long)c * Cr + r] * dsg[c];
    a = wave_sum(a);
    if (lane == 0) {
      const float v = z[(long long)n * Cr + r] > 0.f ? a : 0.f;
      dzs[r] = v;
      dz[(long long)n * Cr + r] = v;
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    float a = 0.f;
    for (int r = 0; r < Cr; ++r) a += w1[(long long)r * C + c] * dzs[r];
    dpool[(long long)n * C + c] = a;
  }
}

// stand-alone launch of the SE weight gradients (se_wgrad.h); the engine normally lets the gated GroupNorm backward do it
__global__ __launch_bounds__(256) void se_wgrad_kernel(SeWgradArgs a) {
  __shared__ float part[8][33];
  se_wgrad_chunk(a, blockIdx.x, part);
}

// ------------------------------------------------------------------------------------------------ spatial gate fwd
// map[n,0,p] = mean_c(a2*s), map[n,1,p] = max_c(a2*s).  Workgroup = 64 pixels x 4 channel slices (one wave each);
// slices are combined through LDS in a fixed order (the max is order independent, the sum order is fixed).
__global__ __launch_bounds__(256) void spatial_stats_kernel(const float* __restrict__ a2, const float* __restrict__ s,
                                                             float* __restrict__ map, int C, int HW) {
  __shared__ float ssum[4][64], smax[4][64];
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + lane;
  const bool live = p < HW;
  const float* ap = a2 + (long long)n * C * HW + (live ? p : 0);
  const float* sp = s + (long long)n * C;
  const int cper = (C + 3) / 4, c0 = slice * cper, c1 = min(C, c0 + cper);
  float sum = 0.f, mx = -INFINITY;
#pragma unroll 8
  for (int c = c0; c < c1; ++c) {
    const float u = ap[(long long)c * HW] * sp[c];
    sum += u;
    mx = fmaxf(mx, u);
  }
  ssum[slice][lane] = sum;
  smax[slice][lane] = mx;
  __syncthreads();
  if (slice == 0 && live) {
    const float t = ((ssum[0][lane] + ssum[1][lane]) + ssum[2][lane]) + ssum[3][lane];
    const float m = fmaxf(fmaxf(smax[0][lane], smax[1][lane]), fmaxf(smax[2][lane], smax[3][lane]));
    map[((long long)n * 2) * HW + p] = t / (float)C;
    map[((long long)n * 2 + 1) * HW + p] = m;
  }
}

// SE excite + spatial statistics in one launch: every workgroup recomputes its sample's s = sigmoid(W2 relu(W1 p)) into
// LDS (2*C*Cr MACs, nothing next to its 64 x C pixel reads; bit-identical across workgroups: same order), the first
// pixel block of each sample also stores z and s for the backward pass.
template <int VEC>
__global__ __launch_bounds__(256) void se_spatial_stats_kernel(const float* __restrict__ pooled,
                                                                const float* __restrict__ w1,
                                                                const float* __restrict__ w2,
                                                                const float* __restrict__ a2, float* __restrict__ z,
                                                                float* __restrict__ s, float* __restrict__ map, int C,
                                                                int Cr, int HW) {
  extern __shared__ float sh[];  // [C] pooled, then s; [Cr] hidden
  __shared__ float ssum[4][64 * VEC], smax[4][64 * VEC];
  float* p = sh;
  float* hsh = sh + C;
  const int n = blockIdx.y, tid = threadIdx.x;
  const int lane = tid & 63, slice = tid >> 6;
  for (int c = tid; c < C; c += 256) p[c] = pooled[(long long)n * C + c];
  __syncthreads();
  for (int r = slice; r < Cr; r += 4) {
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a += w1[(long long)r * C + c] * p[c];
    a = wave_sum(a);
    if (lane == 0) {
      if (blockIdx.x == 0) z[(long long)n * Cr + r] = a;
      hsh[r] = fmaxf(a, 0.f);
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float a = 0.f;
    for (int r = 0; r < Cr; ++r) a += w2[(long long)c * Cr + r] * hsh[r];
    const float sv = sigmoid_acc(a);
    p[c] = sv;                                       // pooled[c] is dead: every thread only rewrites its own c
    if (blockIdx.x == 0) s[(long long)n * C + c] = sv;
  }
  __syncthreads();
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const int px = (blockIdx.x * 64 + lane) * VEC;              // HW % VEC == 0 (host)
  const bool live = px < HW;
  const float* ap = a2 + (long long)n * C * HW + (live ? px : 0);
  const int cper = (C + 3) / 4, c0 = slice * cper, c1 = min(C, c0 + cper);
  vec_t sum, mx;
#pragma unroll
  for (int q = 0; q < VEC; ++q) { sum[q] = 0.f; mx[q] = -INFINITY; }
#pragma unroll 8
  for (int c = c0; c < c1; ++c) {
    const vec_t av = *reinterpret_cast<const vec_t*>(ap + (long long)c * HW);
    const float sc = p[c];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      const float u = av[q] * sc;
      sum[q] += u;
      mx[q] = fmaxf(mx[q], u);
    }
  }
#pragma unroll
  for (int q = 0; q < VEC; ++q) {
    ssum[slice][lane * VEC + q] = sum[q];
    smax[slice][lane * VEC + q] = mx[q];
  }
  __syncthreads();
  if (slice == 0 && live) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      const int l = lane * VEC + q;
      const float t = ((ssum[0][l] + ssum[1][l]) + ssum[2][l]) + ssum[3][l];
      const float m = fmaxf(fmaxf(smax[0][l], smax[1][l]), fmaxf(smax[2][l], smax[3][l]));
      map[((long long)n * 2) * HW + px + q] = t / (float)C;
      map[((long long)n * 2 + 1) * HW + px + q] = m;
    }
  }
}

// gate = sigmoid(conv7x7(map)); out = a2 * s * gate.
// Workgroup = (row band, sample, channel split): the band's map rows (+-3, zero padded) are staged in LDS, the gate of
// the band's pixels is computed once into LDS (XB adjacent pixels per thread share their 7x(XB+6) window reads), then
// the band -- a contiguous run of BAND*W floats per channel -- is streamed with VEC-wide accesses.
constexpr int GATE_BAND = 8;
// POOL: the band's rows are streamed in pairs and MaxPool2d(2) of `out` (src/unet_convlstm_attention.py:21,25) is
// written next to it, which saves the pooling launch and its re-read of `out`.
template <int VEC, int XB, bool POOL>
__global__ __launch_bounds__(256) void spatial_apply_kernel(const float* __restrict__ a2, const float* __restrict__ s,
                                                             const float* __restrict__ map,
                                                             const float* __restrict__ w7, float* __restrict__ gate,
                                                             float* __restrict__ out, float* __restrict__ pooled,
                                                             int C, int H, int W, int c_per_split) {
  constexpr int BAND = GATE_BAND;
  extern __shared__ float sh[];
  const int PW = W + 6 + 4;                    // row pitch of the padded map tile (room for the XB <= 4 overshoot)
  float* msh = sh;                             // [2][BAND+6][PW]
  float* gsh = sh + 2 * (BAND + 6) * PW;       // [BAND*W] gate of the band
  __shared__ float wsh[98];
  const int tid = threadIdx.x;
  const int HW = H * W, n = blockIdx.y, y0 = blockIdx.x * BAND;
  const int rows = min(BAND, H - y0), npx = rows * W;
  if (tid < 98) wsh[tid] = w7[tid];
  const int plane = (BAND + 6) * PW;
  for (int i = tid; i < 2 * plane; i += 256) {
    const int ch = i / plane, r = (i % plane) / PW, cpos = (i % plane) % PW;
    const int yy = y0 - 3 + r, xx = cpos - 3;
    float v = 0.f;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = map[((long long)n * 2 + ch) * HW + yy * W + xx];
    msh[i] = v;
  }
  __syncthreads();
  const int xblocks = (W + XB - 1) / XB;
  for (int i = tid; i < rows * xblocks; i += 256) {
    const int r = i / xblocks, x0 = (i % xblocks) * XB;
    float acc[XB];
#pragma unroll
    for (int j = 0; j < XB; ++j) acc[j] = 0.f;
    // fixed accumulation order: ch, dy, dx (out-of-image taps contribute exact zeros from the padded tile)
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
      for (int dy = 0; dy < 7; ++dy) {
        const float* mr = msh + ch * plane + (r + dy) * PW + x0;
        float win[XB + 6];
#pragma unroll
        for (int q = 0; q < XB + 6; ++q) win[q] = mr[q];
#pragma unroll
        for (int dx = 0; dx < 7; ++dx) {
          const float wv = wsh[ch * 49 + dy * 7 + dx];
#pragma unroll
          for (int j = 0; j < XB; ++j) acc[j] += wv * win[j + dx];
        }
      }
#pragma unroll
    for (int j = 0; j < XB; ++j)
      if (x0 + j < W) gsh[r * W + x0 + j] = sigmoid_acc(acc[j]);
  }
  __syncthreads();
  if (blockIdx.z == 0)
    for (int i = tid; i < npx; i += 256) gate[(long long)n * HW + y0 * W + i] = gsh[i];
  const int c0 = blockIdx.z * c_per_split;
  const int nc = min(C, c0 + c_per_split) - c0;
  const float* sp = s + (long long)n * C;
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  if constexpr (POOL) {                        // host guarantees even rows, W % VEC == 0, VEC in {2, 4}
    const int wv = W / VEC, nvp = (rows / 2) * wv, Wo = W / 2;
    const unsigned magic = 0xFFFFFFFFu / (unsigned)max(nvp, 1) + 1u;
    for (int i = tid; i < nc * nvp; i += 256) {
      const int q = nvp == 1 ? i : (int)__umulhi((unsigned)i, magic);
      const int c = c0 + q, v = i - q * nvp;
      const int rp = v / wv, xv = v - rp * wv;
      const int g0 = (2 * rp) * W + xv * VEC;
      const long long off = ((long long)n * C + c) * HW + y0 * W + g0;
      const float sc = sp[c];
      const vec_t a0 = *reinterpret_cast<const vec_t*>(a2 + off);
      const vec_t a1 = *reinterpret_cast<const vec_t*>(a2 + off + W);
      vec_t o0, o1;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        o0[e] = (a0[e] * sc) * gsh[g0 + e];
        o1[e] = (a1[e] * sc) * gsh[g0 + W + e];
      }
      *reinterpret_cast<vec_t*>(out + off) = o0;
      *reinterpret_cast<vec_t*>(out + off + W) = o1;
      float* pp = pooled + (((long long)n * C + c) * (H / 2) + y0 / 2 + rp) * Wo + xv * (VEC / 2);
#pragma unroll
      for (int e = 0; e < VEC / 2; ++e)
        pp[e] = fmaxf(fmaxf(o0[2 * e], o0[2 * e + 1]), fmaxf(o1[2 * e], o1[2 * e + 1]));
    }
    return;
  }
  const int nv = npx / VEC;                    // host guarantees npx % VEC == 0 and aligned bases
  const unsigned magic = 0xFFFFFFFFu / (unsigned)max(nv, 1) + 1u;   // i / nv == umulhi(i, magic) for i*nv < 2^32
  for (int i = tid; i < nc * nv; i += 256) {
    const int q = nv == 1 ? i : (int)__umulhi((unsigned)i, magic);   // (magic overflows to 0 for nv == 1)
    const int c = c0 + q, v = i - q * nv;
    const long long off = ((long long)n * C + c) * HW + y0 * W + v * VEC;
    const float sc = sp[c];
    if constexpr (VEC == 1) {
      out[off] = (a2[off] * sc) * gsh[v];
    } else {
      const vec_t av = *reinterpret_cast<const vec_t*>(a2 + off);
      vec_t ov;
#pragma unroll
      for (int q = 0; q < VEC; ++q) ov[q] = (av[q] * sc) * gsh[v * VEC + q];
      *reinterpret_cast<vec_t*>(out + off) = ov;
    }
  }
}

// ------------------------------------------------------------------------------------------------ spatial gate bwd
// per pixel: dgate = sum_c dout*U; dgpre = dgate*gate*(1-gate); umax = max_c U; cnt = #{c: U == umax}
// Workgroup = 64*VEC pixels x 4 channel slices, combined through LDS (VEC = 4: 16-byte loads on wide images).
//
// The tie bookkeeping of the backward is SELF-CONSISTENT: (umax, cnt) are both derived here, in one pass, from the
// very products U = a2*s this launch computes, and the two consumers (se_bwd_reduce, the gated GroupNorm backward)
// compare their own U against THIS umax -- never against the map the forward stored.  cnt >= 1 by construction (the
// channel that set the running maximum counts itself), so `dmap_max / cnt` cannot divide by zero whatever the
// forward wrote.  (Round 1 rebuilt cnt by testing a2*s against the forward's stored maximum: any disagreement
// between the two evaluations produced cnt = 0 and non-finite gradients.)
template <int VEC>
__global__ __launch_bounds__(256) void gate_bwd_reduce_kernel(const float* __restrict__ dout,
                                                               const float* __restrict__ a2,
                                                               const float* __restrict__ s,
                                                               const float* __restrict__ gate,
                                                               float* __restrict__ dgpre, float* __restrict__ cnt,
                                                               float* __restrict__ umax, int C, int HW) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  __shared__ float sdg[4][64 * VEC], scn[4][64 * VEC], smx[4][64 * VEC];
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int p = (blockIdx.x * 64 + lane) * VEC;               // HW % VEC == 0 (host)
  const bool live = p < HW;
  const int pp = live ? p : 0;
  const float* sp = s + (long long)n * C;
  const int cper = (C + 3) / 4, c0 = slice * cper, c1 = min(C, c0 + cper);
  vec_t dg, k, mx;
#pragma unroll
  for (int q = 0; q < VEC; ++q) { dg[q] = 0.f; k[q] = 0.f; mx[q] = -INFINITY; }
  // one channel: running (max, multiplicity), branch-free -- a new maximum restarts the count at 1, an equal value adds 1
  auto channel = [&](int c, float sc) {
    const long long i = ((long long)n * C + c) * HW + pp;
    const vec_t av = *reinterpret_cast<const vec_t*>(a2 + i), dv = *reinterpret_cast<const vec_t*>(dout + i);
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      const float u = av[q] * sc;
      dg[q] += dv[q] * u;
      const float same = (u == mx[q]) ? 1.f : 0.f;
      k[q] = (u > mx[q]) ? 1.f : k[q] + same;
      mx[q] = fmaxf(mx[q], u);
    }
  };
  if (((cper | C) & 3) == 0) {
    // the wave's channel range is a multiple of four: the SE scales come as 16-byte (uniform-address) loads
    for (int c = c0; c < c1; c += 4) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(sp + c);
      channel(c, s4.x);
      channel(c + 1, s4.y);
      channel(c + 2, s4.z);
      channel(c + 3, s4.w);
    }
  } else {
#pragma unroll 4
    for (int c = c0; c < c1; ++c) channel(c, sp[c]);
  }
#pragma unroll
  for (int q = 0; q < VEC; ++q) {
    sdg[slice][lane * VEC + q] = dg[q];
    scn[slice][lane * VEC + q] = k[q];
    smx[slice][lane * VEC + q] = mx[q];
  }
  __syncthreads();
  if (slice == 0 && live) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      const int l = lane * VEC + q;
      const float t = ((sdg[0][l] + sdg[1][l]) + sdg[2][l]) + sdg[3][l];
      const float m = fmaxf(fmaxf(smx[0][l], smx[1][l]), fmaxf(smx[2][l], smx[3][l]));
      float kk = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) kk += (smx[j][l] == m) ? scn[j][l] : 0.f;
      const float g = gate[(long long)n * HW + p + q];
      dgpre[(long long)n * HW + p + q] = t * g * (1.f - g);
      cnt[(long long)n * HW + p + q] = fmaxf(kk, 1.f);      // (>= 1 already unless every U is NaN)
      umax[(long long)n * HW + p + q] = m;
    }
  }
}

// dmap[n,ch,p] = sum_taps dgpre[n, p - (tap-3)] * w7[ch][tap];  dW7[ch][tap] += sum_{n,p} map[n,ch,p+tap-3]*dgpre[n,p]
// One workgroup per (row band, sample); the band's dgpre rows (+-3) and map rows (+-3) are staged zero padded in LDS.
// Both parts are register blocked over XB adjacent pixels (a 7-tap row needs XB+6 window reads for 7*XB FMAs):
//   dmap: thread <-> (channel, row, x block);   dW7: thread <-> ((channel, dy), row, x half), 7 dx accumulators,
//   the 16 (row, half) partials of a tap are combined through LDS in a fixed order, one atomic per tap and workgroup.
template <int BAND>
__global__ __launch_bounds__(256) void conv7_bwd_kernel(const float* __restrict__ dgpre,
                                                         const float* __restrict__ map,
                                                         const float* __restrict__ w7, float* __restrict__ dmap,
                                                         float* __restrict__ partials, int H, int W) {
  constexpr int XB = 4;
  extern __shared__ float sh[];
  const int PW = W + 6 + XB;
  const int plane = (BAND + 6) * PW;
  float* dsh = sh;                          // [(BAND+6)][PW]  dgpre rows y0-3 .. y0+BAND+2, zero padded
  float* msh = sh + plane;                  // [2][(BAND+6)][PW] map rows, zero padded
  float* part = sh + 3 * plane;             // [2*BAND][98]
  __shared__ float wsh[98];
  const int n = blockIdx.y, y0 = blockIdx.x * BAND, HW = H * W, tid = threadIdx.x;
  if (tid < 98) wsh[tid] = w7[tid];
  for (int i = tid; i < 3 * plane; i += 256) {
    const int which = i / plane, r = (i % plane) / PW, cpos = (i % plane) % PW;
    const int yy = y0 - 3 + r, xx = cpos - 3;
    float v = 0.f;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      v = which == 0 ? dgpre[(long long)n * HW + yy * W + xx]
                     : map[((long long)n * 2 + (which - 1)) * HW + yy * W + xx];
    }
    sh[i] = v;
  }
  __syncthreads();
  // ---- dmap (flipped kernel) ----
  const int rows = min(BAND, H - y0);
  const int xblocks = (W + XB - 1) / XB;
  for (int i = tid; i < 2 * rows * xblocks; i += 256) {
    const int ch = i / (rows * xblocks), r = (i / xblocks) % rows, x0 = (i % xblocks) * XB;
    float acc[XB];
#pragma unroll
    for (int j = 0; j < XB; ++j) acc[j] = 0.f;
    for (int dy = 0; dy < 7; ++dy) {
      const float* dr = dsh + (r + 6 - dy) * PW + x0;
      float win[XB + 6];
#pragma unroll
      for (int q = 0; q < XB + 6; ++q) win[q] = dr[q];
#pragma unroll
      for (int dx = 0; dx < 7; ++dx) {
        const float wv = wsh[ch * 49 + dy * 7 + dx];
#pragma unroll
        for (int j = 0; j < XB; ++j) acc[j] += wv * win[j + 6 - dx];
      }
    }
    float* dst = dmap + ((long long)n * 2 + ch) * HW + (y0 + r) * W + x0;
#pragma unroll
    for (int j = 0; j < XB; ++j)
      if (x0 + j < W) dst[j] = acc[j];
  }
  // ---- weight gradient ----
  if (tid < 14 * BAND * 2) {
    const int chdy = tid % 14, rs = tid / 14;          // rs = row*2 + half
    const int ch = chdy / 7, dy = chdy % 7, r = rs >> 1, hs = rs & 1;
    const int seg = ((W + 2 * XB - 1) / (2 * XB)) * XB; // x-half length, a multiple of XB
    const float* mr = msh + ch * plane + (r + dy) * PW;
    const float* dr = dsh + (r + 3) * PW + 3;
    float acc[7];
#pragma unroll
    for (int dx = 0; dx < 7; ++dx) acc[dx] = 0.f;
    for (int x0 = hs * seg; x0 < min(W, (hs + 1) * seg); x0 += XB) {
      float win[XB + 6], dv[XB];
#pragma unroll
      for (int q = 0; q < XB + 6; ++q) win[q] = mr[x0 + q];
#pragma unroll
      for (int j = 0; j < XB; ++j) dv[j] = dr[x0 + j];   // zero beyond the image (padded tile)
#pragma unroll
      for (int dx = 0; dx < 7; ++dx)
#pragma unroll
        for (int j = 0; j < XB; ++j) acc[dx] += win[j + dx] * dv[j];
    }
#pragma unroll
    for (int dx = 0; dx < 7; ++dx) part[rs * 98 + ch * 49 + dy * 7 + dx] = acc[dx];
  }
  __syncthreads();
  // Thousands of workgroups adding into the same 98 addresses serialise in the L2 (measured: +15 us at 1152
  // workgroups); every workgroup stores its 98 partial sums instead and conv7_fold_kernel adds them up.
  if (tid < 98) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 2 * BAND; ++k) t += part[k * 98 + tid];
    partials[(long long)(blockIdx.y * gridDim.x + blockIdx.x) * 98 + tid] = t;
  }
}

// dw7[tap] += sum_rows partials[row][tap]; workgroup b takes rows [b*rpb, (b+1)*rpb), two row-interleaved halves
__device__ __forceinline__ void conv7_fold(const float* __restrict__ partials, int nrows, int rpb,
                                           float* __restrict__ dw7) {
  __shared__ float sh[128];
  const int tap = threadIdx.x & 127, hs = threadIdx.x >> 7;
  const int r0 = blockIdx.x * rpb, r1 = min(nrows, r0 + rpb);
  float a0 = 0.f, a1 = 0.f;
  if (tap < 98) {
    int r = r0 + hs;
    for (; r + 2 < r1; r += 4) {
      a0 += partials[(long long)r * 98 + tap];
      a1 += partials[(long long)(r + 2) * 98 + tap];
    }
    if (r < r1) a0 += partials[(long long)r * 98 + tap];
  }
  if (hs == 1) sh[tap] = a0 + a1;
  __syncthreads();
  if (hs == 0 && tap < 98) unsafeAtomicAdd(dw7 + tap, (a0 + a1) + sh[tap]);
}

__global__ __launch_bounds__(256) void conv7_fold_kernel(const float* __restrict__ partials, int nrows, int rpb,
                                                          float* __restrict__ dw7) {
  conv7_fold(partials, nrows, rpb, dw7);
}

// ds[n,c] = sum_p dU * a2, with dU = dout*gate + dmapA/C + dmapM*[U==umax]/cnt.  One wave per (n, CG channels), VEC
// pixels per lane and load; the five per-pixel maps are loaded once for the CG channels.  umax / cnt [N,HW] are the
// pair written by gate_bwd_reduce (see there).
template <int VEC, int CG>
__global__ __launch_bounds__(256) void se_bwd_reduce_kernel(const float* __restrict__ dout,
                                                             const float* __restrict__ a2,
                                                             const float* __restrict__ s,
                                                             const float* __restrict__ gate,
                                                             const float* __restrict__ dmap,
                                                             const float* __restrict__ umax,
                                                             const float* __restrict__ cnt, float* __restrict__ ds,
                                                             int NC, int C, int HW,
                                                             const float* __restrict__ c7_partials, int c7_rows,
                                                             int c7_rpb, float* __restrict__ dw7) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  // side duty of the first workgroups: fold the preceding cm_conv7_bwd's per-workgroup dW7 partials (saves a launch)
  if (c7_partials && blockIdx.x * c7_rpb < c7_rows) conv7_fold(c7_partials, c7_rows, c7_rpb, dw7);
  const int lane = threadIdx.x & 63;
  const int nc = (blockIdx.x * 4 + (threadIdx.x >> 6)) * CG;     // first of this wave's CG channels (C % CG == 0)
  if (nc >= NC) return;
  const int n = nc / C;
  const float inv_c = 1.f / (float)C;
  const vec_t* gp = reinterpret_cast<const vec_t*>(gate + (long long)n * HW);
  const vec_t* da = reinterpret_cast<const vec_t*>(dmap + (long long)n * 2 * HW);
  const vec_t* dm = reinterpret_cast<const vec_t*>(dmap + ((long long)n * 2 + 1) * HW);
  const vec_t* mx = reinterpret_cast<const vec_t*>(umax + (long long)n * HW);
  const vec_t* ct = reinterpret_cast<const vec_t*>(cnt + (long long)n * HW);
  const vec_t* dop = reinterpret_cast<const vec_t*>(dout + (long long)nc * HW);
  const vec_t* ap = reinterpret_cast<const vec_t*>(a2 + (long long)nc * HW);
  float sc[CG], acc[CG];
#pragma unroll
  for (int j = 0; j < CG; ++j) {
    sc[j] = s[nc + j];
    acc[j] = 0.f;
  }
  const int nv = HW / VEC;
  for (int p = lane; p < nv; p += 64) {
    const vec_t g = gp[p], va = da[p], vm = dm[p], m = mx[p], k = ct[p];
#pragma unroll
    for (int j = 0; j < CG; ++j) {
      const vec_t a = ap[(long long)j * nv + p], d = dop[(long long)j * nv + p];
#pragma unroll
      for (int q = 0; q < VEC; ++q) {
        const float u = a[q] * sc[j];
        float dU = d[q] * g[q] + va[q] * inv_c;
        if (u == m[q]) dU += vm[q] / k[q];
        acc[j] += dU * a[q];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < CG; ++j) {
    const float t = wave_sum(acc[j]);
    if (lane == 0) ds[nc + j] = t;
  }
}

}  // namespace

extern "C" {

int cm_se_excite_fwd(const float* pooled, const float* w1, const float* w2, float* z, float* s, int n, int c, int cr,
                     cm_stream stream) {
  if (n <= 0 || c <= 0 || cr <= 0) return -22;
  se_excite_fwd_kernel<<<n, 256, (c + cr) * sizeof(float), (hipStream_t)stream>>>(pooled, w1, w2, z, s, c, cr);
  CM_CHECK_LAUNCH();
  return 0;
}

/* (A single-launch variant whose weight-gradient workgroups recompute dsig / dz instead of waiting for them was
 * measured 2x SLOWER under graph replay -- 164 vs 75 us over the seven blocks: its serial per-sample dots are latency
 * bound -- so the two-launch form stays.  tools/se_bwd_bench.py times this entry point.) */
int cm_se_excite_bwd(const float* ds, const float* s, const float* z, const float* pooled, const float* w1,
                     const float* w2, float* dsig, float* dz, float* dpool, float* dw1, float* dw2, int n, int c,
                     int cr, cm_stream stream) {
  if (n <= 0 || c <= 0 || cr <= 0) return -22;
  se_excite_bwd_kernel<<<n, 256, (c + cr) * sizeof(float), (hipStream_t)stream>>>(ds, s, z, w1, w2, dsig, dz, dpool,
                                                                                   c, cr);
  CM_CHECK_LAUNCH();
  if (dw1 && dw2) {      // (NULL: the caller hands the weight gradients to cm_gn_silu_bwd_gated as a side duty)
    SeWgradArgs wa;
    wa.dsig = dsig; wa.dz = dz; wa.z = z; wa.pooled = pooled; wa.dw1 = dw1; wa.dw2 = dw2; wa.N = n; wa.C = c; wa.Cr = cr;
    se_wgrad_kernel<<<cdiv(2 * c * cr, 32), 256, 0, (hipStream_t)stream>>>(wa);
    CM_CHECK_LAUNCH();
  }
  return 0;
}

int cm_spatial_stats(const float* a2, const float* s, float* map, int n, int c, int hw, cm_stream stream) {
  if (n <= 0 || c <= 0 || hw <= 0) return -22;
  spatial_stats_kernel<<<dim3(cdiv(hw, 64), n), 256, 0, (hipStream_t)stream>>>(a2, s, map, c, hw);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_se_spatial_stats(const float* pooled, const float* w1, const float* w2, const float* a2, float* z, float* s,
                        float* map, int n, int c, int cr, int hw, cm_stream stream) {
  if (n <= 0 || c <= 0 || cr <= 0 || hw <= 0) return -22;
  const size_t lds = (c + cr) * sizeof(float);
  // wide images: 4 pixels per lane (16-byte loads); the workgroup count must still cover the 256 CUs
  if (hw % 4 == 0 && ((uintptr_t)a2 & 15) == 0 && (long long)n * (hw / 256) >= 1024)
    se_spatial_stats_kernel<4><<<dim3(cdiv(hw, 256), n), 256, lds, (hipStream_t)stream>>>(pooled, w1, w2, a2, z, s, map,
                                                                                           c, cr, hw);
  else
    se_spatial_stats_kernel<1><<<dim3(cdiv(hw, 64), n), 256, lds, (hipStream_t)stream>>>(pooled, w1, w2, a2, z, s, map,
                                                                                          c, cr, hw);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_spatial_apply(const float* a2, const float* s, const float* map, const float* w7, float* gate, float* out,
                     float* pooled, int n, int c, int h, int w, cm_stream stream) {
  if (n <= 0 || c <= 0 || h <= 0 || w <= 0 || (pooled && ((h | w) & 1))) return -22;
  const int hw = h * w, bands = cdiv(h, GATE_BAND);
  const size_t lds = (size_t)(2 * (GATE_BAND + 6) * (w + 10) + GATE_BAND * w) * sizeof(float);
  if (lds > 60 * 1024) return -22;  // W up to ~400
  int splits = 1;
  while ((long long)bands * n * splits < 1024 && splits * 8 <= c) splits *= 2;  // enough workgroups for 256 CUs
  const int cps = cdiv(c, splits);
  const dim3 grid(bands, n, cdiv(c, cps));
  // every band starts at a multiple of BAND*W floats inside a plane of HW floats; the last band may be shorter
  const int last = (h - (bands - 1) * GATE_BAND) * w;
  const bool al = (((uintptr_t)a2 | (uintptr_t)out) & 15) == 0;
  const int band_px = GATE_BAND * w;
  int vec = (al && hw % 4 == 0 && band_px % 4 == 0 && last % 4 == 0)   ? 4
            : (al && hw % 2 == 0 && band_px % 2 == 0 && last % 2 == 0) ? 2
                                                                       : 1;
  if (pooled) {
    if (!al) return -22;
    vec = (w % 4 == 0) ? 4 : 2;                     // row pairs: vectors must not straddle rows
  }
  // x blocking of the gate phase: 1.  The blocked forms (4 for W >= 64, 2 for W >= 32: ~150 busy threads per band at
  // every level) cost 220-250 VGPRs = 2 waves per SIMD, and the streaming phase that follows is what the kernel's time
  // is made of: 182 -> 162 us per step with 4 waves per SIMD (CM_APPLY_XB=4|2 selects the blocked forms).
  static const int xb_force = getenv("CM_APPLY_XB") ? atoi(getenv("CM_APPLY_XB")) : 0;
  const int xb = (xb_force == 4 || xb_force == 2) ? xb_force : 1;
#define CM_APPLY(V, X)                                                                                               \
  if (vec == V && xb == X) {                                                                                         \
    if (pooled && V > 1)                                                                                             \
      spatial_apply_kernel<(V > 1 ? V : 2), X, true><<<grid, 256, lds, (hipStream_t)stream>>>(a2, s, map, w7, gate,  \
                                                                                               out, pooled, c, h, w, \
                                                                                               cps);                 \
    else                                                                                                             \
      spatial_apply_kernel<V, X, false><<<grid, 256, lds, (hipStream_t)stream>>>(a2, s, map, w7, gate, out, nullptr, \
                                                                                 c, h, w, cps);                      \
  }
  CM_APPLY(4, 4) CM_APPLY(4, 2) CM_APPLY(4, 1) CM_APPLY(2, 4) CM_APPLY(2, 2) CM_APPLY(2, 1) CM_APPLY(1, 4)
  CM_APPLY(1, 2) CM_APPLY(1, 1)
#undef CM_APPLY
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_gate_bwd_reduce(const float* dout, const float* a2, const float* s, const float* gate, float* dgpre,
                       float* cnt, float* umax, int n, int c, int hw, cm_stream stream) {
  if (n <= 0 || c <= 0 || hw <= 0 || !cnt || !umax) return -22;
  const bool wide = hw % 4 == 0 && (((uintptr_t)dout | (uintptr_t)a2) & 15) == 0 && (long long)n * (hw / 256) >= 1024;
  if (wide)
    gate_bwd_reduce_kernel<4><<<dim3(cdiv(hw, 256), n), 256, 0, (hipStream_t)stream>>>(dout, a2, s, gate, dgpre, cnt,
                                                                                        umax, c, hw);
  else
    gate_bwd_reduce_kernel<1><<<dim3(cdiv(hw, 64), n), 256, 0, (hipStream_t)stream>>>(dout, a2, s, gate, dgpre, cnt,
                                                                                       umax, c, hw);
  CM_CHECK_LAUNCH();
  return 0;
}

long long cm_conv7_bwd_scratch_elems(int n, int h) {
  return (n <= 0 || h <= 0) ? -22 : (long long)n * cdiv(h, 8) * 98;
}

static void conv7_fold_split(int nrows, int* fb, int* rpb) {
  *fb = max(1, min(8, nrows / 32));
  *rpb = cdiv(nrows, *fb);
}

int cm_conv7_bwd(const float* dgpre, const float* map, const float* w7, float* dmap, float* dw7, float* scratch, int n,
                 int h, int w, cm_stream stream) {
  if (n <= 0 || h <= 0 || w <= 0 || !scratch) return -22;
  constexpr int BAND = 8;
  const size_t lds = ((size_t)3 * (BAND + 6) * (w + 10) + 2 * BAND * 98) * sizeof(float);
  if (lds > 60 * 1024) return -22;  // W up to ~320
  const int bands = cdiv(h, BAND), nrows = bands * n;
  conv7_bwd_kernel<BAND><<<dim3(bands, n), 256, lds, (hipStream_t)stream>>>(dgpre, map, w7, dmap, scratch, h, w);
  CM_CHECK_LAUNCH();
  if (dw7) {
    int fb, rpb;
    conv7_fold_split(nrows, &fb, &rpb);
    conv7_fold_kernel<<<cdiv(nrows, rpb), 256, 0, (hipStream_t)stream>>>(scratch, nrows, rpb, dw7);
    CM_CHECK_LAUNCH();
  }
  return 0;
}

int cm_se_bwd_reduce(const float* dout, const float* a2, const float* s, const float* gate, const float* dmap,
                     const float* umax, const float* cnt, float* ds, int n, int c, int hw, const float* c7_partials,
                     int c7_rows, float* dw7, cm_stream stream) {
  if (n <= 0 || c <= 0 || hw <= 0 || (c7_partials && (c7_rows <= 0 || !dw7))) return -22;
  const int cg = (c % 4 == 0 && (long long)n * c >= 4096) ? 4 : 1;   // channels per wave (needs enough waves)
  const unsigned grid = (unsigned)cdiv((long long)n * c / cg, 4);
  int fb = 1, rpb = 0;
  if (c7_partials) {
    conv7_fold_split(c7_rows, &fb, &rpb);
    if ((unsigned)cdiv(c7_rows, rpb) > grid) return -22;
  }
  const bool al = (((uintptr_t)dout | (uintptr_t)a2 | (uintptr_t)gate | (uintptr_t)dmap | (uintptr_t)umax |
                    (uintptr_t)cnt) & 15) == 0;
#define CM_SEBR(V)                                                                                               \
  do {                                                                                                           \
    if (cg == 4)                                                                                                 \
      se_bwd_reduce_kernel<V, 4><<<grid, 256, 0, (hipStream_t)stream>>>(dout, a2, s, gate, dmap, umax, cnt, ds,   \
                                                                        n * c, c, hw, c7_partials, c7_rows, rpb, \
                                                                        dw7);                                    \
    else                                                                                                         \
      se_bwd_reduce_kernel<V, 1><<<grid, 256, 0, (hipStream_t)stream>>>(dout, a2, s, gate, dmap, umax, cnt, ds,   \
                                                                        n * c, c, hw, c7_partials, c7_rows, rpb, \
                                                                        dw7);                                    \
  } while (0)
  if (al && hw % 4 == 0) CM_SEBR(4);
  else if (al && hw % 2 == 0) CM_SEBR(2);
  else CM_SEBR(1);
#undef CM_SEBR
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
