// SE excite weight gradients, shared by attention_gates.hip (stand-alone launch) and norm_act.hip (side duty of the
// gated GroupNorm backward, which follows cm_se_excite_bwd in the chain and saves the launch):
//   dW2[c][r] += sum_n dsig[n,c] relu(z[n,r]);   dW1[r][c] += sum_n dz[n,r] pooled[n,c]
// One call handles 32 weights x 8 sample slices with 256 threads; slices are combined through LDS in a fixed order.
#pragma once
#include "common.h"

struct SeWgradArgs {
  const float* dsig;    // [N,C]   (NULL: no work)
  const float* dz;      // [N,Cr]
  const float* z;       // [N,Cr]
  const float* pooled;  // [N,C]
  float* dw1;           // [Cr,C]  accumulated
  float* dw2;           // [C,Cr]  accumulated
  int N, C, Cr;
};

__device__ __forceinline__ void se_wgrad_chunk(const SeWgradArgs& a, int chunk, float (*part)[33]) {
  const int wl = threadIdx.x & 31, sl = threadIdx.x >> 5;      // the first 256 threads work (callers may have 512)
  const bool act = threadIdx.x < 256;
  const int i = chunk * 32 + wl;
  const int CCr = a.C * a.Cr, total = 2 * CCr;
  float acc = 0.f;
  if (act && i < total) {
    if (i < CCr) {  // dW2[c][r]
      const int c = i / a.Cr, r = i % a.Cr;
      for (int n = sl; n < a.N; n += 8) acc += a.dsig[(long long)n * a.C + c] * fmaxf(a.z[(long long)n * a.Cr + r], 0.f);
    } else {        // dW1[r][c]
      const int j = i - CCr;
      const int r = j / a.C, c = j % a.C;
      for (int n = sl; n < a.N; n += 8) acc += a.dz[(long long)n * a.Cr + r] * a.pooled[(long long)n * a.C + c];
    }
  }
  if (act) part[sl][wl] = acc;
  __syncthreads();
  if (sl == 0 && i < total) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += part[k][wl];
    if (i < CCr) a.dw2[i] += t; else a.dw1[i - CCr] += t;
  }
  __syncthreads();
}
