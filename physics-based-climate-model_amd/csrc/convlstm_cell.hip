// ConvLSTM cell: the pointwise gate / state update stage, forward and backward (BPTT step).
//
// Reference: ConvLSTMCell.forward (src/convlstm.py:11-19): gates = conv(cat[x,h]); i,f,o,g = chunk(4);
//   i,f,o = sigmoid; g = tanh; c' = f*c + i*g; h' = o*tanh(c').  Gate order along channels: i, f, o, g.
// The gate pre-activations are produced by cm_conv3x3 (x-projection for all T at once + per-step h-projection with
// the x-projection as residual, see engine).  This stage is the HBM-bound part of the cell: per element it reads 4
// pre-activations + c and writes 4 activations + c' + h' (training) -- 11 floats, all coalesced along pixels.
// Forward overwrites the pre-activations with the activations; backward overwrites the activations with d(pre-act).
#include "common.h"
#include "../../include/climate_hip.h"

namespace {

// parts (nullable): nparts partial sums of the recurrent projection ([nparts][B][4 Ch][HW] slices, zs apart, sample stride
// sp: what a "partial slices" cm_conv3x3_h3 launch stores), added to the pre-activations in slice order
__global__ void lstm_gates_fwd_kernel(float* __restrict__ gates, long long sg, const float* __restrict__ parts,
                                      long long sp, long long zs, int nparts, const float* __restrict__ c_prev,
                                      long long scp, float* __restrict__ c_out, long long sco,
                                      float* __restrict__ h_out, long long sho, int B, int Ch, int HW) {
  const long long per = (long long)Ch * HW, total = (long long)B * per;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long b = idx / per, r = idx % per;
    float* gp = gates + b * sg + r;
    float pi = gp[0], pf = gp[per], po = gp[2 * per], pg = gp[3 * per];
    for (int z = 0; z < nparts; ++z) {
      const float* pp = parts + z * zs + b * sp + r;
      pi += pp[0]; pf += pp[per]; po += pp[2 * per]; pg += pp[3 * per];
    }
    const float i = sigmoid_acc(pi);
    const float f = sigmoid_acc(pf);
    const float o = sigmoid_acc(po);
    const float g = tanhf(pg);
    const float cp = c_prev ? c_prev[b * scp + r] : 0.f;
    const float cn = f * cp + i * g;
    gp[0] = i; gp[per] = f; gp[2 * per] = o; gp[3 * per] = g;
    c_out[b * sco + r] = cn;
    h_out[b * sho + r] = o * tanhf(cn);
  }
}

// dc buffer: in = dL/dc_t carried from step t+1 (ignored when first != 0), out = dL/dc_{t-1}
// dh_b: nb >= 1 slices zb apart (the recurrent data gradient as partial sums), added in slice order
__global__ void lstm_gates_bwd_kernel(float* __restrict__ gates, long long sg, const float* __restrict__ c_prev,
                                      long long scp, const float* __restrict__ c_cur, long long scc,
                                      const float* __restrict__ dh_a, long long sa, const float* __restrict__ dh_b,
                                      long long sb, long long zb, int nb, float* __restrict__ dc, int first, int B,
                                      int Ch, int HW) {
  const long long per = (long long)Ch * HW, total = (long long)B * per;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long b = idx / per, r = idx % per;
    float* gp = gates + b * sg + r;
    const float i = gp[0], f = gp[per], o = gp[2 * per], g = gp[3 * per];
    float dh = 0.f;
    if (dh_a) dh += dh_a[b * sa + r];
    if (dh_b)
      for (int z = 0; z < nb; ++z) dh += dh_b[z * zb + b * sb + r];
    const float tc = tanhf(c_cur[b * scc + r]);
    const float cp = c_prev ? c_prev[b * scp + r] : 0.f;
    const float dct = (first ? 0.f : dc[idx]) + dh * o * (1.f - tc * tc);
    const float d_o = dh * tc;
    gp[0] = dct * g * i * (1.f - i);
    gp[per] = dct * cp * f * (1.f - f);
    gp[2 * per] = d_o * o * (1.f - o);
    gp[3 * per] = dct * i * (1.f - g * g);
    dc[idx] = dct * f;
  }
}

inline int grid_for(long long total, int bs) {
  long long b = (total + bs - 1) / bs;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int cm_lstm_gates_fwd(float* gates, long long sg, const float* c_prev, long long scp, float* c_out, long long sco,
                      float* h_out, long long sho, int b, int ch, int hw, cm_stream stream) {
  if (b <= 0 || ch <= 0 || hw <= 0) return -22;
  lstm_gates_fwd_kernel<<<grid_for((long long)b * ch * hw, 256), 256, 0, (hipStream_t)stream>>>(
      gates, sg, nullptr, 0, 0, 0, c_prev, scp, c_out, sco, h_out, sho, b, ch, hw);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_lstm_gates_fwd_parts(float* gates, long long sg, const float* parts, long long sp, long long zs, int nparts,
                            const float* c_prev, long long scp, float* c_out, long long sco, float* h_out,
                            long long sho, int b, int ch, int hw, cm_stream stream) {
  if (b <= 0 || ch <= 0 || hw <= 0 || nparts < 0 || (nparts > 0 && !parts)) return -22;
  lstm_gates_fwd_kernel<<<grid_for((long long)b * ch * hw, 256), 256, 0, (hipStream_t)stream>>>(
      gates, sg, parts, sp, zs, nparts, c_prev, scp, c_out, sco, h_out, sho, b, ch, hw);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_lstm_gates_bwd(float* gates, long long sg, const float* c_prev, long long scp, const float* c_cur,
                      long long scc, const float* dh_a, long long sa, const float* dh_b, long long sb, float* dc,
                      int first, int b, int ch, int hw, cm_stream stream) {
  if (b <= 0 || ch <= 0 || hw <= 0) return -22;
  lstm_gates_bwd_kernel<<<grid_for((long long)b * ch * hw, 256), 256, 0, (hipStream_t)stream>>>(
      gates, sg, c_prev, scp, c_cur, scc, dh_a, sa, dh_b, sb, 0, 1, dc, first, b, ch, hw);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_lstm_gates_bwd_parts(float* gates, long long sg, const float* c_prev, long long scp, const float* c_cur,
                            long long scc, const float* dh_a, long long sa, const float* dh_parts, long long sb,
                            long long zb, int nparts, float* dc, int first, int b, int ch, int hw, cm_stream stream) {
  if (b <= 0 || ch <= 0 || hw <= 0 || nparts < 1 || !dh_parts) return -22;
  lstm_gates_bwd_kernel<<<grid_for((long long)b * ch * hw, 256), 256, 0, (hipStream_t)stream>>>(
      gates, sg, c_prev, scp, c_cur, scc, dh_a, sa, dh_parts, sb, zb, nparts, dc, first, b, ch, hw);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
