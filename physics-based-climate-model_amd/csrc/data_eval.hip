// The callers either side of the hot path, on the device:
//
//  * cm_build_windows -- ClimateDataset.__getitem__ + DataLoader collation (main_final.py:97-154, 483-494): the batch
//    x[b, t] = inputs[idx_b - T + 1 + t] (all-zero frame where that index is negative: left padding in normalised space,
//    main_final.py:76,127-131), y[b] = outputs[idx_b], gathered from the device-resident normalised data set straight
//    into the trainer's input buffers.  Replaces a per-sample python loop + torch.stack + a 13 MB host-to-device copy
//    per step by one HBM-bound launch (26 MB moved at config 2).
//  * cm_eval_accumulate / cm_eval_finalize -- validation_step + _evaluate_predictions (main_final.py:563-668):
//    Normalizer.inverse_transform_output (src/utils_final.py:130-206: zscore, minimax, log1p -> expm1, sqrt, pow) fused
//    with the running sums the three area-weighted climate metrics need (src/utils_final.py:282-302, 387-406; same
//    numbers as _climate_kaggle_metric.py:109-142): per (variable, pixel) sum p, sum p^2, sum t, sum t^2, sum (p-t)^2 in
//    float64, then monthly RMSE, time-mean RMSE and time-stddev MAE with cos(latitude) weights.  Replaces
//    .cpu().numpy() + xarray per validation batch.
#include "common.h"
#include "../../include/climate_hip.h"

namespace {

// one workgroup row per (b, t) frame; frames are chw floats, copied as 16-byte vectors when aligned
__global__ __launch_bounds__(256) void build_windows_kernel(const float* __restrict__ inputs,
                                                            const float* __restrict__ outputs,
                                                            const long long* __restrict__ idx, float* __restrict__ x,
                                                            float* __restrict__ y, int T, long long chw_in,
                                                            long long chw_out, long long total) {
  const int f = blockIdx.y;                 // frame id: b * (T + 1) + slot; slot T = the target
  const int b = f / (T + 1), slot = f % (T + 1);
  const long long i0 = idx[b];
  const float* src;
  float* dst;
  long long n;
  bool zero = false;
  if (slot == T) {
    n = chw_out;
    zero = i0 < 0 || i0 >= total;
    src = outputs + (zero ? 0 : i0) * chw_out;
    dst = y + (long long)b * chw_out;
  } else {
    const long long it = i0 - T + 1 + slot;   // main_final.py:122
    n = chw_in;
    zero = it < 0 || it >= total;             // before the start of the data: the all-zero padding template
    src = inputs + (zero ? 0 : it) * chw_in;
    dst = x + ((long long)b * T + slot) * chw_in;
  }
  const long long stride = (long long)gridDim.x * blockDim.x, t0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (((n | (long long)(uintptr_t)src | (long long)(uintptr_t)dst) & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* d4 = reinterpret_cast<f32x4*>(dst);
    for (long long i = t0; i < n / 4; i += stride) d4[i] = zero ? f32x4{0.f, 0.f, 0.f, 0.f} : s4[i];
  } else {
    for (long long i = t0; i < n; i += stride) dst[i] = zero ? 0.f : src[i];
  }
}

// inverse of Normalizer.normalize for one output variable: method codes follow cm_denorm_method
__device__ __forceinline__ double denorm(double v, int method, double a, double b, double lam) {
  switch (method) {
    case 1: return v * b + a;                       // zscore: x * std + mean
    case 2: return v * (b - a) + a;                 // minimax: x * (max - min) + min
    case 3: return expm1(v * b + a);                // log1p: expm1(x * std_of_log + mean_of_log)
    case 4: { const double s = v * b + a; return s * s; }          // sqrt
    case 5: return pow(v * b + a, 1.0 / lam);       // pow
    default: return v;                              // 0: pass through (no config for this variable)
  }
}

// moments[c][k][p], k = sum p, sum p^2, sum t, sum t^2, sum (p-t)^2 (float64); one thread per (c, pixel), loop over n
__global__ __launch_bounds__(256) void eval_accumulate_kernel(const float* __restrict__ pred,
                                                              const float* __restrict__ target,
                                                              const double* __restrict__ params,   // [C][4]
                                                              double* __restrict__ moments, int N, int C, int HW,
                                                              int target_is_normalized) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (p >= HW) return;
  const int method = (int)params[c * 4 + 0];
  const double a = params[c * 4 + 1], b = params[c * 4 + 2], lam = params[c * 4 + 3];
  double sp = 0, spp = 0, st = 0, stt = 0, sd = 0;
  for (int n = 0; n < N; ++n) {
    const long long i = ((long long)n * C + c) * HW + p;
    const double pv = denorm((double)pred[i], method, a, b, lam);
    const double tv = target_is_normalized ? denorm((double)target[i], method, a, b, lam) : (double)target[i];
    sp += pv; spp += pv * pv; st += tv; stt += tv * tv;
    sd += (pv - tv) * (pv - tv);
  }
  double* m = moments + (long long)c * 5 * HW + p;
  m[0] += sp; m[HW] += spp; m[2 * HW] += st; m[3 * HW] += stt; m[4 * HW] += sd;
}

// out[c][0..2] = monthly RMSE, time-mean RMSE, time-stddev MAE: weighted means over (y, x) with weights w[y] / sum w
__global__ __launch_bounds__(256) void eval_finalize_kernel(const double* __restrict__ moments,
                                                            const double* __restrict__ lat_w, double count,
                                                            double* __restrict__ out, int H, int W) {
  __shared__ double red[3][4];
  const int c = blockIdx.x, HW = H * W;
  const double* m = moments + (long long)c * 5 * HW;
  double a0 = 0, a1 = 0, a2 = 0, wsum = 0;
  for (int y = threadIdx.x; y < H; y += blockDim.x) wsum += lat_w[y];
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const double w = lat_w[p / W];
    const double mp = m[p] / count, mt = m[2 * HW + p] / count;
    const double vp = fmax(m[HW + p] / count - mp * mp, 0.0), vt = fmax(m[3 * HW + p] / count - mt * mt, 0.0);
    a0 += w * m[4 * HW + p] / count;                 // mean_t (p - t)^2
    a1 += w * (mp - mt) * (mp - mt);
    a2 += w * fabs(sqrt(vp) - sqrt(vt));             // population standard deviation (ddof = 0), as xarray / numpy
  }
  auto wsum64 = [](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  };
  a0 = wsum64(a0); a1 = wsum64(a1); a2 = wsum64(a2); wsum = wsum64(wsum);
  __shared__ double wred[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[0][wv] = a0; red[1][wv] = a1; red[2][wv] = a2; wred[wv] = wsum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s[3] = {0, 0, 0}, ws = 0;
    for (int k = 0; k < 4; ++k) { s[0] += red[0][k]; s[1] += red[1][k]; s[2] += red[2][k]; ws += wred[k]; }
    const double norm = ws * (double)W;              // sum of the weights over every (y, x)
    out[c * 3 + 0] = sqrt(s[0] / norm);
    out[c * 3 + 1] = sqrt(s[1] / norm);
    out[c * 3 + 2] = s[2] / norm;
  }
}

}  // namespace

extern "C" {

int cm_build_windows(const float* inputs, const float* outputs, const long long* idx_dev, float* x, float* y, int b,
                     int t, long long chw_in, long long chw_out, long long total, cm_stream stream) {
  if (b <= 0 || t <= 0 || chw_in <= 0 || chw_out <= 0 || total <= 0 || !inputs || !outputs || !idx_dev || !x || !y)
    return -22;
  const int bx = (int)((chw_in / 4 + 255) / 256 < 64 ? max(1LL, (chw_in / 4 + 255) / 256) : 64);
  build_windows_kernel<<<dim3(bx, b * (t + 1)), 256, 0, (hipStream_t)stream>>>(inputs, outputs, idx_dev, x, y, t, chw_in,
                                                                               chw_out, total);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_eval_accumulate(const float* pred, const float* target, const double* params_dev, double* moments, int n, int c,
                       int hw, int target_is_normalized, cm_stream stream) {
  if (n <= 0 || c <= 0 || hw <= 0 || !pred || !target || !params_dev || !moments) return -22;
  eval_accumulate_kernel<<<dim3(cdiv(hw, 256), c), 256, 0, (hipStream_t)stream>>>(pred, target, params_dev, moments, n, c,
                                                                                  hw, target_is_normalized);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_eval_finalize(const double* moments, const double* lat_w_dev, double count, double* out_dev, int c, int h, int w,
                     cm_stream stream) {
  if (c <= 0 || h <= 0 || w <= 0 || count <= 0 || !moments || !lat_w_dev || !out_dev) return -22;
  eval_finalize_kernel<<<c, 256, 0, (hipStream_t)stream>>>(moments, lat_w_dev, count, out_dev, h, w);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
