// MaxPool2d(2) of DownPoolEnc and the time-mean skip aggregation.
//
// Reference: nn.MaxPool2d(2) (src/unet_convlstm_attention.py:21,25) and torch.stack(all_s, 0).mean(0)
// (src/unet_convlstm_attention.py:91-93).  Encoder tensors are [B*T, C, H, W] with sample n = b*T + t (the frame
// loop is folded into the batch), so the time mean of sample b reads T consecutive samples.
// Backward: max_pool2d_with_indices_backward routes the gradient to the FIRST maximal element in row-major window
// order (golden fixture maxpool.npz); the mean's backward is a broadcast of dskip/T, fused here.
#include "common.h"
#include "../../include/climate_hip.h"

namespace {

__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes, int H,
                                    int W) {
  const int Ho = H / 2, Wo = W / 2;
  const long long total = planes * Ho * Wo;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int xo = (int)(i % Wo);
    const long long t = i / Wo;
    const int yo = (int)(t % Ho);
    const long long pl = t / Ho;
    const float* p = x + pl * H * W + (long long)(2 * yo) * W + 2 * xo;
    const float2 r0 = *reinterpret_cast<const float2*>(p);
    const float2 r1 = *reinterpret_cast<const float2*>(p + W);
    y[i] = fmaxf(fmaxf(r0.x, r0.y), fmaxf(r1.x, r1.y));
  }
}

// dx[n,c,2yo+i,2xo+j] = (first argmax ? dy[n,c,yo,xo] : 0) + scale * dskip[n / T, c, 2yo+i, 2xo+j]
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                    const float* __restrict__ dskip, long long sds, float* __restrict__ dx, int N,
                                    int C, int H, int W, int T, float scale) {
  const int Ho = H / 2, Wo = W / 2;
  const long long total = (long long)N * C * Ho * Wo;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int xo = (int)(i % Wo);
    long long t = i / Wo;
    const int yo = (int)(t % Ho);
    t /= Ho;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    const long long off = ((long long)n * C + c) * H * W + (long long)(2 * yo) * W + 2 * xo;
    const float2 r0 = *reinterpret_cast<const float2*>(x + off);
    const float2 r1 = *reinterpret_cast<const float2*>(x + off + W);
    int arg = 0;
    float m = r0.x;
    if (r0.y > m) { m = r0.y; arg = 1; }
    if (r1.x > m) { m = r1.x; arg = 2; }
    if (r1.y > m) { m = r1.y; arg = 3; }
    const float g = dy[i];
    float2 o0 = make_float2(arg == 0 ? g : 0.f, arg == 1 ? g : 0.f);
    float2 o1 = make_float2(arg == 2 ? g : 0.f, arg == 3 ? g : 0.f);
    if (dskip) {
      const long long so = (long long)(n / T) * sds + (long long)c * H * W + (long long)(2 * yo) * W + 2 * xo;
      const float2 s0 = *reinterpret_cast<const float2*>(dskip + so);
      const float2 s1 = *reinterpret_cast<const float2*>(dskip + so + W);
      o0.x += scale * s0.x; o0.y += scale * s0.y; o1.x += scale * s1.x; o1.y += scale * s1.y;
    }
    *reinterpret_cast<float2*>(dx + off) = o0;
    *reinterpret_cast<float2*>(dx + off + W) = o1;
  }
}

// y[b, i] = (1/T) sum_t x[(b*T + t), i]
__global__ void time_mean_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int T, long long chw) {
  const long long total = (long long)B * chw;
  const float inv = 1.f / (float)T;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / chw, r = i % chw;
    const float* p = x + b * T * chw + r;
    float a = 0.f;
    for (int t = 0; t < T; ++t) a += p[(long long)t * chw];
    y[i] = a * inv;
  }
}

// same, 4 elements per thread (chw % 4 == 0, 16-byte aligned bases): one 16-byte load per frame
__global__ void time_mean_vec4_kernel(const float4* __restrict__ x, float4* __restrict__ y, int B, int T,
                                      unsigned chw4) {
  const unsigned total = (unsigned)B * chw4;
  const float inv = 1.f / (float)T;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned b = i / chw4, r = i - b * chw4;
    const float4* p = x + (size_t)b * T * chw4 + r;
    float4 a = p[0];
    for (int t = 1; t < T; ++t) {
      const float4 v = p[(size_t)t * chw4];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    y[i] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
  }
}

// per-channel sum over samples and pixels: out[c] += sum_{n,p} x[n, c, p]   (bias gradients)
// (n, p) is walked as one flat index so that small images still fill the workgroup.
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, long long st,
                                                           float* __restrict__ out, int N, int HW, int nsplit) {
  __shared__ float red[32];
  const int c = blockIdx.x, sp = blockIdx.y;
  const long long total = (long long)N * HW;
  const float* xc = x + (long long)c * HW;
  float a = 0.f;
  for (long long i = sp * 256 + threadIdx.x; i < total; i += (long long)nsplit * 256) {
    const long long n = i / HW;
    const int p = (int)(i - n * HW);
    a += xc[n * st + p];
  }
  a = block_sum(a, red);
  if (threadIdx.x == 0) unsafeAtomicAdd(out + c, a);
}

inline int grid_for(long long total, int bs) {
  long long b = (total + bs - 1) / bs;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int cm_maxpool2_fwd(const float* x, float* y, long long planes, int h, int w, cm_stream stream) {
  if (planes <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 1)) return -22;
  maxpool2_fwd_kernel<<<grid_for(planes * (h / 2) * (w / 2), 256), 256, 0, (hipStream_t)stream>>>(x, y, planes, h, w);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_maxpool2_bwd(const float* x, const float* dy, const float* dskip, long long st_dskip, float* dx, int n, int c,
                    int h, int w, int t, float scale, cm_stream stream) {
  if (n <= 0 || c <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 1) || t <= 0) return -22;
  maxpool2_bwd_kernel<<<grid_for((long long)n * c * (h / 2) * (w / 2), 256), 256, 0, (hipStream_t)stream>>>(
      x, dy, dskip, st_dskip, dx, n, c, h, w, t, scale);
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_time_mean(const float* x, float* y, int b, int t, long long chw, cm_stream stream) {
  if (b <= 0 || t <= 0 || chw <= 0) return -22;
  if (chw % 4 == 0 && (long long)b * t * chw < (1ll << 33) && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
    const long long total4 = (long long)b * (chw / 4);
    const long long blocks = (total4 + 255) / 256;
    time_mean_vec4_kernel<<<(unsigned)(blocks > 8192 ? 8192 : blocks), 256, 0, (hipStream_t)stream>>>(
        (const float4*)x, (float4*)y, b, t, (unsigned)(chw / 4));
  } else {
    time_mean_kernel<<<grid_for((long long)b * chw, 256), 256, 0, (hipStream_t)stream>>>(x, y, b, t, chw);
  }
  CM_CHECK_LAUNCH();
  return 0;
}

int cm_channel_sum(const float* x, long long st, float* out, int n, int c, int hw, cm_stream stream) {
  if (n <= 0 || c <= 0 || hw <= 0) return -22;
  int nsplit = 1;
  while (c * nsplit < 1024 && (long long)nsplit * 256 * 8 < (long long)n * hw) nsplit *= 2;
  channel_sum_kernel<<<dim3(c, nsplit), 256, 0, (hipStream_t)stream>>>(x, st, out, n, hw, nsplit);
  CM_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
