// Minimal reproducer attempt for the round-1 "stale uniform load under concurrent queues" finding.
//   writer : buf[i] = val
//   reader : every wave reads buf[c .. c+7] at a wave-uniform address (the compiler merges it into 16-byte loads) for
//            all c and compares with `expect`; mismatches are counted
//   spinner: long-running kernel with a large LDS footprint and MFMA work, launched on a second stream
// Host loop: spinner on stream B, then on stream A repeatedly {writer(it), filler kernels, reader(it)}.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__global__ void writer(float* buf, int n, float val) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = val + (float)(i & 7);
}

__global__ void reader(const float* __restrict__ buf, int rows, int C, float expect, const float* __restrict__ big,
                       int hw, unsigned* bad) {
  // same shape as gate_bwd_reduce: 4 channel slices x 64 pixels, s-like row read uniformly, a2-like tensor per lane
  const int n = blockIdx.y, lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + lane;
  const float* sp = buf + (long long)n * C;
  const int cper = C / 4, c0 = slice * cper, c1 = c0 + cper;
  float acc = 0.f;
  unsigned wrong = 0;
#pragma unroll 8
  for (int c = c0; c < c1; ++c) {
    const float sv = sp[c];
    if (sv != expect + (float)(((long long)n * C + c) & 7)) ++wrong;
    acc += big[((long long)n * C + c) * hw + (p < hw ? p : 0)] * sv;
  }
  if (wrong) atomicAdd(bad, wrong);
  if (acc == 12345.678f) atomicAdd(bad, 1u);
}

__global__ __launch_bounds__(192) void spinner(const float* __restrict__ x, float* out, int iters) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < 19000; i += blockDim.x) lds[i] = 0x7fc07fc0u;   // bf16 NaN patterns, ~76 KB
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  bf16x8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = (short)0x3f80; b[k] = (short)0x3f80; }
  for (int it = 0; it < iters; ++it) {
    const float v = x[(blockIdx.x * 192 + threadIdx.x + it * 4096) & 0xfffff];
    a[0] = (short)(__float_as_uint(v) >> 16);
    for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    lds[(threadIdx.x + it) % 19000] = __float_as_uint(acc[0]);
  }
  if (acc[3] == 1.2345f) out[0] = acc[3];
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 200;
  const int N = 192, C = 256, HW = 54;
  float *buf, *big, *x, *out;
  unsigned* bad;
  hipMalloc(&buf, N * C * 4);
  hipMalloc(&big, (size_t)N * C * HW * 4);
  hipMalloc(&x, (1 << 20) * 4);
  hipMalloc(&out, 4);
  hipMalloc(&bad, 4);
  hipMemset(big, 0, (size_t)N * C * HW * 4);
  hipMemset(x, 0, (1 << 20) * 4);
  hipMemset(bad, 0, 4);
  hipStream_t sa, sb;
  hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  hipFuncSetAttribute((const void*)spinner, hipFuncAttributeMaxDynamicSharedMemorySize, 77 * 1024);
  for (int mode = 0; mode < 2; ++mode) {        // 0: serial, 1: spinner concurrently on stream B
    hipMemset(bad, 0, 4);
    hipDeviceSynchronize();
    for (int it = 1; it <= steps; ++it) {
      if (mode == 1) spinner<<<512, 192, 77 * 1024, sb>>>(x, out, 300);
      writer<<<64, 256, 0, sa>>>(buf, N * C, (float)it);
      for (int f = 0; f < 6; ++f) writer<<<256, 256, 0, sa>>>(big, N * C * HW, (float)f);   // filler traffic
      reader<<<dim3(1, N), 256, 0, sa>>>(buf, N, C, (float)it, big, HW, bad);
      reader<<<dim3(1, N), 256, 0, sa>>>(buf, N, C, (float)it, big, HW, bad);
    }
    hipDeviceSynchronize();
    unsigned h = 0;
    hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("mode %d (%s): %u stale / wrong uniform reads in %d steps\n", mode, mode ? "spinner on a second stream" : "serial",
           h, steps);
  }
  return 0;
}
