// Cycles per MFMA: v_mfma_f32_16x16x32_bf16 (K = 32) against the older v_mfma_f32_16x16x16_bf16 (K = 16) on gfx950.
// hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float seed) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed * (threadIdx.x % 7 + j)); b[j] = (__bf16)(seed * (threadIdx.x % 5 - j)); }
  s16x4 a4 = {(short)(threadIdx.x * 3), (short)(threadIdx.x * 5), (short)(threadIdx.x * 7), 11};
  s16x4 b4 = {(short)(threadIdx.x * 13), (short)(threadIdx.x * 17), 3, 5};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a4), "v"(b4));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 20000;
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters, 0.01f);
      else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters, 0.01f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      const double per = (double)h[0] / ((double)iters * 8);
      const double flop = (mode == 0 ? 16384.0 : 8192.0) * iters * 8 * 4 * 256;
      printf("%s: %.2f memtime-ticks per MFMA (one wave per SIMD), %.3f ms, %.1f TFLOP/s\n", mode == 0 ? "16x16x32_bf16" : "16x16x16_bf16",
             per, ms, flop / ms / 1e9);
    }
  }
  return 0;
}
