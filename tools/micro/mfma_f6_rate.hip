// Issue rate of v_mfma_f32_16x16x128_f8f6f4 by operand format, with the instruction written in inline assembly (through the builtin the
// fp6 / fp4 forms drew ~50 accumulator copies per loop iteration from the compiler: tools/micro/mfma_f6_scaled.hip measures those, not
// the instruction).  8 independent accumulators per wave, one wave per SIMD, random operands.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f6_rate tools/micro/mfma_f6_rate.hip && /tmp/mfma_f6_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(6))) int i32x6;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>      // 0 e4m3 x e4m3, 1 e2m3 x e2m3, 2 e2m1 x e2m1, 3 e2m3 x e2m3 with scale registers
__global__ __launch_bounds__(256) void rate(float* out, unsigned long long* cyc, int iters, const unsigned* rnd) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  i32x8 a8, b8; i32x6 a6, b6; i32x4 a4, b4;
  for (int j = 0; j < 8; ++j) { a8[j] = (int)(rnd[(threadIdx.x * 8 + j) & 4095] & 0x3f3f3f3fu); b8[j] = (int)(rnd[(threadIdx.x * 8 + j + 2048) & 4095] & 0x3f3f3f3fu); }
  for (int j = 0; j < 6; ++j) { a6[j] = (int)(rnd[(threadIdx.x * 6 + j) & 4095] & 0x5d75d75du); b6[j] = (int)(rnd[(threadIdx.x * 6 + j + 2048) & 4095] & 0x5d75d75du); }
  for (int j = 0; j < 4; ++j) { a4[j] = (int)rnd[(threadIdx.x * 4 + j) & 4095]; b4[j] = (int)rnd[(threadIdx.x * 4 + j + 2048) & 4095]; }
  const int sa = 127 - (int)(threadIdx.x & 1), sb = 127 - (int)((threadIdx.x >> 1) & 1);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a8), "v"(b8));
      else if (MODE == 1) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0 cbsz:2 blgp:2" : "+v"(acc[i]) : "v"(a6), "v"(b6));
      else if (MODE == 2) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+v"(acc[i]) : "v"(a4), "v"(b4));
      else asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2" : "+v"(acc[i]) : "v"(a6), "v"(b6), "v"(sa), "v"(sb));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc; unsigned* rnd; unsigned hr[4096];
  srand(2);
  for (int i = 0; i < 4096; ++i) hr[i] = (unsigned)rand() * 2654435761u;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8); (void)hipMalloc(&rnd, sizeof(hr));
  (void)hipMemcpy(rnd, hr, sizeof(hr), hipMemcpyHostToDevice);
  const int iters = 40000;
  const char* names[4] = {"e4m3 x e4m3", "e2m3 x e2m3 (fp6)", "e2m1 x e2m1 (fp4)", "e2m3 x e2m3 with e8m0 scale registers"};
  for (int mode = 0; mode < 4; ++mode) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters, rnd);
      else if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters, rnd);
      else if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters, rnd);
      else hipLaunchKernelGGL(rate<3>, dim3(256), dim3(256), 0, 0, out, cyc, iters, rnd);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      printf("16x16x128 f8f6f4 %s: %.2f memtime-ticks per MFMA (one wave per SIMD), %.3f ms, %.1f TFLOP/s\n", names[mode], (double)h[0] / ((double)iters * 8), ms,
             2.0 * 16 * 16 * 128 * iters * 8 * 4 * 256 / ms / 1e9);
    }
  }
  return 0;
}
