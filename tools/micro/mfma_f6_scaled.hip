// The K = 128 matrix instruction of gfx950 with FP6 (e2m3) operands and e8m0 block scales (v_mfma_scale_f32_16x16x128_f8f6f4, cbsz = blgp = 2):
// (1) operand packing -- 32 six-bit values per lane as one 192-bit little-endian stream in the first six operand registers, lane map k = 32
// (l >> 4) + j as for fp8 -- checked with exact small values against a host sum; (2) the scale operands (one e8m0 byte per lane and
// operand, 127 = 1.0) checked the same way; (3) the rate on random operands beside the e4m3 form.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f6_scaled tools/micro/mfma_f6_scaled.hip && /tmp/mfma_f6_scaled
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// e2m3: sign, 2 exponent bits (bias 1), 3 mantissa bits: 0, 0.125 .. 0.875 (subnormal), 1 .. 1.875, 2 .. 3.75, 4 .. 7.5
__host__ __device__ inline float e2m3_value(unsigned c) {
  const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
  const float v = e == 0 ? m * 0.125f : (1.0f + m * 0.125f) * (float)(1 << (e - 1));
  return s ? -v : v;
}

// MODE 0: e4m3 operands, scale operands constant 0 (the non-scaled form); 1: e2m3 operands with scale registers; 2: e2m3, constant 0 scales;
// 3: e4m3 with scale registers; 4: e2m3 A (weights) x e4m3 B; 5: fp4 (e2m1) with constant 0 scales
template <int MODE>
__global__ __launch_bounds__(256) void rate(float* out, unsigned long long* cyc, int iters, const unsigned* rnd) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  i32x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (int)(rnd[(threadIdx.x * 8 + j) & 4095] & 0x7f7f7f7fu & ~0x40404040u);
                                b[j] = (int)(rnd[(threadIdx.x * 8 + j + 2048) & 4095] & ~0x40404040u); }
  const int sa = 127 - (int)(threadIdx.x & 3), sb = 127 - (int)((threadIdx.x >> 2) & 3);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0, 0, 0);
      else if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 2, 2, 0, sa, 0, sb);
      else if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 2, 2, 0, 0, 0, 0);
      else if (MODE == 3) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, sa, 0, sb);
      else if (MODE == 4) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 2, 0, 0, 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 4, 4, 0, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// one wave: D = (sa A) (16 x 128) * (sb B) (128 x 16); codes [16][128] / [128][16] of 6 bits each, scales one e8m0 byte per (row, 32-k block)
__global__ void check(const unsigned char* A, const unsigned char* B, const unsigned char* SA, const unsigned char* SB, float* D) {
  const int l = threadIdx.x;
  unsigned long long bitsA[3] = {0, 0, 0}, bitsB[3] = {0, 0, 0};
  for (int j = 0; j < 32; ++j) {
    const int k = 32 * (l >> 4) + j;
    const unsigned long long ca = A[(l & 15) * 128 + k] & 63, cb = B[k * 16 + (l & 15)] & 63;
    const int bit = 6 * j, w = bit >> 6, o = bit & 63;
    bitsA[w] |= ca << o; bitsB[w] |= cb << o;
    if (o > 58) { bitsA[w + 1] |= ca >> (64 - o); bitsB[w + 1] |= cb >> (64 - o); }
  }
  i32x8 a, b;
  for (int w = 0; w < 3; ++w) { a[2 * w] = (int)(bitsA[w] & 0xffffffffu); a[2 * w + 1] = (int)(bitsA[w] >> 32);
                                b[2 * w] = (int)(bitsB[w] & 0xffffffffu); b[2 * w + 1] = (int)(bitsB[w] >> 32); }
  a[6] = a[7] = b[6] = b[7] = 0;
  const int sa = SA[(l & 15) * 4 + (l >> 4)], sb = SB[(l >> 4) * 16 + (l & 15)];
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, sa, 0, sb);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

int main() {
  unsigned char hA[16 * 128], hB[128 * 16], hSA[64], hSB[64], *dA, *dB, *dSA, *dSB; float hD[256], *dD;
  srand(1);
  for (int i = 0; i < 16 * 128; ++i) { hA[i] = rand() & 63; hB[i] = rand() & 63; }
  for (int i = 0; i < 64; ++i) { hSA[i] = 127 + rand() % 5 - 2; hSB[i] = 127 + rand() % 5 - 2; }
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dSA, 64); hipMalloc(&dSB, 64); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipMemcpy(dSA, hSA, 64, hipMemcpyHostToDevice); hipMemcpy(dSB, hSB, 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
    double s = 0;
    for (int k = 0; k < 128; ++k)
      s += (double)e2m3_value(hA[r * 128 + k]) * ldexp(1.0, hSA[r * 4 + k / 32] - 127) * (double)e2m3_value(hB[k * 16 + c]) * ldexp(1.0, hSB[(k / 32) * 16 + c] - 127);
    if ((float)s != hD[r * 16 + c]) { if (bad < 5) printf("mismatch D[%d][%d] = %.9g, expected %.9g\n", r, c, hD[r * 16 + c], s); ++bad; }
  }
  printf("16x16x128 f8f6f4 with e2m3 operands (192-bit stream, k = 32 (l >> 4) + j) and e8m0 scales per (row / column, 32-k block): %s (%d of 256 wrong)\n",
         bad ? "WRONG" : "exact", bad);
  float* out; unsigned long long* cyc; unsigned* rnd; unsigned hr[4096];
  for (int i = 0; i < 4096; ++i) hr[i] = (unsigned)rand() * 2654435761u;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&rnd, sizeof(hr));
  hipMemcpy(rnd, hr, sizeof(hr), hipMemcpyHostToDevice);
  const int iters = 40000;
  const char* names[6] = {"e4m3, unit scales", "e2m3, scale registers", "e2m3, unit scales", "e4m3, scale registers", "e2m3 A x e4m3 B, unit scales", "e2m1 (fp4), unit scales"};
  for (int mode = 0; mode < 6; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters, rnd);
      else if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters, rnd);
      else if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters, rnd);
      else if (mode == 3) hipLaunchKernelGGL(rate<3>, dim3(256), dim3(256), 0, 0, out, cyc, iters, rnd);
      else if (mode == 4) hipLaunchKernelGGL(rate<4>, dim3(256), dim3(256), 0, 0, out, cyc, iters, rnd);
      else hipLaunchKernelGGL(rate<5>, dim3(256), dim3(256), 0, 0, out, cyc, iters, rnd);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      printf("16x16x128 %s: %.2f memtime-ticks per MFMA (one wave per SIMD), %.3f ms, %.1f TFLOP/s\n", names[mode], (double)h[0] / ((double)iters * 8), ms,
             2.0 * 16 * 16 * 128 * iters * 8 * 4 * 256 / ms / 1e9);
    }
  }
  return 0;
}
