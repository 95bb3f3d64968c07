// The K = 128 fp8 matrix instruction of gfx950 (v_mfma[_scale]_f32_16x16x128_f8f6f4, e4m3 operands, unit scales) against the K = 32
// form (v_mfma_f32_16x16x32_fp8_fp8): operand lane map (exact-integer check against a host sum) and rate on random operands.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f8_scaled tools/micro/mfma_f8_scaled.hip && /tmp/mfma_f8_scaled
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef long i64;

// e4m3 byte of a small integer value (|v| <= 15 exactly representable up to 16 in steps of 1 below 16)
__host__ __device__ inline unsigned char e4m3_of_int(int v) {
  unsigned char s = v < 0 ? 0x80 : 0;
  int a = v < 0 ? -v : v;
  if (a == 0) return s;
  int e = 0;
  while ((a >> (e + 1)) != 0) ++e;                    // a in [2^e, 2^(e+1))
  const int mant = ((a << 3) >> e) & 7;               // 3 mantissa bits (exact for a < 16)
  return s | (unsigned char)(((e + 7) << 3) | mant);
}

// MODE 0: 16x16x32 fp8 (K = 32), 1: 16x16x128 f8f6f4 (K = 128), 2: 32x32x64 f8f6f4 (K = 64)
template <int MODE>
__global__ __launch_bounds__(256) void rate(float* out, unsigned long long* cyc, int iters, const unsigned* rnd) {
  f32x4 acc[8];
  f32x16 acc32[4];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
  i32x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (int)(rnd[(threadIdx.x * 8 + j) & 4095] & 0x7f7f7f7fu & ~0x40404040u);      // |v| < 2: no overflow
                                b[j] = (int)(rnd[(threadIdx.x * 8 + j + 2048) & 4095] & ~0x40404040u); }
  const i64 a8 = ((i64)a[0] << 32) | (unsigned)a[1], b8 = ((i64)b[0] << 32) | (unsigned)b[1];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a8, b8, acc[i], 0, 0, 0);
      else if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0, 0, 0);
      else acc32[i & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc32[i & 3], 0, 0, 0, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc32[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// one wave: D = A (16 x 128) * B (128 x 16) with small integers; assumed map: lane l holds A[l & 15][32 (l >> 4) + j], B[32 (l >> 4) + j][l & 15]
__global__ void check(const unsigned char* A, const unsigned char* B, float* D) {
  const int l = threadIdx.x;
  i32x8 a, b;
  for (int w = 0; w < 8; ++w) {
    unsigned va = 0, vb = 0;
    for (int j = 0; j < 4; ++j) {
      const int k = 32 * (l >> 4) + 4 * w + j;
      va |= (unsigned)A[(l & 15) * 128 + k] << (8 * j);
      vb |= (unsigned)B[k * 16 + (l & 15)] << (8 * j);
    }
    a[w] = (int)va; b[w] = (int)vb;
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];       // row = 4 (l >> 4) + r, col = l & 15
}

int main() {
  // ---- lane map
  unsigned char hA[16 * 128], hB[128 * 16], *dA, *dB; float hD[256], *dD;
  int iA[16 * 128], iB[128 * 16];
  srand(1);
  for (int i = 0; i < 16 * 128; ++i) { iA[i] = rand() % 15 - 7; hA[i] = e4m3_of_int(iA[i]); iB[i] = rand() % 13 - 6; hB[i] = e4m3_of_int(iB[i]); }
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
    long s = 0;
    for (int k = 0; k < 128; ++k) s += (long)iA[r * 128 + k] * iB[k * 16 + c];
    if ((float)s != hD[r * 16 + c]) { if (bad < 5) printf("mismatch D[%d][%d] = %g, expected %ld\n", r, c, hD[r * 16 + c], s); ++bad; }
  }
  printf("16x16x128 f8f6f4 (e4m3, scale operands 0): lane map k = 32 (l >> 4) + j: %s (%d of 256 wrong)\n", bad ? "WRONG" : "exact", bad);

  // ---- rate
  float* out; unsigned long long* cyc; unsigned* rnd; unsigned hr[4096];
  for (int i = 0; i < 4096; ++i) hr[i] = (unsigned)rand() * 2654435761u;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&rnd, sizeof(hr));
  hipMemcpy(rnd, hr, sizeof(hr), hipMemcpyHostToDevice);
  const int iters = 40000;
  const char* names[3] = {"16x16x32_fp8_fp8", "16x16x128_f8f6f4", "32x32x64_f8f6f4"};
  const double flop_per[3] = {2.0 * 16 * 16 * 32, 2.0 * 16 * 16 * 128, 2.0 * 32 * 32 * 64};
  for (int mode = 0; mode < 3; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters, rnd);
      else if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters, rnd);
      else hipLaunchKernelGGL(rate<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters, rnd);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      const double per = (double)h[0] / ((double)iters * 8);
      const double flop = flop_per[mode] * iters * 8 * 4 * 256;
      printf("%s: %.2f memtime-ticks per MFMA (one wave per SIMD), %.3f ms, %.1f TFLOP/s\n", names[mode], per, ms, flop / ms / 1e9);
    }
  }
  return 0;
}
