// v_cvt_scalef32_pk_fp8_f16 against the three-instruction path it can replace in x2m_hi8 (common.h): for every finite f16 bit pattern h,
// e4m3(h * 2^-8) by (float)h * 2^-8 -> clamp -> v_cvt_pk_fp8_f32 (the production path) and by the packed scaled conversion with scale 2^8
// (and, to pin the scale's direction, 2^-8).   hipcc --offload-arch=gfx950 -O2 tools/micro/cvt_scale_fp8_f16.hip -o /tmp/cvt && /tmp/cvt
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));

__global__ void k(unsigned char* ref, unsigned char* up, unsigned char* down, unsigned char* hiword) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;          // f16 bit pattern
  const unsigned short bits = (unsigned short)i;
  const f16 h = __builtin_bit_cast(f16, bits);
  const float c = __builtin_amdgcn_fmed3f((float)h * 0.00390625f, -448.0f, 448.0f);
  ref[i] = (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(c, 0.f, 0, false) & 0xff);
  const h2 v = h2{h, (f16)0};
  s2 z = s2{0, 0};
  s2 a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(z, v, 256.0f, false);
  s2 b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(z, v, 0.00390625f, false);
  s2 w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(z, h2{(f16)0, h}, 256.0f, true);      // value in element 1, result into the high word
  up[i] = (unsigned char)(__builtin_bit_cast(unsigned, a) & 0xff);
  down[i] = (unsigned char)(__builtin_bit_cast(unsigned, b) & 0xff);
  hiword[i] = (unsigned char)((__builtin_bit_cast(unsigned, w) >> 24) & 0xff);
}

int main() {
  unsigned char *d, h[4][65536];
  hipMalloc(&d, 4 * 65536);
  hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d, d + 65536, d + 2 * 65536, d + 3 * 65536);
  hipMemcpy(h, d, 4 * 65536, hipMemcpyDeviceToHost);
  int neq_up = 0, neq_down = 0, neq_hi = 0, n = 0, first = -1;
  for (int i = 0; i < 65536; ++i) {
    if (((i >> 10) & 31) == 31) continue;                     // inf / nan
    ++n;
    if (h[0][i] != h[1][i]) { if (first < 0) first = i; ++neq_up; }
    neq_down += h[0][i] != h[2][i];
    neq_hi += h[0][i] != h[3][i];
  }
  printf("%d finite f16 patterns: scale 2^8 differs from the reference on %d, scale 2^-8 on %d, the high-word form (scale 2^8) on %d\n", n, neq_up, neq_down, neq_hi);
  if (first >= 0) printf("first difference at pattern 0x%04x: reference 0x%02x, scaled 0x%02x\n", first, h[0][first], h[1][first]);
  printf("samples (pattern: ref / 2^8 / 2^-8): 0x3c00 (1.0): %02x %02x %02x; 0x5c00 (256): %02x %02x %02x; 0x7bff (65504): %02x %02x %02x; 0x1400: %02x %02x %02x\n",
         h[0][0x3c00], h[1][0x3c00], h[2][0x3c00], h[0][0x5c00], h[1][0x5c00], h[2][0x5c00], h[0][0x7bff], h[1][0x7bff], h[2][0x7bff],
         h[0][0x1400], h[1][0x1400], h[2][0x1400]);
  return 0;
}
