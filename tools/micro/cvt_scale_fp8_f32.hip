// v_cvt_scalef32_pk_fp8_f32 (scale 2^-4) against r * 16 -> clamp(+-448) -> v_cvt_pk_fp8_f32, the lo8 byte of x2m_split8 (common.h), over
// ALL 2^32 f32 bit patterns: mismatches among the finite ones, and among those with |r| <= 28 (the conversions agree wherever r * 16 does
// not saturate).   hipcc --offload-arch=gfx950 -O2 tools/micro/cvt_scale_fp8_f32.hip -o /tmp/cvt32 && /tmp/cvt32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned long long* cnt) {
  const unsigned long long base = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 256;
  unsigned long long bad_all = 0, bad_rng = 0, n_rng = 0, bad_hi = 0;
  for (int i = 0; i < 256; ++i) {
    const unsigned bits = (unsigned)(base + i);
    const float r = __builtin_bit_cast(float, bits);
    if (((bits >> 23) & 255) == 255) continue;
    const float c = __builtin_amdgcn_fmed3f(r * 16.0f, -448.0f, 448.0f);
    const unsigned ref = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(c, 0.f, 0, false) & 0xff;
    s2 z = s2{0, 0};
    const s2 a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, r, 0.f, 0.0625f, false);
    const s2 b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, 0.f, r, 0.0625f, true);
    const unsigned va = __builtin_bit_cast(unsigned, a) & 0xff, vb = (__builtin_bit_cast(unsigned, b) >> 24) & 0xff;
    bad_all += va != ref;
    bad_hi += vb != ref;
    if (fabsf(r) <= 28.0f) { ++n_rng; bad_rng += va != ref; }
  }
  atomicAdd(cnt, bad_all); atomicAdd(cnt + 1, bad_rng); atomicAdd(cnt + 2, n_rng); atomicAdd(cnt + 3, bad_hi);
}
int main() {
  unsigned long long *d, h[4] = {0, 0, 0, 0};
  (void)hipMalloc(&d, 32); (void)hipMemset(d, 0, 32);
  hipLaunchKernelGGL(k, dim3(65536), dim3(256), 0, 0, d);
  (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  printf("all finite f32 patterns: %llu differ (low-byte form), %llu (high-byte form); |r| <= 28: %llu of %llu differ\n", h[0], h[3], h[1], h[2]);
  return 0;
}
