// Device side of the reference's whole-volume prediction (predict.py:201-256): the
// float32 accumulators `pred` / `weight` live in HBM instead of in temporary Zarr
// arrays on disk.  All kernels are HBM-bound byte / float32 streaming work.
#include "common.h"
#include <algorithm>

namespace {

// predict.py:291-316 (np.pad mode='reflect' of the clipped block), as a gather.
// lo/hi = clipped extent, p0 = padded block start; rel = (p0 + i) - lo reflected into
// [0, n) with period 2n-2.
__device__ __forceinline__ int reflect_rel(int rel, int n) {
  if (n == 1) return 0;
  const int period = 2 * n - 2;
  int a = rel < 0 ? -rel : rel;
  a %= period;
  return a < n ? a : period - a;
}

__global__ __launch_bounds__(256) void gather_block_kernel(const unsigned char* __restrict__ vol, int Vz, int Vy, int Vx,
                                                           int i0, int j0, int k0, int S,
                                                           unsigned char* __restrict__ out) {
  const long long total = (long long)S * S * S;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % S), y = (int)((i / S) % S), z = (int)(i / ((long long)S * S));
  const int lz = max(i0, 0), ly = max(j0, 0), lx = max(k0, 0);
  const int hz = min(i0 + S, Vz), hy = min(j0 + S, Vy), hx = min(k0 + S, Vx);
  const int sz = lz + reflect_rel(i0 + z - lz, hz - lz);
  const int sy = ly + reflect_rel(j0 + y - ly, hy - ly);
  const int sx = lx + reflect_rel(k0 + x - lx, hx - lx);
  out[i] = vol[((long long)sz * Vy + sy) * Vx + sx];
}

// predict.py:244-245: pred[blk] += P[local] * win[local][..., None]; weight[blk] += win[local]
// P is [S,S,S,C] float32 (block orientation), pred [Vz,Vy,Vx,C], weight [Vz,Vy,Vx].
// Product and sum are rounded separately (no fma) to match numpy bit for bit.
__global__ __launch_bounds__(256) void blend_kernel(float* __restrict__ pred, float* __restrict__ weight,
                                                    const float* __restrict__ P, const float* __restrict__ win,
                                                    int Vy, int Vx, int C, int S, int b0z, int b0y, int b0x, int l0z,
                                                    int l0y, int l0x, int ez, int ey, int ex) {
  const long long total = (long long)ez * ey * ex;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % ex), y = (int)((i / ex) % ey), z = (int)(i / ((long long)ex * ey));
  const long long lidx = ((long long)(l0z + z) * S + (l0y + y)) * S + (l0x + x);
  const long long vidx = ((long long)(b0z + z) * Vy + (b0y + y)) * Vx + (b0x + x);
  const float w = win[lidx];
  weight[vidx] = __fadd_rn(weight[vidx], w);
  for (int c = 0; c < C; ++c)
    pred[vidx * C + c] = __fadd_rn(pred[vidx * C + c], __fmul_rn(P[lidx * C + c], w));
}

// predict.py:255: uint8(255 * pred / max(weight, eps)) -- truncating cast.
__global__ __launch_bounds__(256) void normalize_quantize_kernel(const float* __restrict__ pred,
                                                                 const float* __restrict__ weight,
                                                                 unsigned char* __restrict__ out, long long nvox,
                                                                 int C, float eps) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvox * C) return;
  const float w = fmaxf(weight[i / C], eps);
  const float v = __fdiv_rn(__fmul_rn(255.0f, pred[i]), w);
  out[i] = (unsigned char)(int)v;     // values are in [0, 255]: truncation toward zero
}

__global__ __launch_bounds__(256) void div_kernel(float* __restrict__ p, long long n, float d) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = __fdiv_rn(p[i], d);
}

// class map -> palette colours (predict.py:41-45: one-hot * 255 -> utils.convert_categorical_to_color): 3 bytes per pixel
__global__ __launch_bounds__(256) void colorize_kernel(const unsigned char* __restrict__ cls, long long n,
                                                      unsigned char* __restrict__ out,
                                                      const unsigned char* __restrict__ palette, int ncls) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = cls[i];
  unsigned char r = 0, g = 0, b = 0;
  if (c < ncls) { r = palette[c * 3]; g = palette[c * 3 + 1]; b = palette[c * 3 + 2]; }
  out[i * 3] = r; out[i * 3 + 1] = g; out[i * 3 + 2] = b;
}

// Calibration figure of the default prediction mode (engine_auto.py): out[0] = max |a - b|, out[1] = max |a| over two fp32 logit
// tensors of one shape.  Non-negative floats order like their bit patterns, so the reduction is an integer atomicMax (exact,
// order-independent); NaN patterns sort above every finite value and therefore surface as a NaN result.
__global__ __launch_bounds__(256) void logit_diff_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                        unsigned int* __restrict__ out) {
  float d = 0.f, m = 0.f;
  unsigned int dn = 0u, mn = 0u;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float x = a[i], y = b[i];
    const float e = fabsf(x - y), s = fabsf(x);
    if (e != e) dn = 0x7fc00000u;
    if (s != s) mn = 0x7fc00000u;
    d = fmaxf(d, e);
    m = fmaxf(m, s);
  }
  unsigned int du = max(__float_as_uint(d), dn), mu = max(__float_as_uint(m), mn);
  for (int o = 32; o > 0; o >>= 1) {
    du = max(du, (unsigned int)__shfl_xor((int)du, o));
    mu = max(mu, (unsigned int)__shfl_xor((int)mu, o));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(out, du);
    atomicMax(out + 1, mu);
  }
}

}  // namespace

extern "C" {

// max |a - b| and max |a| of two fp32 tensors of n elements -> out[0], out[1] (device floats, overwritten).  The selection rule of the
// default prediction mode compares the logits of the x2m and the fp16x2 forward of one calibration tile with it.
int iunet_logit_diff(const void* a, const void* b, long long n, void* out2, void* stream) {
  IUNET_REQUIRE(a && b && out2 && n > 0, "logit_diff: null pointer or empty tensor");
  IUNET_CHECK_HIP(hipMemsetAsync(out2, 0, 8, (hipStream_t)stream));
  const unsigned grid = (unsigned)std::min<long long>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(logit_diff_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b, n,
                     (unsigned int*)out2);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

// predict.get_padded_block (predict.py:291-316) on a device-resident uint8 volume.
int iunet_gather_block(const void* vol, int Vz, int Vy, int Vx, int i0, int j0, int k0, int S, void* out, void* stream) {
  IUNET_REQUIRE(vol && out, "gather_block: null pointer");
  IUNET_REQUIRE(S > 0 && Vz > 0 && Vy > 0 && Vx > 0, "gather_block: bad shape");
  IUNET_REQUIRE(i0 < Vz && j0 < Vy && k0 < Vx && i0 + S > 0 && j0 + S > 0 && k0 + S > 0,
                "gather_block: block does not intersect the volume");
  const long long total = (long long)S * S * S;
  hipLaunchKernelGGL(gather_block_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)vol, Vz, Vy, Vx, i0, j0, k0, S, (unsigned char*)out);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

// blend-accumulate of predict.py:244-245.  block = clipped volume coords [6], local = coords inside the block [6].
int iunet_blend_accumulate(void* pred, void* weight, const void* P, const void* window, int Vz, int Vy, int Vx, int C,
                           int S, const int* block, const int* local, void* stream) {
  IUNET_REQUIRE(pred && weight && P && window && block && local, "blend: null pointer");
  IUNET_REQUIRE(C > 0 && S > 0, "blend: bad class count %d / block size %d", C, S);
  const int ez = block[3] - block[0], ey = block[4] - block[1], ex = block[5] - block[2];
  IUNET_REQUIRE(ez > 0 && ey > 0 && ex > 0, "blend: empty block");
  IUNET_REQUIRE(block[0] >= 0 && block[1] >= 0 && block[2] >= 0 && block[3] <= Vz && block[4] <= Vy && block[5] <= Vx,
                "blend: block outside the volume");
  IUNET_REQUIRE(local[0] >= 0 && local[1] >= 0 && local[2] >= 0 && local[0] + ez <= S && local[1] + ey <= S &&
                local[2] + ex <= S, "blend: local coords outside the block");
  const long long total = (long long)ez * ey * ex;
  hipLaunchKernelGGL(blend_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (float*)pred, (float*)weight, (const float*)P, (const float*)window, Vy, Vx, C, S, block[0],
                     block[1], block[2], local[0], local[1], local[2], ez, ey, ex);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

// normalise + quantise of predict.py:252-256.
int iunet_normalize_quantize(const void* pred, const void* weight, void* out_u8, long long nvox, int C, float eps,
                             void* stream) {
  IUNET_REQUIRE(pred && weight && out_u8, "normalize_quantize: null pointer");
  IUNET_REQUIRE(nvox > 0 && C > 0, "normalize_quantize: %lld voxels, %d classes", nvox, C);
  const long long total = nvox * C;
  hipLaunchKernelGGL(normalize_quantize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const float*)pred, (const float*)weight, (unsigned char*)out_u8, nvox, C, eps);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_div_f32(void* p, long long n, float d, void* stream) {
  IUNET_REQUIRE(p && n > 0, "div_f32: null pointer or empty range");
  hipLaunchKernelGGL(div_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float*)p, n, d);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

// out[i] = palette[cls[i]] (uint8 [n][3]); classes >= ncls are black.  palette: uint8 [ncls][3] on the device.
int iunet_colorize(const void* cls, long long n, const void* palette, int ncls, void* out_rgb, void* stream) {
  IUNET_REQUIRE(cls && palette && out_rgb && n >= 0 && ncls > 0, "colorize: bad arguments");
  if (n == 0) return IUNET_OK;
  hipLaunchKernelGGL(colorize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)cls, n, (unsigned char*)out_rgb, (const unsigned char*)palette, ncls);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

}  // extern "C"
