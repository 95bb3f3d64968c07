// GroupNorm(groups) + ReLU of the tolerance-meeting prediction modes (north_star: "GroupNorm/BN"; the reference's normalisation lives in
// smp, unet.py:33-61): the fp32 mode's planar tensors (precise_f32.hip) and the split-precision mode's fp16 word pairs (split16.hip).
//
// GroupNorm statistics are per (sample, group) and nothing folds into the operators, so a stage conv writes its RAW output and three
// launches follow (all HBM-bound streaming passes):
//   partial   one workgroup per (segment, sample, 8-channel plane | channel): sum and sum of squares in double -> slab rows
//   finalize  one workgroup per (sample, group): the rows of its channels in fixed order -> mean, 1 / sqrt(var + eps) (double) ->
//             per (sample, channel) scale = rstd * gamma, shift = beta - mean * rstd * gamma (fp32)
//   apply     y = relu(x * scale + shift), written in the consumer's format (split words: hi + lo planes, range flag)
// The biased variance E[x^2] - mean^2 is formed in double from fp32-exact inputs (a split value hi + lo has 22 bits): 1e-13 relative,
// against F.group_norm's own fp32 arithmetic in oracle/unet_ref.py.
#include "common.h"
#include <algorithm>

namespace {

constexpr int GN_SEG = 8192;                // voxels per partial workgroup

__host__ __device__ inline int gn_parts(long long vox) {
  long long p = (vox + GN_SEG - 1) / GN_SEG;
  return (int)(p < 1 ? 1 : p > 512 ? 512 : p);
}

// slab: double [N][C][parts][2]
// FMT 1: split fp16 words, plane = 8 channels [vox][8], lo planes x_lo planes further on; values are act_scale x activation
template <int FMT>
__global__ __launch_bounds__(256) void gn_partial_kernel(const void* __restrict__ xv, long long x_ss, int x_lo, float inv_act, int C,
                                                        long long vox, int parts, double* __restrict__ slab) {
  const int part = blockIdx.x, n = blockIdx.z, tid = threadIdx.x;
  const long long seg = (vox + parts - 1) / parts;
  const long long v0 = (long long)part * seg, v1 = min(v0 + seg, vox);
  __shared__ double red[256 * 2];
  if constexpr (FMT == 1) {
    const int pl = blockIdx.y;                           // 8-channel plane
    const f16* xh = (const f16*)xv + n * x_ss + (long long)pl * vox * 8;
    const f16* xl = xh + (long long)x_lo * vox * 8;
    double s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.0; q[j] = 0.0; }
    for (long long v = v0 + tid; v < v1; v += 256) {
      const f16x8 h = *(const f16x8*)(xh + v * 8), l = *(const f16x8*)(xl + v * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const double t = (double)(((float)h[j] + (float)l[j]) * inv_act);          // (the sum is exact in fp32, the power of two too)
        s[j] += t;
        q[j] += t * t;
      }
    }
    for (int j = 0; j < 8; ++j) {
      red[tid] = s[j]; red[256 + tid] = q[j];
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[tid] += red[tid + o]; red[256 + tid] += red[256 + tid + o]; }
        __syncthreads();
      }
      if (tid == 0) {
        double* d = slab + (((long long)n * C + pl * 8 + j) * parts + part) * 2;
        d[0] = red[0]; d[1] = red[256];
      }
      __syncthreads();
    }
  } else {
    const int c = blockIdx.y;
    const float* x = (const float*)xv + n * x_ss + (long long)c * vox;
    double s = 0.0, q = 0.0;
    for (long long v = v0 + tid; v < v1; v += 256) { const double t = (double)x[v]; s += t; q += t * t; }
    red[tid] = s; red[256 + tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) { red[tid] += red[tid + o]; red[256 + tid] += red[256 + tid + o]; }
      __syncthreads();
    }
    if (tid == 0) {
      double* d = slab + (((long long)n * C + c) * parts + part) * 2;
      d[0] = red[0]; d[1] = red[256];
    }
  }
}

// out_scale: the factor the stored output carries (act_scale of the split mode, 1 for fp32): shift is pre-multiplied by it
__global__ __launch_bounds__(64) void gn_finalize_kernel(const double* __restrict__ slab, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int C, int groups, long long vox, int parts,
                                                        float eps, float out_scale, float* __restrict__ scale, float* __restrict__ shift) {
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / groups;
  __shared__ double cs[64], cq[64];
  __shared__ double mean_s, rstd_s;
  for (int c0 = 0; c0 < cpg; c0 += 64) {                  // (cpg <= 64 for every network here; the loop keeps the kernel general)
    const int c = c0 + tid;
    double s = 0.0, q = 0.0;
    if (c < cpg) {
      const double* d = slab + ((long long)n * C + g * cpg + c) * parts * 2;
      for (int p = 0; p < parts; ++p) { s += d[2 * p]; q += d[2 * p + 1]; }
    }
    if (c0 == 0) { cs[tid] = s; cq[tid] = q; } else { cs[tid] += s; cq[tid] += q; }
    __syncthreads();
  }
  if (tid == 0) {
    double s = 0.0, q = 0.0;
    for (int c = 0; c < min(cpg, 64); ++c) { s += cs[c]; q += cq[c]; }
    const double cnt = (double)cpg * (double)vox;
    const double m = s / cnt;
    double var = q / cnt - m * m;
    if (var < 0.0) var = 0.0;
    mean_s = m;
    rstd_s = 1.0 / sqrt(var + (double)eps);
  }
  __syncthreads();
  for (int c = tid; c < cpg; c += 64) {
    const int ch = g * cpg + c;
    const double a = rstd_s * (double)gamma[ch];
    scale[(long long)n * C + ch] = (float)a;
    shift[(long long)n * C + ch] = (float)(((double)beta[ch] - mean_s * a) * (double)out_scale);
  }
}

// FMT 1 with y8 != null: the x2m form of the output (conv3_x2m.hip) -- hi planes + lo8 planes (e4m3 of the fp16 rounding residual x 16, the
// words x2m_split8 makes: what every x2m producer stores), and lo planes only where y_lo >= 0
template <int FMT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const void* __restrict__ xv, long long x_ss, int x_lo, void* __restrict__ yv,
                                                      long long y_ss, int y_lo, unsigned char* __restrict__ y8, long long y8_ss,
                                                      const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int C, long long vox, int* __restrict__ sat) {
  const int n = blockIdx.z;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if constexpr (FMT == 1) {
    const int pl = blockIdx.y;
    if (i >= vox) return;
    const f16* xh = (const f16*)xv + n * x_ss + (long long)pl * vox * 8 + i * 8;
    const f16x8 h = *(const f16x8*)xh, l = *(const f16x8*)(xh + (long long)x_lo * vox * 8);
    const float* sc = scale + (long long)n * C + pl * 8;
    const float* sh = shift + (long long)n * C + pl * 8;
    f16x8 oh, ol;
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = fmaxf(fmaf((float)h[j] + (float)l[j], sc[j], sh[j]), 0.f);
    f16* yh = (f16*)yv + n * y_ss + (long long)pl * vox * 8 + i * 8;
    if (y8 != nullptr) {
      u32x2_t l8, h8;
      x2m_split8(r, oh, ol, l8, h8);
      *(u32x2_t*)(y8 + n * y8_ss + x2m_off(pl, i, vox)) = l8;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f16 a, b;
        split16<f16>(r[j], a, b);
        oh[j] = a; ol[j] = b;
      }
    }
    *(f16x8*)yh = oh;
    if (y_lo >= 0) *(f16x8*)(yh + (long long)y_lo * vox * 8) = ol;
    if (sat != nullptr) x2_note_saturation(sat, oh);
  } else {
    const int c = blockIdx.y;
    if (i >= vox) return;
    const float v = ((const float*)xv)[n * x_ss + (long long)c * vox + i];
    ((float*)yv)[n * y_ss + (long long)c * vox + i] = fmaxf(fmaf(v, scale[(long long)n * C + c], shift[(long long)n * C + c]), 0.f);
  }
}

template <int FMT>
int gn_launch(const char* what, const void* x, long long x_ss, int x_lo, void* y, long long y_ss, int y_lo, void* y8, long long y8_ss,
              const void* gamma, const void* beta, int groups, float eps, float act_scale, void* slab, void* scale, void* shift, int C, int N,
              long long vox, void* sat, void* stream) {
  IUNET_REQUIRE(x && y && gamma && beta && slab && scale && shift, "%s: null pointer", what);
  IUNET_REQUIRE(N > 0 && vox > 0 && C > 0 && groups > 0 && C % groups == 0 && (FMT == 0 || C % 8 == 0),
                "%s: %d channels in %d groups, %d samples of %lld voxels", what, C, groups, N, vox);
  IUNET_REQUIRE(N <= 65535 && C <= 65535, "%s: grid limits (N %d, C %d)", what, N, C);
  IUNET_REQUIRE(act_scale > 0.f, "%s: act_scale %g", what, (double)act_scale);
  hipStream_t s = (hipStream_t)stream;
  const int parts = gn_parts(vox);
  const int ny = FMT == 1 ? C / 8 : C;
  hipLaunchKernelGGL(gn_partial_kernel<FMT>, dim3(parts, ny, N), dim3(256), 0, s, x, x_ss, x_lo, 1.0f / act_scale, C, vox, parts, (double*)slab);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, N), dim3(64), 0, s, (const double*)slab, (const float*)gamma, (const float*)beta, C, groups,
                     vox, parts, eps, act_scale, (float*)scale, (float*)shift);
  // the apply reads act_scale x value and its scale carries rstd * gamma alone: (A v) * sc + A * sh = A * (v * sc + sh)
  hipLaunchKernelGGL(gn_apply_kernel<FMT>, dim3((unsigned)((vox + 255) / 256), ny, N), dim3(256), 0, s, x, x_ss, x_lo, y, y_ss, y_lo,
                     (unsigned char*)y8, y8_ss, (const float*)scale, (const float*)shift, C, vox, (int*)sat);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

}  // namespace

extern "C" {

// bytes of the statistics slab (double [N][C][parts][2]) of the two launches below
long long iunet_gn_precise_slab_bytes(int N, int C, long long vox) {
  if (N <= 0 || C <= 0 || vox <= 0) return 0;
  return (long long)N * C * gn_parts(vox) * 2 * (long long)sizeof(double);
}

// fp32 mode: y = relu(group_norm(x)) on planar fp32 tensors [N][C][vox] (sample strides x_ss / y_ss in elements: y may be a half of a
// concat buffer); scale / shift: fp32 [N][C] scratch that receives the per-sample affine pair.
int iunet_f32_gn_relu_fwd(const void* x, long long x_ss, void* y, long long y_ss, const void* gamma, const void* beta, int groups, float eps,
                          void* slab, void* scale, void* shift, int C, int N, long long vox, void* stream) {
  return gn_launch<0>("f32_gn_relu_fwd", x, x_ss, 0, y, y_ss, 0, nullptr, 0, gamma, beta, groups, eps, 1.0f, slab, scale, shift, C, N, vox, nullptr, stream);
}

// split precision: x / y = C / 8 hi planes + lo planes x_lo / y_lo planes further on (split16.hip's layout), values act_scale x
// activation in both; sat: the optional range flag.
int iunet_x2_gn_relu_fwd(const void* x, long long x_ss, int x_lo, void* y, long long y_ss, int y_lo, const void* gamma, const void* beta,
                         int groups, float eps, float act_scale, void* slab, void* scale, void* shift, int C, int N, long long vox,
                         void* sat, void* stream) {
  IUNET_REQUIRE(x_lo > 0 && y_lo > 0, "x2_gn_relu_fwd: lo plane offsets %d / %d", x_lo, y_lo);
  return gn_launch<1>("x2_gn_relu_fwd", x, x_ss, x_lo, y, y_ss, y_lo, nullptr, 0, gamma, beta, groups, eps, act_scale, slab, scale, shift, C, N, vox, sat, stream);
}

// the same with the output in the x2m form (conv3_x2m.hip): hi planes at y (y_ss elements per sample), lo8 planes at y8 (y8_ss BYTES per sample;
// C / 16 planes of [vox][16 B]), and lo planes only where y_lo >= 0 (a tensor a transposed conv or the head reads) -- what the x2m stage convs
// take: the GroupNorm network in the default prediction mode's fast form.  The input is the raw conv output as hi + lo planes.
int iunet_x2m_gn_relu_fwd(const void* x, long long x_ss, int x_lo, void* y, long long y_ss, int y_lo, void* y8, long long y8_ss, const void* gamma,
                          const void* beta, int groups, float eps, float act_scale, void* slab, void* scale, void* shift, int C, int N, long long vox,
                          void* sat, void* stream) {
  IUNET_REQUIRE(x_lo > 0 && y8 != nullptr && C % 16 == 0, "x2m_gn_relu_fwd: lo plane offset %d, lo8 planes %p, %d channels", x_lo, y8, C);
  return gn_launch<1>("x2m_gn_relu_fwd", x, x_ss, x_lo, y, y_ss, y_lo, y8, y8_ss, gamma, beta, groups, eps, act_scale, slab, scale, shift, C, N, vox, sat, stream);
}

}  // extern "C"
