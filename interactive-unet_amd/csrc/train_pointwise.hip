// Training-side HBM-bound kernels: BatchNorm batch statistics / apply / backward, the
// fused 1x1 head + softmax + weighted soft-confusion loss (metrics.py) forward and
// backward, max-pool backward, slab reductions, AdamW.  All reductions go through
// per-block partial slabs summed in a fixed order (deterministic, no float atomics).
#include "common.h"

namespace {

template <typename T> using V8T = typename Vec8<T>::type;

__device__ __forceinline__ float block_sum_256(float v, float* red /* [4] */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// Block reduction of NV per-thread values with ONE barrier: wave shuffles, then 4 wave
// partials through LDS (lds: [4][NV] floats), thread i < NV writes out[i].  Fixed order.
template <int NV>
__device__ __forceinline__ void block_reduce_store(const float (&vals)[NV], float* lds, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float v = wave_sum(vals[i]);
    if (lane == 0) lds[wave * NV + i] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NV; i += 256) out[i] = lds[i] + lds[NV + i] + lds[2 * NV + i] + lds[3 * NV + i];
}

// ------------------------------------------------------------------ BN statistics -> scale/shift
// slab [nparts][C][2] (sum, sumsq of the raw conv output).  One block per channel.
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ slab, int nparts, int C, double count,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* running_mean, float* running_var, float momentum,
                                                           float eps, float* scale, float* shift, float* mean_o,
                                                           float* invstd_o) {
  // 256 threads, or 1 024 for slabs of thousands of rows (the first conv writes one row per TILE: 8 192 at 2 x 128^3, 16 us on 256 threads)
  const int c = blockIdx.x, nt = (int)blockDim.x, nw = nt >> 6;
  double s = 0.0, s2 = 0.0;
  for (int p = threadIdx.x; p < nparts; p += nt) {
    const float2 v = *(const float2*)(slab + ((long long)p * C + c) * 2);
    s += (double)v.x; s2 += (double)v.y;
  }
  // wave sums by lane exchange, the waves meet in LDS: one barrier instead of the nine of a 256-wide LDS tree (these
  // kernels are a few microseconds of pure latency between two convolutions, 28 of them per training step)
  __shared__ double red[2][16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); s2 += __shfl_xor(s2, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    // (256 threads: ((w0 + w1) + (w2 + w3)), the order this kernel has always had; 1 024: the same tree over four such groups)
    double g0[4], g1[4];
    for (int g = 0; g < nw / 4; ++g) {
      g0[g] = (red[0][4 * g] + red[0][4 * g + 1]) + (red[0][4 * g + 2] + red[0][4 * g + 3]);
      g1[g] = (red[1][4 * g] + red[1][4 * g + 1]) + (red[1][4 * g + 2] + red[1][4 * g + 3]);
    }
    const double t0 = nw == 4 ? g0[0] : (g0[0] + g0[1]) + (g0[2] + g0[3]), t1 = nw == 4 ? g1[0] : (g1[0] + g1[1]) + (g1[2] + g1[3]);
    const double mean = t0 / count;
    double var = t1 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    mean_o[c] = (float)mean;
    invstd_o[c] = invstd;
    if (running_mean) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  }
}

// z = relu(scale[c] * y + shift[c]), blocked layout.  The channel plane is a grid dimension, so the
// per-channel constants are wave-uniform (scalar loads); two 16-byte items per thread in flight.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(const T* __restrict__ y, long long y_ss, T* __restrict__ z,
                                                          long long z_ss, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int planes, long long vox,
                                                          int pss = 0 /* per-sample stride of scale / shift (GroupNorm: C; BatchNorm: 0) */) {
  const int pl = blockIdx.y, n = blockIdx.z;
  const long long v0 = (long long)blockIdx.x * 512 + threadIdx.x;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = scale[n * pss + pl * 8 + j]; sh[j] = shift[n * pss + pl * 8 + j]; }
  const long long base = (long long)pl * vox * 8;
  V8T<T> in[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const long long v = v0 + u * 256;
    if (v < vox) in[u] = *(const V8T<T>*)(y + n * y_ss + base + v * 8);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const long long v = v0 + u * 256;
    if (v >= vox) continue;
    V8T<T> o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = from_f32<T>(fmaxf(fmaf(sc[j], to_f32<T>(in[u][j]), sh[j]), 0.f));
    *(V8T<T>*)(z + n * z_ss + base + v * 8) = o;
  }
}

// z = relu(scale[c] * y + shift[c]) AND its 2^d max-pool in one pass (encoder stages: the pool would re-read z right
// away).  One thread per pooled voxel and channel plane: 2^d reads of y, 2^d writes of z, one write of the pooled value.
template <typename T, int ND>
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const T* __restrict__ y, long long y_ss, T* __restrict__ z,
                                                               long long z_ss, T* __restrict__ pooled, long long p_ss,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               int Do, int Ho, int Wo, int pss = 0 /* per-sample stride of scale / shift (GroupNorm: C) */) {
  const long long ovox = (long long)Do * Ho * Wo;
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= ovox) return;
  const int pl = blockIdx.y, n = blockIdx.z;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = scale[n * pss + pl * 8 + j]; sh[j] = shift[n * pss + pl * 8 + j]; }
  const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), oz = (int)(r / ((long long)Wo * Ho));
  const int Di = ND == 3 ? Do * 2 : 1, Hi = Ho * 2, Wi = Wo * 2;
  const long long ipl = (long long)pl * Di * Hi * Wi * 8;
  constexpr int NW = ND == 3 ? 8 : 4;
  V8T<T> win[NW];
  long long off[NW];
#pragma unroll
  for (int s = 0; s < NW; ++s) {
    const int a = ND == 3 ? (s >> 2) : 0, b = (s >> 1) & 1, c = s & 1;
    const int zz = ND == 3 ? oz * 2 + a : 0;
    off[s] = ipl + (((long long)zz * Hi + oy * 2 + b) * Wi + ox * 2 + c) * 8;
    win[s] = *(const V8T<T>*)(y + n * y_ss + off[s]);
  }
  float m[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
  for (int s = 0; s < NW; ++s) {
    V8T<T> o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o[j] = from_f32<T>(fmaxf(fmaf(sc[j], to_f32<T>(win[s][j]), sh[j]), 0.f));     // = bn_relu_fwd_kernel
      m[j] = fmaxf(m[j], to_f32<T>(o[j]));                                            // = maxpool_kernel on the stored value
    }
    *(V8T<T>*)(z + n * z_ss + off[s]) = o;
  }
  V8T<T> o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = from_f32<T>(m[j]);
  *(V8T<T>*)(pooled + n * p_ss + (long long)pl * ovox * 8 + r * 8) = o;
}

// BN+ReLU backward, pass 1: per-channel s1 = sum(dyh), s2 = sum(dyh * xhat), dyh = dz * (z > 0).
// grid (chunks, planes, N); slab [(n*chunks + chunk)][C][2]
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dz, long long dz_ss,
                                                            const T* __restrict__ z, long long z_ss,
                                                            const T* __restrict__ y, long long y_ss,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            int C, long long vox, int per_block, float* __restrict__ slab,
                                                            int pss = 0 /* per-sample stride of mean / invstd / scale / shift (GroupNorm: C) */) {
  const int pl = blockIdx.y, n = blockIdx.z;
  const long long v0 = (long long)blockIdx.x * per_block;
  const long long v1 = min(v0 + per_block, vox);
  // the ReLU mask (z > 0) is recomputed from y when z is not given: one tensor read less per pass
  float s1[8], s2[8], mu[8], is[8], sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = n * pss + pl * 8 + j;
    s1[j] = 0.f; s2[j] = 0.f; mu[j] = mean[c]; is[j] = invstd[c];
    sc[j] = scale[c]; sh[j] = shift[c];
  }
  const long long po = (long long)pl * vox * 8;
#pragma unroll 4
  for (long long v = v0 + threadIdx.x; v < v1; v += 256) {        // unrolled: 8 x 16 B in flight per thread
    const V8T<T> g = *(const V8T<T>*)(dz + n * dz_ss + po + v * 8);
    const V8T<T> yy = *(const V8T<T>*)(y + n * y_ss + po + v * 8);
    V8T<T> zz;
    if (z) zz = *(const V8T<T>*)(z + n * z_ss + po + v * 8);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) zz[j] = from_f32<T>(fmaf(sc[j], to_f32<T>(yy[j]), sh[j]));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = to_f32<T>(zz[j]) > 0.f ? to_f32<T>(g[j]) : 0.f;
      s1[j] += d;
      s2[j] += d * (to_f32<T>(yy[j]) - mu[j]) * is[j];
    }
  }
  __shared__ float red[4 * 16];
  const long long part = (long long)n * gridDim.x + blockIdx.x;
  float vals[16];
#pragma unroll
  for (int j = 0; j < 8; ++j) { vals[2 * j] = s1[j]; vals[2 * j + 1] = s2[j]; }
  block_reduce_store<16>(vals, red, slab + (part * C + pl * 8) * 2);
}

// BatchNorm + ReLU backward of an encoder stage's second conv with the max-pool backward folded in: the gradient of the
// stage output is dz = dskip + route(dpool) (dskip = the decoder's gradient of the skip connection, route = to the first
// maximum of each 2^d window of z = relu(bn(y)), recomputed from y) and is never written.  One thread per pooled voxel
// and channel plane.  PASS 1: per-channel sums (as bn_bwd_reduce_kernel), PASS 2: dy (as bn_bwd_apply_kernel).
// Every value is rounded where the three-kernel sequence maxpool_bwd -> reduce -> apply rounds it.
template <typename T, int ND, int PASS, bool GN = false>
__global__ __launch_bounds__(256) void bn_pool_bwd_kernel(const T* __restrict__ dskip, long long ds_ss,
                                                          const T* __restrict__ dpool, long long dp_ss,
                                                          const T* __restrict__ y, long long y_ss, T* __restrict__ dy,
                                                          long long dy_ss, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ coef,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          int C, int Do, int Ho, int Wo, int per_block, float* __restrict__ slab,
                                                          int pss = 0 /* per-sample stride of the per-channel parameters (GroupNorm: C) */) {
  constexpr int NW = ND == 3 ? 8 : 4;
  const int pl = blockIdx.y, n = blockIdx.z;
  const long long ovox = (long long)Do * Ho * Wo;
  const int Di = ND == 3 ? Do * 2 : 1, Hi = Ho * 2, Wi = Wo * 2;
  const long long ipl = (long long)pl * Di * Hi * Wi * 8;
  float mu[8], is[8], sc[8], sh[8], ca[8], c1[8], c2[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = n * pss + pl * 8 + j;
    mu[j] = mean[c]; is[j] = invstd[c]; sc[j] = scale[c]; sh[j] = shift[c]; s1[j] = 0.f; s2[j] = 0.f;
    if (PASS == 2) { ca[j] = coef[c * 3]; c1[j] = coef[c * 3 + 1]; c2[j] = coef[c * 3 + 2]; }
  }
  const long long r0 = (long long)blockIdx.x * per_block, r1 = min(r0 + per_block, ovox);
  for (long long r = r0 + threadIdx.x; r < r1; r += 256) {
    const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), oz = (int)(r / ((long long)Wo * Ho));
    V8T<T> yy[NW], gs[NW];
    long long off[NW];
#pragma unroll
    for (int s = 0; s < NW; ++s) {
      const int a = ND == 3 ? (s >> 2) : 0, b = (s >> 1) & 1, c = s & 1;
      const int zz = ND == 3 ? oz * 2 + a : 0;
      off[s] = ipl + (((long long)zz * Hi + oy * 2 + b) * Wi + ox * 2 + c) * 8;
      yy[s] = *(const V8T<T>*)(y + n * y_ss + off[s]);
      gs[s] = *(const V8T<T>*)(dskip + n * ds_ss + off[s]);
    }
    const V8T<T> gp = *(const V8T<T>*)(dpool + n * dp_ss + (long long)pl * ovox * 8 + r * 8);
    int best[8];
    float zv[NW][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float m = 0.f;
#pragma unroll
      for (int s = 0; s < NW; ++s) {
        zv[s][j] = to_f32<T>(from_f32<T>(fmaxf(fmaf(sc[j], to_f32<T>(yy[s][j]), sh[j]), 0.f)));     // z as bn_relu_fwd stored it
        if (s == 0) { m = zv[0][j]; best[j] = 0; } else if (zv[s][j] > m) { m = zv[s][j]; best[j] = s; }   // first maximum
      }
    }
#pragma unroll
    for (int s = 0; s < NW; ++s) {
      V8T<T> o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // dz as maxpool_bwd_kernel (add_skip) stored it, then the ReLU mask
        const float dz = to_f32<T>(from_f32<T>(to_f32<T>(gs[s][j]) + (best[j] == s ? to_f32<T>(gp[j]) : 0.f)));
        const float d = zv[s][j] > 0.f ? dz : 0.f;
        const float xh = (to_f32<T>(yy[s][j]) - mu[j]) * is[j];
        if (PASS == 1) { s1[j] += d; s2[j] += d * xh; }
        else o[j] = from_f32<T>(GN ? ca[j] * d - c1[j] - xh * c2[j] : ca[j] * (d - c1[j] - xh * c2[j]));      // (GN: bn_bwd_apply_kernel's GroupNorm form)
      }
      if (PASS == 2) *(V8T<T>*)(dy + n * dy_ss + off[s]) = o;
    }
  }
  if (PASS == 1) {
    __shared__ float red[4 * 16];
    const long long part = (long long)n * gridDim.x + blockIdx.x;
    float vals[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) { vals[2 * j] = s1[j]; vals[2 * j + 1] = s2[j]; }
    block_reduce_store<16>(vals, red, slab + (part * C + pl * 8) * 2);
  }
}

// pass 1b: dgamma = s2, dbeta = s1 (times grad_unscale), coefficients for pass 2
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ slab, int nparts, int C, double count,
                                                              const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                              float* dgamma, float* dbeta, float* coef /* [C][3]: a, c1, c2 */) {
  const int c = blockIdx.x;
  double s = 0.0, s2 = 0.0;
  for (int p = threadIdx.x; p < nparts; p += 256) {
    s += (double)slab[((long long)p * C + c) * 2];
    s2 += (double)slab[((long long)p * C + c) * 2 + 1];
  }
  // wave sums by lane exchange, the four waves meet in LDS: one barrier instead of the nine of a 256-wide LDS tree (these
  // kernels are a few microseconds of pure latency between two convolutions, 28 of them per training step)
  __shared__ double red[2][4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); s2 += __shfl_xor(s2, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    red[0][0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    red[1][0] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
  if (threadIdx.x == 0) {
    dbeta[c] = (float)red[0][0];
    dgamma[c] = (float)red[1][0];
    coef[c * 3 + 0] = gamma[c] * invstd[c];
    coef[c * 3 + 1] = (float)(red[0][0] / count);
    coef[c * 3 + 2] = (float)(red[1][0] / count);
  }
}

// pass 2: dy = a * (dyh - c1 - xhat * c2); plane = grid dimension (wave-uniform constants), two items per thread
// GN (GroupNorm form): coef = (gamma invstd, invstd m1, invstd m2) and dy = a dz' - b1 - xhat b2 -- the statistics of a GROUP depend on
// every channel of it, so a channel with gamma = 0 still has the gradient -invstd (m1 + xhat m2), which the BatchNorm factorisation
// a (dz' - c1 - xhat c2) (a = gamma invstd) cannot express.
template <typename T, bool GN = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dz, long long dz_ss, const T* __restrict__ z,
                                                           long long z_ss, const T* __restrict__ y, long long y_ss,
                                                           T* __restrict__ dy, long long dy_ss, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ coef,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           int planes, long long vox,
                                                           int pss = 0 /* per-sample stride of the per-channel parameters (GroupNorm: C) */) {
  const int pl = blockIdx.y, n = blockIdx.z;
  const long long v0 = (long long)blockIdx.x * 512 + threadIdx.x;
  float mu[8], is[8], ca[8], c1[8], c2[8], sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = n * pss + pl * 8 + j;
    mu[j] = mean[c]; is[j] = invstd[c]; ca[j] = coef[c * 3]; c1[j] = coef[c * 3 + 1]; c2[j] = coef[c * 3 + 2];
    sc[j] = scale[c]; sh[j] = shift[c];
  }
  const long long base = (long long)pl * vox * 8;
  V8T<T> g[2], yy[2], zz[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const long long v = v0 + u * 256;
    if (v < vox) {
      g[u] = *(const V8T<T>*)(dz + n * dz_ss + base + v * 8);
      yy[u] = *(const V8T<T>*)(y + n * y_ss + base + v * 8);
      if (z) zz[u] = *(const V8T<T>*)(z + n * z_ss + base + v * 8);
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const long long v = v0 + u * 256;
    if (v >= vox) continue;
    V8T<T> o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float yv = to_f32<T>(yy[u][j]);
      const float zv = z ? to_f32<T>(zz[u][j]) : to_f32<T>(from_f32<T>(fmaf(sc[j], yv, sh[j])));
      const float d = zv > 0.f ? to_f32<T>(g[u][j]) : 0.f;
      const float xh = (yv - mu[j]) * is[j];
      o[j] = from_f32<T>(GN ? ca[j] * d - c1[j] - xh * c2[j] : ca[j] * (d - c1[j] - xh * c2[j]));
    }
    *(V8T<T>*)(dy + n * dy_ss + base + v * 8) = o;
  }
}

// ------------------------------------------------------------------ GroupNorm (north_star "GroupNorm/BN"; SURVEY 8d: GroupNorm(8) variant)
// Statistics are per (sample, group) -- the same at training and at inference, nothing to fold.  Everything per-channel of
// the BatchNorm kernels above becomes per (sample, channel): the passes over the tensors are the BatchNorm ones launched
// per sample with that sample's rows of scale / shift / mean / invstd / coef; only the small finalize steps are new.
// pass 1 of the forward: per-(sample, channel) sum and sum of squares of the raw conv output; grid (chunks, planes, N);
// slab [(n * chunks + chunk)][C][2]
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ y, long long y_ss, int C, long long vox, int per_block,
                                                       float* __restrict__ slab) {
  const int pl = blockIdx.y, n = blockIdx.z;
  const long long v0 = (long long)blockIdx.x * per_block, v1 = min(v0 + per_block, vox);
  const long long po = (long long)pl * vox * 8;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll 4
  for (long long v = v0 + threadIdx.x; v < v1; v += 256) {
    const V8T<T> yy = *(const V8T<T>*)(y + n * y_ss + po + v * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float a = to_f32<T>(yy[j]); s1[j] += a; s2[j] = fmaf(a, a, s2[j]); }
  }
  __shared__ float red[4 * 16];
  float vals[16];
#pragma unroll
  for (int j = 0; j < 8; ++j) { vals[2 * j] = s1[j]; vals[2 * j + 1] = s2[j]; }
  block_reduce_store<16>(vals, red, slab + (((long long)n * gridDim.x + blockIdx.x) * C + pl * 8) * 2);
}

// one block per (group, sample): mean / biased variance over the group's channels x voxels (double, fixed order) ->
// per-(sample, channel) scale = gamma * invstd, shift = beta - mean * scale, mean, invstd  ([N][C] each)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ slab, int chunks, int C, int groups, double vox,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                         float* scale, float* shift, float* mean_o, float* invstd_o) {
  const int g = blockIdx.x, n = blockIdx.y, cpg = C / groups;
  double s = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < chunks * cpg; i += 256) {
    const int ch = i / cpg, c = g * cpg + i % cpg;
    const float* p = slab + (((long long)n * chunks + ch) * C + c) * 2;
    s += (double)p[0]; s2 += (double)p[1];
  }
  __shared__ double red[2][4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); s2 += __shfl_xor(s2, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    red[0][0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    red[1][0] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
  __syncthreads();
  const double cnt = vox * cpg, mean = red[0][0] / cnt;
  double var = red[1][0] / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  for (int k = threadIdx.x; k < cpg; k += 256) {
    const int c = g * cpg + k;
    const float sc = gamma[c] * invstd;
    scale[n * C + c] = sc;
    shift[n * C + c] = beta[c] - (float)mean * sc;
    mean_o[n * C + c] = (float)mean;
    invstd_o[n * C + c] = invstd;
  }
}

// backward finalize.  slab [(n * chunks + chunk)][C][2] = (s1, s2) of bn_bwd_reduce_kernel run per sample with that sample's
// mean / invstd: s1 = sum dz', s2 = sum dz' * xhat (dz' = dz where relu passed).  One block of 256 threads per GROUP (C <= 1024):
//   dgamma[c] = sum_n s2, dbeta[c] = sum_n s1;
//   per (n, group): m1 = sum_c gamma_c s1 / M, m2 = sum_c gamma_c s2 / M, M = channels per group x voxels;
//   dy = invstd_g (gamma_c dz' - m1 - xhat m2) = a dz' - b1 - xhat b2 with coef[n][c] = (gamma_c invstd, invstd m1, invstd m2)
// The slab rows of a channel are summed by the whole block (double, fixed order: per-thread strided partials, wave shuffles, four wave
// partials) -- [r4] the first version gave every channel ONE thread of ONE block that walked its 1 024 rows per sample alone: 209 us per
// launch on average, 2.9 ms of the 10.8 ms C3 step with GroupNorm.
__global__ __launch_bounds__(1024) void gn_bwd_finalize_kernel(const float* __restrict__ slab, int chunks, int C, int groups, int N,
                                                             double vox, const float* __restrict__ gamma,
                                                             const float* __restrict__ invstd, float* dgamma, float* dbeta,
                                                             float* coef) {
  // [r5] one WAVE per (sample, channel) row sum -- lane-strided partials, one shuffle tree, no block barrier inside -- and a batch of
  // the group's N x cpg sums in flight over the block's sixteen waves; one barrier before the batch's per-sample coefficients.  (The block-wide sum per
  // (sample, channel) with two barriers each took 19.7 us per launch, 14 launches per C3 step.)
  const int g = blockIdx.x, cpg = C / groups, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  __shared__ double sums[2048][2];                          // [sample of the batch][channel of the group]: NB x cpg rows at a time
  const int NB = max(1, 2048 / cpg);                        // (cpg <= 1024: iunet_gn_relu_bwd checks C <= 1024)
  const double M = vox * cpg;
  double acc_a = 0.0, acc_b = 0.0;                          // dbeta / dgamma of channel t (cpg <= 1024 threads): samples added in order
  for (int n0 = 0; n0 < N; n0 += NB) {
    const int nb = min(NB, N - n0);
    for (int i = wave; i < nb * cpg; i += 16) {
      const int n = n0 + i / cpg, c = g * cpg + i % cpg;
      double a = 0.0, b = 0.0;
      for (int ch = lane; ch < chunks; ch += 64) {
        const float2 v = *(const float2*)(slab + (((long long)n * chunks + ch) * C + c) * 2);
        a += (double)v.x; b += (double)v.y;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
      if (lane == 0) { sums[i][0] = a; sums[i][1] = b; }
    }
    __syncthreads();
    for (int i = t; i < nb * cpg; i += 1024) {
      const int nl = i / cpg, n = n0 + nl, c = g * cpg + i % cpg;
      double m1 = 0.0, m2 = 0.0;
      for (int k = 0; k < cpg; ++k) { const double gk = (double)gamma[g * cpg + k]; m1 += gk * sums[nl * cpg + k][0]; m2 += gk * sums[nl * cpg + k][1]; }
      const float is = invstd[n * C + c];
      coef[((long long)n * C + c) * 3 + 0] = gamma[c] * is;
      coef[((long long)n * C + c) * 3 + 1] = is * (float)(m1 / M);
      coef[((long long)n * C + c) * 3 + 2] = is * (float)(m2 / M);
    }
    if (t < cpg) for (int nl = 0; nl < nb; ++nl) { acc_a += sums[nl * cpg + t][0]; acc_b += sums[nl * cpg + t][1]; }
    __syncthreads();                                        // sums is rewritten for the next batch of samples
  }
  if (t < cpg) { dbeta[g * cpg + t] = (float)acc_a; dgamma[g * cpg + t] = (float)acc_b; }
}

// ------------------------------------------------------------------ max-pool backward (+ skip gradient)
// dz[v] = (dskip ? dskip[v] : 0) + (v is the FIRST maximum of its window ? dpool : 0), in place on dskip.
template <typename T, int ND>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ z, long long z_ss, const T* __restrict__ dpool,
                                                          long long dp_ss, T* __restrict__ dz, long long dz_ss, int add_skip,
                                                          int planes, int Do, int Ho, int Wo) {
  const long long ovox = (long long)Do * Ho * Wo;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= ovox * planes) return;
  const int n = blockIdx.y;
  const int pl = (int)(i / ovox);
  const long long r = i - (long long)pl * ovox;
  const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), oz = (int)(r / ((long long)Wo * Ho));
  const int Di = ND == 3 ? Do * 2 : 1, Hi = Ho * 2, Wi = Wo * 2;
  const long long ipl = (long long)pl * Di * Hi * Wi * 8;
  constexpr int NW = ND == 3 ? 8 : 4;
  V8T<T> win[NW];
  long long off[NW];
#pragma unroll
  for (int s = 0; s < NW; ++s) {
    const int a = ND == 3 ? (s >> 2) : 0, b = (s >> 1) & 1, c = s & 1;
    const int zz = ND == 3 ? oz * 2 + a : 0;
    off[s] = ipl + (((long long)zz * Hi + oy * 2 + b) * Wi + ox * 2 + c) * 8;
    win[s] = *(const V8T<T>*)(z + n * z_ss + off[s]);
  }
  const V8T<T> g = *(const V8T<T>*)(dpool + n * dp_ss + (long long)pl * ovox * 8 + r * 8);
  int best[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float m = to_f32<T>(win[0][j]); best[j] = 0;
#pragma unroll
    for (int s = 1; s < NW; ++s) { const float v = to_f32<T>(win[s][j]); if (v > m) { m = v; best[j] = s; } }
  }
#pragma unroll
  for (int s = 0; s < NW; ++s) {
    V8T<T> o;
    if (add_skip) o = *(const V8T<T>*)(dz + n * dz_ss + off[s]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float base = add_skip ? to_f32<T>(o[j]) : 0.f;
      o[j] = from_f32<T>(base + (best[j] == s ? to_f32<T>(g[j]) : 0.f));
    }
    *(V8T<T>*)(dz + n * dz_ss + off[s]) = o;
  }
}

// ------------------------------------------------------------------ fused head + softmax + loss
// targets / weights: [N][ncls][vox] contiguous, f16 (tdtype 1) or f32 (tdtype 0).
__device__ __forceinline__ float load_t(const void* p, long long off, int dt) {
  return dt == 0 ? ((const float*)p)[off] : (float)((const f16*)p)[off];
}

struct HeadLossParams {
  const void* x; long long x_ss; int planes;
  const float* w; const float* bias;
  const void* target; const void* weight; int tdtype;
  float* slab;          // fwd: [nblocks][ncls][8]
  const float* coef;    // bwd: [ncls][3] (A, B, CE)
  void* dx; long long dx_ss;   // bwd: gradient wrt head input (T, blocked)
  float* dwslab;        // bwd: [nblocks][ncls*(C0+1)]
  float loss_scale;
  const float* loss_scale_dev;   // non-null: the loss scale lives on the device (the training handle's state), read here
  int N; long long vox;
  // optional [C0] pair: the head input is relu(in_scale * x + in_shift) rounded to T -- the BatchNorm + ReLU of the last stage conv,
  // applied while loading (training: that activation is read only by the head, so it is never written; bit-identical to reading
  // the tensor bn_relu_fwd_kernel would have stored)
  const float* in_scale; const float* in_shift;
  int pss;              // per-sample stride of in_scale / in_shift (GroupNorm: C0; BatchNorm: 0)
};

template <typename T>
__device__ __forceinline__ V8T<T> head_act(const HeadLossParams& p, V8T<T> v, int pl, int n) {
  if (p.in_scale == nullptr) return v;
  V8T<T> o;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    o[j] = from_f32<T>(fmaxf(fmaf(p.in_scale[n * p.pss + pl * 8 + j], to_f32<T>(v[j]), p.in_shift[n * p.pss + pl * 8 + j]), 0.f));
  return o;
}

#define HEAD_FWD_ITER 8
// sums per class: 0 sw, 1 swy, 2 swp, 3 swyp, 4 swy*log(p+eps), 5 sw*ry, 6 sw*rp, 7 sw*ry*rp
template <typename T, int NCLS>
__global__ __launch_bounds__(256) void head_loss_fwd_kernel(HeadLossParams p) {
  const int n = blockIdx.y;
  float acc[NCLS][8];
#pragma unroll
  for (int c = 0; c < NCLS; ++c)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[c][k] = 0.f;
  // HEAD_FWD_ITER x 256 voxels per workgroup: one block reduction (and one slab row for the finalize pass) per 2048 voxels
  for (int it = 0; it < HEAD_FWD_ITER; ++it) {
    const long long v = ((long long)blockIdx.x * HEAD_FWD_ITER + it) * 256 + threadIdx.x;
    if (v >= p.vox) break;
    const T* xin = (const T*)p.x + n * p.x_ss + v * 8;
    float l[NCLS];
#pragma unroll
    for (int c = 0; c < NCLS; ++c) l[c] = p.bias[c];
    for (int pl = 0; pl < p.planes; ++pl) {
      const V8T<T> xv = head_act<T>(p, *(const V8T<T>*)(xin + (long long)pl * p.vox * 8), pl, n);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float a = to_f32<T>(xv[j]);
#pragma unroll
        for (int c = 0; c < NCLS; ++c) l[c] = fmaf(a, p.w[c * p.planes * 8 + pl * 8 + j], l[c]);
      }
    }
    float mx = l[0];
#pragma unroll
    for (int c = 1; c < NCLS; ++c) mx = fmaxf(mx, l[c]);
    float e[NCLS], s = 0.f;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) { e[c] = __expf(l[c] - mx); s += e[c]; }
    const float inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
      const float pr = e[c] * inv;
      const long long to = ((long long)n * NCLS + c) * p.vox + v;
      const float y = load_t(p.target, to, p.tdtype);
      const float w = p.weight ? load_t(p.weight, to, p.tdtype) : 1.f;
      const float ry = rintf(y), rp = rintf(pr);
      acc[c][0] += w; acc[c][1] += w * y; acc[c][2] += w * pr; acc[c][3] += w * y * pr;
      acc[c][4] += w * y * __logf(pr + 1e-12f);
      acc[c][5] += w * ry; acc[c][6] += w * rp; acc[c][7] += w * ry * rp;
    }
  }
  __shared__ float red[4 * NCLS * 8];
  const long long part = (long long)n * gridDim.x + blockIdx.x;
  float vals[NCLS * 8];
#pragma unroll
  for (int c = 0; c < NCLS; ++c)
#pragma unroll
    for (int k = 0; k < 8; ++k) vals[c * 8 + k] = acc[c][k];
  block_reduce_store<NCLS * 8>(vals, red, p.slab + part * NCLS * 8);
}

// One block: reduce the slab, evaluate the reference loss (metrics.py:3-187 with
// axes = batch + spatial, mean over classes) and the rounded Dice/IoU/MCC of unet.py:75-86,
// and emit per-class gradient coefficients: dL/dp = w*(A + B*y) - CE * w*y/(p+eps).
// kind: 0 ce, 1 dice, 2 iou, 3 mcc, 4 dice+ce, 5 iou+ce, 6 mcc+ce.
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ slab, int nparts, int ncls, int kind,
                                                            int has_weight, double nvox_total, float* out /* [4]: loss, dice, iou, mcc */,
                                                            float* coef /* [ncls][3] */) {
  __shared__ double sums[10][8];
  __shared__ double red[256];
  // one coalesced pass over the slab: thread t owns item (class, k) = t % (ncls*8) of every (256 / items)-th part;
  // fixed summation order -> deterministic
  const int items = ncls * 8, groups = 256 / items, t = threadIdx.x;
  double s = 0.0;
  if (t < groups * items) {
    const long long total = (long long)nparts * items;
    long long e = t;
    for (; e + 3LL * groups * items < total; e += 4LL * groups * items) {
      const float a = slab[e], b = slab[e + (long long)groups * items], c = slab[e + 2LL * groups * items],
                  d = slab[e + 3LL * groups * items];
      s += (double)a; s += (double)b; s += (double)c; s += (double)d;
    }
    for (; e < total; e += (long long)groups * items) s += (double)slab[e];
  }
  red[t] = s;
  __syncthreads();
  if (t < items) {
    double a = 0.0;
    for (int g = 0; g < groups; ++g) a += red[g * items + t];
    sums[t >> 3][t & 7] = a;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double eps = 1e-12;
  const bool use_ce = (kind == 0 || kind >= 4);
  const int base = kind >= 4 ? kind - 3 : kind;      // 0 ce only, 1 dice, 2 iou, 3 mcc
  double loss = 0.0, dice_r = 0.0, iou_r = 0.0, mcc_r = 0.0;
  const double G = (double)ncls;
  for (int c = 0; c < ncls; ++c) {
    const double cnt = has_weight ? sums[c][0] : nvox_total;
    const double tp = sums[c][3] / cnt, fp = (sums[c][2] - sums[c][3]) / cnt, fn = (sums[c][1] - sums[c][3]) / cnt;
    const double tn = (sums[c][0] - sums[c][1] - sums[c][2] + sums[c][3]) / cnt;
    double s_tp = 0, s_tn = 0, s_fp = 0, s_fn = 0, score = 0;
    if (base == 1) {
      const double num = 2 * tp + eps, den = 2 * tp + fp + fn + eps;
      score = num / den; s_tp = (2 * den - 2 * num) / (den * den); s_fp = -num / (den * den); s_fn = s_fp;
    } else if (base == 2) {
      const double num = tp + eps, den = tp + fp + fn + eps;
      score = num / den; s_tp = (den - num) / (den * den); s_fp = -num / (den * den); s_fn = s_fp;
    } else if (base == 3) {
      const double a = tp + fp, b = tp + fn, cc = tn + fp, d = tn + fn;
      const double root = sqrt(a * b * cc * d), num = tp * tn - fp * fn + eps, den = root + eps;
      score = num / den;
      const double half = 0.5 / root;
      const double r_tp = half * (b * cc * d + a * cc * d), r_tn = half * (a * b * d + a * b * cc);
      const double r_fp = half * (b * cc * d + a * b * d), r_fn = half * (a * cc * d + a * b * cc);
      s_tp = (tn * den - num * r_tp) / (den * den); s_tn = (tp * den - num * r_tn) / (den * den);
      s_fp = (-fn * den - num * r_fp) / (den * den); s_fn = (-fp * den - num * r_fn) / (den * den);
    }
    if (base != 0) loss += (1.0 - score) / G;
    if (use_ce) loss += (-sums[c][4] / cnt) / G;
    // dL/dp = -(1/(G cnt)) w [ y (s_tp - s_fn) + (1-y)(s_fp - s_tn) ]
    const double k = -1.0 / (G * cnt);
    coef[c * 3 + 0] = (float)(k * (s_fp - s_tn));
    coef[c * 3 + 1] = (float)(k * ((s_tp - s_fn) - (s_fp - s_tn)));
    coef[c * 3 + 2] = use_ce ? (float)(1.0 / (G * cnt)) : 0.f;
    // rounded metrics
    const double rtp = sums[c][7] / cnt, rfp = (sums[c][6] - sums[c][7]) / cnt, rfn = (sums[c][5] - sums[c][7]) / cnt;
    const double rtn = (sums[c][0] - sums[c][5] - sums[c][6] + sums[c][7]) / cnt;
    dice_r += ((2 * rtp + eps) / (2 * rtp + rfp + rfn + eps)) / G;
    iou_r += ((rtp + eps) / (rtp + rfp + rfn + eps)) / G;
    mcc_r += ((rtp * rtn - rfp * rfn + eps) / (sqrt((rtp + rfp) * (rtp + rfn) * (rtn + rfp) * (rtn + rfn)) + eps)) / G;
  }
  out[0] = (float)loss; out[1] = (float)dice_r; out[2] = (float)iou_r; out[3] = (float)mcc_r;
}

// backward through loss, softmax and the 1x1 head.  Each thread walks HB_ITER voxels and keeps its
// dW / db partial sums in registers (PL planes x NCLS x 8 + NCLS floats), so the block reduction
// (shuffles + one barrier per plane) is paid once per HB_ITER * 256 voxels.  Instantiated for
// PL = 4 (base 32) and PL = 8 (base 64); wide heads (NCLS > 4) use HB_ITER = 1 to stay in registers.
template <int NCLS> struct HeadBwdIter { static constexpr int value = NCLS <= 4 ? 8 : 1; };

template <typename T, int NCLS, int PL>
__global__ __launch_bounds__(256) void head_loss_bwd_kernel(HeadLossParams p) {
  constexpr int ITER = (PL * NCLS <= 16) ? HeadBwdIter<NCLS>::value : 1;
  constexpr int C0 = PL * 8;
  const int n = blockIdx.y;
  const float lscale = p.loss_scale_dev ? *p.loss_scale_dev : p.loss_scale;
  float accw[PL][NCLS][8], accb[NCLS];
#pragma unroll
  for (int pl = 0; pl < PL; ++pl)
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) accw[pl][c][j] = 0.f;
#pragma unroll
  for (int c = 0; c < NCLS; ++c) accb[c] = 0.f;
  float wreg[NCLS][C0 > 32 ? 1 : C0];      // head weights in registers when they fit (base 32)
  if (C0 <= 32) {
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
      for (int k = 0; k < (C0 > 32 ? 1 : C0); ++k) wreg[c][k] = p.w[c * C0 + k];
  }
  auto W = [&](int c, int k) { return C0 <= 32 ? wreg[c][C0 > 32 ? 0 : k] : p.w[c * C0 + k]; };
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
    const long long v = ((long long)blockIdx.x * ITER + it) * 256 + threadIdx.x;
    if (v >= p.vox) break;
    const T* xin = (const T*)p.x + n * p.x_ss + v * 8;
    V8T<T> xv[PL];
#pragma unroll
    for (int pl = 0; pl < PL; ++pl) xv[pl] = head_act<T>(p, *(const V8T<T>*)(xin + (long long)pl * p.vox * 8), pl, n);
    float l[NCLS];
#pragma unroll
    for (int c = 0; c < NCLS; ++c) l[c] = p.bias[c];
#pragma unroll
    for (int pl = 0; pl < PL; ++pl)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float a = to_f32<T>(xv[pl][j]);
#pragma unroll
        for (int c = 0; c < NCLS; ++c) l[c] = fmaf(a, W(c, pl * 8 + j), l[c]);
      }
    float mx = l[0];
#pragma unroll
    for (int c = 1; c < NCLS; ++c) mx = fmaxf(mx, l[c]);
    float e[NCLS], s = 0.f;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) { e[c] = __expf(l[c] - mx); s += e[c]; }
    const float inv = 1.f / s;
    float g[NCLS], dl[NCLS], dot = 0.f;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
      const float pr = e[c] * inv;
      const long long to = ((long long)n * NCLS + c) * p.vox + v;
      const float y = load_t(p.target, to, p.tdtype);
      const float w = p.weight ? load_t(p.weight, to, p.tdtype) : 1.f;
      g[c] = w * (p.coef[c * 3] + p.coef[c * 3 + 1] * y) - p.coef[c * 3 + 2] * w * y / (pr + 1e-12f);
      e[c] = pr;
      dot += g[c] * pr;
    }
#pragma unroll
    for (int c = 0; c < NCLS; ++c) { dl[c] = e[c] * (g[c] - dot) * lscale; accb[c] += dl[c]; }   // softmax backward
    T* dxo = (T*)p.dx + n * p.dx_ss + v * 8;
#pragma unroll
    for (int pl = 0; pl < PL; ++pl) {
      V8T<T> o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xa = to_f32<T>(xv[pl][j]);
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < NCLS; ++c) { a = fmaf(dl[c], W(c, pl * 8 + j), a); accw[pl][c][j] = fmaf(dl[c], xa, accw[pl][c][j]); }
        o[j] = from_f32<T>(a);                        // dx = W^T dl
      }
      *(V8T<T>*)(dxo + (long long)pl * p.vox * 8) = o;
    }
  }
  // slab layout per part: [planes][NCLS][8] weight partials, then [NCLS] bias partials
  __shared__ float red[4 * NCLS * 8];
  const long long part = (long long)n * gridDim.x + blockIdx.x;
  float* slab = p.dwslab + part * (NCLS * (C0 + 1));
#pragma unroll
  for (int pl = 0; pl < PL; ++pl) {
    float vals[NCLS * 8];
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) vals[c * 8 + j] = accw[pl][c][j];
    __syncthreads();                       // previous plane's LDS partials are consumed
    block_reduce_store<NCLS * 8>(vals, red, slab + pl * NCLS * 8);
  }
  __syncthreads();
  block_reduce_store<NCLS>(accb, red, slab + PL * NCLS * 8);
}

// [r5] The head's backward AND the BatchNorm + ReLU backward of the last stage conv in two passes over that conv's raw output y -- the
// head's input gradient dz is never written.  The three-kernel sequence it replaces (head_loss_bwd_kernel -> bn_bwd_reduce_kernel ->
// bn_bwd_apply_kernel, with head_act) read y three times and wrote / read dz 1 + 2 times: 1.88 GB per C3 step at level 0 against 0.8.
// A workgroup walks 64 voxels at a time, wave pl its channel plane pl (C0 = 32: four waves; every per-channel constant is wave-uniform,
// i.e. lives in scalar registers): z = relu(bn(y)) as bn_relu_fwd would have stored it, the plane's share of the logits -> LDS, one
// barrier, the four shares summed in a fixed order (the same bits in all four waves), softmax and loss gradient per lane, dz = W^T dl
// for the wave's 8 channels ROUNDED to T as head_loss_bwd_kernel stored it.  (A quad of lanes per voxel with the planes across the quad
// kept the constants in 72 vector registers per lane: 3 waves per SIMD, 260 + 236 us for the two passes at 2 x 128^3.)
//   PASS 1: s1 += dz', s2 += dz' xhat (bn_bwd_reduce_kernel's arithmetic), dW += dl z, db += dl -- 16 + 8 NCLS + NCLS partial sums
//           per lane -> one BatchNorm row [C][2] and one head row [planes][NCLS][8] + [NCLS] per 2 048 voxels;
//   PASS 2: dy = a (dz' - c1 - xhat c2) (bn_bwd_apply_kernel's arithmetic) with the coefficients of bn_bwd_finalize_kernel.
// The logits add their 32 terms plane by plane (head_loss_fwd_kernel adds them in channel order): the gradient is that of a logit one
// rounding away from the forward's.
struct HeadBnBwdParams {
  const void* y; long long y_ss;
  const float* w; const float* bias;                 // head [NCLS][32], [NCLS]
  const void* target; const void* weight; int tdtype;
  const float* coef;                                 // loss coefficients [NCLS][3] (loss_finalize_kernel)
  float loss_scale; const float* loss_scale_dev;
  const float* scale; const float* shift; const float* mean; const float* invstd;      // the last conv's BatchNorm [32]
  const float* bncoef;                               // PASS 2: [32][3] (bn_bwd_finalize_kernel)
  float* bnslab; float* dwslab;                      // PASS 1 outputs: [parts][32][2], [parts][NCLS * 33]
  void* dy; long long dy_ss;                         // PASS 2 output
  float* dl;                                         // [N][vox][NCLS]: the logit gradients, written by PASS 1 (plane 0's wave), read by PASS 2
  long long vox; int per_block;
  int pss;                                           // per-sample stride of scale / shift / mean / invstd / bncoef rows (GroupNorm: C0; BatchNorm: 0)
};

#ifndef HBB_U
#define HBB_U 1
#endif
#ifndef HBB_OCC
#define HBB_OCC 1
#endif
template <typename T, int NCLS, int PASS, int PL>      // PL: channel planes of the head input = waves per workgroup (4: base 32, 8: base 64)
__global__ __launch_bounds__(PL * 64, HBB_OCC) void head_bn_bwd_kernel(HeadBnBwdParams p) {
  constexpr int C0 = PL * 8;
  constexpr int NV = 16 + NCLS * 8 + NCLS;           // PASS 1: partial sums per lane
  const int n = blockIdx.y, lane = threadIdx.x & 63;
  const int pl = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave = channel plane: every per-channel constant is wave-uniform (scalar registers)
  const float lscale = p.loss_scale_dev ? *p.loss_scale_dev : p.loss_scale;
  float sc[8], sh[8], mu[8], is[8], W[NCLS][8];
  [[maybe_unused]] float ca[8], c1[8], c2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = pl * 8 + j, cs = n * p.pss + c;
    sc[j] = p.scale[cs]; sh[j] = p.shift[cs]; mu[j] = p.mean[cs]; is[j] = p.invstd[cs];
#pragma unroll
    for (int k = 0; k < NCLS; ++k) W[k][j] = p.w[k * C0 + c];
    if (PASS == 2) { ca[j] = p.bncoef[cs * 3]; c1[j] = p.bncoef[cs * 3 + 1]; c2[j] = p.bncoef[cs * 3 + 2]; }
  }
  float bias[NCLS], lc[NCLS][3];
#pragma unroll
  for (int k = 0; k < NCLS; ++k) { bias[k] = p.bias[k]; lc[k][0] = p.coef[k * 3]; lc[k][1] = p.coef[k * 3 + 1]; lc[k][2] = p.coef[k * 3 + 2]; }
  [[maybe_unused]] float acc[PASS == 1 ? NV : 1];
  if (PASS == 1) {
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
  }
  constexpr int U = HBB_U;                           // voxels per lane and iteration: 64 U voxels per barrier
  __shared__ float lp[2][PL][NCLS][64 * U];           // the planes' shares of the logits, by iteration parity
  __shared__ float tv[2][2 * NCLS][64 * U];          // targets and weights of the voxels: value 2 k + which, loaded ONCE per voxel (wave pl takes
                                                     // values pl, pl + PL) -- every wave loading all of them cost 150-190 us of the two passes' 430
  constexpr int NTV = (2 * NCLS + PL - 1) / PL;      // values per wave
  const long long v0 = (long long)blockIdx.x * p.per_block, v1 = min(v0 + p.per_block, p.vox);
  const T* yin = (const T*)p.y + n * p.y_ss + (long long)pl * p.vox * 8;
  const int niter = (int)((v1 - v0 + 64 * U - 1) / (64 * U));
  auto load_y = [&](long long base, V8T<T> (&yy)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) yy[u] = *(const V8T<T>*)(yin + min(base + u * 64 + lane, v1 - 1) * 8);
  };
  auto load_tv = [&](long long base, float (&t)[U][NTV]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < NTV; ++i) {
        const int idx = pl + PL * i;                 // wave-uniform
        if (idx < 2 * NCLS) {
          const long long to = ((long long)n * NCLS + (idx >> 1)) * p.vox + min(base + u * 64 + lane, v1 - 1);
          t[u][i] = (idx & 1) ? (p.weight ? load_t(p.weight, to, p.tdtype) : 1.f) : load_t(p.target, to, p.tdtype);
        }
      }
  };
  V8T<T> ynext[U];
  float tnext[U][NTV];
  load_y(v0, ynext);
  load_tv(v0, tnext);
  for (int it = 0; it < niter; ++it) {
    const long long base = v0 + (long long)it * 64 * U;
    V8T<T> yy[U];
    float tcur[U][NTV];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      yy[u] = ynext[u];
#pragma unroll
      for (int i = 0; i < NTV; ++i) tcur[u][i] = tnext[u][i];
    }
    if (it + 1 < niter) { load_y(base + 64 * U, ynext); load_tv(base + 64 * U, tnext); }      // the next voxels are in flight over this iteration's barrier
    const int par = it & 1;
    float yv[U][8], zv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float part[NCLS];
#pragma unroll
      for (int k = 0; k < NCLS; ++k) part[k] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        yv[u][j] = to_f32<T>(yy[u][j]);
        zv[u][j] = to_f32<T>(from_f32<T>(fmaxf(fmaf(sc[j], yv[u][j], sh[j]), 0.f)));      // z as bn_relu_fwd_kernel would have stored it (head_act)
#pragma unroll
        for (int k = 0; k < NCLS; ++k) part[k] = fmaf(zv[u][j], W[k][j], part[k]);
      }
#pragma unroll
      for (int k = 0; k < NCLS; ++k) lp[par][pl][k][u * 64 + lane] = part[k];
#pragma unroll
      for (int i = 0; i < NTV; ++i) if (pl + PL * i < 2 * NCLS) tv[par][pl + PL * i][u * 64 + lane] = tcur[u][i];
    }
    __syncthreads();                                 // (one barrier per 64 U voxels: the other parity is being written by nobody yet -- its readers passed this barrier)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long v = base + u * 64 + lane;
      const bool live = v < v1;
      const int li = u * 64 + lane;
      float l[NCLS], mx, yt[NCLS], wt[NCLS];
#pragma unroll
      for (int k = 0; k < NCLS; ++k) {
        float lsum = (lp[par][0][k][li] + lp[par][1][k][li]) + (lp[par][2][k][li] + lp[par][3][k][li]);
        if (PL == 8) lsum += (lp[par][4][k][li] + lp[par][5][k][li]) + (lp[par][6][k][li] + lp[par][7][k][li]);
        l[k] = lsum + bias[k];
        yt[k] = tv[par][2 * k][li]; wt[k] = tv[par][2 * k + 1][li];
      }
      mx = l[0];
#pragma unroll
      for (int k = 1; k < NCLS; ++k) mx = fmaxf(mx, l[k]);
      float e[NCLS], ssum = 0.f;
#pragma unroll
      for (int k = 0; k < NCLS; ++k) { e[k] = __expf(l[k] - mx); ssum += e[k]; }
      const float inv = __builtin_amdgcn_rcpf(ssum);      // (1 ulp: far inside the rounding of dz to T; an IEEE division is ~10 instructions)
      float g[NCLS], dl[NCLS], dot = 0.f;
#pragma unroll
      for (int k = 0; k < NCLS; ++k) {
        const float pr = e[k] * inv;
        g[k] = wt[k] * (lc[k][0] + lc[k][1] * yt[k]) - lc[k][2] * wt[k] * yt[k] * __builtin_amdgcn_rcpf(pr + 1e-12f);
        e[k] = pr;
        dot += g[k] * pr;
      }
#pragma unroll
      for (int k = 0; k < NCLS; ++k) dl[k] = live ? e[k] * (g[k] - dot) * lscale : 0.f;      // softmax backward (head_loss_bwd_kernel); a lane past the end adds zeros
      [[maybe_unused]] V8T<T> o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < NCLS; ++k) a = fmaf(dl[k], W[k][j], a);
        const float dz = to_f32<T>(from_f32<T>(a));                    // the gradient head_loss_bwd_kernel stored
        const float d = zv[u][j] > 0.f ? dz : 0.f;
        if (PASS == 1) {
          acc[2 * j] += d;
          acc[2 * j + 1] += d * (yv[u][j] - mu[j]) * is[j];
#pragma unroll
          for (int k = 0; k < NCLS; ++k) acc[16 + k * 8 + j] = fmaf(dl[k], zv[u][j], acc[16 + k * 8 + j]);
        } else {
          const float xh = (yv[u][j] - mu[j]) * is[j];
          o[j] = from_f32<T>(ca[j] * (d - c1[j] - xh * c2[j]));
        }
      }
      if (PASS == 1) {
#pragma unroll
        for (int k = 0; k < NCLS; ++k) acc[16 + NCLS * 8 + k] += dl[k];      // (every plane's wave: plane 0's copy is the one kept)
        if (pl == 0 && live) {
#pragma unroll
          for (int k = 0; k < NCLS; ++k) p.dl[((long long)n * p.vox + v) * NCLS + k] = dl[k];
        }
      } else if (live) {
        *(V8T<T>*)((T*)p.dy + n * p.dy_ss + (long long)pl * p.vox * 8 + v * 8) = o;
      }
    }
  }
  if (PASS == 1) {
    // a wave owns its plane: its 64 lanes' sums are the block's row of that plane (no cross-wave step)
    const long long part = (long long)n * gridDim.x + blockIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float v = wave_sum(acc[i]);
      if (lane == 0) {
        if (i < 16) p.bnslab[(part * C0 + pl * 8 + (i >> 1)) * 2 + (i & 1)] = v;
        else if (i < 16 + NCLS * 8) p.dwslab[part * (NCLS * (C0 + 1)) + pl * NCLS * 8 + (i - 16)] = v;      // [planes][NCLS][8]
        else if (pl == 0) p.dwslab[part * (NCLS * (C0 + 1)) + PL * NCLS * 8 + (i - 16 - NCLS * 8)] = v;     // [NCLS] bias partials
      }
    }
  }
}

// PASS 2 without the head: dy = a (dz' - c1 - xhat c2) with dz = W^T dl from the logit gradients PASS 1 left (8 NCLS bytes per voxel
// instead of the logits, the softmax and the loss terms again).  Plane = grid dimension (wave-uniform constants), two voxels per thread.
template <typename T, int NCLS, int PL, bool GN = false>      // GN: bn_bwd_apply_kernel's GroupNorm form (a d - c1 - xhat c2, per-sample rows)
__global__ __launch_bounds__(256) void head_bn_apply_kernel(HeadBnBwdParams p) {
  constexpr int C0 = PL * 8;
  const int pl = blockIdx.y, n = blockIdx.z;
  const long long vb = (long long)blockIdx.x * 512 + threadIdx.x;
  float sc[8], sh[8], mu[8], is[8], W[NCLS][8], ca[8], c1[8], c2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = pl * 8 + j, cs = n * p.pss + c;
    sc[j] = p.scale[cs]; sh[j] = p.shift[cs]; mu[j] = p.mean[cs]; is[j] = p.invstd[cs];
    ca[j] = p.bncoef[cs * 3]; c1[j] = p.bncoef[cs * 3 + 1]; c2[j] = p.bncoef[cs * 3 + 2];
#pragma unroll
    for (int k = 0; k < NCLS; ++k) W[k][j] = p.w[k * C0 + c];
  }
  const long long po = (long long)pl * p.vox * 8;
  V8T<T> yy[2];
  float dl[2][NCLS];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const long long v = vb + u * 256;
    if (v < p.vox) {
      yy[u] = *(const V8T<T>*)((const T*)p.y + n * p.y_ss + po + v * 8);
#pragma unroll
      for (int k = 0; k < NCLS; ++k) dl[u][k] = p.dl[((long long)n * p.vox + v) * NCLS + k];
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const long long v = vb + u * 256;
    if (v >= p.vox) continue;
    V8T<T> o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float yv = to_f32<T>(yy[u][j]);
      const float zv = to_f32<T>(from_f32<T>(fmaxf(fmaf(sc[j], yv, sh[j]), 0.f)));
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < NCLS; ++k) a = fmaf(dl[u][k], W[k][j], a);
      const float dz = to_f32<T>(from_f32<T>(a));
      const float d = zv > 0.f ? dz : 0.f;
      const float xh = (yv - mu[j]) * is[j];
      o[j] = from_f32<T>(GN ? ca[j] * d - c1[j] - xh * c2[j] : ca[j] * (d - c1[j] - xh * c2[j]));
    }
    *(V8T<T>*)((T*)p.dy + n * p.dy_ss + po + v * 8) = o;
  }
}

// The same backward for the shapes whose dW partials do not fit the register file (PL * NCLS > 16: base 64 with three or
// more classes, base 32 with five or more).  Two phases per 256-voxel chunk: (1) one thread per voxel computes dl and
// dx = W^T dl and parks its x planes and dl in LDS; (2) the workgroup re-reads them as 256 / PL voxel lanes per channel
// plane, so a thread carries NCLS x 8 dW partials only.  One cross-lane reduction per HEAD_BWD_WIDE_ITER chunks.
#define HEAD_BWD_WIDE_ITER 8
template <typename T, int NCLS, int PL>
__global__ __launch_bounds__(256) void head_loss_bwd_wide_kernel(HeadLossParams p) {
 
  constexpr int C0 = PL * 8, SUBS = 256 / PL;
  const int n = blockIdx.y, t = threadIdx.x;
  const float lscale = p.loss_scale_dev ? *p.loss_scale_dev : p.loss_scale;
  __shared__ V8T<T> xs[PL][256];
  __shared__ float dls[256][NCLS];
  __shared__ float red[4 * NCLS];
  // head weights and the optional input activation's constants in LDS (broadcast reads): left as global loads the compiler
  // keeps all NCLS x C0 + 2 C0 of them in vector registers -- 256 VGPRs, one wave per SIMD, 468 us at C5's 64 x 4 head
  __shared__ float wsm[NCLS * C0];
  __shared__ float actp[2 * C0];
  for (int i = threadIdx.x; i < NCLS * C0; i += 256) wsm[i] = p.w[i];
  if (p.in_scale != nullptr)
    for (int i = threadIdx.x; i < C0; i += 256) { actp[i] = p.in_scale[n * p.pss + i]; actp[C0 + i] = p.in_shift[n * p.pss + i]; }
  __syncthreads();
  float accw[NCLS][8], accb[NCLS];
#pragma unroll
  for (int c = 0; c < NCLS; ++c) {
    accb[c] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) accw[c][j] = 0.f;
  }
  const int mypl = t / SUBS, sub = t % SUBS;
#pragma unroll 1
  for (int it = 0; it < HEAD_BWD_WIDE_ITER; ++it) {
    const long long v0 = ((long long)blockIdx.x * HEAD_BWD_WIDE_ITER + it) * 256;
    if (v0 >= p.vox) break;
    const long long v = v0 + t;
    const bool live = v < p.vox;
    // Phase 1, one channel plane at a time in ROLLED loops: with the planes unrolled the compiler keeps all NCLS x C0 head weights
    // (and the planes themselves) live at once -- 366 registers at C5's 64 x 4 head, one wave per SIMD or 200 spills.
    __syncthreads();                                    // the previous chunk's phase 2 has read xs / dls
    const T* xin = (const T*)p.x + n * p.x_ss + (live ? v : 0) * 8;
    float l[NCLS];
#pragma unroll
    for (int c = 0; c < NCLS; ++c) l[c] = p.bias[c];
#pragma unroll 1
    for (int pl = 0; pl < PL; ++pl) {
      V8T<T> x;
      if (live) {
        x = *(const V8T<T>*)(xin + (long long)pl * p.vox * 8);
        if (p.in_scale != nullptr) {
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = from_f32<T>(fmaxf(fmaf(actp[pl * 8 + j], to_f32<T>(x[j]), actp[C0 + pl * 8 + j]), 0.f));
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = from_f32<T>(0.f);
      }
      xs[pl][t] = x;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xa = to_f32<T>(x[j]);
#pragma unroll
        for (int c = 0; c < NCLS; ++c) l[c] = fmaf(xa, wsm[c * C0 + pl * 8 + j], l[c]);
      }
    }
    float dl[NCLS];
#pragma unroll
    for (int c = 0; c < NCLS; ++c) dl[c] = 0.f;
    if (live) {
      float mx = l[0];
#pragma unroll
      for (int c = 1; c < NCLS; ++c) mx = fmaxf(mx, l[c]);
      float e[NCLS], s = 0.f;
#pragma unroll
      for (int c = 0; c < NCLS; ++c) { e[c] = __expf(l[c] - mx); s += e[c]; }
      const float inv = 1.f / s;
      float g[NCLS], dot = 0.f;
#pragma unroll
      for (int c = 0; c < NCLS; ++c) {
        const float pr = e[c] * inv;
        const long long to = ((long long)n * NCLS + c) * p.vox + v;
        const float y = load_t(p.target, to, p.tdtype);
        const float w = p.weight ? load_t(p.weight, to, p.tdtype) : 1.f;
        g[c] = w * (p.coef[c * 3] + p.coef[c * 3 + 1] * y) - p.coef[c * 3 + 2] * w * y / (pr + 1e-12f);
        e[c] = pr;
        dot += g[c] * pr;
      }
#pragma unroll
      for (int c = 0; c < NCLS; ++c) { dl[c] = e[c] * (g[c] - dot) * lscale; accb[c] += dl[c]; }   // softmax backward
      T* dxo = (T*)p.dx + n * p.dx_ss + v * 8;
#pragma unroll 1
      for (int pl = 0; pl < PL; ++pl) {
        V8T<T> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float a = 0.f;
#pragma unroll
          for (int c = 0; c < NCLS; ++c) a = fmaf(dl[c], wsm[c * C0 + pl * 8 + j], a);
          o[j] = from_f32<T>(a);                        // dx = W^T dl
        }
        *(V8T<T>*)(dxo + (long long)pl * p.vox * 8) = o;
      }
    }
#pragma unroll
    for (int c = 0; c < NCLS; ++c) dls[t][c] = dl[c];   // dl = 0 for a voxel past the end: its zero x adds nothing
    __syncthreads();
#pragma unroll 2
    for (int i = 0; i < PL; ++i) {                      // dW[c][mypl*8 + j] += dl[c][v] x[mypl*8 + j][v]
      const int vv = sub + i * SUBS;
      const V8T<T> xq = xs[mypl][vv];
      float d[NCLS];
#pragma unroll
      for (int c = 0; c < NCLS; ++c) d[c] = dls[vv][c];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xa = to_f32<T>(xq[j]);
#pragma unroll
        for (int c = 0; c < NCLS; ++c) accw[c][j] = fmaf(d[c], xa, accw[c][j]);
      }
    }
  }
  // slab layout per part: [planes][NCLS][8] weight partials, then [NCLS] bias partials
  const long long part = (long long)n * gridDim.x + blockIdx.x;
  float* slab = p.dwslab + part * (NCLS * (C0 + 1));
#pragma unroll
  for (int c = 0; c < NCLS; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = accw[c][j];
#pragma unroll
      for (int o = SUBS / 2; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      if (sub == 0) slab[mypl * NCLS * 8 + c * 8 + j] = a;
    }
  __syncthreads();
  block_reduce_store<NCLS>(accb, red, slab + PL * NCLS * 8);
}

// out[i] = alpha * sum_p slab[p][i] (+ out[i] if accumulate); fixed order.
__global__ __launch_bounds__(256) void reduce_slab_kernel(const float* __restrict__ slab, int nparts, long long n,
                                                          float* __restrict__ out, float alpha, int accumulate) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += slab[(long long)p * n + i];
  out[i] = alpha * s + (accumulate ? out[i] : 0.f);
}

// narrow slabs: 4 columns x 64 row groups per block; a thread walks nparts / 64 rows, four loads in flight, and the groups meet
// in LDS (fixed order).  The one-thread-per-column kernel above walks ALL rows in one thread: 2 048 loads, each waited for, for
// the head gradient.
__global__ __launch_bounds__(256) void reduce_slab_tree_kernel(const float* __restrict__ slab, int nparts, long long n,
                                                               float* __restrict__ out, float alpha, int accumulate) {
  __shared__ float red[64][4];
  const int col = threadIdx.x & 3, grp = threadIdx.x >> 2;
  const long long i = (long long)blockIdx.x * 4 + col;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int p = grp;
    for (; p + 192 < nparts; p += 256) {
      const float a = slab[(long long)p * n + i], b = slab[(long long)(p + 64) * n + i];
      const float c = slab[(long long)(p + 128) * n + i], d = slab[(long long)(p + 192) * n + i];
      s0 += a; s1 += b; s2 += c; s3 += d;
    }
    for (; p < nparts; p += 64) s0 += slab[(long long)p * n + i];
  }
  red[grp][col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp != 0 || i >= n) return;
  float s = red[0][col];
  for (int g = 1; g < 64; ++g) s += red[g][col];
  out[i] = alpha * s + (accumulate ? out[i] : 0.f);
}

// stage A of a wide reduction, in place: slab[g][i] += slab[g + G][i] + slab[g + 2G][i] + ...  (g < G)
__global__ __launch_bounds__(256) void fold_slab_kernel(float* __restrict__ slab, int nparts, long long n, int G) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;
  if (i >= n) return;
  float s = slab[(long long)g * n + i];
  for (int p = g + G; p < nparts; p += G) s += slab[(long long)p * n + i];
  slab[(long long)g * n + i] = s;
}

// ------------------------------------------------------------------ AdamW (torch defaults, decoupled decay)
// flag[0] != 0 (non-finite gradient found) -> the step is skipped entirely (fp16 loss scaling).
__global__ __launch_bounds__(256) void check_finite_kernel(const float* __restrict__ g, long long n, int* flag) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  bool bad = false;
  for (; i < n; i += (long long)gridDim.x * 256) bad |= !isfinite(g[i]);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2_sqrt, float ginv,
                                                    const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gr = g[i] * ginv;
  float pv = p[i] * (1.f - lr * wd);
  const float mn = b1 * m[i] + (1.f - b1) * gr;
  const float vn = b2 * v[i] + (1.f - b2) * gr * gr;
  m[i] = mn; v[i] = vn;
  const float denom = sqrtf(vn) / bc2_sqrt + eps;
  pv -= (lr / bc1) * (mn / denom);
  p[i] = pv;
}

// ---- the training state on the device (csrc/train_net.hip; interactive_unet/train_engine.py): no host read per step
// state[0] float loss scale, [1] int completed optimiser steps, [2] int good steps since the last scale change, [3] int overflow flag of
// the step in flight, [4] float bc1 = 1 - b1^step, [5] float sqrt(1 - b2^step), [6] float 1 / (loss scale x world), [7] int dynamic scale
__global__ void train_state_coef_kernel(float* __restrict__ st, float b1, float b2, float world) {
  int* si = (int*)st;
  // in double, as torch.optim.AdamW's host arithmetic: 1 - 0.999^step in fp32 loses 1.3e-5 of its value to cancellation at step 1
  // (ADVICE r4); one thread, once per step
  const double step = (double)(si[1] + 1);
  st[4] = (float)(1.0 - pow((double)b1, step));
  st[5] = (float)sqrt(1.0 - pow((double)b2, step));
  st[6] = 1.0f / (st[0] * world);
}
// GradScaler semantics (the reference trains under precision='16-mixed', trainer.py:59): an overflowing step is skipped, halves the
// scale and is not counted; 2000 good steps in a row double it.  A fixed scale counts the APPLIED steps too: a step skipped for a
// non-finite fp16 gradient must not advance the bias corrections.
__global__ void train_state_update_kernel(float* __restrict__ st) {
  int* si = (int*)st;
  const bool dyn = si[7] != 0, bad = si[3] != 0;
  if (!dyn) { if (!bad) si[1] += 1; return; }
  if (bad) { st[0] = fmaxf(st[0] * 0.5f, 1.0f); si[2] = 0; }
  else {
    si[1] += 1;
    si[2] += 1;
    if (si[2] >= 2000) { st[0] *= 2.0f; si[2] = 0; }
  }
}
__global__ __launch_bounds__(256) void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long long n, float lr, float b1, float b2,
                                                        float eps, float wd, const float* __restrict__ st) {
  if (((const int*)st)[3]) return;                       // the gradient overflowed: the step is skipped
  const float bc1 = st[4], bc2_sqrt = st[5], ginv = st[6];
  auto one = [&](float& pv, float gv, float& mv, float& vv) {      // (per element as ever; four elements per thread: 16-byte accesses, 27 -> ~12 us at 1.9 M parameters)
    const float gr = gv * ginv;
    float pn = pv * (1.f - lr * wd);
    const float mn = b1 * mv + (1.f - b1) * gr;
    const float vn = b2 * vv + (1.f - b2) * gr * gr;
    mv = mn; vv = vn;
    const float denom = sqrtf(vn) / bc2_sqrt + eps;
    pn -= (lr / bc1) * (mn / denom);
    pv = pn;
  };
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 pv = *(float4*)(p + i), mv = *(float4*)(m + i), vv = *(float4*)(v + i);
    const float4 gv = *(const float4*)(g + i);
    one(pv.x, gv.x, mv.x, vv.x); one(pv.y, gv.y, mv.y, vv.y); one(pv.z, gv.z, mv.z, vv.z); one(pv.w, gv.w, mv.w, vv.w);
    *(float4*)(p + i) = pv; *(float4*)(m + i) = mv; *(float4*)(v + i) = vv;
  } else {
    for (long long k = i; k < n; ++k) one(p[k], g[k], m[k], v[k]);
  }
}
// head gradient from the reduced slab row: [C0 / 8][ncls][8] weight sums, then [ncls] bias sums -> dW [ncls][C0], db [ncls]
__global__ void head_grad_scatter_kernel(const float* __restrict__ t, float* __restrict__ dw, float* __restrict__ db, int ncls, int C0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ncls * C0) {
    const int c = i / C0, ch = i - c * C0;
    dw[i] = t[((ch >> 3) * ncls + c) * 8 + (ch & 7)];
  } else if (i < ncls * C0 + ncls) {
    db[i - ncls * C0] = t[i];
  }
}

}  // namespace

#define DT_OK(dt) IUNET_REQUIRE((dt) == 0 || (dt) == 1, "dtype must be 0 (f16) or 1 (bf16), got %d", (dt))
#define LAUNCH_T(kern, grid, ...)                                                              \
  do {                                                                                         \
    if (dtype == 0) hipLaunchKernelGGL((kern<f16>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);  \
    else hipLaunchKernelGGL((kern<bf16>), grid, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
  } while (0)

// (train_f32.hip: the fp32 parity form of the head + loss forward shares the finalize pass)
int iunet_loss_finalize_launch(const float* slab, int nparts, int ncls, int kind, int has_weight, double nvox_total, float* out4,
                               float* coef, hipStream_t stream) {
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, stream, slab, nparts, ncls, kind, has_weight, nvox_total, out4, coef);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

extern "C" {

int iunet_bn_finalize(const void* slab, int nparts, int C, double count, const void* gamma, const void* beta,
                      void* running_mean, void* running_var, float momentum, float eps, void* scale, void* shift,
                      void* mean, void* invstd, void* stream) {
  IUNET_REQUIRE(slab && gamma && beta && scale && shift && mean && invstd, "bn_finalize: null pointer");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(nparts >= 2048 ? 1024 : 256), 0, (hipStream_t)stream, (const float*)slab, nparts, C, count,
                     (const float*)gamma, (const float*)beta, (float*)running_mean, (float*)running_var, momentum, eps,
                     (float*)scale, (float*)shift, (float*)mean, (float*)invstd);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_bn_relu_fwd(int dtype, const void* y, long long y_ss, void* z, long long z_ss, const void* scale,
                      const void* shift, int C, int N, long long vox, void* stream) {
  DT_OK(dtype);
  dim3 grid((unsigned)((vox + 511) / 512), C / 8, N);
  if (dtype == 0) hipLaunchKernelGGL(bn_relu_fwd_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)y, y_ss, (f16*)z, z_ss, (const float*)scale, (const float*)shift, C / 8, vox);
  else hipLaunchKernelGGL(bn_relu_fwd_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)y, y_ss, (bf16*)z, z_ss, (const float*)scale, (const float*)shift, C / 8, vox);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

// bn_relu_fwd and the 2^d max-pool of its output in one pass; (Do, Ho, Wo) = pooled grid, y / z on the 2x grid
int iunet_bn_relu_pool_fwd(int dtype, int nd, const void* y, long long y_ss, void* z, long long z_ss, void* pooled,
                           long long p_ss, const void* scale, const void* shift, int C, int N, int Do, int Ho, int Wo,
                           void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(y && z && pooled && scale && shift, "bn_relu_pool_fwd: null pointer");
  IUNET_REQUIRE(nd == 2 || nd == 3, "bn_relu_pool_fwd: nd must be 2 or 3");
  const long long ovox = (long long)Do * Ho * Wo;
  dim3 grid((unsigned)((ovox + 255) / 256), C / 8, N);
#define BRP(TT, NDV) hipLaunchKernelGGL((bn_relu_pool_fwd_kernel<TT, NDV>), grid, dim3(256), 0, (hipStream_t)stream, (const TT*)y, y_ss, (TT*)z, z_ss, (TT*)pooled, p_ss, (const float*)scale, (const float*)shift, Do, Ho, Wo)
  if (dtype == 0) { if (nd == 3) BRP(f16, 3); else BRP(f16, 2); } else { if (nd == 3) BRP(bf16, 3); else BRP(bf16, 2); }
#undef BRP
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

static const int BN_BWD_PER_BLOCK = 2048;       // (measured at level 0 of C3: 8192 -> 139 us, 2048 -> 122 us, 1024 -> 147 us)
static const int BN_POOL_PER_BLOCK = 8192;      // input voxels per workgroup of the pooled variant's first pass (its 2^d windows make 2048 slower)

int iunet_bn_bwd_num_parts(int N, long long vox) {
  const int per_block = BN_BWD_PER_BLOCK;
  return N * (int)((vox + per_block - 1) / per_block);
}

int iunet_bn_relu_bwd(int dtype, const void* dz, long long dz_ss, const void* z, long long z_ss, const void* y,
                      long long y_ss, void* dy, long long dy_ss, const void* mean, const void* invstd, const void* gamma,
                      const void* scale, const void* shift, void* dgamma, void* dbeta, void* slab, void* coef, int C, int N,
                      long long vox, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(dz && y && slab && coef && scale && shift, "bn_relu_bwd: null pointer");      // dy NULL: sums + coefficients only
  const int per_block = BN_BWD_PER_BLOCK;
  const int chunks = (int)((vox + per_block - 1) / per_block);
  dim3 g1(chunks, C / 8, N);
  if (dtype == 0) hipLaunchKernelGGL(bn_bwd_reduce_kernel<f16>, g1, dim3(256), 0, (hipStream_t)stream, (const f16*)dz, dz_ss, (const f16*)z, z_ss, (const f16*)y, y_ss, (const float*)mean, (const float*)invstd, (const float*)scale, (const float*)shift, C, vox, per_block, (float*)slab);
  else hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16>, g1, dim3(256), 0, (hipStream_t)stream, (const bf16*)dz, dz_ss, (const bf16*)z, z_ss, (const bf16*)y, y_ss, (const float*)mean, (const float*)invstd, (const float*)scale, (const float*)shift, C, vox, per_block, (float*)slab);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, (const float*)slab, chunks * N, C,
                     (double)N * (double)vox, (const float*)gamma, (const float*)invstd, (float*)dgamma, (float*)dbeta, (float*)coef);
  if (dy == nullptr) { IUNET_CHECK_HIP(hipGetLastError()); return IUNET_OK; }      // the consumer applies pass 2 while staging
  dim3 g2((unsigned)((vox + 511) / 512), C / 8, N);
  if (dtype == 0) hipLaunchKernelGGL(bn_bwd_apply_kernel<f16>, g2, dim3(256), 0, (hipStream_t)stream, (const f16*)dz, dz_ss, (const f16*)z, z_ss, (const f16*)y, y_ss, (f16*)dy, dy_ss, (const float*)mean, (const float*)invstd, (const float*)coef, (const float*)scale, (const float*)shift, C / 8, vox);
  else hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16>, g2, dim3(256), 0, (hipStream_t)stream, (const bf16*)dz, dz_ss, (const bf16*)z, z_ss, (const bf16*)y, y_ss, (bf16*)dy, dy_ss, (const float*)mean, (const float*)invstd, (const float*)coef, (const float*)scale, (const float*)shift, C / 8, vox);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

// iunet_bn_relu_bwd whose first pass has already been done: `slab` holds nparts rows [C][2] of (sum dz', sum dz' * xhat) written by
// the data-gradient launch that produced dz (iunet_conv3_dgrad_bnstats).  Finalize (dgamma, dbeta, coefficients) + pass 2
// (dy; NULL: the consumer applies it).
int iunet_bn_relu_bwd_apply(int dtype, const void* dz, long long dz_ss, const void* y, long long y_ss, void* dy, long long dy_ss,
                            const void* mean, const void* invstd, const void* gamma, const void* scale, const void* shift,
                            void* dgamma, void* dbeta, const void* slab, int nparts, void* coef, int C, int N, long long vox,
                            void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(dz && y && slab && coef && scale && shift && mean && invstd && gamma && dgamma && dbeta, "bn_relu_bwd_apply: null pointer");
  IUNET_REQUIRE(C > 0 && C % 8 == 0 && N > 0 && vox > 0 && nparts > 0, "bn_relu_bwd_apply: C %d, N %d, %lld voxels, %d rows", C, N, vox, nparts);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, (const float*)slab, nparts, C,
                     (double)N * (double)vox, (const float*)gamma, (const float*)invstd, (float*)dgamma, (float*)dbeta, (float*)coef);
  if (dy == nullptr) { IUNET_CHECK_HIP(hipGetLastError()); return IUNET_OK; }
  dim3 g2((unsigned)((vox + 511) / 512), C / 8, N);
  if (dtype == 0) hipLaunchKernelGGL(bn_bwd_apply_kernel<f16>, g2, dim3(256), 0, (hipStream_t)stream, (const f16*)dz, dz_ss, (const f16*)nullptr, 0LL, (const f16*)y, y_ss, (f16*)dy, dy_ss, (const float*)mean, (const float*)invstd, (const float*)coef, (const float*)scale, (const float*)shift, C / 8, vox);
  else hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16>, g2, dim3(256), 0, (hipStream_t)stream, (const bf16*)dz, dz_ss, (const bf16*)nullptr, 0LL, (const bf16*)y, y_ss, (bf16*)dy, dy_ss, (const float*)mean, (const float*)invstd, (const float*)coef, (const float*)scale, (const float*)shift, C / 8, vox);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

/* ---- GroupNorm + ReLU (north_star "GroupNorm/BN") ------------------------------------------------------- */
int iunet_gn_num_parts(int N, long long vox) { return iunet_bn_bwd_num_parts(N, vox); }

// z = relu(group_norm(y, groups, gamma, beta, eps)): statistics pass + finalize + one normalise pass per sample.
// slab: iunet_gn_num_parts(N, vox) * C * 2 floats; scale / shift / mean / invstd: fp32 [N][C] outputs (the backward reads them).
// rows > 0: `slab` already holds the statistics, [N][rows][C][2] partial sums written by the conv that produced y
// (iunet_conv3_fwd_sample_stats): no statistics pass.
int iunet_gn_relu_fwd_rows(int dtype, const void* y, long long y_ss, void* z, long long z_ss, const void* gamma, const void* beta,
                           int groups, float eps, void* slab, int rows, void* scale, void* shift, void* mean, void* invstd, int C, int N,
                           long long vox, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(y && gamma && beta && slab && scale && shift && mean && invstd, "gn_relu_fwd: null pointer");      // z NULL: statistics -> scale / shift only (the consumer applies them: iunet_head_loss_fwd_act_ps)
  IUNET_REQUIRE(C > 0 && C % 8 == 0 && N > 0 && vox > 0 && rows >= 0, "gn_relu_fwd: C %d (multiple of 8), N %d, %lld voxels, %d rows", C, N, vox, rows);
  IUNET_REQUIRE(groups > 0 && C % groups == 0, "gn_relu_fwd: %d channels do not split into %d groups", C, groups);
  const int per_block = BN_BWD_PER_BLOCK;
  const int chunks = rows > 0 ? rows : (int)((vox + per_block - 1) / per_block);
  dim3 g1(chunks, C / 8, N);
  if (rows > 0) {}
  else if (dtype == 0) hipLaunchKernelGGL(gn_stats_kernel<f16>, g1, dim3(256), 0, (hipStream_t)stream, (const f16*)y, y_ss, C, vox, per_block, (float*)slab);
  else hipLaunchKernelGGL(gn_stats_kernel<bf16>, g1, dim3(256), 0, (hipStream_t)stream, (const bf16*)y, y_ss, C, vox, per_block, (float*)slab);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, N), dim3(256), 0, (hipStream_t)stream, (const float*)slab, chunks, C, groups,
                     (double)vox, (const float*)gamma, (const float*)beta, eps, (float*)scale, (float*)shift, (float*)mean, (float*)invstd);
  dim3 g2((unsigned)((vox + 511) / 512), C / 8, N);        // one launch over the samples: sample n reads its row of scale / shift
  if (z == nullptr) {}
  else if (dtype == 0) hipLaunchKernelGGL(bn_relu_fwd_kernel<f16>, g2, dim3(256), 0, (hipStream_t)stream, (const f16*)y, y_ss, (f16*)z, z_ss, (const float*)scale, (const float*)shift, C / 8, vox, C);
  else hipLaunchKernelGGL(bn_relu_fwd_kernel<bf16>, g2, dim3(256), 0, (hipStream_t)stream, (const bf16*)y, y_ss, (bf16*)z, z_ss, (const float*)scale, (const float*)shift, C / 8, vox, C);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_gn_relu_fwd(int dtype, const void* y, long long y_ss, void* z, long long z_ss, const void* gamma, const void* beta,
                      int groups, float eps, void* slab, void* scale, void* shift, void* mean, void* invstd, int C, int N,
                      long long vox, void* stream) {
  return iunet_gn_relu_fwd_rows(dtype, y, y_ss, z, z_ss, gamma, beta, groups, eps, slab, 0, scale, shift, mean, invstd, C, N, vox, stream);
}

// GroupNorm statistics of ONE sample from the BatchNorm-statistics epilogue of the convolutions (stats = [nparts][C][2] partial sums
// (sum, sum of squares) of a launch with N = 1: iunet_conv3_fwd / iunet_first_conv_fwd) -> scale / shift / mean / invstd [C] of that
// sample: what iunet_conv3_fwd_act (the next conv's loader waves) and iunet_bn_relu_fwd / _pool_fwd apply.  The statistics pass of
// iunet_gn_relu_fwd over the tensor goes away.
int iunet_gn_finalize(const void* stats, int nparts, int C, int groups, long long vox, const void* gamma, const void* beta, float eps,
                      void* scale, void* shift, void* mean, void* invstd, void* stream) {
  IUNET_REQUIRE(stats && gamma && beta && scale && shift && mean && invstd, "gn_finalize: null pointer");
  IUNET_REQUIRE(nparts > 0 && C > 0 && groups > 0 && C % groups == 0 && vox > 0, "gn_finalize: %d parts, %d channels in %d groups, %lld voxels",
                nparts, C, groups, vox);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, 1), dim3(256), 0, (hipStream_t)stream, (const float*)stats, nparts, C, groups,
                     (double)vox, (const float*)gamma, (const float*)beta, eps, (float*)scale, (float*)shift, (float*)mean, (float*)invstd);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

// backward of z = relu(group_norm(y)): dy, dgamma, dbeta from dz and y; scale / shift / mean / invstd [N][C] from the forward;
// slab as above, coef: N * C * 3 floats of scratch.  C <= 1024.
int iunet_gn_relu_bwd_rows(int dtype, const void* dz, long long dz_ss, const void* y, long long y_ss, void* dy, long long dy_ss,
                           const void* gamma, int groups, const void* scale, const void* shift, const void* mean, const void* invstd,
                           void* dgamma, void* dbeta, void* slab, int rows, void* coef, int C, int N, long long vox, void* stream);
int iunet_gn_relu_bwd(int dtype, const void* dz, long long dz_ss, const void* y, long long y_ss, void* dy, long long dy_ss,
                      const void* gamma, int groups, const void* scale, const void* shift, const void* mean, const void* invstd,
                      void* dgamma, void* dbeta, void* slab, void* coef, int C, int N, long long vox, void* stream) {
  return iunet_gn_relu_bwd_rows(dtype, dz, dz_ss, y, y_ss, dy, dy_ss, gamma, groups, scale, shift, mean, invstd, dgamma, dbeta, slab, 0, coef, C, N, vox, stream);
}
// rows > 0: `slab` already holds the first pass, [N][rows][C][2] = (sum dz', sum dz' xhat) per sample from the data-gradient launch that
// produced dz (iunet_conv3_dgrad_sample_bnstats): no reduction pass over dz and y.
int iunet_gn_relu_bwd_rows(int dtype, const void* dz, long long dz_ss, const void* y, long long y_ss, void* dy, long long dy_ss,
                           const void* gamma, int groups, const void* scale, const void* shift, const void* mean, const void* invstd,
                           void* dgamma, void* dbeta, void* slab, int rows, void* coef, int C, int N, long long vox, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(dz && y && dy && gamma && scale && shift && mean && invstd && dgamma && dbeta && slab && coef, "gn_relu_bwd: null pointer");
  IUNET_REQUIRE(C > 0 && C % 8 == 0 && C <= 1024 && N > 0 && vox > 0, "gn_relu_bwd: C %d (multiple of 8, <= 1024), N %d, %lld voxels", C, N, vox);
  IUNET_REQUIRE(groups > 0 && C % groups == 0, "gn_relu_bwd: %d channels do not split into %d groups", C, groups);
  const int per_block = BN_BWD_PER_BLOCK;
  const int chunks = rows > 0 ? rows : (int)((vox + per_block - 1) / per_block);
  dim3 g1(chunks, C / 8, N), g2((unsigned)((vox + 511) / 512), C / 8, N);      // one launch over the samples: sample n reads its rows of the parameters
  const float *mu = (const float*)mean, *is = (const float*)invstd, *sc = (const float*)scale, *sh = (const float*)shift;
  if (rows > 0) {}
  else if (dtype == 0) hipLaunchKernelGGL(bn_bwd_reduce_kernel<f16>, g1, dim3(256), 0, (hipStream_t)stream, (const f16*)dz, dz_ss, (const f16*)nullptr, 0LL, (const f16*)y, y_ss, mu, is, sc, sh, C, vox, per_block, (float*)slab, C);
  else hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16>, g1, dim3(256), 0, (hipStream_t)stream, (const bf16*)dz, dz_ss, (const bf16*)nullptr, 0LL, (const bf16*)y, y_ss, mu, is, sc, sh, C, vox, per_block, (float*)slab, C);
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(groups), dim3(1024), 0, (hipStream_t)stream, (const float*)slab, chunks, C, groups, N,
                     (double)vox, (const float*)gamma, (const float*)invstd, (float*)dgamma, (float*)dbeta, (float*)coef);
  const float* cf = (const float*)coef;
  if (dtype == 0) hipLaunchKernelGGL((bn_bwd_apply_kernel<f16, true>), g2, dim3(256), 0, (hipStream_t)stream, (const f16*)dz, dz_ss, (const f16*)nullptr, 0LL, (const f16*)y, y_ss, (f16*)dy, dy_ss, mu, is, cf, sc, sh, C / 8, vox, C);
  else hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16, true>), g2, dim3(256), 0, (hipStream_t)stream, (const bf16*)dz, dz_ss, (const bf16*)nullptr, 0LL, (const bf16*)y, y_ss, (bf16*)dy, dy_ss, mu, is, cf, sc, sh, C / 8, vox, C);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

// iunet_gn_relu_fwd whose normalise pass also writes the 2^d max-pool of z (encoder stages: the pool would re-read z right away):
// statistics pass + finalize + ONE pass that writes z and the pooled tensor ((Do, Ho, Wo) grid, p_ss elements per sample).
int iunet_gn_relu_pool_fwd_rows(int dtype, int nd, const void* y, long long y_ss, void* z, long long z_ss, void* pooled, long long p_ss,
                                const void* gamma, const void* beta, int groups, float eps, void* slab, int rows, void* scale, void* shift,
                                void* mean, void* invstd, int C, int N, int Do, int Ho, int Wo, void* stream);
int iunet_gn_relu_pool_fwd(int dtype, int nd, const void* y, long long y_ss, void* z, long long z_ss, void* pooled, long long p_ss,
                           const void* gamma, const void* beta, int groups, float eps, void* slab, void* scale, void* shift, void* mean,
                           void* invstd, int C, int N, int Do, int Ho, int Wo, void* stream) {
  return iunet_gn_relu_pool_fwd_rows(dtype, nd, y, y_ss, z, z_ss, pooled, p_ss, gamma, beta, groups, eps, slab, 0, scale, shift, mean, invstd, C, N,
                                     Do, Ho, Wo, stream);
}
// (rows > 0: as iunet_gn_relu_fwd_rows)
int iunet_gn_relu_pool_fwd_rows(int dtype, int nd, const void* y, long long y_ss, void* z, long long z_ss, void* pooled, long long p_ss,
                                const void* gamma, const void* beta, int groups, float eps, void* slab, int rows, void* scale, void* shift,
                                void* mean, void* invstd, int C, int N, int Do, int Ho, int Wo, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(y && z && pooled && gamma && beta && slab && scale && shift && mean && invstd, "gn_relu_pool_fwd: null pointer");
  IUNET_REQUIRE(nd == 2 || nd == 3, "gn_relu_pool_fwd: nd must be 2 or 3");
  IUNET_REQUIRE(C > 0 && C % 8 == 0 && N > 0 && Do > 0 && Ho > 0 && Wo > 0, "gn_relu_pool_fwd: bad shape");
  IUNET_REQUIRE(groups > 0 && C % groups == 0, "gn_relu_pool_fwd: %d channels do not split into %d groups", C, groups);
  const long long ovox = (long long)Do * Ho * Wo, vox = ovox * (nd == 3 ? 8 : 4);
  const int per_block = BN_BWD_PER_BLOCK;
  const int chunks = rows > 0 ? rows : (int)((vox + per_block - 1) / per_block);
  dim3 g1(chunks, C / 8, N);
  if (rows > 0) {}
  else if (dtype == 0) hipLaunchKernelGGL(gn_stats_kernel<f16>, g1, dim3(256), 0, (hipStream_t)stream, (const f16*)y, y_ss, C, vox, per_block, (float*)slab);
  else hipLaunchKernelGGL(gn_stats_kernel<bf16>, g1, dim3(256), 0, (hipStream_t)stream, (const bf16*)y, y_ss, C, vox, per_block, (float*)slab);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, N), dim3(256), 0, (hipStream_t)stream, (const float*)slab, chunks, C, groups,
                     (double)vox, (const float*)gamma, (const float*)beta, eps, (float*)scale, (float*)shift, (float*)mean, (float*)invstd);
  dim3 grid((unsigned)((ovox + 255) / 256), C / 8, N);
#define GRP(TT, NDV) hipLaunchKernelGGL((bn_relu_pool_fwd_kernel<TT, NDV>), grid, dim3(256), 0, (hipStream_t)stream, (const TT*)y, y_ss, (TT*)z, z_ss, (TT*)pooled, p_ss, (const float*)scale, (const float*)shift, Do, Ho, Wo, C)
  if (dtype == 0) { if (nd == 3) GRP(f16, 3); else GRP(f16, 2); } else { if (nd == 3) GRP(bf16, 3); else GRP(bf16, 2); }
#undef GRP
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

// iunet_maxpool_bwd (add_skip) + iunet_gn_relu_bwd of an encoder stage's second conv in two passes instead of three (the GroupNorm form of
// iunet_bn_relu_pool_bwd): dz = dskip + route(dpool) is formed on the fly in both passes and never written.  slab / coef as iunet_gn_relu_bwd.
int iunet_gn_relu_pool_bwd(int dtype, int nd, const void* dskip, long long ds_ss, const void* dpool, long long dp_ss, const void* y,
                           long long y_ss, void* dy, long long dy_ss, const void* gamma, int groups, const void* scale, const void* shift,
                           const void* mean, const void* invstd, void* dgamma, void* dbeta, void* slab, void* coef, int C, int N, int Do,
                           int Ho, int Wo, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(dskip && dpool && y && dy && gamma && scale && shift && mean && invstd && dgamma && dbeta && slab && coef, "gn_relu_pool_bwd: null pointer");
  IUNET_REQUIRE(nd == 2 || nd == 3, "gn_relu_pool_bwd: nd must be 2 or 3");
  IUNET_REQUIRE(C > 0 && C % 8 == 0 && C <= 1024 && N > 0 && Do > 0 && Ho > 0 && Wo > 0, "gn_relu_pool_bwd: bad shape");
  IUNET_REQUIRE(groups > 0 && C % groups == 0, "gn_relu_pool_bwd: %d channels do not split into %d groups", C, groups);
  const long long ovox = (long long)Do * Ho * Wo, vox = ovox * (nd == 3 ? 8 : 4);
  const int per_block = BN_POOL_PER_BLOCK / (nd == 3 ? 8 : 4);
  const int chunks = (int)((ovox + per_block - 1) / per_block);
  IUNET_REQUIRE(chunks * N <= iunet_bn_bwd_num_parts(N, vox), "gn_relu_pool_bwd: slab part count");
  dim3 g1(chunks, C / 8, N), g2((unsigned)((ovox + 255) / 256), C / 8, N);
#define GPB(TT, NDV, PASSV, GRID, PB) hipLaunchKernelGGL((bn_pool_bwd_kernel<TT, NDV, PASSV, true>), GRID, dim3(256), 0, (hipStream_t)stream, \
    (const TT*)dskip, ds_ss, (const TT*)dpool, dp_ss, (const TT*)y, y_ss, (TT*)dy, dy_ss, (const float*)mean, (const float*)invstd, \
    (const float*)coef, (const float*)scale, (const float*)shift, C, Do, Ho, Wo, PB, (float*)slab, C)
  if (dtype == 0) { if (nd == 3) GPB(f16, 3, 1, g1, per_block); else GPB(f16, 2, 1, g1, per_block); }
  else { if (nd == 3) GPB(bf16, 3, 1, g1, per_block); else GPB(bf16, 2, 1, g1, per_block); }
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(groups), dim3(1024), 0, (hipStream_t)stream, (const float*)slab, chunks, C, groups, N,
                     (double)vox, (const float*)gamma, (const float*)invstd, (float*)dgamma, (float*)dbeta, (float*)coef);
  if (dtype == 0) { if (nd == 3) GPB(f16, 3, 2, g2, 256); else GPB(f16, 2, 2, g2, 256); }
  else { if (nd == 3) GPB(bf16, 3, 2, g2, 256); else GPB(bf16, 2, 2, g2, 256); }
#undef GPB
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

// iunet_maxpool_bwd (add_skip) + iunet_bn_relu_bwd of an encoder stage's second conv in two passes instead of three: dskip is
// the decoder's gradient of the skip connection (2x grid), dpool the gradient of the pooled tensor ((Do, Ho, Wo) grid);
// slab: iunet_bn_bwd_num_parts(N, 2^nd * Do*Ho*Wo) * C * 2 floats.
int iunet_bn_relu_pool_bwd(int dtype, int nd, const void* dskip, long long ds_ss, const void* dpool, long long dp_ss,
                           const void* y, long long y_ss, void* dy, long long dy_ss, const void* mean, const void* invstd,
                           const void* gamma, const void* scale, const void* shift, void* dgamma, void* dbeta, void* slab,
                           void* coef, int C, int N, int Do, int Ho, int Wo, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(dskip && dpool && y && dy && slab && coef && scale && shift, "bn_relu_pool_bwd: null pointer");
  IUNET_REQUIRE(nd == 2 || nd == 3, "bn_relu_pool_bwd: nd must be 2 or 3");
  const long long ovox = (long long)Do * Ho * Wo, vox = ovox * (nd == 3 ? 8 : 4);
  const int per_block = BN_POOL_PER_BLOCK / (nd == 3 ? 8 : 4);             // pooled voxels per workgroup (fewer parts than bn_relu_bwd's slab holds)
  const int chunks = (int)((ovox + per_block - 1) / per_block);
  IUNET_REQUIRE(chunks * N <= iunet_bn_bwd_num_parts(N, vox), "bn_relu_pool_bwd: slab part count");
  dim3 g1(chunks, C / 8, N), g2((unsigned)((ovox + 255) / 256), C / 8, N);
#define BPB(TT, NDV, PASSV, GRID, PB) hipLaunchKernelGGL((bn_pool_bwd_kernel<TT, NDV, PASSV>), GRID, dim3(256), 0, (hipStream_t)stream, \
    (const TT*)dskip, ds_ss, (const TT*)dpool, dp_ss, (const TT*)y, y_ss, (TT*)dy, dy_ss, (const float*)mean, (const float*)invstd, \
    (const float*)coef, (const float*)scale, (const float*)shift, C, Do, Ho, Wo, PB, (float*)slab)
  if (dtype == 0) { if (nd == 3) BPB(f16, 3, 1, g1, per_block); else BPB(f16, 2, 1, g1, per_block); }
  else { if (nd == 3) BPB(bf16, 3, 1, g1, per_block); else BPB(bf16, 2, 1, g1, per_block); }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, (const float*)slab, chunks * N, C,
                     (double)N * (double)vox, (const float*)gamma, (const float*)invstd, (float*)dgamma, (float*)dbeta, (float*)coef);
  if (dtype == 0) { if (nd == 3) BPB(f16, 3, 2, g2, 256); else BPB(f16, 2, 2, g2, 256); }
  else { if (nd == 3) BPB(bf16, 3, 2, g2, 256); else BPB(bf16, 2, 2, g2, 256); }
#undef BPB
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_maxpool_bwd(int dtype, int nd, const void* z, long long z_ss, const void* dpool, long long dp_ss, void* dz,
                      long long dz_ss, int add_skip, int C, int N, int Do, int Ho, int Wo, void* stream) {
  DT_OK(dtype);
  const long long total = (long long)Do * Ho * Wo * (C / 8);
  dim3 grid((unsigned)((total + 255) / 256), N);
#define MPB(TT, NDV) hipLaunchKernelGGL((maxpool_bwd_kernel<TT, NDV>), grid, dim3(256), 0, (hipStream_t)stream, (const TT*)z, z_ss, (const TT*)dpool, dp_ss, (TT*)dz, dz_ss, add_skip, C / 8, Do, Ho, Wo)
  if (dtype == 0) { if (nd == 3) MPB(f16, 3); else MPB(f16, 2); } else { if (nd == 3) MPB(bf16, 3); else MPB(bf16, 2); }
#undef MPB
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_head_loss_num_parts(int N, long long vox) { return N * (int)((vox + 256 * HEAD_FWD_ITER - 1) / (256 * HEAD_FWD_ITER)); }

// partial-sum rows written by iunet_head_loss_bwd (each workgroup covers 256 * iter voxels)
int iunet_head_loss_bwd_num_parts(int N, long long vox, int ncls, int C0) {
  const int iter = (C0 / 8) * ncls <= 16 ? 8 : HEAD_BWD_WIDE_ITER;       // register kernel : LDS kernel (head_loss_bwd_wide_kernel)
  return N * (int)((vox + 256LL * iter - 1) / (256LL * iter));
}

#define HEAD_SWITCH(KERN, TT)                                                                                   \
  switch (ncls) {                                                                                               \
    case 2: hipLaunchKernelGGL((KERN<TT, 2>), grid, dim3(256), 0, (hipStream_t)stream, p); break;               \
    case 3: hipLaunchKernelGGL((KERN<TT, 3>), grid, dim3(256), 0, (hipStream_t)stream, p); break;               \
    case 4: hipLaunchKernelGGL((KERN<TT, 4>), grid, dim3(256), 0, (hipStream_t)stream, p); break;               \
    case 5: hipLaunchKernelGGL((KERN<TT, 5>), grid, dim3(256), 0, (hipStream_t)stream, p); break;               \
    case 6: hipLaunchKernelGGL((KERN<TT, 6>), grid, dim3(256), 0, (hipStream_t)stream, p); break;               \
    case 7: hipLaunchKernelGGL((KERN<TT, 7>), grid, dim3(256), 0, (hipStream_t)stream, p); break;               \
    case 8: hipLaunchKernelGGL((KERN<TT, 8>), grid, dim3(256), 0, (hipStream_t)stream, p); break;               \
    case 9: hipLaunchKernelGGL((KERN<TT, 9>), grid, dim3(256), 0, (hipStream_t)stream, p); break;               \
    default: hipLaunchKernelGGL((KERN<TT, 10>), grid, dim3(256), 0, (hipStream_t)stream, p); break;             \
  }

// forward: head + softmax + loss sums -> loss value, rounded metrics, gradient coefficients
static int head_loss_fwd_impl(int dtype, const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls,
                              const void* target, const void* weight, int tdtype, int kind, void* slab, void* out4, void* coef,
                              int N, long long vox, const void* in_scale, const void* in_shift, void* stream, int per_sample = 0) {
  DT_OK(dtype);
  IUNET_REQUIRE(x && w && bias && target && slab && out4 && coef, "head_loss_fwd: null pointer");
  IUNET_REQUIRE(ncls >= 2 && ncls <= 10, "head_loss: num_classes must be 2..10");
  IUNET_REQUIRE(kind >= 0 && kind <= 6, "head_loss: unknown loss kind %d", kind);
  IUNET_REQUIRE(tdtype == 0 || tdtype == 1, "head_loss: target dtype must be 0 (f32) or 1 (f16)");
  HeadLossParams p{};
  p.x = x; p.x_ss = x_ss; p.planes = C0 / 8; p.w = (const float*)w; p.bias = (const float*)bias;
  p.target = target; p.weight = weight; p.tdtype = tdtype; p.slab = (float*)slab; p.N = N; p.vox = vox;
  p.in_scale = (const float*)in_scale; p.in_shift = (const float*)in_shift; p.pss = per_sample ? C0 : 0;
  dim3 grid((unsigned)(iunet_head_loss_num_parts(N, vox) / N), N);
  if (dtype == 0) { HEAD_SWITCH(head_loss_fwd_kernel, f16) } else { HEAD_SWITCH(head_loss_fwd_kernel, bf16) }
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)slab,
                     iunet_head_loss_num_parts(N, vox), ncls, kind, weight != nullptr, (double)N * (double)vox,
                     (float*)out4, (float*)coef);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_head_loss_fwd(int dtype, const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls,
                        const void* target, const void* weight, int tdtype, int kind, void* slab, void* out4, void* coef,
                        int N, long long vox, void* stream) {
  return head_loss_fwd_impl(dtype, x, x_ss, C0, w, bias, ncls, target, weight, tdtype, kind, slab, out4, coef, N, vox, nullptr, nullptr, stream);
}

// the same with the head input given as relu(in_scale[c] * x + in_shift[c]) (see iunet_conv3_fwd_act): the BatchNorm + ReLU of the
// last stage conv is applied while loading, its output tensor is never written
int iunet_head_loss_fwd_act(int dtype, const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls,
                            const void* target, const void* weight, int tdtype, int kind, void* slab, void* out4, void* coef,
                            const void* in_scale, const void* in_shift, int N, long long vox, void* stream) {
  IUNET_REQUIRE(in_scale && in_shift, "head_loss_fwd_act: null scale / shift");
  return head_loss_fwd_impl(dtype, x, x_ss, C0, w, bias, ncls, target, weight, tdtype, kind, slab, out4, coef, N, vox, in_scale, in_shift, stream);
}

// ... with per-sample rows of in_scale / in_shift ([N][C0]: GroupNorm)
int iunet_head_loss_fwd_act_ps(int dtype, const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls,
                               const void* target, const void* weight, int tdtype, int kind, void* slab, void* out4, void* coef,
                               const void* in_scale, const void* in_shift, int per_sample, int N, long long vox, void* stream) {
  IUNET_REQUIRE(in_scale && in_shift, "head_loss_fwd_act: null scale / shift");
  return head_loss_fwd_impl(dtype, x, x_ss, C0, w, bias, ncls, target, weight, tdtype, kind, slab, out4, coef, N, vox, in_scale, in_shift, stream, per_sample);
}

// backward: dx (gradient wrt head input), dW/db slabs [num_parts][ncls*(C0+1)]
static int head_loss_bwd_impl(int dtype, const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls,
                              const void* target, const void* weight, int tdtype, const void* coef, float loss_scale, void* dx,
                              long long dx_ss, void* dwslab, int N, long long vox, const void* in_scale, const void* in_shift,
                              void* stream, const void* loss_scale_dev = nullptr) {
  DT_OK(dtype);
  IUNET_REQUIRE(x && w && bias && target && coef && dx && dwslab, "head_loss_bwd: null pointer");
  IUNET_REQUIRE(ncls >= 2 && ncls <= 10, "head_loss: num_classes must be 2..10");
  HeadLossParams p{};
  p.x = x; p.x_ss = x_ss; p.planes = C0 / 8; p.w = (const float*)w; p.bias = (const float*)bias;
  p.target = target; p.weight = weight; p.tdtype = tdtype; p.coef = (const float*)coef; p.loss_scale = loss_scale; p.loss_scale_dev = (const float*)loss_scale_dev;
  p.dx = dx; p.dx_ss = dx_ss; p.dwslab = (float*)dwslab; p.N = N; p.vox = vox;
  p.in_scale = (const float*)in_scale; p.in_shift = (const float*)in_shift;
  IUNET_REQUIRE(C0 == 32 || C0 == 64, "head_loss_bwd: head input must have 32 or 64 channels (got %d)", C0);
  dim3 grid((unsigned)(iunet_head_loss_bwd_num_parts(N, vox, ncls, C0) / N), N);
#define HB(TT, NC, PLN) do { if constexpr (PLN == 4 && NC <= 4) hipLaunchKernelGGL((head_loss_bwd_kernel<TT, NC, PLN>), grid, dim3(256), 0, (hipStream_t)stream, p); \
    else hipLaunchKernelGGL((head_loss_bwd_wide_kernel<TT, NC, PLN>), grid, dim3(256), 0, (hipStream_t)stream, p); } while (0)
#define HB_SWITCH(TT, PLN) switch (ncls) { case 2: HB(TT, 2, PLN); break; case 3: HB(TT, 3, PLN); break; case 4: HB(TT, 4, PLN); break; \
    case 5: HB(TT, 5, PLN); break; case 6: HB(TT, 6, PLN); break; case 7: HB(TT, 7, PLN); break; case 8: HB(TT, 8, PLN); break; \
    case 9: HB(TT, 9, PLN); break; default: HB(TT, 10, PLN); break; }
  if (dtype == 0) { if (C0 == 32) { HB_SWITCH(f16, 4) } else { HB_SWITCH(f16, 8) } }
  else { if (C0 == 32) { HB_SWITCH(bf16, 4) } else { HB_SWITCH(bf16, 8) } }
#undef HB_SWITCH
#undef HB
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_head_loss_bwd(int dtype, const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls,
                        const void* target, const void* weight, int tdtype, const void* coef, float loss_scale, void* dx,
                        long long dx_ss, void* dwslab, int N, long long vox, void* stream) {
  return head_loss_bwd_impl(dtype, x, x_ss, C0, w, bias, ncls, target, weight, tdtype, coef, loss_scale, dx, dx_ss, dwslab, N, vox,
                            nullptr, nullptr, stream);
}

// dx is then the gradient of the ACTIVATION relu(in_scale * x + in_shift) (what the BatchNorm backward of that layer takes)
int iunet_head_loss_bwd_act(int dtype, const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls,
                            const void* target, const void* weight, int tdtype, const void* coef, float loss_scale, void* dx,
                            long long dx_ss, void* dwslab, const void* in_scale, const void* in_shift, int N, long long vox,
                            void* stream) {
  IUNET_REQUIRE(in_scale && in_shift, "head_loss_bwd_act: null scale / shift");
  return head_loss_bwd_impl(dtype, x, x_ss, C0, w, bias, ncls, target, weight, tdtype, coef, loss_scale, dx, dx_ss, dwslab, N, vox,
                            in_scale, in_shift, stream);
}

int iunet_reduce_slab(void* slab, int nparts, long long n, void* out, float alpha, int accumulate, void* stream) {
  IUNET_REQUIRE(slab && out, "reduce_slab: null pointer");
  if (n <= 16384 && nparts > 16) {       // narrow slab (head / bias gradients: tens of columns, hundreds of rows): one launch, 64 row groups
    hipLaunchKernelGGL(reduce_slab_tree_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)slab, nparts, n, (float*)out, alpha, accumulate);
    IUNET_CHECK_HIP(hipGetLastError());
    return IUNET_OK;
  }
  const int G = 64;
  if (nparts > 2 * G) {      // wide slab: fold it to G rows first (in place; the slab is scratch)
    hipLaunchKernelGGL(fold_slab_kernel, dim3((unsigned)((n + 255) / 256), G), dim3(256), 0, (hipStream_t)stream,
                       (float*)slab, nparts, n, G);
    nparts = G;
  }
  hipLaunchKernelGGL(reduce_slab_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)slab, nparts, n, (float*)out, alpha, accumulate);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_check_finite(const void* g, long long n, void* flag, void* stream) {
  hipLaunchKernelGGL(check_finite_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, (const float*)g, n, (int*)flag);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

// torch.optim.AdamW(params, lr) defaults of unet.py:71-73: betas (0.9, 0.999), eps 1e-8, wd 1e-2.
int iunet_adamw_step(void* p, const void* g, void* m, void* v, long long n, float lr, float b1, float b2, float eps,
                     float wd, int step, float grad_scale_inv, const void* skip_flag, void* stream) {
  IUNET_REQUIRE(p && g && m && v && step >= 1, "adamw: bad arguments");
  const float bc1 = 1.f - powf(b1, (float)step);
  const float bc2s = sqrtf(1.f - powf(b2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float*)p,
                     (const float*)g, (float*)m, (float*)v, n, lr, b1, b2, eps, wd, bc1, bc2s, grad_scale_inv,
                     (const int*)skip_flag);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

/* ---- training state on the device: loss scale, step count and the overflow back-off without a host read per step ------------- */
int iunet_train_state_init(void* state, float loss_scale, int dynamic, void* stream) {
  IUNET_REQUIRE(state && loss_scale > 0.f, "train_state_init: bad arguments");
  float h[8] = {loss_scale, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  ((int*)h)[7] = dynamic ? 1 : 0;
  IUNET_CHECK_HIP(hipMemcpyAsync(state, h, sizeof(h), hipMemcpyHostToDevice, (hipStream_t)stream));
  IUNET_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));          // (h is a stack buffer; initialisation is not on the hot path)
  return IUNET_OK;
}
/* head + loss backward with the loss scale read from state[0] (in_scale / in_shift null: the head input is the stored activation) */
int iunet_head_loss_bwd_dev(int dtype, const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls,
                            const void* target, const void* weight, int tdtype, const void* coef, const void* state, void* dx,
                            long long dx_ss, void* dwslab, const void* in_scale, const void* in_shift, int N, long long vox,
                            void* stream) {
  IUNET_REQUIRE(state != nullptr, "head_loss_bwd_dev: null state");
  IUNET_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "head_loss_bwd_dev: in_scale and in_shift come together");
  return head_loss_bwd_impl(dtype, x, x_ss, C0, w, bias, ncls, target, weight, tdtype, coef, 0.f, dx, dx_ss, dwslab, N, vox,
                            in_scale, in_shift, stream, state);
}
/* [r5] head backward + the last stage conv's BatchNorm + ReLU backward in two passes over that conv's raw output y (head_bn_bwd_kernel):
 * dwslab as iunet_head_loss_bwd ([iunet_head_loss_bwd_num_parts][ncls * 33]), dgamma / dbeta / bncoef as iunet_bn_relu_bwd, dy = the
 * gradient of y.  bnslab: iunet_bn_bwd_num_parts(N, vox) * 2 C0 floats; dl_scratch: N * vox * ncls floats (the logit gradients between the
 * two passes).  loss scale: state[0] when state is given, else loss_scale. */
int iunet_head_bn_bwd_ok(int C0, int ncls) { return (C0 == 32 || C0 == 64) && ncls >= 2 && ncls <= 4; }
static int head_norm_bwd_impl(int dtype, const void* y, long long y_ss, int C0, const void* w, const void* bias, int ncls, const void* target,
                              const void* weight, int tdtype, const void* coef, float loss_scale, const void* state, const void* scale,
                              const void* shift, const void* mean, const void* invstd, const void* gamma, void* dgamma, void* dbeta, void* dy,
                              long long dy_ss, void* dwslab, void* bnslab, void* bncoef, void* dl_scratch, int N, long long vox, void* stream,
                              int groups /* 0: BatchNorm ([C0] parameters); > 0: GroupNorm ([N][C0] rows) */) {
  DT_OK(dtype);
  IUNET_REQUIRE(y && w && bias && target && coef && scale && shift && mean && invstd && gamma && dgamma && dbeta && dy && dwslab && bnslab && bncoef && dl_scratch,
                "head_bn_bwd: null pointer");
  IUNET_REQUIRE(iunet_head_bn_bwd_ok(C0, ncls), "head_bn_bwd: 32 or 64 head input channels and 2..4 classes (got %d, %d): run iunet_head_loss_bwd + iunet_bn_relu_bwd", C0, ncls);
  IUNET_REQUIRE(N > 0 && vox > 0, "head_bn_bwd: N %d, %lld voxels", N, vox);
  HeadBnBwdParams p{};
  p.y = y; p.y_ss = y_ss; p.w = (const float*)w; p.bias = (const float*)bias; p.target = target; p.weight = weight; p.tdtype = tdtype;
  p.coef = (const float*)coef; p.loss_scale = loss_scale; p.loss_scale_dev = (const float*)state;
  p.scale = (const float*)scale; p.shift = (const float*)shift; p.mean = (const float*)mean; p.invstd = (const float*)invstd;
  p.bncoef = (const float*)bncoef; p.bnslab = (float*)bnslab; p.dwslab = (float*)dwslab; p.dy = dy; p.dy_ss = dy_ss;
  p.dl = (float*)dl_scratch;
  p.vox = vox; p.per_block = BN_BWD_PER_BLOCK; p.pss = groups > 0 ? C0 : 0;
  static_assert(BN_BWD_PER_BLOCK == 256 * 8, "head_bn_bwd: one row per block for both slabs (iunet_head_loss_bwd_num_parts = iunet_bn_bwd_num_parts)");
  const int chunks = (int)((vox + p.per_block - 1) / p.per_block);
  dim3 grid(chunks, N);
#define HBB(TT, PLN) switch (ncls) { case 2: hipLaunchKernelGGL((head_bn_bwd_kernel<TT, 2, 1, PLN>), grid, dim3(PLN * 64), 0, (hipStream_t)stream, p); break; \
    case 3: hipLaunchKernelGGL((head_bn_bwd_kernel<TT, 3, 1, PLN>), grid, dim3(PLN * 64), 0, (hipStream_t)stream, p); break; \
    default: hipLaunchKernelGGL((head_bn_bwd_kernel<TT, 4, 1, PLN>), grid, dim3(PLN * 64), 0, (hipStream_t)stream, p); break; }
  if (dtype == 0) { if (C0 == 32) { HBB(f16, 4) } else { HBB(f16, 8) } } else { if (C0 == 32) { HBB(bf16, 4) } else { HBB(bf16, 8) } }
#undef HBB
  if (groups > 0)      // the rows of pass 1 are [sample][chunk][C0][2]: gn_bwd_finalize_kernel's slab
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(groups), dim3(1024), 0, (hipStream_t)stream, (const float*)bnslab, chunks, C0, groups, N,
                       (double)vox, (const float*)gamma, (const float*)invstd, (float*)dgamma, (float*)dbeta, (float*)bncoef);
  else
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C0), dim3(256), 0, (hipStream_t)stream, (const float*)bnslab, chunks * N, C0,
                       (double)N * (double)vox, (const float*)gamma, (const float*)invstd, (float*)dgamma, (float*)dbeta, (float*)bncoef);
  dim3 g2((unsigned)((vox + 511) / 512), C0 / 8, N);
#define HBA(TT, PLN, GNV) switch (ncls) { case 2: hipLaunchKernelGGL((head_bn_apply_kernel<TT, 2, PLN, GNV>), g2, dim3(256), 0, (hipStream_t)stream, p); break; \
    case 3: hipLaunchKernelGGL((head_bn_apply_kernel<TT, 3, PLN, GNV>), g2, dim3(256), 0, (hipStream_t)stream, p); break; \
    default: hipLaunchKernelGGL((head_bn_apply_kernel<TT, 4, PLN, GNV>), g2, dim3(256), 0, (hipStream_t)stream, p); break; }
#define HBA2(TT, PLN) if (groups > 0) { HBA(TT, PLN, true) } else { HBA(TT, PLN, false) }
  if (dtype == 0) { if (C0 == 32) { HBA2(f16, 4) } else { HBA2(f16, 8) } } else { if (C0 == 32) { HBA2(bf16, 4) } else { HBA2(bf16, 8) } }
#undef HBA2
#undef HBA
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}
int iunet_head_bn_bwd(int dtype, const void* y, long long y_ss, int C0, const void* w, const void* bias, int ncls, const void* target,
                      const void* weight, int tdtype, const void* coef, float loss_scale, const void* state, const void* scale,
                      const void* shift, const void* mean, const void* invstd, const void* gamma, void* dgamma, void* dbeta, void* dy,
                      long long dy_ss, void* dwslab, void* bnslab, void* bncoef, void* dl_scratch, int N, long long vox, void* stream) {
  return head_norm_bwd_impl(dtype, y, y_ss, C0, w, bias, ncls, target, weight, tdtype, coef, loss_scale, state, scale, shift, mean, invstd, gamma, dgamma,
                            dbeta, dy, dy_ss, dwslab, bnslab, bncoef, dl_scratch, N, vox, stream, 0);
}
/* the GroupNorm form: scale / shift / mean / invstd are [N][C0] rows, bncoef N * C0 * 3 floats (iunet_gn_relu_bwd's), C0 <= 1024 / groups as there */
int iunet_head_gn_bwd(int dtype, const void* y, long long y_ss, int C0, const void* w, const void* bias, int ncls, const void* target,
                      const void* weight, int tdtype, const void* coef, float loss_scale, const void* state, const void* scale,
                      const void* shift, const void* mean, const void* invstd, const void* gamma, int groups, void* dgamma, void* dbeta, void* dy,
                      long long dy_ss, void* dwslab, void* bnslab, void* bncoef, void* dl_scratch, int N, long long vox, void* stream) {
  IUNET_REQUIRE(groups > 0 && C0 % groups == 0, "head_gn_bwd: %d channels do not split into %d groups", C0, groups);
  return head_norm_bwd_impl(dtype, y, y_ss, C0, w, bias, ncls, target, weight, tdtype, coef, loss_scale, state, scale, shift, mean, invstd, gamma, dgamma,
                            dbeta, dy, dy_ss, dwslab, bnslab, bncoef, dl_scratch, N, vox, stream, groups);
}
/* dW [ncls][C0], db [ncls] of the head from the reduced slab row of iunet_head_loss_bwd */
int iunet_head_grad_scatter(const void* row, void* dw, void* db, int ncls, int C0, void* stream) {
  IUNET_REQUIRE(row && dw && db && ncls > 0 && C0 % 8 == 0, "head_grad_scatter: bad arguments");
  const int n = ncls * C0 + ncls;
  hipLaunchKernelGGL(head_grad_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)row, (float*)dw, (float*)db, ncls, C0);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}
/* the optimiser step on the device state: overflow check of the flat gradient (fp16 training: check != 0), the step's coefficients,
 * AdamW (skipped on overflow), then the scale / step-count update.  world: ranks the gradient was summed over. */
int iunet_adamw_step_dev(void* p, const void* g, void* m, void* v, long long n, float lr, float b1, float b2, float eps, float wd,
                         void* state, int check, float world, void* stream) {
  IUNET_REQUIRE(p && g && m && v && state && n > 0 && world >= 1.f, "adamw_dev: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  IUNET_CHECK_HIP(hipMemsetAsync((int*)state + 3, 0, sizeof(int), s));
  if (check) hipLaunchKernelGGL(check_finite_kernel, dim3(1024), dim3(256), 0, s, (const float*)g, n, (int*)state + 3);
  hipLaunchKernelGGL(train_state_coef_kernel, dim3(1), dim3(1), 0, s, (float*)state, b1, b2, world);
  hipLaunchKernelGGL(adamw_dev_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, (float*)p, (const float*)g, (float*)m, (float*)v,
                     n, lr, b1, b2, eps, wd, (const float*)state);
  hipLaunchKernelGGL(train_state_update_kernel, dim3(1), dim3(1), 0, s, (float*)state);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

}  // extern "C"
