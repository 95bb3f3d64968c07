// fp32 parity form of the TRAINING step (interactive_unet/train_engine_f32.py): what the 16-bit training path is checked against on
// the device, and what `UNet(act_dtype='fp32')` trains with.  Planar fp32 tensors (precise_f32.hip's layout: C planes of [D][H][W],
// `*_ss` = elements between samples); every product and sum is an fp32 (or wider) operation, so a whole step differs from CPU
// autograd (oracle/unet_ref.py in training mode + torch.autograd) only by the order of the sums: the parameter gradients agree to
// ~1e-5 relative (tests/test_gpu_train_f32.py holds them to 1e-4).  Convolutions, transposed convolutions and the data gradients run
// on precise_f32.hip's f32-input MFMA kernel (the data gradient of a conv is the conv with the flipped / transposed operator; that of
// a transposed conv a 1x1 GEMM over the space-to-depth view of dy).  This file adds what a training step needs beyond it:
//   f32_bn_stats / f32_bn_relu_fwd / f32_bn_relu_bwd   BatchNorm with batch statistics + ReLU (unet.py:88-102 in train mode)
//   f32_maxpool_bwd                                     gradient to the first maximum of each window
//   f32_wgrad                                           weight gradient on v_mfma_f32_16x16x4_f32, voxels as k, slab + fixed-order reduce
//   f32_head_loss_fwd / _bwd                            1x1 head + softmax + metrics.py loss sums, and their gradient
//   f32_channel_sum                                     bias gradients
// A checking mode: simple, deterministic (no atomics), not tuned -- 1/16 of the 16-bit matrix rate at best.
#include "common.h"

int iunet_loss_finalize_launch(const float* slab, int nparts, int ncls, int kind, int has_weight, double nvox_total, float* out4,
                               float* coef, hipStream_t stream);

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double block_sum_d(double v, double* red /* [256] */) {
  __syncthreads();
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  return red[0];
}

// ------------------------------------------------------------------ BatchNorm (batch statistics) + ReLU
// one workgroup per channel, two passes (mean, then the sum of squared deviations): mean, biased variance, std = sqrt(var + eps)
// as the oracle computes them (torch mean / var(unbiased=False)); running statistics with momentum and the UNBIASED variance.
__global__ __launch_bounds__(256) void f32_bn_stats_kernel(const float* __restrict__ y, long long y_ss, int N, long long vox, float eps,
                                                          float momentum, float* __restrict__ mean, float* __restrict__ stdv,
                                                          float* __restrict__ run_mean, float* __restrict__ run_var) {
  __shared__ double red[256];
  const int c = blockIdx.x;
  const double M = (double)N * (double)vox;
  double s = 0.0;
  for (int n = 0; n < N; ++n) {
    const float* p = y + n * y_ss + (long long)c * vox;
    for (long long i = threadIdx.x; i < vox; i += 256) s += (double)p[i];
  }
  const double mu = block_sum_d(s, red) / M;
  double s2 = 0.0;
  for (int n = 0; n < N; ++n) {
    const float* p = y + n * y_ss + (long long)c * vox;
    for (long long i = threadIdx.x; i < vox; i += 256) { const double d = (double)p[i] - mu; s2 += d * d; }
  }
  const double var = block_sum_d(s2, red) / M;
  if (threadIdx.x == 0) {
    mean[c] = (float)mu;
    stdv[c] = sqrtf((float)var + eps);
    if (run_mean) {
      run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * (float)mu;
      run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (float)(var * (M > 1.0 ? M / (M - 1.0) : 1.0));
    }
  }
}

// z = relu(((y - mean) / std) * gamma + beta): the oracle's operation order, each operation rounded on its own
__global__ __launch_bounds__(256) void f32_bn_relu_fwd_kernel(const float* __restrict__ y, long long y_ss, float* __restrict__ z,
                                                             long long z_ss, const float* __restrict__ mean,
                                                             const float* __restrict__ stdv, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int C, long long vox) {
#pragma clang fp contract(off)
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)C * vox) return;
  const int n = blockIdx.y, c = (int)(i / vox);
  const float t = (y[n * y_ss + i] - mean[c]) / stdv[c];
  z[n * z_ss + i] = fmaxf(t * gamma[c] + beta[c], 0.f);
}

// backward of z = relu(bn(y)): g = dz where z > 0; dbeta = sum g, dgamma = sum g xhat, dy = gamma / std * (g - dbeta / M - xhat dgamma / M)
// (the gradient of the batch-statistics BatchNorm: mean and variance depend on y).  One workgroup per channel, two passes.
__global__ __launch_bounds__(256) void f32_bn_relu_bwd_kernel(const float* __restrict__ dz, long long dz_ss, const float* __restrict__ y,
                                                             long long y_ss, float* __restrict__ dy, long long dy_ss,
                                                             const float* __restrict__ mean, const float* __restrict__ stdv,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int N, long long vox) {
#pragma clang fp contract(off)
  __shared__ double red[256];
  const int c = blockIdx.x;
  const float mu = mean[c], sd = stdv[c], ga = gamma[c], be = beta[c];
  const double M = (double)N * (double)vox;
  double s1 = 0.0, s2 = 0.0;
  for (int n = 0; n < N; ++n) {
    const float* yp = y + n * y_ss + (long long)c * vox;
    const float* gp = dz + n * dz_ss + (long long)c * vox;
    for (long long i = threadIdx.x; i < vox; i += 256) {
      const float xh = (yp[i] - mu) / sd;
      const float g = (xh * ga + be) > 0.f ? gp[i] : 0.f;
      s1 += (double)g; s2 += (double)g * (double)xh;
    }
  }
  const double S1 = block_sum_d(s1, red), S2 = block_sum_d(s2, red);
  if (threadIdx.x == 0) { dbeta[c] = (float)S1; dgamma[c] = (float)S2; }
  const float m1 = (float)(S1 / M), m2 = (float)(S2 / M), k = ga / sd;
  for (int n = 0; n < N; ++n) {
    const float* yp = y + n * y_ss + (long long)c * vox;
    const float* gp = dz + n * dz_ss + (long long)c * vox;
    float* op = dy + n * dy_ss + (long long)c * vox;
    for (long long i = threadIdx.x; i < vox; i += 256) {
      const float xh = (yp[i] - mu) / sd;
      const float g = (xh * ga + be) > 0.f ? gp[i] : 0.f;
      op[i] = k * (g - m1 - xh * m2);
    }
  }
}

// ------------------------------------------------------------------ max-pool 2^d backward
// dz (+)= dpool at the FIRST maximum of each window in scan order (what torch's max_pool backward routes to), 0 elsewhere
template <int ND>
__global__ __launch_bounds__(256) void f32_maxpool_bwd_kernel(const float* __restrict__ z, long long z_ss, const float* __restrict__ dpool,
                                                             long long dp_ss, float* __restrict__ dz, long long dz_ss, int C, int Do,
                                                             int Ho, int Wo, int accumulate) {
  const long long ovox = (long long)Do * Ho * Wo, total = ovox * C;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int n = blockIdx.y;
  const int c = (int)(i / ovox);
  const long long r = i - (long long)c * ovox;
  const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), oz = (int)(r / ((long long)Wo * Ho));
  const int Di = ND == 3 ? Do * 2 : 1, Hi = Ho * 2, Wi = Wo * 2;
  const long long ivox = (long long)Di * Hi * Wi;
  const float* zp = z + n * z_ss + (long long)c * ivox;
  float* gp = dz + n * dz_ss + (long long)c * ivox;
  const float g = dpool[n * dp_ss + i];
  float m = -INFINITY;
  long long arg = 0;
#pragma unroll
  for (int a = 0; a < (ND == 3 ? 2 : 1); ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const long long o = ((long long)(ND == 3 ? oz * 2 + a : 0) * Hi + oy * 2 + b) * Wi + ox * 2 + cc;
        const float v = zp[o];
        if (v > m) { m = v; arg = o; }
      }
#pragma unroll
  for (int a = 0; a < (ND == 3 ? 2 : 1); ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const long long o = ((long long)(ND == 3 ? oz * 2 + a : 0) * Hi + oy * 2 + b) * Wi + ox * 2 + cc;
        const float v = o == arg ? g : 0.f;
        gp[o] = accumulate ? gp[o] + v : v;
      }
}

// ------------------------------------------------------------------ weight gradient
// dW[co][ci][tap] = sum over samples and voxels of dy[co][v] * x[ci][v + tap - 1] on v_mfma_f32_16x16x4_f32: A = dy (rows = 16 couts,
// k = 4 consecutive x voxels), B = the tap-shifted x (cols = 16 cins).  Workgroup = 4 waves on one (32 couts x 16 cins) block of
// the operator; it walks its share of the voxel tiles (3-D 4 x 4 x 16, 2-D 16 x 16), stages the dy tile and the x halo tile of the
// tile in LDS (planar, odd plane strides) and every wave accumulates the taps t = wave, wave + 4, ... in registers.  Output: one slab
// row [Cout][Cin][TAPS] per split; iunet_reduce_slab sums the rows in a fixed order.  TAPS = 1: pointwise (transposed-conv and head
// weight gradients).
struct F32WgradParams {
  const float* x; long long x_ss;        // [N][Cin][vox]
  const float* dy; long long dy_ss;      // [N][Cout][vox]
  float* slab;                           // [splits][Cout][Cin][TAPS]
  int N, D, H, W, Cin, Cout, splits;
};

template <int ND, int TAPS>
__global__ __launch_bounds__(256) void f32_wgrad_kernel(F32WgradParams p) {
  constexpr int TZ = ND == 3 ? 4 : 1, TY = ND == 3 ? 4 : 16, TX = 16;
  constexpr int HALO = TAPS == 1 ? 0 : 1, HZ = ND == 3 ? HALO : 0;
  constexpr int PZ = TZ + 2 * HZ, PY = TY + 2 * HALO, PX = TX + 2 * HALO, NPIX = PZ * PY * PX, NVOX = TZ * TY * TX;
  constexpr int XS = NPIX | 1, YS = NVOX + 1;               // odd plane strides (words)
  constexpr int NT = (TAPS + 3) / 4;                         // taps per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                                          // [16][XS]
  float* ys = smem + 16 * XS;                                // [32][YS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
  const int cib = blockIdx.x, cob = blockIdx.y, split = blockIdx.z;
  const int tilesZ = (p.D + TZ - 1) / TZ, tilesY = (p.H + TY - 1) / TY, tilesX = (p.W + TX - 1) / TX;
  const long long tps = (long long)tilesZ * tilesY * tilesX, ntiles = tps * p.N;
  const long long vox = (long long)p.D * p.H * p.W;
  f32x4v acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t) { acc[t][0] = f32x4v{0, 0, 0, 0}; acc[t][1] = f32x4v{0, 0, 0, 0}; }
  for (long long tile = split; tile < ntiles; tile += p.splits) {
    const int n = (int)(tile / tps);
    long long trem = tile - n * tps;
    const int tzi = (int)(trem / ((long long)tilesY * tilesX));
    trem -= (long long)tzi * tilesY * tilesX;
    const int tyi = (int)(trem / tilesX), txi = (int)(trem - (long long)tyi * tilesX);
    const int z0 = tzi * TZ, y0 = tyi * TY, x0 = txi * TX;
    __syncthreads();                                         // the previous tile's reads are done
    for (int e = tid; e < 16 * NPIX; e += 256) {
      const int ci = e / NPIX, pix = e - ci * NPIX;
      const int px = pix % PX, t2 = pix / PX, py = t2 % PY, pz = t2 / PY;
      const int gz = z0 + pz - HZ, gy = y0 + py - HALO, gx = x0 + px - HALO, cg = cib * 16 + ci;
      float v = 0.f;
      if (cg < p.Cin && (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
        v = p.x[n * p.x_ss + cg * vox + ((long long)gz * p.H + gy) * p.W + gx];
      xs[ci * XS + pix] = v;
    }
    for (int e = tid; e < 32 * NVOX; e += 256) {
      const int co = e / NVOX, v = e - co * NVOX;
      const int vx = v % TX, t2 = v / TX, vy = t2 % TY, vz = t2 / TY;
      const int gz = z0 + vz, gy = y0 + vy, gx = x0 + vx, cg = cob * 32 + co;
      float g = 0.f;
      if (cg < p.Cout && gz < p.D && gy < p.H && gx < p.W) g = p.dy[n * p.dy_ss + cg * vox + ((long long)gz * p.H + gy) * p.W + gx];
      ys[co * YS + v] = g;
    }
    __syncthreads();
#pragma unroll 2
    for (int kg = 0; kg < NVOX / 4; ++kg) {
      const int v = kg * 4 + kq;                              // this lane's voxel of the k-group
      const int vx = v % TX, t2 = v / TX, vy = t2 % TY, vz = t2 / TY;
      const float a0 = ys[l15 * YS + v], a1 = ys[(16 + l15) * YS + v];
      const int xb = l15 * XS + ((vz + HZ) * PY + vy + HALO) * PX + vx + HALO;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int tap = wave + 4 * t;
        if (tap < TAPS) {
          const int dz = TAPS == 27 ? tap / 9 - 1 : 0, dyy = TAPS == 1 ? 0 : (tap / 3) % 3 - 1, dx = TAPS == 1 ? 0 : tap % 3 - 1;
          const float b = xs[xb + (dz * PY + dyy) * PX + dx];
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[t][1], 0, 0, 0);
        }
      }
    }
  }
  // D[row = cout 4 (lane >> 4) + j][col = cin lane & 15]
  float* out = p.slab + (long long)split * p.Cout * p.Cin * TAPS;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tap = wave + 4 * t;
    if (tap >= TAPS) continue;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = cob * 32 + h * 16 + kq * 4 + j, ci = cib * 16 + l15;
        if (co < p.Cout && ci < p.Cin) out[((long long)co * p.Cin + ci) * TAPS + tap] = acc[t][h][j];
      }
  }
}

// ------------------------------------------------------------------ head + softmax + loss
struct F32HeadLossParams {
  const float* x; long long x_ss; int C0;
  const float* w; const float* bias;
  const void* target; const void* weight; int tdtype;
  float* slab;                   // fwd: [parts][ncls][8]
  const float* coef;             // bwd: [ncls][3]
  float* dlogits; long long dl_ss;      // bwd: [N][dl channels][vox] (the first ncls planes are written)
  float* dx; long long dx_ss;           // bwd: [N][C0][vox]
  int N; long long vox;
};

__device__ __forceinline__ float f32_load_t(const void* p, long long off, int dt) {
  return dt == 0 ? ((const float*)p)[off] : (float)((const f16*)p)[off];
}

#define F32_HEAD_ITER 8
// the 8 sums per class of train_pointwise.hip's head_loss_fwd_kernel (0 sw, 1 swy, 2 swp, 3 swyp, 4 swy log(p + eps), 5 sw ry,
// 6 sw rp, 7 sw ry rp), from fp32 features with expf / logf / correctly rounded divisions
template <int NCLS>
__global__ __launch_bounds__(256) void f32_head_loss_fwd_kernel(F32HeadLossParams p) {
  __shared__ float red[4 * NCLS * 8];
  const int n = blockIdx.y;
  float acc[NCLS][8];
#pragma unroll
  for (int c = 0; c < NCLS; ++c)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[c][k] = 0.f;
  for (int it = 0; it < F32_HEAD_ITER; ++it) {
    const long long v = ((long long)blockIdx.x * F32_HEAD_ITER + it) * 256 + threadIdx.x;
    if (v >= p.vox) break;
    const float* xin = p.x + n * p.x_ss + v;
    float l[NCLS];
#pragma unroll
    for (int c = 0; c < NCLS; ++c) l[c] = 0.f;
    for (int ch = 0; ch < p.C0; ++ch) {
      const float a = xin[(long long)ch * p.vox];
#pragma unroll
      for (int c = 0; c < NCLS; ++c) l[c] = fmaf(a, p.w[c * p.C0 + ch], l[c]);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) { l[c] = __fadd_rn(l[c], p.bias[c]); mx = fmaxf(mx, l[c]); }
    float e[NCLS], s = 0.f;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) { e[c] = expf(l[c] - mx); s += e[c]; }
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
      const float pr = __fdiv_rn(e[c], s);
      const long long to = ((long long)n * NCLS + c) * p.vox + v;
      const float y = f32_load_t(p.target, to, p.tdtype);
      const float w = p.weight ? f32_load_t(p.weight, to, p.tdtype) : 1.f;
      const float ry = rintf(y), rp = rintf(pr);
      acc[c][0] += w; acc[c][1] += w * y; acc[c][2] += w * pr; acc[c][3] += w * y * pr;
      acc[c][4] += w * y * logf(pr + 1e-12f);
      acc[c][5] += w * ry; acc[c][6] += w * rp; acc[c][7] += w * ry * rp;
    }
  }
  const long long part = (long long)n * gridDim.x + blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < NCLS; ++c)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float v = wave_sum(acc[c][k]);
      if (lane == 0) red[wave * NCLS * 8 + c * 8 + k] = v;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < NCLS * 8; i += 256)
    p.slab[part * NCLS * 8 + i] = red[i] + red[NCLS * 8 + i] + red[2 * NCLS * 8 + i] + red[3 * NCLS * 8 + i];
}

// dL/dp_c = w (A_c + B_c y) - CE_c w y / (p_c + eps) (loss_finalize_kernel's coefficients), through the softmax to the logits, and
// on to the head input: dlogits planes for the head's weight / bias gradients (f32_wgrad TAPS = 1, f32_channel_sum), dx planar
template <int NCLS>
__global__ __launch_bounds__(256) void f32_head_loss_bwd_kernel(F32HeadLossParams p) {
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
  if (v >= p.vox) return;
  const int n = blockIdx.y;
  const float* xin = p.x + n * p.x_ss + v;
  float l[NCLS];
#pragma unroll
  for (int c = 0; c < NCLS; ++c) l[c] = 0.f;
  for (int ch = 0; ch < p.C0; ++ch) {
    const float a = xin[(long long)ch * p.vox];
#pragma unroll
    for (int c = 0; c < NCLS; ++c) l[c] = fmaf(a, p.w[c * p.C0 + ch], l[c]);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < NCLS; ++c) { l[c] = __fadd_rn(l[c], p.bias[c]); mx = fmaxf(mx, l[c]); }
  float pr[NCLS], s = 0.f;
#pragma unroll
  for (int c = 0; c < NCLS; ++c) { pr[c] = expf(l[c] - mx); s += pr[c]; }
  float dp[NCLS], dot = 0.f;
#pragma unroll
  for (int c = 0; c < NCLS; ++c) {
    pr[c] = __fdiv_rn(pr[c], s);
    const long long to = ((long long)n * NCLS + c) * p.vox + v;
    const float y = f32_load_t(p.target, to, p.tdtype);
    const float w = p.weight ? f32_load_t(p.weight, to, p.tdtype) : 1.f;
    dp[c] = w * (p.coef[c * 3 + 0] + p.coef[c * 3 + 1] * y) - p.coef[c * 3 + 2] * w * y / (pr[c] + 1e-12f);
    dot += pr[c] * dp[c];
  }
  float dl[NCLS];
#pragma unroll
  for (int c = 0; c < NCLS; ++c) {
    dl[c] = pr[c] * (dp[c] - dot);
    p.dlogits[n * p.dl_ss + (long long)c * p.vox + v] = dl[c];
  }
  float* dx = p.dx + n * p.dx_ss + v;
  for (int ch = 0; ch < p.C0; ++ch) {
    float g = 0.f;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) g = fmaf(dl[c], p.w[c * p.C0 + ch], g);
    dx[(long long)ch * p.vox] = g;
  }
}

// out[c] = sum over samples and voxels of t[n][c][v]: one workgroup per channel, double accumulation
__global__ __launch_bounds__(256) void f32_channel_sum_kernel(const float* __restrict__ t, long long t_ss, float* __restrict__ out, int N,
                                                             long long vox) {
  __shared__ double red[256];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int n = 0; n < N; ++n) {
    const float* p = t + n * t_ss + (long long)c * vox;
    for (long long i = threadIdx.x; i < vox; i += 256) s += (double)p[i];
  }
  s = block_sum_d(s, red);
  if (threadIdx.x == 0) out[c] = (float)s;
}

int wgrad_splits(int nd, int N, int D, int H, int W, int Cin, int Cout) {
  const int TZ = nd == 3 ? 4 : 1, TY = nd == 3 ? 4 : 16;
  const long long tiles = (long long)N * ((D + TZ - 1) / TZ) * ((H + TY - 1) / TY) * ((W + 15) / 16);
  const long long blocks = (long long)((Cin + 15) / 16) * ((Cout + 31) / 32);
  long long s = 1024 / blocks;
  if (s < 1) s = 1;
  if (s > tiles) s = tiles;
  if (s > 256) s = 256;
  return (int)s;
}

}  // namespace

extern "C" {

int iunet_f32_bn_stats(const void* y, long long y_ss, int C, int N, long long vox, float eps, float momentum, void* mean, void* stdv,
                       void* run_mean, void* run_var, void* stream) {
  IUNET_REQUIRE(y && mean && stdv, "f32_bn_stats: null pointer");
  IUNET_REQUIRE(C > 0 && N > 0 && vox > 0, "f32_bn_stats: bad shape");
  IUNET_REQUIRE((run_mean == nullptr) == (run_var == nullptr), "f32_bn_stats: running mean and variance come together");
  hipLaunchKernelGGL(f32_bn_stats_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, (const float*)y, y_ss, N, vox, eps, momentum,
                     (float*)mean, (float*)stdv, (float*)run_mean, (float*)run_var);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_f32_bn_relu_fwd(const void* y, long long y_ss, void* z, long long z_ss, const void* mean, const void* stdv, const void* gamma,
                          const void* beta, int C, int N, long long vox, void* stream) {
  IUNET_REQUIRE(y && z && mean && stdv && gamma && beta, "f32_bn_relu_fwd: null pointer");
  IUNET_REQUIRE(C > 0 && N > 0 && vox > 0, "f32_bn_relu_fwd: bad shape");
  dim3 grid((unsigned)(((long long)C * vox + 255) / 256), N);
  hipLaunchKernelGGL(f32_bn_relu_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float*)y, y_ss, (float*)z, z_ss,
                     (const float*)mean, (const float*)stdv, (const float*)gamma, (const float*)beta, C, vox);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_f32_bn_relu_bwd(const void* dz, long long dz_ss, const void* y, long long y_ss, void* dy, long long dy_ss, const void* mean,
                          const void* stdv, const void* gamma, const void* beta, void* dgamma, void* dbeta, int C, int N, long long vox,
                          void* stream) {
  IUNET_REQUIRE(dz && y && dy && mean && stdv && gamma && beta && dgamma && dbeta, "f32_bn_relu_bwd: null pointer");
  IUNET_REQUIRE(C > 0 && N > 0 && vox > 0, "f32_bn_relu_bwd: bad shape");
  hipLaunchKernelGGL(f32_bn_relu_bwd_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, (const float*)dz, dz_ss, (const float*)y, y_ss,
                     (float*)dy, dy_ss, (const float*)mean, (const float*)stdv, (const float*)gamma, (const float*)beta,
                     (float*)dgamma, (float*)dbeta, N, vox);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_f32_maxpool_bwd(int nd, const void* z, long long z_ss, const void* dpool, long long dp_ss, void* dz, long long dz_ss, int C,
                          int N, int Do, int Ho, int Wo, int accumulate, void* stream) {
  IUNET_REQUIRE(z && dpool && dz, "f32_maxpool_bwd: null pointer");
  IUNET_REQUIRE(nd == 2 || nd == 3, "f32_maxpool_bwd: nd must be 2 or 3");
  IUNET_REQUIRE(C > 0 && N > 0 && Do > 0 && Ho > 0 && Wo > 0, "f32_maxpool_bwd: bad shape");
  const long long total = (long long)Do * Ho * Wo * C;
  dim3 grid((unsigned)((total + 255) / 256), N);
  if (nd == 3) hipLaunchKernelGGL((f32_maxpool_bwd_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)z, z_ss,
                                  (const float*)dpool, dp_ss, (float*)dz, dz_ss, C, Do, Ho, Wo, accumulate);
  else hipLaunchKernelGGL((f32_maxpool_bwd_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)z, z_ss, (const float*)dpool,
                          dp_ss, (float*)dz, dz_ss, C, Do, Ho, Wo, accumulate);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

/* slab rows iunet_f32_wgrad writes (and iunet_reduce_slab must sum); taps 9 / 27 (nd 2 / 3) or 1 (pointwise) */
int iunet_f32_wgrad_splits(int nd, int N, int D, int H, int W, int Cin, int Cout) {
  if ((nd != 2 && nd != 3) || N < 1 || D < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return 0;
  return wgrad_splits(nd, N, D, H, W, Cin, Cout);
}

int iunet_f32_wgrad(int nd, const void* x, long long x_ss, const void* dy, long long dy_ss, void* slab, int N, int D, int H, int W,
                    int Cin, int Cout, int taps, void* stream) {
  IUNET_REQUIRE(x && dy && slab, "f32_wgrad: null pointer");
  IUNET_REQUIRE(nd == 2 || nd == 3, "f32_wgrad: nd must be 2 or 3");
  IUNET_REQUIRE_GRID("f32_wgrad", N, D, H, W);
  IUNET_REQUIRE(nd == 3 || D == 1, "f32_wgrad: 2-D needs D == 1");
  IUNET_REQUIRE(Cin > 0 && Cout > 0, "f32_wgrad: bad channel counts %d, %d", Cin, Cout);
  IUNET_REQUIRE(taps == 1 || taps == (nd == 3 ? 27 : 9), "f32_wgrad: taps must be 1 or 3^nd (got %d)", taps);
  F32WgradParams p;
  p.x = (const float*)x; p.x_ss = x_ss; p.dy = (const float*)dy; p.dy_ss = dy_ss; p.slab = (float*)slab;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.splits = wgrad_splits(nd, N, D, H, W, Cin, Cout);
  dim3 grid((Cin + 15) / 16, (Cout + 31) / 32, p.splits);
  const int halo = taps == 1 ? 0 : 1;
  const int npix = nd == 3 ? (4 + 2 * halo) * (4 + 2 * halo) * (16 + 2 * halo) : (16 + 2 * halo) * (16 + 2 * halo);
  const int lds = (16 * (npix | 1) + 32 * 257) * 4;
#define F32WG(NDV, TP) do { IUNET_SET_MAX_LDS((f32_wgrad_kernel<NDV, TP>), lds); \
    hipLaunchKernelGGL((f32_wgrad_kernel<NDV, TP>), grid, dim3(256), lds, (hipStream_t)stream, p); } while (0)
  if (nd == 3) { if (taps == 1) F32WG(3, 1); else F32WG(3, 27); } else { if (taps == 1) F32WG(2, 1); else F32WG(2, 9); }
#undef F32WG
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_f32_head_loss_num_parts(int N, long long vox) { return N * (int)((vox + 256 * F32_HEAD_ITER - 1) / (256 * F32_HEAD_ITER)); }

/* unet.py:88-102 on fp32 planar features: head + softmax + the loss of `kind` (0 ce, 1 dice, 2 iou, 3 mcc, 4-6 the + ce combinations;
 * metrics.py semantics over batch + spatial axes) -> out4 [loss, dice, iou, mcc on rounded tensors], coef [ncls][3] for the backward;
 * slab: iunet_f32_head_loss_num_parts x ncls x 8 floats; target / weight [N][ncls][vox], tdtype 0 f32 / 1 f16 */
int iunet_f32_head_loss_fwd(const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls, const void* target,
                            const void* weight, int tdtype, int kind, void* slab, void* out4, void* coef, int N, long long vox,
                            void* stream) {
  IUNET_REQUIRE(x && w && bias && target && slab && out4 && coef, "f32_head_loss_fwd: null pointer");
  IUNET_REQUIRE(ncls >= 2 && ncls <= 10, "f32_head_loss: num_classes must be 2..10");
  IUNET_REQUIRE(kind >= 0 && kind <= 6, "f32_head_loss: unknown loss kind %d", kind);
  IUNET_REQUIRE(tdtype == 0 || tdtype == 1, "f32_head_loss: target dtype must be 0 (f32) or 1 (f16)");
  IUNET_REQUIRE(C0 > 0 && N > 0 && vox > 0, "f32_head_loss: bad shape");
  F32HeadLossParams p{};
  p.x = (const float*)x; p.x_ss = x_ss; p.C0 = C0; p.w = (const float*)w; p.bias = (const float*)bias; p.target = target;
  p.weight = weight; p.tdtype = tdtype; p.slab = (float*)slab; p.N = N; p.vox = vox;
  const int parts = iunet_f32_head_loss_num_parts(N, vox);
  dim3 grid((unsigned)(parts / N), N);
#define F32HL(NC) case NC: hipLaunchKernelGGL((f32_head_loss_fwd_kernel<NC>), grid, dim3(256), 0, (hipStream_t)stream, p); break;
  switch (ncls) { F32HL(2) F32HL(3) F32HL(4) F32HL(5) F32HL(6) F32HL(7) F32HL(8) F32HL(9) F32HL(10) }
#undef F32HL
  IUNET_CHECK_HIP(hipGetLastError());
  return iunet_loss_finalize_launch((const float*)slab, parts, ncls, kind, weight != nullptr, (double)N * (double)vox, (float*)out4,
                                    (float*)coef, (hipStream_t)stream);
}

/* gradient of the loss wrt the logits (dlogits [N][dl_ss / vox planes][vox], the first ncls planes written) and wrt the head input
 * (dx [N][C0][vox]) */
int iunet_f32_head_loss_bwd(const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls, const void* target,
                            const void* weight, int tdtype, const void* coef, void* dlogits, long long dl_ss, void* dx, long long dx_ss,
                            int N, long long vox, void* stream) {
  IUNET_REQUIRE(x && w && bias && target && coef && dlogits && dx, "f32_head_loss_bwd: null pointer");
  IUNET_REQUIRE(ncls >= 2 && ncls <= 10, "f32_head_loss: num_classes must be 2..10");
  IUNET_REQUIRE(tdtype == 0 || tdtype == 1, "f32_head_loss: target dtype must be 0 (f32) or 1 (f16)");
  IUNET_REQUIRE(C0 > 0 && N > 0 && vox > 0, "f32_head_loss: bad shape");
  F32HeadLossParams p{};
  p.x = (const float*)x; p.x_ss = x_ss; p.C0 = C0; p.w = (const float*)w; p.bias = (const float*)bias; p.target = target;
  p.weight = weight; p.tdtype = tdtype; p.coef = (const float*)coef; p.dlogits = (float*)dlogits; p.dl_ss = dl_ss;
  p.dx = (float*)dx; p.dx_ss = dx_ss; p.N = N; p.vox = vox;
  dim3 grid((unsigned)((vox + 255) / 256), N);
#define F32HB(NC) case NC: hipLaunchKernelGGL((f32_head_loss_bwd_kernel<NC>), grid, dim3(256), 0, (hipStream_t)stream, p); break;
  switch (ncls) { F32HB(2) F32HB(3) F32HB(4) F32HB(5) F32HB(6) F32HB(7) F32HB(8) F32HB(9) F32HB(10) }
#undef F32HB
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_f32_channel_sum(const void* t, long long t_ss, void* out, int C, int N, long long vox, void* stream) {
  IUNET_REQUIRE(t && out, "f32_channel_sum: null pointer");
  IUNET_REQUIRE(C > 0 && N > 0 && vox > 0, "f32_channel_sum: bad shape");
  hipLaunchKernelGGL(f32_channel_sum_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, (const float*)t, t_ss, (float*)out, N, vox);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

}  // extern "C"
