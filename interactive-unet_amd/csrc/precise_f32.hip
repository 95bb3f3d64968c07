// fp32 parity mode of the U-Net forward (BASELINE.json north_star: "outputs match the reference CPU PyTorch path
// within 1e-3 on logits, integer-exact on the argmax class map").  The 16-bit paths round every activation to
// fp16 / bf16 in HBM, which moves the logits of the 18-conv network by 5e-3 .. 7e-2; this path keeps fp32 end to
// end: fp32 activations in HBM, fp32 weights, and the convolutions on the f32-input matrix instruction
// v_mfma_f32_16x16x4_f32, whose result is bit for bit a k-ordered fmaf chain (cdna_hip_programming.md "FP32-input
// MFMA") -- so the only difference to the CPU oracle is the order of the sums.  157 TFLOP/s peak, 1/16 of bf16:
// a checking mode, not the throughput path.
//
// Layout (this mode only): PLANAR fp32, a tensor of C channels = C planes [D][H][W]; consecutive samples are
// `sample stride` elements apart, so the two halves of a skip-concat buffer are views.  Planar because the B operand
// of 16x16x4 is ONE f32 per lane (16 consecutive x voxels of one input channel): x-innermost planes make the LDS halo
// image a straight copy and every global access a run along x.
//
// One kernel serves every GEMM-shaped layer:
//   f32_conv_kernel<ND, false>   3^d conv, pad 1 (+ folded-BatchNorm bias, ReLU); the input is read through element
//                                strides and a dtype code, so the first conv takes the caller's uint8 / NCHW / 2.5-D
//                                view directly (channels beyond Cin are zero rows of the packed operator)
//   f32_conv_kernel<ND, true>    ConvTranspose k2 s2 = 2^d 1x1 GEMMs (blockIdx.z = output position) with a stride-2
//                                scatter store
// plus a max-pool and the 1x1 head + softmax / argmax.
#include "common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct F32ConvParams {
  const void* x;                       // input, generic element strides
  long long sN, sC, sD, sH, sW;
  int in_dtype;                        // 0 f32, 1 f16, 2 u8 (x / 255, predict.py:30), 3 bf16
  float* y; long long y_ss;            // planar output [Cout][Do][Ho][Wo]
  const float* w;                      // packed operator (f32_pack_conv_kernel)
  const float* bias;                   // [Cout] or null
  int N, D, H, W, Cin, Cout, relu;     // D, H, W: input grid (the output grid of a transposed conv is 2x)
  int dense;                           // CT instantiation only: 1 = plain 1x1 conv (output voxel = input voxel, one position): the
                                       // pointwise GEMMs of the fp32 training mode (transposed-conv data gradient)
};

__device__ __forceinline__ float f32_load_in(const void* p, long long off, int dt) {
  switch (dt) {
    case 0: return ((const float*)p)[off];
    case 1: return (float)((const f16*)p)[off];
    case 2: return __fdiv_rn((float)((const unsigned char*)p)[off], 255.0f);
    default: return (float)((const bf16*)p)[off];
  }
}

// Workgroup = 4 waves = a tile of 16 rows x 16 x voxels (3-D: 4 z x 4 y rows; 2-D: 16 y rows) x 32 output channels;
// wave w owns rows 4w .. 4w+3: 2 cout tiles x 4 rows of 16x16 accumulators.  The input channels go by in chunks of 8:
// halo tile [8][PZ][PY][PX] (plane stride = 16 mod 32 words: the two channel planes a 32-lane group reads sit on
// disjoint banks) + the chunk's operator [tap][k pair][cout tile][k parity][16] (the 32 lanes of a group read 32
// consecutive words), both staged global -> registers -> LDS with the next chunk's loads in flight during the MFMAs.
template <int ND, bool CT>
__global__ __launch_bounds__(256) void f32_conv_kernel(F32ConvParams p) {
  constexpr int TZ = ND == 3 ? 4 : 1, TY = ND == 3 ? 4 : 16, TX = 16;
  constexpr int HALO = CT ? 0 : 1, HZ = ND == 3 ? HALO : 0;
  constexpr int PZ = TZ + 2 * HZ, PY = TY + 2 * HALO, PX = TX + 2 * HALO, NPIX = PZ * PY * PX;
  constexpr int PS = ((NPIX + 15) / 32) * 32 + 16;
  constexpr int TAPS = CT ? 1 : (ND == 3 ? 27 : 9);
  constexpr int CK = 8, WCH = TAPS * CK * 32;
  constexpr int NP = (NPIX + 255) / 256;            // halo positions per thread
  constexpr int NW = (WCH / 4 + 255) / 256;         // float4 weight loads per thread
  __shared__ float xs[CK * PS];
  __shared__ __attribute__((aligned(16))) float wsm[WCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15;
  const int tilesZ = (p.D + TZ - 1) / TZ, tilesY = (p.H + TY - 1) / TY, tilesX = (p.W + TX - 1) / TX;
  const int tps = tilesZ * tilesY * tilesX;
  const int n = blockIdx.x / tps;
  int trem = blockIdx.x - n * tps;
  const int tzi = trem / (tilesY * tilesX);
  trem -= tzi * tilesY * tilesX;
  const int tyi = trem / tilesX, txi = trem - tyi * tilesX;
  const int z0 = tzi * TZ, y0 = tyi * TY, x0 = txi * TX;
  const int cob = blockIdx.y, pos = CT ? blockIdx.z : 0;
  const int nchunks = (p.Cin + CK - 1) / CK;

  // halo positions of this thread (the same for every channel chunk)
  long long goff[NP];
  bool gok[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int pix = tid + 256 * i;
    const int px = pix % PX, t2 = pix / PX, py = t2 % PY, pz = t2 / PY;
    const int gz = z0 + pz - HZ, gy = y0 + py - HALO, gx = x0 + px - HALO;
    gok[i] = pix < NPIX && (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    goff[i] = n * p.sN + gz * p.sD + gy * p.sH + gx * p.sW;
  }
  const f32x4v* wbase = (const f32x4v*)p.w + ((long long)(pos * (int)gridDim.y + cob) * nchunks) * (WCH / 4);

  float xr[NP][CK];
  f32x4v wr[NW];
  auto fetch = [&](int c) {
#pragma unroll
    for (int ci = 0; ci < CK; ++ci) {
      const int cg = c * CK + ci;
#pragma unroll
      for (int i = 0; i < NP; ++i)
        xr[i][ci] = (gok[i] && cg < p.Cin) ? f32_load_in(p.x, goff[i] + cg * p.sC, p.in_dtype) : 0.f;
    }
    const f32x4v* wc = wbase + (long long)c * (WCH / 4);
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tid + 256 * i;
      wr[i] = e < WCH / 4 ? wc[e] : f32x4v{0, 0, 0, 0};
    }
  };

  f32x4v acc[2][4];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[ct][i] = f32x4v{0, 0, 0, 0};

  // per-row LDS base of this wave's four rows (+ the lane's x and its k plane)
  int rbase[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 4 * wave + i, tz = r / TY, ty = r % TY;
    rbase[i] = (lane >> 4) * PS + (tz * PY + ty) * PX + l15;
  }

  fetch(0);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();                                   // the previous chunk's reads are done
#pragma unroll
    for (int ci = 0; ci < CK; ++ci)
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int pix = tid + 256 * i;
        if (NPIX % 256 == 0 || pix < NPIX) xs[ci * PS + pix] = xr[i][ci];
      }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tid + 256 * i;
      if (WCH / 4 % 256 == 0 || e < WCH / 4) *(f32x4v*)(wsm + 4 * e) = wr[i];
    }
    __syncthreads();
    if (c + 1 < nchunks) fetch(c + 1);                 // in flight during the MFMAs below
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int dz = ND == 3 ? tap / 9 : 0, dy = CT ? 0 : (tap / 3) % 3, dx = CT ? 0 : tap % 3;
      const int toff = (dz * PY + dy) * PX + dx;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int q = ks * 2 + (lane >> 5);
        const float a0 = wsm[((tap * 4 + q) * 2 + 0) * 32 + (lane & 31)];
        const float a1 = wsm[((tap * 4 + q) * 2 + 1) * 32 + (lane & 31)];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float b = xs[rbase[i] + ks * 4 * PS + toff];
          acc[0][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[0][i], 0, 0, 0);
          acc[1][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[1][i], 0, 0, 0);
        }
      }
    }
  }

  // epilogue: D[cout = 16 ct + 4 (lane >> 4) + j][voxel x = lane & 15]
  const bool up = CT && !p.dense;
  const int Do = up ? (ND == 3 ? 2 * p.D : 1) : p.D, Ho = up ? 2 * p.H : p.H, Wo = up ? 2 * p.W : p.W;
  const long long ovox = (long long)Do * Ho * Wo;
  const int pa = ND == 3 ? (pos >> 2) & 1 : 0, pb = (pos >> 1) & 1, pc = pos & 1;
  float* yout = p.y + n * p.y_ss;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 4 * wave + i, tz = r / TY, ty = r % TY;
    const int gz = z0 + tz, gy = y0 + ty, gx = x0 + l15;
    if (gz >= p.D || gy >= p.H || gx >= p.W) continue;
    const int oz = up ? (ND == 3 ? 2 * gz + pa : 0) : gz, oy = up ? 2 * gy + pb : gy, ox = up ? 2 * gx + pc : gx;
    const long long o = ((long long)oz * Ho + oy) * Wo + ox;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = cob * 32 + ct * 16 + (lane >> 4) * 4 + j;
        float v = acc[ct][i][j];
        if (p.bias) v = __fadd_rn(v, p.bias[co]);
        if (p.relu) v = fmaxf(v, 0.f);
        yout[co * ovox + o] = v;
      }
  }
}

// fp32 master weights -> the operator order above.  conv: w [Cout][Cin][taps]; transposed: w [Cin][Cout][npos].
// dst [npos or 1][Cout/32][chunks][taps or 1][4 k pairs][2 cout tiles][2 k parity][16]; channels >= Cin are zeros.
// A non-null gamma folds an eval-mode BatchNorm exactly as the oracle does (separately rounded fp32 operations):
// a = gamma / sqrt(var + eps), w' = w * a, bias_out = beta - mean * a.
__global__ void f32_pack_conv_kernel(const float* __restrict__ w, float* __restrict__ dst, float* __restrict__ bias_out,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ mean, const float* __restrict__ var, float eps, int Cout,
                                     int Cin, int taps, int transposed) {
  const int nchunks = (Cin + 7) / 8, ncob = Cout / 32;
  const int T = transposed ? 1 : taps, NPOS = transposed ? taps : 1;
  const long long total = (long long)NPOS * ncob * nchunks * T * 256;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int c16 = r & 15; r >>= 4;
    const int kk = r & 1; r >>= 1;
    const int ct = r & 1; r >>= 1;
    const int q = r & 3; r >>= 2;
    const int tap = r % T; r /= T;
    const int chunk = r % nchunks; r /= nchunks;
    const int cob = r % ncob;
    const int pos = (int)(r / ncob);
    const int co = cob * 32 + ct * 16 + c16, ci = chunk * 8 + q * 2 + kk;
    float v = 0.f;
    if (ci < Cin) {
      v = transposed ? w[((long long)ci * Cout + co) * taps + pos] : w[((long long)co * Cin + ci) * taps + tap];
      if (gamma) v = __fmul_rn(v, __fdiv_rn(gamma[co], __fsqrt_rn(__fadd_rn(var[co], eps))));
    }
    dst[i] = v;
  }
  if (gamma && bias_out)
    for (int co = blockIdx.x * blockDim.x + threadIdx.x; co < Cout; co += gridDim.x * blockDim.x) {
      const float a = __fdiv_rn(gamma[co], __fsqrt_rn(__fadd_rn(var[co], eps)));
      bias_out[co] = __fsub_rn(beta[co], __fmul_rn(mean[co], a));
    }
}

template <int ND>
__global__ __launch_bounds__(256) void f32_maxpool_kernel(const float* __restrict__ x, long long x_ss, float* __restrict__ y,
                                                          long long y_ss, int C, int Do, int Ho, int Wo) {
  const long long ovox = (long long)Do * Ho * Wo, total = ovox * C;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int n = blockIdx.y;
  const int c = (int)(i / ovox);
  const long long r = i - (long long)c * ovox;
  const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), oz = (int)(r / ((long long)Wo * Ho));
  const int Di = ND == 3 ? Do * 2 : 1, Hi = Ho * 2, Wi = Wo * 2;
  const float* xp = x + n * x_ss + (long long)c * Di * Hi * Wi;
  float m = -INFINITY;
#pragma unroll
  for (int a = 0; a < (ND == 3 ? 2 : 1); ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int z = ND == 3 ? oz * 2 + a : 0;
      const float2 v = *(const float2*)(xp + ((long long)z * Hi + oy * 2 + b) * Wi + ox * 2);
      m = fmaxf(m, fmaxf(v.x, v.y));
    }
  y[n * y_ss + i] = m;
}

struct F32HeadParams {
  const float* x; long long x_ss; int C0;
  const float* w; const float* bias;
  float* logits; float* probs; unsigned char* cls;
  long long oN, oC, oD, oH, oW;
  float divisor; int accumulate;
  int N, D, H, W;
};

// 1x1 head + softmax (unet.py:63-69) + class map (predict.py:38) from planar fp32 features; output contract of
// iunet_head_fwd (strided logits / probabilities, 2.5-D accumulation of predict.py:101-110).
template <int NCLS>
__global__ __launch_bounds__(256) void f32_head_kernel(F32HeadParams p) {
  const long long vox = (long long)p.D * p.H * p.W;
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
  if (v >= vox) return;
  const int n = blockIdx.y;
  const float* xin = p.x + n * p.x_ss + v;
  float l[NCLS];
#pragma unroll
  for (int c = 0; c < NCLS; ++c) l[c] = 0.f;
  for (int ch = 0; ch < p.C0; ++ch) {
    const float a = xin[(long long)ch * vox];
#pragma unroll
    for (int c = 0; c < NCLS; ++c) l[c] = fmaf(a, p.w[c * p.C0 + ch], l[c]);
  }
#pragma unroll
  for (int c = 0; c < NCLS; ++c) l[c] = __fadd_rn(l[c], p.bias[c]);
  const int gx = (int)(v % p.W), gy = (int)((v / p.W) % p.H), gz = (int)(v / ((long long)p.W * p.H));
  const long long obase = n * p.oN + gz * p.oD + gy * p.oH + gx * p.oW;
  float mx = l[0];
#pragma unroll
  for (int c = 1; c < NCLS; ++c) mx = fmaxf(mx, l[c]);
  if (p.logits) {
#pragma unroll
    for (int c = 0; c < NCLS; ++c) p.logits[obase + c * p.oC] = l[c];
  }
  float e[NCLS], s = 0.f;
#pragma unroll
  for (int c = 0; c < NCLS; ++c) { e[c] = expf(l[c] - mx); s += e[c]; }
  float pm = __fdiv_rn(e[0], s); int am = 0;
  float pr[NCLS];
  pr[0] = pm;
#pragma unroll
  for (int c = 1; c < NCLS; ++c) { pr[c] = __fdiv_rn(e[c], s); if (pr[c] > pm) { pm = pr[c]; am = c; } }
  if (p.cls) p.cls[n * vox + v] = (unsigned char)am;
  if (p.probs) {
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
      float* o = p.probs + obase + c * p.oC;
      float r = p.accumulate ? __fadd_rn(*o, pr[c]) : pr[c];
      if (p.divisor != 1.0f) r = __fdiv_rn(r, p.divisor);
      *o = r;
    }
  }
}

}  // namespace

// ------------------------------------------------------------------ host launchers
long long iunet_f32_pack_size(int Cout, int Cin, int taps) {
  return (long long)(Cout / 32) * ((Cin + 7) / 8) * taps * 256;      // the same for a transposed conv (taps = npos)
}

int iunet_f32_pack_launch(const float* w, float* dst, float* bias_out, const float* gamma, const float* beta,
                          const float* mean, const float* var, float eps, int Cout, int Cin, int taps, int transposed,
                          hipStream_t stream) {
  const long long total = iunet_f32_pack_size(Cout, Cin, taps);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(f32_pack_conv_kernel, dim3(blocks), dim3(256), 0, stream, w, dst, bias_out, gamma, beta, mean, var, eps,
                     Cout, Cin, taps, transposed);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_f32_conv_launch(int nd, const void* x, int in_dtype, const long long* st, float* y, long long y_ss, const float* wpk,
                          const float* bias, int N, int D, int H, int W, int Cin, int Cout, int relu, int transposed,
                          hipStream_t stream) {
  // transposed: 0 = 3^d conv, 1 = ConvTranspose k2 s2, 2 = 1x1 conv (dense pointwise GEMM; operator packed with taps = 1)
  F32ConvParams p;
  p.dense = transposed == 2;
  p.x = x; p.sN = st[0]; p.sC = st[1]; p.sD = st[2]; p.sH = st[3]; p.sW = st[4]; p.in_dtype = in_dtype;
  p.y = y; p.y_ss = y_ss; p.w = wpk; p.bias = bias;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
  const int TZ = nd == 3 ? 4 : 1, TY = nd == 3 ? 4 : 16;
  const long long tiles = (long long)N * ((D + TZ - 1) / TZ) * ((H + TY - 1) / TY) * ((W + 15) / 16);
  IUNET_REQUIRE(tiles < (1ll << 31), "f32 conv: too many tiles");
  dim3 grid((unsigned)tiles, Cout / 32, transposed == 1 ? (nd == 3 ? 8 : 4) : 1);
  if (nd == 3) {
    if (transposed) hipLaunchKernelGGL((f32_conv_kernel<3, true>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((f32_conv_kernel<3, false>), grid, dim3(256), 0, stream, p);
  } else {
    if (transposed) hipLaunchKernelGGL((f32_conv_kernel<2, true>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((f32_conv_kernel<2, false>), grid, dim3(256), 0, stream, p);
  }
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_f32_maxpool_launch(int nd, const float* x, long long x_ss, float* y, long long y_ss, int C, int N, int Do, int Ho,
                             int Wo, hipStream_t stream) {
  const long long total = (long long)Do * Ho * Wo * C;
  dim3 grid((unsigned)((total + 255) / 256), N);
  if (nd == 3) hipLaunchKernelGGL((f32_maxpool_kernel<3>), grid, dim3(256), 0, stream, x, x_ss, y, y_ss, C, Do, Ho, Wo);
  else hipLaunchKernelGGL((f32_maxpool_kernel<2>), grid, dim3(256), 0, stream, x, x_ss, y, y_ss, C, Do, Ho, Wo);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_f32_head_launch(const float* x, long long x_ss, int C0, const float* w, const float* bias, int ncls, float* logits,
                          float* probs, unsigned char* cls, const long long* os, float divisor, int accumulate, int N, int D,
                          int H, int W, hipStream_t stream) {
  F32HeadParams p;
  p.x = x; p.x_ss = x_ss; p.C0 = C0; p.w = w; p.bias = bias; p.logits = logits; p.probs = probs; p.cls = cls;
  p.oN = os[0]; p.oC = os[1]; p.oD = os[2]; p.oH = os[3]; p.oW = os[4]; p.divisor = divisor; p.accumulate = accumulate;
  p.N = N; p.D = D; p.H = H; p.W = W;
  const long long vox = (long long)D * H * W;
  dim3 grid((unsigned)((vox + 255) / 256), N);
#define IUNET_F32_HEAD(NC) case NC: hipLaunchKernelGGL((f32_head_kernel<NC>), grid, dim3(256), 0, stream, p); break;
  switch (ncls) { IUNET_F32_HEAD(2) IUNET_F32_HEAD(3) IUNET_F32_HEAD(4) IUNET_F32_HEAD(5) IUNET_F32_HEAD(6)
                  IUNET_F32_HEAD(7) IUNET_F32_HEAD(8) IUNET_F32_HEAD(9) IUNET_F32_HEAD(10) }
#undef IUNET_F32_HEAD
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}
