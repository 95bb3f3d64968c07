// fp16x2 split precision ("x2" mode): the tolerance-meeting forward on the 16-bit matrix cores.
//
// BASELINE.json north_star wants the logits within 1e-3 of the CPU fp32 path and the class map integer-exact; the reference
// predicts in fp32 (predict.py:30-35).  The 16-bit throughput modes round every activation to 11 / 8 bits in HBM (3.6e-3 /
// 3e-2 on the logits); the f32-input MFMA of precise_f32.hip is exact but runs at 1/16 of the 16-bit rate.  Here every value
// -- activation and operator entry -- is carried as TWO fp16 words, hi = f16(v) and lo = f16(v - hi): 22 significant bits.
// A product (x_hi + x_lo)(w_hi + w_lo) is evaluated as x_hi w_hi + x_lo w_hi + x_hi w_lo on v_mfma_f32_16x16x32_f16 (the
// dropped lo x lo term is 2^-22 of the product, the size of the representation error itself); all three terms accumulate in
// the same fp32 accumulator: three matrix instructions per product instead of sixteen.
//
// Layout: a tensor of C channels = C/8 "hi" planes [D][H][W][8] of fp16 (the NHWC8c planes of the 16-bit modes) + C/8 "lo" planes
// at a caller-given plane offset (so the two halves of a skip-concat buffer stay views: [skip_hi | up_hi | skip_lo | up_lo]).
// Scaling: fp16 has 5 exponent bits, so the lo word of a value below 2^-2 is subnormal (exact to 2^-24 only).  Activations
// are therefore kept multiplied by a power of two `act_scale` (default 2^6: exact, undone in the next operator's scale), and
// every operator row is scaled per output channel by the power of two that puts its largest entry in [2^9, 2^10); the
// epilogues multiply the accumulator by the inverse powers (exact) before the bias.  Values are clamped to +-65504.
//
// The convolutions run as the 16-bit kernels over Cin' = 3 Cin VIRTUAL input channels: parts [x_hi | x_lo | x_hi] against operator
// rows [w_hi | w_hi | w_lo] (conv3_v4.hip, template flag SPL).  This file holds what surrounds them: the operator preparation
// (BatchNorm fold in the oracle's fp32 operation order, scaling, split into the virtual fp32 operator that the ordinary pack
// kernels then reorder), the first convolution, the max-pool, the transposed convolution and the head.
#include "common.h"
#include "x2_prep_desc.h"

int iunet_conv3_v4_x2_launch(int nd, const void* x, long long x_sstride, int x_lo, void* y, long long y_sstride, int y_lo, const void* wpk,
                             const float* oscale, const float* bias, int N, int D, int H, int W, int Cin, int Cout, int epi,
                             int* sat, hipStream_t stream);
int iunet_conv3_v4_x2_pack_mode(int nd);

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ operator preparation
// kind 0: conv  w [Cout][Cin][taps]   -> wv [Cout][3 Cin][taps]: per chunk of kc input channels (the consuming kernel's step: 16 in
//         3-D, 32 in 2-D, all Cin for the first conv) the rows [w_hi | w_hi | w_lo]; the activation parts are [x_lo | x_hi | x_hi]
// kind 1: convT w [Cin][Cout][npos]   -> wv [3 Cin][Cout][npos], [hi | hi | lo] over all channels (activation parts [hi | lo | hi])
// kind 2: convT w [Cin][Cout][npos]   -> wv [2 Cin][Cout][npos]: both words of every entry once, in chunks of kc k-steps (of 32
// channels) [chunk][hi | lo][kc][32] -- the operator x2_convT_lds_kernel keeps in LDS.  One workgroup per output channel:
// a = gamma / sqrt(var + eps), w' = w * a, bias' = beta - mean * a (each operation rounded on its own, oracle/unet_ref.py fold_bn),
// s = 2^k with max |w'| * s in [2^9, 2^10), w'' = w' * s (exact), hi = f16(w''), lo = f16(w'' - hi);
// oscale = act_out / (act_in * s), bias_out = bias' * act_out (powers of two: exact).
__device__ __forceinline__ void x2_prep_row(const float* __restrict__ w, float* __restrict__ wv, float* __restrict__ oscale,
                                            float* __restrict__ bias_out, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, const float* __restrict__ mean,
                                            const float* __restrict__ var, const float* __restrict__ bias_in, float eps,
                                            float act_in, float act_out, int Cout, int Cin, int taps, int kind, int kc, int co, float* red) {
#pragma clang fp contract(off)
  const int tid = threadIdx.x;
  float a = 1.0f;
  if (gamma) { const float s = var[co] + eps; a = gamma[co] / sqrtf(s); }
  const int n = Cin * taps;
  auto src = [&](int i) -> float {                       // i = ci * taps + tap
    const int ci = i / taps, t = i - ci * taps;
    const float v = kind == 0 ? w[((long long)co * Cin + ci) * taps + t] : w[((long long)ci * Cout + co) * taps + t];
    return gamma ? v * a : v;
  };
  float m = 0.f;
  for (int i = tid; i < n; i += 256) m = fmaxf(m, fabsf(src(i)));
  red[tid] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
  m = red[0];
  float s = 1.0f;
  if (m > 0.f && m < INFINITY) {
    int e;
    (void)frexpf(m, &e);                                 // m = f 2^e, f in [0.5, 1)
    int k = 10 - e;
    k = k < -40 ? -40 : k > 40 ? 40 : k;
    s = ldexpf(1.0f, k);
  }
  for (int i = tid; i < n; i += 256) {
    const int ci = i / taps, t = i - ci * taps;
    const float v = src(i) * s;
    f16 hi, lo;
    split16<f16>(v, hi, lo);
    const float fh = (float)hi, fl = (float)lo;
    if (kind == 0) {
      // kc = channels per chunk of the consuming kernel (16 / 32 / Cin): virtual channel (3 c + part) * kc + i
      const int c = ci / kc, i = ci - c * kc;
      float* d = wv + ((long long)co * 3 * Cin) * taps + t;
      d[(long long)((3 * c + 0) * kc + i) * taps] = fh; d[(long long)((3 * c + 1) * kc + i) * taps] = fh;
      d[(long long)((3 * c + 2) * kc + i) * taps] = fl;
    } else if (kind == 1) {
      float* d = wv + (long long)co * taps + t;
      d[((long long)ci * Cout) * taps] = fh; d[((long long)(Cin + ci) * Cout) * taps] = fh; d[((long long)(2 * Cin + ci) * Cout) * taps] = fl;
    } else {
      const int ks = ci >> 5, chunk = ks / kc, j = ks - chunk * kc, i = ci & 31;
      float* d = wv + (long long)co * taps + t;
      d[((long long)(((chunk * 2 + 0) * kc + j) * 32 + i) * Cout) * taps] = fh;
      d[((long long)(((chunk * 2 + 1) * kc + j) * 32 + i) * Cout) * taps] = fl;
    }
  }
  if (tid == 0) {
    oscale[co] = act_out / (act_in * s);
    float b = 0.f;
    if (gamma) { const float t = mean[co] * a; b = beta[co] - t; }
    else if (bias_in) b = bias_in[co];
    bias_out[co] = b * act_out;
  }
}
__global__ __launch_bounds__(256) void x2_prep_kernel(const float* __restrict__ w, float* __restrict__ wv, float* __restrict__ oscale,
                                                     float* __restrict__ bias_out, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ mean,
                                                     const float* __restrict__ var, const float* __restrict__ bias_in, float eps,
                                                     float act_in, float act_out, int Cout, int Cin, int taps, int kind, int kc) {
  __shared__ float red[256];
  x2_prep_row(w, wv, oscale, bias_out, gamma, beta, mean, var, bias_in, eps, act_in, act_out, Cout, Cin, taps, kind, kc, blockIdx.x, red);
}
// every operator of a table in one launch: workgroup -> (operator, output channel) by the table's running row count
__global__ __launch_bounds__(256) void x2_prep_batch_kernel(const X2PrepDesc* __restrict__ table, int n) {
  __shared__ float red[256];
  int i = 0;
  while (i + 1 < n && (int)blockIdx.x >= table[i + 1].row0) ++i;
  const X2PrepDesc d = table[i];
  const int co = (int)blockIdx.x - d.row0;
  if (co >= d.Cout) return;
  x2_prep_row(d.w, d.out, d.oscale, d.bias_out, d.gamma, d.beta, d.mean, d.var, d.bias_in, d.eps, d.act_in, d.act_out, d.Cout, d.Cin, d.taps,
              d.kind, d.kc, co, red);
}

// ------------------------------------------------------------------ first conv
struct X2FirstParams {
  const void* x; long long sN, sC, sD, sH, sW; int in_dtype;       // caller's tensor, generic element strides; 0 f32, 1 f16, 2 u8 (/ 255), 3 bf16
  void* y; long long y_sstride; int y_lo;             // y_lo < 0: no lo planes (x2m: a tensor only 3x3x3 convs read)
  void* y8; long long y8_sstride;                      // x2m: the lo8 planes of the output (conv3_x2m.hip; bytes) or null
  int* sat;                                            // optional range flag (common.h: x2_note_saturation)
  const void* w;                        // virtual operator [Cout][3 Cin][taps] in the first conv's fragment order (pack_first_conv, "Cin" = 3 Cin)
  const float* oscale; const float* bias;
  float act_scale;
  int N, D, H, W, Cout, relu;
};

// The structure of pointwise.hip's first_conv_kernel (K = taps x channels padded to 32, im2col operand gathered per lane from an
// LDS image of the halo tile) over the 3 CIN virtual channels [lo | hi | hi] of act_scale * x (operator rows [w_hi | w_hi | w_lo]).
template <int ND, int CIN>
__global__ __launch_bounds__(256) void x2_first_conv_kernel(X2FirstParams p) {
  constexpr int VC = 3 * CIN;
  constexpr int TZ = ND == 3 ? 4 : 1, TY = ND == 3 ? 8 : 16, TX = ND == 3 ? 16 : 32, PADZ = ND == 3 ? 1 : 0;
  constexpr int PZ = TZ + 2 * PADZ, PY = TY + 2, PX = TX + 2, NPIX = PZ * PY * PX;
  constexpr int TAPS = ND == 3 ? 27 : 9, KK = TAPS * VC, KS = (KK + 31) / 32;
  constexpr int FX = TX / 16, NI = 8;
  __shared__ f16 xs[VC * NPIX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, q = lane >> 4;
  const int tilesZ = (p.D + TZ - 1) / TZ, tilesY = (p.H + TY - 1) / TY, tilesX = (p.W + TX - 1) / TX;
  const int tps = tilesZ * tilesY * tilesX;
  const int tile = blockIdx.x, n = tile / tps;
  int trem = tile - n * tps;
  const int tz_i = trem / (tilesY * tilesX);
  trem -= tz_i * tilesY * tilesX;
  const int ty_i = trem / tilesX, tx_i = trem - ty_i * tilesX;
  const int z0 = tz_i * TZ, y0 = ty_i * TY, x0 = tx_i * TX;
  const int cob = blockIdx.y;
  for (int it = tid; it < NPIX * CIN; it += 256) {
    const int c = it / NPIX, pix = it - c * NPIX;
    const int px = pix % PX, t2 = pix / PX, py = t2 % PY, pz = t2 / PY;
    const int gz = z0 + pz - PADZ, gy = y0 + py - 1, gx = x0 + px - 1;
    float v = 0.f;
    if ((unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
      v = x2_load_in(p.x, n * p.sN + c * p.sC + gz * p.sD + gy * p.sH + gx * p.sW, p.in_dtype);
    f16 hi, lo;
    split16<f16>(v * p.act_scale, hi, lo);
    xs[c * NPIX + pix] = lo; xs[(CIN + c) * NPIX + pix] = hi; xs[(2 * CIN + c) * NPIX + pix] = hi;      // parts [x_lo | x_hi | x_hi]
  }
  const f16x8* wp = (const f16x8*)p.w + (long long)cob * KS * 2 * 64 + lane;
  // per-lane LDS element offsets of the 8 k entries of its k-quad (k = 32 ks + 8 q + j = tap * VC + c, pack_first_conv_kernel)
  int koff[KS][8];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 32 * ks + 8 * q + j;
      const int kc = k < KK ? k : 0;                  // padded k: zero weight, any valid address
      const int tap = kc / VC, c = kc % VC;
      const int dz = ND == 3 ? tap / 9 : 0, dy = (tap / 3) % 3, dx = tap % 3;
      koff[ks][j] = c * NPIX + (dz * PY + dy) * PX + dx;
    }
  __syncthreads();
  f16* yout = (f16*)p.y + (long long)n * p.y_sstride;
  const long long plane_stride = (long long)p.D * p.H * p.W * 8;
  float bias[8], osc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { bias[j] = p.bias[cob * 32 + 8 * q + j]; osc[j] = p.oscale[cob * 32 + 8 * q + j]; }
#pragma unroll 1
  for (int nf = 0; nf < NI; ++nf) {
    const int f = wave * NI + nf;
    const int xh = f % FX, row = f / FX, fy = row % TY, fz = row / TY;
    const int base = (fz * PY + fy) * PX + xh * 16 + l15;
    f32x4 acc0 = f32x4{0, 0, 0, 0}, acc1 = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f16x8 b;
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = xs[base + koff[ks][j]];
      acc0 = mfma16<f16>(wp[(ks * 2 + 0) * 64], b, acc0);
      acc1 = mfma16<f16>(wp[(ks * 2 + 1) * 64], b, acc1);
    }
    const int gz = z0 + fz, gy = y0 + fy, gx = x0 + xh * 16 + l15;
    const bool ok = gz < p.D && gy < p.H && gx < p.W;
    f16x8 o, ol;
    float rr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      rr[j] = fmaf(j < 4 ? acc0[j & 3] : acc1[j & 3], osc[j], bias[j]);
      if (p.relu) rr[j] = fmaxf(rr[j], 0.f);
    }
    u32x2_t l8, h8;
    x2m_split8(rr, o, ol, l8, h8);                     // (hi, lo: split16's words)
    if (ok) {
      const long long off = (((long long)gz * p.H + gy) * p.W + gx) * 8;
      *(f16x8*)(yout + (long long)(cob * 4 + q) * plane_stride + off) = o;
      if (p.y_lo >= 0) *(f16x8*)(yout + (long long)(p.y_lo + cob * 4 + q) * plane_stride + off) = ol;
      if (p.sat != nullptr) x2_note_saturation(p.sat, o);
      if (p.y8 != nullptr) {
        unsigned char* y8 = (unsigned char*)p.y8 + (long long)n * p.y8_sstride + x2m_off(cob * 4 + q, off / 8, plane_stride / 8);
        *(u32x2_t*)y8 = l8;
      }
    }
  }
}

// ------------------------------------------------------------------ max-pool 2^d
// the larger hi + lo sum (exact in fp32: 22 bits) wins and its word pair is copied -- no re-split, the pooled tensor holds exactly
// the values of its source
template <int ND>
__global__ __launch_bounds__(256) void x2_maxpool_kernel(const f16* __restrict__ x, long long x_ss, int x_lo, f16* __restrict__ y,
                                                         long long y_ss, int y_lo, int planes, int Do, int Ho, int Wo) {
  const long long ovox = (long long)Do * Ho * Wo;
  const long long total = ovox * planes;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int n = blockIdx.y;
  const int pl = (int)(i / ovox);
  const long long r = i - (long long)pl * ovox;
  const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), oz = (int)(r / ((long long)Wo * Ho));
  const int Di = ND == 3 ? Do * 2 : 1, Hi = Ho * 2, Wi = Wo * 2;
  const long long ivox = (long long)Di * Hi * Wi;
  const f16* xh = x + n * x_ss + (long long)pl * ivox * 8;
  const f16* xl = xh + (long long)x_lo * ivox * 8;
  float m[8];
  f16x8 oh, ol;
#pragma unroll
  for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
  for (int a = 0; a < (ND == 3 ? 2 : 1); ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int z = ND == 3 ? oz * 2 + a : 0;
        const long long off = (((long long)z * Hi + oy * 2 + b) * Wi + ox * 2 + c) * 8;
        const f16x8 vh = *(const f16x8*)(xh + off), vl = *(const f16x8*)(xl + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = (float)vh[j] + (float)vl[j];
          if (v > m[j]) { m[j] = v; oh[j] = vh[j]; ol[j] = vl[j]; }
        }
      }
  f16* yh = y + n * y_ss + (long long)pl * ovox * 8 + r * 8;
  *(f16x8*)yh = oh;
  *(f16x8*)(yh + (long long)y_lo * ovox * 8) = ol;
}

// ------------------------------------------------------------------ transposed conv k2 s2
// pointwise.hip's transposed conv (one wave = 16 input x voxels x 32 couts x all 2^d output positions, B operand straight from
// global, stride-2 interleave of the two x positions for full-line stores).
struct X2ConvTParams {
  const void* x; long long x_sstride; int x_lo;
  void* y; long long y_sstride; int y_lo;             // y_lo < 0: no lo planes
  void* y8; long long y8_sstride;                      // x2m: the lo8 planes of the output (bytes) or null
  int* sat;                                            // optional range flag
  const void* wpk; const float* oscale; const float* bias;
  int N, D, H, W, Cin, Cout;            // input grid; Cin = real input channels
};

// The Cout tile's weights sit in LDS -- BOTH words of every entry once ([hi | lo], prep kind 2), in chunks of KC k-steps -- and a wave
// takes two 16-voxel groups: a first version that fetched 3 x 16 KB of weight fragments per k-step and wave through the vector
// cache (the virtual [hi | hi | lo] operator) spent 12 % of the 128^3 forward on 1.7 % of its FLOPs.  One chunk (Cin <= 64): loaded once per workgroup, which then
// walks its voxel groups; more: double-buffered by LDS-DMA through every pass (the structure of pointwise.hip's convT_chunk_kernel).
// Per (k-step, position, cout half): 2 fragment reads feed 3 MFMAs per voxel group: x_hi w_hi + x_lo w_hi + x_hi w_lo.
// RES (one chunk, resident): the output positions go in two halves -- 64 accumulator registers instead of 128, the kernel then fits
// 256 registers and two workgroups share a CU (the first form held 418: one wave per SIMD, its loads and its heavy epilogue in the open).
template <int ND, int KC, bool RES = false>
__global__ __launch_bounds__(256, RES ? 2 : 1) void x2_convT_lds_kernel(X2ConvTParams p) {
  constexpr int NPOS = ND == 3 ? 8 : 4;
  constexpr int G = 2;
  constexpr int CHB = 2 * KC * NPOS * 2 * 1024;                 // bytes of one chunk
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int cob = blockIdx.y;
  const int nchunks = (p.Cin >> 5) / KC;
  const u32x4* wsrc = (const u32x4*)p.wpk + (long long)cob * nchunks * (CHB / 16);
  const int xg = (p.W + 15) / 16;
  const long long rows = (long long)p.D * p.H * xg, ngroups = rows * p.N;
  const long long in_plane = (long long)p.D * p.H * p.W * 8;
  const int Do = ND == 3 ? p.D * 2 : 1, Ho = p.H * 2, Wo = p.W * 2;
  const long long out_plane = (long long)Do * Ho * Wo * 8;
  float bias[RES ? 1 : 8], osc[RES ? 1 : 8];
  if constexpr (!RES) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { bias[j] = p.bias[cob * 32 + q * 8 + j]; osc[j] = p.oscale[cob * 32 + q * 8 + j]; }
  } else if (threadIdx.x < 64) {      // RES: [oscale 32 | bias 32] behind the chunk, read per epilogue (16 registers less); published by the first pass's barrier
    ((float*)(smem + CHB))[threadIdx.x] = threadIdx.x < 32 ? p.oscale[cob * 32 + threadIdx.x] : p.bias[cob * 32 + threadIdx.x - 32];
  }
  const int src_lo = ((lane & 48) | (l15 >> 1)) * 4, src_hi = src_lo + 8 * 4;
  const bool odd = l15 & 1;
  const f16x8* wl = (const f16x8*)smem + lane;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  auto dma_chunk = [&](int ch) {
    const u32x4* src = wsrc + (long long)ch * (CHB / 16);
#pragma unroll
    for (int i = 0; i < CHB / 16 / 256; ++i) {
      const int base = (i * 4 + wave) * 64;                       // first 16-byte item of this wave instruction
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (ch & 1) * CHB + base * 16);
      const u32x4* gsrc = src + base + lane;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
    }
  };
  bool resident = false;                                           // one chunk: it stays in LDS over the passes
  for (long long base = (long long)blockIdx.x * 4 * G; base < ngroups; base += (long long)gridDim.x * 4 * G) {      // uniform trip count
    const long long g0 = base + wave * G;
    int n_[G], z_[G], y_[G], xb_[G];
    const f16* xin_[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const long long wid = g0 + g < ngroups ? g0 + g : ngroups - 1;      // a missing partner recomputes the last group (never stored)
      n_[g] = (int)(wid / rows);
      const long long r = wid - n_[g] * rows;
      xb_[g] = (int)(r % xg); y_[g] = (int)((r / xg) % p.H); z_[g] = (int)(r / ((long long)xg * p.H));
      const int xc = min(xb_[g] * 16 + l15, p.W - 1);
      xin_[g] = (const f16*)p.x + n_[g] * p.x_sstride + (((long long)z_[g] * p.H + y_[g]) * p.W + xc) * 8;
    }
    auto load_b = [&](int ch, f16x8 (&bb)[G][2][KC]) {
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < KC; ++j) {
          const long long pl = (long long)((ch * KC + j) * 4 + q);
          bb[g][0][j] = *(const f16x8*)(xin_[g] + pl * in_plane);
          bb[g][1][j] = *(const f16x8*)(xin_[g] + (pl + p.x_lo) * in_plane);
        }
    };
    constexpr int NH = RES ? 2 : 1, HP = NPOS / NH;                // passes over the output positions, positions per pass
    f32x4 acc[G][HP][2];
    auto zero_acc = [&]() {
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int s = 0; s < HP; ++s) { acc[g][s][0] = f32x4{0, 0, 0, 0}; acc[g][s][1] = f32x4{0, 0, 0, 0}; }
    };
    f16x8 bcur[G][2][KC], bnext[G][2][KC];
    auto mfmas = [&](int ch, int hf) {                             // chunk ch on positions hf * HP ..
      const f16x8* wb = wl + (ch & 1) * (CHB / 16);
#pragma unroll
      for (int j = 0; j < KC; ++j)
#pragma unroll
        for (int s = 0; s < HP; ++s)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const f16x8 ah = wb[(((0 * KC + j) * NPOS + hf * HP + s) * 2 + t) * 64];
            const f16x8 al = wb[(((1 * KC + j) * NPOS + hf * HP + s) * 2 + t) * 64];
#pragma unroll
            for (int g = 0; g < G; ++g) {
              acc[g][s][t] = mfma16<f16>(ah, bcur[g][0][j], acc[g][s][t]);
              acc[g][s][t] = mfma16<f16>(ah, bcur[g][1][j], acc[g][s][t]);
              acc[g][s][t] = mfma16<f16>(al, bcur[g][0][j], acc[g][s][t]);
            }
          }
    };
    zero_acc();
    if (!resident) {
      __syncthreads();                                             // the previous pass's last chunk is read
      dma_chunk(0);
    }
    load_b(0, bcur);
    for (int ch = 0; ch < nchunks; ++ch) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this chunk's weights and fragments have landed
      __syncthreads();
      if (ch + 1 < nchunks) { dma_chunk(ch + 1); load_b(ch + 1, bnext); }
      mfmas(ch, 0);
      if (ch + 1 < nchunks) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < KC; ++j) bcur[g][h][j] = bnext[g][h][j];
      }
    }
    resident = nchunks == 1;
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
    if (hf > 0) { zero_acc(); mfmas(0, hf); }                        // RES: the second half of the positions on the same fragments
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g0 + g >= ngroups) break;
      f16* yout = (f16*)p.y + n_[g] * p.y_sstride + (long long)(cob * 4 + q) * out_plane;
      const int x0 = xb_[g] * 16;
#pragma unroll
      for (int s2 = 0; s2 < HP / 2; ++s2) {
        const int sp = hf * (HP / 2) + s2;
        const int a = ND == 3 ? (sp >> 1) : 0, b = sp & 1;
        i32x4 oc[2][2];                                  // [x position][hi | lo]
        [[maybe_unused]] f32x4 cs[4];                    // RES: this lane's 8 channels of [oscale | bias]
        if constexpr (RES) {
          const f32x4* sp4 = (const f32x4*)(smem + CHB);
          cs[0] = sp4[2 * q]; cs[1] = sp4[2 * q + 1]; cs[2] = sp4[8 + 2 * q]; cs[3] = sp4[8 + 2 * q + 1];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          f16x8 o, ol;
          float rr[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if constexpr (RES) rr[j] = fmaf(j < 4 ? acc[g][s2 * 2 + c][0][j & 3] : acc[g][s2 * 2 + c][1][j & 3], cs[j >> 2][j & 3], cs[2 + (j >> 2)][j & 3]);
            else rr[j] = fmaf(j < 4 ? acc[g][s2 * 2 + c][0][j & 3] : acc[g][s2 * 2 + c][1][j & 3], osc[j], bias[j]);
          }
          if (p.y8 == nullptr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              f16 hi, lo;
              split16<f16>(rr[j], hi, lo);
              o[j] = hi; ol[j] = lo;
            }
          } else {
            // x2m: this lane's half-granules of the lo8 planes go straight out (output voxel 2 x + c of row (oz, oy): 8-byte stores)
            u32x2_t l8, h8;
            x2m_split8(rr, o, ol, l8, h8);
            if (x0 + l15 < p.W) {
              const long long ovox = (long long)Do * Ho * Wo;
              const long long vo = ((long long)(ND == 3 ? z_[g] * 2 + a : 0) * Ho + y_[g] * 2 + b) * Wo + 2 * (x0 + l15) + c;
              unsigned char* y8 = (unsigned char*)p.y8 + n_[g] * p.y8_sstride + x2m_off(cob * 4 + q, vo, ovox);
              *(u32x2_t*)y8 = l8;
            }
          }
          if (p.sat != nullptr && x0 + l15 < p.W) x2_note_saturation(p.sat, o);
          oc[c][0] = __builtin_bit_cast(i32x4, o);
          oc[c][1] = __builtin_bit_cast(i32x4, ol);
        }
        const int oz = ND == 3 ? z_[g] * 2 + a : 0;
        f16* row = yout + (((long long)oz * Ho + y_[g] * 2 + b) * Wo + 2 * x0) * 8;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          if (w == 1 && p.y_lo < 0) continue;              // (x2m: the lo planes of a tensor only 3x3x3 convs read are not written)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int src = h ? src_hi : src_lo;
            i32x4 v;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const int t0 = __builtin_amdgcn_ds_bpermute(src, oc[0][w][d]);
              const int t1 = __builtin_amdgcn_ds_bpermute(src, oc[1][w][d]);
              v[d] = odd ? t1 : t0;
            }
            if (2 * x0 + 16 * h + l15 < Wo) *(i32x4*)(row + (w ? (long long)p.y_lo * out_plane : 0) + (16 * h + l15) * 8) = v;
          }
        }
      }
    }
    }
  }
}

// ------------------------------------------------------------------ 1x1 head + softmax / argmax
struct X2HeadParams {
  const void* x; long long x_sstride; int planes, x_lo;
  const float* w; const float* bias; float inv_act;
  float* logits; float* probs; unsigned char* cls;
  long long oN, oC, oD, oH, oW;
  float divisor; int accumulate;
  int N, D, H, W;
};

// fp32 arithmetic of precise_f32.hip's head (fmaf chain over the channels, expf, correctly rounded division) on the
// reconstructed features (hi + lo) / act_scale (exact)
template <int NCLS>
__global__ __launch_bounds__(256) void x2_head_kernel(X2HeadParams p) {
  const long long vox = (long long)p.D * p.H * p.W;
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
  if (v >= vox) return;
  const int n = blockIdx.y;
  const f16* xin = (const f16*)p.x + n * p.x_sstride + v * 8;
  float l[NCLS];
#pragma unroll
  for (int c = 0; c < NCLS; ++c) l[c] = 0.f;
  // Summation order (shared with the head fused into the last x2m conv's epilogue, conv3_x2m.hip: x2m_head): one fmaf chain per
  // 8-channel plane, the planes of a group of four added as (p0 + p1) + (p2 + p3) -- what two butterfly exchanges between the four lane
  // groups of an MFMA fragment give --, the groups (base 64: two) added in order.  The loads of a group are issued before the first is
  // consumed.
  for (int pl0 = 0; pl0 < p.planes; pl0 += 4) {
    f16x8 xh[4], xl[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pl = min(pl0 + u, p.planes - 1);
      xh[u] = *(const f16x8*)(xin + (long long)pl * vox * 8);
      xl[u] = *(const f16x8*)(xin + (long long)(p.x_lo + pl) * vox * 8);
    }
    float part[4][NCLS];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pl = pl0 + u;
#pragma unroll
      for (int c = 0; c < NCLS; ++c) part[u][c] = 0.f;
      if (pl < p.planes) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float a = ((float)xh[u][j] + (float)xl[u][j]) * p.inv_act;
#pragma unroll
          for (int c = 0; c < NCLS; ++c) part[u][c] = fmaf(a, p.w[c * p.planes * 8 + pl * 8 + j], part[u][c]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
      const float g = __fadd_rn(__fadd_rn(part[0][c], part[1][c]), __fadd_rn(part[2][c], part[3][c]));
      l[c] = pl0 == 0 ? g : __fadd_rn(l[c], g);
    }
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
  float pr[NCLS];
  pr[0] = __fdiv_rn(e[0], s);
  float pm = pr[0]; int am = 0;
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

bool pow2(float v) { int e; return v > 0.f && frexpf(v, &e) == 0.5f; }

}  // namespace

extern "C" {

/* k-steps (of 32 input channels) per LDS chunk of the transposed conv's [hi | lo] operator (iunet_x2_prep transposed = 2) */
int iunet_x2_convT_kc(int Cin) { return (Cin / 32) % 2 == 0 ? 2 : 1; }

/* fp32 operator of a stage conv (transposed = 0: VIRTUAL [Cout][3 Cin][taps] = [w_hi | w_hi | w_lo]) or a transposed conv (transposed = 1:
 * virtual [3 Cin][Cout][npos]; transposed = 2: [2 Cin][Cout][npos], both words once in chunks of iunet_x2_convT_kc k-steps -- what
 * iunet_x2_convT_fwd takes) + its accumulator scale and scaled bias; gamma..var: the eval-mode BatchNorm folded in (or null),
 * bias_in: the layer's own bias (or null) */
int iunet_x2_prep(const void* w, void* wv, void* oscale, void* bias_out, const void* gamma, const void* beta, const void* mean,
                  const void* var, const void* bias_in, float eps, float act_in, float act_out, int Cout, int Cin, int taps,
                  int transposed, int chunk, void* stream) {
  IUNET_REQUIRE(w && wv && oscale && bias_out, "x2_prep: null pointer");
  IUNET_REQUIRE(transposed || (chunk > 0 && Cin % chunk == 0), "x2_prep: the chunk (%d) must divide Cin (%d)", chunk, Cin);
  IUNET_REQUIRE(Cout > 0 && Cin > 0 && taps > 0, "x2_prep: bad operator %d x %d x %d", Cout, Cin, taps);
  IUNET_REQUIRE(transposed >= 0 && transposed <= 2, "x2_prep: transposed must be 0, 1 or 2 (got %d)", transposed);
  IUNET_REQUIRE(transposed != 2 || Cin % 32 == 0, "x2_prep: the chunked transposed operator needs Cin %% 32 == 0 (got %d)", Cin);
  IUNET_REQUIRE(!gamma || (beta && mean && var), "x2_prep: a BatchNorm fold needs gamma, beta, mean and var");
  IUNET_REQUIRE(pow2(act_in) && pow2(act_out), "x2_prep: the activation scales must be powers of two (got %g, %g)", act_in, act_out);
  hipLaunchKernelGGL(x2_prep_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, (const float*)w, (float*)wv, (float*)oscale,
                     (float*)bias_out, (const float*)gamma, (const float*)beta, (const float*)mean, (const float*)var,
                     (const float*)bias_in, eps, act_in, act_out, Cout, Cin, taps, transposed,
                     transposed ? iunet_x2_convT_kc(Cin) : chunk);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

/* iunet_x2_prep for every operator of a device-resident table of n X2PrepDesc rows (kinds 0..2; x2_prep_desc.h) in ONE launch of `rows`
 * workgroups (= the table's running sum of Cout); the caller validated each row as iunet_x2_prep would (engine_x2.py builds the rows from the
 * arguments it used to pass layer by layer).  Same kernel body per row: the same bits. */
int iunet_x2_prep_desc_bytes(void) { return (int)sizeof(X2PrepDesc); }
int iunet_x2_prep_batch(const void* table, int n, int rows, void* stream) {
  IUNET_REQUIRE(table != nullptr && n > 0 && rows > 0, "x2_prep_batch: empty table");
  hipLaunchKernelGGL(x2_prep_batch_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const X2PrepDesc*)table, n);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

/* unet.py:65-69 first conv of the split-precision forward: the caller's tensor (strides, dtype as iunet_first_conv_fwd) ->
 * split(relu?(conv * oscale + bias)); w = iunet_pack_first_conv of the virtual operator with "Cin" = 3 Cin */
int iunet_x2m_first_conv_fwd(int nd, const void* x, int in_dtype, const long long* in_strides, void* y, long long y_sstride, int y_lo,
                             void* y8, long long y8_sstride, const void* w, const void* oscale, const void* bias, float act_scale, int N,
                             int D, int H, int W, int Cin, int Cout, int relu, void* sat, void* stream);
int iunet_x2_first_conv_fwd(int nd, const void* x, int in_dtype, const long long* in_strides, void* y, long long y_sstride, int y_lo,
                            const void* w, const void* oscale, const void* bias, float act_scale, int N, int D, int H, int W, int Cin,
                            int Cout, int relu, void* stream) {
  IUNET_REQUIRE(y_lo >= 0, "x2_first_conv: y_lo must not be negative");
  return iunet_x2m_first_conv_fwd(nd, x, in_dtype, in_strides, y, y_sstride, y_lo, nullptr, 0, w, oscale, bias, act_scale, N, D, H, W, Cin, Cout,
                                  relu, nullptr, stream);
}

/* the same first conv writing, beside the hi planes, the lo8 planes of its output (y8, y8_sstride bytes per sample; null: none) and the
 * lo planes only when y_lo >= 0; sat: optional device int raised to 0x7bff when a stored hi word saturated */
int iunet_x2m_first_conv_fwd(int nd, const void* x, int in_dtype, const long long* in_strides, void* y, long long y_sstride, int y_lo,
                             void* y8, long long y8_sstride, const void* w, const void* oscale, const void* bias, float act_scale, int N,
                             int D, int H, int W, int Cin, int Cout, int relu, void* sat, void* stream) {
  IUNET_REQUIRE(x && y && w && oscale && bias && in_strides, "x2_first_conv: null pointer");
  IUNET_REQUIRE(nd == 2 || nd == 3, "x2_first_conv: nd must be 2 or 3");
  IUNET_REQUIRE_GRID("x2_first_conv", N, D, H, W);
  IUNET_REQUIRE(nd == 3 || D == 1, "x2_first_conv: 2-D needs D == 1");
  IUNET_REQUIRE(in_dtype >= 0 && in_dtype <= 3, "x2_first_conv: bad input dtype %d", in_dtype);
  IUNET_REQUIRE(Cin >= 1 && Cin <= 4, "x2_first_conv: Cin must be 1..4 (got %d)", Cin);
  IUNET_REQUIRE(Cout > 0 && Cout % 32 == 0, "x2_first_conv: Cout must be a multiple of 32 (got %d)", Cout);
  IUNET_REQUIRE(pow2(act_scale), "x2_first_conv: the activation scale must be a power of two (got %g)", act_scale);
  X2FirstParams p;
  p.x = x; p.sN = in_strides[0]; p.sC = in_strides[1]; p.sD = in_strides[2]; p.sH = in_strides[3]; p.sW = in_strides[4];
  p.in_dtype = in_dtype; p.y = y; p.y_sstride = y_sstride; p.y_lo = y_lo; p.y8 = y8; p.y8_sstride = y8_sstride; p.sat = (int*)sat; p.w = w; p.oscale = (const float*)oscale;
  p.bias = (const float*)bias; p.act_scale = act_scale; p.N = N; p.D = D; p.H = H; p.W = W; p.Cout = Cout; p.relu = relu;
  const int TZ = nd == 3 ? 4 : 1, TY = nd == 3 ? 8 : 16, TX = nd == 3 ? 16 : 32;
  dim3 grid(N * ((D + TZ - 1) / TZ) * ((H + TY - 1) / TY) * ((W + TX - 1) / TX), Cout / 32);
#define X2FC(NDV, CI) hipLaunchKernelGGL((x2_first_conv_kernel<NDV, CI>), grid, dim3(256), 0, (hipStream_t)stream, p)
#define X2FC_CIN(NDV) switch (Cin) { case 1: X2FC(NDV, 1); break; case 2: X2FC(NDV, 2); break; case 3: X2FC(NDV, 3); break; default: X2FC(NDV, 4); break; }
  if (nd == 3) { X2FC_CIN(3) } else { X2FC_CIN(2) }
#undef X2FC_CIN
#undef X2FC
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

/* pack mode of the stage convs' virtual operator (iunet_pack_conv3 on iunet_x2_prep's output): 2 (padded K16 order) in 3-D, 6 (compact
 * order: the third filter column of the two 16-channel halves of a step shares one k-group -- 9 taps in 9 k-slots) in 2-D */
int iunet_x2_pack_mode(int nd) { return iunet_conv3_v4_x2_pack_mode(nd); }

/* stage conv 3^d of the split-precision forward: x / y = Cin / 8 (Cout / 8) hi planes, the lo planes x_lo / y_lo planes further on;
 * wpk = iunet_pack_conv3 (mode iunet_x2_pack_mode(nd)) of the virtual operator [Cout][3 Cin][taps]; epi as iunet_conv3_fwd */
int iunet_x2_conv3_fwd_flag(int nd, const void* x, long long x_sstride, int x_lo, void* y, long long y_sstride, int y_lo, const void* wpk,
                            const void* oscale, const void* bias, int N, int D, int H, int W, int Cin, int Cout, int epi, void* sat, void* stream) {
  IUNET_REQUIRE(x && y && wpk && oscale, "x2_conv3: null pointer");
  IUNET_REQUIRE(nd == 2 || nd == 3, "x2_conv3: nd must be 2 or 3");
  IUNET_REQUIRE_GRID("x2_conv3", N, D, H, W);
  IUNET_REQUIRE(nd == 3 || D == 1, "x2_conv3: 2-D needs D == 1");
  IUNET_REQUIRE(Cin >= 32 && Cout >= 32 && Cin % 32 == 0 && Cout % 32 == 0, "x2_conv3: channels must be positive multiples of 32 (%d, %d)", Cin, Cout);
  IUNET_REQUIRE(epi >= 0 && epi <= 2, "x2_conv3: bad epilogue %d", epi);
  IUNET_REQUIRE(epi == 0 || bias != nullptr, "x2_conv3: epilogue %d needs a bias", epi);
  return iunet_conv3_v4_x2_launch(nd, x, x_sstride, x_lo, y, y_sstride, y_lo, wpk, (const float*)oscale, (const float*)bias, N, D, H, W,
                                  Cin, Cout, epi, (int*)sat, (hipStream_t)stream);
}

int iunet_x2_conv3_fwd(int nd, const void* x, long long x_sstride, int x_lo, void* y, long long y_sstride, int y_lo, const void* wpk,
                       const void* oscale, const void* bias, int N, int D, int H, int W, int Cin, int Cout, int epi, void* stream) {
  return iunet_x2_conv3_fwd_flag(nd, x, x_sstride, x_lo, y, y_sstride, y_lo, wpk, oscale, bias, N, D, H, W, Cin, Cout, epi, nullptr, stream);
}

int iunet_x2_maxpool_fwd(int nd, const void* x, long long x_ss, int x_lo, void* y, long long y_ss, int y_lo, int C, int N, int Do,
                         int Ho, int Wo, void* stream) {
  IUNET_REQUIRE(x && y, "x2_maxpool: null pointer");
  IUNET_REQUIRE(nd == 2 || nd == 3, "x2_maxpool: nd must be 2 or 3");
  IUNET_REQUIRE(C > 0 && C % 8 == 0 && N > 0 && Do > 0 && Ho > 0 && Wo > 0, "x2_maxpool: bad shape");
  const long long total = (long long)Do * Ho * Wo * (C / 8);
  dim3 grid((unsigned)((total + 255) / 256), N);
  if (nd == 3) hipLaunchKernelGGL((x2_maxpool_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, x_ss, x_lo, (f16*)y, y_ss, y_lo, C / 8, Do, Ho, Wo);
  else hipLaunchKernelGGL((x2_maxpool_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, x_ss, x_lo, (f16*)y, y_ss, y_lo, C / 8, Do, Ho, Wo);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

/* transposed conv k2 s2; wpk = iunet_pack_convT ("Cin" = 2 Cin) of iunet_x2_prep's transposed = 2 operator [2 Cin][Cout][npos] */
int iunet_x2m_convT_fwd(int nd, const void* x, long long x_ss, int x_lo, void* y, long long y_ss, int y_lo, void* y8, long long y8_ss,
                        const void* wpk, const void* oscale, const void* bias, int N, int D, int H, int W, int Cin, int Cout, void* sat, void* stream);
int iunet_x2_convT_fwd(int nd, const void* x, long long x_ss, int x_lo, void* y, long long y_ss, int y_lo, const void* wpk,
                       const void* oscale, const void* bias, int N, int D, int H, int W, int Cin, int Cout, void* stream) {
  IUNET_REQUIRE(y_lo >= 0, "x2_convT: y_lo must not be negative");
  return iunet_x2m_convT_fwd(nd, x, x_ss, x_lo, y, y_ss, y_lo, nullptr, 0, wpk, oscale, bias, N, D, H, W, Cin, Cout, nullptr, stream);
}

/* the same transposed conv writing, beside the hi planes, the lo8 planes of its output (y8, y8_ss bytes per sample; null: none) and the lo
 * planes only when y_lo >= 0; sat: optional range flag */
int iunet_x2m_convT_fwd(int nd, const void* x, long long x_ss, int x_lo, void* y, long long y_ss, int y_lo, void* y8, long long y8_ss,
                        const void* wpk, const void* oscale, const void* bias, int N, int D, int H, int W, int Cin, int Cout, void* sat, void* stream) {
  IUNET_REQUIRE(x && y && wpk && oscale && bias, "x2_convT: null pointer");
  IUNET_REQUIRE(nd == 2 || nd == 3, "x2_convT: nd must be 2 or 3");
  IUNET_REQUIRE_GRID("x2_convT", N, D, H, W);
  IUNET_REQUIRE(Cin >= 32 && Cout >= 32 && Cin % 32 == 0 && Cout % 32 == 0, "x2_convT: channels must be positive multiples of 32 (%d, %d)", Cin, Cout);
  X2ConvTParams p;
  p.x = x; p.x_sstride = x_ss; p.x_lo = x_lo; p.y = y; p.y_sstride = y_ss; p.y_lo = y_lo; p.y8 = y8; p.y8_sstride = y8_ss; p.sat = (int*)sat; p.wpk = wpk;
  p.oscale = (const float*)oscale; p.bias = (const float*)bias; p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  const long long waves = (long long)N * D * H * ((W + 15) / 16);
  const int kc = iunet_x2_convT_kc(Cin), nchunks = Cin / 32 / kc, npos = nd == 3 ? 8 : 4;
  const int chb = 2 * kc * npos * 2 * 1024, lds = nchunks == 1 ? chb + 256 : 2 * chb;      // (resident form: + its [oscale | bias])
  int gx = (int)((waves + 7) / 8);
  const int per_cu = lds <= 80 * 1024 ? 2 : 1;
  const int cap = per_cu * 256 / (Cout / 32) > 1 ? per_cu * 256 / (Cout / 32) : 1;
  if (gx > cap) gx = cap;
  dim3 grid(gx, Cout / 32);
  static const int res_on = getenv("IUNET_X2CT_RES") ? atoi(getenv("IUNET_X2CT_RES")) : 1;      // A/B: the two-half resident form
#define X2CT(NDV, KCV) do { if (nchunks == 1 && res_on) { IUNET_SET_MAX_LDS((x2_convT_lds_kernel<NDV, KCV, true>), lds); \
    hipLaunchKernelGGL((x2_convT_lds_kernel<NDV, KCV, true>), grid, dim3(256), lds, (hipStream_t)stream, p); } else { IUNET_SET_MAX_LDS((x2_convT_lds_kernel<NDV, KCV>), lds); \
    hipLaunchKernelGGL((x2_convT_lds_kernel<NDV, KCV>), grid, dim3(256), lds, (hipStream_t)stream, p); } } while (0)
  if (nd == 3) { if (kc == 2) X2CT(3, 2); else X2CT(3, 1); } else { if (kc == 2) X2CT(2, 2); else X2CT(2, 1); }
#undef X2CT
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

/* 1x1 head + softmax + class map (unet.py:63-69, predict.py:38) from split features; output contract of iunet_head_fwd */
int iunet_x2_head_fwd(const void* x, long long x_ss, int x_lo, int C0, const void* w, const void* bias, float act_scale, int ncls,
                      void* logits, void* probs, void* cls, const long long* out_strides, float divisor, int accumulate, int N, int D,
                      int H, int W, void* stream) {
  IUNET_REQUIRE(x && w && bias && out_strides, "x2_head: null pointer");
  IUNET_REQUIRE(C0 > 0 && C0 % 8 == 0, "x2_head: C0 must be a multiple of 8");
  IUNET_REQUIRE(ncls >= 2 && ncls <= 10, "x2_head: num_classes must be 2..10 (got %d)", ncls);
  IUNET_REQUIRE_GRID("x2_head", N, D, H, W);
  IUNET_REQUIRE(pow2(act_scale), "x2_head: the activation scale must be a power of two (got %g)", act_scale);
  X2HeadParams p;
  p.x = x; p.x_sstride = x_ss; p.planes = C0 / 8; p.x_lo = x_lo; p.w = (const float*)w; p.bias = (const float*)bias;
  p.inv_act = 1.0f / act_scale; p.logits = (float*)logits; p.probs = (float*)probs; p.cls = (unsigned char*)cls;
  p.oN = out_strides[0]; p.oC = out_strides[1]; p.oD = out_strides[2]; p.oH = out_strides[3]; p.oW = out_strides[4];
  p.divisor = divisor; p.accumulate = accumulate; p.N = N; p.D = D; p.H = H; p.W = W;
  const long long vox = (long long)D * H * W;
  dim3 grid((unsigned)((vox + 255) / 256), N);
#define X2_HEAD(NC) case NC: hipLaunchKernelGGL((x2_head_kernel<NC>), grid, dim3(256), 0, (hipStream_t)stream, p); break;
  switch (ncls) { X2_HEAD(2) X2_HEAD(3) X2_HEAD(4) X2_HEAD(5) X2_HEAD(6) X2_HEAD(7) X2_HEAD(8) X2_HEAD(9) X2_HEAD(10) }
#undef X2_HEAD
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

}  // extern "C"
