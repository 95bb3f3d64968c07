// HBM-bound forward kernels around the MFMA convolutions: the first convolution
// (tiny Cin, reads the caller's NCHW / uint8 / strided block directly), 2x max-pool,
// the k2 s2 transposed convolution (MFMA, operands straight from global: it is bound by
// its 8x larger output, not by the matrix cores) and the 1x1 head fused with
// softmax / argmax (writes the caller's NCHW fp32 / uint8 / strided accumulator).
#include "common.h"

namespace {

// ------------------------------------------------------------------ first conv
struct FirstConvParams {
  const void* x;       // input, generic strides (elements)
  long long sN, sC, sD, sH, sW;
  int in_dtype;        // 0 f32, 1 f16, 2 u8 (scaled by 1/255), 3 bf16
  void* y; long long y_sstride;
  const void* w;       // packed weights [cob32][kstep][2][64][8] of T (iunet_pack_first_conv)
  const float* bias;   // [Cout] or null
  float* stats;        // [ntiles][Cout][2] or null (ntiles = iunet_conv3_num_tiles)
  int N, D, H, W, Cin, Cout, nd, relu;
  int out8;            // 1: y = e4m3 planes [Cout / 16][D][H][W][16 B], y_sstride in bytes (the 16-bit result rounded once more)
};

__device__ __forceinline__ float load_in(const void* p, long long off, int dt) {
  switch (dt) {
    case 0: return ((const float*)p)[off];
    case 1: return (float)((const f16*)p)[off];
    case 2: return (float)((const unsigned char*)p)[off] / 255.0f;
    default: return (float)((const bf16*)p)[off];
  }
}

// First conv on the matrix cores: K = taps * Cin (27..108) padded to a multiple of 32; the im2col
// operand is gathered per lane from a 16-bit LDS image of the halo tile (8 scalar reads feed one
// k-quad), the packed weights live in registers.  The kernel is bound by its 64 B/voxel output.
template <typename T, int ND, int CIN>
__global__ __launch_bounds__(256) void first_conv_kernel(FirstConvParams p) {
  using V8 = typename Vec8<T>::type;
  constexpr int TZ = ND == 3 ? 4 : 1, TY = ND == 3 ? 8 : 16, TX = ND == 3 ? 16 : 32, PADZ = ND == 3 ? 1 : 0;
  constexpr int PZ = TZ + 2 * PADZ, PY = TY + 2, PX = TX + 2, NPIX = PZ * PY * PX;
  constexpr int TAPS = ND == 3 ? 27 : 9, KK = TAPS * CIN, KS = (KK + 31) / 32;
  constexpr int FX = TX / 16, NI = 8;
  __shared__ T xs[CIN * NPIX];
  __shared__ float red[4 * 4 * 8 * 2];
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
  // ---- stage the halo tile (caller's layout and dtype, rounded to T) ----
  for (int it = tid; it < NPIX * CIN; it += 256) {
    const int c = it / NPIX, pix = it - c * NPIX;
    const int px = pix % PX, t2 = pix / PX, py = t2 % PY, pz = t2 / PY;
    const int gz = z0 + pz - PADZ, gy = y0 + py - 1, gx = x0 + px - 1;
    float v = 0.f;
    if ((unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
      v = load_in(p.x, n * p.sN + c * p.sC + gz * p.sD + gy * p.sH + gx * p.sW, p.in_dtype);
    xs[it] = from_f32<T>(v);
  }
  // ---- weights: [cob][kstep][2][64][8] ----
  V8 a[KS][2];
  const V8* wp = (const V8*)p.w + (long long)cob * KS * 2 * 64 + lane;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) { a[ks][0] = wp[(ks * 2 + 0) * 64]; a[ks][1] = wp[(ks * 2 + 1) * 64]; }
  // per-lane LDS element offsets of the 8 k entries of its k-quad (k = 32 ks + 8 q + j -> tap, channel)
  int koff[KS][8];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 32 * ks + 8 * q + j;
      const int kc = k < KK ? k : 0;                  // padded k: zero weight, any valid address
      const int tap = kc / CIN, c = kc % CIN;
      const int dz = ND == 3 ? tap / 9 : 0, dy = (tap / 3) % 3, dx = tap % 3;
      koff[ks][j] = c * NPIX + (dz * PY + dy) * PX + dx;
    }
  __syncthreads();
  T* yout = (T*)p.y + (long long)n * p.y_sstride;
  const long long plane_stride = (long long)p.D * p.H * p.W * 8;
  float bias[8], s_sum[8], s_sq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { bias[j] = p.bias ? p.bias[cob * 32 + 8 * q + j] : 0.f; s_sum[j] = 0.f; s_sq[j] = 0.f; }
#pragma unroll
  for (int nf = 0; nf < NI; ++nf) {
    const int f = wave * NI + nf;
    const int xh = f % FX, row = f / FX, fy = row % TY, fz = row / TY;
    const int base = (fz * PY + fy) * PX + xh * 16 + l15;
    f32x4 acc0 = f32x4{0, 0, 0, 0}, acc1 = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      V8 b;
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = xs[base + koff[ks][j]];
      acc0 = mfma16<T>(a[ks][0], b, acc0);
      acc1 = mfma16<T>(a[ks][1], b, acc1);
    }
    const int gz = z0 + fz, gy = y0 + fy, gx = x0 + xh * 16 + l15;
    const bool ok = gz < p.D && gy < p.H && gx < p.W;
    float vals[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { vals[j] = acc0[j]; vals[4 + j] = acc1[j]; }
    if (p.stats && ok) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { s_sum[j] += vals[j]; s_sq[j] += vals[j] * vals[j]; }
    }
    V8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float r = vals[j] + bias[j];
      if (p.relu) r = fmaxf(r, 0.f);
      o[j] = from_f32<T>(r);
    }
    if (ok) {
      if (!p.out8) *(V8*)(yout + (long long)(cob * 4 + q) * plane_stride + (((long long)gz * p.H + gy) * p.W + gx) * 8) = o;
      else e4m3_store8<T>((unsigned char*)p.y + (long long)n * p.y_sstride, cob * 4 + q, ((long long)gz * p.H + gy) * p.W + gx,
                          (long long)p.D * p.H * p.W, __builtin_bit_cast(u32x4, o));
    }
  }
  if (p.stats) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float s1 = s_sum[j], s2 = s_sq[j];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
      if (l15 == 0) { red[((wave * 4 + q) * 8 + j) * 2] = s1; red[((wave * 4 + q) * 8 + j) * 2 + 1] = s2; }
    }
    __syncthreads();
    if (tid < 64) {
      const int c = tid >> 1, which = tid & 1;
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) sum += red[((w * 4 + (c >> 3)) * 8 + (c & 7)) * 2 + which];
      p.stats[((long long)tile * p.Cout + cob * 32 + c) * 2 + which] = sum;
    }
  }
}

// fp32 [Cout][Cin][taps] (x optional per-cout scale) -> [cob32][kstep][2][64][8] of T, k = tap * Cin + c
template <typename T>
__global__ void pack_first_conv_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ dst,
                                       int Cout, int Cin, int taps) {
  const int KK = taps * Cin, KS = (KK + 31) / 32;
  const int total = (Cout / 32) * KS * 2 * 64 * 8;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int r = i;
    const int j = r & 7; r >>= 3;
    const int lane = r & 63; r >>= 6;
    const int t = r & 1; r >>= 1;
    const int ks = r % KS;
    const int cob = r / KS;
    const int row = lane & 15, qq = lane >> 4;
    const int co = cob * 32 + 8 * (row >> 2) + 4 * t + (row & 3);
    const int k = 32 * ks + 8 * qq + j;
    float v = 0.f;
    if (k < KK) { v = w[(co * Cin + k % Cin) * taps + k / Cin]; if (scale) v *= scale[co]; }
    dst[i] = from_f32<T>(v);
  }
}

// ------------------------------------------------------------------ max-pool 2^d
template <typename T, int ND>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ x, long long x_ss, T* __restrict__ y,
                                                      long long y_ss, int planes, int Do, int Ho, int Wo) {
  using V8 = typename Vec8<T>::type;
  const long long ovox = (long long)Do * Ho * Wo;
  const long long total = ovox * planes;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int n = blockIdx.y;
  const int pl = (int)(i / ovox);
  const long long r = i - (long long)pl * ovox;
  const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), oz = (int)(r / ((long long)Wo * Ho));
  const int Di = ND == 3 ? Do * 2 : 1, Hi = Ho * 2, Wi = Wo * 2;
  const T* xp = x + n * x_ss + (long long)pl * Di * Hi * Wi * 8;
  float m[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
  for (int a = 0; a < (ND == 3 ? 2 : 1); ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int z = ND == 3 ? oz * 2 + a : 0;
        const V8 v = *(const V8*)(xp + (((long long)z * Hi + oy * 2 + b) * Wi + ox * 2 + c) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], to_f32<T>(v[j]));
      }
  V8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = from_f32<T>(m[j]);
  *(V8*)(y + n * y_ss + (long long)pl * ovox * 8 + r * 8) = o;
}

// max-pool of e4m3 planes (16 channels = 16 B per voxel and plane): rounding to e4m3 is monotonic, so the maximum of the rounded
// values IS the rounded maximum -- the pooled tensor holds the bytes the 16-bit pool + the consumer's loader rounding would give
template <int ND>
__global__ __launch_bounds__(256) void maxpool_q_kernel(const unsigned char* __restrict__ x, long long x_ss, unsigned char* __restrict__ y,
                                                        long long y_ss, int planes16, int Do, int Ho, int Wo) {
  const long long ovox = (long long)Do * Ho * Wo;
  const long long total = ovox * planes16;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int n = blockIdx.y;
  const int pl = (int)(i / ovox);
  const long long r = i - (long long)pl * ovox;
  const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), oz = (int)(r / ((long long)Wo * Ho));
  const int Di = ND == 3 ? Do * 2 : 1, Hi = Ho * 2, Wi = Wo * 2;
  const unsigned char* xp = x + n * x_ss + (long long)pl * Di * Hi * Wi * 16;
  float m[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) m[j] = -INFINITY;
#pragma unroll
  for (int a = 0; a < (ND == 3 ? 2 : 1); ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int z = ND == 3 ? oz * 2 + a : 0;
        const u32x4 v = *(const u32x4*)(xp + (((long long)z * Hi + oy * 2 + b) * Wi + ox * 2 + c) * 16);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          m[4 * d + 0] = fmaxf(m[4 * d + 0], __builtin_amdgcn_cvt_f32_fp8((int)v[d], 0));
          m[4 * d + 1] = fmaxf(m[4 * d + 1], __builtin_amdgcn_cvt_f32_fp8((int)v[d], 1));
          m[4 * d + 2] = fmaxf(m[4 * d + 2], __builtin_amdgcn_cvt_f32_fp8((int)v[d], 2));
          m[4 * d + 3] = fmaxf(m[4 * d + 3], __builtin_amdgcn_cvt_f32_fp8((int)v[d], 3));
        }
      }
  u32x4 o;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(m[4 * d], m[4 * d + 1], 0, false);      // exact: the inputs are e4m3 values
    w = __builtin_amdgcn_cvt_pk_fp8_f32(m[4 * d + 2], m[4 * d + 3], w, true);
    o[d] = (unsigned)w;
  }
  *(u32x4*)(y + n * y_ss + ((long long)pl * ovox + r) * 16) = o;
}

// ------------------------------------------------------------------ transposed conv k2 s2
// One wave = 16 consecutive input x voxels x 32 output channels x all 2^d output
// positions.  A = packed weights [cob32][kstep][pos][t][64][8], B = activations.
struct ConvTParams {
  const void* x; long long x_sstride;
  void* y; long long y_sstride;
  const void* wpk; const float* bias;
  int N, D, H, W, Cin, Cout;   // input grid
  int out8;                    // 1: y = e4m3 planes [Cout / 16][2D][2H][2W][16 B], y_sstride in bytes
};

template <typename T, int ND, bool O8 = false>      // O8: e4m3 planes out (a template flag: its store code costs the 16-bit forms registers)
__global__ __launch_bounds__(256) void convT_kernel(ConvTParams p) {
  using V8 = typename Vec8<T>::type;
  constexpr int NPOS = ND == 3 ? 8 : 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int xg = (p.W + 15) / 16;
  const long long rows = (long long)p.D * p.H * xg;
  const long long wid = (long long)blockIdx.x * 4 + wave;
  if (wid >= rows * p.N) return;
  const int n = (int)(wid / rows);
  const long long r = wid - n * rows;
  const int xb = (int)(r % xg), y = (int)((r / xg) % p.H), z = (int)(r / ((long long)xg * p.H));
  const int cob = blockIdx.y;
  const int x = xb * 16 + l15;
  const bool ok = x < p.W;
  const int xc = ok ? x : p.W - 1;
  const long long in_plane = (long long)p.D * p.H * p.W * 8;
  const T* xin = (const T*)p.x + n * p.x_sstride + (((long long)z * p.H + y) * p.W + xc) * 8;
  const int nk = p.Cin >> 5;
  const V8* wp = (const V8*)p.wpk + (long long)cob * nk * NPOS * 2 * 64 + lane;

  f32x4 acc[NPOS][2];
#pragma unroll
  for (int s = 0; s < NPOS; ++s) { acc[s][0] = f32x4{0, 0, 0, 0}; acc[s][1] = f32x4{0, 0, 0, 0}; }
  for (int ks = 0; ks < nk; ++ks) {
    const V8 b = *(const V8*)(xin + (long long)(ks * 4 + q) * in_plane);
#pragma unroll
    for (int s = 0; s < NPOS; ++s) {
      const V8 a0 = wp[((ks * NPOS + s) * 2 + 0) * 64];
      const V8 a1 = wp[((ks * NPOS + s) * 2 + 1) * 64];
      acc[s][0] = mfma16<T>(a0, b, acc[s][0]);
      acc[s][1] = mfma16<T>(a1, b, acc[s][1]);
    }
  }
  const int Do = ND == 3 ? p.D * 2 : 1, Ho = p.H * 2, Wo = p.W * 2;
  const long long out_plane = (long long)Do * Ho * Wo * 8;
  T* yout = (T*)p.y + n * p.y_sstride + (long long)(cob * 4 + q) * out_plane;
  float bias[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias[j] = p.bias ? p.bias[cob * 32 + q * 8 + j] : 0.f;
  // Stores: a lane owns input voxel x, i.e. the output voxel pair (2x, 2x + 1) = 32 contiguous bytes per plane, but one
  // store instruction moves 16 B per lane -- written lane by lane, each instruction would fill every other 16 B of its
  // cache lines.  So the two x positions (c = 0, 1) are exchanged across lanes first (ds_bpermute): instruction "h"
  // then writes the 16 consecutive output voxels 2 x0 + 16 h + l15, full 256-B runs per channel plane.
  const int src_lo = ((lane & 48) | (l15 >> 1)) * 4, src_hi = src_lo + 8 * 4;     // byte index of the source lane
  const bool odd = l15 & 1;
  const int x0 = xb * 16;
#pragma unroll
  for (int sp = 0; sp < NPOS / 2; ++sp) {
    const int a = ND == 3 ? (sp >> 1) : 0, b = sp & 1;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 oc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      V8 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = from_f32<T>(acc[sp * 2 + c][0][j] + bias[j]);
        o[4 + j] = from_f32<T>(acc[sp * 2 + c][1][j] + bias[4 + j]);
      }
      oc[c] = __builtin_bit_cast(i32x4, o);
    }
    const int oz = ND == 3 ? z * 2 + a : 0;
    T* row = yout + (((long long)oz * Ho + y * 2 + b) * Wo + 2 * x0) * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int src = h ? src_hi : src_lo;
      i32x4 v;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int t0 = __builtin_amdgcn_ds_bpermute(src, oc[0][d]);
        const int t1 = __builtin_amdgcn_ds_bpermute(src, oc[1][d]);
        v[d] = odd ? t1 : t0;
      }
      const int xo = 2 * x0 + 16 * h + l15;                    // output x of this lane in instruction h
      if (xo < Wo) {
        if constexpr (!O8) *(i32x4*)(row + (16 * h + l15) * 8) = v;
        else e4m3_store8<T>((unsigned char*)p.y + (long long)n * p.y_sstride, cob * 4 + q, ((long long)oz * Ho + y * 2 + b) * Wo + xo,
                            (long long)Do * Ho * Wo, __builtin_bit_cast(u32x4, v));
      }
    }
  }
}

// Same operator with the weights of the Cout tile resident in LDS (Cin <= 128: 16 KB per 32 input channels in 3-D) and two 16-voxel
// groups per wave and step: the kernel above fetches 16 KB of weight fragments per k-step and wave through the vector cache to
// move 10 KB of activations -- 4x more cache traffic for the weights than for the tensor, 3.0 TB/s.  Here the only
// global traffic is the tensor itself.
// [r3] Two workgroups per CU (256 registers per lane) and a software pipeline across the iterations: the first form let the
// compiler hoist the loop-invariant weight fragments out of the voxel loop into registers (64 x NK of them: 360 registers at NK = 2,
// 512 + 52 B of scratch at NK = 4 -- one wave per SIMD, every load's latency in the open; 128 -> 64 @ 64^3 ran at 0.3 PF and
// 1.6 TB/s).  Now the fragments stay in LDS (their offset is opaque per iteration) and the activation fragments of the NEXT
// iteration are fetched into the registers of k-step ks as soon as its MFMAs are issued.
// NT = threads per workgroup: 256 (two workgroups per CU) while the weights take at most 64 KB, 512 (one workgroup, 8 waves on one
// copy of the weights) for the 128 KB of Cin = 256 in 3-D.
template <typename T, int ND, int NK, bool O8 = false, int NT = 256>
__global__ __launch_bounds__(NT, NT == 256 ? 2 : 1) void convT_lds_kernel(ConvTParams p) {
  constexpr int NW = NT / 64;                                    // waves per workgroup
  using V8 = typename Vec8<T>::type;
  constexpr int NPOS = ND == 3 ? 8 : 4;
  constexpr int G = 2;                                           // voxel groups per wave and step
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int cob = blockIdx.y;
  {
    const u32x4* wsrc = (const u32x4*)p.wpk + (long long)cob * NK * NPOS * 2 * 64;
    stage_to_lds(smem, wsrc, NK * NPOS * 2 * 64, threadIdx.x, NT);
    if (threadIdx.x < 32) ((float*)(smem + NK * NPOS * 2 * 1024))[threadIdx.x] = p.bias ? p.bias[cob * 32 + threadIdx.x] : 0.f;      // read per epilogue: 8 registers less
  }
  __syncthreads();
  const int xg = (p.W + 15) / 16;
  const long long rows = (long long)p.D * p.H * xg, ngroups = rows * p.N;
  const long long in_plane = (long long)p.D * p.H * p.W * 8;
  const int Do = ND == 3 ? p.D * 2 : 1, Ho = p.H * 2, Wo = p.W * 2;
  const long long out_plane = (long long)Do * Ho * Wo * 8;
  const int src_lo = ((lane & 48) | (l15 >> 1)) * 4, src_hi = src_lo + 8 * 4;
  const bool odd = l15 & 1;

  struct Where { int n, z, y, xb; const T* xin; };
  auto locate = [&](long long wid) -> Where {                      // voxel group wid (clamped: a missing group re-reads the last one, never stored)
    wid = wid < ngroups ? wid : ngroups - 1;
    Where w;
    w.n = (int)(wid / rows);
    const long long r = wid - w.n * rows;
    w.xb = (int)(r % xg); w.y = (int)((r / xg) % p.H); w.z = (int)(r / ((long long)xg * p.H));
    const int xc = min(w.xb * 16 + l15, p.W - 1);
    w.xin = (const T*)p.x + w.n * p.x_sstride + (((long long)w.z * p.H + w.y) * p.W + xc) * 8 + (long long)q * in_plane;
    return w;
  };
  const long long stride = (long long)gridDim.x * NW * G;
  long long g0 = ((long long)blockIdx.x * NW + wave) * G;
  if (g0 >= ngroups) return;
  Where cur[G];
  V8 b[G][NK];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    cur[g] = locate(g0 + g);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) b[g][ks] = *(const V8*)(cur[g].xin + (long long)(ks * 4) * in_plane);
  }
  for (; g0 < ngroups; g0 += stride) {
    Where nxt[G];
#pragma unroll
    for (int g = 0; g < G; ++g) nxt[g] = locate(g0 + stride + g);
    unsigned woff = lane * 16;
    asm volatile("" : "+v"(woff));                               // the weight fragments are read from LDS every iteration (see the header)
    const V8* wl = (const V8*)(smem + woff);
    // the output positions in two halves (z' = 0, 1 in 3-D; y' in 2-D): 64 accumulator registers instead of 128, so that two
    // workgroups fit on a CU without scratch; the activation fragments serve both halves and are replaced during the second
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      constexpr int HP = NPOS / 2;
      f32x4 acc[G][HP][2];
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int s = 0; s < HP; ++s) { acc[g][s][0] = f32x4{0, 0, 0, 0}; acc[g][s][1] = f32x4{0, 0, 0, 0}; }
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
        for (int s = 0; s < HP; ++s) {
          const V8 a0 = wl[((ks * NPOS + hf * HP + s) * 2 + 0) * 64];
          const V8 a1 = wl[((ks * NPOS + hf * HP + s) * 2 + 1) * 64];
#pragma unroll
          for (int g = 0; g < G; ++g) {
            acc[g][s][0] = mfma16<T>(a0, b[g][ks], acc[g][s][0]);
            acc[g][s][1] = mfma16<T>(a1, b[g][ks], acc[g][s][1]);
          }
        }
        if (hf == 1) {
#pragma unroll
          for (int g = 0; g < G; ++g) b[g][ks] = *(const V8*)(nxt[g].xin + (long long)(ks * 4) * in_plane);      // the next iteration's, in place
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (g0 + g >= ngroups) break;
        T* yout = (T*)p.y + cur[g].n * p.y_sstride + (long long)(cob * 4 + q) * out_plane;
        const int x0 = cur[g].xb * 16;
#pragma unroll
        for (int s2 = 0; s2 < HP / 2; ++s2) {
          const int sp = hf * (HP / 2) + s2;
          const int a = ND == 3 ? (sp >> 1) : 0, bb = sp & 1;
          typedef int i32x4 __attribute__((ext_vector_type(4)));
          i32x4 oc[2];
          const f32x4 bias0 = ((const f32x4*)(smem + NK * NPOS * 2 * 1024))[2 * q], bias1 = ((const f32x4*)(smem + NK * NPOS * 2 * 1024))[2 * q + 1];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            V8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              o[j] = from_f32<T>(acc[g][s2 * 2 + c][0][j] + bias0[j]);
              o[4 + j] = from_f32<T>(acc[g][s2 * 2 + c][1][j] + bias1[j]);
            }
            oc[c] = __builtin_bit_cast(i32x4, o);
          }
          const int oz = ND == 3 ? cur[g].z * 2 + a : 0;
          if constexpr (O8) {
            // e4m3 planes: a 16-byte granule = 16 channels of one output voxel = the 8 bytes of this lane and of its partner 16 lanes
            // on (q ^ 1).  Even q assembles the granule of voxel 2x, odd q of voxel 2x + 1: each sends the partner the half it does
            // not store (2 exchanges instead of 16), and one instruction writes 2 planes x 32 consecutive voxels x 16 B.
            unsigned E[2][2];
            e4m3_pack8<T>(__builtin_bit_cast(u32x4, oc[0]), E[0][0], E[0][1]);
            e4m3_pack8<T>(__builtin_bit_cast(u32x4, oc[1]), E[1][0], E[1][1]);
            const bool oq = q & 1;
            const int partner = (lane ^ 16) * 4;
            const unsigned r0 = (unsigned)__builtin_amdgcn_ds_bpermute(partner, (int)(oq ? E[0][0] : E[1][0]));
            const unsigned r1 = (unsigned)__builtin_amdgcn_ds_bpermute(partner, (int)(oq ? E[0][1] : E[1][1]));
            const u32x4 gran = oq ? u32x4{r0, r1, E[1][0], E[1][1]} : u32x4{E[0][0], E[0][1], r0, r1};
            if (x0 + l15 < p.W)
              *(u32x4*)((unsigned char*)p.y + (long long)cur[g].n * p.y_sstride +
                        ((long long)((cob * 4 + q) >> 1) * ((long long)Do * Ho * Wo) + ((long long)oz * Ho + cur[g].y * 2 + bb) * Wo + 2 * (x0 + l15) + (oq ? 1 : 0)) * 16) = gran;
            continue;
          }
          T* row = yout + (((long long)oz * Ho + cur[g].y * 2 + bb) * Wo + 2 * x0) * 8;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int src = h ? src_hi : src_lo;
            i32x4 v;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const int t0 = __builtin_amdgcn_ds_bpermute(src, oc[0][d]);
              const int t1 = __builtin_amdgcn_ds_bpermute(src, oc[1][d]);
              v[d] = odd ? t1 : t0;
            }
            if (2 * x0 + 16 * h + l15 < Wo) {
              if constexpr (!O8) *(i32x4*)(row + (16 * h + l15) * 8) = v;
              else e4m3_store8<T>((unsigned char*)p.y + (long long)cur[g].n * p.y_sstride, cob * 4 + q,
                                  ((long long)oz * Ho + cur[g].y * 2 + bb) * Wo + 2 * x0 + 16 * h + l15, (long long)Do * Ho * Wo,
                                  __builtin_bit_cast(u32x4, v));
            }
          }
        }
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) cur[g] = nxt[g];
  }
}

// Cin > 128: the Cout tile's weights (16 KB per 32 input channels in 3-D) no longer fit in LDS at once, and the direct kernel at
// the top fetches them per wave and k-step through the vector cache (1 KB of weight fragments per voxel and k-step: C5's
// 256 / 512 / 1024-channel levels ran 60-115 us on a few MB of tensor).  Here they pass through LDS in chunks of NK = 4 k-steps
// shared by the workgroup's 4 waves x 2 voxel groups; the accumulators live across the chunks.
template <typename T, int ND, bool O8 = false>
__global__ __launch_bounds__(256) void convT_chunk_kernel(ConvTParams p) {
  constexpr int NK = 4;
  using V8 = typename Vec8<T>::type;
  constexpr int NPOS = ND == 3 ? 8 : 4;
  constexpr int G = 2;                                           // voxel groups per wave and step
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int cob = blockIdx.y;
  const int nchunks = (p.Cin >> 5) / NK;
  const u32x4* wsrc = (const u32x4*)p.wpk + (long long)cob * nchunks * NK * NPOS * 2 * 64;
  const int xg = (p.W + 15) / 16;
  const long long rows = (long long)p.D * p.H * xg, ngroups = rows * p.N;
  const long long in_plane = (long long)p.D * p.H * p.W * 8;
  const int Do = ND == 3 ? p.D * 2 : 1, Ho = p.H * 2, Wo = p.W * 2;
  const long long out_plane = (long long)Do * Ho * Wo * 8;
  float bias[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias[j] = p.bias ? p.bias[cob * 32 + q * 8 + j] : 0.f;
  const int src_lo = ((lane & 48) | (l15 >> 1)) * 4, src_hi = src_lo + 8 * 4;
  const bool odd = l15 & 1;
  const V8* wl = (const V8*)smem + lane;

  for (long long base = (long long)blockIdx.x * 4 * G; base < ngroups; base += (long long)gridDim.x * 4 * G) {      // uniform trip count: barriers inside
    const long long g0 = base + wave * G;
    int n_[G], z_[G], y_[G], xb_[G];
    const T* xin_[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const long long wid = g0 + g < ngroups ? g0 + g : ngroups - 1;      // a missing partner recomputes the last group (never stored)
      n_[g] = (int)(wid / rows);
      const long long r = wid - n_[g] * rows;
      xb_[g] = (int)(r % xg); y_[g] = (int)((r / xg) % p.H); z_[g] = (int)(r / ((long long)xg * p.H));
      const int xc = min(xb_[g] * 16 + l15, p.W - 1);
      xin_[g] = (const T*)p.x + n_[g] * p.x_sstride + (((long long)z_[g] * p.H + y_[g]) * p.W + xc) * 8;
    }
    f32x4 acc[G][NPOS][2];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int s = 0; s < NPOS; ++s) { acc[g][s][0] = f32x4{0, 0, 0, 0}; acc[g][s][1] = f32x4{0, 0, 0, 0}; }
    // chunk ch's weights travel global -> LDS buffer ch & 1 by LDS-DMA (lane-linear, no registers) while chunk ch - 1 is
    // multiplied; the activation fragments of the next chunk are fetched into a second register set meanwhile.  One barrier per
    // chunk: behind it every wave has finished the MFMAs that read the buffer the next copy overwrites.
    constexpr int CHB = NK * NPOS * 2 * 1024;                      // bytes of one chunk (64 KB in 3-D)
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
    auto load_b = [&](int ch, V8 (&bb)[G][NK]) {
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) bb[g][ks] = *(const V8*)(xin_[g] + (long long)((ch * NK + ks) * 4 + q) * in_plane);
    };
    V8 bcur[G][NK], bnext[G][NK];
    __syncthreads();                                               // the previous pass's last chunk is read
    dma_chunk(0);
    load_b(0, bcur);
    for (int ch = 0; ch < nchunks; ++ch) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this chunk's weights and fragments have landed
      __syncthreads();
      if (ch + 1 < nchunks) { dma_chunk(ch + 1); load_b(ch + 1, bnext); }
      const V8* wb = wl + (ch & 1) * (CHB / 16);
#pragma unroll
      for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int s = 0; s < NPOS; ++s) {
          const V8 a0 = wb[((ks * NPOS + s) * 2 + 0) * 64];
          const V8 a1 = wb[((ks * NPOS + s) * 2 + 1) * 64];
#pragma unroll
          for (int g = 0; g < G; ++g) {
            acc[g][s][0] = mfma16<T>(a0, bcur[g][ks], acc[g][s][0]);
            acc[g][s][1] = mfma16<T>(a1, bcur[g][ks], acc[g][s][1]);
          }
        }
      if (ch + 1 < nchunks) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int ks = 0; ks < NK; ++ks) bcur[g][ks] = bnext[g][ks];
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g0 + g >= ngroups) break;
      T* yout = (T*)p.y + n_[g] * p.y_sstride + (long long)(cob * 4 + q) * out_plane;
      const int x0 = xb_[g] * 16;
#pragma unroll
      for (int sp = 0; sp < NPOS / 2; ++sp) {
        const int a = ND == 3 ? (sp >> 1) : 0, bb = sp & 1;
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        i32x4 oc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          V8 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[j] = from_f32<T>(acc[g][sp * 2 + c][0][j] + bias[j]);
            o[4 + j] = from_f32<T>(acc[g][sp * 2 + c][1][j] + bias[4 + j]);
          }
          oc[c] = __builtin_bit_cast(i32x4, o);
        }
        const int oz = ND == 3 ? z_[g] * 2 + a : 0;
        T* row = yout + (((long long)oz * Ho + y_[g] * 2 + bb) * Wo + 2 * x0) * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int src = h ? src_hi : src_lo;
          i32x4 v;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const int t0 = __builtin_amdgcn_ds_bpermute(src, oc[0][d]);
            const int t1 = __builtin_amdgcn_ds_bpermute(src, oc[1][d]);
            v[d] = odd ? t1 : t0;
          }
          if (2 * x0 + 16 * h + l15 < Wo) {
            if constexpr (!O8) *(i32x4*)(row + (16 * h + l15) * 8) = v;
            else e4m3_store8<T>((unsigned char*)p.y + (long long)n_[g] * p.y_sstride, cob * 4 + q,
                                ((long long)oz * Ho + y_[g] * 2 + bb) * Wo + 2 * x0 + 16 * h + l15, (long long)Do * Ho * Wo,
                                __builtin_bit_cast(u32x4, v));
          }
        }
      }
    }
  }
}

// pack ConvTranspose weights fp32 [Cin][Cout][2^d] -> [cob32][kstep][pos][t][64][8]
template <typename T>
__global__ void pack_convT_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cin, int Cout, int npos) {
  const long long total = (long long)Cin * Cout * npos;
  const int nk = Cin >> 5;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int j = r & 7; r >>= 3;
    const int lane = r & 63; r >>= 6;
    const int t = r & 1; r >>= 1;
    const int s = r % npos; r /= npos;
    const int ks = r % nk;
    const int cob = r / nk;
    const int row = lane & 15, qq = lane >> 4;
    const int co = cob * 32 + 8 * (row >> 2) + 4 * t + (row & 3);
    const int ci = ks * 32 + 8 * qq + j;
    dst[i] = from_f32<T>(w[((long long)ci * Cout + co) * npos + s]);
  }
}

// ------------------------------------------------------------------ 1x1 head + softmax / argmax
struct HeadParams {
  const void* x; long long x_sstride; int planes;     // C0/8
  const float* w;      // [ncls][C0] fp32
  const float* bias;   // [ncls]
  float* logits;       // optional, generic strides
  float* probs;        // optional, generic strides
  unsigned char* cls;  // optional, [N][vox]
  long long oN, oC, oD, oH, oW;   // output strides (elements) for logits / probs
  float divisor; int accumulate;  // probs: out = ((accumulate ? out : 0) + p) / divisor  (predict.py:101-110)
  int N, D, H, W;
};

template <typename T, int NCLS>
__global__ __launch_bounds__(256) void head_kernel(HeadParams p) {
  using V8 = typename Vec8<T>::type;
  const long long vox = (long long)p.D * p.H * p.W;
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
  if (v >= vox) return;
  const int n = blockIdx.y;
  const T* xin = (const T*)p.x + n * p.x_sstride + v * 8;
  float l[NCLS];
#pragma unroll
  for (int c = 0; c < NCLS; ++c) l[c] = p.bias[c];
  for (int pl = 0; pl < p.planes; ++pl) {
    const V8 xv = *(const V8*)(xin + (long long)pl * vox * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = to_f32<T>(xv[j]);
#pragma unroll
      for (int c = 0; c < NCLS; ++c) l[c] = fmaf(a, p.w[c * p.planes * 8 + pl * 8 + j], l[c]);
    }
  }
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
  for (int c = 0; c < NCLS; ++c) { e[c] = __expf(l[c] - mx); s += e[c]; }
  const float inv = 1.0f / s;
  // class map = first maximum of the probabilities, exactly what np.argmax over the
  // returned softmax gives (predict.py:38)
  float pm = e[0] * inv; int am = 0;
#pragma unroll
  for (int c = 1; c < NCLS; ++c) { const float pc = e[c] * inv; if (pc > pm) { pm = pc; am = c; } }
  if (p.cls) p.cls[n * vox + v] = (unsigned char)am;
  if (p.probs) {
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
      const float pr = e[c] * inv;
      float* o = p.probs + obase + c * p.oC;
      float r = p.accumulate ? __fadd_rn(*o, pr) : pr;
      if (p.divisor != 1.0f) r = __fdiv_rn(r, p.divisor);
      *o = r;
    }
  }
}

}  // namespace

// ------------------------------------------------------------------ host launchers
int iunet_first_conv_launch(int dtype, int nd, const void* x, int in_dtype, long long sN, long long sC, long long sD,
                            long long sH, long long sW, void* y, long long y_sstride, const void* w,
                            const float* bias, float* stats, int N, int D, int H, int W, int Cin, int Cout, int relu,
                            hipStream_t stream, int out8) {
  IUNET_REQUIRE(Cin >= 1 && Cin <= 4, "first_conv: Cin must be 1..4 (got %d)", Cin);
  IUNET_REQUIRE(Cout % 32 == 0, "first_conv: Cout must be a multiple of 32 (got %d)", Cout);
  FirstConvParams p;
  p.x = x; p.sN = sN; p.sC = sC; p.sD = sD; p.sH = sH; p.sW = sW; p.in_dtype = in_dtype;
  p.y = y; p.y_sstride = y_sstride; p.w = w; p.bias = bias; p.stats = stats;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.nd = nd; p.relu = relu; p.out8 = out8;
  const int TZ = nd == 3 ? 4 : 1, TY = nd == 3 ? 8 : 16, TX = nd == 3 ? 16 : 32;
  dim3 grid(N * ((D + TZ - 1) / TZ) * ((H + TY - 1) / TY) * ((W + TX - 1) / TX), Cout / 32);
#define IUNET_FC(TT, NDV, CI) hipLaunchKernelGGL((first_conv_kernel<TT, NDV, CI>), grid, dim3(256), 0, stream, p)
#define IUNET_FC_CIN(TT, NDV)                                              \
  switch (Cin) { case 1: IUNET_FC(TT, NDV, 1); break; case 2: IUNET_FC(TT, NDV, 2); break; \
                 case 3: IUNET_FC(TT, NDV, 3); break; default: IUNET_FC(TT, NDV, 4); break; }
  if (dtype == 0) { if (nd == 3) { IUNET_FC_CIN(f16, 3) } else { IUNET_FC_CIN(f16, 2) } }
  else            { if (nd == 3) { IUNET_FC_CIN(bf16, 3) } else { IUNET_FC_CIN(bf16, 2) } }
#undef IUNET_FC_CIN
#undef IUNET_FC
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_pack_first_conv_launch(int dtype, const float* w, const float* scale, void* dst, int Cout, int Cin, int taps,
                                 hipStream_t stream) {
  const int total = Cout * (((taps * Cin + 31) / 32) * 32);
  if (dtype == 0) hipLaunchKernelGGL(pack_first_conv_kernel<f16>, dim3((total + 255) / 256), dim3(256), 0, stream, w, scale, (f16*)dst, Cout, Cin, taps);
  else hipLaunchKernelGGL(pack_first_conv_kernel<bf16>, dim3((total + 255) / 256), dim3(256), 0, stream, w, scale, (bf16*)dst, Cout, Cin, taps);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_maxpool_launch(int dtype, int nd, const void* x, long long x_ss, void* y, long long y_ss, int planes, int N,
                         int Do, int Ho, int Wo, hipStream_t stream) {
  const long long total = (long long)Do * Ho * Wo * planes;
  dim3 grid((unsigned)((total + 255) / 256), N);
  if (dtype == 0) {
    if (nd == 3) hipLaunchKernelGGL((maxpool_kernel<f16, 3>), grid, dim3(256), 0, stream, (const f16*)x, x_ss, (f16*)y, y_ss, planes, Do, Ho, Wo);
    else hipLaunchKernelGGL((maxpool_kernel<f16, 2>), grid, dim3(256), 0, stream, (const f16*)x, x_ss, (f16*)y, y_ss, planes, Do, Ho, Wo);
  } else {
    if (nd == 3) hipLaunchKernelGGL((maxpool_kernel<bf16, 3>), grid, dim3(256), 0, stream, (const bf16*)x, x_ss, (bf16*)y, y_ss, planes, Do, Ho, Wo);
    else hipLaunchKernelGGL((maxpool_kernel<bf16, 2>), grid, dim3(256), 0, stream, (const bf16*)x, x_ss, (bf16*)y, y_ss, planes, Do, Ho, Wo);
  }
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_maxpool_q_launch(int nd, const void* x, long long x_ss, void* y, long long y_ss, int planes16, int N, int Do, int Ho, int Wo,
                           hipStream_t stream) {
  const long long total = (long long)Do * Ho * Wo * planes16;
  dim3 grid((unsigned)((total + 255) / 256), N);
  if (nd == 3) hipLaunchKernelGGL((maxpool_q_kernel<3>), grid, dim3(256), 0, stream, (const unsigned char*)x, x_ss, (unsigned char*)y, y_ss, planes16, Do, Ho, Wo);
  else hipLaunchKernelGGL((maxpool_q_kernel<2>), grid, dim3(256), 0, stream, (const unsigned char*)x, x_ss, (unsigned char*)y, y_ss, planes16, Do, Ho, Wo);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_convT_launch(int dtype, int nd, const void* x, long long x_ss, void* y, long long y_ss, const void* wpk,
                       const float* bias, int N, int D, int H, int W, int Cin, int Cout, hipStream_t stream, int out8) {
  IUNET_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, "convT: Cin (%d) and Cout (%d) must be multiples of 32", Cin, Cout);
  IUNET_REQUIRE(!out8 || nd == 3, "convT: e4m3 planes out exist in 3-D only (the K = 128 fp8 convolution's format)");
  ConvTParams p;
  p.x = x; p.x_sstride = x_ss; p.y = y; p.y_sstride = y_ss; p.wpk = wpk; p.bias = bias;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.out8 = out8;
  const long long waves = (long long)N * D * H * ((W + 15) / 16);
  const int nk = Cin / 32;
  // Cin = 128 (4 k-steps): the resident-weights form again since its pipelined rewrite (128 -> 64 @ 64^3 -> e4m3 planes 122 -> 81 us,
  // @ 2 x 32^3 40 -> 31 us against the chunked-weights kernel); IUNET_CONVT_CHUNK4=1: A/B
  static const int chunk4 = getenv("IUNET_CONVT_CHUNK4") ? atoi(getenv("IUNET_CONVT_CHUNK4")) : 0;
  // Cin = 256 in 3-D (128 KB of weights per Cout tile, one workgroup of 8 waves per CU): resident when every workgroup walks several
  // sweeps (256 -> 128 @ 32^3: 47 -> 34 us; at 2 x 16^3 the 128 KB per workgroup cost more than they save: 14.5 -> 17.5 us, so the
  // chunked-weights kernel keeps the small grids -- both kernels sum in the same order, the bits do not depend on the choice).
  // IUNET_CONVT_RES8=0: A/B
  static const int res8 = getenv("IUNET_CONVT_RES8") ? atoi(getenv("IUNET_CONVT_RES8")) : 1;
  if ((nk <= 4 || (nk == 8 && res8 && nd == 3 && waves >= 2048)) && waves >= 256 && !(chunk4 && nk == 4)) {
    // weights of the Cout tile in LDS, grid-stride walk over the voxel groups
    const int npos = nd == 3 ? 8 : 4;
    const int lds = nk * npos * 2 * 1024 + 128;        // + the Cout tile's bias
    const bool big = lds > 65 * 1024;                  // 8 waves on one copy of the weights, one workgroup per CU
    const int per_wg = big ? 16 : 8;                   // voxel groups per workgroup and sweep
    int gx = (int)((waves + per_wg - 1) / per_wg);
    static const int cap_all = getenv("IUNET_CONVT_CAP") ? atoi(getenv("IUNET_CONVT_CAP")) : 512;      // workgroups per launch: two per CU, each walking its share of the groups
    const int cap = (big ? cap_all / 2 : cap_all) / (Cout / 32) > 8 ? (big ? cap_all / 2 : cap_all) / (Cout / 32) : 8;
    if (gx > cap) gx = cap;
    dim3 g2(gx, Cout / 32);
#define CTL(TT, NDV, NKV, NTV) do { if (out8 && NDV == 3) { IUNET_SET_MAX_LDS((convT_lds_kernel<TT, 3, NKV, true, NTV>), lds); \
    hipLaunchKernelGGL((convT_lds_kernel<TT, 3, NKV, true, NTV>), g2, dim3(NTV), lds, stream, p); } else { IUNET_SET_MAX_LDS((convT_lds_kernel<TT, NDV, NKV, false, NTV>), lds); \
    hipLaunchKernelGGL((convT_lds_kernel<TT, NDV, NKV, false, NTV>), g2, dim3(NTV), lds, stream, p); } } while (0)
#define CTL_NK(TT, NDV) switch (nk) { case 1: CTL(TT, NDV, 1, 256); break; case 2: CTL(TT, NDV, 2, 256); break; case 3: CTL(TT, NDV, 3, 256); break; \
    case 4: CTL(TT, NDV, 4, 256); break; default: CTL(TT, NDV, 8, (NDV == 3 ? 512 : 256)); break; }
    if (dtype == 0) { if (nd == 3) { CTL_NK(f16, 3) } else { CTL_NK(f16, 2) } }
    else            { if (nd == 3) { CTL_NK(bf16, 3) } else { CTL_NK(bf16, 2) } }
#undef CTL_NK
#undef CTL
    IUNET_CHECK_HIP(hipGetLastError());
    return IUNET_OK;
  }
  if (nk >= 4 && nk % 4 == 0 && waves >= 64) {
    // weights through LDS in chunks of 4 k-steps (convT_chunk_kernel): few workgroups, each walking many voxel groups per weight pass
    const int npos = nd == 3 ? 8 : 4;
    const int lds = 2 * 4 * npos * 2 * 1024;           // two chunk buffers
    int gx = (int)((waves + 7) / 8);
    const int cap = 512 / (Cout / 32) > 1 ? 512 / (Cout / 32) : 1;
    if (gx > cap) gx = cap;
    dim3 g2(gx, Cout / 32);
#define CTC(TT, NDV) do { if (out8 && NDV == 3) { IUNET_SET_MAX_LDS((convT_chunk_kernel<TT, 3, true>), lds); \
    hipLaunchKernelGGL((convT_chunk_kernel<TT, 3, true>), g2, dim3(256), lds, stream, p); } else { IUNET_SET_MAX_LDS((convT_chunk_kernel<TT, NDV>), lds); \
    hipLaunchKernelGGL((convT_chunk_kernel<TT, NDV>), g2, dim3(256), lds, stream, p); } } while (0)
    if (dtype == 0) { if (nd == 3) CTC(f16, 3); else CTC(f16, 2); } else { if (nd == 3) CTC(bf16, 3); else CTC(bf16, 2); }
#undef CTC
    IUNET_CHECK_HIP(hipGetLastError());
    return IUNET_OK;
  }
  dim3 grid((unsigned)((waves + 3) / 4), Cout / 32);
  if (dtype == 0) {
    if (nd == 3 && out8) hipLaunchKernelGGL((convT_kernel<f16, 3, true>), grid, dim3(256), 0, stream, p);
    else if (nd == 3) hipLaunchKernelGGL((convT_kernel<f16, 3>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((convT_kernel<f16, 2>), grid, dim3(256), 0, stream, p);
  } else {
    if (nd == 3 && out8) hipLaunchKernelGGL((convT_kernel<bf16, 3, true>), grid, dim3(256), 0, stream, p);
    else if (nd == 3) hipLaunchKernelGGL((convT_kernel<bf16, 3>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((convT_kernel<bf16, 2>), grid, dim3(256), 0, stream, p);
  }
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_pack_convT_launch(int dtype, const float* w, void* dst, int Cin, int Cout, int npos, hipStream_t stream) {
  const long long total = (long long)Cin * Cout * npos;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == 0) hipLaunchKernelGGL(pack_convT_kernel<f16>, dim3(blocks), dim3(256), 0, stream, w, (f16*)dst, Cin, Cout, npos);
  else hipLaunchKernelGGL(pack_convT_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, w, (bf16*)dst, Cin, Cout, npos);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_head_launch(int dtype, const void* x, long long x_ss, int C0, const float* w, const float* bias, int ncls,
                      float* logits, float* probs, unsigned char* cls, long long oN, long long oC, long long oD,
                      long long oH, long long oW, float divisor, int accumulate, int N, int D, int H, int W,
                      hipStream_t stream) {
  IUNET_REQUIRE(ncls >= 2 && ncls <= 10, "head: num_classes must be 2..10 (got %d)", ncls);
  HeadParams p;
  p.x = x; p.x_sstride = x_ss; p.planes = C0 / 8; p.w = w; p.bias = bias; p.logits = logits; p.probs = probs; p.cls = cls;
  p.oN = oN; p.oC = oC; p.oD = oD; p.oH = oH; p.oW = oW; p.divisor = divisor; p.accumulate = accumulate;
  p.N = N; p.D = D; p.H = H; p.W = W;
  const long long vox = (long long)D * H * W;
  dim3 grid((unsigned)((vox + 255) / 256), N);
#define IUNET_HEAD(TT, NC) case NC: hipLaunchKernelGGL((head_kernel<TT, NC>), grid, dim3(256), 0, stream, p); break;
#define IUNET_HEAD_ALL(TT) switch (ncls) { IUNET_HEAD(TT, 2) IUNET_HEAD(TT, 3) IUNET_HEAD(TT, 4) IUNET_HEAD(TT, 5) \
  IUNET_HEAD(TT, 6) IUNET_HEAD(TT, 7) IUNET_HEAD(TT, 8) IUNET_HEAD(TT, 9) IUNET_HEAD(TT, 10) }
  if (dtype == 0) { IUNET_HEAD_ALL(f16) } else { IUNET_HEAD_ALL(bf16) }
#undef IUNET_HEAD_ALL
#undef IUNET_HEAD
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}
