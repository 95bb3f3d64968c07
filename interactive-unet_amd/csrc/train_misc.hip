// Backward kernels of the small layers: transposed-conv data gradient (MFMA, operands from
// global), transposed-conv weight/bias gradient and first-conv weight gradient (VALU + LDS;
// < 2 % of the training FLOPs).  Partial results go to per-block slabs reduced in a fixed
// order by iunet_reduce_slab.
#include "common.h"

namespace {

template <typename T> using V8T = typename Vec8<T>::type;

// ------------------------------------------------------------------ convT k2 s2: dx = W . dy(gathered)
// dx[ci][v] = sum_{pos, co} W[ci][co][pos] * dy[co][2v + pos].
// wave = 16 input voxels x 32 ci; A = packed [cib32][pos][kc][t][64][8] (rows ci, k = co).
struct ConvTDgradParams {
  const void* dy; long long dy_ss;
  void* dx; long long dx_ss;
  const void* wpk;
  int N, D, H, W, Cin, Cout;    // input grid of the forward transposed conv
};

template <typename T, int ND>
__global__ __launch_bounds__(256) void convT_dgrad_kernel(ConvTDgradParams p) {
  using V8 = V8T<T>;
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
  const int cib = blockIdx.y;
  const int x = xb * 16 + l15;
  const bool ok = x < p.W;
  const int xc = ok ? x : p.W - 1;
  const int Do = ND == 3 ? p.D * 2 : 1, Ho = p.H * 2, Wo = p.W * 2;
  const long long out_plane = (long long)Do * Ho * Wo * 8;
  const T* dyin = (const T*)p.dy + n * p.dy_ss;
  const int nk = p.Cout >> 5;
  const V8* wp = (const V8*)p.wpk + (long long)cib * NPOS * nk * 2 * 64 + lane;
  f32x4 acc0 = f32x4{0, 0, 0, 0}, acc1 = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < NPOS; ++s) {
    const int a = ND == 3 ? (s >> 2) : 0, b = (s >> 1) & 1, c = s & 1;
    const int oz = ND == 3 ? z * 2 + a : 0;
    const long long voff = (((long long)oz * Ho + y * 2 + b) * Wo + xc * 2 + c) * 8;
    for (int kc = 0; kc < nk; ++kc) {
      const V8 bf = *(const V8*)(dyin + (long long)(kc * 4 + q) * out_plane + voff);
      const V8 a0 = wp[((s * nk + kc) * 2 + 0) * 64];
      const V8 a1 = wp[((s * nk + kc) * 2 + 1) * 64];
      acc0 = mfma16<T>(a0, bf, acc0);
      acc1 = mfma16<T>(a1, bf, acc1);
    }
  }
  const long long in_plane = (long long)p.D * p.H * p.W * 8;
  V8 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) { o[j] = from_f32<T>(acc0[j]); o[4 + j] = from_f32<T>(acc1[j]); }
  if (ok)
    *(V8*)((T*)p.dx + n * p.dx_ss + (long long)(cib * 4 + q) * in_plane + (((long long)z * p.H + y) * p.W + x) * 8) = o;
}

// fp32 [Cin][Cout][npos] -> [cib32][pos][kc][t][64][8], rows = ci (8g + 4t + r), k = co
template <typename T>
__global__ void pack_convT_dgrad_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cin, int Cout, int npos) {
  const long long total = (long long)Cin * Cout * npos;
  const int nk = Cout >> 5;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int j = r & 7; r >>= 3;
    const int lane = r & 63; r >>= 6;
    const int t = r & 1; r >>= 1;
    const int kc = r % nk; r /= nk;
    const int s = r % npos;
    const int cib = r / npos;
    const int row = lane & 15, qq = lane >> 4;
    const int ci = cib * 32 + 8 * (row >> 2) + 4 * t + (row & 3);
    const int co = kc * 32 + 8 * qq + j;
    dst[i] = from_f32<T>(w[((long long)ci * Cout + co) * npos + s]);
  }
}

// ------------------------------------------------------------------ convT weight + bias gradient (VALU)
// dW[ci][co][pos] = sum_v x[ci][v] * dy[co][2v+pos];  db[co] = sum dy[co][.]
// block: 32 ci x 32 co x all pos over a run of input voxels; thread = (ci, 4 co).
struct ConvTWgradParams {
  const void* x; long long x_ss;
  const void* dy; long long dy_ss;
  float* wslab;    // [nb][Cin][Cout][NPOS]
  float* bslab;    // [nb][Cout]
  int N, D, H, W, Cin, Cout, per_block;
};

template <typename T, int ND>
__global__ __launch_bounds__(256) void convT_wgrad_kernel(ConvTWgradParams p) {
  constexpr int NPOS = ND == 3 ? 8 : 4;
  constexpr int VC = 32;                       // voxels per LDS sub-chunk
  __shared__ float xs[VC][33];
  __shared__ float dys[VC][NPOS][32];
  const int t = threadIdx.x;
  const int ci = t & 31, cog = t >> 5;
  const int cib = blockIdx.y, cob = blockIdx.z;
  const long long vox = (long long)p.D * p.H * p.W;
  const long long total = vox * p.N;
  const long long v0 = (long long)blockIdx.x * p.per_block, v1 = min(v0 + p.per_block, total);
  const int Do = ND == 3 ? p.D * 2 : 1, Ho = p.H * 2, Wo = p.W * 2;
  const long long in_plane = vox * 8, out_plane = (long long)Do * Ho * Wo * 8;
  float acc[NPOS][4], bacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // bacc: channels (t & 3) * 8 + j
#pragma unroll
  for (int s = 0; s < NPOS; ++s)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[s][k] = 0.f;
  for (long long vb = v0; vb < v1; vb += VC) {
    __syncthreads();
    // stage x: 32 voxels x 32 ci (4 planes x 8)  -> 128 16-B items; dy: 32 voxels x NPOS x 4 planes
    for (int it = t; it < VC * 4; it += 256) {
      const int vv = it >> 2, pl = it & 3;
      const long long gv = vb + vv;
      V8T<T> val;
#pragma unroll
      for (int j = 0; j < 8; ++j) val[j] = from_f32<T>(0.f);
      if (gv < v1) {
        const int n = (int)(gv / vox);
        const long long r = gv - n * vox;
        val = *(const V8T<T>*)((const T*)p.x + n * p.x_ss + (long long)(cib * 4 + pl) * in_plane + r * 8);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) xs[vv][pl * 8 + j] = to_f32<T>(val[j]);
    }
    for (int it = t; it < VC * NPOS * 4; it += 256) {
      const int pl = it & 3, s = (it >> 2) % NPOS, vv = it / (4 * NPOS);
      const long long gv = vb + vv;
      V8T<T> val;
#pragma unroll
      for (int j = 0; j < 8; ++j) val[j] = from_f32<T>(0.f);
      if (gv < v1) {
        const int n = (int)(gv / vox);
        const long long r = gv - n * vox;
        const int x = (int)(r % p.W), y = (int)((r / p.W) % p.H), z = (int)(r / ((long long)p.W * p.H));
        const int a = ND == 3 ? (s >> 2) : 0, b = (s >> 1) & 1, c = s & 1;
        const int oz = ND == 3 ? z * 2 + a : 0;
        val = *(const V8T<T>*)((const T*)p.dy + n * p.dy_ss + (long long)(cob * 4 + pl) * out_plane +
                               (((long long)oz * Ho + y * 2 + b) * Wo + x * 2 + c) * 8);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = to_f32<T>(val[j]); dys[vv][s][pl * 8 + j] = f; bacc[j] += f; }
    }
    __syncthreads();
#pragma unroll 4
    for (int vv = 0; vv < VC; ++vv) {
      const float xv = xs[vv][ci];
#pragma unroll
      for (int s = 0; s < NPOS; ++s)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float d = dys[vv][s][cog * 4 + k];
          acc[s][k] = fmaf(xv, d, acc[s][k]);
        }
    }
  }
  float* ws = p.wslab + (long long)blockIdx.x * p.Cin * p.Cout * NPOS;
#pragma unroll
  for (int s = 0; s < NPOS; ++s)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      ws[((long long)(cib * 32 + ci) * p.Cout + cob * 32 + cog * 4 + k) * NPOS + s] = acc[s][k];
  if (cib == 0) {       // bias gradient: threads with equal (t & 3) own the same 8 channels -> sum them through LDS
    __syncthreads();
    float* red = &dys[0][0][0];                       // reuse: [256][8]
#pragma unroll
    for (int j = 0; j < 8; ++j) red[t * 8 + j] = bacc[j];
    __syncthreads();
    if (t < 32) {
      const int pl = t >> 3, j = t & 7;
      float sum = 0.f;
      for (int k = pl; k < 256; k += 4) sum += red[k * 8 + j];
      p.bslab[(long long)blockIdx.x * p.Cout + cob * 32 + pl * 8 + j] = sum;
    }
  }
}

// ------------------------------------------------------------------ first conv weight gradient (VALU)
// dW[co][ci][tap] = sum_v dy[co][v] * x[ci][v + tap - 1]; x read with the caller's strides.
struct FirstWgradParams {
  const void* x; long long sN, sC, sD, sH, sW; int in_dtype;
  const void* dy; long long dy_ss;
  float* slab;     // [ntiles][Cout][Cin][taps]
  int N, D, H, W, Cin, Cout;
  int tilesZ, tilesY, tilesX;
};

__device__ __forceinline__ float load_in2(const void* p, long long off, int dt) {
  switch (dt) {
    case 0: return ((const float*)p)[off];
    case 1: return (float)((const f16*)p)[off];
    case 2: return (float)((const unsigned char*)p)[off] / 255.0f;
    default: return (float)((const bf16*)p)[off];
  }
}

template <typename T, int ND, int CIN>
__global__ __launch_bounds__(256) void first_wgrad_kernel(FirstWgradParams p) {
  constexpr int TZ = ND == 3 ? 4 : 1, TY = ND == 3 ? 8 : 16, TX = ND == 3 ? 16 : 32, PADZ = ND == 3 ? 1 : 0;
  constexpr int PZ = TZ + 2 * PADZ, PY = TY + 2, PX = TX + 2, NPIX = PZ * PY * PX, NVOX = TZ * TY * TX;
  constexpr int TAPS = ND == 3 ? 27 : 9;
  constexpr int TPG = (TAPS + 7) / 8;          // taps per thread group
  __shared__ float xs[CIN][NPIX];
  __shared__ T dys[NVOX][34];
  const int t = threadIdx.x;
  const int co = t & 31, tg = t >> 5;
  const int cob = blockIdx.y;
  const int tiles_per_sample = p.tilesZ * p.tilesY * p.tilesX;
  const int tile = blockIdx.x;
  const int n = tile / tiles_per_sample;
  int trem = tile - n * tiles_per_sample;
  const int tz_i = trem / (p.tilesY * p.tilesX);
  trem -= tz_i * p.tilesY * p.tilesX;
  const int ty_i = trem / p.tilesX, tx_i = trem - ty_i * p.tilesX;
  const int z0 = tz_i * TZ, y0 = ty_i * TY, x0 = tx_i * TX;
  for (int it = t; it < NPIX * CIN; it += 256) {
    const int c = it / NPIX, pix = it - c * NPIX;
    const int px = pix % PX, t2 = pix / PX, py = t2 % PY, pz = t2 / PY;
    const int gz = z0 + pz - PADZ, gy = y0 + py - 1, gx = x0 + px - 1;
    float v = 0.f;
    if ((unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
      v = to_f32<T>(from_f32<T>(load_in2(p.x, n * p.sN + c * p.sC + gz * p.sD + gy * p.sH + gx * p.sW, p.in_dtype)));
    xs[c][pix] = v;
  }
  const long long plane = (long long)p.D * p.H * p.W * 8;
  for (int it = t; it < NVOX * 4; it += 256) {
    const int pl = it & 3, vv = it >> 2;
    const int px = vv % TX, t2 = vv / TX, py = t2 % TY, pz = t2 / TY;
    const int gz = z0 + pz, gy = y0 + py, gx = x0 + px;
    V8T<T> val;
#pragma unroll
    for (int j = 0; j < 8; ++j) val[j] = from_f32<T>(0.f);
    if (gz < p.D && gy < p.H && gx < p.W)
      val = *(const V8T<T>*)((const T*)p.dy + n * p.dy_ss + (long long)(cob * 4 + pl) * plane +
                             (((long long)gz * p.H + gy) * p.W + gx) * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) dys[vv][pl * 8 + j] = val[j];
  }
  __syncthreads();
  float acc[TPG][CIN];
  int toff[TPG];
#pragma unroll
  for (int k = 0; k < TPG; ++k) {
    const int tap = min(tg * TPG + k, TAPS - 1);
    const int dz = ND == 3 ? tap / 9 : 0, dy_ = (tap / 3) % 3, dx = tap % 3;
    toff[k] = (dz * PY + dy_) * PX + dx;
#pragma unroll
    for (int c = 0; c < CIN; ++c) acc[k][c] = 0.f;
  }
  for (int vv = 0; vv < NVOX; ++vv) {
    const int px = vv % TX, t2 = vv / TX, py = t2 % TY, pz = t2 / TY;
    const int pix0 = (pz * PY + py) * PX + px;
    const float d = to_f32<T>(dys[vv][co]);
#pragma unroll
    for (int k = 0; k < TPG; ++k)
#pragma unroll
      for (int c = 0; c < CIN; ++c) acc[k][c] = fmaf(d, xs[c][pix0 + toff[k]], acc[k][c]);
  }
  float* slab = p.slab + (long long)tile * p.Cout * CIN * TAPS;
#pragma unroll
  for (int k = 0; k < TPG; ++k) {
    const int tap = tg * TPG + k;
    if (tap < TAPS) {
#pragma unroll
      for (int c = 0; c < CIN; ++c) slab[((long long)(cob * 32 + co) * CIN + c) * TAPS + tap] = acc[k][c];
    }
  }
}

}  // namespace

#define DT_OK(dt) IUNET_REQUIRE((dt) == 0 || (dt) == 1, "dtype must be 0 (f16) or 1 (bf16), got %d", (dt))

extern "C" {

int iunet_pack_convT_dgrad(int dtype, const void* w, void* dst, int Cin, int Cout, int npos, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(w && dst, "pack_convT_dgrad: null pointer");
  IUNET_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, "pack_convT_dgrad: channels must be multiples of 32");
  const long long total = (long long)Cin * Cout * npos;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == 0) hipLaunchKernelGGL(pack_convT_dgrad_kernel<f16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)w, (f16*)dst, Cin, Cout, npos);
  else hipLaunchKernelGGL(pack_convT_dgrad_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)w, (bf16*)dst, Cin, Cout, npos);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

// N, D, H, W, Cin, Cout describe the FORWARD transposed conv (x: Cin planes on D,H,W; dy: Cout planes on 2x grid)
int iunet_convT_dgrad(int dtype, int nd, const void* dy, long long dy_ss, void* dx, long long dx_ss, const void* wpk,
                      int N, int D, int H, int W, int Cin, int Cout, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(dy && dx && wpk, "convT_dgrad: null pointer");
  ConvTDgradParams p;
  p.dy = dy; p.dy_ss = dy_ss; p.dx = dx; p.dx_ss = dx_ss; p.wpk = wpk; p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  const long long waves = (long long)N * D * H * ((W + 15) / 16);
  dim3 grid((unsigned)((waves + 3) / 4), Cin / 32);
  if (dtype == 0) { if (nd == 3) hipLaunchKernelGGL((convT_dgrad_kernel<f16, 3>), grid, dim3(256), 0, (hipStream_t)stream, p);
                    else hipLaunchKernelGGL((convT_dgrad_kernel<f16, 2>), grid, dim3(256), 0, (hipStream_t)stream, p); }
  else { if (nd == 3) hipLaunchKernelGGL((convT_dgrad_kernel<bf16, 3>), grid, dim3(256), 0, (hipStream_t)stream, p);
         else hipLaunchKernelGGL((convT_dgrad_kernel<bf16, 2>), grid, dim3(256), 0, (hipStream_t)stream, p); }
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_convT_wgrad_blocks(int N, int D, int H, int W) {
  const long long total = (long long)N * D * H * W;
  long long nb = (total + 2047) / 2048;
  if (nb > 256) nb = 256;
  return (int)(nb < 1 ? 1 : nb);
}

// wslab: [blocks][Cin][Cout][npos] floats, bslab: [blocks][Cout]; reduce both with iunet_reduce_slab
int iunet_convT_wgrad(int dtype, int nd, const void* x, long long x_ss, const void* dy, long long dy_ss, void* wslab,
                      void* bslab, int N, int D, int H, int W, int Cin, int Cout, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(x && dy && wslab && bslab, "convT_wgrad: null pointer");
  IUNET_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, "convT_wgrad: channels must be multiples of 32");
  ConvTWgradParams p;
  p.x = x; p.x_ss = x_ss; p.dy = dy; p.dy_ss = dy_ss; p.wslab = (float*)wslab; p.bslab = (float*)bslab;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  const int nb = iunet_convT_wgrad_blocks(N, D, H, W);
  const long long total = (long long)N * D * H * W;
  p.per_block = (int)(((total + nb - 1) / nb + 31) / 32 * 32);
  dim3 grid(nb, Cin / 32, Cout / 32);
  if (dtype == 0) { if (nd == 3) hipLaunchKernelGGL((convT_wgrad_kernel<f16, 3>), grid, dim3(256), 0, (hipStream_t)stream, p);
                    else hipLaunchKernelGGL((convT_wgrad_kernel<f16, 2>), grid, dim3(256), 0, (hipStream_t)stream, p); }
  else { if (nd == 3) hipLaunchKernelGGL((convT_wgrad_kernel<bf16, 3>), grid, dim3(256), 0, (hipStream_t)stream, p);
         else hipLaunchKernelGGL((convT_wgrad_kernel<bf16, 2>), grid, dim3(256), 0, (hipStream_t)stream, p); }
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_first_conv_wgrad_tiles(int nd, int N, int D, int H, int W) {
  const int TZ = nd == 3 ? 4 : 1, TY = nd == 3 ? 8 : 16, TX = nd == 3 ? 16 : 32;
  return N * ((D + TZ - 1) / TZ) * ((H + TY - 1) / TY) * ((W + TX - 1) / TX);
}

// slab: [tiles][Cout][Cin][taps] floats; reduce with iunet_reduce_slab
int iunet_first_conv_wgrad(int dtype, int nd, const void* x, int in_dtype, const long long* in_strides, const void* dy,
                           long long dy_ss, void* slab, int N, int D, int H, int W, int Cin, int Cout, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(x && dy && slab && in_strides, "first_conv_wgrad: null pointer");
  IUNET_REQUIRE(Cin >= 1 && Cin <= 4 && Cout % 32 == 0, "first_conv_wgrad: Cin 1..4, Cout multiple of 32");
  FirstWgradParams p;
  p.x = x; p.sN = in_strides[0]; p.sC = in_strides[1]; p.sD = in_strides[2]; p.sH = in_strides[3]; p.sW = in_strides[4];
  p.in_dtype = in_dtype; p.dy = dy; p.dy_ss = dy_ss; p.slab = (float*)slab;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  const int TZ = nd == 3 ? 4 : 1, TY = nd == 3 ? 8 : 16, TX = nd == 3 ? 16 : 32;
  p.tilesZ = (D + TZ - 1) / TZ; p.tilesY = (H + TY - 1) / TY; p.tilesX = (W + TX - 1) / TX;
  dim3 grid(p.tilesZ * p.tilesY * p.tilesX * N, Cout / 32);
#define FW(TT, NDV, CI) hipLaunchKernelGGL((first_wgrad_kernel<TT, NDV, CI>), grid, dim3(256), 0, (hipStream_t)stream, p)
#define FW_CIN(TT, NDV) switch (Cin) { case 1: FW(TT, NDV, 1); break; case 2: FW(TT, NDV, 2); break; case 3: FW(TT, NDV, 3); break; default: FW(TT, NDV, 4); break; }
  if (dtype == 0) { if (nd == 3) { FW_CIN(f16, 3) } else { FW_CIN(f16, 2) } }
  else            { if (nd == 3) { FW_CIN(bf16, 3) } else { FW_CIN(bf16, 2) } }
#undef FW_CIN
#undef FW
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

}  // extern "C"
