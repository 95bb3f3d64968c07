// 3^d convolution for Cout tiles of 32 channels (the full-resolution layers that hold ~65 % of
// the network's FLOPs), second structure.  Ablation of the first structure (tools/ablate_conv.py,
// 64->32 at 128^3): MFMA-only 138 us, full kernel 297 us -- activation staging, per-wave weight
// fetches through the vector cache and the stores were serialised with the MFMA phase.  Here
//   * BOTH MFMA operands come from LDS: the halo tile of a 16-channel chunk (2 planes) and that
//     chunk's packed weights (28 KB), 62 KB per workgroup -> two workgroups per CU;
//   * a k-step of 32 is 2 taps x 16 channels; the two taps of a k-step differ in (dz, dx) only, so
//     one set of NR+2 activation row fragments serves the three dy taps of that k-step column:
//     16 LDS fragment reads per 48 MFMAs instead of 30 (at Cout = 32 the LDS pipe, not the matrix
//     pipe, is the binding unit);
//   * each workgroup is persistent over a contiguous run of tiles and prefetches the NEXT step's
//     bytes global -> registers (16 x 16 B per lane) while the current step runs on the matrix
//     cores; no other VMEM instruction is issued in that phase, so hipcc's in-order vmcnt waits
//     cannot drain the prefetch early; the registers go to LDS behind one barrier.
#include "common.h"
#include <cstdlib>

namespace {

template <int ND> struct Tile2;
template <> struct Tile2<3> { static constexpr int TZ = 4, TY = 8, TX = 16, PADZ = 1, TAPS = 27; };
template <> struct Tile2<2> { static constexpr int TZ = 1, TY = 16, TX = 32, PADZ = 0, TAPS = 9; };

struct ConvV2Params {
  const void* x;  long long x_sstride;
  void* y;        long long y_sstride;
  const void* wpk;                            // packed weights [cob][chunk16][column pair][dy][2][64][8]
  const float* bias;
  float* stats;                               // [ntiles][Cout][2] or null
  int N, D, H, W, Cin, Cout;
  int tilesZ, tilesY, tilesX;
  int epi;
  int dbg;                                    // profiling only (IUNET_V2_DBG): 1 no refill after step 0, 2 no MFMA phase, 4 no stores
};

template <typename T, int ND>
__global__ __launch_bounds__(256, 2) void conv3_v2_kernel(ConvV2Params p) {
  using TL = Tile2<ND>;
  using V8 = typename Vec8<T>::type;
  constexpr int MI = 2;
  constexpr int TZ = TL::TZ, TY = TL::TY, TX = TL::TX, PADZ = TL::PADZ, TAPS = TL::TAPS;
  constexpr int PZ = TZ + 2 * PADZ, PY = TY + 2, PX = TX + 2;
  constexpr int NPIX = PZ * PY * PX;
  constexpr int PLANE = ((NPIX * 16 + 255) / 256) * 256;
  constexpr int FX = TX / 16;
  constexpr int NI = 8;
  constexpr int CP = 2;                                // planes (of 8 channels) per chunk
  constexpr int NCOL = TAPS / 3;                       // (dz, dx) columns of the filter: 9 (3-D) or 3 (2-D)
  constexpr int NCMB = (NCOL + 1) / 2;                 // column pairs (the last one is half empty)
  constexpr int KS = NCMB * 3;                         // k-steps per chunk: pair x dy
  constexpr int NR = NI / FX;                          // output rows per wave and x half
  constexpr int AIT = (NPIX + 255) / 256;              // activation pixels per thread
  constexpr int WBYTES = KS * MI * 1024;               // one chunk of packed weights
  constexpr int WIT = (WBYTES / 16 + 255) / 256;       // 16-byte weight items per thread
  constexpr int OFF_W = CP * PLANE;

  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int cob = blockIdx.y;
  const int tiles_per_sample = p.tilesZ * p.tilesY * p.tilesX;
  const int ntiles = tiles_per_sample * p.N;
  // workgroups b and b + 8 share an XCD (and its L2): give each XCD a contiguous stretch of the tile
  // sequence so that neighbouring runs re-use each other's halo lines in that L2 (speed only)
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int t_begin = (int)((long long)lb * ntiles / gridDim.x);
  const int t_end = (int)((long long)(lb + 1) * ntiles / gridDim.x);
  const int nchunk = p.Cin >> 4;
  const int nsteps = (t_end - t_begin) * nchunk;
  if (nsteps <= 0) return;
  const long long plane_stride = (long long)p.D * p.H * p.W * 8;

  // per-thread pixel slots of the halo tile: packed (pz, py, px)
  int pcoord[AIT];
#pragma unroll
  for (int it = 0; it < AIT; ++it) {
    const int pix = min(tid + it * 256, NPIX - 1);
    const int px = pix % PX, t2 = pix / PX;
    pcoord[it] = px | ((t2 % PY) << 8) | ((t2 / PY) << 16);
  }
  // lanes with q < 2 read the first column of a pair, lanes with q >= 2 the second one
  int col_off[NCMB];
#pragma unroll
  for (int c = 0; c < NCMB; ++c) {
    const int col = min(2 * c + (q >> 1), NCOL - 1);           // the missing partner re-reads a valid column (zero weights)
    const int dz = col / 3, dx = col % 3;
    col_off[c] = (dz * PY * PX + dx) * 16;
  }
  // LDS byte address of this lane's piece of halo row r (0 .. NR+1) of x half xh, column (0, 0)
  const int row0 = (wave * NI) / FX;                         // first output row of this wave (tile row index)
  const int rbase = (q & 1) * PLANE + ((((row0 / TY) * PY + (row0 % TY)) * PX) + l15) * 16;
  float bias_r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias_r[j] = (p.epi != 0) ? p.bias[cob * 32 + 8 * q + j] : 0.f;

  const V8* wbase = (const V8*)p.wpk + (long long)cob * nchunk * (WBYTES / 16);

  u32x4 areg[AIT][CP];
  u32x4 wreg[WIT];
  unsigned okmask = 0;      // bit it: pixel slot `it` of the prefetched tile lies inside the image

  auto tile_origin = [&](int tile, int& n_img, int& z0, int& y0, int& x0) {
    n_img = tile / tiles_per_sample;
    int trem = tile - n_img * tiles_per_sample;
    const int tz_i = trem / (p.tilesY * p.tilesX);
    trem -= tz_i * p.tilesY * p.tilesX;
    const int ty_i = trem / p.tilesX;
    z0 = tz_i * TZ; y0 = ty_i * TY; x0 = (trem - ty_i * p.tilesX) * TX;
  };

  // issue the global loads of step s (activation chunk + weight chunk) into registers
  auto prefetch = [&](int s) {
    const int tile = t_begin + s / nchunk, chunk = s - (s / nchunk) * nchunk;
    int n_img, z0, y0, x0;
    tile_origin(tile, n_img, z0, y0, x0);
    const T* xc = (const T*)p.x + (long long)n_img * p.x_sstride + (long long)chunk * CP * plane_stride;
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      const int px = pcoord[it] & 255, py = (pcoord[it] >> 8) & 255, pz = pcoord[it] >> 16;
      const int gz = z0 + pz - PADZ, gy = y0 + py - 1, gx = x0 + px - 1;
      const bool ok = (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      const int cz = min(max(gz, 0), p.D - 1), cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
      const long long goff = (((long long)cz * p.H + cy) * p.W + cx) * 8;
      // raw loads only: nothing may consume them before commit(), or the wait lands in front of the MFMA phase
#pragma unroll
      for (int k = 0; k < CP; ++k) areg[it][k] = *(const u32x4*)(xc + k * plane_stride + goff);
      okmask = ok ? (okmask | (1u << it)) : (okmask & ~(1u << it));
    }
    if (nchunk > 1 || s == 0) {
      const u32x4* wsrc = (const u32x4*)(wbase + (long long)chunk * (WBYTES / 16));
#pragma unroll
      for (int it = 0; it < WIT; ++it) {
        const int idx = tid + it * 256;
        wreg[it] = wsrc[min(idx, WBYTES / 16 - 1)];
      }
    }
  };
  auto commit = [&](int s) {      // registers -> LDS
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      const int pix = tid + it * 256;
      if (pix < NPIX) {
        const bool ok = (okmask >> it) & 1u;
#pragma unroll
        for (int k = 0; k < CP; ++k) *(u32x4*)(smem + k * PLANE + pix * 16) = ok ? areg[it][k] : u32x4{0u, 0u, 0u, 0u};
      }
    }
    if (nchunk > 1 || s == 0) {
#pragma unroll
      for (int it = 0; it < WIT; ++it) {
        const int idx = tid + it * 256;
        if (idx < WBYTES / 16) *(u32x4*)(smem + OFF_W + idx * 16) = wreg[it];
      }
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int m = 0; m < MI; ++m)
#pragma unroll
    for (int n = 0; n < NI; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  prefetch(0);
  commit(0);
  __syncthreads();

  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps && !(p.dbg & 1)) prefetch(s + 1);          // in flight during the MFMA phase below

    // ---- all k-steps of this (tile, 16-channel chunk) from LDS ----
    if (!(p.dbg & 2))
#pragma unroll
    for (int c = 0; c < NCMB; ++c) {
      V8 R[FX][NR + 2];
#pragma unroll
      for (int xh = 0; xh < FX; ++xh)
#pragma unroll
        for (int r = 0; r < NR + 2; ++r) R[xh][r] = *(const V8*)(smem + rbase + (r * PX + xh * 16) * 16 + col_off[c]);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const V8 a0 = *(const V8*)(smem + OFF_W + ((c * 3 + dy) * MI + 0) * 1024 + lane * 16);
        const V8 a1 = *(const V8*)(smem + OFF_W + ((c * 3 + dy) * MI + 1) * 1024 + lane * 16);
#pragma unroll
        for (int n = 0; n < NI; ++n) {
          const V8 b = R[n % FX][n / FX + dy];
          acc[0][n] = mfma16<T>(a0, b, acc[0][n]);
          acc[1][n] = mfma16<T>(a1, b, acc[1][n]);
        }
      }
    }

    const int tile = t_begin + s / nchunk, chunk = s - (s / nchunk) * nchunk;
    if (chunk == nchunk - 1) {
      // ---- epilogue of this tile ----
      int n_img, z0, y0, x0;
      tile_origin(tile, n_img, z0, y0, x0);
      T* yout = (T*)p.y + (long long)n_img * p.y_sstride;
      float s_sum[8], s_sq[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { s_sum[j] = 0.f; s_sq[j] = 0.f; }
#pragma unroll
      for (int n = 0; n < NI; ++n) {
        const int f = wave * NI + n;
        const int xh = f % FX, row = f / FX;
        const int gz = z0 + row / TY, gy = y0 + row % TY, gx = x0 + xh * 16 + l15;
        const bool ok = gz < p.D && gy < p.H && gx < p.W;
        float vals[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { vals[j] = acc[0][n][j]; vals[4 + j] = acc[1][n][j]; }
        if (p.stats != nullptr && ok) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { s_sum[j] += vals[j]; s_sq[j] += vals[j] * vals[j]; }
        }
        V8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float r = vals[j] + bias_r[j];
          if (p.epi == 2) r = fmaxf(r, 0.f);
          o[j] = from_f32<T>(r);
        }
        if (ok && !(p.dbg & 4)) *(V8*)(yout + (long long)(cob * 4 + q) * plane_stride + (((long long)gz * p.H + gy) * p.W + gx) * 8) = o;
        acc[0][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[1][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (p.stats != nullptr) {
        // partial BatchNorm sums of this tile: 16 x-lanes by shuffles, 4 waves through LDS
        // (the LDS image is dead after the barrier; the next commit comes after another one)
        __syncthreads();
        float* red = (float*)smem;               // [4 waves][4 q][8][2]
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float a = s_sum[j], b = s_sq[j];
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
          if (l15 == 0) { red[((wave * 4 + q) * 8 + j) * 2] = a; red[((wave * 4 + q) * 8 + j) * 2 + 1] = b; }
        }
        __syncthreads();
        if (tid < 64) {
          const int c = tid >> 1, which = tid & 1;          // c = 8 g + j
          float sum = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) sum += red[((w * 4 + (c >> 3)) * 8 + (c & 7)) * 2 + which];
          p.stats[((long long)tile * p.Cout + cob * 32 + c) * 2 + which] = sum;
        }
      }
    }
    __syncthreads();                 // every wave is done reading this step's LDS image
    if (s + 1 < nsteps && !(p.dbg & 1)) {
      commit(s + 1);
      __syncthreads();
    }
  }
}

template <typename T, int ND>
int launch_v2(const ConvV2Params& p, hipStream_t stream) {
  using TL = Tile2<ND>;
  constexpr int PZ = TL::TZ + 2 * TL::PADZ, PY = TL::TY + 2, PX = TL::TX + 2;
  constexpr int PLANE = ((PZ * PY * PX * 16 + 255) / 256) * 256;
  constexpr int LDS = 2 * PLANE + ((TL::TAPS / 3 + 1) / 2) * 3 * 2 * 1024;
  IUNET_SET_MAX_LDS((conv3_v2_kernel<T, ND>), LDS);
  const int ntiles = p.tilesZ * p.tilesY * p.tilesX * p.N;
  const int ncob = p.Cout / 32;
  // two workgroups per CU; with several cout blocks the CUs are split between them
  int gx = 512 / ncob;
  if (gx < 1) gx = 1;
  if (gx > ntiles) gx = ntiles;
  dim3 grid(gx, ncob);
  hipLaunchKernelGGL((conv3_v2_kernel<T, ND>), grid, dim3(256), LDS, stream, p);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

}  // namespace

int iunet_conv3_v2_launch(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride,
                          const void* wpk, const float* bias, float* stats, int N, int D, int H, int W, int Cin,
                          int Cout, int epi, hipStream_t stream) {
  ConvV2Params p;
  p.x = x; p.x_sstride = x_sstride; p.y = y; p.y_sstride = y_sstride; p.wpk = wpk; p.bias = bias; p.stats = stats;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.epi = epi;
  static const int dbg = getenv("IUNET_V2_DBG") ? atoi(getenv("IUNET_V2_DBG")) : 0;
  p.dbg = dbg;
  const int TZ = nd == 3 ? 4 : 1, TY = nd == 3 ? 8 : 16, TX = nd == 3 ? 16 : 32;
  p.tilesZ = (D + TZ - 1) / TZ; p.tilesY = (H + TY - 1) / TY; p.tilesX = (W + TX - 1) / TX;
  if (dtype == 0) return nd == 3 ? launch_v2<f16, 3>(p, stream) : launch_v2<f16, 2>(p, stream);
  return nd == 3 ? launch_v2<bf16, 3>(p, stream) : launch_v2<bf16, 2>(p, stream);
}
