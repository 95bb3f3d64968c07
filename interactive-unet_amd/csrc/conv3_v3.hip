// 3x3x3 convolution for Cout tiles of 32 channels, third structure: WEIGHT-STATIONARY.
//
// The second structure (conv3_v2.hip) re-fetches the 30 KB weight chunk with every 34.5 KB activation
// chunk: 127 B per output voxel and 16-channel chunk through the vector memory path, 16.8 B/clk per CU
// if the matrix pipe were to run flat out -- the measured 8-9 B/clk is what holds it at ~50 % MFMA.
// Here one persistent workgroup per CU (8 waves) keeps ALL weights of its Cout tile in LDS for the
// whole launch (Cin <= 64: 61 / 123 KB in the K16 fragment order of layout 1) and streams only
// activations:
//   Cin <= 48: tile 8 x 8 x 16 (halo 10 x 10 x 18 -> 1.76x re-fetch, 57.6 KB per chunk), 128 voxels per wave
//   Cin <= 64: tile 4 x 8 x 16 (34.5 KB per chunk, LDS is full at 157.7 KB),               64 voxels per wave
// i.e. 56 / 67 B per voxel-chunk instead of 127, and no weight traffic on the LDS store path.
// The step structure is that of conv3_v2: k-step = 2 filter columns x 16 channels, NR + 2 activation row
// fragments reused over the three dy taps, next chunk prefetched global -> registers during the MFMA phase
// and committed to LDS behind one barrier.  BatchNorm partial sums are written per 4 x 8 x 16 sub-tile in
// the statistics grid of the other structures (the 8 x 8 x 16 tile also in their summation order).
#include "common.h"
#include <cstdlib>

namespace {

struct ConvV3Params {
  const void* x;  long long x_sstride;
  void* y;        long long y_sstride;
  const void* wpk;                            // K16 order: [cob][chunk16][column pair][dy][2][64][8]
  const float* bias;
  float* stats;                               // [N * tiles(4x8x16)][Cout][2] or null
  int N, D, H, W, Cin, Cout;
  int tilesZ, tilesY, tilesX;                 // in units of this kernel's tile
  int stilesZ;                                // z tiles of the 4 x 8 x 16 statistics grid
  int epi;
  int dbg;                                    // profiling only (IUNET_V3_DBG): 1 no refill after step 0, 2 no MFMA phase, 4 no stores, 8 no LDS fragment reads
};

template <typename T, int TZ, int NI>
__global__ __launch_bounds__(512, 1) void conv3_v3_kernel(ConvV3Params p) {
  using V8 = typename Vec8<T>::type;
  constexpr int NW = 8, NT = NW * 64;
  constexpr int TY = 8, TX = 16, TAPS = 27;
  constexpr int PZ = TZ + 2, PY = TY + 2, PX = TX + 2;
  constexpr int NPIX = PZ * PY * PX;
  constexpr int PLANE = ((NPIX * 16 + 255) / 256) * 256;
  constexpr int CP = 2;                                // planes (of 8 channels) per chunk
  constexpr int NCOL = TAPS / 3, NCMB = (NCOL + 1) / 2, KS = NCMB * 3;
  constexpr int AIT = (NPIX + NT - 1) / NT;            // activation pixels per thread
  constexpr int WBYTES = KS * 2 * 1024;                // one 16-channel chunk of packed weights
  constexpr int OFF_W = CP * PLANE;
  static_assert(NW * NI == TZ * TY, "waves x rows must cover the tile");

  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int cob = blockIdx.y;
  const int tiles_per_sample = p.tilesZ * p.tilesY * p.tilesX;
  const int ntiles = tiles_per_sample * p.N;
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int t_begin = (int)((long long)lb * ntiles / gridDim.x);
  const int t_end = (int)((long long)(lb + 1) * ntiles / gridDim.x);
  const int nchunk = p.Cin >> 4;
  const int nsteps = (t_end - t_begin) * nchunk;
  if (nsteps <= 0) return;
  const long long plane_stride = (long long)p.D * p.H * p.W * 8;

  // ---- the weights of this Cout tile: global -> LDS, once ----
  {
    const u32x4* wsrc = (const u32x4*)p.wpk + (long long)cob * nchunk * (WBYTES / 16);
    const int nitems = nchunk * (WBYTES / 16);
    for (int i = tid; i < nitems; i += NT) *(u32x4*)(smem + OFF_W + i * 16) = wsrc[i];
  }

  int pcoord[AIT];
#pragma unroll
  for (int it = 0; it < AIT; ++it) {
    const int pix = min(tid + it * NT, NPIX - 1);
    const int px = pix % PX, t2 = pix / PX;
    pcoord[it] = px | ((t2 % PY) << 8) | ((t2 / PY) << 16);
  }
  int col_off[NCMB];
#pragma unroll
  for (int c = 0; c < NCMB; ++c) {
    const int col = min(2 * c + (q >> 1), NCOL - 1);           // the missing partner re-reads a valid column (zero weights)
    const int dz = col / 3, dx = col % 3;
    col_off[c] = (dz * PY * PX + dx) * 16;
  }
  const int row0 = wave * NI;                                  // first output row (z * TY + y) of this wave
  const int rbase = (q & 1) * PLANE + ((((row0 / TY) * PY + (row0 % TY)) * PX) + l15) * 16;
  float bias_r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bias_r[j] = (p.epi != 0) ? p.bias[cob * 32 + 8 * q + j] : 0.f;

  u32x4 areg[AIT][CP];
  unsigned okmask = 0;

  auto tile_origin = [&](int tile, int& n_img, int& z0, int& y0, int& x0) {
    n_img = tile / tiles_per_sample;
    int trem = tile - n_img * tiles_per_sample;
    const int tz_i = trem / (p.tilesY * p.tilesX);
    trem -= tz_i * p.tilesY * p.tilesX;
    const int ty_i = trem / p.tilesX;
    z0 = tz_i * TZ; y0 = ty_i * TY; x0 = (trem - ty_i * p.tilesX) * TX;
  };

  auto prefetch = [&](int s) {
    const int tile = t_begin + s / nchunk, chunk = s - (s / nchunk) * nchunk;
    int n_img, z0, y0, x0;
    tile_origin(tile, n_img, z0, y0, x0);
    const T* xc = (const T*)p.x + (long long)n_img * p.x_sstride + (long long)chunk * CP * plane_stride;
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      const int px = pcoord[it] & 255, py = (pcoord[it] >> 8) & 255, pz = pcoord[it] >> 16;
      const int gz = z0 + pz - 1, gy = y0 + py - 1, gx = x0 + px - 1;
      const bool ok = (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      const int cz = min(max(gz, 0), p.D - 1), cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
      const long long goff = (((long long)cz * p.H + cy) * p.W + cx) * 8;
      // raw loads only: nothing may consume them before commit(), or the wait lands in front of the MFMA phase
#pragma unroll
      for (int k = 0; k < CP; ++k) areg[it][k] = *(const u32x4*)(xc + k * plane_stride + goff);
      okmask = ok ? (okmask | (1u << it)) : (okmask & ~(1u << it));
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      const int pix = tid + it * NT;
      if (pix < NPIX) {
        const bool ok = (okmask >> it) & 1u;
#pragma unroll
        for (int k = 0; k < CP; ++k) *(u32x4*)(smem + k * PLANE + pix * 16) = ok ? areg[it][k] : u32x4{0u, 0u, 0u, 0u};
      }
    }
  };

  f32x4 acc[2][NI];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NI; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  prefetch(0);
  commit();
  __syncthreads();

  for (int s = 0; s < nsteps; ++s) {
    if (s + 1 < nsteps && !(p.dbg & 1)) prefetch(s + 1);      // in flight during the MFMA phase below
    const int tile = t_begin + s / nchunk, chunk = s - (s / nchunk) * nchunk;
    const unsigned char* wl = smem + OFF_W + chunk * WBYTES + lane * 16;

    if (!(p.dbg & 2))
#pragma unroll
    for (int c = 0; c < NCMB; ++c) {
      V8 R[NI + 2];
      if (p.dbg & 8) {
#pragma unroll
        for (int r = 0; r < NI + 2; ++r) R[r] = *(const V8*)&acc[0][r % NI];
      } else
#pragma unroll
      for (int r = 0; r < NI + 2; ++r) R[r] = *(const V8*)(smem + rbase + r * PX * 16 + col_off[c]);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const V8 a0 = *(const V8*)(wl + ((c * 3 + dy) * 2 + 0) * 1024);
        const V8 a1 = *(const V8*)(wl + ((c * 3 + dy) * 2 + 1) * 1024);
#pragma unroll
        for (int n = 0; n < NI; ++n) {
          acc[0][n] = mfma16<T>(a0, R[n + dy], acc[0][n]);
          acc[1][n] = mfma16<T>(a1, R[n + dy], acc[1][n]);
        }
      }
    }

    __syncthreads();                 // every wave is done reading this step's LDS image
    if (s + 1 < nsteps && !(p.dbg & 1)) {
      commit();
      __syncthreads();
    }
    // the epilogue comes AFTER the commit: its stores are then never in front of a vmcnt wait for the
    // prefetched loads (they drain during the next MFMA phase)
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
        const int row = row0 + n;
        const int gz = z0 + row / TY, gy = y0 + row % TY, gx = x0 + l15;
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
        // partial BatchNorm sums per 4 x 8 x 16 sub-tile (the statistics grid of the other structures): 16 x-lanes by
        // shuffles, then the waves of a sub-tile in wave order through LDS
        constexpr int WPS = 32 / NI;                           // waves per 4 x 8 x 16 sub-tile
        constexpr int NSUB = NW / WPS;                         // sub-tiles per tile (along z)
        float* red = (float*)(smem + OFF_W + nchunk * WBYTES);   // [8 waves][4 q][8][2], behind the weights
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float a = s_sum[j], b = s_sq[j];
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
          if (l15 == 0) { red[((wave * 4 + q) * 8 + j) * 2] = a; red[((wave * 4 + q) * 8 + j) * 2 + 1] = b; }
        }
        __syncthreads();
        if (tid < 64 * NSUB) {
          const int sub = tid >> 6, c = (tid & 63) >> 1, which = tid & 1;      // c = 8 g + j
          float sum = 0.f;
#pragma unroll
          for (int w = 0; w < WPS; ++w) sum += red[(((sub * WPS + w) * 4 + (c >> 3)) * 8 + (c & 7)) * 2 + which];
          const int sz = z0 / 4 + sub;                         // z index in the statistics grid
          if (sz < p.stilesZ) {
            const long long stile = (((long long)n_img * p.stilesZ + sz) * p.tilesY + y0 / TY) * p.tilesX + x0 / TX;
            p.stats[(stile * p.Cout + cob * 32 + c) * 2 + which] = sum;
          }
        }
      }
    }
  }
}

template <typename T, int TZ, int NI>
int launch_v3(ConvV3Params p, hipStream_t stream) {
  constexpr int PLANE = ((((TZ + 2) * 10 * 18) * 16 + 255) / 256) * 256;
  const int lds = 2 * PLANE + (p.Cin / 16) * 30720 + 2048;
  static int attr_lds = 0;
  if (lds > attr_lds) {
    IUNET_CHECK_HIP(hipFuncSetAttribute((const void*)conv3_v3_kernel<T, TZ, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_lds = lds;
  }
  p.tilesZ = (p.D + TZ - 1) / TZ; p.tilesY = (p.H + 7) / 8; p.tilesX = (p.W + 15) / 16;
  p.stilesZ = (p.D + 3) / 4;
  const int ntiles = p.tilesZ * p.tilesY * p.tilesX * p.N;
  const int ncob = p.Cout / 32;
  int gx = 256 / ncob;                          // one workgroup per CU
  if (gx < 1) gx = 1;
  if (gx > ntiles) gx = ntiles;
  hipLaunchKernelGGL((conv3_v3_kernel<T, TZ, NI>), dim3(gx, ncob), dim3(512), lds, stream, p);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

}  // namespace

// 1 if the weight-stationary structure can run this shape (3-D, all weights of a Cout tile fit in LDS)
int iunet_conv3_v3_ok(int nd, int Cin, int Cout) { return nd == 3 && Cin % 16 == 0 && Cin <= 64 && Cout % 32 == 0; }

int iunet_conv3_v3_launch(int dtype, const void* x, long long x_sstride, void* y, long long y_sstride, const void* wpk,
                          const float* bias, float* stats, int N, int D, int H, int W, int Cin, int Cout, int epi,
                          hipStream_t stream) {
  IUNET_REQUIRE(iunet_conv3_v3_ok(3, Cin, Cout), "conv3 layout 2 needs a 3-D conv with Cin <= 64 (got %d -> %d)", Cin, Cout);
  ConvV3Params p;
  p.x = x; p.x_sstride = x_sstride; p.y = y; p.y_sstride = y_sstride; p.wpk = wpk; p.bias = bias; p.stats = stats;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.epi = epi;
  p.tilesZ = p.tilesY = p.tilesX = p.stilesZ = 0;
  static const int dbg = getenv("IUNET_V3_DBG") ? atoi(getenv("IUNET_V3_DBG")) : 0;
  p.dbg = dbg;
  if (Cin <= 48) return dtype == 0 ? launch_v3<f16, 8, 8>(p, stream) : launch_v3<bf16, 8, 8>(p, stream);
  return dtype == 0 ? launch_v3<f16, 4, 4>(p, stream) : launch_v3<bf16, 4, 4>(p, stream);
}
