// 3^d convolution (d = 2, 3), stride 1, pad 1, as an im2col-free implicit GEMM on the
// gfx950 matrix cores.  Used for the forward stage convs and (with repacked weights)
// for their data gradients.
//
// GEMM view (swapped so that the epilogue owns 8 consecutive channels of one voxel):
//     D[cout][voxel] = sum_{tap, cin} W[cout][tap, cin] * X[tap-shifted voxel][cin]
//   A operand = packed weights  (rows = 16 couts,  k = 32 cin of one tap)   <- global/L1
//   B operand = activations     (cols = 16 consecutive x voxels, k = 32 cin) <- LDS
// One workgroup (4 waves, 256 threads) owns a TZ x TY x TX voxel tile and 16*MI output
// channels.  Per 32-channel chunk of Cin the input tile plus its 1-voxel halo is staged
// once into LDS as 4 planes of [pixel][8 ch] (16 B per pixel, the HBM layout itself), and
// every tap reads its B fragments from that single image at a shifted pixel offset: the
// 27 (9) taps re-use the LDS bytes, nothing is materialised.  A fragment read is 64 lanes
// x 16 B = four contiguous 256-B runs, one per plane, planes 0 mod 256 B apart:
// conflict-free for ds_read_b128 at any pixel shift.
//
// Each wave owns 8 voxel fragments (NI = 8) x MI cout tiles: 8*MI MFMA 16x16x32 per tap
// and chunk against 8 LDS fragment reads and MI weight-fragment loads.
#include "common.h"
#include <cstdlib>

namespace {

template <int ND> struct Tile;
template <> struct Tile<3> { static constexpr int TZ = 4, TY = 8, TX = 16, PADZ = 1, TAPS = 27; };
template <> struct Tile<2> { static constexpr int TZ = 1, TY = 16, TX = 32, PADZ = 0, TAPS = 9; };

struct Conv3Params {
  const void* x;  long long x_sstride;        // input view (Cin/8 planes)
  void* y;        long long y_sstride;        // output view (Cout/8 planes)
  const void* wpk;                            // packed weights [cob][chunk][tap][MI][64][8]
  const float* bias;                          // [Cout] or null
  float* stats;                               // [ntiles][Cout][2] partial sum / sumsq, or null
  int N, D, H, W, Cin, Cout;
  int tilesZ, tilesY, tilesX;                 // tiles per sample along each axis
  int epi;                                    // 0 none, 1 +bias, 2 +bias,relu
};

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_RELU = 2 };

// EXP: ablation switches for profiling builds only (bit 0 no weight loads, 1 no LDS fragment reads,
// 2 no staging, 3 no epilogue stores); production instantiations use EXP = 0.
template <typename T, int ND, int MI, int EXP = 0>
__global__ __launch_bounds__(256, 2) void conv3_mfma_kernel(Conv3Params p) {
  using TL = Tile<ND>;
  using V8 = typename Vec8<T>::type;
  constexpr int TZ = TL::TZ, TY = TL::TY, TX = TL::TX, PADZ = TL::PADZ;
  constexpr int PZ = TZ + 2 * PADZ, PY = TY + 2, PX = TX + 2;
  constexpr int NPIX = PZ * PY * PX;
  constexpr int PLANE = ((NPIX * 16 + 255) / 256) * 256;   // bytes, 0 mod 256
  constexpr int FX = TX / 16;                               // x fragments per row
  constexpr int NI = 8;
  static_assert(TZ * TY * FX == 32, "tile must hold 32 fragments (4 waves x 8)");
  constexpr int KD = (ND == 3) ? 3 : 1;

  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;

  // ---- which tile ----
  const int tiles_per_sample = p.tilesZ * p.tilesY * p.tilesX;
  const int ntiles = tiles_per_sample * p.N;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int n_img = tile / tiles_per_sample;
  int trem = tile - n_img * tiles_per_sample;
  const int tz_i = trem / (p.tilesY * p.tilesX);
  trem -= tz_i * p.tilesY * p.tilesX;
  const int ty_i = trem / p.tilesX;
  const int tx_i = trem - ty_i * p.tilesX;
  const int z0 = tz_i * TZ, y0 = ty_i * TY, x0 = tx_i * TX;
  const int cob = blockIdx.y;

  const long long plane_stride = (long long)p.D * p.H * p.W * 8;   // elements
  const T* xin = (const T*)p.x + (long long)n_img * p.x_sstride;
  const int nchunk = p.Cin >> 5;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int m = 0; m < MI; ++m)
#pragma unroll
    for (int n = 0; n < NI; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane LDS byte address of fragment n at tap (0,0,0)
  int frag_addr[NI];
#pragma unroll
  for (int n = 0; n < NI; ++n) {
    const int f = wave * NI + n;
    const int xh = f % FX, row = f / FX;
    const int fy = row % TY, fz = row / TY;
    frag_addr[n] = q * PLANE + (((fz * PY + fy) * PX) + xh * 16 + l15) * 16;
  }

  const V8* wbase = (const V8*)p.wpk + ((long long)cob * nchunk * TL::TAPS * MI) * 64 + lane;

  for (int chunk = 0; chunk < nchunk; ++chunk) {
    __syncthreads();   // previous chunk's fragment reads are done
    // ---- stage the halo tile of 32 channels: global -> registers -> LDS ----
    if (!(EXP & 4)) {
      const T* xc = xin + (long long)chunk * 4 * plane_stride;
      constexpr int ITERS = (NPIX + 255) / 256;
      // MI = 2 has the registers to keep all four planes of every pixel in flight at once (one
      // latency round); MI = 4 stages in two rounds.  Loads are unconditional (clamped address,
      // zero selected afterwards): no exec-mask branches between the load instructions.
      constexpr int ROUNDS = (MI == 2) ? 1 : 2;
      constexpr int PPR = 4 / ROUNDS;                    // planes per round
#pragma unroll
      for (int rnd = 0; rnd < ROUNDS; ++rnd) {
        u32x4 v[ITERS][PPR];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          const int pix = tid + it * 256;
          const int px = pix % PX, t2 = pix / PX;
          const int py = t2 % PY, pz = (t2 / PY) % PZ;
          const int gz = z0 + pz - PADZ, gy = y0 + py - 1, gx = x0 + px - 1;
          const bool ok = (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
          const int cz = min(max(gz, 0), p.D - 1), cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
          const long long goff = (((long long)cz * p.H + cy) * p.W + cx) * 8;
#pragma unroll
          for (int k = 0; k < PPR; ++k) {
            const u32x4 val = *(const u32x4*)(xc + (rnd * PPR + k) * plane_stride + goff);
            v[it][k] = ok ? val : u32x4{0u, 0u, 0u, 0u};
          }
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          const int pix = tid + it * 256;
          if (pix < NPIX) {
#pragma unroll
            for (int k = 0; k < PPR; ++k)
              *(u32x4*)(smem + (rnd * PPR + k) * PLANE + pix * 16) = v[it][k];
          }
        }
      }
    }
    __syncthreads();

    // ---- all taps of this chunk ----
    const V8* wc = wbase + (long long)chunk * TL::TAPS * MI * 64;
    // weight fragments come straight from global/L2 into a register ring, PF taps ahead:
    // one tap is only 8*MI MFMAs (128 MFMA-cycles at MI = 2), far less than an L2 round trip
    constexpr int PF = (MI == 2) ? 4 : 1;
    V8 ring[PF + 1][MI];
#pragma unroll
    for (int t = 0; t < PF; ++t)
#pragma unroll
      for (int m = 0; m < MI; ++m) ring[t][m] = wc[(t * MI + m) * 64];
#pragma unroll
    for (int tap = 0; tap < TL::TAPS; ++tap) {
      if (tap + PF < TL::TAPS && !(EXP & 1)) {
#pragma unroll
        for (int m = 0; m < MI; ++m) ring[(tap + PF) % (PF + 1)][m] = wc[((tap + PF) * MI + m) * 64];
      }
      const int dz = (ND == 3) ? tap / 9 : 0;
      const int dy = (tap / 3) % 3, dx = tap % 3;
      const int tapoff = ((dz * PY + dy) * PX + dx) * 16;
      (void)KD;
#pragma unroll
      for (int n = 0; n < NI; ++n) {
        V8 b;
        if (EXP & 2) { b = ring[0][0]; asm volatile("" : "+v"(b)); }
        else b = *(const V8*)(smem + frag_addr[n] + tapoff);
#pragma unroll
        for (int m = 0; m < MI; ++m) acc[m][n] = mfma16<T>(ring[(EXP & 1) ? (tap % PF) : (tap % (PF + 1))][m], b, acc[m][n]);
      }
    }
  }

  // ---- epilogue: lane (g = q, x = l15) holds couts 32u+8g+{0..7} of MI/2 plane groups ----
  T* yout = (T*)p.y + (long long)n_img * p.y_sstride;
  float s_sum[MI / 2][8], s_sq[MI / 2][8];
#pragma unroll
  for (int u = 0; u < MI / 2; ++u)
#pragma unroll
    for (int j = 0; j < 8; ++j) { s_sum[u][j] = 0.f; s_sq[u][j] = 0.f; }

#pragma unroll
  for (int n = 0; n < NI; ++n) {
    const int f = wave * NI + n;
    const int xh = f % FX, row = f / FX;
    const int fy = row % TY, fz = row / TY;
    const int gz = z0 + fz, gy = y0 + fy, gx = x0 + xh * 16 + l15;
    const bool ok = gz < p.D && gy < p.H && gx < p.W;
    const long long voff = (((long long)gz * p.H + gy) * p.W + gx) * 8;
#pragma unroll
    for (int u = 0; u < MI / 2; ++u) {
      const int cbase = cob * 16 * MI + 32 * u + 8 * q;
      float vals[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { vals[j] = acc[2 * u][n][j]; vals[4 + j] = acc[2 * u + 1][n][j]; }
      if (p.stats != nullptr && ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { s_sum[u][j] += vals[j]; s_sq[u][j] += vals[j] * vals[j]; }
      }
      if (p.epi != EPI_NONE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          vals[j] += p.bias[cbase + j];
          if (p.epi == EPI_BIAS_RELU) vals[j] = fmaxf(vals[j], 0.f);
        }
      }
      V8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = from_f32<T>(vals[j]);
      if (ok && !((EXP & 8) && vals[0] != 12345.f)) {
        const int plane = (cbase >> 3);
        *(V8*)(yout + (long long)plane * plane_stride + voff) = o;
      }
    }
  }

  if (p.stats != nullptr) {
    // reduce over the 16 x-lanes of each q group, then over the 4 waves through LDS
    __syncthreads();
    float* red = (float*)smem;   // [4 waves][MI/2][4 q][8][2]
#pragma unroll
    for (int u = 0; u < MI / 2; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = s_sum[u][j], b = s_sq[u][j];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (l15 == 0) {
          const int idx = (((wave * (MI / 2) + u) * 4 + q) * 8 + j) * 2;
          red[idx] = a; red[idx + 1] = b;
        }
      }
    __syncthreads();
    constexpr int NCH = 16 * MI;
    if (tid < NCH * 2) {
      const int c = tid >> 1, which = tid & 1;          // c = 32u + 8g + j
      const int u = c >> 5, g = (c >> 3) & 3, j = c & 7;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += red[(((w * (MI / 2) + u) * 4 + g) * 8 + j) * 2 + which];
      p.stats[((long long)tile * p.Cout + cob * NCH + c) * 2 + which] = s;
    }
  }
}

// ---- weight packing: fp32 [Cout][Cin][taps] -> fragment order, optional per-cout scale ----
// mode 0: forward weights.  mode 1: data-gradient weights: out channel = original Cin,
// in channel = original Cout, taps flipped (correlation with the transposed, mirrored filter).
template <typename T>
__global__ void pack_conv3_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ dst,
                                  int CoutP, int CinP, int taps, int MI, int mode, int CoutO, int CinO) {
  // CoutP / CinP: channels of the packed (possibly transposed) operator; CoutO / CinO: original dims
  const long long total = (long long)CoutP * CinP * taps;
  const int nchunk = CinP >> 5;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int j = r & 7; r >>= 3;
    const int lane = r & 63; r >>= 6;
    const int m = r % MI; r /= MI;
    const int tap = r % taps; r /= taps;
    const int chunk = r % nchunk;
    const int cob = r / nchunk;
    const int row = lane & 15, qq = lane >> 4;
    const int co = cob * 16 * MI + 32 * (m >> 1) + 8 * (row >> 2) + 4 * (m & 1) + (row & 3);
    const int ci = chunk * 32 + 8 * qq + j;
    float v;
    if (mode == 0) {
      v = w[((long long)co * CinO + ci) * taps + tap];
      if (scale) v *= scale[co];
    } else {
      v = w[((long long)ci * CinO + co) * taps + (taps - 1 - tap)];
    }
    dst[i] = from_f32<T>(v);
  }
}

// K16 packing for the Cout-32 structure (conv3_v2.hip): [cob32][chunk16][column pair][dy][2][64][8].
// A filter "column" is a (dz, dx) pair (9 in 3-D, 3 in 2-D); a k-step holds two columns x 16 channels:
// lane (row = l & 15, q = l >> 4), element j -> column 2 pair + (q >> 1), cin = 16 chunk + 8 (q & 1) + j.
template <typename T>
__global__ void pack_conv3_k16_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ dst,
                                      int CoutP, int CinP, int taps, int dgrad, int CinO) {
  const int ncol = taps / 3, ncmb = (ncol + 1) / 2, nchunk = CinP >> 4;
  const long long total = (long long)(CoutP / 32) * nchunk * ncmb * 3 * 2 * 64 * 8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int j = r & 7; r >>= 3;
    const int lane = r & 63; r >>= 6;
    const int m = r & 1; r >>= 1;
    const int dy = r % 3; r /= 3;
    const int c = r % ncmb; r /= ncmb;
    const int chunk = r % nchunk;
    const int cob = r / nchunk;
    const int row = lane & 15, qq = lane >> 4;
    const int co = cob * 32 + 8 * (row >> 2) + 4 * m + (row & 3);
    const int ci = chunk * 16 + 8 * (qq & 1) + j;
    const int col = 2 * c + (qq >> 1);
    float v = 0.f;
    if (col < ncol) {
      const int tap = ((col / 3) * 3 + dy) * 3 + (col % 3);       // (dz, dy, dx) -> linear tap
      if (!dgrad) { v = w[((long long)co * CinO + ci) * taps + tap]; if (scale) v *= scale[co]; }
      else v = w[((long long)ci * CinO + co) * taps + (taps - 1 - tap)];
    }
    dst[i] = from_f32<T>(v);
  }
}

// Compact K16 order (mode bit 2; conv3_v4.hip's padding-free step): the last filter column -- (dz, dx) column 8 of the 3^3 filter, dx
// column 2 of the 3^2 filter -- of two consecutive 16-channel chunks shares one k-step instead of being padded to a pair with zeros.
// Per Cout tile and chunk PAIR (32 channels), NR = 4 (3-D) / 1 (2-D) regular column pairs:
//   [even chunk: column pairs 0..NR-1][dy][2][64][8]  (24 / 6 KB)  |  [odd chunk: the same]  |  [cross: dy][2][64][8]  (6 KB),
// cross lanes q >> 1 = 0: the last column of the even chunk, q >> 1 = 1: of the odd chunk.  Cout x Cin x taps elements: no padding.
// 2-D: the pair block IS one 32-channel step of conv3_v4.hip -- three k-groups instead of four (-25 % MFMAs and fragment reads).
template <typename T>
__global__ void pack_conv3_k16c_kernel(const float* __restrict__ w, const float* __restrict__ scale, T* __restrict__ dst,
                                       int CoutP, int CinP, int taps, int dgrad, int CinO) {
  constexpr int FR = 512;                                           // elements of one fragment (64 lanes x 8)
  const int ncol = taps / 3, nreg = ncol / 2;                       // 9 columns: 4 regular pairs; 3 columns: 1
  const int EVEN = nreg * 3 * 2 * FR, PAIR = 2 * EVEN + 3 * 2 * FR, NF = nreg * 6;      // elements; fragments of a chunk's regular part
  const int npair = CinP >> 5;
  const long long total = (long long)(CoutP / 32) * npair * PAIR;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(i % PAIR);
    long long r = i / PAIR;
    const int pr = (int)(r % npair), cob = (int)(r / npair);
    const int j = e & 7, lane = (e >> 3) & 63;
    const int row = lane & 15, qq = lane >> 4;
    int frag = e >> 9, chunk, col;                                  // fragment index within the pair block
    if (frag < 2 * NF) { chunk = 2 * pr + frag / NF; frag %= NF; col = 2 * (frag / 6) + (qq >> 1); frag %= 6; }
    else { frag -= 2 * NF; chunk = 2 * pr + (qq >> 1); col = ncol - 1; }
    const int dy = frag >> 1, m = frag & 1;
    const int co = cob * 32 + 8 * (row >> 2) + 4 * m + (row & 3);
    const int ci = chunk * 16 + 8 * (qq & 1) + j;
    const int tap = ((col / 3) * 3 + dy) * 3 + (col % 3);
    float v;
    if (!dgrad) { v = w[((long long)co * CinO + ci) * taps + tap]; if (scale) v *= scale[co]; }
    else v = w[((long long)ci * CinO + co) * taps + (taps - 1 - tap)];
    dst[i] = from_f32<T>(v);
  }
}

template <typename T, int ND, int MI>
int launch_conv3(const Conv3Params& p, hipStream_t stream) {
  using TL = Tile<ND>;
  constexpr int PZ = TL::TZ + 2 * TL::PADZ, PY = TL::TY + 2, PX = TL::TX + 2;
  constexpr int PLANE = ((PZ * PY * PX * 16 + 255) / 256) * 256;
  constexpr int LDS = 4 * PLANE;
  IUNET_SET_MAX_LDS((conv3_mfma_kernel<T, ND, MI>), LDS);
  dim3 grid(p.tilesZ * p.tilesY * p.tilesX * p.N, p.Cout / (16 * MI));
  hipLaunchKernelGGL((conv3_mfma_kernel<T, ND, MI>), grid, dim3(256), LDS, stream, p);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

}  // namespace

int iunet_conv3_v2_launch(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride,
                          const void* wpk, const float* bias, float* stats, int N, int D, int H, int W, int Cin,
                          int Cout, int epi, hipStream_t stream);
int iunet_conv3_v4_launch(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride, const void* wpk,
                          const float* bias, float* stats, int N, int D, int H, int W, int Cin, int Cout, int epi,
                          const float* in_scale, const float* in_shift, hipStream_t stream, const void* bw_y = nullptr,
                          long long bw_y_ss = 0, const float* const* bw_par = nullptr, int compact = 0, int per_sample = 0,
                          int* query_rows = nullptr);

// Host entry used by the net runtime and the per-kernel C ABI.
int iunet_conv3_launch(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride,
                       const void* wpk, const float* bias, float* stats, int N, int D, int H, int W, int Cin,
                       int Cout, int epi, int layout, hipStream_t stream, const float* in_scale, const float* in_shift,
                       const void* bw_y, long long bw_y_ss, const float* const* bw_par) {
  IUNET_REQUIRE(nd == 2 || nd == 3, "conv3: nd must be 2 or 3 (got %d)", nd);
  IUNET_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, "conv3: Cin (%d) and Cout (%d) must be multiples of 32", Cin, Cout);
  IUNET_REQUIRE(nd == 3 || D == 1, "conv3: 2-D conv needs D == 1");
  IUNET_REQUIRE(epi == 0 || bias != nullptr, "conv3: epilogue %d needs a bias", epi);
  Conv3Params p;
  p.x = x; p.x_sstride = x_sstride; p.y = y; p.y_sstride = y_sstride; p.wpk = wpk; p.bias = bias; p.stats = stats;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.epi = epi;
  const int TZ = nd == 3 ? 4 : 1, TY = nd == 3 ? 8 : 16, TX = nd == 3 ? 16 : 32;
  p.tilesZ = (D + TZ - 1) / TZ; p.tilesY = (H + TY - 1) / TY; p.tilesX = (W + TX - 1) / TX;
  const bool wide = (Cout % 64 == 0);
  IUNET_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv3: in_scale and in_shift come together");
  IUNET_REQUIRE(in_scale == nullptr || layout >= 2, "conv3: a fused input activation needs layout 2 or 3 (got %d)", layout);
  IUNET_REQUIRE(bw_y == nullptr || layout == 2 || (layout == 3 && nd == 2 && Cin <= 64), "conv3: the fused BatchNorm-backward sums need layout 2, or layout 3 in 2-D up to 64 input channels (got %d)", layout);
  if (layout >= 2) {
    return iunet_conv3_v4_launch(dtype, nd, x, x_sstride, y, y_sstride, wpk, bias, stats, N, D, H, W, Cin, Cout, epi,
                                 in_scale, in_shift, stream, bw_y, bw_y_ss, bw_par, layout == 3);
  }
  if (layout == 1)
    return iunet_conv3_v2_launch(dtype, nd, x, x_sstride, y, y_sstride, wpk, bias, stats, N, D, H, W, Cin, Cout, epi, stream);
#define IUNET_DISPATCH(TT)                                                                   \
  if (nd == 3) return wide ? launch_conv3<TT, 3, 4>(p, stream) : launch_conv3<TT, 3, 2>(p, stream); \
  else         return wide ? launch_conv3<TT, 2, 4>(p, stream) : launch_conv3<TT, 2, 2>(p, stream);
  if (dtype == 0) { IUNET_DISPATCH(f16) } else { IUNET_DISPATCH(bf16) }
#undef IUNET_DISPATCH
}

int iunet_conv3_tiles(int nd, int N, int D, int H, int W) {
  const int TZ = nd == 3 ? 4 : 1, TY = nd == 3 ? 8 : 16, TX = nd == 3 ? 16 : 32;
  return N * ((D + TZ - 1) / TZ) * ((H + TY - 1) / TY) * ((W + TX - 1) / TX);
}

int iunet_conv3_mi(int Cout) { return (Cout % 64 == 0) ? 4 : 2; }

// Weight layout / kernel structure of a launch: 0 = first structure (conv3_mfma_kernel, 32-channel chunks),
// 1 = LDS-fed persistent Cout-32 structure (conv3_v2.hip, K16 fragment order), 2 = wave-specialised structure on the
// same K16 operator (conv3_v4.hip; 3-D).  The second one is used for
// Cout tiles of 32 and whenever the first structure's grid would under-fill the chip (deep levels):
// measured at 128^3 / N = 1, levels 2 and 3 run 1.6-2x faster on it, level 1 (Cout 64, 512 tiles) 10 % slower.
// IUNET_CONV_V1=1 / IUNET_CONV_V2_ALL=1 force one structure (A/B runs).
int iunet_conv3_pick(int nd, int N, int D, int H, int W, int Cin, int Cout) {
  static const bool force_v1 = getenv("IUNET_CONV_V1") != nullptr;
  static const bool force_v2 = getenv("IUNET_CONV_V2_ALL") != nullptr;
  static const int v3 = getenv("IUNET_CONV_V3") ? atoi(getenv("IUNET_CONV_V3")) : 1;   // 0: layouts 0 / 1 only (A/B runs)
  if (force_v1) return 0;
  const long long tiles = iunet_conv3_tiles(nd, N, D, H, W);
  // 3-D: the wave-specialised structure wins at every level of the U-Net (profiles/r01_conv_levels.md), marginally
  // behind layout 0 only for 128 -> 64 at 64^3 (-4 %)
  // 2-D: the same structure wins where the weights stay resident in LDS (Cin <= 64: +5-12 % on the HBM-bound level-0
  // layers); for wider inputs layout 0 has no K16 padding (a quarter of the 2-D k-steps) and stays ahead
  if (v3 && Cin >= 32 && (nd == 3 || Cin <= 64)) return 2;
  if (force_v2 || Cout % 64 != 0) return 1;
  const long long blocks = tiles * (Cout / 64);
  return blocks < 512 ? 1 : 0;
}

// elements of the packed operator (the K16 order pads the tap count to an even number)
long long iunet_pack_conv3_size(int Cout, int Cin, int taps, int mode) {
  if (mode & 4) return (long long)Cout * Cin * taps;                   // compact K16: no padding
  const int t = (mode & 2) ? ((taps / 3 + 1) / 2) * 6 : taps;      // K16 order pads the filter columns to pairs
  return (long long)Cout * Cin * t;
}

int iunet_pack_conv3_launch(int dtype, const float* w, const float* scale, void* dst, int Cout, int Cin, int taps,
                            int mode, hipStream_t stream) {
  // mode bit 0: data-gradient operator (Cin x Cout, taps mirrored); bit 1: K16 fragment order (layout 1)
  const int dg = mode & 1;
  const int CoutP = dg == 0 ? Cout : Cin, CinP = dg == 0 ? Cin : Cout;
  IUNET_REQUIRE(CoutP % 32 == 0 && CinP % 32 == 0, "pack_conv3: channel counts must be multiples of 32 (%d, %d)", CoutP, CinP);
  const int MI = iunet_conv3_mi(CoutP);
  if (mode & 4) {     // compact K16 (conv3_v4.hip: layout 3 in 3-D, the cross-pair step of the 2-D split-precision conv)
    const long long tot = (long long)CoutP * CinP * taps;
    const int nb = (int)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
    if (dtype == 0) hipLaunchKernelGGL(pack_conv3_k16c_kernel<f16>, dim3(nb), dim3(256), 0, stream, w, scale, (f16*)dst, CoutP, CinP, taps, dg, Cin);
    else hipLaunchKernelGGL(pack_conv3_k16c_kernel<bf16>, dim3(nb), dim3(256), 0, stream, w, scale, (bf16*)dst, CoutP, CinP, taps, dg, Cin);
    IUNET_CHECK_HIP(hipGetLastError());
    return IUNET_OK;
  }
  if (mode & 2) {     // the LDS-fed structure's K16 fragment order
    const long long tot = (long long)(CoutP / 32) * (CinP / 16) * ((taps / 3 + 1) / 2) * 3 * 1024;
    const int nb = (int)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
    if (dtype == 0) hipLaunchKernelGGL(pack_conv3_k16_kernel<f16>, dim3(nb), dim3(256), 0, stream, w, scale, (f16*)dst, CoutP, CinP, taps, dg, Cin);
    else hipLaunchKernelGGL(pack_conv3_k16_kernel<bf16>, dim3(nb), dim3(256), 0, stream, w, scale, (bf16*)dst, CoutP, CinP, taps, dg, Cin);
    IUNET_CHECK_HIP(hipGetLastError());
    return IUNET_OK;
  }
  const long long total = (long long)CoutP * CinP * taps;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == 0)
    hipLaunchKernelGGL(pack_conv3_kernel<f16>, dim3(blocks), dim3(256), 0, stream, w, scale, (f16*)dst, CoutP, CinP, taps, MI, dg, Cout, Cin);
  else
    hipLaunchKernelGGL(pack_conv3_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, w, scale, (bf16*)dst, CoutP, CinP, taps, MI, dg, Cout, Cin);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}
