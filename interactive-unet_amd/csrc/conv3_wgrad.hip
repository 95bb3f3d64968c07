// Weight gradient of the 3^d convolution on the matrix cores:
//     dW[co][ci][tap] = sum_{n, v} dy[n][co][v] * x[n][ci][v + tap - 1]
// GEMM per tap with the VOXELS as the reduction (k) dimension: A[co][k = voxel] = dy^T,
// B[k = voxel][ci] = x shifted by the tap.  Both operands need "8 consecutive voxels of one
// channel" per lane while the HBM/LDS layout (NHWC8c) has the channels innermost, so the
// fragments are read with gfx950's transposing LDS read ds_read_b64_tr_b16 straight from
// the channel-innermost tile images (no transposed copy of the activations anywhere).
//
// One workgroup (4 waves) walks voxel tiles for one (32 co) x (32 ci) block of the filter
// and keeps all TAPS x 32 x 32 partial sums in registers (split over the waves by
// (tap, ci half)); at the end it stores one fp32 slab; a second kernel sums the slabs in a
// fixed order into dW (deterministic, no float atomics).
//
// LDS plane strides are 64 mod 256 bytes: a 32-lane half of the tr-read touches planes
// p, p+1 and pixels P..P+3, P+8..P+11 -> 32 distinct 8-byte slots of the 256-B bank row.
#include "common.h"
#include <cstdlib>

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s4_ptr;

template <int ND> struct WTile;
template <> struct WTile<3> { static constexpr int TZ = 2, TY = 8, TX = 16, PADZ = 1, TAPS = 27; };
template <> struct WTile<2> { static constexpr int TZ = 1, TY = 16, TX = 32, PADZ = 0, TAPS = 9; };

struct WgradParams {
  const void* x;  long long x_ss;     // conv input  (Cin/8 planes)
  const void* dy; long long dy_ss;    // output grad (Cout/8 planes)
  float* slab;                        // [gridDim.x][Cout/32][Cin/32][TAPS][32][32]
  const float* x_scale;               // optional [Cin] pair: the conv input is relu(x_scale * x + x_shift) (applied while staging)
  const float* x_shift;
  int N, D, H, W, Cin, Cout;
  int tilesZ, tilesY, tilesX;
};

template <typename T>
__device__ __forceinline__ typename Vec8<T>::type tr_frag(unsigned addr) {
  // two transposing reads: k = 0..3 and k = 4..7 (4 voxels = 64 bytes further) of this lane's k-quad
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)addr);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(addr + 64));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(typename Vec8<T>::type, v);
}

template <typename T, int ND>
__global__ __launch_bounds__(256, 2) void conv3_wgrad_kernel(WgradParams p) {
  using TL = WTile<ND>;
  constexpr int TZ = TL::TZ, TY = TL::TY, TX = TL::TX, PADZ = TL::PADZ, TAPS = TL::TAPS;
  constexpr int PZ = TZ + 2 * PADZ, PY = TY + 2, PX = TX + 2;
  constexpr int NPIX = PZ * PY * PX;
  constexpr int NVOX = TZ * TY * TX;
  constexpr int FX = TX / 16;
  constexpr int NFRAG = NVOX / 16;
  constexpr int NKS = NFRAG / 2;                                   // k-steps of 32 voxels
  constexpr int PLANE_X = ((NPIX * 16 + 255) / 256) * 256 + 64;    // 64 mod 256
  constexpr int PLANE_Y = ((NVOX * 16 + 255) / 256) * 256 + 64;
  constexpr int OFF_Y = 4 * PLANE_X;
  constexpr int NU = TAPS * 2;                                     // units = (tap, ci half)
  constexpr int MAXU = (NU + 3) / 4;

  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform -> scalar registers
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  const int gh = g >> 1, gl = g & 1;

  // lane part of the tr-read addresses (bytes): voxel (gl*8 + q) of fragment gh, channels 4pp..4pp+3
  const unsigned laneY = lds0 + OFF_Y + (pp >> 1) * PLANE_Y + (pp & 1) * 8 + (gh * 16 + gl * 8 + q) * 16;
  // x image: fragment gh is the next y row (3-D, FX = 1) or the next 16 x voxels (2-D, FX = 2)
  const int ghpix = (FX == 1) ? gh * PX : gh * 16;
  const unsigned laneX = lds0 + (pp >> 1) * PLANE_X + (pp & 1) * 8 + (ghpix + gl * 8 + q) * 16;

  // units of this wave: u = wave + 4 i  ->  tap = u >> 1, ci half = u & 1
  unsigned unit_off[MAXU];
#pragma unroll
  for (int i = 0; i < MAXU; ++i) {
    const int u = wave + 4 * i;
    const int tap = (u < NU ? u : 0) >> 1, cih = u & 1;
    const int dz = ND == 3 ? tap / 9 : 0, dy_ = (tap / 3) % 3, dx = tap % 3;
    unit_off[i] = (unsigned)(((dz * PY + dy_) * PX + dx) * 16 + cih * 2 * PLANE_X);
  }

  f32x4 acc[MAXU][2];
#pragma unroll
  for (int i = 0; i < MAXU; ++i) { acc[i][0] = f32x4{0, 0, 0, 0}; acc[i][1] = f32x4{0, 0, 0, 0}; }

  const int tiles_per_sample = p.tilesZ * p.tilesY * p.tilesX;
  const int ntiles = tiles_per_sample * p.N;
  const int cob = blockIdx.y, cib = blockIdx.z;
  const long long plane_stride = (long long)p.D * p.H * p.W * 8;

  // contiguous run of tiles per workgroup, runs of one XCD adjacent (halo lines stay in that XCD's L2)
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int t_begin = (int)((long long)lb * ntiles / gridDim.x), t_end = (int)((long long)(lb + 1) * ntiles / gridDim.x);
  // split staging: the loads of tile t + 1 are issued into registers before the MFMA phase of tile t and written to LDS
  // after it (their latency hides behind the compute)
  constexpr int XITERS = (NPIX + 255) / 256, YITERS = NVOX / 256;
  static_assert(NVOX % 256 == 0, "tile voxels must be a multiple of 256");
  u32x4 xr[XITERS][4], yr[YITERS][4];
  unsigned okx = 0;                   // bit it: halo pixel slot `it` of the staged tile lies inside the image
  auto load_tile = [&](int tile) {
    okx = 0;
    const int n_img = tile / tiles_per_sample;
    int trem = tile - n_img * tiles_per_sample;
    const int tz_i = trem / (p.tilesY * p.tilesX);
    trem -= tz_i * p.tilesY * p.tilesX;
    const int ty_i = trem / p.tilesX, tx_i = trem - ty_i * p.tilesX;
    const int z0 = tz_i * TZ, y0 = ty_i * TY, x0 = tx_i * TX;
    const T* xin = (const T*)p.x + (long long)n_img * p.x_ss + (long long)cib * 4 * plane_stride;
    const T* dyin = (const T*)p.dy + (long long)n_img * p.dy_ss + (long long)cob * 4 * plane_stride;
#pragma unroll
    for (int it = 0; it < XITERS; ++it) {
      const int pix = tid + it * 256;
      const int px = pix % PX, t2 = pix / PX;
      const int py = t2 % PY, pz = t2 / PY;
      const int gz = z0 + pz - PADZ, gy = y0 + py - 1, gx = x0 + px - 1;
      const bool ok = (pix < NPIX) && (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H &&
                      (unsigned)gx < (unsigned)p.W;
      const long long goff = (((long long)gz * p.H + gy) * p.W + gx) * 8;
      okx |= ok ? (1u << it) : 0u;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        u32x4 val = u32x4{0u, 0u, 0u, 0u};
        if (ok) val = *(const u32x4*)(xin + k * plane_stride + goff);
        xr[it][k] = val;
      }
    }
#pragma unroll
    for (int it = 0; it < YITERS; ++it) {
      const int pix = tid + it * 256;
      const int px = pix % TX, t2 = pix / TX;
      const int py = t2 % TY, pz = t2 / TY;
      const int gz = z0 + pz, gy = y0 + py, gx = x0 + px;
      const bool ok = gz < p.D && gy < p.H && gx < p.W;
      const long long goff = (((long long)gz * p.H + gy) * p.W + gx) * 8;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        u32x4 val = u32x4{0u, 0u, 0u, 0u};
        if (ok) val = *(const u32x4*)(dyin + k * plane_stride + goff);
        yr[it][k] = val;
      }
    }
  };
  auto commit_tile = [&]() {
#pragma unroll
    for (int it = 0; it < XITERS; ++it) {
      const int pix = tid + it * 256;
      if (pix < NPIX) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          u32x4 v = xr[it][k];
          if (p.x_scale != nullptr && ((okx >> it) & 1u)) {       // z = relu(scale * y + shift) as bn_relu_fwd_kernel; padding stays 0
            const typename Vec8<T>::type in = __builtin_bit_cast(typename Vec8<T>::type, v);
            typename Vec8<T>::type o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int c = cib * 32 + k * 8 + j;
              o[j] = from_f32<T>(fmaxf(fmaf(p.x_scale[c], to_f32<T>(in[j]), p.x_shift[c]), 0.f));
            }
            v = __builtin_bit_cast(u32x4, o);
          }
          *(u32x4*)(smem + k * PLANE_X + pix * 16) = v;
        }
      }
    }
#pragma unroll
    for (int it = 0; it < YITERS; ++it) {
      const int pix = tid + it * 256;
#pragma unroll
      for (int k = 0; k < 4; ++k) *(u32x4*)(smem + OFF_Y + k * PLANE_Y + pix * 16) = yr[it][k];
    }
  };
  if (t_begin < t_end) load_tile(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();    // previous tile's reads are done
    commit_tile();
    __syncthreads();
    if (tile + 1 < t_end) load_tile(tile + 1);

    // ---- k loop over 32-voxel steps (fragments 2ks, 2ks+1) ----
#pragma unroll 2
    for (int ks = 0; ks < NKS; ++ks) {
      // fragment 2ks -> (fz, fy, xh); the lane's own fragment adds gh (folded into laneX / laneY)
      const int f = 2 * ks;
      const int xh = f % FX, row = f / FX;
      const int fy = row % TY, fz = row / TY;
      const unsigned offY = (unsigned)(f * 16 * 16);
      const unsigned offX = (unsigned)((((fz * PY + fy) * PX) + xh * 16) * 16);
      const typename Vec8<T>::type a0 = tr_frag<T>(laneY + offY);                    // co 0..15
      const typename Vec8<T>::type a1 = tr_frag<T>(laneY + offY + 2 * PLANE_Y);      // co 16..31
      // branch-free: a wave whose last unit does not exist (u >= NU) recomputes tap 0 into an
      // accumulator that is never stored (<= 1/14 of its MFMAs) -- keeps the read/MFMA stream pipelined
#pragma unroll
      for (int i = 0; i < MAXU; ++i) {
        const typename Vec8<T>::type b = tr_frag<T>(laneX + offX + unit_off[i]);
        acc[i][0] = mfma16<T>(a0, b, acc[i][0]);
        acc[i][1] = mfma16<T>(a1, b, acc[i][1]);
      }
    }
  }

  // ---- store the slab: D rows = co (4g + j), cols = ci (lane & 15) ----
  const int ncob = gridDim.y, ncib = gridDim.z;
  float* slab = p.slab + ((((long long)blockIdx.x * ncob + cob) * ncib + cib) * TAPS) * 1024;
#pragma unroll
  for (int i = 0; i < MAXU; ++i) {
    const int u = wave + 4 * i;
    if (u < NU) {
      const int tap = u >> 1, cih = u & 1;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          slab[tap * 1024 + (t * 16 + 4 * g + j) * 32 + cih * 16 + i16] = acc[i][t][j];
    }
  }
}

// dW[co][ci][tap] = alpha * sum_b slab[b][cob][cib][tap][co%32][ci%32]; threads walk the slab
// order (coalesced reads of every part), the small dW write is scattered.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int nb, int Cout, int Cin, int taps,
                                                           float* __restrict__ dW, float alpha) {
  // 64 slab columns per block as 16 float4 lanes x 16 row groups (per_b is a multiple of 1024): 16-byte loads, nb / 16 of them
  // per thread; fixed summation order
  __shared__ f32x4 red[16][16];
  const long long per_b4 = (long long)Cout * Cin * taps / 4;
  const int col4 = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const long long j4 = (long long)blockIdx.x * 16 + col4;
  const f32x4* slab4 = (const f32x4*)slab;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int b = grp; b < nb; b += 16) s += slab4[b * per_b4 + j4];
  red[grp][col4] = s;
  __syncthreads();
  if (grp != 0) return;
#pragma unroll
  for (int g = 1; g < 16; ++g) s += red[g][col4];
  const int ncib = Cin / 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long j = j4 * 4 + k;
    const int c = (int)(j & 31), r = (int)((j >> 5) & 31);
    long long t = j >> 10;
    const int tap = (int)(t % taps); t /= taps;
    const int cib = (int)(t % ncib), cob = (int)(t / ncib);
    dW[((long long)(cob * 32 + r) * Cin + cib * 32 + c) * taps + tap] = alpha * s[k];
  }
}

// Wide layers (many 32 x 32 filter blocks, few slab rows -- C5's 512- and 1024-channel convs have one row of 57-113 MB): one
// workgroup per filter block sums its rows into LDS ([tap][32 co][32 ci], tap stride 1025) and writes dW as whole (ci, tap) runs
// of 864 floats per output channel.  The column-parallel kernel above scatters 4-byte writes at a stride of `taps` floats:
// 1.2 TB/s on those layers.
__global__ __launch_bounds__(256) void wgrad_reduce_block_kernel(const float* __restrict__ slab, int nb, int Cout, int Cin, int taps,
                                                                 float* __restrict__ dW, float alpha) {
  extern __shared__ float blk[];                                   // taps x 1025
  const int ncib = Cin / 32;
  const int cib = blockIdx.x % ncib, cob = blockIdx.x / ncib;
  const long long per_b = (long long)Cout * Cin * taps;
  const float* src = slab + (long long)blockIdx.x * taps * 1024;  // this block's [tap][co][ci] in row 0
  const int n4 = taps * 256;                                       // float4 items
  for (int i = threadIdx.x; i < n4; i += 256) {
    f32x4 s = *(const f32x4*)(src + i * 4);
    for (int b = 1; b < nb; ++b) s += *(const f32x4*)(src + b * per_b + i * 4);
    const int tap = i >> 8, rem = (i & 255) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) blk[tap * 1025 + rem + k] = s[k];
  }
  __syncthreads();
  const int run = 32 * taps;                                       // (ci, tap) elements of one output channel of this block
  for (int r = 0; r < 32; ++r) {
    float* dst = dW + ((long long)(cob * 32 + r) * Cin + cib * 32) * taps;
    for (int e = threadIdx.x; e < run; e += 256) {
      const int c = e / taps, tap = e - c * taps;
      dst[e] = alpha * blk[tap * 1025 + r * 32 + c];
    }
  }
}

template <typename T, int ND>
int launch_wgrad(const WgradParams& p, int nb, hipStream_t stream) {
  using TL = WTile<ND>;
  constexpr int PZ = TL::TZ + 2 * TL::PADZ, PY = TL::TY + 2, PX = TL::TX + 2;
  constexpr int PLANE_X = ((PZ * PY * PX * 16 + 255) / 256) * 256 + 64;
  constexpr int PLANE_Y = ((TL::TZ * TL::TY * TL::TX * 16 + 255) / 256) * 256 + 64;
  constexpr int LDS = 4 * PLANE_X + 4 * PLANE_Y;
  IUNET_SET_MAX_LDS((conv3_wgrad_kernel<T, ND>), LDS);
  dim3 grid(nb, p.Cout / 32, p.Cin / 32);
  hipLaunchKernelGGL((conv3_wgrad_kernel<T, ND>), grid, dim3(256), LDS, stream, p);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

}  // namespace

int iunet_conv3_wgrad_v2_blocks(int N, int D, int H, int W, int Cin, int Cout);
int iunet_conv3_wgrad_v2_launch(int dtype, const void* x, long long x_ss, const void* dy, long long dy_ss, float* slab,
                                int N, int D, int H, int W, int Cin, int Cout, const float* x_scale, const float* x_shift,
                                hipStream_t stream);
int iunet_conv2_wgrad_v2_blocks(int N, int H, int W, int Cin, int Cout, int* rows);
int iunet_conv2_wgrad_v2_launch(int dtype, const void* x, long long x_ss, const void* dy, long long dy_ss, float* slab, int N, int H, int W,
                                int Cin, int Cout, const float* x_scale, const float* x_shift, hipStream_t stream);
static bool wgrad_use_v2(int nd) {
  static const bool off = getenv("IUNET_WGRAD_V1") != nullptr;       // A/B runs: the two-workgroups-per-CU structure (both 2-D and 3-D)
  static const bool off2 = getenv("IUNET_WGRAD2D_V1") != nullptr;    // ... in 2-D only
  return !off && !(nd == 2 && off2);
}

extern "C" {

// number of slab rows (one or two per voxel-walking workgroup) of a (co, ci) block and the slab size they need
int iunet_conv3_wgrad_blocks(int nd, int N, int D, int H, int W, int Cin, int Cout) {
  if (N < 1 || D < 1 || H < 1 || W < 1 || Cin < 32 || Cout < 32 || (nd != 2 && nd != 3)) return 0;
  if (wgrad_use_v2(nd)) {
    if (nd == 3) return iunet_conv3_wgrad_v2_blocks(N, D, H, W, Cin, Cout);
    int rows = 0;
    iunet_conv2_wgrad_v2_blocks(N, H, W, Cin, Cout, &rows);            // (the 32 x 32 block writes two slab rows per workgroup)
    return rows;
  }
  const int TZ = nd == 3 ? 2 : 1, TY = nd == 3 ? 8 : 16, TX = nd == 3 ? 16 : 32;
  const long long ntiles = (long long)N * ((D + TZ - 1) / TZ) * ((H + TY - 1) / TY) * ((W + TX - 1) / TX);
  const int pairs = (Cin / 32) * (Cout / 32);
  long long nb = (512 + pairs - 1) / pairs;          // about two workgroups per CU in total
  if (nb > ntiles) nb = ntiles;
  if (nb < 1) nb = 1;
  return (int)nb;
}

long long iunet_conv3_wgrad_slab_floats(int nd, int N, int D, int H, int W, int Cin, int Cout) {
  const int taps = nd == 3 ? 27 : 9;
  return (long long)iunet_conv3_wgrad_blocks(nd, N, D, H, W, Cin, Cout) * (Cout / 32) * (Cin / 32) * taps * 1024;
}

static int wgrad_impl(int dtype, int nd, const void* x, long long x_ss, const void* dy, long long dy_ss, void* slab,
                      void* dW, float alpha, int N, int D, int H, int W, int Cin, int Cout, const float* x_scale,
                      const float* x_shift, void* stream) {
  IUNET_REQUIRE(dtype == 0 || dtype == 1, "conv3_wgrad: bad dtype %d", dtype);
  IUNET_REQUIRE(nd == 2 || nd == 3, "conv3_wgrad: nd must be 2 or 3");
  IUNET_REQUIRE(Cin >= 32 && Cout >= 32 && Cin % 32 == 0 && Cout % 32 == 0, "conv3_wgrad: channels must be positive multiples of 32 (%d, %d)", Cin, Cout);
  IUNET_REQUIRE_GRID("conv3_wgrad", N, D, H, W);
  IUNET_REQUIRE(nd == 3 || D == 1, "conv3_wgrad: 2-D needs D == 1");
  IUNET_REQUIRE(x && dy && slab && dW, "conv3_wgrad: null pointer");
  WgradParams p;
  p.x = x; p.x_ss = x_ss; p.dy = dy; p.dy_ss = dy_ss; p.slab = (float*)slab; p.x_scale = x_scale; p.x_shift = x_shift;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  const int TZ = nd == 3 ? 2 : 1, TY = nd == 3 ? 8 : 16, TX = nd == 3 ? 16 : 32;
  p.tilesZ = (D + TZ - 1) / TZ; p.tilesY = (H + TY - 1) / TY; p.tilesX = (W + TX - 1) / TX;
  const int nb = iunet_conv3_wgrad_blocks(nd, N, D, H, W, Cin, Cout);
  int rc;
  if (wgrad_use_v2(nd) && nd == 2) rc = iunet_conv2_wgrad_v2_launch(dtype, x, x_ss, dy, dy_ss, (float*)slab, N, H, W, Cin, Cout, x_scale, x_shift, (hipStream_t)stream);
  else if (wgrad_use_v2(nd)) rc = iunet_conv3_wgrad_v2_launch(dtype, x, x_ss, dy, dy_ss, (float*)slab, N, D, H, W, Cin, Cout, x_scale, x_shift, (hipStream_t)stream);
  else if (dtype == 0) rc = nd == 3 ? launch_wgrad<f16, 3>(p, nb, (hipStream_t)stream) : launch_wgrad<f16, 2>(p, nb, (hipStream_t)stream);
  else rc = nd == 3 ? launch_wgrad<bf16, 3>(p, nb, (hipStream_t)stream) : launch_wgrad<bf16, 2>(p, nb, (hipStream_t)stream);
  if (rc != IUNET_OK) return rc;
  const int taps = nd == 3 ? 27 : 9;
  const long long total = (long long)Cout * Cin * taps;
  const int nblk = (Cout / 32) * (Cin / 32);
  if (nblk >= 128 && nb <= 4) {                        // wide layer: LDS-transposing reduce, one workgroup per filter block
    const int lds = taps * 1025 * 4;
    IUNET_SET_MAX_LDS(wgrad_reduce_block_kernel, lds);
    hipLaunchKernelGGL(wgrad_reduce_block_kernel, dim3(nblk), dim3(256), lds, (hipStream_t)stream,
                       (const float*)slab, nb, Cout, Cin, taps, (float*)dW, alpha);
  } else {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(total / 64)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)slab, nb, Cout, Cin, taps, (float*)dW, alpha);
  }
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

// dW fp32 [Cout][Cin][taps] = alpha * sum over samples and voxels of dy (x) shifted x
int iunet_conv3_wgrad(int dtype, int nd, const void* x, long long x_ss, const void* dy, long long dy_ss, void* slab,
                      void* dW, float alpha, int N, int D, int H, int W, int Cin, int Cout, void* stream) {
  return wgrad_impl(dtype, nd, x, x_ss, dy, dy_ss, slab, dW, alpha, N, D, H, W, Cin, Cout, nullptr, nullptr, stream);
}

// the same with the conv input given as relu(x_scale[c] * x + x_shift[c]) (see iunet_conv3_fwd_act)
int iunet_conv3_wgrad_act(int dtype, int nd, const void* x, long long x_ss, const void* dy, long long dy_ss, void* slab,
                          void* dW, float alpha, const void* x_scale, const void* x_shift, int N, int D, int H, int W,
                          int Cin, int Cout, void* stream) {
  IUNET_REQUIRE(x_scale && x_shift, "conv3_wgrad_act: null scale / shift");
  return wgrad_impl(dtype, nd, x, x_ss, dy, dy_ss, slab, dW, alpha, N, D, H, W, Cin, Cout, (const float*)x_scale,
                    (const float*)x_shift, stream);
}

}  // extern "C"
