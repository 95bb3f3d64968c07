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

// Same operator with the weights of the ci block resident in LDS (Cout <= 128: 16 KB per 32 output channels of the
// forward conv) and two 16-voxel groups per wave and step: as in convT_lds_kernel (pointwise.hip), the per-wave weight
// fetches through the vector cache were 2x the tensor traffic.
// NCI: input-channel blocks (of 32) per workgroup.  Every ci block needs the whole gradient tensor dy; with one block per
// workgroup the level-0 launch (Cin = 64) read its 268 MB twice and ran at the HBM rate of the doubled traffic.
// [r3] The operator fragments are loop-invariant: left alone, the compiler hoists them out of the voxel loop into registers (NCI x NK x
// 2^d x 2 of them: 352 registers at NK = 2, 512 + 464 B of scratch at NK = 4 -- one wave per SIMD).  Their LDS offset is made opaque
// per iteration, and the operator sizes that leave room for two workgroups per CU (<= 64 KB) are compiled for 256 registers.
template <typename T, int ND, int NK, int NCI>
__global__ __launch_bounds__(256, (NCI * NK * (ND == 3 ? 16 : 8) <= 64) ? 2 : 1) void convT_dgrad_lds_kernel(ConvTDgradParams p) {
  using V8 = V8T<T>;
  constexpr int NPOS = ND == 3 ? 8 : 4, G = 2;
  constexpr int WCI = NPOS * NK * 2 * 64;                        // 16-byte granules of one ci block's operator
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int cib0 = blockIdx.y * NCI;
  {
    const u32x4* wsrc = (const u32x4*)p.wpk + (long long)cib0 * WCI;
    stage_to_lds(smem, wsrc, NCI * WCI, threadIdx.x, 256);
  }
  __syncthreads();
  const int xg = (p.W + 15) / 16;
  const long long rows = (long long)p.D * p.H * xg, ngroups = rows * p.N;
  const int Do = ND == 3 ? p.D * 2 : 1, Ho = p.H * 2, Wo = p.W * 2;
  const long long out_plane = (long long)Do * Ho * Wo * 8, in_plane = (long long)p.D * p.H * p.W * 8;
  for (long long g0 = ((long long)blockIdx.x * 4 + wave) * G; g0 < ngroups; g0 += (long long)gridDim.x * 4 * G) {
    unsigned woff = lane * 16;
    asm volatile("" : "+v"(woff));
    const V8* wl = (const V8*)(smem + woff);
    f32x4 acc[NCI][G][2];
    const T* dyb[G];
    long long xoff[G];
    bool okg[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const long long wid = g0 + g < ngroups ? g0 + g : ngroups - 1;
      const int n = (int)(wid / rows);
      const long long r = wid - n * rows;
      const int xb = (int)(r % xg), y = (int)((r / xg) % p.H), z = (int)(r / ((long long)xg * p.H));
      const int x = xb * 16 + l15, xc = min(x, p.W - 1);
      okg[g] = x < p.W && g0 + g < ngroups;
      const int oz = ND == 3 ? z * 2 : 0;
      dyb[g] = (const T*)p.dy + n * p.dy_ss + (((long long)oz * Ho + y * 2) * Wo + xc * 2) * 8;
      xoff[g] = n * p.dx_ss + (long long)(cib0 * 4 + q) * in_plane + (((long long)z * p.H + y) * p.W + x) * 8;
#pragma unroll
      for (int c = 0; c < NCI; ++c) { acc[c][g][0] = f32x4{0, 0, 0, 0}; acc[c][g][1] = f32x4{0, 0, 0, 0}; }
    }
#pragma unroll
    for (int s = 0; s < NPOS; ++s) {
      const int a = ND == 3 ? (s >> 2) : 0, b = (s >> 1) & 1, c = s & 1;
      const long long voff = (((long long)a * Ho + b) * Wo + c) * 8;
#pragma unroll
      for (int kc = 0; kc < NK; ++kc) {
        V8 bf[G];
#pragma unroll
        for (int g = 0; g < G; ++g) bf[g] = *(const V8*)(dyb[g] + (long long)(kc * 4 + q) * out_plane + voff);
#pragma unroll
        for (int c = 0; c < NCI; ++c) {
          const V8 a0 = wl[c * WCI + ((s * NK + kc) * 2 + 0) * 64];
          const V8 a1 = wl[c * WCI + ((s * NK + kc) * 2 + 1) * 64];
#pragma unroll
          for (int g = 0; g < G; ++g) {
            acc[c][g][0] = mfma16<T>(a0, bf[g], acc[c][g][0]);
            acc[c][g][1] = mfma16<T>(a1, bf[g], acc[c][g][1]);
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCI; ++c)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        V8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = from_f32<T>(acc[c][g][0][j]); o[4 + j] = from_f32<T>(acc[c][g][1][j]); }
        if (okg[g]) *(V8*)((T*)p.dx + xoff[g] + (long long)c * 4 * in_plane) = o;
      }
  }
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

// ------------------------------------------------------------------ convT weight + bias gradient (MFMA)
// dW[ci][co][pos] = sum_v x[ci][v] * dy[co][2v+pos];  db[co] = sum dy[co][.]
// Voxels are the k dimension: A[ci][k] = x^T, B[k][co] = dy at the 2x-upsampled position, both read
// with the transposing LDS read from channel-innermost images (as conv3_wgrad).  One workgroup walks
// input-voxel tiles for a (32 ci) x (32 co) block; its 4 waves split the 2^d output positions.
typedef short s16x4w __attribute__((ext_vector_type(4)));
typedef short s16x8w __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4w* lds_s4w_ptr;

template <typename T>
__device__ __forceinline__ typename Vec8<T>::type tr_frag2(unsigned addr, unsigned hi_off) {
  const s16x4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4w_ptr)addr);
  const s16x4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4w_ptr)(addr + hi_off));
  const s16x8w v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(typename Vec8<T>::type, v);
}

struct ConvTWgradParams {
  const void* x; long long x_ss;
  const void* dy; long long dy_ss;
  float* wslab;    // [nb][Cin/32][Cout/32][NPOS][32][32]
  float* bslab;    // [nb][Cout]
  int N, D, H, W, Cin, Cout;
  int tilesZ, tilesY, tilesX;
};

// NCI: input-channel blocks (of 32) per workgroup.  The 64 KB dy tile is what this kernel moves; every ci block of a workgroup
// uses the one copy in LDS (its own 8 KB x tile is swapped in between two MFMA phases), so dy is read once per NCI ci blocks.
template <typename T, int ND, int NCI>
__global__ __launch_bounds__(256, 2) void convT_wgrad_kernel(ConvTWgradParams p) {   // 2 per CU: 74 KB of LDS each
  using V8 = V8T<T>;
  constexpr int NPOS = ND == 3 ? 8 : 4;
  constexpr int TZ = ND == 3 ? 2 : 1, TY = ND == 3 ? 4 : 8, TX = 16;       // input-voxel tile (128 voxels)
  constexpr int NVI = TZ * TY * TX;
  constexpr int OZ = ND == 3 ? 2 * TZ : 1, OY = 2 * TY, OX = 2 * TX;         // output tile
  constexpr int NVO = OZ * OY * OX;
  constexpr int PLANE_X = NVI * 16 + 64, PLANE_Y = NVO * 16 + 64;            // 64 mod 256
  constexpr int OFF_Y = 4 * PLANE_X;
  constexpr int NKS = NVI / 32;
  constexpr int PL = NPOS / 4;                                               // positions per wave
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, gh = g >> 1, gl = g & 1;
  const int cib0 = blockIdx.y * NCI, cob = blockIdx.z;
  // lane parts of the tr-read addresses
  const unsigned laneX = lds0 + (pp >> 1) * PLANE_X + (pp & 1) * 8 + (gh * 16 + gl * 8 + q) * 16;
  const unsigned laneY = lds0 + OFF_Y + (pp >> 1) * PLANE_Y + (pp & 1) * 8 + (gh * 2 * OX + 2 * (gl * 8 + q)) * 16;
  unsigned pos_off[PL];
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    const int s = wave * PL + i;
    const int a = ND == 3 ? (s >> 2) : 0, b = (s >> 1) & 1, c = s & 1;
    pos_off[i] = (unsigned)(((a * OY + b) * OX + c) * 16);
  }
  f32x4 acc[NCI][PL][2][2];
#pragma unroll
  for (int c = 0; c < NCI; ++c)
#pragma unroll
    for (int i = 0; i < PL; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t) { acc[c][i][t][0] = f32x4{0, 0, 0, 0}; acc[c][i][t][1] = f32x4{0, 0, 0, 0}; }
  float bacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                   // channels (tid & 3) * 8 + j of this co block

  const int tps = p.tilesZ * p.tilesY * p.tilesX, ntiles = tps * p.N;
  const int Do = ND == 3 ? p.D * 2 : 1, Ho = p.H * 2, Wo = p.W * 2;
  const long long in_plane = (long long)p.D * p.H * p.W * 8, out_plane = (long long)Do * Ho * Wo * 8;
  // Split staging: the global loads of tile t + 1 are issued (into registers) before the MFMA phase of tile t and are
  // written to LDS after it, so their latency hides behind the compute instead of sitting between two barriers.
  constexpr int XIT = NVI * 4 / 256, YIT = NVO * 4 / 256;      // 16-byte items per thread: x 2, dy 16 (3-D) / 8 (2-D)
  u32x4 xr[NCI * XIT], xkeep[NCI > 1 ? XIT : 1];              // xkeep: the second ci block's x tile of the tile being consumed
  V8 yr[YIT];
  auto load_tile = [&](int tile) {
    const int n = tile / tps;
    int trem = tile - n * tps;
    const int tz_i = trem / (p.tilesY * p.tilesX);
    trem -= tz_i * p.tilesY * p.tilesX;
    const int ty_i = trem / p.tilesX, tx_i = trem - ty_i * p.tilesX;
    const int z0 = tz_i * TZ, y0 = ty_i * TY, x0 = tx_i * TX;
    const T* xin = (const T*)p.x + (long long)n * p.x_ss + (long long)cib0 * 4 * in_plane;
    const T* dyin = (const T*)p.dy + (long long)n * p.dy_ss + (long long)cob * 4 * out_plane;
#pragma unroll
    for (int k = 0; k < NCI * XIT; ++k) {                        // x tile: 4 planes per ci block
      const int it = tid + k * 256;
      const int pl = it / NVI, pix = it - pl * NVI;
      const int px = pix % TX, t2 = pix / TX, py = t2 % TY, pz = t2 / TY;
      const int gz = z0 + pz, gy = y0 + py, gx = x0 + px;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (gz < p.D && gy < p.H && gx < p.W) v = *(const u32x4*)(xin + pl * in_plane + (((long long)gz * p.H + gy) * p.W + gx) * 8);
      xr[k] = v;
    }
#pragma unroll
    for (int k = 0; k < YIT; ++k) {                              // dy tile: plane = it & 3 = tid & 3
      const int it = tid + k * 256;
      const int pl = it & 3, pix = it >> 2;
      const int px = pix % OX, t2 = pix / OX, py = t2 % OY, pz = t2 / OY;
      const int gz = 2 * z0 + pz, gy = 2 * y0 + py, gx = 2 * x0 + px;
      V8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = from_f32<T>(0.f);
      if (gz < Do && gy < Ho && gx < Wo) v = *(const V8*)(dyin + pl * out_plane + (((long long)gz * Ho + gy) * Wo + gx) * 8);
      yr[k] = v;
    }
  };
  auto commit_x = [&](const u32x4* src) {                      // one ci block's 4 planes
#pragma unroll
    for (int k = 0; k < XIT; ++k) {
      const int it = tid + k * 256;
      const int pl = it / NVI, pix = it - pl * NVI;
      *(u32x4*)(smem + pl * PLANE_X + pix * 16) = src[k];
    }
  };
  auto commit_tile = [&]() {
    commit_x(xr);
    if (NCI > 1) {
#pragma unroll
      for (int k = 0; k < XIT; ++k) xkeep[k] = xr[XIT + k];
    }
#pragma unroll
    for (int k = 0; k < YIT; ++k) {
      const int it = tid + k * 256;
      const int pl = it & 3, pix = it >> 2;
      *(V8*)(smem + OFF_Y + pl * PLANE_Y + pix * 16) = yr[k];
#pragma unroll
      for (int j = 0; j < 8; ++j) bacc[j] += to_f32<T>(yr[k][j]);
    }
  };
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();                                             // the previous tile's fragment reads are done
    commit_tile();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);       // in flight during the MFMA phase
#pragma unroll
    for (int c = 0; c < NCI; ++c) {
      if (c > 0) {                                               // next ci block: its x tile replaces the first one's, dy stays
        __syncthreads();
        commit_x(xkeep);
        __syncthreads();
      }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        // in-voxel rows 2ks, 2ks+1 (the lane's own row adds gh, folded into laneX / laneY)
        const int r = 2 * ks, rz = r / TY, ry = r % TY;
        const unsigned offX = (unsigned)(r * TX * 16);
        const unsigned offY = (unsigned)(((2 * rz * OY + 2 * ry) * OX) * 16);
        const V8 a0 = tr_frag2<T>(laneX + offX, 64);
        const V8 a1 = tr_frag2<T>(laneX + offX + 2 * PLANE_X, 64);
#pragma unroll
        for (int i = 0; i < PL; ++i) {
          const V8 b0 = tr_frag2<T>(laneY + offY + pos_off[i], 128);                 // 4 voxels further = 8 output pixels
          const V8 b1 = tr_frag2<T>(laneY + offY + pos_off[i] + 2 * PLANE_Y, 128);
          acc[c][i][0][0] = mfma16<T>(a0, b0, acc[c][i][0][0]);
          acc[c][i][0][1] = mfma16<T>(a0, b1, acc[c][i][0][1]);
          acc[c][i][1][0] = mfma16<T>(a1, b0, acc[c][i][1][0]);
          acc[c][i][1][1] = mfma16<T>(a1, b1, acc[c][i][1][1]);
        }
      }
    }
  }
  // slab: rows = ci (4g + j), cols = co (lane & 15)
#pragma unroll
  for (int c = 0; c < NCI; ++c) {
    float* ws = p.wslab + ((((long long)blockIdx.x * (gridDim.y * NCI) + cib0 + c) * gridDim.z + cob) * NPOS) * 1024;
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      const int s = wave * PL + i;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) ws[s * 1024 + (t * 16 + 4 * g + j) * 32 + u * 16 + i16] = acc[c][i][t][u][j];
    }
  }
  if (blockIdx.y == 0) {       // bias gradient: threads with equal (tid & 3) own the same 8 channels
    __syncthreads();
    float* red = (float*)smem;
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid * 8 + j] = bacc[j];
    __syncthreads();
    if (tid < 32) {
      const int pl = tid >> 3, j = tid & 7;
      float sum = 0.f;
      for (int k = pl; k < 256; k += 4) sum += red[k * 8 + j];
      p.bslab[(long long)blockIdx.x * p.Cout + cob * 32 + pl * 8 + j] = sum;
    }
  }
}

// dW[ci][co][pos] = sum_b slab[b][cib][cob][pos][ci%32][co%32] (threads walk the slab order)
__global__ __launch_bounds__(256) void convT_wgrad_reduce_kernel(const float* __restrict__ slab, int nb, int Cin, int Cout, int npos,
                                                                 float* __restrict__ dW) {
  // 64 slab columns per block as 16 float4 lanes x 16 row groups (16-byte loads, nb / 16 of them per thread), fixed order
  __shared__ f32x4 red[16][16];
  const long long per_b4 = (long long)Cin * Cout * npos / 4;
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
  const int ncob = Cout / 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long j = j4 * 4 + k;
    const int c = (int)(j & 31), r = (int)((j >> 5) & 31);
    long long t = j >> 10;
    const int pos = (int)(t % npos); t /= npos;
    const int cob = (int)(t % ncob), cib = (int)(t / ncob);
    dW[((long long)(cib * 32 + r) * Cout + cob * 32 + c) * npos + pos] = s[k];
  }
}

// ------------------------------------------------------------------ first conv weight gradient (MFMA)
// dW[co][c][tap] = sum_v dy[co][v] * x[c][v + tap - 1]: voxels are the k dimension; A = dy^T through the
// transposing LDS read, B = im2col columns gathered from a 16-bit LDS image of the input halo tile
// (8 scalar reads per k-quad).  x is read with the caller's strides / dtype.  Bound by reading dy.
struct FirstWgradParams {
  const void* x; long long sN, sC, sD, sH, sW; int in_dtype;
  const void* dy; long long dy_ss;
  // optional BatchNorm + ReLU backward applied while staging: `dy` is then the gradient of the activation (dz), `yraw` the
  // conv's raw output, and the operand is a * (dz * mask - c1 - xhat * c2) rounded as bn_bwd_apply_kernel stores it
  const void* yraw; long long y_ss;
  const float* mean; const float* invstd; const float* coef; const float* scale; const float* shift;
  float* slab;     // [nb][Cout][KKP]
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
__global__ __launch_bounds__(256, 2) void first_wgrad_kernel(FirstWgradParams p) {
  using V8 = V8T<T>;
  constexpr int TZ = ND == 3 ? 4 : 1, TY = ND == 3 ? 8 : 16, TX = ND == 3 ? 16 : 32, PADZ = ND == 3 ? 1 : 0;
  constexpr int PZ = TZ + 2 * PADZ, PY = TY + 2, PX = TX + 2, NPIX = PZ * PY * PX, NVOX = TZ * TY * TX;
  constexpr int TAPS = ND == 3 ? 27 : 9, KK = TAPS * CIN, NT = (KK + 15) / 16, KKP = NT * 16;
  constexpr int FX = TX / 16, NKS = NVOX / 32;
  constexpr int PLANE_Y = NVOX * 16 + 64;                      // 64 mod 256: conflict-free tr reads
  constexpr int XS_BYTES = ((CIN * NPIX * 2 + 255) / 256) * 256;
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  T* xs = (T*)smem;                                           // [CIN][NPIX]
  unsigned char* dys = smem + XS_BYTES;                        // 4 planes
  __shared__ __attribute__((aligned(16))) float par[7 * 32];   // [scale | shift | mean | invstd | coef0 | coef1 | coef2][32 channels of this Cout tile]
  if (p.yraw != nullptr && threadIdx.x < 7 * 32) {
    const int k = threadIdx.x >> 5, c = blockIdx.y * 32 + (threadIdx.x & 31);
    par[threadIdx.x] = k == 0 ? p.scale[c] : k == 1 ? p.shift[c] : k == 2 ? p.mean[c] : k == 3 ? p.invstd[c] : p.coef[c * 3 + (k - 4)];
  }
  const unsigned lds_y = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)dys;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, gh = g >> 1, gl = g & 1;
  const int cob = blockIdx.y;
  const unsigned laneY = lds_y + (pp >> 1) * PLANE_Y + (pp & 1) * 8 + (gh * 16 + gl * 8 + q) * 16;
  // this lane's im2col column of every column tile: kk = 16 ct + i16 -> (tap, channel) -> element offset
  int col_off[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
    const int kk = ct * 16 + i16, kc = kk < KK ? kk : 0;        // padded columns are never stored
    const int tap = kc / CIN, c = kc % CIN;
    const int dz = ND == 3 ? tap / 9 : 0, dy_ = (tap / 3) % 3, dx = tap % 3;
    col_off[ct] = c * NPIX + (dz * PY + dy_) * PX + dx;
  }
  f32x4 acc[2][NT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) acc[t][ct] = f32x4{0, 0, 0, 0};
  const int tps = p.tilesZ * p.tilesY * p.tilesX, ntiles = tps * p.N;
  const long long plane = (long long)p.D * p.H * p.W * 8;
  // split staging: the loads of tile t + 1 (input halo and dy tile) are issued into registers before the MFMA phase of
  // tile t and written to LDS after it
  constexpr int XIT = (NPIX * CIN + 255) / 256, YIT = NVOX * 4 / 256;
  float xr[XIT];
  u32x4 yr[YIT], yw[YIT];
  unsigned oky = 0;                  // bit it: dy item `it` of the staged tile lies inside the image
  auto load_tile = [&](int tile) {
    oky = 0;
    const int n = tile / tps;
    int trem = tile - n * tps;
    const int tz_i = trem / (p.tilesY * p.tilesX);
    trem -= tz_i * p.tilesY * p.tilesX;
    const int ty_i = trem / p.tilesX, tx_i = trem - ty_i * p.tilesX;
    const int z0 = tz_i * TZ, y0 = ty_i * TY, x0 = tx_i * TX;
#pragma unroll
    for (int k = 0; k < XIT; ++k) {
      const int it = tid + k * 256;
      const int c = it / NPIX, pix = it - c * NPIX;
      const int px = pix % PX, t2 = pix / PX, py = t2 % PY, pz = t2 / PY;
      const int gz = z0 + pz - PADZ, gy = y0 + py - 1, gx = x0 + px - 1;
      float v = 0.f;
      if (it < NPIX * CIN && (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
        v = load_in2(p.x, n * p.sN + c * p.sC + gz * p.sD + gy * p.sH + gx * p.sW, p.in_dtype);
      xr[k] = v;
    }
    const T* dyin = (const T*)p.dy + (long long)n * p.dy_ss + (long long)cob * 4 * plane;
#pragma unroll
    for (int it = 0; it < YIT; ++it) {
      const int idx = tid + it * 256, pl = idx / NVOX, pix = idx - pl * NVOX;
      const int px = pix % TX, t2 = pix / TX, py = t2 % TY, pz = t2 / TY;
      const int gz = z0 + pz, gy = y0 + py, gx = x0 + px;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (gz < p.D && gy < p.H && gx < p.W) v = *(const u32x4*)(dyin + pl * plane + (((long long)gz * p.H + gy) * p.W + gx) * 8);
      yr[it] = v;
      oky |= (gz < p.D && gy < p.H && gx < p.W) ? (1u << it) : 0u;
      if (p.yraw != nullptr) {
        u32x4 w = u32x4{0u, 0u, 0u, 0u};
        if (gz < p.D && gy < p.H && gx < p.W)
          w = *(const u32x4*)((const T*)p.yraw + (long long)n * p.y_ss + (long long)(cob * 4 + pl) * plane + (((long long)gz * p.H + gy) * p.W + gx) * 8);
        yw[it] = w;
      }
    }
  };
  auto commit_tile = [&]() {
#pragma unroll
    for (int k = 0; k < XIT; ++k) {
      const int it = tid + k * 256;
      if (it < NPIX * CIN) xs[it] = from_f32<T>(xr[k]);
    }
#pragma unroll
    for (int it = 0; it < YIT; ++it) {
      const int idx = tid + it * 256, pl = idx / NVOX, pix = idx - pl * NVOX;
      u32x4 v = yr[it];
      if (p.yraw != nullptr && ((oky >> it) & 1u)) {      // dy = a * (dz * [z > 0] - c1 - xhat * c2) as bn_bwd_apply_kernel; outside the image: 0
        const V8 g = __builtin_bit_cast(V8, v), yy = __builtin_bit_cast(V8, yw[it]);
        V8 o;
        // the 7 per-channel constants of this plane's 8 channels: broadcast LDS reads (one global load per constant and
        // element -- 448 per thread and tile -- was what this kernel spent its time on)
        const f32x4* pp = (const f32x4*)(par + pl * 8);
        float cs[7][8];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
          const f32x4 lo = pp[k * 8], hi = pp[k * 8 + 1];
#pragma unroll
          for (int j = 0; j < 4; ++j) { cs[k][j] = lo[j]; cs[k][4 + j] = hi[j]; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float yv = to_f32<T>(yy[j]);
          const float zv = to_f32<T>(from_f32<T>(fmaf(cs[0][j], yv, cs[1][j])));
          const float d = zv > 0.f ? to_f32<T>(g[j]) : 0.f;
          const float xh = (yv - cs[2][j]) * cs[3][j];
          o[j] = from_f32<T>(cs[4][j] * (d - cs[5][j] - xh * cs[6][j]));
        }
        v = __builtin_bit_cast(u32x4, o);
      }
      *(u32x4*)(dys + pl * PLANE_Y + pix * 16) = v;
    }
  };
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    commit_tile();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
    for (int ks = wave; ks < NKS; ks += 4) {                 // the four waves split the k-steps
      const int f = 2 * ks + gh;                             // this lane's fragment (16 x voxels)
      const int xh = f % FX, row = f / FX, fy = row % TY, fz = row / TY;
      const int vbase = (fz * PY + fy) * PX + xh * 16 + gl * 8;      // first of the lane's 8 voxels, tap (0,0,0)
      const V8 a0 = tr_frag2<T>(laneY + (unsigned)(2 * ks * 256), 64);
      const V8 a1 = tr_frag2<T>(laneY + (unsigned)(2 * ks * 256) + 2 * PLANE_Y, 64);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        V8 b;
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = xs[vbase + j + col_off[ct]];
        acc[0][ct] = mfma16<T>(a0, b, acc[0][ct]);
        acc[1][ct] = mfma16<T>(a1, b, acc[1][ct]);
      }
    }
  }
  // sum the four waves' accumulators through LDS, then one slab row per workgroup
  __syncthreads();
  float* red = (float*)smem;                                  // [4 waves][32 co][KKP]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(wave * 32 + t * 16 + 4 * g + j) * KKP + ct * 16 + i16] = acc[t][ct][j];
  __syncthreads();
  float* slab = p.slab + ((long long)blockIdx.x * p.Cout + cob * 32) * KKP;
  for (int i = tid; i < 32 * KKP; i += 256)
    slab[i] = red[i] + red[32 * KKP + i] + red[2 * 32 * KKP + i] + red[3 * 32 * KKP + i];
}

// dW[co][c][tap] = sum_b slab[b][co][tap * Cin + c]: 32 outputs x 8 row groups per block, fixed order
__global__ __launch_bounds__(256) void first_wgrad_reduce_kernel(const float* __restrict__ slab, int nb, int Cout, int Cin, int taps,
                                                                 int KKP, float* __restrict__ dW) {
  __shared__ float red[8][32];
  const int total = Cout * Cin * taps;
  const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + o;
  float s = 0.f;
  if (i < total) {
    const int tap = i % taps, c = (i / taps) % Cin, co = i / (taps * Cin);
    const float* src = slab + (long long)co * KKP + tap * Cin + c;
    const long long st = (long long)Cout * KKP;
    int b = grp;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (; b + 24 < nb; b += 32) {                       // four loads in flight (one waited-for load per row took 21 us)
      const float a0 = src[b * st], a1 = src[(b + 8) * st], a2 = src[(b + 16) * st], a3 = src[(b + 24) * st];
      s += a0; s1 += a1; s2 += a2; s3 += a3;
    }
    for (; b < nb; b += 8) s += src[b * st];
    s = (s + s1) + (s2 + s3);
  }
  red[grp][o] = s;
  __syncthreads();
  if (grp == 0 && i < total) {
    float a = red[0][o];
#pragma unroll
    for (int g = 1; g < 8; ++g) a += red[g][o];
    dW[i] = a;
  }
}

}  // namespace

#define DT_OK(dt) IUNET_REQUIRE((dt) == 0 || (dt) == 1, "dtype must be 0 (f16) or 1 (bf16), got %d", (dt))

extern "C" int iunet_reduce_slab(void* slab, int nparts, long long n, void* out, float alpha, int accumulate, void* stream);

extern "C" {

int iunet_pack_convT_dgrad(int dtype, const void* w, void* dst, int Cin, int Cout, int npos, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(w && dst, "pack_convT_dgrad: null pointer");
  IUNET_REQUIRE(Cin >= 32 && Cout >= 32 && Cin % 32 == 0 && Cout % 32 == 0, "pack_convT_dgrad: channels must be positive multiples of 32");
  IUNET_REQUIRE(npos == 4 || npos == 8, "pack_convT_dgrad: npos must be 4 or 8");
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
  IUNET_REQUIRE(nd == 2 || nd == 3, "convT_dgrad: nd must be 2 or 3");
  IUNET_REQUIRE_GRID("convT_dgrad", N, D, H, W);
  IUNET_REQUIRE(Cin >= 32 && Cout >= 32 && Cin % 32 == 0 && Cout % 32 == 0, "convT_dgrad: channels must be positive multiples of 32 (%d, %d)", Cin, Cout);
  ConvTDgradParams p;
  p.dy = dy; p.dy_ss = dy_ss; p.dx = dx; p.dx_ss = dx_ss; p.wpk = wpk; p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  const long long waves = (long long)N * D * H * ((W + 15) / 16);
  const int nk = Cout / 32;
  if (nk <= 4 && waves >= 256) {
    // two ci blocks per workgroup (dy read once for both) where their operators stay within 32 KB of LDS
    const int nci = (nk <= 2 && (Cin / 32) % 2 == 0 && (nd == 3 ? 8 : 4) * nk * 2 * 1024 * 2 <= 32768) ? 2 : 1;
    const int lds = (nd == 3 ? 8 : 4) * nk * 2 * 1024 * nci;
    int gx = (int)((waves + 7) / 8);
    const int cap = 1024 / (Cin / 32 / nci);
    if (gx > cap) gx = cap;
    dim3 g2(gx, Cin / 32 / nci);
#define CDL(TT, NDV, NKV) do { if (nci == 2) { IUNET_SET_MAX_LDS((convT_dgrad_lds_kernel<TT, NDV, NKV, 2>), lds); \
      hipLaunchKernelGGL((convT_dgrad_lds_kernel<TT, NDV, NKV, 2>), g2, dim3(256), lds, (hipStream_t)stream, p); } \
    else { IUNET_SET_MAX_LDS((convT_dgrad_lds_kernel<TT, NDV, NKV, 1>), lds); \
      hipLaunchKernelGGL((convT_dgrad_lds_kernel<TT, NDV, NKV, 1>), g2, dim3(256), lds, (hipStream_t)stream, p); } } while (0)
#define CDL_NK(TT, NDV) switch (nk) { case 1: CDL(TT, NDV, 1); break; case 2: CDL(TT, NDV, 2); break; case 3: CDL(TT, NDV, 3); break; default: CDL(TT, NDV, 4); break; }
    if (dtype == 0) { if (nd == 3) { CDL_NK(f16, 3) } else { CDL_NK(f16, 2) } }
    else            { if (nd == 3) { CDL_NK(bf16, 3) } else { CDL_NK(bf16, 2) } }
#undef CDL_NK
#undef CDL
    IUNET_CHECK_HIP(hipGetLastError());
    return IUNET_OK;
  }
  dim3 grid((unsigned)((waves + 3) / 4), Cin / 32);
  if (dtype == 0) { if (nd == 3) hipLaunchKernelGGL((convT_dgrad_kernel<f16, 3>), grid, dim3(256), 0, (hipStream_t)stream, p);
                    else hipLaunchKernelGGL((convT_dgrad_kernel<f16, 2>), grid, dim3(256), 0, (hipStream_t)stream, p); }
  else { if (nd == 3) hipLaunchKernelGGL((convT_dgrad_kernel<bf16, 3>), grid, dim3(256), 0, (hipStream_t)stream, p);
         else hipLaunchKernelGGL((convT_dgrad_kernel<bf16, 2>), grid, dim3(256), 0, (hipStream_t)stream, p); }
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_convT_wgrad_blocks(int nd, int N, int D, int H, int W, int Cin, int Cout) {
  if (N < 1 || D < 1 || H < 1 || W < 1 || Cin < 32 || Cout < 32) return 0;
  const int TZ = nd == 3 ? 2 : 1, TY = nd == 3 ? 4 : 8, TX = 16;
  const long long ntiles = (long long)N * ((D + TZ - 1) / TZ) * ((H + TY - 1) / TY) * ((W + TX - 1) / TX);
  const int nci = (Cin / 32) % 2 == 0 ? 2 : 1;                       // ci blocks per workgroup (convT_wgrad_kernel: NCI)
  const int pairs = (Cin / 32 / nci) * (Cout / 32);
  long long nb = (512 + pairs - 1) / pairs;
  if (nb > ntiles) nb = ntiles;
  return (int)(nb < 1 ? 1 : nb);
}

// wslab: blocks*Cin*Cout*npos floats, bslab: blocks*Cout floats (scratch); dW fp32 [Cin][Cout][npos], db fp32 [Cout]
int iunet_convT_wgrad(int dtype, int nd, const void* x, long long x_ss, const void* dy, long long dy_ss, void* wslab,
                      void* bslab, void* dW, void* db, int N, int D, int H, int W, int Cin, int Cout, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(x && dy && wslab && bslab && dW && db, "convT_wgrad: null pointer");
  IUNET_REQUIRE(Cin >= 32 && Cout >= 32 && Cin % 32 == 0 && Cout % 32 == 0, "convT_wgrad: channels must be positive multiples of 32");
  IUNET_REQUIRE(nd == 2 || nd == 3, "convT_wgrad: nd must be 2 or 3");
  IUNET_REQUIRE_GRID("convT_wgrad", N, D, H, W);
  ConvTWgradParams p;
  p.x = x; p.x_ss = x_ss; p.dy = dy; p.dy_ss = dy_ss; p.wslab = (float*)wslab; p.bslab = (float*)bslab;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  const int TZ = nd == 3 ? 2 : 1, TY = nd == 3 ? 4 : 8, TX = 16;
  p.tilesZ = (D + TZ - 1) / TZ; p.tilesY = (H + TY - 1) / TY; p.tilesX = (W + TX - 1) / TX;
  const int nb = iunet_convT_wgrad_blocks(nd, N, D, H, W, Cin, Cout);
  const int npos = nd == 3 ? 8 : 4;
  const int nvi = 128, nvo = nd == 3 ? 1024 : 512;
  const int lds = 4 * (nvi * 16 + 64) + 4 * (nvo * 16 + 64);
  const int nci = (Cin / 32) % 2 == 0 ? 2 : 1;
  dim3 grid(nb, Cin / 32 / nci, Cout / 32);
#define CTW(TT, NDV) do { if (nci == 2) { IUNET_SET_MAX_LDS((convT_wgrad_kernel<TT, NDV, 2>), lds); \
      hipLaunchKernelGGL((convT_wgrad_kernel<TT, NDV, 2>), grid, dim3(256), lds, (hipStream_t)stream, p); } \
    else { IUNET_SET_MAX_LDS((convT_wgrad_kernel<TT, NDV, 1>), lds); \
      hipLaunchKernelGGL((convT_wgrad_kernel<TT, NDV, 1>), grid, dim3(256), lds, (hipStream_t)stream, p); } } while (0)
  if (dtype == 0) { if (nd == 3) CTW(f16, 3); else CTW(f16, 2); } else { if (nd == 3) CTW(bf16, 3); else CTW(bf16, 2); }
#undef CTW
  const long long total = (long long)Cin * Cout * npos;
  hipLaunchKernelGGL(convT_wgrad_reduce_kernel, dim3((unsigned)(total / 64)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)wslab, nb, Cin, Cout, npos, (float*)dW);
  IUNET_CHECK_HIP(hipGetLastError());
  return iunet_reduce_slab(bslab, nb, Cout, db, 1.0f, 0, stream);
}

int iunet_first_conv_wgrad_blocks(int nd, int N, int D, int H, int W) {
  if (N < 1 || D < 1 || H < 1 || W < 1) return 0;
  const int TZ = nd == 3 ? 4 : 1, TY = nd == 3 ? 8 : 16, TX = nd == 3 ? 16 : 32;
  const long long ntiles = (long long)N * ((D + TZ - 1) / TZ) * ((H + TY - 1) / TY) * ((W + TX - 1) / TX);
  return (int)(ntiles < 512 ? ntiles : 512);
}

// slab: iunet_first_conv_wgrad_blocks * Cout * 112 floats of scratch; dW fp32 [Cout][Cin][taps]
static int first_wgrad_impl(int dtype, int nd, const void* x, int in_dtype, const long long* in_strides, const void* dy,
                            long long dy_ss, void* slab, void* dW, int N, int D, int H, int W, int Cin, int Cout,
                            const void* yraw, long long y_ss, const float* mean, const float* invstd, const float* coef,
                            const float* scale, const float* shift, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(x && dy && slab && dW && in_strides, "first_conv_wgrad: null pointer");
  IUNET_REQUIRE(Cin >= 1 && Cin <= 4 && Cout >= 32 && Cout % 32 == 0, "first_conv_wgrad: Cin 1..4, Cout a positive multiple of 32");
  IUNET_REQUIRE(nd == 2 || nd == 3, "first_conv_wgrad: nd must be 2 or 3");
  IUNET_REQUIRE(in_dtype >= 0 && in_dtype <= 3, "first_conv_wgrad: bad input dtype %d", in_dtype);
  IUNET_REQUIRE_GRID("first_conv_wgrad", N, D, H, W);
  FirstWgradParams p;
  p.x = x; p.sN = in_strides[0]; p.sC = in_strides[1]; p.sD = in_strides[2]; p.sH = in_strides[3]; p.sW = in_strides[4];
  p.in_dtype = in_dtype; p.dy = dy; p.dy_ss = dy_ss; p.slab = (float*)slab;
  p.yraw = yraw; p.y_ss = y_ss; p.mean = mean; p.invstd = invstd; p.coef = coef; p.scale = scale; p.shift = shift;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  const int TZ = nd == 3 ? 4 : 1, TY = nd == 3 ? 8 : 16, TX = nd == 3 ? 16 : 32;
  p.tilesZ = (D + TZ - 1) / TZ; p.tilesY = (H + TY - 1) / TY; p.tilesX = (W + TX - 1) / TX;
  const int nb = iunet_first_conv_wgrad_blocks(nd, N, D, H, W);
  const int taps = nd == 3 ? 27 : 9, KKP = ((taps * Cin + 15) / 16) * 16;
  const int npix = (nd == 3 ? 6 : 1) * (TY + 2) * (TX + 2);
  const int xs_bytes = ((Cin * npix * 2 + 255) / 256) * 256;
  int lds = xs_bytes + 4 * (512 * 16 + 64);
  if (lds < 4 * 32 * KKP * 4) lds = 4 * 32 * KKP * 4;
  dim3 grid(nb, Cout / 32);
#define FW(TT, NDV, CI) do { IUNET_SET_MAX_LDS((first_wgrad_kernel<TT, NDV, CI>), lds); \
    hipLaunchKernelGGL((first_wgrad_kernel<TT, NDV, CI>), grid, dim3(256), lds, (hipStream_t)stream, p); } while (0)
#define FW_CIN(TT, NDV) switch (Cin) { case 1: FW(TT, NDV, 1); break; case 2: FW(TT, NDV, 2); break; case 3: FW(TT, NDV, 3); break; default: FW(TT, NDV, 4); break; }
  if (dtype == 0) { if (nd == 3) { FW_CIN(f16, 3) } else { FW_CIN(f16, 2) } }
  else            { if (nd == 3) { FW_CIN(bf16, 3) } else { FW_CIN(bf16, 2) } }
#undef FW_CIN
#undef FW
  const int total = Cout * Cin * taps;
  hipLaunchKernelGGL(first_wgrad_reduce_kernel, dim3((total + 31) / 32), dim3(256), 0, (hipStream_t)stream,
                     (const float*)slab, nb, Cout, Cin, taps, KKP, (float*)dW);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_first_conv_wgrad(int dtype, int nd, const void* x, int in_dtype, const long long* in_strides, const void* dy,
                           long long dy_ss, void* slab, void* dW, int N, int D, int H, int W, int Cin, int Cout,
                           void* stream) {
  return first_wgrad_impl(dtype, nd, x, in_dtype, in_strides, dy, dy_ss, slab, dW, N, D, H, W, Cin, Cout, nullptr, 0, nullptr,
                          nullptr, nullptr, nullptr, nullptr, stream);
}

// first_conv_wgrad with the second BatchNorm-backward pass folded in: dz = gradient of the first conv's activation, y = its raw
// output; mean / invstd / scale / shift from the forward pass, coef [Cout][3] from the BatchNorm-backward reduction
// (iunet_bn_relu_bwd with dy = NULL).  The gradient of the raw output is never written.
int iunet_first_conv_wgrad_bn(int dtype, int nd, const void* x, int in_dtype, const long long* in_strides, const void* dz,
                              long long dz_ss, const void* y, long long y_ss, const void* mean, const void* invstd,
                              const void* coef, const void* scale, const void* shift, void* slab, void* dW, int N, int D, int H,
                              int W, int Cin, int Cout, void* stream) {
  IUNET_REQUIRE(y && mean && invstd && coef && scale && shift, "first_conv_wgrad_bn: null pointer");
  return first_wgrad_impl(dtype, nd, x, in_dtype, in_strides, dz, dz_ss, slab, dW, N, D, H, W, Cin, Cout, y, y_ss,
                          (const float*)mean, (const float*)invstd, (const float*)coef, (const float*)scale,
                          (const float*)shift, stream);
}

}  // extern "C"
