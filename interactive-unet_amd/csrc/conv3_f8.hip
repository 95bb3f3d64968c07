// 3^d convolution on the fp8 matrix cores (BASELINE config C5: "fp8 weights ... on CDNA4 fp8 MFMA"), inference only.
//
// Same wave-specialised structure as conv3_v4.hip (one persistent workgroup per CU: consumer waves read LDS and issue
// MFMAs, loader waves feed the other LDS buffer, one barrier per step, per-XCD brick schedule) -- see that file for
// the measurements behind the structure.  What changes:
//   * the packed weights are OCP e4m3 BYTES in HBM and in LDS ([cob32][chunk16][column pair][dy][2][64][8 bytes]:
//     the K16 order of layouts 1 / 2 with one-byte elements) + one fp32 power-of-two scale per output channel,
//     applied to the fp32 accumulator in the epilogue (exact);
//   * the activations stay 16-bit in HBM (what C5 names), and are rounded to e4m3 (saturating at 448) by the LOADER
//     waves on their way into LDS -- the 16 channels of a chunk become ONE 16-byte granule per halo voxel, so a
//     step's LDS image is half as large and the consumers' fragment reads are ds_read_b64;
//   * the product runs on v_mfma_f32_16x16x32_fp8_fp8 (lane l holds k = 8 (l >> 4) + j in byte j, the bf16 map).
// Half the weight bytes make the filter of every layer up to Cin = 128 resident in LDS for the whole launch
// (16-bit: Cin = 32 only), so 4 loader waves stream activations alone and the consumers keep the cross-step fragment
// pipeline with its 168-register budget.
//
// gfx950 has no mixed fp8 x bf16 MFMA: both operands must be fp8, so "fp8 weights / bf16 activations" can only mean
// bf16 activations in memory.  Numerics: products of two e4m3 values are exact in fp32; the result differs from the
// CPU emulation (oracle/unet_ref.py, weight_quant + act_quant) only by the order of the fp32 sums.
#include "common.h"
#include <cstdlib>
#include <type_traits>

int iunet_conv3_f8k_launch(int dtype, const void* x, long long x_sstride, void* y, long long y_sstride, const void* wpk,
                           const float* wscale, const float* bias, int N, int D, int H, int W, int Cin, int Cout, int epi,
                           int ksplit, float* partial, int small, int in8, int out8, hipStream_t stream);

namespace {

typedef long i64;

template <int ND, bool SMALL> struct F8Tile;
template <> struct F8Tile<3, false> { static constexpr int TZ = 4, TY = 8, TX = 16, PADZ = 1, NCOL = 9, S16 = 1, NCW = 8; };
template <> struct F8Tile<3, true>  { static constexpr int TZ = 2, TY = 8, TX = 16, PADZ = 1, NCOL = 9, S16 = 1, NCW = 4; };
template <bool SMALL> struct F8Tile<2, SMALL> { static constexpr int TZ = 1, TY = 16, TX = 32, PADZ = 0, NCOL = 3, S16 = 2, NCW = 8; };

struct ConvF8Params {
  const void* x;  long long x_sstride;        // 16-bit NHWC8c activations
  void* y;        long long y_sstride;
  const void* wpk;                            // e4m3 bytes, K16 order: [cob][chunk16][column pair][dy][2][64][8]
  const float* wscale;                        // [Cout] power-of-two dequantisation scales
  const float* bias;
  int N, D, H, W, Cin, Cout;
  int tilesZ, tilesY, tilesX;
  int bz, by, bx;
  int nbz, nby, nbx;
  int epi;
  int ksplit;                                 // > 1: blockIdx.z owns Cin / ksplit input channels and writes fp32 partial sums
  float* partial;                             // [ksplit][N][Cout / 8][voxels][8] fp32 (split-K only)
};

template <typename T, int ND, bool WS, bool SMALL>
__global__ __launch_bounds__((F8Tile<ND, SMALL>::NCW * 64 + (WS ? 256 : 512)), 1) void conv3_f8_kernel(ConvF8Params p) {
  using V8 = typename Vec8<T>::type;
  using TL = F8Tile<ND, SMALL>;
  constexpr int NCW = TL::NCW, NLT = WS ? 256 : 512;
  constexpr int TZ = TL::TZ, TY = TL::TY, TX = TL::TX, PADZ = TL::PADZ, NCOL = TL::NCOL, S16 = TL::S16;
  constexpr int FX = TX / 16, NI = TZ * TY * FX / NCW, NR = NI / FX;
  constexpr int PZ = TZ + 2 * PADZ, PY = TY + 2, PX = TX + 2;
  constexpr int NPIX = PZ * PY * PX;
  constexpr int PLANE = ((NPIX * 16 + 255) / 256) * 256;       // one 16-channel sub-chunk of the halo tile: 16 B per voxel
  constexpr int CP = 2 * S16;                                  // 8-channel planes of the 16-bit input per step
  constexpr int ABUF = S16 * PLANE;
  constexpr int NCMB = (NCOL + 1) / 2, KS = NCMB * 3;
  constexpr int WBYTES = KS * 2 * 512;                         // one 16-channel chunk of packed e4m3 weights
  constexpr int WSTEP = S16 * WBYTES;
  constexpr int OFF_W = 2 * ABUF;
  constexpr int AIT = (NPIX + NLT - 1) / NLT;
  constexpr int WIT = (WSTEP / 16 + NLT - 1) / NLT;
  constexpr int NGRP = S16 * NCMB;
  constexpr int NRD = FX * (NR + 2) + 6;
  static_assert(NCW * NI == TZ * TY * FX, "consumer waves x fragments must cover the tile");

  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cob = blockIdx.y;
  // blocks b and b + 8 share an XCD whatever blockIdx.y / .z are (the grid's x extent is a multiple of 8), so on a grid with
  // fewer bricks than XCDs the brick range of a workgroup is rotated by its Cout tile and split, and the launch spreads over
  // all XCDs; larger grids keep the Cout tiles of one brick on one XCD (they read the same input: the second one hits in L2)
  const int nbricks = p.N * p.nbz * p.nby * p.nbx;
  const int xcd = (blockIdx.x + (nbricks < 8 ? blockIdx.y + 3 * blockIdx.z : 0)) & 7, slot = blockIdx.x >> 3;
  const int sx = slot % p.bx, sy = (slot / p.bx) % p.by, sz = slot / (p.bx * p.by);
  const int b_begin = (int)((long long)xcd * nbricks / 8), b_end = (int)((long long)(xcd + 1) * nbricks / 8);
  // split-K: this workgroup's share of the input channels (chunks chunk0 .. chunk0 + nchunk - 1 of every tile)
  const int nchunk_all = p.Cin / (16 * S16);
  const int nchunk = nchunk_all / p.ksplit;
  const int chunk0 = (int)blockIdx.z * nchunk;
  const int nsteps = (b_end - b_begin) * nchunk;
  if (nsteps <= 0) return;
  const long long plane_stride = (long long)p.D * p.H * p.W * 8;
  const u32x4* wsrc = (const u32x4*)p.wpk + ((long long)cob * nchunk_all + chunk0) * (WSTEP / 16);

  auto tile_origin = [&](int k, int& n_img, int& z0, int& y0, int& x0) -> bool {
    int b = b_begin + k;
    const int Bx = b % p.nbx; b /= p.nbx;
    const int By = b % p.nby; b /= p.nby;
    const int Bz = b % p.nbz; n_img = b / p.nbz;
    const int tz = Bz * p.bz + sz, ty = By * p.by + sy, tx = Bx * p.bx + sx;
    z0 = tz * TZ; y0 = ty * TY; x0 = tx * TX;
    return tz < p.tilesZ && ty < p.tilesY && tx < p.tilesX;
  };

  if (WS) {     // all weights of this Cout tile: global -> LDS once, by everybody
    const int nitems = nchunk * (WSTEP / 16);
    for (int i = tid; i < nitems; i += NCW * 64 + NLT) *(u32x4*)(smem + OFF_W + i * 16) = wsrc[i];
  }

  if (wave >= NCW) {
    // ================================================================== loader waves
    const int lt = tid - NCW * 64;
    int pcoord[AIT];
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      const int pix = min(lt + it * NLT, NPIX - 1);
      const int px = pix % PX, t2 = pix / PX;
      pcoord[it] = px | ((t2 % PY) << 8) | ((t2 / PY) << 16);
    }
    struct Staged { u32x4 a[AIT][CP]; unsigned ok; };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int lw = __builtin_amdgcn_readfirstlane(lt >> 6);
    auto dma_weights = [&](int s, int buf) {           // streamed weights: global -> LDS directly (LDS-DMA)
      const int chunk = s - (s / nchunk) * nchunk;
      const u32x4* ws = wsrc + (long long)chunk * (WSTEP / 16);
#pragma unroll
      for (int it = 0; it < WIT; ++it) {
        const int base = it * NLT + lw * 64;
        if (base < WSTEP / 16) {
          const u32x4* gsrc = ws + min(base + (lt & 63), WSTEP / 16 - 1);
          const unsigned dst = lds0 + OFF_W + buf * WSTEP + base * 16;
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
        }
      }
    };
    auto load = [&](int s, Staged& r) {
      const int chunk = s - (s / nchunk) * nchunk;
      int n_img, z0, y0, x0;
      tile_origin(s / nchunk, n_img, z0, y0, x0);
      const T* xc = (const T*)p.x + (long long)n_img * p.x_sstride + (long long)(chunk0 + chunk) * CP * plane_stride;
      r.ok = 0;
#pragma unroll
      for (int it = 0; it < AIT; ++it) {
        const int px = pcoord[it] & 255, py = (pcoord[it] >> 8) & 255, pz = pcoord[it] >> 16;
        const int gz = z0 + pz - PADZ, gy = y0 + py - 1, gx = x0 + px - 1;
        const bool ok = (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        const int cz = min(max(gz, 0), p.D - 1), cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
        const long long goff = (((long long)cz * p.H + cy) * p.W + cx) * 8;
#pragma unroll
        for (int k = 0; k < CP; ++k) r.a[it][k] = *(const u32x4*)(xc + k * plane_stride + goff);
        r.ok |= ok ? (1u << it) : 0u;
      }
    };
    // registers -> LDS buffer s & 1: the two 8-channel planes of a 16-channel sub-chunk become one 16-byte e4m3 granule
    auto commit = [&](int s, const Staged& r) {
      unsigned char* ab = smem + (s & 1) * ABUF;
#pragma unroll
      for (int h = 0; h < S16; ++h)
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
          const int pix = lt + it * NLT;
          u32x4 v0 = r.a[it][2 * h], v1 = r.a[it][2 * h + 1];
          asm volatile("" : "+v"(v0), "+v"(v1));     // the loads are waited for on EVERY path (see conv3_v4.hip)
          if (pix < NPIX) {
            const bool ok = (r.ok >> it) & 1u;
            unsigned o0, o1, o2, o3;
            e4m3_pack8<T>(v0, o0, o1);
            e4m3_pack8<T>(v1, o2, o3);
            *(u32x4*)(ab + h * PLANE + pix * 16) = ok ? u32x4{o0, o1, o2, o3} : u32x4{0u, 0u, 0u, 0u};
          }
        }
    };
    const int last = nsteps - 1;
    Staged r;
    if (!WS) dma_weights(0, 0);
    load(0, r);
    commit(0, r);
    load(min(1, last), r);
    if (!WS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (WS) {
      for (int s = 0; s < nsteps; ++s) {
        commit(s + 1, r);
        load(min(s + 2, last), r);
        lds_barrier();
      }
    } else {
      Staged r2;
      int s = 0;
      for (; s + 1 < nsteps; s += 2) {
        dma_weights(s + 1, 1);
        load(min(s + 2, last), r2);
        commit(s + 1, r);
        lds_barrier();
        dma_weights(min(s + 2, last), 0);
        load(min(s + 3, last), r);
        commit(s + 2, r2);
        lds_barrier();
      }
      if (s < nsteps) lds_barrier();
    }
    return;
  }

  // ==================================================================== consumer waves
  const int l15 = lane & 15, q = lane >> 4;
  int col_off[NCMB];
#pragma unroll
  for (int c = 0; c < NCMB; ++c) {
    const int col = min(2 * c + (q >> 1), NCOL - 1);           // the missing partner re-reads a valid column (zero weights)
    const int dz = ND == 3 ? col / 3 : 0, dx = ND == 3 ? col % 3 : col;
    col_off[c] = (dz * PY * PX + dx) * 16;
  }
  const int f0 = wave * NI;
  const int row_first = f0 / FX;
  const int rbase = (q & 1) * 8 + ((((row_first / TY) * PY + (row_first % TY)) * PX) + l15) * 16;
  float bias_r[8], ws_r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bias_r[j] = (p.epi != 0) ? p.bias[cob * 32 + 8 * q + j] : 0.f;
    ws_r[j] = p.wscale[cob * 32 + 8 * q + j];
  }

  f32x4 acc[2][NI];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NI; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  lds_barrier();                                             // step 0 (and the resident weights) are in LDS

  i64 R[2][FX][NR + 2], A[2][3][2];
  auto step_ptrs = [&](int s, const unsigned char*& ab, const unsigned char*& wl) {
    const int chunk = s - (s / nchunk) * nchunk;
    ab = smem + (s & 1) * ABUF + rbase;
    wl = smem + OFF_W + (WS ? chunk : (s & 1)) * WSTEP + lane * 8;
  };
  auto load_group = [&](const unsigned char* ab, const unsigned char* wl, int g, auto BUF) {
    constexpr int b = decltype(BUF)::value;
    const int h = g / NCMB, c = g - h * NCMB;
#pragma unroll
    for (int xh = 0; xh < FX; ++xh)
#pragma unroll
      for (int r = 0; r < NR + 2; ++r) R[b][xh][r] = *(const i64*)(ab + h * PLANE + (r * PX + xh * 16) * 16 + col_off[c]);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      A[b][dy][0] = *(const i64*)(wl + h * WBYTES + ((c * 3 + dy) * 2 + 0) * 512);
      A[b][dy][1] = *(const i64*)(wl + h * WBYTES + ((c * 3 + dy) * 2 + 1) * 512);
    }
  };
  auto tile_epilogue = [&](int s) {
    const int chunk = s - (s / nchunk) * nchunk;
    if (chunk == nchunk - 1) {
      int n_img, z0, y0, x0;
      tile_origin(s / nchunk, n_img, z0, y0, x0);
      T* yout = (T*)p.y + (long long)n_img * p.y_sstride;
#pragma unroll
      for (int n = 0; n < NI; ++n) {
        const int f = f0 + n, row = f / FX;
        const int gz = z0 + (ND == 3 ? row / TY : 0), gy = y0 + row % TY, gx = x0 + (f % FX) * 16 + l15;
        const bool ok = gz < p.D && gy < p.H && gx < p.W;
        const long long vo = (((long long)gz * p.H + gy) * p.W + gx) * 8;
        if (p.ksplit > 1) {          // raw fp32 partial sums; scale, bias and activation happen in the reduction
          float* po = p.partial + ((long long)((int)blockIdx.z * p.N + n_img) * (p.Cout / 8) + cob * 4 + q) * plane_stride + vo;
          if (ok) { *(f32x4*)po = acc[0][n]; *(f32x4*)(po + 4) = acc[1][n]; }
        } else {
          V8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float a = j < 4 ? acc[0][n][j] : acc[1][n][j - 4];
            float r = __fmul_rn(a, ws_r[j]) + bias_r[j];          // the scale is a power of two: the product is exact
            if (p.epi == 2) r = fmaxf(r, 0.f);
            o[j] = from_f32<T>(r);
          }
          if (ok) *(V8*)(yout + (long long)(cob * 4 + q) * plane_stride + vo) = o;
        }
        acc[0][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[1][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto group_mfmas = [&](auto BUF, bool reads_pending) {
    constexpr int b = decltype(BUF)::value;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int n = 0; n < NI; ++n) {
        acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(A[b][dy][0], R[b][n % FX][n / FX + dy], acc[0][n], 0, 0, 0);
        acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(A[b][dy][1], R[b][n % FX][n / FX + dy], acc[1][n], 0, 0, 0);
      }
    if (reads_pending) {
      constexpr int MPR = (3 * NI * 2) / NRD > 0 ? (3 * NI * 2) / NRD : 1;
#pragma unroll
      for (int i = 0; i < NRD; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // one LDS read
        __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);     // MPR MFMAs
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // cross-step pipeline (conv3_v4.hip): the barrier sits in front of the step's last group, behind which the first
  // fragments of the next step are read
  auto step_groups = [&](int s, auto PAR) {
    constexpr int par = decltype(PAR)::value;
    using B0 = std::integral_constant<int, par>;
    using B1 = std::integral_constant<int, par ^ 1>;
    using BL = std::integral_constant<int, (NGRP - 1 + par) & 1>;
    using BN = std::integral_constant<int, (NGRP + par) & 1>;
    const unsigned char *ab, *wl, *abn, *wln;
    step_ptrs(s, ab, wl);
    step_ptrs(min(s + 1, nsteps - 1), abn, wln);
#pragma unroll
    for (int g = 0; g + 1 < NGRP; ++g) {
      if ((g & 1) == 0) { load_group(ab, wl, g + 1, B1{}); group_mfmas(B0{}, true); }
      else              { load_group(ab, wl, g + 1, B0{}); group_mfmas(B1{}, true); }
    }
    lds_barrier();
    load_group(abn, wln, 0, BN{});
    group_mfmas(BL{}, true);
    tile_epilogue(s);
  };
  const unsigned char *ab0, *wl0;
  step_ptrs(0, ab0, wl0);
  load_group(ab0, wl0, 0, std::integral_constant<int, 0>{});
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (NGRP % 2 == 1) {
    for (int s = 0; s < nsteps; s += 2) {
      step_groups(s, std::integral_constant<int, 0>{});
      if (s + 1 < nsteps) step_groups(s + 1, std::integral_constant<int, 1>{});
    }
  } else {
    for (int s = 0; s < nsteps; ++s) step_groups(s, std::integral_constant<int, 0>{});
  }
}

template <typename T, int ND, bool WS, bool SMALL>
int launch_f8(ConvF8Params p, hipStream_t stream) {
  using TL = F8Tile<ND, SMALL>;
  constexpr int NPIX = (TL::TZ + 2 * TL::PADZ) * (TL::TY + 2) * (TL::TX + 2);
  constexpr int PLANE = ((NPIX * 16 + 255) / 256) * 256;
  constexpr int WSTEP = TL::S16 * ((TL::NCOL + 1) / 2) * 3 * 2 * 512;
  const int lds = 2 * TL::S16 * PLANE + (WS ? p.Cin / p.ksplit / (16 * TL::S16) : 2) * WSTEP;
  IUNET_SET_MAX_LDS((conv3_f8_kernel<T, ND, WS, SMALL>), lds);
  p.tilesZ = (p.D + TL::TZ - 1) / TL::TZ; p.tilesY = (p.H + TL::TY - 1) / TL::TY; p.tilesX = (p.W + TL::TX - 1) / TL::TX;
  const int ncob = p.Cout / 32;
  iunet_brick_shape(ND, ncob, p.tilesZ, p.tilesY, p.tilesX, &p.bz, &p.by, &p.bx);
  p.nbz = (p.tilesZ + p.bz - 1) / p.bz; p.nby = (p.tilesY + p.by - 1) / p.by; p.nbx = (p.tilesX + p.bx - 1) / p.bx;
  const int gx = 8 * p.bz * p.by * p.bx;
  hipLaunchKernelGGL((conv3_f8_kernel<T, ND, WS, SMALL>), dim3(gx, ncob, p.ksplit), dim3(TL::NCW * 64 + (WS ? 256 : 512)), lds, stream, p);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

// fp32 [Cout][Cin][taps] -> e4m3 bytes in the K16 order + the per-output-channel scales.  One workgroup per output
// channel: max |folded weight| -> scale 2^k (k minimal with max / 2^k <= 448), then every element of that channel.
__device__ __forceinline__ float f8_round_e4m3(float x) {       // pack_batch.hip: round_e4m3
  const float a = fabsf(x);
  int e;
  (void)frexpf(a, &e);
  const int fl = (a == 0.f || e - 1 < -6) ? -6 : e - 1;
  const float step = ldexpf(1.0f, fl - 3);
  return copysignf(rintf(a / step) * step, x);
}
__device__ __forceinline__ unsigned char f8_encode_e4m3(float v) {   // v is an e4m3 value: sign | 4 exponent bits (bias 7) | 3 mantissa bits
  const float a = fabsf(v);
  unsigned char s = v < 0.f || (v == 0.f && __builtin_signbit(v)) ? 0x80 : 0;
  if (a == 0.f) return s;
  int e;
  const float m = frexpf(a, &e);                                  // a = m 2^e, m in [0.5, 1)
  if (e - 1 < -6) return s | (unsigned char)(int)ldexpf(a, 9);    // subnormal: a / 2^-9
  return s | (unsigned char)(((e - 1 + 7) << 3) | ((int)ldexpf(m, 4) - 8));
}

__global__ __launch_bounds__(256) void pack_f8_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ mean,
                                                      const float* __restrict__ var, float eps, unsigned char* __restrict__ dst,
                                                      float* __restrict__ wscale, float* __restrict__ bias_out, int Cout, int Cin,
                                                      int taps, int k128) {
#pragma clang fp contract(off)
  const int co = blockIdx.x;
  const float fs = gamma ? gamma[co] / sqrtf(var[co] + eps) : 1.0f;
  const float* wc = w + (long long)co * Cin * taps;
  __shared__ float red[256];
  float m = 0.f;
  for (int i = threadIdx.x; i < Cin * taps; i += 256) m = fmaxf(m, fabsf(wc[i] * fs));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
  float sc = 1.0f;
  if (red[0] > 0.f) { int e; const float mm = frexpf(red[0] / 448.0f, &e); sc = ldexpf(1.0f, mm == 0.5f ? e - 1 : e); }
  if (threadIdx.x == 0) {
    wscale[co] = sc;
    if (bias_out) bias_out[co] = gamma ? beta[co] - mean[co] * fs : 0.f;
  }
  // elements of this output channel in the K16 order [cob32][chunk16][column pair][dy][2][64][8]: one thread = the 8 bytes of
  // one lane slot (8 consecutive input channels of one tap), written as one 8-byte store
  const int ncol = taps / 3, ncmb = (ncol + 1) / 2, nchunk = Cin >> 4;
  const int cob = co >> 5, r32 = co & 31;
  const int mt = (r32 >> 2) & 1, row = (r32 >> 3) * 4 + (r32 & 3);       // co = cob*32 + 8 (row >> 2) + 4 m + (row & 3)
  for (int i = threadIdx.x; i < nchunk * ncmb * 3 * 4; i += 256) {       // (chunk, c, dy, qq)
    int r = i;
    const int qq = r & 3; r >>= 2;
    const int dy = r % 3; r /= 3;
    const int c = r % ncmb;
    const int chunk = r / ncmb;
    const int col = 2 * c + (qq >> 1);
    unsigned long long pk = 0;
    if (col < ncol) {
      const int tap = ((col / 3) * 3 + dy) * 3 + (col % 3);
      const int ci0 = chunk * 16 + 8 * (qq & 1);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        pk |= (unsigned long long)f8_encode_e4m3(f8_round_e4m3((wc[(ci0 + j) * taps + tap] * fs) / sc)) << (8 * j);
    }
    const int lane = qq * 16 + row;
    long long o = ((((((long long)cob * nchunk + chunk) * ncmb + c) * 3 + dy) * 2 + mt) * 64 + lane) * 8;
    if (k128) {                                      // conv3_f8k.hip's order: blocks of 32 input channels, no padded column
      if (col >= ncol) continue;
      o = ((long long)cob * (nchunk >> 1) + (chunk >> 1)) * F8K_WSTEP + f8k_offset(col, dy, mt, chunk & 1, qq & 1, row);
    }
    *(unsigned long long*)(dst + o) = pk;
  }
}

// split-K reduction: y = epilogue(wscale[c] * sum over the splits + bias[c]), fixed order; one thread per voxel and
// 8-channel plane
template <typename T>
__global__ __launch_bounds__(256) void f8_splitk_reduce_kernel(const float* __restrict__ partial, int ksplit, T* __restrict__ y,
                                                               long long y_ss, const float* __restrict__ wscale,
                                                               const float* __restrict__ bias, int N, int planes, long long vox,
                                                               int epi, int out8) {
  using V8 = typename Vec8<T>::type;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= vox) return;
  const int pl = blockIdx.y, n = blockIdx.z;
  float a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = 0.f;
  for (int z = 0; z < ksplit; ++z) {
    const float* pp = partial + (((long long)(z * N + n) * planes + pl) * vox + i) * 8;
    const f32x4 v0 = *(const f32x4*)pp, v1 = *(const f32x4*)(pp + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] += v0[j]; a[4 + j] += v1[j]; }
  }
  V8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float r = __fmul_rn(a[j], wscale[pl * 8 + j]) + (epi != 0 ? bias[pl * 8 + j] : 0.f);
    if (epi == 2) r = fmaxf(r, 0.f);
    o[j] = from_f32<T>(r);
  }
  if (!out8) { *(V8*)(y + n * y_ss + ((long long)pl * vox + i) * 8) = o; return; }
  // e4m3 planes [Cout / 16][voxels][16 B], y_ss in bytes: the 16-bit result rounded once more (conv3_f8k.hip's epilogue)
  unsigned o0, o1;
  e4m3_pack8<T>(__builtin_bit_cast(u32x4, o), o0, o1);
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  *(u32x2*)((unsigned char*)y + n * y_ss + ((long long)(pl >> 1) * vox + i) * 16 + (pl & 1) * 8) = u32x2{o0, o1};
}

}  // namespace

// Split-K policy.  A layer whose (voxel tiles x Cout tiles) leave most CUs idle and whose input channels make a long serial
// loop per workgroup (C5's 16^3 and 8^3 levels: 64 .. 128 workgroups of 32 .. 64 steps each, 90 .. 470 TFLOP/s) is cut along
// Cin into `ksplit` shares, each its own workgroup: the chip fills, the loop shortens, the shares' filters fit in LDS.
int iunet_conv3_f8_ksplit(int nd, int N, int D, int H, int W, int Cin, int Cout) {
  static const int forced = getenv("IUNET_F8_KSPLIT") ? atoi(getenv("IUNET_F8_KSPLIT")) : 0;
  // PER-SAMPLE tiles: the split changes the order of the fp32 sums, and a layer must keep one summation order whatever the number
  // of blocks in a launch (predict.BLOCK_BATCH with a tail of 1, per-rank block runs: shard.py's byte identity across world sizes)
  (void)N;
  const long long tiles = nd == 3 ? (long long)((D + 3) / 4) * ((H + 7) / 8) * ((W + 15) / 16)
                                  : (long long)((H + 15) / 16) * ((W + 31) / 32);
  const long long tasks = tiles * (Cout / 32);                   // (tile, Cout tile) pairs of one sample: one workgroup each
  const int nchunk = Cin / ((nd == 3 && !iunet_f8_k128(27, Cin)) ? 16 : 32);      // steps of the Cin loop
  // measured on C5's 16^3 level (tools/bench_conv.py --f8 1, IUNET_F8_KSPLIT sweep): two shares win from Cin = 512 on
  // (512 -> 512: 69 -> 56 us, 1024 -> 512: 130 -> 88 us); four are slower again (more partial sums than they save), and
  // shorter loops (Cin <= 256) lose to the reduction pass
  int ks = forced > 0 ? forced : ((tasks < 256 && Cin >= 512) ? 2 : 1);
  while (ks > 1 && (nchunk % ks != 0)) ks >>= 1;
  return ks;
}

long long iunet_conv3_f8_workspace_floats(int nd, int N, int D, int H, int W, int Cin, int Cout) {
  const int ks = iunet_conv3_f8_ksplit(nd, N, D, H, W, Cin, Cout);
  return ks > 1 ? (long long)ks * N * Cout * D * H * W : 0;
}

long long iunet_f8_pack_bytes(int Cout, int Cin, int taps) {
  return (long long)Cout * Cin * ((taps / 3 + 1) / 2) * 6;       // K16 order pads the filter columns to pairs; one byte each
}

int iunet_f8_pack_launch(const float* w, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                         void* dst, float* wscale, float* bias_out, int Cout, int Cin, int taps, hipStream_t stream) {
  hipLaunchKernelGGL(pack_f8_kernel, dim3(Cout), dim3(256), 0, stream, w, gamma, beta, mean, var, eps, (unsigned char*)dst, wscale,
                     bias_out, Cout, Cin, taps, iunet_f8_k128(taps, Cin));
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

// x_fmt / y_fmt: 0 = 16-bit NHWC8c planes (strides in elements), 1 = e4m3 planes [C / 16][D][H][W][16 B] (strides in bytes; the K = 128
// path only: 3-D, iunet_f8_k128).  An e4m3 output is the 16-bit result rounded once more -- bit for bit what a consumer conv's loader
// makes of the 16-bit tensor -- so a chain of convs keeps its values whichever format the tensors between them have.
int iunet_conv3_f8_launch_q(int dtype, int nd, const void* x, long long x_sstride, int x_fmt, void* y, long long y_sstride, int y_fmt,
                            const void* wpk, const float* wscale, const float* bias, int N, int D, int H, int W, int Cin, int Cout,
                            int epi, float* workspace, hipStream_t stream) {
  if ((x_fmt || y_fmt) && !(nd == 3 && iunet_f8_k128(27, Cin))) {
    iunet_set_error("conv3_f8: e4m3 activation planes need the K = 128 path (3-D, Cin %% 32 == 0; got nd %d, Cin %d)", nd, Cin);
    return IUNET_ERR_UNSUPPORTED;
  }
  ConvF8Params p;
  p.ksplit = workspace ? iunet_conv3_f8_ksplit(nd, N, D, H, W, Cin, Cout) : 1;
  p.partial = workspace;
  p.x = x; p.x_sstride = x_sstride; p.y = y; p.y_sstride = y_sstride; p.wpk = wpk; p.wscale = wscale; p.bias = bias;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.epi = epi;
  p.tilesZ = p.tilesY = p.tilesX = 0;
  p.bz = p.by = p.bx = p.nbz = p.nby = p.nbx = 0;
  // resident weights when they fit beside the two activation buffers (160 KB of LDS): 3-D up to Cin = 128 (2 x 17 KB +
  // 15 KB per 16 channels), 2-D up to Cin = 256 (2 x 19.5 KB + 12 KB per 32 channels)
  const int cin_wg = Cin / p.ksplit;
  // resident weights when they fit beside the two activation buffers (160 KB of LDS: 3-D up to 128 input channels per
  // workgroup, 2-D up to 256) AND the workgroup walks at least two tiles: the resident filter is loaded before the first
  // MFMA, which a one-tile workgroup (the 16^3 level at N = 1: 21 us resident, 15 us streamed) cannot amortise
  const long long big_tiles = nd == 3 ? (long long)N * ((D + 3) / 4) * ((H + 7) / 8) * ((W + 15) / 16)
                                      : (long long)N * ((H + 15) / 16) * ((W + 31) / 32);
  const long long slots = Cout / 32 >= 256 ? 1 : 256 / (Cout / 32);      // workgroups (CUs) per Cout tile
  const bool fits = nd == 3 ? cin_wg <= 128 : cin_wg <= 256;
  const bool ws = fits && big_tiles >= 2 * slots;
  const bool small = nd == 3 && !ws && big_tiles * (Cout / 32) * p.ksplit < 128;
  if (nd == 3 && iunet_f8_k128(27, Cin)) {            // the K = 128 matrix instruction (conv3_f8k.hip): weights always stream
    const bool small_k = big_tiles * (Cout / 32) * p.ksplit < 128;
    const int rc = iunet_conv3_f8k_launch(dtype, x, x_sstride, y, y_sstride, wpk, wscale, bias, N, D, H, W, Cin, Cout, epi, p.ksplit,
                                          workspace, small_k ? 1 : 0, x_fmt, p.ksplit > 1 ? 0 : y_fmt, stream);
    if (rc != IUNET_OK || p.ksplit == 1) return rc;
    const long long voxk = (long long)D * H * W;
    dim3 gridk((unsigned)((voxk + 255) / 256), Cout / 8, N);
    if (dtype == 0) hipLaunchKernelGGL(f8_splitk_reduce_kernel<f16>, gridk, dim3(256), 0, stream, workspace, p.ksplit, (f16*)y, y_sstride, wscale, bias, N, Cout / 8, voxk, epi, y_fmt);
    else hipLaunchKernelGGL(f8_splitk_reduce_kernel<bf16>, gridk, dim3(256), 0, stream, workspace, p.ksplit, (bf16*)y, y_sstride, wscale, bias, N, Cout / 8, voxk, epi, y_fmt);
    IUNET_CHECK_HIP(hipGetLastError());
    return IUNET_OK;
  }
#define F8_GO(TT) (nd == 3 ? (ws ? launch_f8<TT, 3, true, false>(p, stream)                                          \
                                 : (small ? launch_f8<TT, 3, false, true>(p, stream) : launch_f8<TT, 3, false, false>(p, stream))) \
                           : (ws ? launch_f8<TT, 2, true, false>(p, stream) : launch_f8<TT, 2, false, false>(p, stream)))
  const int rc = dtype == 0 ? F8_GO(f16) : F8_GO(bf16);
#undef F8_GO
  if (rc != IUNET_OK || p.ksplit == 1) return rc;
  const long long vox = (long long)D * H * W;
  dim3 grid((unsigned)((vox + 255) / 256), Cout / 8, N);
  if (dtype == 0) hipLaunchKernelGGL(f8_splitk_reduce_kernel<f16>, grid, dim3(256), 0, stream, workspace, p.ksplit, (f16*)y, y_sstride, wscale, bias, N, Cout / 8, vox, epi, 0);
  else hipLaunchKernelGGL(f8_splitk_reduce_kernel<bf16>, grid, dim3(256), 0, stream, workspace, p.ksplit, (bf16*)y, y_sstride, wscale, bias, N, Cout / 8, vox, epi, 0);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

int iunet_conv3_f8_launch(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride, const void* wpk,
                          const float* wscale, const float* bias, int N, int D, int H, int W, int Cin, int Cout, int epi,
                          float* workspace, hipStream_t stream) {
  return iunet_conv3_f8_launch_q(dtype, nd, x, x_sstride, 0, y, y_sstride, 0, wpk, wscale, bias, N, D, H, W, Cin, Cout, epi, workspace, stream);
}
