// Weight gradient of the 3x3x3 convolution, wave-specialised structure (see conv3_wgrad.hip for the math and the
// transposing-LDS-read fragment scheme, conv3_v4.hip for why the roles are split).
//
// One persistent 12-wave workgroup per CU and (32 co) x (32 ci) filter block:
//   * 4 LOADER waves stage the next tile (x halo 4 x 10 x 18 voxels x 32 ci + dy 2 x 8 x 16 voxels x 32 co, 62 KB)
//     global -> registers -> the other LDS buffer; the loads of tile t + 2 are issued right after tile t + 1 has been
//     written, so they are in flight during a whole consumer tile.  Tiles are walked z-fastest, so two of the four halo z-planes of every tile were fetched by
//     this CU one step earlier (L2 hits);
//   * 8 CONSUMER waves = 2 voxel (k) halves x 4 unit waves: a unit wave owns 14 of the 54 (tap, ci half) units
//     (112 accumulator registers, as before); each k half walks 4 of the tile's 8 k-steps.  One barrier per tile.
//   At the end the two k halves are added through LDS and one fp32 slab is stored; the slab reduce is unchanged
//   (and reads half as many slabs: one workgroup per CU instead of two).
#include "common.h"
#include <cstdlib>
#include <type_traits>

#ifdef IUNET_STAMPS
// Diagnostic build only (build.sh never defines IUNET_STAMPS): cycles and real time around the consumers' tile loop, to
// read the in-kernel clock (s_memtime / s_memrealtime x 100 MHz) and the cycles per tile; tools/wgrad_clock.py reads them.
__device__ unsigned long long g_wg2_stamps[4 * 512];
extern "C" int iunet_wg2_stamps_read(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg2_stamps), sizeof(unsigned long long) * 4 * 512);
}
#endif
// 16 zero bytes: the source of the voxels outside the image when a tile goes global -> LDS without registers
__device__ __attribute__((aligned(16))) unsigned int g_wg2_zero16[4] = {0u, 0u, 0u, 0u};

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s4_ptr;

struct WgradV2Params {
  const void* x;  long long x_ss;     // conv input  (Cin/8 planes)
  const void* dy; long long dy_ss;    // output grad (Cout/8 planes)
  const float* x_scale;               // optional [Cin] pair: the conv input is relu(x_scale * x + x_shift), applied by the
  const float* x_shift;               // loader waves (the BatchNorm + ReLU of the previous conv, never materialised)
  float* slab;                        // [gridDim.x][Cout/32][Cin/32][27][32][32]
  int N, D, H, W, Cin, Cout;
  int tilesZ, tilesY, tilesX;
  int dbg;                            // profiling only (IUNET_WG2_DBG): 1 no refill after tile 0, 2 no MFMA phase, 4 register-staged loads always
};

template <typename T>
__device__ __forceinline__ typename Vec8<T>::type tr_frag_v2(unsigned addr) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)addr);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(addr + 64));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(typename Vec8<T>::type, v);
}

// DYR ("dy reuse"): the consumers' second form.  A k-step is the 16 x voxels of row y in BOTH z slices of the tile (instead of two
// consecutive rows of one slice), so the x fragment of halo row h serves the three taps dy = 0, 1, 2 (output rows h, h - 1, h - 2):
// 10 x fragments per (dz, dx, ci half) and tile instead of 24, 43 % fewer LDS fragment reads in all (the first form reads 1 KB of
// fragments per 1.6 MFMAs: 164 B/clk of the LDS's 256 beside the loaders' stores -- what held its matrix pipe at 65 %).  Work items =
// (dz, dx, ci half, co half): 36, five for waves 0-3 and four for waves 4-7 (waves w and w + 4 share a SIMD: nine items each).
template <typename T, bool DYR>
__global__ __launch_bounds__(768, 1) void conv3_wgrad_v2_kernel(WgradV2Params p) {
  constexpr int TZ = 2, TY = 8, TX = 16, TAPS = 27;
  constexpr int PY = TY + 2, PX = TX + 2;
  constexpr int ZPIX = PY * PX;                                    // 180 voxels per halo z-plane
  constexpr int NSLOT = 6;                                         // z-plane ring: 4 in use + 2 incoming
  constexpr int NVOX = TZ * TY * TX;                               // 256
  constexpr int NKS = NVOX / 32;                                   // 8 k-steps of 32 voxels
  constexpr int PLANE_X = NSLOT * ZPIX * 16 + 192;                 // 17 472 = 64 mod 256: conflict-free transposing reads
  constexpr int PLANE_Y = ((NVOX * 16 + 255) / 256) * 256 + 64;
  constexpr int OFF_Y = 4 * PLANE_X;
  constexpr int YBUF = 4 * PLANE_Y;
  constexpr int NCW = 8, NLT = 256;
  constexpr int NU = TAPS * 2, MAXU = (NU + NCW - 1) / NCW;        // 54 units, 7 per consumer wave
  constexpr int XIT = (4 * ZPIX + 63) / 64;                        // 16-byte x items per loader lane: 4 z-planes of one channel plane (12)
  constexpr int YIT = NVOX * 4 / NLT;                              // dy items per loader thread (4)
  static_assert(PLANE_X % 256 == 64, "x plane stride must be 64 mod 256");

  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cob = blockIdx.y, cib = blockIdx.z;
  const int tiles_per_sample = p.tilesZ * p.tilesY * p.tilesX;
  const int ntiles = tiles_per_sample * p.N;
  const long long plane_stride = (long long)p.D * p.H * p.W * 8;
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int t_begin = (int)((long long)lb * ntiles / gridDim.x), t_end = (int)((long long)(lb + 1) * ntiles / gridDim.x);
  const int nt = t_end - t_begin;

  // tiles are walked z-fastest: inside a z column the halo of tile k + 1 shares two of its four z-planes with tile k,
  // and those stay where they are in the LDS ring -- only the two new planes travel (x bytes per tile halved)
  auto tile_origin = [&](int k, int& n_img, int& z0, int& y0, int& x0) {
    const int tile = t_begin + k;
    n_img = tile / tiles_per_sample;
    int trem = tile - n_img * tiles_per_sample;
    const int tz_i = trem % p.tilesZ; trem /= p.tilesZ;
    const int tx_i = trem % p.tilesX, ty_i = trem / p.tilesX;
    z0 = tz_i * TZ; y0 = ty_i * TY; x0 = tx_i * TX;
  };
  auto fresh = [&](int k) { return k == 0 || (t_begin + k) % p.tilesZ == 0; };      // first tile of a z column (or of the run)

  if (wave >= NCW) {
    const unsigned* zero16 = iunet_opaque_ptr((const unsigned*)g_wg2_zero16);      // (common.h: one address computation per kernel, not one per DMA piece)
    // ================================================================== loader waves
    const int lt = tid - NCW * 64;
    if (p.x_scale == nullptr && !(p.dbg & 4)) {
      // ---- everything by LDS-DMA (launches whose input needs no arithmetic on the way; IUNET_WG2_DBG=4: the register path always).
      // Loader wave w copies channel plane w of x (3 wave instructions per halo z-plane: 180 voxels) and plane w of dy (4 instructions:
      // 256 voxels) global -> LDS without registers; voxels outside the image read 16 zero bytes.  Tile k + 1 is copied while the
      // consumers work on tile k -- into the ring slots / dy buffer they are not reading --, waited for and published by the tile's barrier.
      const int lw = __builtin_amdgcn_readfirstlane(lt >> 6), ll = lt & 63;
      int xcd[3], ycd[4];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int pix = j * 64 + ll;
        xcd[j] = pix < ZPIX ? ((pix / PX) << 8) | (pix % PX) : -1;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pix = j * 64 + ll;
        ycd[j] = (pix % TX) | (((pix / TX) % TY) << 8) | ((pix / (TX * TY)) << 16);
      }
      auto dma16 = [&](const void* gsrc, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
      };
      // halo planes pz in [pz_lo, pz_hi) of tile k -> ring slots (slot0 + pz - pz_lo ... ) given by slot_of(pz)
      auto dma_x = [&](int k, int pz_lo, int pz_hi, int slot_first) {
        int n_img, z0, y0, x0;
        tile_origin(k, n_img, z0, y0, x0);
        const T* xin = (const T*)p.x + (long long)n_img * p.x_ss + (long long)(cib * 4 + lw) * plane_stride;
        for (int pz = pz_lo; pz < pz_hi; ++pz) {
          const int slot = (slot_first + pz - pz_lo) % NSLOT;
          const int gz = z0 + pz - 1;
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            if (xcd[j] >= 0) {
              const int gy = y0 + (xcd[j] >> 8) - 1, gx = x0 + (xcd[j] & 255) - 1;
              const bool ok = (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
              const void* gsrc = ok ? (const void*)(xin + (((long long)gz * p.H + gy) * p.W + gx) * 8) : (const void*)zero16;
              dma16(gsrc, __builtin_amdgcn_readfirstlane(lds0 + lw * PLANE_X + slot * (ZPIX * 16) + j * 1024));
            }
          }
        }
      };
      auto dma_y = [&](int k) {
        int n_img, z0, y0, x0;
        tile_origin(k, n_img, z0, y0, x0);
        const T* dyin = (const T*)p.dy + (long long)n_img * p.dy_ss + (long long)(cob * 4 + lw) * plane_stride;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int gz = z0 + (ycd[j] >> 16), gy = y0 + ((ycd[j] >> 8) & 255), gx = x0 + (ycd[j] & 255);
          const bool ok = gz < p.D && gy < p.H && gx < p.W;
          const void* gsrc = ok ? (const void*)(dyin + (((long long)gz * p.H + gy) * p.W + gx) * 8) : (const void*)zero16;
          dma16(gsrc, __builtin_amdgcn_readfirstlane(lds0 + OFF_Y + (k & 1) * YBUF + lw * PLANE_Y + j * 1024));
        }
      };
      auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
      const bool refill = !(p.dbg & 1);
      int base = 0;
      if (nt > 0) { dma_x(0, 0, 4, 0); dma_y(0); }
      landed();
      lds_barrier();
      for (int k = 0; k < nt; ++k) {
        const bool more = k + 1 < nt && refill;
        const bool fr = more && fresh(k + 1);
        if (more) {
          // incoming planes go to the two free slots base + 4, base + 5: pz 2, 3 of a continuing column, pz 0, 1 of a new one
          if (fr) dma_x(k + 1, 0, 2, base + 4); else dma_x(k + 1, 2, 4, base + 4);
          dma_y(k + 1);
        }
        landed();
        lds_barrier();                              // tile k is consumed
        if (k + 1 < nt && fresh(k + 1)) {             // (same condition as the consumers': the barrier count must match)
          if (fr) dma_x(k + 1, 2, 4, base + 6);       // pz 2, 3 of the new column into the slots tile k just released (base, base + 1)
          landed();
          lds_barrier();
          base = (base + 4) % NSLOT;
        } else {
          base = (base + 2) % NSLOT;
        }
      }
      return;
    }
    struct Staged { u32x4 x[XIT]; u32x4 y[YIT]; unsigned okx, oky; };
    // per-thread item tables, computed once: the loaders' integer arithmetic per tile is what their rate depends on
    // x: loader wave w owns channel plane w (its BatchNorm constants are then wave-uniform); item = (zi, voxel of the
    // z-plane): packed zi | py << 8 | px << 16, and its LDS offset
    const int lw = lt >> 6, ll = lt & 63;
    int xc[XIT], xl[XIT], yc[YIT], yl[YIT];
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int idx = min(ll + it * 64, 4 * ZPIX - 1);
      const int zi = idx / ZPIX, pix = idx - zi * ZPIX;
      const int py = pix / PX, px = pix - py * PX;
      xc[it] = zi | (py << 8) | (px << 16);
      xl[it] = lw * PLANE_X + pix * 16;
    }
    float xsc[8], xsh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      xsc[j] = p.x_scale ? p.x_scale[cib * 32 + lw * 8 + j] : 1.f;
      xsh[j] = p.x_scale ? p.x_shift[cib * 32 + lw * 8 + j] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < YIT; ++it) {
      const int item = lt + it * NLT;
      const int pl = item / NVOX, pix = item - pl * NVOX;
      const int px = pix % TX, t2 = pix / TX, py = t2 % TY, pz = t2 / TY;
      yc[it] = pz | (pl << 4) | (py << 8) | (px << 16);
      yl[it] = pl * PLANE_Y + pix * 16;
    }
    // tile k's new planes: all four (pz 0..3) when fresh, else pz 2, 3
    auto load = [&](int k, Staged& r) {
      int n_img, z0, y0, x0;
      tile_origin(k, n_img, z0, y0, x0);
      const bool fr = fresh(k);
      const int nitems = (fr ? 4 : 2) * ZPIX, pz0 = fr ? 0 : 2;       // per channel plane
      const T* xin = (const T*)p.x + (long long)n_img * p.x_ss + (long long)(cib * 4 + lw) * plane_stride;
      const T* dyin = (const T*)p.dy + (long long)n_img * p.dy_ss + (long long)cob * 4 * plane_stride;
      r.okx = 0; r.oky = 0;
#pragma unroll
      for (int it = 0; it < XIT; ++it) {
        if (ll + it * 64 < nitems) {
          const int zi = xc[it] & 15, py = (xc[it] >> 8) & 255, px = xc[it] >> 16;
          const int gz = z0 + pz0 + zi - 1, gy = y0 + py - 1, gx = x0 + px - 1;
          const bool ok = (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
          const int cz = min(max(gz, 0), p.D - 1), cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
          r.x[it] = *(const u32x4*)(xin + (long long)((cz * p.H + cy) * p.W + cx) * 8);
          r.okx |= ok ? (1u << it) : 0u;
        }
      }
#pragma unroll
      for (int it = 0; it < YIT; ++it) {
        const int pz = yc[it] & 15, pl = (yc[it] >> 4) & 15, py = (yc[it] >> 8) & 255, px = yc[it] >> 16;
        const int gz = z0 + pz, gy = y0 + py, gx = x0 + px;
        const bool ok = gz < p.D && gy < p.H && gx < p.W;
        const int cz = min(gz, p.D - 1), cy = min(gy, p.H - 1), cx = min(gx, p.W - 1);
        r.y[it] = *(const u32x4*)(dyin + pl * plane_stride + (long long)((cz * p.H + cy) * p.W + cx) * 8);
        r.oky |= ok ? (1u << it) : 0u;
      }
    };
    // x planes zi in [zi_lo, zi_hi) of the staged tile -> ring slots (slot0 + zi) mod 6
    auto commit_x = [&](const Staged& r, int nitems, int zi_lo, int zi_hi, int slot0) {      // nitems: per channel plane
#pragma unroll
      for (int it = 0; it < XIT; ++it) {
        const int zi = xc[it] & 15;
        if (ll + it * 64 < nitems && zi >= zi_lo && zi < zi_hi) {
          const int slot = (slot0 + zi) % NSLOT;
          u32x4 v = r.x[it];
          if (p.x_scale != nullptr) {                    // z = relu(scale * y + shift), the arithmetic of bn_relu_fwd_kernel
            const typename Vec8<T>::type in = __builtin_bit_cast(typename Vec8<T>::type, v);
            typename Vec8<T>::type o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = from_f32<T>(fmaxf(fmaf(xsc[j], to_f32<T>(in[j]), xsh[j]), 0.f));
            v = __builtin_bit_cast(u32x4, o);
          }
          *(u32x4*)(smem + xl[it] + slot * (ZPIX * 16)) = ((r.okx >> it) & 1u) ? v : u32x4{0u, 0u, 0u, 0u};
        }
      }
    };
    auto commit_y = [&](int k, const Staged& r) {
      unsigned char* b = smem + OFF_Y + (k & 1) * YBUF;
#pragma unroll
      for (int it = 0; it < YIT; ++it) *(u32x4*)(b + yl[it]) = ((r.oky >> it) & 1u) ? r.y[it] : u32x4{0u, 0u, 0u, 0u};
    };
    const bool refill = !(p.dbg & 1);
    Staged r;
    int base = 0;                                   // ring slot of halo plane pz = 0 of the tile being consumed
    if (nt > 0) { load(0, r); commit_x(r, 4 * ZPIX, 0, 4, 0); commit_y(0, r); }
    if (nt > 1 && refill) load(1, r);
    lds_barrier();
    for (int k = 0; k < nt; ++k) {
      const bool more = k + 1 < nt && refill;
      const bool fr = more && fresh(k + 1);
      if (more) {
        // incoming planes go to the two free slots base + 4, base + 5: pz 2, 3 of a continuing column, pz 0, 1 of a new one
        if (fr) commit_x(r, 4 * ZPIX, 0, 2, base + 4); else commit_x(r, 2 * ZPIX, 0, 2, base + 4);
        commit_y(k + 1, r);
      }
      lds_barrier();                              // tile k is consumed
      if (k + 1 < nt && fresh(k + 1)) {             // (same condition as the consumers': the barrier count must match)
        if (fr) commit_x(r, 4 * ZPIX, 2, 4, base + 4);      // pz 2, 3 of the new column into the slots tile k just released
        lds_barrier();
        base = (base + 4) % NSLOT;
      } else {
        base = (base + 2) % NSLOT;
      }
      if (k + 2 < nt && refill) load(k + 2, r);
    }
    return;
  }

  // ==================================================================== consumer waves
  if constexpr (DYR) {
    using V8 = typename Vec8<T>::type;
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
    const int gh = g >> 1, gl = g & 1;
    // lane part of the tr-read addresses (bytes): k-group gh = z slice (dy image: 128 voxels further; x image: the next ring slot),
    // voxel (gl * 8 + q) of the row, channels 4 pp .. 4 pp + 3
    const unsigned laneY = lds0 + OFF_Y + (pp >> 1) * PLANE_Y + (pp & 1) * 8 + (gh * (TY * TX) + gl * 8 + q) * 16;
    const unsigned laneXb = lds0 + (pp >> 1) * PLANE_X + (pp & 1) * 8 + (gl * 8 + q) * 16;
    // units (filter column c = dz * 3 + dx, ci half): 2 w and 2 w + 1 with both co halves; waves 0-3 also one co half of unit 16 + (w >> 1)
    unsigned ustat[3];
    int udz[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int u = k < 2 ? 2 * wave + k : 16 + ((wave & 3) >> 1);
      const int c = u >> 1, cih = u & 1;
      ustat[k] = (unsigned)__builtin_amdgcn_readfirstlane((c % 3) * 16 + cih * 2 * PLANE_X);
      udz[k] = __builtin_amdgcn_readfirstlane(c / 3);
    }
    const int xco = wave & 1;                                       // co half of the fifth item (waves 0-3)
    f32x4 acc[5][3];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int d = 0; d < 3; ++d) acc[i][d] = f32x4{0, 0, 0, 0};
    int base = 0;
    lds_barrier();                                                 // tile 0 is in LDS
    // one tile: halo rows h = 0 .. 9; row h multiplies the dy-gradient rows h (dy 0), h - 1 (dy 1), h - 2 (dy 2)
    auto tile_body = [&](int k, auto NITEMS) {
      constexpr int NIT = decltype(NITEMS)::value, NUN = NIT == 5 ? 3 : 2;
      unsigned xoff[NUN];
#pragma unroll
      for (int u = 0; u < NUN; ++u) {
        const unsigned s0 = (unsigned)(((base + udz[u]) % NSLOT) * ZPIX * 16), s1 = (unsigned)(((base + udz[u] + 1) % NSLOT) * ZPIX * 16);
        xoff[u] = laneXb + ustat[u] + (gh ? s1 : s0);
      }
      const unsigned baseY = laneY + (unsigned)((k & 1) * YBUF);
      auto rdA = [&](int y, int coh) { return tr_frag_v2<T>(baseY + (unsigned)(y * 16 * 16 + coh * 2 * PLANE_Y)); };
      auto rdB = [&](int u, int h) { return tr_frag_v2<T>(xoff[u] + (unsigned)(h * PX * 16)); };
      V8 Ar[4][2], Bq[2][NUN];
      Ar[0][0] = rdA(0, 0); Ar[0][1] = rdA(0, 1);
#pragma unroll
      for (int u = 0; u < NUN; ++u) Bq[0][u] = rdB(u, 0);
#pragma unroll
      for (int h = 0; h < TY + 2; ++h) {
        if (h + 1 < TY) { Ar[(h + 1) & 3][0] = rdA(h + 1, 0); Ar[(h + 1) & 3][1] = rdA(h + 1, 1); }
        if (h + 1 < TY + 2) {
#pragma unroll
          for (int u = 0; u < NUN; ++u) Bq[(h + 1) & 1][u] = rdB(u, h + 1);
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const int y = h - d;
          if (y >= 0 && y < TY) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
              const int u = i >> 1;
              const V8 a = i < 4 ? Ar[y & 3][i & 1] : (xco ? Ar[y & 3][1] : Ar[y & 3][0]);
              acc[i][d] = mfma16<T>(a, Bq[h & 1][u], acc[i][d]);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    for (int k = 0; k < nt; ++k) {
      if (!(p.dbg & 2)) {
        if (wave < 4) tile_body(k, std::integral_constant<int, 5>{}); else tile_body(k, std::integral_constant<int, 4>{});
      }
      lds_barrier();
      if (k + 1 < nt && fresh(k + 1)) {
        lds_barrier();                                             // the loaders complete the new column's first tile
        base = (base + 4) % NSLOT;
      } else {
        base = (base + 2) % NSLOT;
      }
    }
    // ---- store the slab: rows = co (coh * 16 + 4 g + j), cols = ci ----
    float* slab = p.slab + ((((long long)blockIdx.x * gridDim.y + cob) * gridDim.z + cib) * TAPS) * 1024;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (i == 4 && wave >= 4) break;
      const int u = i < 4 ? 2 * wave + (i >> 1) : 16 + ((wave & 3) >> 1);
      const int c = u >> 1, cih = u & 1, coh = i < 4 ? (i & 1) : xco;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int tap = ((c / 3) * 3 + d) * 3 + (c % 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) slab[tap * 1024 + (coh * 16 + 4 * g + j) * 32 + cih * 16 + i16] = acc[i][d][j];
      }
    }
    return;
  }
  // Unit wave uw owns units u = uw + 8 i (unit = tap x ci half: a 32 co x 16 ci block of dW) and walks ALL eight k-steps of
  // a tile, so its 7 x 2 accumulators are complete sums: no reduction between waves, and 56 accumulator registers leave
  // room for a deep fragment pipeline.  (Splitting the voxels over two wave groups instead halved the dy fragment reads
  // but needed 112 accumulator registers: at the 168-register cap the reads could not be issued ahead of their MFMAs.)
  using V8 = typename Vec8<T>::type;
  const int uw = wave;
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  const int gh = g >> 1, gl = g & 1;
  // lane part of the tr-read addresses (bytes): voxel (gl*8 + q) of fragment gh, channels 4pp..4pp+3
  const unsigned laneY = lds0 + OFF_Y + (pp >> 1) * PLANE_Y + (pp & 1) * 8 + (gh * 16 + gl * 8 + q) * 16;
  const unsigned laneX = lds0 + (pp >> 1) * PLANE_X + (pp & 1) * 8 + (gh * PX + gl * 8 + q) * 16;
  unsigned unit_off[MAXU];                                         // in-plane part: (dy, dx) shift and ci half
  int unit_dz[MAXU];
#pragma unroll
  for (int i = 0; i < MAXU; ++i) {
    const int u = uw + NCW * i;
    const int tap = (u < NU ? u : 0) >> 1, cih = u & 1;
    const int dz = tap / 9, dy_ = (tap / 3) % 3, dx = tap % 3;
    unit_off[i] = (unsigned)__builtin_amdgcn_readfirstlane((dy_ * PX + dx) * 16 + cih * 2 * PLANE_X);   // wave-uniform: scalar registers
    unit_dz[i] = __builtin_amdgcn_readfirstlane(dz);
  }
  f32x4 acc[MAXU][2];
#pragma unroll
  for (int i = 0; i < MAXU; ++i) { acc[i][0] = f32x4{0, 0, 0, 0}; acc[i][1] = f32x4{0, 0, 0, 0}; }

  int base = 0;
  lds_barrier();                                                 // tile 0 is in LDS
#ifdef IUNET_STAMPS
  const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int k = 0; k < nt; ++k) {
    // k-step ks = output z slice ks / 4, rows 2 (ks % 4), 2 (ks % 4) + 1; unit i reads halo plane pz = z slice + dz_i
    // -> ring slot (base + pz) mod 6
    unsigned uoff[2][MAXU];
#pragma unroll
    for (int kz = 0; kz < 2; ++kz)
#pragma unroll
      for (int i = 0; i < MAXU; ++i)
        uoff[kz][i] = (unsigned)__builtin_amdgcn_readfirstlane((int)unit_off[i] + ((base + kz + unit_dz[i]) % NSLOT) * ZPIX * 16);
    if (!(p.dbg & 2)) {
      // One flat, software-pipelined sequence of NKS x MAXU unit steps (2 MFMAs each): the x fragment of step t + DEPTH
      // is read right after the MFMAs of step t, the dy fragments of the next k-step at the start of the current one, so
      // every LDS read has DEPTH - 1 unit steps (64 MFMA cycles each) to land.  (Left to the compiler, each read was
      // issued directly before its two MFMAs and waited for with lgkmcnt(0): the matrix pipe was busy 29 % of the time.)
#ifndef WG2_DEPTH
#define WG2_DEPTH 5
#endif
      constexpr int NSTEP = NKS * MAXU, DEPTH = WG2_DEPTH;
      const unsigned baseY = laneY + (unsigned)((k & 1) * YBUF);
      auto rdA = [&](int ks, int h) { return tr_frag_v2<T>(baseY + (unsigned)(ks * 2 * 16 * 16 + h * 2 * PLANE_Y)); };
      auto rdB = [&](int t) {
        const int ks = t / MAXU;
        return tr_frag_v2<T>(laneX + (unsigned)(2 * (ks & 3) * PX * 16) + uoff[ks >> 2][t % MAXU]);
      };
      V8 a[2][2], bq[DEPTH];
      a[0][0] = rdA(0, 0); a[0][1] = rdA(0, 1);
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) bq[d] = rdB(d);
#pragma unroll
      for (int t = 0; t < NSTEP; ++t) {
        const int ks = t / MAXU, i = t % MAXU;
        if (i == 0 && ks + 1 < NKS) { a[(ks + 1) & 1][0] = rdA(ks + 1, 0); a[(ks + 1) & 1][1] = rdA(ks + 1, 1); }
        // branch-free: a wave whose last unit does not exist (u >= NU) recomputes tap 0 into an accumulator that is never stored
        const V8 cur = bq[t % DEPTH];
#ifdef WG2_NOMFMA                                          // ablation builds (never defined by build.sh)
        asm volatile("" :: "v"(cur), "v"(a[ks & 1][0]), "v"(a[ks & 1][1]));
#else
        acc[i][0] = mfma16<T>(a[ks & 1][0], cur, acc[i][0]);
        acc[i][1] = mfma16<T>(a[ks & 1][1], cur, acc[i][1]);
#endif
#ifndef WG2_NOREAD
        if (t + DEPTH < NSTEP) bq[t % DEPTH] = rdB(t + DEPTH);
#endif
        __builtin_amdgcn_sched_barrier(0);             // keep this order (the scheduler sinks the reads back to their MFMAs)
      }
    }
    lds_barrier();
    if (k + 1 < nt && fresh(k + 1)) {
      lds_barrier();                                             // the loaders complete the new column's first tile
      base = (base + 4) % NSLOT;
    } else {
      base = (base + 2) % NSLOT;
    }
  }

#ifdef IUNET_STAMPS
  if (wave == 0 && lane == 0 && cob == 0 && cib == 0 && blockIdx.x < 512) {
    const unsigned long long st1 = __builtin_amdgcn_s_memtime(), sr1 = __builtin_amdgcn_s_memrealtime();
    g_wg2_stamps[blockIdx.x * 4 + 0] = st1 - st0; g_wg2_stamps[blockIdx.x * 4 + 1] = sr1 - sr0; g_wg2_stamps[blockIdx.x * 4 + 2] = nt;
  }
#endif
  // ---- store the slab: rows = co (4g + j), cols = ci ----
  const int ncob = gridDim.y, ncib = gridDim.z;
  float* slab = p.slab + ((((long long)blockIdx.x * ncob + cob) * ncib + cib) * TAPS) * 1024;
#pragma unroll
  for (int i = 0; i < MAXU; ++i) {
    const int u = uw + NCW * i;
    if (u < NU) {
      const int tap = u >> 1, cih = u & 1;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) slab[tap * 1024 + (t * 16 + 4 * g + j) * 32 + cih * 16 + i16] = acc[i][t][j];
    }
  }
}

}  // namespace

int iunet_conv3_wgrad_v2_blocks(int N, int D, int H, int W, int Cin, int Cout) {
  const long long ntiles = (long long)N * ((D + 1) / 2) * ((H + 7) / 8) * ((W + 15) / 16);
  const int pairs = (Cin / 32) * (Cout / 32);
  long long nb = (256 + pairs - 1) / pairs;             // one workgroup per CU in total
  if (nb > ntiles) nb = ntiles;
  if (nb < 1) nb = 1;
  return (int)nb;
}

int iunet_conv3_wgrad_v2_launch(int dtype, const void* x, long long x_ss, const void* dy, long long dy_ss, float* slab,
                                int N, int D, int H, int W, int Cin, int Cout, const float* x_scale, const float* x_shift,
                                hipStream_t stream) {
  WgradV2Params p;
  p.x = x; p.x_ss = x_ss; p.dy = dy; p.dy_ss = dy_ss; p.slab = slab; p.x_scale = x_scale; p.x_shift = x_shift;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.tilesZ = (D + 1) / 2; p.tilesY = (H + 7) / 8; p.tilesX = (W + 15) / 16;
#ifdef IUNET_ABLATE      // result-destroying profiling switches exist in diagnostic builds only (tools/ab_build.sh <file> -DIUNET_ABLATE): ADVICE r3
  static const int dbg = getenv("IUNET_WG2_DBG") ? atoi(getenv("IUNET_WG2_DBG")) : 0;
#else
  static const int dbg = 0;
#endif
  p.dbg = dbg;
  static const int dyr = getenv("IUNET_WG2_DYR") ? atoi(getenv("IUNET_WG2_DYR")) : 1;      // A/B switch: 0 = the first consumer form
  constexpr int PLANE_X = 6 * 180 * 16 + 192, PLANE_Y = 256 * 16 + 64;
  constexpr int LDS = 4 * PLANE_X + 2 * 4 * PLANE_Y;      // 103 168 B: x z-plane ring + two dy buffers
  const int nb = iunet_conv3_wgrad_v2_blocks(N, D, H, W, Cin, Cout);
  dim3 grid(nb, Cout / 32, Cin / 32);
  if (dtype == 0) {
    if (dyr) { IUNET_SET_MAX_LDS((conv3_wgrad_v2_kernel<f16, true>), LDS); hipLaunchKernelGGL((conv3_wgrad_v2_kernel<f16, true>), grid, dim3(768), LDS, stream, p); }
    else { IUNET_SET_MAX_LDS((conv3_wgrad_v2_kernel<f16, false>), LDS); hipLaunchKernelGGL((conv3_wgrad_v2_kernel<f16, false>), grid, dim3(768), LDS, stream, p); }
  } else {
    if (dyr) { IUNET_SET_MAX_LDS((conv3_wgrad_v2_kernel<bf16, true>), LDS); hipLaunchKernelGGL((conv3_wgrad_v2_kernel<bf16, true>), grid, dim3(768), LDS, stream, p); }
    else { IUNET_SET_MAX_LDS((conv3_wgrad_v2_kernel<bf16, false>), LDS); hipLaunchKernelGGL((conv3_wgrad_v2_kernel<bf16, false>), grid, dim3(768), LDS, stream, p); }
  }
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}
