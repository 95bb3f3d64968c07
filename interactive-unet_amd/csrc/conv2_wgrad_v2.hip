// Weight gradient of the 3 x 3 convolution (2-D), wave-specialised structure: the 2-D form of conv3_wgrad_v2.hip's dy-reuse kernel
// (see conv3_wgrad.hip for the math and the transposing-LDS-read fragment scheme, conv3_v4.hip for why the roles are split).
//
//     dW[co][ci][dy][dx] = sum_{n, y, x} g[n][co][y][x] * x[n][ci][y + dy - 1][x + dx - 1]          (g = the output gradient)
//
// One persistent 12-wave workgroup per CU and filter block:
//   * 4 LOADER waves stage the next tile -- the x halo (TY + 2) x 34 pixels x 32 ci and the gradient TY x 32 pixels x BCO co -- into the
//     other LDS buffer while the consumers work: by LDS-DMA (global_load_lds_dwordx4, no registers) when the input needs no arithmetic on
//     the way, else through registers with the BatchNorm + ReLU of the producing conv applied (iunet_conv3_wgrad_act);
//   * 8 CONSUMER waves.  A k-step is one tile row of 32 pixels; the x fragment of halo row h at column shift dx serves the three filter
//     rows (gradient rows h, h - 1, h - 2): per halo row a wave reads one gradient fragment and three x fragments for nine MFMAs
//     (0.53 fragment reads per MFMA; the first form, conv3_wgrad_kernel<T, 2>: 0.7, two workgroups of 4 waves per CU with loads and
//     MFMAs in the same waves).  A wave owns a (16 co) x (16 ci) corner of the block with all three dx: 9 accumulators.
// Two block shapes:
//   * CO64 (Cout % 64 == 0): block = 64 co x 32 ci, tile 8 x 32 pixels; wave = (ci half, co quarter), every wave walks the whole tile.
//     54.5 KB per tile for 576 MFMAs -- the launch is bound by the bytes that cross the L2 -> CU fabric, and the 32 x 32 block needs
//     72 KB for the same MFMAs;
//   * 32 co x 32 ci (the level-0 layers: Cout = 32, HBM-bound): tile 16 x 32 pixels, wave = (row half, ci half, co half); the two row
//     halves write two slab rows (the slab reduce adds them with the other workgroups' rows).
// One barrier per tile, two LDS buffers.
#include "common.h"
#include <cstdlib>
#include <type_traits>

__device__ __attribute__((aligned(16))) unsigned int g_wg2d_zero16[4] = {0u, 0u, 0u, 0u};

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s4_ptr;

struct Wgrad2Params {
  const void* x;  long long x_ss;     // conv input  (Cin / 8 planes)
  const void* dy; long long dy_ss;    // output gradient (Cout / 8 planes)
  const float* x_scale;               // optional [Cin] pair: the conv input is relu(x_scale * x + x_shift), applied by the loader waves
  const float* x_shift;
  float* slab;                        // [rows][Cout / 32][Cin / 32][9][32][32]
  int N, H, W, Cin, Cout;
  int tilesY, tilesX;
};

template <typename T>
__device__ __forceinline__ typename Vec8<T>::type tr_frag_2d(unsigned addr) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)addr);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(addr + 64));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(typename Vec8<T>::type, v);
}

template <bool CO64> struct W2Tile {
  static constexpr int TY = CO64 ? 8 : 16, TX = 32, PY = TY + 2, PX = TX + 2;
  static constexpr int NPIX = PY * PX, NVOX = TY * TX;
  static constexpr int NPY = CO64 ? 8 : 4;                               // gradient planes per block
  static constexpr int PLANE_X = NPIX * 16;                              // 5 440 / 9 792: both 64 mod 256 (conflict-free transposing reads)
  static constexpr int PLANE_Y = NVOX * 16 + 64;
  static constexpr int BUF = 4 * PLANE_X + NPY * PLANE_Y;                // 55 040 / 72 192
  static_assert(PLANE_X % 256 == 64 && PLANE_Y % 256 == 64, "plane strides must be 64 mod 256");
};

template <typename T, bool CO64>
__global__ __launch_bounds__(768, 1) void conv2_wgrad_v2_kernel(Wgrad2Params p) {
  using TL = W2Tile<CO64>;
  constexpr int TY = TL::TY, TX = TL::TX, PX = TL::PX, NPIX = TL::NPIX, NVOX = TL::NVOX, NPY = TL::NPY;
  constexpr int PLANE_X = TL::PLANE_X, PLANE_Y = TL::PLANE_Y, BUF = TL::BUF, OFF_Y = 4 * PLANE_X;
  constexpr int NCW = 8, NLT = 256;
  constexpr int XIT = (NPIX + 63) / 64;                                  // 16-byte x items per loader lane: one channel plane per wave (6 / 10)
  constexpr int YPW = NPY / 4;                                           // gradient planes per loader wave (2 / 1)
  constexpr int YIT = YPW * NVOX / 64;                                   // gradient items per loader lane (8 / 8)
  using V8 = typename Vec8<T>::type;

  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cobk = blockIdx.y, cib = blockIdx.z;                         // block of 64 (CO64) or 32 output channels; 32 input channels
  const int tiles_per_sample = p.tilesY * p.tilesX;
  const int ntiles = tiles_per_sample * p.N;
  const long long plane_stride = (long long)p.H * p.W * 8;
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  const int t_begin = (int)((long long)lb * ntiles / gridDim.x), t_end = (int)((long long)(lb + 1) * ntiles / gridDim.x);
  const int nt = t_end - t_begin;

  auto tile_origin = [&](int k, int& n_img, int& y0, int& x0) {
    const int tile = t_begin + k;
    n_img = tile / tiles_per_sample;
    const int trem = tile - n_img * tiles_per_sample;
    const int ty_i = trem / p.tilesX, tx_i = trem - ty_i * p.tilesX;
    y0 = ty_i * TY; x0 = tx_i * TX;
  };

  if (wave >= NCW) {
    const unsigned* zero16 = iunet_opaque_ptr((const unsigned*)g_wg2d_zero16);      // (common.h: one address computation per kernel, not one per DMA piece)
    // ================================================================== loader waves: wave w owns channel plane w of x and YPW planes of the gradient
    const int lt = tid - NCW * 64;
    const int lw = __builtin_amdgcn_readfirstlane(lt >> 6), ll = lt & 63;
    int xc[XIT];                                                         // halo pixel of item it: py << 8 | px, -1 past the end
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int pix = it * 64 + ll;
      xc[it] = pix < NPIX ? ((pix / PX) << 8) | (pix % PX) : -1;
    }
    if (p.x_scale == nullptr) {
      // ---- everything by LDS-DMA: 64 consecutive pixels of one plane per wave instruction; pixels outside the image read 16 zero bytes
      auto dma16 = [&](const void* gsrc, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
      };
      auto dma_tile = [&](int k) {
        int n_img, y0, x0;
        tile_origin(k, n_img, y0, x0);
        const unsigned buf = lds0 + (unsigned)((k & 1) * BUF);
        const T* xin = (const T*)p.x + (long long)n_img * p.x_ss + (long long)(cib * 4 + lw) * plane_stride;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
          if (xc[it] >= 0) {                                              // (the last instruction of a plane is partial: lanes past the halo skip)
            const int gy = y0 + (xc[it] >> 8) - 1, gx = x0 + (xc[it] & 255) - 1;
            const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            const void* gsrc = ok ? (const void*)(xin + ((long long)gy * p.W + gx) * 8) : (const void*)zero16;
            dma16(gsrc, __builtin_amdgcn_readfirstlane(buf + lw * PLANE_X + it * 1024));
          }
        }
#pragma unroll
        for (int it = 0; it < YIT; ++it) {
          const int pl = lw * YPW + it / (NVOX / 64), pix = (it % (NVOX / 64)) * 64 + ll;
          const int gy = y0 + pix / TX, gx = x0 + pix % TX;
          const bool ok = gy < p.H && gx < p.W;
          const T* dyin = (const T*)p.dy + (long long)n_img * p.dy_ss + (long long)(cobk * NPY + pl) * plane_stride;
          const void* gsrc = ok ? (const void*)(dyin + ((long long)gy * p.W + gx) * 8) : (const void*)zero16;
          dma16(gsrc, __builtin_amdgcn_readfirstlane(buf + OFF_Y + pl * PLANE_Y + (it % (NVOX / 64)) * 1024));
        }
      };
      auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
      if (nt > 0) dma_tile(0);
      landed();
      lds_barrier();
      for (int k = 0; k < nt; ++k) {
        if (k + 1 < nt) dma_tile(k + 1);
        landed();
        lds_barrier();                                                    // tile k is consumed, tile k + 1 is in LDS
      }
      return;
    }
    // ---- through registers: the loads of tile k + 1 are in flight during tile k, committed (with the fused activation) before its barrier
    float xsc[8], xsh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { xsc[j] = p.x_scale[cib * 32 + lw * 8 + j]; xsh[j] = p.x_shift[cib * 32 + lw * 8 + j]; }
    struct Staged { u32x4 x[XIT]; u32x4 y[YIT]; unsigned okx, oky; };
    auto load = [&](int k, Staged& r) {
      int n_img, y0, x0;
      tile_origin(k, n_img, y0, x0);
      const T* xin = (const T*)p.x + (long long)n_img * p.x_ss + (long long)(cib * 4 + lw) * plane_stride;
      r.okx = 0; r.oky = 0;
#pragma unroll
      for (int it = 0; it < XIT; ++it) {
        if (xc[it] >= 0) {
          const int gy = y0 + (xc[it] >> 8) - 1, gx = x0 + (xc[it] & 255) - 1;
          const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
          const int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
          r.x[it] = *(const u32x4*)(xin + ((long long)cy * p.W + cx) * 8);
          r.okx |= ok ? (1u << it) : 0u;
        }
      }
#pragma unroll
      for (int it = 0; it < YIT; ++it) {
        const int pl = lw * YPW + it / (NVOX / 64), pix = (it % (NVOX / 64)) * 64 + ll;
        const int gy = y0 + pix / TX, gx = x0 + pix % TX;
        const bool ok = gy < p.H && gx < p.W;
        const int cy = min(gy, p.H - 1), cx = min(gx, p.W - 1);
        const T* dyin = (const T*)p.dy + (long long)n_img * p.dy_ss + (long long)(cobk * NPY + pl) * plane_stride;
        r.y[it] = *(const u32x4*)(dyin + ((long long)cy * p.W + cx) * 8);
        r.oky |= ok ? (1u << it) : 0u;
      }
    };
    auto commit = [&](int k, const Staged& r) {
      unsigned char* buf = smem + (k & 1) * BUF;
#pragma unroll
      for (int it = 0; it < XIT; ++it) {
        if (xc[it] >= 0) {
          const typename Vec8<T>::type in = __builtin_bit_cast(typename Vec8<T>::type, r.x[it]);
          typename Vec8<T>::type o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = from_f32<T>(fmaxf(fmaf(xsc[j], to_f32<T>(in[j]), xsh[j]), 0.f));      // z = relu(scale * y + shift), as bn_relu_fwd_kernel
          *(u32x4*)(buf + lw * PLANE_X + (it * 64 + ll) * 16) = ((r.okx >> it) & 1u) ? __builtin_bit_cast(u32x4, o) : u32x4{0u, 0u, 0u, 0u};
        }
      }
#pragma unroll
      for (int it = 0; it < YIT; ++it) {
        const int pl = lw * YPW + it / (NVOX / 64), pix = (it % (NVOX / 64)) * 64 + ll;
        *(u32x4*)(buf + OFF_Y + pl * PLANE_Y + pix * 16) = ((r.oky >> it) & 1u) ? r.y[it] : u32x4{0u, 0u, 0u, 0u};
      }
    };
    Staged r;
    if (nt > 0) { load(0, r); commit(0, r); }
    if (nt > 1) load(1, r);
    lds_barrier();
    for (int k = 0; k < nt; ++k) {
      if (k + 1 < nt) commit(k + 1, r);
      if (k + 2 < nt) load(k + 2, r);                                     // in flight during the whole of tile k + 1
      lds_barrier();
    }
    return;
  }

  // ==================================================================== consumer waves
  const int g = lane >> 4, i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
  const int gh = g >> 1, gl = g & 1;
  // wave -> (first row of its 8 rows, ci half, co group of 16)
  const int r0 = CO64 ? 0 : (wave >> 2) * 8;
  const int cih = wave & 1, cog = CO64 ? (wave >> 1) : ((wave >> 1) & 1);
  // lane part of the tr-read addresses (bytes): pixel gh * 16 + gl * 8 + q of the row, channels 4 pp .. 4 pp + 3 of the 16-channel group
  const unsigned laneY = lds0 + OFF_Y + (cog * 2 + (pp >> 1)) * PLANE_Y + (pp & 1) * 8 + (r0 * TX + gh * 16 + gl * 8 + qq) * 16;
  const unsigned laneX = lds0 + (cih * 2 + (pp >> 1)) * PLANE_X + (pp & 1) * 8 + (r0 * PX + gh * 16 + gl * 8 + qq) * 16;
  f32x4 acc[3][3];                                                       // [dx][filter row]
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int d = 0; d < 3; ++d) acc[a][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  lds_barrier();                                                         // tile 0 is in LDS
  for (int k = 0; k < nt; ++k) {
    const unsigned bofs = (unsigned)((k & 1) * BUF);
    auto rdA = [&](int y) { return tr_frag_2d<T>(laneY + bofs + (unsigned)(y * TX * 16)); };
    auto rdB = [&](int dx, int h) { return tr_frag_2d<T>(laneX + bofs + (unsigned)((h * PX + dx) * 16)); };
    // halo rows h = 0 .. 9 of this wave's 8 rows; row h multiplies the gradient rows h (filter row 0), h - 1 (row 1), h - 2 (row 2)
    V8 Ar[4], Bq[2][3];
    Ar[0] = rdA(0);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) Bq[0][dx] = rdB(dx, 0);
#pragma unroll
    for (int h = 0; h < 10; ++h) {
      if (h + 1 < 8) Ar[(h + 1) & 3] = rdA(h + 1);
      if (h + 1 < 10) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) Bq[(h + 1) & 1][dx] = rdB(dx, h + 1);
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int y = h - d;
        if (y >= 0 && y < 8) {
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) acc[dx][d] = mfma16<T>(Ar[y & 3], Bq[h & 1][dx], acc[dx][d]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);                                 // keep this order (the scheduler sinks the reads back to their MFMAs)
    }
    lds_barrier();
  }
  // ---- store the slab: rows = co (4 g + j of this wave's 16), cols = ci (cih * 16 + i16)
  const int ncob = p.Cout / 32, ncib = gridDim.z;
  const int cob = CO64 ? cobk * 2 + (cog >> 1) : cobk, coh = CO64 ? (cog & 1) : cog;
  const long long row = CO64 ? (long long)blockIdx.x : (long long)blockIdx.x * 2 + (wave >> 2);
  float* slab = p.slab + (((row * ncob + cob) * ncib + cib) * 9) * 1024;
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int tap = d * 3 + dx;
#pragma unroll
      for (int j = 0; j < 4; ++j) slab[tap * 1024 + (coh * 16 + 4 * g + j) * 32 + cih * 16 + i16] = acc[dx][d][j];
    }
}

inline bool wg2_co64(int Cout) {
  static const int force32 = getenv("IUNET_WG2D_CO32") != nullptr;      // A/B switch: the 32 x 32 block everywhere
  return Cout % 64 == 0 && !force32;
}

}  // namespace

// workgroups along the tile axis and slab rows they write
int iunet_conv2_wgrad_v2_blocks(int N, int H, int W, int Cin, int Cout, int* rows) {
  const bool co64 = wg2_co64(Cout);
  const int ty = co64 ? 8 : 16;
  const long long ntiles = (long long)N * ((H + ty - 1) / ty) * ((W + 31) / 32);
  const int blocks = (Cin / 32) * (Cout / (co64 ? 64 : 32));
  long long nb = (256 + blocks - 1) / blocks;                            // one workgroup per CU in total
  if (nb > ntiles) nb = ntiles;
  if (nb < 1) nb = 1;
  if (rows) *rows = (int)(co64 ? nb : 2 * nb);
  return (int)nb;
}

int iunet_conv2_wgrad_v2_launch(int dtype, const void* x, long long x_ss, const void* dy, long long dy_ss, float* slab, int N, int H, int W,
                                int Cin, int Cout, const float* x_scale, const float* x_shift, hipStream_t stream) {
  const bool co64 = wg2_co64(Cout);
  Wgrad2Params p;
  p.x = x; p.x_ss = x_ss; p.dy = dy; p.dy_ss = dy_ss; p.slab = slab; p.x_scale = x_scale; p.x_shift = x_shift;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.tilesY = (H + (co64 ? 8 : 16) - 1) / (co64 ? 8 : 16); p.tilesX = (W + 31) / 32;
  const int nb = iunet_conv2_wgrad_v2_blocks(N, H, W, Cin, Cout, nullptr);
  dim3 grid(nb, Cout / (co64 ? 64 : 32), Cin / 32);
#define WG2_LAUNCH(TT, C64)                                                                                   \
  do {                                                                                                        \
    constexpr int LDS = 2 * W2Tile<C64>::BUF;                                                                 \
    IUNET_SET_MAX_LDS((conv2_wgrad_v2_kernel<TT, C64>), LDS);                                                 \
    hipLaunchKernelGGL((conv2_wgrad_v2_kernel<TT, C64>), grid, dim3(768), LDS, stream, p);                    \
  } while (0)
  if (dtype == 0) { if (co64) WG2_LAUNCH(f16, true); else WG2_LAUNCH(f16, false); }
  else { if (co64) WG2_LAUNCH(bf16, true); else WG2_LAUNCH(bf16, false); }
#undef WG2_LAUNCH
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}
