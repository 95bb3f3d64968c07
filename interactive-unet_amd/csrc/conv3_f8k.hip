// 3x3x3 convolution on the K = 128 fp8 matrix instruction of gfx950 (v_mfma_f32_16x16x128_f8f6f4 with e4m3 operands and unit
// block scales: twice the multiply-adds per clock of the K = 32 form that conv3_f8.hip uses, MI355X_MICROARCH.md "Matrix cores";
// tools/micro/mfma_f8_scaled.hip checks the lane map with exact integers and measures 4.2-4.5 PF in a bare loop against the
// 1.5-1.66 PF of the 16-bit loop).  BASELINE config C5, inference only; the structure is conv3_f8.hip's (one persistent workgroup
// per CU: consumer waves read LDS and issue MFMAs, loader waves feed the other LDS buffer, one barrier per step, per-XCD bricks).
//
// What K = 128 changes.  A lane of the instruction holds 32 consecutive k-bytes (k = 32 (lane >> 4) + j), i.e. the 32 input
// channels of a step at ONE filter column (dz, dx); the four lane groups q take four columns, the row shift dy stays a choice of
// activation row fragment (each fragment feeds the three dy taps, as in the K16 order).  The 9 columns of a 3x3x3 filter are two
// K = 128 groups (8 columns) + ONE K = 32 instruction of the old form for the ninth: 80 matrix cycles per (32 channels, dy, 16 x 16
// outputs) against the 160 of ten K = 32 instructions (five column pairs x two 16-channel chunks), 27 of 30 k-slots in use in both.
//   * step = 32 input channels; LDS halo image = two 16-channel planes [voxel][16 B e4m3] with the z-plane stride padded from
//     180 to 192 voxels: the lane groups q = 0 / 1 (and 2 / 3) of a ds_read_b128 share one 16-lane LDS pass, and their columns
//     are chosen to differ in dz only, so the pass sees 16 distinct 16-byte slots (conflict-free);
//   * operator in HBM and LDS per (32 Cout, 32 Cin): [group 2][dy 3][m 2][half 2][64 lanes][16 B] + [dy 3][m 2][64][8 B] for
//     the ninth column (27 648 B; pack_batch.hip kind 5 / pack_f8_kernel write this order when iunet_f8_k128 says so);
//   * the operator streams by LDS-DMA into two step buffers (64 input channels per workgroup = two steps: both stay resident);
//   * activations (template flag IN8): e4m3 planes [C / 16][D][H][W][16 B] -- the format the fp8 layers hand each other (engine.py:
//     q_planes), the HBM tensor IS the LDS image, 4 loader waves copy it by LDS-DMA -- or 16-bit NHWC8c planes rounded to e4m3 by 8
//     loader waves (the stand-alone entry point; that form was bound by its loaders: 453 us with the MFMAs compiled out against 529 us
//     in all on 128 -> 64 @ 128^3, which is why the planes exist).  The epilogue writes 16-bit planes or e4m3 planes (out8).
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef long i64;
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <bool SMALL> struct F8KTile { static constexpr int TZ = SMALL ? 2 : 4, TY = 8, TX = 16, NCW = SMALL ? 4 : 8; };

struct ConvF8KParams {
  const void* x;  long long x_sstride;        // 16-bit NHWC8c activations, or (IN8) e4m3 planes [Cin / 16][D][H][W][16 B], strides in bytes
  void* y;        long long y_sstride;
  const void* wpk;                            // e4m3 bytes, K128 order (header)
  const float* wscale;                        // [Cout] power-of-two dequantisation scales
  const float* bias;
  int N, D, H, W, Cin, Cout;
  int tilesZ, tilesY, tilesX;
  int bz, by, bx;
  int nbz, nby, nbx;
  int epi;
  int ksplit;                                 // > 1: blockIdx.z owns Cin / ksplit input channels and writes fp32 partial sums
  float* partial;                             // [ksplit][N][Cout / 8][voxels][8] fp32 (split-K only)
  int out8;                                   // 1: y is e4m3 planes [Cout / 16][D][H][W][16 B] (strides in bytes), the 16-bit result rounded once more
  int dbg;                                    // profiling only (IUNET_F8K_DBG): 1 no e4m3 conversion, 2 no activation loads, 8 no operator copies
};

// loader threads: 8 waves when the 16-bit input is rounded on the way into LDS (tools/f8k_ab.sh: 529 us against 574 with 4 on
// 128 -> 64 @ 128^3), 4 when the input is e4m3 already and everything moves by LDS-DMA
#ifndef F8K_NLT
#define F8K_NLT 512
#endif
#ifndef F8K_NLT8
#define F8K_NLT8 256
#endif
// 16 zero bytes: the source of the halo voxels outside the image when the halo tile goes global -> LDS without registers
__device__ __attribute__((aligned(16))) unsigned int g_f8k_zero16[4] = {0u, 0u, 0u, 0u};
constexpr int f8k_loader_threads(bool in8) { return in8 ? F8K_NLT8 : F8K_NLT; }

template <typename T, bool SMALL, bool IN8>
__global__ __launch_bounds__((F8KTile<SMALL>::NCW * 64 + f8k_loader_threads(IN8)), 1) void conv3_f8k_kernel(ConvF8KParams p) {
  using V8 = typename Vec8<T>::type;
  using TL = F8KTile<SMALL>;
  constexpr int NCW = TL::NCW, NLT = f8k_loader_threads(IN8), NLW = NLT / 64;
  constexpr int TZ = TL::TZ, TY = TL::TY, TX = TL::TX;
  constexpr int NI = TZ * TY / NCW, NR = NI;                   // 4 tile rows (16 voxels each) per consumer wave
  constexpr int PZ = TZ + 2, PY = TY + 2, PX = TX + 2;
  constexpr int ZS = 192;                                      // voxels per halo z-plane in LDS (180 real + 12 pad: 12 x 16)
  constexpr int NPIX = PZ * PY * PX;                           // real halo voxels (1080 / 720)
  constexpr int PLANE = PZ * ZS * 16;                          // one 16-channel plane of the halo tile
  constexpr int ABUF = 2 * PLANE;
  constexpr int W128 = 2 * 3 * 2 * 2 * 1024, W32 = 3 * 2 * 512, WSTEP = W128 + W32;      // 27 648
  constexpr int OFF_W = 2 * ABUF;
  constexpr int OFF_E = OFF_W + 2 * WSTEP;                     // [scale 32 | bias 32] floats of this Cout tile
  constexpr int AIT = (NPIX + NLT - 1) / NLT;                  // halo voxels per loader thread (5 / 3)
  constexpr int WIT = (WSTEP / 1024 + NLW - 1) / NLW;          // 1-KB weight pieces per loader wave (7 / 4)
  static_assert(NI == 4 && NCW * NI == TZ * TY, "a consumer wave owns 4 rows of one z slice");
  static_assert(WSTEP % 1024 == 0, "the operator of a step is whole LDS-DMA pieces");

  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cob = blockIdx.y;
  const int nbricks = p.N * p.nbz * p.nby * p.nbx;
  const int xcd = (blockIdx.x + (nbricks < 8 ? blockIdx.y + 3 * blockIdx.z : 0)) & 7, slot = blockIdx.x >> 3;
  const int sx = slot % p.bx, sy = (slot / p.bx) % p.by, sz = slot / (p.bx * p.by);
  const int b_begin = (int)((long long)xcd * nbricks / 8), b_end = (int)((long long)(xcd + 1) * nbricks / 8);
  const int nchunk_all = p.Cin / 32;
  const int nchunk = nchunk_all / p.ksplit;
  const int chunk0 = (int)blockIdx.z * nchunk;
  const int nsteps = (b_end - b_begin) * nchunk;
  if (nsteps <= 0) return;
  const long long plane_stride = (long long)p.D * p.H * p.W * 8;
  const unsigned char* wsrc = (const unsigned char*)p.wpk + ((long long)cob * nchunk_all + chunk0) * WSTEP;

  auto tile_origin = [&](int k, int& n_img, int& z0, int& y0, int& x0) -> bool {
    int b = b_begin + k;
    const int Bx = b % p.nbx; b /= p.nbx;
    const int By = b % p.nby; b /= p.nby;
    const int Bz = b % p.nbz; n_img = b / p.nbz;
    const int tz = Bz * p.bz + sz, ty = By * p.by + sy, tx = Bx * p.bx + sx;
    z0 = tz * TZ; y0 = ty * TY; x0 = tx * TX;
    return tz < p.tilesZ && ty < p.tilesY && tx < p.tilesX;
  };
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  if (tid < 64) ((float*)(smem + OFF_E))[tid] = tid < 32 ? p.wscale[cob * 32 + tid] : (p.epi != 0 ? p.bias[cob * 32 + tid - 32] : 0.f);

  if (wave >= NCW) {
    const unsigned* zero16 = iunet_opaque_ptr((const unsigned*)g_f8k_zero16);      // (common.h: one address computation per kernel, not one per DMA piece)
    // ================================================================== loader waves
    const int lt = tid - NCW * 64;
    const int lw = __builtin_amdgcn_readfirstlane(lt >> 6);
    auto dma_weights = [&](int s, int buf) {           // the operator of step s: global -> LDS directly (LDS-DMA), lane-linear both sides
      if (p.dbg & 8) return;
      const int chunk = s - (s / nchunk) * nchunk;
      const unsigned char* ws = wsrc + (long long)chunk * WSTEP;
#pragma unroll
      for (int it = 0; it < WIT; ++it) {
        const int piece = it * NLW + lw;
        if (piece < WSTEP / 1024) {
          const unsigned char* gsrc = ws + piece * 1024 + (lt & 63) * 16;
          const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + OFF_W + buf * WSTEP + piece * 1024);
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
        }
      }
    };
    if constexpr (IN8) {
      // e4m3 planes in HBM ARE the LDS image: the halo tile goes global -> LDS without registers (conv3_v4.hip: dma_acts).  A wave
      // instruction fills 64 consecutive 16-byte slots of a plane (z-plane stride 192: the 12 pad slots and the voxels outside the
      // image read 16 zero bytes).  Issued, waited for and published by the next barrier within one step.
      constexpr int NSLOT = PZ * ZS;                             // slots per 16-channel plane (1152 / 768: whole wave instructions)
      constexpr int DIT = (NSLOT + NLT - 1) / NLT;
      static_assert(NSLOT % 64 == 0, "a plane is whole wave instructions");
      int dcoord[DIT];
#pragma unroll
      for (int it = 0; it < DIT; ++it) {
        const int slot = lt + it * NLT;
        const int pz = slot / ZS, rem = slot - pz * ZS, py = rem / PX, px = rem - py * PX;
        dcoord[it] = (slot < NSLOT && rem < PY * PX) ? (px | (py << 8) | (pz << 16)) : -1;
      }
      const long long plane16 = (long long)p.D * p.H * p.W * 16;
      auto dma_acts = [&](int s) {
        if (p.dbg & 2) return;
        const int chunk = s - (s / nchunk) * nchunk;
        int n_img, z0, y0, x0;
        tile_origin(s / nchunk, n_img, z0, y0, x0);
        const unsigned char* xc = (const unsigned char*)p.x + (long long)n_img * p.x_sstride + (long long)(chunk0 + chunk) * 2 * plane16;
        const unsigned abuf = lds0 + (s & 1) * ABUF;
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
          const int base = it * NLT + lw * 64;                     // first slot of this wave instruction
          if (base < NSLOT) {
            const int c = dcoord[it];
            const int gz = z0 + (c >> 16) - 1, gy = y0 + ((c >> 8) & 255) - 1, gx = x0 + (c & 255) - 1;
            const bool ok = c >= 0 && (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            const long long goff = (((long long)gz * p.H + gy) * p.W + gx) * 16;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const unsigned char* gsrc = ok ? xc + e * plane16 + goff : (const unsigned char*)zero16;
              const unsigned dst = __builtin_amdgcn_readfirstlane(abuf + e * PLANE + base * 16);
              unsigned keep;
              asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                           : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
            }
          }
        }
      };
      // two steps per tile (64 input channels per workgroup): the two operator buffers hold the two steps' operators for the whole
      // launch -- copied once, 27 KB of every step's 64 KB less across the L2 -> CU fabric
      const bool wres = nchunk == 2;
      dma_weights(0, 0);
      if (wres && nsteps > 1) dma_weights(1, 1);
      dma_acts(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) { if (!wres) dma_weights(s + 1, (s + 1) & 1); dma_acts(s + 1); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
      }
      return;
    }
    int pcoord[AIT], plds[AIT];
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      const int pix = min(lt + it * NLT, NPIX - 1);
      const int px = pix % PX, t2 = pix / PX, py = t2 % PY, pz = t2 / PY;
      pcoord[it] = px | (py << 8) | (pz << 16);
      plds[it] = (pz * ZS + py * PX + px) * 16;
    }
    struct Staged { u32x4 a[AIT][4]; unsigned ok; };
    auto load = [&](int s, Staged& r) {
      const int chunk = s - (s / nchunk) * nchunk;
      int n_img, z0, y0, x0;
      tile_origin(s / nchunk, n_img, z0, y0, x0);
      const T* xc = (const T*)p.x + (long long)n_img * p.x_sstride + (long long)(chunk0 + chunk) * 4 * plane_stride;
      r.ok = 0;
      if (p.dbg & 2) return;
#pragma unroll
      for (int it = 0; it < AIT; ++it) {
        const int px = pcoord[it] & 255, py = (pcoord[it] >> 8) & 255, pz = pcoord[it] >> 16;
        const int gz = z0 + pz - 1, gy = y0 + py - 1, gx = x0 + px - 1;
        const bool ok = (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        const int cz = min(max(gz, 0), p.D - 1), cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
        const long long goff = (((long long)cz * p.H + cy) * p.W + cx) * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) r.a[it][k] = *(const u32x4*)(xc + k * plane_stride + goff);
        r.ok |= ok ? (1u << it) : 0u;
      }
    };
    // registers -> LDS buffer s & 1: four 8-channel planes of the 16-bit input become two 16-byte e4m3 granules per halo voxel
    auto commit = [&](int s, const Staged& r) {
      unsigned char* ab = smem + (s & 1) * ABUF;
#pragma unroll
      for (int it = 0; it < AIT; ++it) {
        const int pix = lt + it * NLT;
        u32x4 v0 = r.a[it][0], v1 = r.a[it][1], v2 = r.a[it][2], v3 = r.a[it][3];
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));     // the loads are waited for on EVERY path (see conv3_v4.hip)
        if (pix < NPIX) {
          const bool ok = (r.ok >> it) & 1u;
          unsigned o0, o1, o2, o3, o4, o5, o6, o7;
          if (p.dbg & 1) { *(u32x4*)(ab + plds[it]) = v0; *(u32x4*)(ab + PLANE + plds[it]) = v2; continue; }
          e4m3_pack8<T>(v0, o0, o1);
          e4m3_pack8<T>(v1, o2, o3);
          e4m3_pack8<T>(v2, o4, o5);
          e4m3_pack8<T>(v3, o6, o7);
          *(u32x4*)(ab + plds[it]) = ok ? u32x4{o0, o1, o2, o3} : u32x4{0u, 0u, 0u, 0u};
          *(u32x4*)(ab + PLANE + plds[it]) = ok ? u32x4{o4, o5, o6, o7} : u32x4{0u, 0u, 0u, 0u};
        }
      }
    };
    const int last = nsteps - 1;
    Staged r;
    const bool wres = nchunk == 2;                                 // (see the e4m3 path above)
    dma_weights(0, 0);
    if (wres && nsteps > 1) dma_weights(1, 1);
    load(0, r);
    commit(0, r);
    load(min(1, last), r);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the first operator copy has landed
    lds_barrier();
    // one register set: in iteration s (consumers on step s) the operator of step s + 1 is copied, step s + 1's activations --
    // loaded during step s - 1 -- are converted and written, and the loads of step s + 2 are issued into the same registers: they
    // stay in flight over the barrier and the consumers' whole next step
    for (int s = 0; s < nsteps; ++s) {
      if (s + 1 < nsteps && !wres) dma_weights(s + 1, (s + 1) & 1);
      commit(s + 1, r);                                          // (buffer (s + 1) & 1: garbage after the last step, unread)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the operator copy has landed before the barrier publishes it
      load(min(s + 2, last), r);
      lds_barrier();
    }
    return;
  }

  // ==================================================================== consumer waves
  const int l15 = lane & 15, q = lane >> 4;
  // byte offset of this lane group's filter column inside a halo plane, per K = 128 group; the ninth column for the K = 32 instruction
  int coff[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int col = q == 0 ? f8k_col(g, 0) : q == 1 ? f8k_col(g, 1) : q == 2 ? f8k_col(g, 2) : f8k_col(g, 3);
    coff[g] = ((col / 3) * ZS + (col % 3)) * 16;
  }
  const int coff8 = (2 * ZS + 2) * 16 + (q >> 1) * PLANE + (q & 1) * 8;
  const int row_first = wave * NI;                             // first tile row (z * TY + y) of this wave
  const int rbase = (((row_first / TY) * ZS + (row_first % TY) * PX) + l15) * 16;

  f32x4 acc[2][NI];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NI; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // operand registers: K = 128 row fragments of the two groups, K = 128 operator fragments of the running and the next (group, dy),
  // and the K = 32 group's 8-byte fragments
  i32x8 R[2][NR + 2], A[2][2];
  i64 R8[NR + 2], A8[3][2];

  auto rd128 = [&](const unsigned char* ptr, int second) -> i32x8 {          // two 16-byte halves `second` bytes apart
    const u32x4 lo = *(const u32x4*)ptr, hi = *(const u32x4*)(ptr + second);
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  // sub-group sg = 0 .. 8 of a step: (group 0, dy 0..2), (group 1, dy 0..2), (ninth column, dy 0..2)
  auto load_sub = [&](const unsigned char* ab, const unsigned char* wl, auto SG) {
    constexpr int sg = decltype(SG)::value, g = sg / 3, dy = sg % 3;
    if constexpr (g < 2) {
#pragma unroll
      for (int m = 0; m < 2; ++m) A[sg & 1][m] = rd128(wl + ((g * 3 + dy) * 2 + m) * 2048 + lane * 16, 1024);
      constexpr int r0 = dy == 0 ? 0 : NR - 1 + dy, r1 = dy == 0 ? NR : NR + dy;          // new rows: 0 .. 3, then 4, then 5
#pragma unroll
      for (int r = r0; r < r1; ++r) R[g][r] = rd128(ab + rbase + coff[g] + r * PX * 16, PLANE);
    } else if constexpr (dy == 0) {
      // the ninth column's operands are 8 bytes per lane: ALL of them (6 operator + 6 row fragments, 24 registers) are read behind the
      // last K = 128 sub-group, so the step's barrier can sit in front of its 24 K = 32 instructions -- the loaders get the buffer
      // 384 matrix cycles earlier, and the next step's first 12 reads have those cycles to land
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int m = 0; m < 2; ++m) A8[d][m] = *(const i64*)(wl + W128 + (d * 2 + m) * 512 + lane * 8);
#pragma unroll
      for (int r = 0; r < NR + 2; ++r) R8[r] = *(const i64*)(ab + rbase + coff8 + r * PX * 16);
    }
  };
  auto mfma_sub = [&](auto SG, auto NREADS) {
    constexpr int sg = decltype(SG)::value, g = sg / 3, dy = sg % 3, nreads = decltype(NREADS)::value;
#ifndef F8K_NOMFMA                                         // ablation builds only (tools/ab_build.sh conv3_f8k.hip -DF8K_NOMFMA)
#pragma unroll
    for (int n = 0; n < NI; ++n)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if constexpr (g < 2) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[sg & 1][m], R[g][n + dy], acc[m][n], 0, 0, 0, 0, 0, 0);
        else acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(A8[dy][m], R8[n + dy], acc[m][n], 0, 0, 0);
      }
#else
    if constexpr (g < 2) { asm volatile("" :: "v"(A[sg & 1][0]), "v"(A[sg & 1][1]), "v"(R[g][dy]), "v"(R[g][dy + 1]), "v"(R[g][dy + 2]), "v"(R[g][dy + 3])); }
    else { asm volatile("" :: "v"(A8[dy][0]), "v"(A8[dy][1]), "v"(R8[dy]), "v"(R8[dy + 1]), "v"(R8[dy + 2]), "v"(R8[dy + 3])); }
#endif
    // spread the next sub-group's LDS reads between this sub-group's 8 MFMAs
    if constexpr (nreads >= 8) {
      constexpr int RPM = (nreads + 7) / 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);       // RPM LDS reads
      }
    } else if constexpr (nreads > 0) {
      constexpr int MPR = 8 / nreads;
#pragma unroll
      for (int i = 0; i < nreads; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);       // MPR MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         // one LDS read
      }
    }
    if constexpr (sg < 6 || sg == 8) __builtin_amdgcn_sched_barrier(0);      // (the three K = 32 sub-groups are one scheduling region)
  };
  auto tile_epilogue = [&](int s) {
    const int chunk = s - (s / nchunk) * nchunk;
    if (chunk == nchunk - 1) {
      int n_img, z0, y0, x0;
      tile_origin(s / nchunk, n_img, z0, y0, x0);
      T* yout = (T*)p.y + (long long)n_img * p.y_sstride;
      // the 8 output channels of this lane group: scale and bias come from LDS, once per tile (16 registers the step loop needs;
      // a global load here would put a vmcnt(0) wait -- i.e. a wait for the previous tile's output stores -- at the head of the loop)
      float bias_r[8], ws_r[8];
      {
        const f32x4* ep = (const f32x4*)(smem + OFF_E) + 2 * q;
        const f32x4 w0 = ep[0], w1 = ep[1], b0 = ep[8], b1 = ep[9];
#pragma unroll
        for (int j = 0; j < 4; ++j) { ws_r[j] = w0[j]; ws_r[4 + j] = w1[j]; bias_r[j] = b0[j]; bias_r[4 + j] = b1[j]; }
      }
#pragma unroll
      for (int n = 0; n < NI; ++n) {
        const int row = row_first + n;
        const int gz = z0 + row / TY, gy = y0 + row % TY, gx = x0 + l15;
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
          if (!p.out8) { if (ok) *(V8*)(yout + (long long)(cob * 4 + q) * plane_stride + vo) = o; }
          else {                     // e4m3 planes: this lane group's 8 channels are half of a 16-byte granule
            unsigned o0, o1;
            e4m3_pack8<T>(__builtin_bit_cast(u32x4, o), o0, o1);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            if (ok) *(u32x2*)((unsigned char*)p.y + (long long)n_img * p.y_sstride + (long long)(cob * 2 + (q >> 1)) * plane_stride * 2 + vo * 2 + (q & 1) * 8) = u32x2{o0, o1};
          }
        }
        acc[0][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[1][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto step_ptrs = [&](int s, const unsigned char*& ab, const unsigned char*& wl) {
    ab = smem + (s & 1) * ABUF;
    wl = smem + OFF_W + (s & 1) * WSTEP;
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
  using I6 = std::integral_constant<int, 6>; using I7 = std::integral_constant<int, 7>; using I8 = std::integral_constant<int, 8>;
  using N4 = std::integral_constant<int, 4>; using N6 = std::integral_constant<int, 6>; using N12 = std::integral_constant<int, 12>;
  using N3 = std::integral_constant<int, 3>;

  lds_barrier();                                             // step 0 is in LDS
  {
    const unsigned char *ab, *wl;
    step_ptrs(0, ab, wl);
    load_sub(ab, wl, I0{});
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int s = 0; s < nsteps; ++s) {
    const unsigned char *ab, *wl, *abn, *wln;
    step_ptrs(s, ab, wl);
    step_ptrs(s + 1 < nsteps ? s + 1 : s, abn, wln);      // (the last step re-reads its own buffers: unused, but one straight path)
    // reads of sub-group i + 1 (LDS read instructions: 4 operator + 8 / 2 / 2 row reads for K = 128; 2 + 4 / 1 / 1 for K = 32)
    load_sub(ab, wl, I1{}); mfma_sub(I0{}, N6{});
    load_sub(ab, wl, I2{}); mfma_sub(I1{}, N6{});
    load_sub(ab, wl, I3{}); mfma_sub(I2{}, N12{});
    load_sub(ab, wl, I4{}); mfma_sub(I3{}, N6{});
    load_sub(ab, wl, I5{}); mfma_sub(I4{}, N6{});
    load_sub(ab, wl, I6{}); mfma_sub(I5{}, N12{});
    lds_barrier();                                           // step s + 1 is published; every read of step s has landed
    load_sub(abn, wln, I0{});                                // 12 reads spread over the 24 K = 32 instructions
    mfma_sub(I6{}, N4{}); mfma_sub(I7{}, N4{}); mfma_sub(I8{}, N4{});
    tile_epilogue(s);
  }
}

template <typename T, bool SMALL, bool IN8>
int launch_f8k(ConvF8KParams p, hipStream_t stream) {
  using TL = F8KTile<SMALL>;
  constexpr int PLANE = (TL::TZ + 2) * 192 * 16;
  constexpr int WSTEP = 2 * 3 * 2 * 2 * 1024 + 3 * 2 * 512;
  const int lds = 2 * 2 * PLANE + 2 * WSTEP + 256;           // 129 280 / 104 704 B
  IUNET_SET_MAX_LDS((conv3_f8k_kernel<T, SMALL, IN8>), lds);
  p.tilesZ = (p.D + TL::TZ - 1) / TL::TZ; p.tilesY = (p.H + TL::TY - 1) / TL::TY; p.tilesX = (p.W + TL::TX - 1) / TL::TX;
  const int ncob = p.Cout / 32;
  iunet_brick_shape(3, ncob, p.tilesZ, p.tilesY, p.tilesX, &p.bz, &p.by, &p.bx);
  p.nbz = (p.tilesZ + p.bz - 1) / p.bz; p.nby = (p.tilesY + p.by - 1) / p.by; p.nbx = (p.tilesX + p.bx - 1) / p.bx;
  const int gx = 8 * p.bz * p.by * p.bx;
  hipLaunchKernelGGL((conv3_f8k_kernel<T, SMALL, IN8>), dim3(gx, ncob, p.ksplit), dim3(TL::NCW * 64 + f8k_loader_threads(IN8)), lds, stream, p);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

}  // namespace

// Which operator order a 3^d fp8 conv uses: 1 = the K = 128 order of this file (3-D, Cin a multiple of 32), 0 = the K16 order of
// conv3_f8.hip.  A function of the layer only (taps, Cin): the packed operator serves every launch size.  IUNET_F8_K128=0: A/B switch.
int iunet_f8_k128(int taps, int Cin) {
  static const int off = getenv("IUNET_F8_K128") ? (atoi(getenv("IUNET_F8_K128")) == 0) : 0;
  return !off && taps == 27 && Cin % 32 == 0;
}

int iunet_conv3_f8k_launch(int dtype, const void* x, long long x_sstride, void* y, long long y_sstride, const void* wpk,
                           const float* wscale, const float* bias, int N, int D, int H, int W, int Cin, int Cout, int epi,
                           int ksplit, float* partial, int small, int in8, int out8, hipStream_t stream) {
  ConvF8KParams p;
  p.ksplit = ksplit; p.partial = partial;
  p.x = x; p.x_sstride = x_sstride; p.y = y; p.y_sstride = y_sstride; p.wpk = wpk; p.wscale = wscale; p.bias = bias;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.epi = epi;
  p.tilesZ = p.tilesY = p.tilesX = 0;
  p.bz = p.by = p.bx = p.nbz = p.nby = p.nbx = 0;
#ifdef IUNET_ABLATE      // result-destroying profiling switches exist in diagnostic builds only (tools/ab_build.sh <file> -DIUNET_ABLATE): ADVICE r3
  static const int dbg = getenv("IUNET_F8K_DBG") ? atoi(getenv("IUNET_F8K_DBG")) : 0;
#else
  static const int dbg = 0;
#endif
  p.dbg = dbg;
  p.out8 = out8;
#define F8K_GO(TT) (in8 ? (small ? launch_f8k<TT, true, true>(p, stream) : launch_f8k<TT, false, true>(p, stream)) \
                        : (small ? launch_f8k<TT, true, false>(p, stream) : launch_f8k<TT, false, false>(p, stream)))
  return dtype == 0 ? F8K_GO(f16) : F8K_GO(bf16);
#undef F8K_GO
}
