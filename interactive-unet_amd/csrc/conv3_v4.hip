// 3x3x3 convolution for Cout tiles of 32 channels, wave-specialised structure (layout 2).
//
// What the ablations of the earlier structures showed (tools/_run_v3.sh): the MFMA loop alone
// runs at the clock-limited peak (1.5-1.66 PF), but a wave that also issues the global loads of the next
// chunk blocks on the CU's vector-memory queue for thousands of cycles (the path sustains 12-25 B/clk per CU),
// and a blocked wave issues no MFMAs; two such workgroups per CU, or half the bytes (weights kept in LDS), did not
// change that.  So here the roles are split inside one persistent workgroup per CU:
//   * 8 CONSUMER waves (two per SIMD) only read LDS and issue MFMAs (plus the tile's output stores);
//   * 4 or 8 LOADER waves (one or two per SIMD) fetch the next 16-channel chunk of the halo tile (and, for Cin > 32, its
//     30 KB of weights) into the OTHER LDS buffer, wait for it, and meet the consumers at the one barrier per step.  Their
//     stalls on the memory queue cost no MFMA issue slots.  Weights always travel by LDS-DMA; the halo tile does too in 3-D
//     launches whose input needs no arithmetic on the way (dma_acts), else through registers (load / commit: the fused
//     BatchNorm + ReLU of the training forward, and the HBM-bound 2-D layers, which need loads in flight across the barrier).
// Two operator orders: the K16 order of layout 1 (9 filter columns padded to 5 pairs: 30 k-slots for 27 taps) and, for 3-D launches
// with streamed weights, the COMPACT order (template flag NP, layout 3): even steps multiply 4 column pairs, odd steps those plus one
// cross pair holding the ninth column of both chunks -- 27 taps in 27 k-slots, a ring of three halo buffers.
// Round 2 measured what is left: with and without the loader waves the consumers take the same cycles per
// step (4 800, 80 % of them MFMA); the difference is the clock the chip holds (1.54 vs 1.75 GHz at 64 -> 32 @ 128^3).
// 3-D: tile 4 x 8 x 16 voxels, a step = one 16-channel chunk; 2-D: tile 16 x 32 pixels, a step = 32 channels (the 2-D
// filter has a third of the taps, so a 16-channel step would be too short between barriers).  64 voxels per consumer
// wave; k-step = 2 filter columns x 16 channels (the K16 operator of layout 1), activation row fragments reused over
// the three dy taps.  LDS: 2 x 34 KB (3-D) / 2 x 39 KB (2-D) activations + the weights (all of them, resident for the
// whole launch, when they fit: Cin <= 32 in 3-D, <= 64 in 2-D; else two step-sized buffers streamed with the activations).
#include "common.h"
#include <cstdlib>
#include <type_traits>

#ifdef IUNET_STAMPS
// Diagnostic build only (build.sh never defines IUNET_STAMPS): cycles and real time around the consumers' step loop, to
// read the in-kernel clock (s_memtime / s_memrealtime x 100 MHz) and the cycles per step; tools/conv_clock.py reads them.
__device__ unsigned long long g_v4_stamps[4 * 512];
extern "C" int iunet_v4_stamps_read(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_v4_stamps), sizeof(unsigned long long) * 4 * 512);
}
#endif
// 16 zero bytes: the source of the halo pixels outside the image when the halo tile goes global -> LDS without registers
__device__ __attribute__((aligned(16))) unsigned int g_v4_zero16[4] = {0u, 0u, 0u, 0u};
int iunet_conv3_v4_stats_parts(int nd, int Cout);
int iunet_conv3_v4_pairs(int nd, int N, int D, int H, int W, int Cin, int Cout, int bw);
int iunet_conv3_v4_x2_pack_mode(int nd);

namespace {

// tile shape, consumer waves (NCW; A/B: 4 waves x 8 fragments on the big 3-D tile measured the same, +-2 %), filter columns,
// 16-channel sub-chunks per step.  SMALL (3-D): half the tile and half the consumer waves, for launches whose 4 x 8 x 16 tiles
// would leave CUs idle (16^3 grids: 16 tiles per sample).
template <int ND, bool SMALL> struct V4Tile;
template <> struct V4Tile<3, false> { static constexpr int TZ = 4, TY = 8, TX = 16, PADZ = 1, NCOL = 9, S16 = 1, NCW = 8; };
template <> struct V4Tile<3, true>  { static constexpr int TZ = 2, TY = 8, TX = 16, PADZ = 1, NCOL = 9, S16 = 1, NCW = 4; };
template <bool SMALL> struct V4Tile<2, SMALL> { static constexpr int TZ = 1, TY = 16, TX = 32, PADZ = 0, NCOL = 3, S16 = 2, NCW = 8; };

// loader threads: 4 loader waves when only activations stream (resident weights), when tiles go in pairs, and for the 2-D split
// launches (their consumers need more than the 128 registers that 16 waves per CU leave: 52 B of scratch per lane otherwise, and the
// operand-sharing schedule moves a third fewer bytes); 8 when the weights stream too
// ([r3] the fused-BatchNorm-backward data gradient runs on 4 loader waves too: 153-165 registers and no scratch instead of 128 + 56 B
// (3-D) / 144 B (2-D) per lane; A/B on one box: C3 training step 6.90-6.92 against 6.97-7.00 ms, 2-D step 13.61-13.66 against 13.75 ms)
#ifndef V4_BW_LT
#define V4_BW_LT 256            // (A/B: -DV4_BW_LT=512)
#endif
#ifndef V4_X2_LT3D
#define V4_X2_LT3D 512          // loader threads of the 3-D split conv (A/B: -DV4_X2_LT3D=256)
#endif
#ifndef V4_RING2
#define V4_RING2 1              // 2-D cross-pair step: three halo buffers, copies two steps ahead -- bit 0: resident weights (64 -> 32 @ 8 x 512^2: 116 -> 107 us), bit 1: streamed weights too (measured: no gain, 128 -> 64 75 -> 77 us; A/B: -DV4_RING2=0 / 3)
#endif
#ifndef V4_NP3_LT
#define V4_NP3_LT 512           // loader threads of the 3-D compact-operator conv (A/B: -DV4_NP3_LT=256)
#endif
constexpr int v4_loader_threads(int nd, bool ws, bool pair, bool spl, bool bwv = false, bool np = false) { return (ws || pair || ((spl || np) && nd == 2)) ? 256 : bwv ? V4_BW_LT : (spl && nd == 3) ? V4_X2_LT3D : (np && nd == 3) ? V4_NP3_LT : 512; }

struct ConvV4Params {
  const void* x;  long long x_sstride;
  void* y;        long long y_sstride;
  const void* wpk;                            // K16 order: [cob][chunk16][column pair][dy][2][64][8]
  const float* bias;
  const float* in_scale;                      // optional [Cin] pair: the input is relu(in_scale * x + in_shift), applied by the
  const float* in_shift;                      // loader waves (training: BatchNorm + ReLU of the previous conv, never materialised)
  float* stats;                               // [gridDim.x][Cout][2] or null: one row of BatchNorm partial sums per workgroup
  int N, D, H, W, Cin, Cout;
  int tilesZ, tilesY, tilesX;
  int bz, by, bx;                             // tiles per brick (bz * by * bx = workgroups per XCD and Cout tile)
  int nbz, nby, nbx;                          // bricks per sample
  int epi;
  int dbg;                                    // profiling only (IUNET_V4_DBG): 2 no MFMA phase, 4 no stores, 64 halo tiles through registers always
  // Data-gradient launches: this launch's output IS the gradient dz of the producer layer's activation z = relu(bn(yp)).  With
  // bw_y set, the epilogue also reads yp at its output voxels and accumulates the BatchNorm-backward sums of that layer --
  // s1 = sum dz', s2 = sum dz' * xhat, dz' = dz where z > 0 (the arithmetic of bn_bwd_reduce_kernel on the STORED, rounded dz)
  // -- into `stats` ([workgroups][Cout][2], as the forward statistics): the separate reduction pass over dz and yp goes away.
  const void* bw_y; long long bw_y_ss;
  const float* bw_mean; const float* bw_invstd; const float* bw_scale; const float* bw_shift;      // [Cout] of the producer layer
  // fp16x2 split precision (template flag SPL, split16.hip): every value travels as hi = f16(v), lo = f16(v - hi) in two plane sets.
  // The launch is a conv over Cin' = 3 Cin virtual channels -- per channel chunk the parts [x_lo | x_hi | x_hi] against the operator
  // rows [w_hi | w_hi | w_lo] -- so the step loop, the LDS images and the operator order are the 16-bit kernel's own; only the source
  // plane of a chunk, the buffer naming and the epilogue differ.  split_nc = channel chunks, x_lo / y_lo = plane offset of the lo planes.
  int split_nc, x_lo, y_lo;
  const float* oscale;                        // [Cout]: power-of-two factor on the accumulator (operator and activation scales)
  int* sat;                                   // SPL: optional range flag (common.h: x2_note_saturation)
  // GroupNorm: statistics per SAMPLE.  per_sample != 0 (host only): the launch is ONE sample's (N = 1 in the kernel's eyes) with the
  // samples along gridDim.z -- workgroup (x, y, z) walks sample z and writes statistics row z * gridDim.x + x: stats [N][gridDim.x][Cout][2],
  // the slab layout of gn_finalize_kernel.  The workgroups of sample z + 1 start as those of sample z retire (one workgroup per CU).
  int per_sample;
  int* query_rows;                            // host only: not null = no launch, *query_rows = rows per sample of a per_sample launch (0: not available on this grid)
};

// BW: the data-gradient variant that also accumulates the BatchNorm-backward sums of the layer its output flows into (bw_y); a
// template parameter so that its extra registers (the prefetched yp fragments) never touch the forward instantiations
// PAIR (streamed weights only): the workgroup walks its tiles two at a time -- steps (chunk c, tile A), (chunk c, tile B), (chunk c + 1,
// tile A), ... -- so a 30 KB weight chunk is streamed once per TWO tiles.  The launch is bound by the bytes that cross the L2 -> CU
// fabric (64 -> 32 @ 2 x 128^3 moves 2.4 GB of halo tiles, weights and outputs in 0.35-0.43 ms = 5.6-6.9 TB/s, the
// rate the chip sustains), and the re-streamed weights are 41 % of them.  Cost: a second accumulator set (32 registers), so 4 loader
// waves instead of 8 (12 waves per CU: 168 registers each).
// NP ("no padding", layout 3, experimental): the compact operator order of pack mode bit 2.  A 16-channel step multiplies column pairs
// 0..3 only; the ninth column waits for the next chunk: the odd step adds one "cross" group whose k-slot holds column 8 of the
// previous chunk (read from the previous step's halo buffer, which a ring of THREE buffers keeps alive) and column 8 of its own.
// 27 taps in 27 K-slots: -10 % MFMAs, fragment reads and weight bytes.  LDS: 3 x 34 816 + 24 576 + 30 720 + scratch = 162 304 B.
template <typename T, int ND, bool WS, bool SMALL, bool BW = false, bool PAIR = false, bool NP = false, bool SPL = false>
__global__ __launch_bounds__((V4Tile<ND, SMALL>::NCW * 64 + v4_loader_threads(ND, WS, PAIR, SPL, BW, NP)), 1) void conv3_v4_kernel(ConvV4Params p) {
  static_assert(!SPL || (!WS && !BW && !PAIR && !(NP && ND == 3)), "split precision: streamed weights (Cin' = 3 Cin >= 96), forward only, padded operator in 3-D");
  static_assert(!PAIR || (!WS && !BW && ND == 3), "tile pairs: the streamed-weight 3-D forward / data-gradient variant only");
  static_assert(!NP || (!PAIR && (ND == 2 || !WS) && (!BW || (ND == 2 && WS))), "padding-free step: the streamed-weight 3-D variants and the 2-D ones; fused BatchNorm-backward sums on the 2-D resident-weights variant only");
  // NP in 3-D: the ninth column of two consecutive 16-channel STEPS shares a k-slot (ring of three halo buffers, below).  NP in 2-D
  // (NP2): a step already holds two 16-channel sub-chunks, so the third filter column of both shares one k-group inside the step --
  // groups (sub-chunk 0: columns 0, 1), (sub-chunk 1: columns 0, 1), (cross: column 2 of both) = 9 taps in 9 k-slots instead of 12:
  // -25 % MFMAs, fragment reads and weight bytes; nothing else of the step changes (two halo buffers, one weight stride).
  constexpr bool NP3 = NP && ND == 3, NP2 = NP && ND == 2;
  using V8 = typename Vec8<T>::type;
  using TL = V4Tile<ND, SMALL>;
  // consumer waves; loader threads: 4 loader waves when only activations stream, 8 when the weights stream too (twice the
  // bytes per step: the extra waves double the loads in flight, -6...-11 % on those layers)
  constexpr int NCW = TL::NCW, NLT = v4_loader_threads(ND, WS, PAIR, SPL, BW, NP);
  constexpr int TZ = TL::TZ, TY = TL::TY, TX = TL::TX, PADZ = TL::PADZ, NCOL = TL::NCOL, S16 = TL::S16;
  constexpr int FX = TX / 16, NI = TZ * TY * FX / NCW, NR = NI / FX;    // x halves; fragments per consumer wave; tile rows per wave
  constexpr int PZ = TZ + 2 * PADZ, PY = TY + 2, PX = TX + 2;
  constexpr int NPIX = PZ * PY * PX;                   // 1080 / 612
  constexpr int PLANE = ((NPIX * 16 + 255) / 256) * 256;
  constexpr int CP = 2 * S16;                          // planes (of 8 channels) per step
  constexpr int ABUF = CP * PLANE;                     // one step of the halo tile
  constexpr int NCMB = (NCOL + 1) / 2, KS = NCMB * 3;
  constexpr int WBYTES = KS * 2 * 1024;                // one 16-channel chunk of packed weights
  constexpr int WSTEP = NP2 ? 3 * 3 * 2 * 1024 : S16 * WBYTES;      // the weights of one step (NP2: three k-groups of 6 KB)
  // RING2: the 2-D cross-pair step keeps THREE halo buffers as well -- its loaders run two steps ahead by LDS-DMA (the copy of step
  // s + 2 is issued before the wait for step s + 1: the memory pipe never drains at a step barrier); streamed weights stay one step ahead
  constexpr bool RING2 = NP2 && !SPL && (WS ? (V4_RING2 & 1) != 0 : (V4_RING2 & 2) != 0);
  constexpr int NBUF = (NP3 || RING2) ? 3 : 2;         // halo buffers
  constexpr int OFF_W = NBUF * ABUF;
  constexpr int WE = 4 * 3 * 2 * 1024, WO = WE + 3 * 2 * 1024;      // NP: bytes of an even step's weights (4 column pairs) / an odd step's (+ the cross pair)
  constexpr int AIT = (NPIX + NLT - 1) / NLT;          // halo pixels per loader thread (5 / 3)
  constexpr int WIT = ((NP3 ? WO : WSTEP) / 16 + NLT - 1) / NLT;    // 16-byte weight items per loader thread (8 / 6)
  constexpr int NGRP = NP2 ? 3 : S16 * NCMB;           // (16-channel sub-chunk, column pair) groups per step
  constexpr int NRD = FX * (NR + 2) + 6;               // LDS fragment reads per group
  static_assert(NCW * NI == TZ * TY * FX, "consumer waves x fragments must cover the tile");

  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cob = blockIdx.y;
  // Brick schedule: the 32 / ncob workgroups of an XCD (blocks b, b + 8, ... share an XCD and its 4 MB L2) work on the
  // tiles of ONE compact brick of bz x by x bx tiles at the same time and move on together, so the halo voxels a tile
  // shares with its neighbours are fetched from HBM once and then hit in that L2 (contiguous per-workgroup runs had
  // every concurrent tile 32 tiles apart: the 2.1x halo re-fetch all went to HBM / Infinity Cache at ~20 GB/s per CU).
  // Slots of a brick that stick out of the tile grid are processed fully masked (speed only, never correctness).
  // blocks b and b + 8 share an XCD whatever blockIdx.y is (the grid's x extent is a multiple of 8): on a grid with fewer
  // bricks than XCDs the brick range of a workgroup is rotated by its Cout tile, so the launch still spreads over all XCDs;
  // larger grids keep the Cout tiles of one brick on one XCD (they read the same input: the second one hits in L2)
  const int nbricks = p.N * p.nbz * p.nby * p.nbx;
  // A brick clamped to a small tile grid has fewer slots than the XCD has workgroups for this Cout tile (2-D slices of 128^2 at the
  // deep levels: 2 of 8): the launch then holds `ngrp` groups of slots per XCD, each walking its own share of the XCD's bricks --
  // a batch of many small images (the 2.5-D block prediction: 128 slices per launch) fills the chip instead of a quarter of it.
  const int bslots = p.bx * p.by * p.bz, ngrp = (int)gridDim.x / (8 * bslots);
  const int xcd = (blockIdx.x + (nbricks < 8 ? blockIdx.y : 0)) & 7, slot_all = blockIdx.x >> 3;
  const int grp = slot_all / bslots, slot = slot_all - grp * bslots;
  const int sx = slot % p.bx, sy = (slot / p.bx) % p.by, sz = slot / (p.bx * p.by);
  const int xb0 = (int)((long long)xcd * nbricks / 8), xb1 = (int)((long long)(xcd + 1) * nbricks / 8);
  const int b_begin = xb0 + (int)((long long)(xb1 - xb0) * grp / ngrp), b_end = xb0 + (int)((long long)(xb1 - xb0) * (grp + 1) / ngrp);
  const int nchunk = p.Cin / (16 * S16);               // steps per tile
  const int nsteps = (b_end - b_begin) * nchunk;
  // step -> (tile of this workgroup, 16-channel chunk).  PAIR: tile = 2 * pair + (s & 1), chunk advances every second step
  auto tile_of = [&](int s) -> int { return PAIR ? 2 * (s / (2 * nchunk)) + (s & 1) : s / nchunk; };
  auto chunk_of = [&](int s) -> int { return PAIR ? (s - (s / (2 * nchunk)) * 2 * nchunk) >> 1 : s - (s / nchunk) * nchunk; };
  if (nsteps <= 0) {                                   // no tile for this workgroup: its statistics row is zero
    if (p.stats != nullptr && tid < 64) p.stats[((long long)(blockIdx.z * gridDim.x + blockIdx.x) * p.Cout + cob * 32 + (tid >> 1)) * 2 + (tid & 1)] = 0.f;
    return;
  }
  const long long plane_stride = (long long)p.D * p.H * p.W * 8;
  const int off_red = OFF_W + (NP3 ? WE + WO : (WS ? nchunk : 2) * WSTEP);    // 2 KB of scratch for the BatchNorm partial sums
  const int off_act = off_red + 2048;                       // the fused input activation: [Cin / 8][scale 8 | shift 8] floats
  const int off_bw = off_act + (p.in_scale != nullptr || !NP ? p.Cin * 8 : 0);                   // bw_y: [mean | invstd | scale | shift][32] floats of this Cout tile
  constexpr bool bw = BW;
  if (bw && tid < 128) {
    const float* src = (tid >> 5) == 0 ? p.bw_mean : (tid >> 5) == 1 ? p.bw_invstd : (tid >> 5) == 2 ? p.bw_scale : p.bw_shift;
    ((float*)(smem + off_bw))[tid] = src[blockIdx.z * p.Cout + cob * 32 + (tid & 31)];       // read in tile epilogues, many barriers later (z > 0: a per-sample launch, the sample's rows -- GroupNorm)
  }
  const u32x4* wsrc = (const u32x4*)p.wpk + (long long)cob * (NP3 ? (nchunk / 2) * ((WE + WO) / 16) : nchunk * (WSTEP / 16));
  // SPL: virtual chunk = 3 * c + part for channel chunk c; part 0 = x_lo w_hi, 1 = x_hi w_hi, 2 = x_hi w_lo.  Consecutive parts share an
  // operand -- 0 -> 1 the weights, 1 -> 2 the halo tile -- so the buffers are named by what they hold instead of by step parity (halo:
  // 0 = lo, 1 = hi; weights: 0 = w_hi, 1 = w_lo) and the loaders skip what is resident: 2 halo tiles + 2 weight chunks per channel
  // chunk instead of 3 + 3.
  auto part_of = [&](int s) -> int { const int c = chunk_of(s); return c - 3 * (c / 3); };
  auto abuf_of = [&](int s) -> int {                                                               // halo buffer of step s
    if constexpr (SPL) return part_of(s) == 0 ? 0 : ABUF;
    return (NP3 || RING2) ? (s - 3 * (s / 3)) * ABUF : (s & 1) * ABUF;
  };
  // first source plane of a chunk
  auto src_plane = [&](int chunk) -> long long {
    if constexpr (SPL) {
      const int c = chunk / 3, part = chunk - 3 * c;
      return (long long)c * CP + (part == 0 ? p.x_lo : 0);
    }
    return (long long)chunk * CP;
  };

  auto tile_origin = [&](int k, int& n_img, int& z0, int& y0, int& x0) -> bool {      // this workgroup's k-th tile
    int b = b_begin + k;
    const int Bx = b % p.nbx; b /= p.nbx;
    const int By = b % p.nby; b /= p.nby;
    const int Bz = b % p.nbz; n_img = b / p.nbz + blockIdx.z;      // (gridDim.z > 1: a per-sample launch, N = 1 per z)
    const int tz = Bz * p.bz + sz, ty = By * p.by + sy, tx = Bx * p.bx + sx;
    z0 = tz * TZ; y0 = ty * TY; x0 = tx * TX;
    return tz < p.tilesZ && ty < p.tilesY && tx < p.tilesX;
  };

  if constexpr (SPL) {     // [oscale 32 | bias 32] of this Cout tile in the (unused) statistics scratch: read by the tile epilogues
    if (tid < 64) ((float*)(smem + off_red))[tid] = tid < 32 ? p.oscale[cob * 32 + tid] : (p.epi != 0 ? p.bias[cob * 32 + tid - 32] : 0.f);
  }
  if (WS) {     // all weights of this Cout tile: global -> LDS once, by everybody
    const int nitems = nchunk * (WSTEP / 16);
    // (a rolled loop on purpose: staged through stage_to_lds -- 8 loads in flight per thread -- these launches measured 13 % SLOWER,
    //  272 against 240 us at 32 -> 32 @ 2 x 128^3)
    for (int i = tid; i < nitems; i += NCW * 64 + NLT) *(u32x4*)(smem + OFF_W + i * 16) = wsrc[i];
  }

  if (p.in_scale != nullptr) {
    float* ap = (float*)(smem + off_act);
    for (int c = tid; c < p.Cin; c += NCW * 64 + NLT) {
      ap[(c >> 3) * 16 + (c & 7)] = p.in_scale[c];
      ap[(c >> 3) * 16 + 8 + (c & 7)] = p.in_shift[c];
    }
    __syncthreads();
  }

  if (wave >= NCW) {
    const unsigned* zero16 = iunet_opaque_ptr((const unsigned*)g_v4_zero16);      // (common.h: one address computation per kernel, not one per DMA piece)
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
    const int lw = __builtin_amdgcn_readfirstlane(lt >> 6);                 // loader wave index
    // streamed weights go global -> LDS directly (LDS-DMA: no staging registers, no ds_write): a lane-linear copy of the
    // step's WSTEP bytes; one wave instruction moves 64 x 16 B to M0-base + lane * 16.  The copy is in flight on the
    // vector-memory counter; the wait for the activation loads that were issued before it retires it too (in order).
    auto dma_weights = [&](int s, int buf, int it0 = 0, int it1 = 1 << 20) {     // the weights of step s -> weight buffer buf (items it0..it1 of each thread)
      const int chunk = chunk_of(s);
      // NP: the pair block of the chunk's pair is [even 24 KB | odd 24 KB | cross 6 KB]; an even step takes the first part into the
      // "even" region, an odd step the rest into the "odd" region (each region is free again after the next step of the other parity)
      const u32x4* ws = NP3 ? wsrc + (long long)(chunk >> 1) * ((WE + WO) / 16) + ((chunk & 1) ? WE / 16 : 0) : wsrc + (long long)chunk * (WSTEP / 16);
      const int nitem = NP3 ? ((chunk & 1) ? WO / 16 : WE / 16) : WSTEP / 16;
      const int woff = NP3 ? ((chunk & 1) ? WE : 0) : buf * WSTEP;
#pragma unroll
      for (int it = 0; it < WIT; ++it) {
        if (it < it0 || it >= it1) continue;
        const int base = it * NLT + lw * 64;                                  // first item of this wave instruction
        if (base < nitem) {
          const u32x4* gsrc = ws + min(base + (lt & 63), nitem - 1);
          const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + OFF_W + woff + base * 16);   // uniform by construction; keeps it in an SGPR for M0
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
        }
      }
    };
    auto load = [&](int s, Staged& r) {               // issue the global loads of step s (nothing consumes them here)
      const int chunk = chunk_of(s);
      int n_img, z0, y0, x0;
      tile_origin(tile_of(s), n_img, z0, y0, x0);
      const T* xc = (const T*)p.x + (long long)n_img * p.x_sstride + src_plane(chunk) * plane_stride;
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
    // The halo tile of step s global -> LDS buffer s & 1 WITHOUT registers (launches whose input needs no arithmetic on the way: every
    // conv but the training forward of a stage's second conv).  Measured with the register path's parts switched off one at a time
    // (64 -> 32 @ 2 x 128^3: 466 us; no activation global loads 366; no ds_write 438; no weight LDS-DMA 456; consumers alone 361):
    // the consumers lose their time to the register-staged loads, not to the same bytes moved by LDS-DMA.  A wave instruction copies
    // 64 consecutive halo pixels (16 B each, per-lane source address) to 1 KB of the plane; pixels outside the image read 16 zero
    // bytes.  The buffer is free once the consumers have passed the previous step's barrier, so the copy is issued, waited for
    // (vmcnt(0)) and published by the next barrier within ONE step -- no prefetch across the barrier, none needed (a second register
    // set in the register path measured no gain: the loads land well within a step).
    auto dma_acts = [&](int s) {
      const int chunk = chunk_of(s);
      int n_img, z0, y0, x0;
      tile_origin(tile_of(s), n_img, z0, y0, x0);
      const T* xc = (const T*)p.x + (long long)n_img * p.x_sstride + src_plane(chunk) * plane_stride;
      const unsigned abuf = lds0 + abuf_of(s);
#pragma unroll
      for (int it = 0; it < AIT; ++it) {
        const int base = it * NLT + lw * 64;                                  // first pixel of this wave instruction
        if (base < PLANE / 16) {
          const int pix = lt + it * NLT;
          const int px = pcoord[it] & 255, py = (pcoord[it] >> 8) & 255, pz = pcoord[it] >> 16;
          const int gz = z0 + pz - PADZ, gy = y0 + py - 1, gx = x0 + px - 1;
          const bool ok = pix < NPIX && (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
          const long long goff = (((long long)gz * p.H + gy) * p.W + gx) * 8;
          if (pix < PLANE / 16) {                                             // (2-D: the last instruction would run past the plane)
#pragma unroll
            for (int k = 0; k < CP; ++k) {
              const u32x4* gsrc = ok ? (const u32x4*)(xc + k * plane_stride + goff) : (const u32x4*)zero16;
              const unsigned dst = __builtin_amdgcn_readfirstlane(abuf + k * PLANE + base * 16);
              unsigned keep;
              asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                           : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
            }
          }
        }
      }
    };
    auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    // registers -> LDS buffer s & 1; ACT: z = relu(scale * y + shift), the arithmetic of bn_relu_fwd_kernel, with the
    // per-channel pairs read from LDS (a global load here would sit on the vector-memory counter and its wait would
    // drain the prefetched next step too)
    auto commit = [&](int s, const Staged& r, auto ACT) {
      constexpr bool act = decltype(ACT)::value;
      unsigned char* ab = smem + abuf_of(s);
      const int cfirst = chunk_of(s) * CP * 8;
#pragma unroll
      for (int k = 0; k < CP; ++k) {
        float sc[8], sh[8];
        if (act) {
          const f32x4* ap = (const f32x4*)(smem + off_act) + (cfirst + k * 8) / 2;        // [channel / 8][scale 8 | shift 8]
          const f32x4 s0 = ap[0], s1 = ap[1], h0 = ap[2], h1 = ap[3];
#pragma unroll
          for (int j = 0; j < 4; ++j) { sc[j] = s0[j]; sc[4 + j] = s1[j]; sh[j] = h0[j]; sh[4 + j] = h1[j]; }
        }
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
          const int pix = lt + it * NLT;
          u32x4 v = r.a[it][k];
          asm volatile("" : "+v"(v));          // the load is waited for on EVERY path (a wait only inside the branch below
                                               // leaves it "pending" for the compiler, which then drains younger loads later)
          if (pix < NPIX) {
            const bool ok = (r.ok >> it) & 1u;
            if (act) {
              const V8 in = __builtin_bit_cast(V8, v);
              V8 o;
#pragma unroll
              for (int j = 0; j < 8; ++j) o[j] = from_f32<T>(fmaxf(fmaf(sc[j], to_f32<T>(in[j]), sh[j]), 0.f));
              v = __builtin_bit_cast(u32x4, o);
            }
            *(u32x4*)(ab + k * PLANE + pix * 16) = ok ? v : u32x4{0u, 0u, 0u, 0u};      // padding stays zero AFTER the activation
          }
        }
      }
    };
    // Straight-line steps: every load / copy / commit of a step index past the end is clamped to the last step (its
    // target buffer is not the one being read) instead of being skipped, so the compiler's vector-memory counting sees
    // ONE path and waits for exactly the loads a commit needs, never for the younger prefetch.
    auto run = [&](auto ACT) {
      const int last = nsteps - 1;
#ifdef V4_NOLOAD                                                    // ablation build (never defined by build.sh): consumers alone
      lds_barrier();
      for (int s = 0; s < nsteps; ++s) lds_barrier();
      return;
#endif
      // 3-D only: the 2-D level-0 layers are HBM-bound and need the register path's loads in flight across the barrier (measured:
      // C2's dec0.conv1 at 45 % of the HBM peak by LDS-DMA against 49 % through registers)
      if constexpr (RING2) if (!decltype(ACT)::value && !(p.dbg & 64)) {
        // ---- two steps ahead: in iteration s (consumers on step s) the copy of step s + 2 goes out -- into the buffer of step s - 1,
        // free since the last barrier -- and only then step s + 1 is waited for: vmcnt(K), K = this wave's copy instructions per step
        int K = 0;
#pragma unroll
        for (int it = 0; it < AIT; ++it) K += (it * NLT + lw * 64 < PLANE / 16) ? CP : 0;
        auto landed_but_last = [&]() {
          if (K == CP * AIT) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(CP * AIT) : "memory");
          else if (K == CP * (AIT - 1)) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(CP * (AIT - 1)) : "memory");
          else landed();
        };
        if (!WS) dma_weights(0, 0);
        dma_acts(0);
        if (last > 0) { dma_acts(1); landed_but_last(); } else landed();
        lds_barrier();
        for (int s = 0; s < nsteps; ++s) {
          if (!WS && s + 1 < nsteps) dma_weights(s + 1, (s + 1) & 1);      // (in front of the halo copy: the wait below leaves only that one in flight)
          if (s + 2 < nsteps) { dma_acts(s + 2); landed_but_last(); } else landed();
          lds_barrier();
        }
        return;
      }
      if ((ND == 3 || SPL || (NP2 && !WS)) && !decltype(ACT)::value && !(p.dbg & 64)) {      // (IUNET_V4_DBG=64: the register path for every launch -- A/B switch)
        // ---- everything by LDS-DMA: in iteration s (consumers on step s) the buffers of step s + 1 are filled ----
        if (!WS) dma_weights(0, 0);
        dma_acts(0);
        landed();
        lds_barrier();
        if constexpr (SPL) {
          // step s + 1 = part 0: lo halo -> buffer 0 (last read two steps ago) + w_hi -> weight buffer 0 (read until step s - 1);
          // part 1: hi halo -> buffer 1 (read until step s - 1), w_hi stays; part 2: w_lo -> weight buffer 1, the hi halo stays
          for (int s = 0; s < nsteps; ++s) {
            if (s + 1 < nsteps) {
              const int pn = part_of(s + 1);
              if (pn != 1) dma_weights(s + 1, pn == 2 ? 1 : 0);
              if (pn != 2) dma_acts(s + 1);
            }
            landed();
            lds_barrier();
          }
          return;
        }
        if (PAIR) {
          for (int s = 0; s + 1 < nsteps; s += 2) {
            // steps s, s + 1: tiles A, B on the weights of one chunk (buffer = chunk parity); the next chunk's weights go half in each
            const bool more = s + 2 < nsteps;                          // (the last pair prefetches no weights: its clamped target is in use)
            const int nb = chunk_of(min(s + 2, last)) & 1;
            if (more) dma_weights(s + 2, nb, 0, WIT / 2);
            dma_acts(s + 1);
            landed();
            lds_barrier();
            if (more) { dma_weights(s + 2, nb, WIT / 2, WIT); dma_acts(s + 2); }
            landed();
            lds_barrier();
          }
        } else {
          for (int s = 0; s < nsteps; ++s) {
            if (s + 1 < nsteps) {                                      // (the last step has nothing to prepare)
              if (!WS) dma_weights(s + 1, (s + 1) & 1);
              dma_acts(s + 1);
            }
            landed();
            lds_barrier();
          }
        }
        return;
      }
      Staged r;
      if (!WS) dma_weights(0, 0);
      load(0, r);
      commit(0, r, ACT);
      load(min(1, last), r);
      if (!WS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the first weight copy has landed
      lds_barrier();
      if (WS) {
        // one register set: step s + 1 is written to LDS, then the loads of step s + 2 are issued into the same registers
        // and stay in flight over the barrier and the consumers' whole next step
        for (int s = 0; s < nsteps; ++s) {
          commit(s + 1, r, ACT);                                       // (buffer (s + 1) & 1: garbage after the last step, unread)
          load(min(s + 2, last), r);
          lds_barrier();
        }
      } else if (PAIR) {
        // steps s (tile A) and s + 1 (tile B) share the weights of one chunk (buffer = chunk parity); the next chunk's weights are
        // copied half in each of the two steps into the other buffer (last read two steps ago).  Each half is issued BEFORE the
        // step's activation loads, so the wait inside the following commit retires it before the step's barrier.
        Staged r2;
        for (int s = 0; s + 1 < nsteps; s += 2) {
          const bool more = s + 2 < nsteps;                            // (the last pair prefetches no weights: its clamped target is in use)
          const int nb = chunk_of(min(s + 2, last)) & 1;
          if (more) dma_weights(s + 2, nb, 0, WIT / 2);
          load(min(s + 2, last), r2);
          commit(s + 1, r, ACT);
          lds_barrier();
          if (more) dma_weights(s + 2, nb, WIT / 2, WIT);
          load(min(s + 3, last), r);
          commit(s + 2, r2, ACT);
          lds_barrier();
        }
      } else {
        // weights stream too (the loaders bound these layers): the weights of step s + 1 go by LDS-DMA into the buffer the
        // consumers released at the last barrier; two register sets for the activations, the loads of step s + 2 are
        // issued BEFORE step s + 1 is waited for and written, so two steps of loads are in flight
        Staged r2;
        int s = 0;
        for (; s + 1 < nsteps; s += 2) {
          dma_weights(s + 1, 1);
          load(min(s + 2, last), r2);
          commit(s + 1, r, ACT);
          lds_barrier();
          if (!NP3 || s + 2 <= last) dma_weights(min(s + 2, last), 0);     // (NP: a clamped copy would land in the region being read)
          load(min(s + 3, last), r);
          commit(s + 2, r2, ACT);
          lds_barrier();
        }
        if (s < nsteps) lds_barrier();                                 // odd step count: the last step has nothing to prefetch
      }
    };
    if (p.in_scale != nullptr) run(std::true_type{}); else run(std::false_type{});
    if (p.stats != nullptr) lds_barrier();            // the consumers' final statistics reduction
    return;
  }

  // ==================================================================== consumer waves
#ifdef V4_PRIO      // A/B (never defined by build.sh): static issue priority of the MFMA waves over the loader waves (1), and of the
                    // younger consumer half over the older one (2)   [MI355X_MICROARCH.md, two waves per SIMD, item 4]
  if (V4_PRIO >= 2 && wave >= NCW / 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
#endif
  const int l15 = lane & 15, q = lane >> 4;
  int col_off[NCMB];
#pragma unroll
  for (int c = 0; c < NCMB; ++c) {
    const int col = min(2 * c + (q >> 1), NCOL - 1);           // the missing partner re-reads a valid column (zero weights)
    const int dz = ND == 3 ? col / 3 : 0, dx = ND == 3 ? col % 3 : col;
    col_off[c] = (dz * PY * PX + dx) * 16;
  }
  const int f0 = wave * NI;                                    // first fragment of this wave: fragment f = row * FX + x half
  const int row_first = f0 / FX;                               // first tile row (z * TY + y)
  const int rbase = (q & 1) * PLANE + ((((row_first / TY) * PY + (row_first % TY)) * PX) + l15) * 16;
  float bias_r[SPL ? 1 : 8];
  if constexpr (!SPL) {
#pragma unroll
    for (int j = 0; j < 8; ++j) bias_r[j] = (p.epi != 0) ? p.bias[cob * 32 + 8 * q + j] : 0.f;
  }

  constexpr int NACC = PAIR ? 2 : 1;                           // accumulator sets: one per tile in flight
  f32x4 acc[NACC][2][NI];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NI; ++n) acc[t][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  float stat_acc = 0.f;      // BatchNorm partial sums over all tiles of this workgroup: lane (q, l15) holds value l15 = which * 8 + j
  V8 bw_yv[BW ? NI : 1];     // bw: the producer layer's raw output at this wave's output voxels, fetched one step ahead of the epilogue
  auto bw_prefetch = [&](int s) {
    const int chunk = chunk_of(s);
    if constexpr (BW) if (chunk == nchunk - 1) {
      int n_img, z0, y0, x0;
      tile_origin(tile_of(s), n_img, z0, y0, x0);
      const T* yp = (const T*)p.bw_y + (long long)n_img * p.bw_y_ss + (long long)(cob * 4 + q) * plane_stride;
#pragma unroll
      for (int n = 0; n < NI; ++n) {
        const int f = f0 + n, row = f / FX;
        const int gz = min(z0 + (ND == 3 ? row / TY : 0), p.D - 1), gy = min(y0 + row % TY, p.H - 1), gx = min(x0 + (f % FX) * 16 + l15, p.W - 1);
        bw_yv[n] = *(const V8*)(yp + (((long long)gz * p.H + gy) * p.W + gx) * 8);
      }
    }
  };
  lds_barrier();                                             // step 0 (and the resident weights) are in LDS
#ifdef IUNET_STAMPS
  const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif

  // Fragment registers: two sets (R: activation rows, A: weights), group t of the launch-wide sequence uses set t & 1.  The
  // software pipeline runs ACROSS the steps: the reads of step s + 1's first group are issued during the last group of step
  // s, and the step's one barrier sits in front of that last group -- there every read of step s's buffers has landed
  // (lgkmcnt(0)), so the loaders may overwrite them, and the loaders have finished step s + 1's buffers.  (With the barrier
  // at the end of the step both consumer waves of a SIMD refilled their pipeline from empty at the same moment, every step.)
  V8 R[2][FX][NR + 2], A[2][3][2];
  auto step_ptrs = [&](int s, const unsigned char*& ab, const unsigned char*& wl) {
    const int chunk = chunk_of(s);
    ab = smem + abuf_of(s) + rbase;
    wl = smem + OFF_W + (NP3 ? ((chunk & 1) ? WE : 0) : (WS ? chunk : PAIR ? (chunk & 1) : SPL ? (chunk - 3 * (chunk / 3) == 2) : (s & 1)) * WSTEP) + lane * 16;
  };
  // NP, odd steps: the cross group.  Lanes q >> 1 = 0 read column 8 of the PREVIOUS step's halo buffer, q >> 1 = 1 of this step's;
  // the weights follow the four regular pairs in the odd region.
  auto load_cross = [&](const unsigned char* ab, const unsigned char* abp, const unsigned char* wl, auto BUF) {
    constexpr int b = decltype(BUF)::value;
    const unsigned char* ax = (q >> 1) ? ab : abp;
#pragma unroll
    for (int xh = 0; xh < FX; ++xh)
#pragma unroll
      for (int r = 0; r < NR + 2; ++r) R[b][xh][r] = *(const V8*)(ax + (r * PX + xh * 16) * 16 + col_off[NCMB - 1]);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      A[b][dy][0] = *(const V8*)(wl + WE + (dy * 2 + 0) * 1024);
      A[b][dy][1] = *(const V8*)(wl + WE + (dy * 2 + 1) * 1024);
    }
  };
  auto load_group = [&](const unsigned char* ab, const unsigned char* wl, int g, auto BUF) {
    constexpr int b = decltype(BUF)::value;
    if constexpr (NP2) {
      // groups 0, 1: columns 0 / 1 (lanes q >> 1) of sub-chunk g; group 2, the cross group: column 2 of sub-chunk q >> 1
      const int hoff = g == 2 ? (q >> 1) * 2 * PLANE + col_off[NCMB - 1] : g * 2 * PLANE + col_off[0];
#pragma unroll
      for (int xh = 0; xh < FX; ++xh)
#pragma unroll
        for (int r = 0; r < NR + 2; ++r) R[b][xh][r] = *(const V8*)(ab + hoff + (r * PX + xh * 16) * 16);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        A[b][dy][0] = *(const V8*)(wl + ((g * 3 + dy) * 2 + 0) * 1024);
        A[b][dy][1] = *(const V8*)(wl + ((g * 3 + dy) * 2 + 1) * 1024);
      }
      return;
    }
    const int h = g / NCMB, c = g - h * NCMB;
#pragma unroll
    for (int xh = 0; xh < FX; ++xh)
#pragma unroll
      for (int r = 0; r < NR + 2; ++r) R[b][xh][r] = *(const V8*)(ab + h * 2 * PLANE + (r * PX + xh * 16) * 16 + col_off[c]);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      A[b][dy][0] = *(const V8*)(wl + h * WBYTES + ((c * 3 + dy) * 2 + 0) * 1024);
      A[b][dy][1] = *(const V8*)(wl + h * WBYTES + ((c * 3 + dy) * 2 + 1) * 1024);
    }
  };
  // the ReLU as ONE v_max against a wave-uniform floor, 0 or -inf (the run-time select `epi == 2 ? max(r, 0) : r` cost a v_cndmask per
  // output value on top; two compiled epilogues cost the 128- / 168-register variants 640 B of scratch).  Without ReLU a NaN leaves as
  // -inf: still not finite for every check downstream (with ReLU it has always left as 0)
  // cross-step pipeline: the variants with resident weights (168-register cap); BW keeps its yp fragments instead; the 2-D cross-pair
  // step has an odd group count (both parities of the pipeline instantiated: 76 B of scratch per lane) and restarts every step too
  constexpr bool XSTEP = WS && !BW && !NP2;
  auto tile_epilogue = [&](int s, auto TSET) {
    constexpr int ts = decltype(TSET)::value;
    const int chunk = chunk_of(s);
    if (chunk == nchunk - 1) {
      const float relu_floor = p.epi == 2 ? 0.f : -__builtin_inff();      // (made here from the scalar p.epi: held across the step loop it took a register)
      // ---- epilogue of this tile ----
      int n_img, z0, y0, x0;
      tile_origin(tile_of(s), n_img, z0, y0, x0);
      T* yout = (T*)p.y + (long long)n_img * p.y_sstride;
      float s_sum[8], s_sq[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { s_sum[j] = 0.f; s_sq[j] = 0.f; }
#pragma unroll
      for (int n = 0; n < NI; ++n) {
        const int f = f0 + n, row = f / FX;
        const int gz = z0 + (ND == 3 ? row / TY : 0), gy = y0 + row % TY, gx = x0 + (f % FX) * 16 + l15;
        const bool ok = gz < p.D && gy < p.H && gx < p.W;
        float vals[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { vals[j] = acc[ts][0][n][j]; vals[4 + j] = acc[ts][1][n][j]; }
        if (p.stats != nullptr && ok && !bw) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { s_sum[j] += vals[j]; s_sq[j] += vals[j] * vals[j]; }
        }
        V8 o;
        [[maybe_unused]] V8 o_lo;
        if constexpr (SPL) {
          const f32x4* sp = (const f32x4*)(smem + off_red) + 2 * q;           // [oscale 32 | bias 32]: this lane's 8 channels of each
          const f32x4 c0 = sp[0], c1 = sp[1], b0 = sp[8], b1 = sp[9];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float r = fmaf(vals[j], j < 4 ? c0[j & 3] : c1[j & 3], j < 4 ? b0[j & 3] : b1[j & 3]);
            if constexpr (XSTEP) { if (p.epi == 2) r = fmaxf(r, 0.f); } else r = fmaxf(r, relu_floor);      // (the cross-step pipeline's epilogue sits at its register cap: the floor form spilled 20 B there)
            T hi, lo;
            split16<T>(r, hi, lo);
            o[j] = hi; o_lo[j] = lo;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float r = vals[j] + bias_r[j];
            if constexpr (XSTEP) { if (p.epi == 2) r = fmaxf(r, 0.f); } else r = fmaxf(r, relu_floor);      // (the cross-step pipeline's epilogue sits at its register cap: the floor form spilled 20 B there)
            o[j] = from_f32<T>(r);
          }
        }
        if constexpr (BW) if (ok) {
          const f32x4* bp = (const f32x4*)(smem + off_bw) + 2 * q;          // [param][32]: this lane's 8 channels of each
          const f32x4 m0 = bp[0], m1 = bp[1], i0 = bp[8], i1 = bp[9], c0 = bp[16], c1 = bp[17], h0 = bp[24], h1 = bp[25];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float mu = j < 4 ? m0[j & 3] : m1[j & 3], is = j < 4 ? i0[j & 3] : i1[j & 3];
            const float sc = j < 4 ? c0[j & 3] : c1[j & 3], sh = j < 4 ? h0[j & 3] : h1[j & 3];
            const float yy = to_f32<T>(bw_yv[n][j]);
            const float zz = to_f32<T>(from_f32<T>(fmaf(sc, yy, sh)));       // the stored activation (bn_bwd_reduce_kernel)
            const float d = zz > 0.f ? to_f32<T>(o[j]) : 0.f;                // the stored gradient
            s_sum[j] += d;
            s_sq[j] += d * (yy - mu) * is;
          }
        }
        if (ok && !(p.dbg & 4)) *(V8*)(yout + (long long)(cob * 4 + q) * plane_stride + (((long long)gz * p.H + gy) * p.W + gx) * 8) = o;
        if constexpr (SPL) {
          if (ok && !(p.dbg & 4)) *(V8*)(yout + (long long)(p.y_lo + cob * 4 + q) * plane_stride + (((long long)gz * p.H + gy) * p.W + gx) * 8) = o_lo;
          if (ok && p.sat != nullptr) x2_note_saturation(p.sat, o);
        }
        acc[ts][0][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[ts][1][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (p.stats != nullptr) {
        // this tile's 16 partial sums per 16-lane group, reduce-scattered over the x lanes (15 exchanges): lane l15 ends
        // up with value l15 and adds it to its running total -- no barrier, no LDS, one register of state
        float vals[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) { vals[j] = s_sum[j]; vals[8 + j] = s_sq[j]; }
#pragma unroll
        for (int h = 8; h >= 1; h >>= 1) {
          const bool up = (l15 & h) != 0;
#pragma unroll
          for (int i = 0; i < h; ++i) {
            const float send = up ? vals[i] : vals[i + h], keep = up ? vals[i + h] : vals[i];
            vals[i] = keep + __shfl_xor(send, h);
          }
        }
        stat_acc += vals[0];
      }
    }
  };

  // The MFMAs of group g on fragment set b, with the LDS reads issued just before them spread between the MFMAs.
  auto group_mfmas = [&](auto BUF, bool reads_pending, auto TSET) {
    constexpr int b = decltype(BUF)::value, ts = decltype(TSET)::value;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int n = 0; n < NI; ++n) {
        acc[ts][0][n] = mfma16<T>(A[b][dy][0], R[b][n % FX][n / FX + dy], acc[ts][0][n]);
        acc[ts][1][n] = mfma16<T>(A[b][dy][1], R[b][n % FX][n / FX + dy], acc[ts][1][n]);
      }
    if (reads_pending) {
      constexpr int MPR = (3 * NI * 2) / NRD > 0 ? (3 * NI * 2) / NRD : 1;      // MFMAs per LDS read in the interleave
#pragma unroll
      for (int i = 0; i < NRD; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // one LDS read
        __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);     // MPR MFMAs
      }
    }
    __builtin_amdgcn_sched_barrier(0);                           // nothing moves across the group boundary
  };
  // The groups of one step of the cross-step pipeline; PAR = fragment set of its first group.
  using TS0 = std::integral_constant<int, 0>;
  using TS1 = std::integral_constant<int, PAIR ? 1 : 0>;
  auto step_groups = [&](int s, auto PAR) {
    constexpr int par = decltype(PAR)::value;
    using B0 = std::integral_constant<int, par>;                       // set of the even groups of this step
    using B1 = std::integral_constant<int, par ^ 1>;
    using BL = std::integral_constant<int, (NGRP - 1 + par) & 1>;      // set of the last group
    using BN = std::integral_constant<int, (NGRP + par) & 1>;          // set of the next step's first group
    const unsigned char *ab, *wl, *abn, *wln;
    step_ptrs(s, ab, wl);
    step_ptrs(min(s + 1, nsteps - 1), abn, wln);           // (after the last step: a harmless re-read of its own buffers)
    bw_prefetch(s);
#pragma unroll
    for (int g = 0; g + 1 < NGRP; ++g) {
      if ((g & 1) == 0) { load_group(ab, wl, g + 1, B1{}); group_mfmas(B0{}, true, TS0{}); }
      else              { load_group(ab, wl, g + 1, B0{}); group_mfmas(B1{}, true, TS0{}); }
    }
    lds_barrier();                   // this step's buffers are read, the next step's are filled
    load_group(abn, wln, 0, BN{});
    group_mfmas(BL{}, true, TS0{});
    tile_epilogue(s, TS0{});
  };
  if (!(p.dbg & 2)) {
    if constexpr (XSTEP) {
      const unsigned char *ab0, *wl0;
      step_ptrs(0, ab0, wl0);
      load_group(ab0, wl0, 0, std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NGRP % 2 == 1) {           // odd group count: the first group's fragment set alternates from step to step
        for (int s = 0; s < nsteps; s += 2) {
          step_groups(s, std::integral_constant<int, 0>{});
          if (s + 1 < nsteps) step_groups(s + 1, std::integral_constant<int, 1>{});
        }
      } else {
        for (int s = 0; s < nsteps; ++s) step_groups(s, std::integral_constant<int, 0>{});
      }
    } else {
      // streamed-weight variants (128-register cap: accumulators + two fragment sets fill it; a set that stays live over the
      // barrier and the epilogue spills): the pipeline restarts every step, the barrier closes the step
      using B0 = std::integral_constant<int, 0>;
      using B1 = std::integral_constant<int, 1>;
      auto one_step = [&](int s, auto TSET, auto ODD) {
        // NP: an even step has the four regular column pairs, an odd step those plus the cross pair (ODD: compile-time parity)
        constexpr int NG = NP3 ? (decltype(ODD)::value ? NCMB : NCMB - 1) : NGRP;
        const unsigned char *ab, *wl, *abp = nullptr;
        step_ptrs(s, ab, wl);
        if constexpr (NP3) abp = smem + abuf_of(s - 1) + rbase;
        bw_prefetch(s);
        load_group(ab, wl, 0, B0{});
        __builtin_amdgcn_sched_barrier(0);
        auto fetch = [&](int g, auto BUF) {
          if (NP3 && g == NCMB - 1) load_cross(ab, abp, wl, BUF); else load_group(ab, wl, g, BUF);
        };
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if ((g & 1) == 0) { if (g + 1 < NG) fetch(g + 1, B1{}); group_mfmas(B0{}, g + 1 < NG, TSET); }
          else              { if (g + 1 < NG) fetch(g + 1, B0{}); group_mfmas(B1{}, g + 1 < NG, TSET); }
        }
        tile_epilogue(s, TSET);
        lds_barrier();               // consumers are done with this step's buffers, the loaders have filled the others
      };
      using EVEN = std::integral_constant<int, 0>;
      using ODDS = std::integral_constant<int, 1>;
      if constexpr (PAIR) {
        for (int s = 0; s + 1 < nsteps; s += 2) { one_step(s, TS0{}, EVEN{}); one_step(s + 1, TS1{}, EVEN{}); }     // tile A, tile B of the same chunk
      } else if constexpr (NP3) {
        for (int s = 0; s + 1 < nsteps; s += 2) { one_step(s, TS0{}, EVEN{}); one_step(s + 1, TS0{}, ODDS{}); }     // chunk parity = step parity (nchunk is even)
      } else {
        for (int s = 0; s < nsteps; ++s) one_step(s, TS0{}, EVEN{});
      }
    }
  } else {
    for (int s = 0; s < nsteps; ++s) lds_barrier();            // profiling only: nothing is computed or stored
  }
#ifdef IUNET_STAMPS
  if (wave == 0 && lane == 0 && cob == 0 && blockIdx.x < 512) {
    const unsigned long long st1 = __builtin_amdgcn_s_memtime(), sr1 = __builtin_amdgcn_s_memrealtime();
    g_v4_stamps[blockIdx.x * 4 + 0] = st1 - st0; g_v4_stamps[blockIdx.x * 4 + 1] = sr1 - sr0; g_v4_stamps[blockIdx.x * 4 + 2] = nsteps;
  }
#endif
  if (p.stats != nullptr) {
    // the consumer waves' totals meet in LDS (wave order: deterministic) and become this workgroup's statistics row
    float* red = (float*)(smem + off_red);                     // [consumer waves][4 q][16]
    red[(wave * 4 + q) * 16 + l15] = stat_acc;
    lds_barrier();
    if (tid < 64) {
      const int c = tid >> 1, which = tid & 1;                 // c = 8 g + j
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < NCW; ++w) sum += red[(w * 4 + (c >> 3)) * 16 + which * 8 + (c & 7)];
      p.stats[((long long)(blockIdx.z * gridDim.x + blockIdx.x) * p.Cout + cob * 32 + c) * 2 + which] = sum;
    }
  }
}

template <typename T, int ND, bool WS, bool SMALL, bool BW = false, bool PAIR = false, bool NP = false, bool SPL = false>
int launch_v4(ConvV4Params p, hipStream_t stream) {
  using TL = V4Tile<ND, SMALL>;
  constexpr int NPIX = (TL::TZ + 2 * TL::PADZ) * (TL::TY + 2) * (TL::TX + 2);
  constexpr int PLANE = ((NPIX * 16 + 255) / 256) * 256;
  constexpr int WSTEP = (NP && ND == 2) ? 3 * 3 * 2 * 1024 : TL::S16 * ((TL::NCOL + 1) / 2) * 3 * 2 * 1024;
  const int lds = (NP && ND == 3) ? 3 * 2 * PLANE + (24576 + 30720) + 2048 + (p.in_scale != nullptr ? p.Cin * 8 : 0) + 512
                     : ((NP && ND == 2 && !SPL && (WS ? (V4_RING2 & 1) != 0 : (V4_RING2 & 2) != 0)) ? 3 : 2) * 2 * TL::S16 * PLANE + (WS ? p.Cin / (16 * TL::S16) : 2) * WSTEP + 2048
                           + ((SPL || (NP && ND == 2 && p.in_scale == nullptr)) ? 0 : p.Cin * 8) + 512;      // (SPL: no fused input activation, and Cin is the 3x virtual count)
  IUNET_REQUIRE(lds <= 160 * 1024, "conv3 layout 3: %d bytes of LDS (a fused input activation fits up to 192 input channels)", lds);
  IUNET_SET_MAX_LDS((conv3_v4_kernel<T, ND, WS, SMALL, BW, PAIR, NP, SPL>), lds);
  p.tilesZ = (p.D + TL::TZ - 1) / TL::TZ; p.tilesY = (p.H + TL::TY - 1) / TL::TY; p.tilesX = (p.W + TL::TX - 1) / TL::TX;
  const int ncob = p.Cout / 32;
  // one workgroup per CU: 8 XCDs x (bz x by x bx) brick slots per Cout tile
  iunet_brick_shape(ND, ncob, p.tilesZ, p.tilesY, p.tilesX, &p.bz, &p.by, &p.bx);
  p.nbz = (p.tilesZ + p.bz - 1) / p.bz; p.nby = (p.tilesY + p.by - 1) / p.by; p.nbx = (p.tilesX + p.bx - 1) / p.bx;
  // slot groups: the brick table's slot count over this (possibly clamped) brick's, while every group still gets a brick per XCD
  int groups = 1;
  const int nz = p.per_sample ? p.N : 1;                 // per-sample statistics: one sample per grid z, N = 1 inside
  if (p.per_sample) p.N = 1;
  const long long nbricks = (long long)p.N * p.nbz * p.nby * p.nbx;
  if (!PAIR) {
    groups = iunet_conv3_v4_stats_parts(ND, p.Cout) / 8 / (p.bz * p.by * p.bx);
    while (groups > 1 && nbricks / 8 < groups) groups >>= 1;
    if (groups < 1) groups = 1;
  }
  const int gx = 8 * p.bz * p.by * p.bx * groups;
  if (p.query_rows != nullptr) {      // per-sample statistics need every XCD to hold a brick of every sample (else the launch would idle CUs: the caller's own pass is cheaper)
    *p.query_rows = (!PAIR && !SPL && nbricks >= 8) ? gx : 0;
    return IUNET_OK;
  }
  IUNET_REQUIRE(!p.per_sample || (!PAIR && !SPL && nbricks >= 8 && p.stats != nullptr), "conv3 layout 2 / 3: per-sample statistics are not available for this launch (iunet_conv3_sample_stats_rows says 0)");
  if (p.stats != nullptr && !p.per_sample) {
    // the caller reduces iunet_conv3_v4_stats_parts rows (the brick table's slot count); a brick clamped to a small tile grid
    // launches fewer workgroups: the rows nobody writes are zeroed
    const int rows = iunet_conv3_v4_stats_parts(ND, p.Cout);
    if (gx < rows)
      IUNET_CHECK_HIP(hipMemsetAsync(p.stats + (long long)gx * p.Cout * 2, 0, (size_t)(rows - gx) * p.Cout * 2 * sizeof(float), stream));
  }
  hipLaunchKernelGGL((conv3_v4_kernel<T, ND, WS, SMALL, BW, PAIR, NP, SPL>), dim3(gx, ncob, nz), dim3(TL::NCW * 64 + v4_loader_threads(ND, WS, PAIR, SPL, BW, NP)), lds, stream, p);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

}  // namespace

// rows of BatchNorm partial sums a layout-2 launch writes: one per workgroup of a Cout tile
int iunet_conv3_v4_stats_parts(int nd, int Cout) {
  (void)nd;                                            // 2-D and 3-D bricks have the same number of slots
  const int ncob = Cout / 32;
  return 8 * (ncob == 1 ? 32 : ncob == 2 ? 16 : ncob <= 4 ? 8 : 4);
}

// does this launch walk its tiles in pairs (PAIR variant: one weight stream per two tiles)?  3-D, big tiles, streamed weights, no
// fused BatchNorm-backward sums, and every XCD's share of the bricks even (nbricks a multiple of 16).  IUNET_V4_PAIR=0: A/B switch.
int iunet_conv3_v4_pairs(int nd, int N, int D, int H, int W, int Cin, int Cout, int bw) {
  static const int pair_on = getenv("IUNET_V4_PAIR") ? atoi(getenv("IUNET_V4_PAIR")) : 1;
  if (nd != 3 || bw || !pair_on || Cin <= 32) return 0;                      // Cin <= 32: resident weights
  const int tz = (D + 3) / 4, ty = (H + 7) / 8, tx = (W + 15) / 16;
  if ((long long)N * tz * ty * tx * (Cout / 32) < 128) return 0;              // the half-size tile variant
  int bz, by, bx;
  iunet_brick_shape(3, Cout / 32, tz, ty, tx, &bz, &by, &bx);
  const long long nbricks = (long long)N * ((tz + bz - 1) / bz) * ((ty + by - 1) / by) * ((tx + bx - 1) / bx);
  return nbricks % 16 == 0;
}

int iunet_conv3_v4_launch(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride, const void* wpk,
                          const float* bias, float* stats, int N, int D, int H, int W, int Cin, int Cout, int epi,
                          const float* in_scale, const float* in_shift, hipStream_t stream, const void* bw_y, long long bw_y_ss,
                          const float* const* bw_par, int compact, int per_sample, int* query_rows) {
  IUNET_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, "conv3 layout 2: Cin %% 32, Cout %% 32 (got %d -> %d)", Cin, Cout);
  ConvV4Params p;
  p.x = x; p.x_sstride = x_sstride; p.y = y; p.y_sstride = y_sstride; p.wpk = wpk; p.bias = bias; p.stats = stats;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.epi = epi;
  p.in_scale = in_scale; p.in_shift = in_shift;
  p.bw_y = bw_y; p.bw_y_ss = bw_y_ss;
  p.bw_mean = bw_y ? bw_par[0] : nullptr; p.bw_invstd = bw_y ? bw_par[1] : nullptr;
  p.bw_scale = bw_y ? bw_par[2] : nullptr; p.bw_shift = bw_y ? bw_par[3] : nullptr;
  IUNET_REQUIRE(bw_y == nullptr || stats != nullptr, "conv3 layout 2: the fused BatchNorm-backward sums need a statistics buffer");
  p.tilesZ = p.tilesY = p.tilesX = 0;
  p.bz = p.by = p.bx = p.nbz = p.nby = p.nbx = 0;
  p.split_nc = 0; p.x_lo = p.y_lo = 0; p.oscale = nullptr; p.sat = nullptr;
  p.per_sample = per_sample; p.query_rows = query_rows;
#ifdef IUNET_ABLATE      // result-destroying profiling switches exist in diagnostic builds only (tools/ab_build.sh <file> -DIUNET_ABLATE): ADVICE r3
  static const int dbg = getenv("IUNET_V4_DBG") ? atoi(getenv("IUNET_V4_DBG")) : 0;
#else
  static const int dbg = 0;
#endif
  p.dbg = dbg;
  // weights resident in LDS for the whole launch when they fit beside the two activation buffers
  const bool ws = nd == 3 ? Cin <= 32 : Cin <= 64;
  // 3-D launches whose 4 x 8 x 16 tiles would occupy fewer than half of the CUs run on the half-size tile (measured on the
  // 16^3 level: 1.4-1.6x faster there; at 128 of 256 CUs the doubled weight streaming costs more than the idle CUs)
  const long long big_tiles = (long long)(per_sample ? 1 : N) * ((D + 3) / 4) * ((H + 7) / 8) * ((W + 15) / 16);      // (per-sample statistics: the launch is one sample's grid, N times)
  const bool small = nd == 3 && !ws && big_tiles * (Cout / 32) < 128;
  if (compact) {      // layout 3: the compact operator, padding-free step (streamed weights, big tiles, no fused BatchNorm-backward sums so far)
    // (every grid size: a layer must not change its summation order with the number of blocks in a launch -- the sharded prediction
    //  is byte-identical across world sizes)
    IUNET_REQUIRE((bw_y == nullptr || (nd == 2 && Cin <= 64)) && (nd == 2 || !ws), "conv3 layout 3: fused BatchNorm-backward sums in 2-D up to 64 input channels only; 3-D: Cin > 32");
    if (nd == 2) {    // the cross-pair step: resident weights up to 64 input channels, streamed beyond
      // (the two variants read the same operator and add in the same order: which one runs is a speed choice.  IUNET_V4_WS2D = the
      //  largest Cin that keeps its weights resident when the input needs no arithmetic -- A/B switch)
      static const int ws2d = getenv("IUNET_V4_WS2D") ? atoi(getenv("IUNET_V4_WS2D")) : 64;
      const bool ws = (in_scale != nullptr || bw_y != nullptr) ? Cin <= 64 : Cin <= ws2d;
      if (ws && bw_y != nullptr) return dtype == 0 ? launch_v4<f16, 2, true, false, true, false, true>(p, stream) : launch_v4<bf16, 2, true, false, true, false, true>(p, stream);
      if (ws) return dtype == 0 ? launch_v4<f16, 2, true, false, false, false, true>(p, stream) : launch_v4<bf16, 2, true, false, false, false, true>(p, stream);
      return dtype == 0 ? launch_v4<f16, 2, false, false, false, false, true>(p, stream) : launch_v4<bf16, 2, false, false, false, false, true>(p, stream);
    }
    if (small) return dtype == 0 ? launch_v4<f16, 3, false, true, false, false, true>(p, stream) : launch_v4<bf16, 3, false, true, false, false, true>(p, stream);
    return dtype == 0 ? launch_v4<f16, 3, false, false, false, false, true>(p, stream) : launch_v4<bf16, 3, false, false, false, false, true>(p, stream);
  }
  const bool pair = !per_sample && iunet_conv3_v4_pairs(nd, N, D, H, W, Cin, Cout, bw_y != nullptr) != 0;
  if (pair) return dtype == 0 ? launch_v4<f16, 3, false, false, false, true>(p, stream) : launch_v4<bf16, 3, false, false, false, true>(p, stream);
#define V4_GO(TT, BWV) (nd == 3 ? (ws ? launch_v4<TT, 3, true, false, BWV>(p, stream)                                          \
                                      : (small ? launch_v4<TT, 3, false, true, BWV>(p, stream) : launch_v4<TT, 3, false, false, BWV>(p, stream))) \
                                : (ws ? launch_v4<TT, 2, true, false, BWV>(p, stream) : launch_v4<TT, 2, false, false, BWV>(p, stream)))
  if (bw_y != nullptr) return dtype == 0 ? V4_GO(f16, true) : V4_GO(bf16, true);
  return dtype == 0 ? V4_GO(f16, false) : V4_GO(bf16, false);
#undef V4_GO
}

// pack mode (iunet_pack_conv3) of the split conv's virtual operator: 2 = the padded K16 order, 6 = the compact order -- 2-D launches
// run the cross-pair step on it (three k-groups per 32-channel step instead of four).  IUNET_X2_NP2=0: A/B switch back to the padded order.
int iunet_conv3_v4_x2_pack_mode(int nd) {
  static const int np2 = getenv("IUNET_X2_NP2") ? atoi(getenv("IUNET_X2_NP2")) : 1;
  return nd == 2 && np2 ? 6 : 2;
}

// fp16x2 split-precision forward (split16.hip): Cin real input channels; x / y are views of Cin / 8 (Cout / 8) hi planes with the lo
// planes x_lo / y_lo planes further on; wpk = the K16 order (iunet_conv3_v4_x2_pack_mode: padded in 3-D, compact in 2-D) of the VIRTUAL operator
// [Cout][3 Cin][taps] = [w_hi | w_hi | w_lo]; y = split(relu?(acc * oscale + bias)).
int iunet_conv3_v4_x2_launch(int nd, const void* x, long long x_sstride, int x_lo, void* y, long long y_sstride, int y_lo, const void* wpk,
                             const float* oscale, const float* bias, int N, int D, int H, int W, int Cin, int Cout, int epi,
                             int* sat, hipStream_t stream) {
  IUNET_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, "conv3 x2: Cin %% 32, Cout %% 32 (got %d -> %d)", Cin, Cout);
  ConvV4Params p;
  p.x = x; p.x_sstride = x_sstride; p.y = y; p.y_sstride = y_sstride; p.wpk = wpk; p.bias = bias; p.stats = nullptr;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = 3 * Cin; p.Cout = Cout; p.epi = epi;
  p.in_scale = p.in_shift = nullptr;
  p.bw_y = nullptr; p.bw_y_ss = 0; p.bw_mean = p.bw_invstd = p.bw_scale = p.bw_shift = nullptr;
  p.tilesZ = p.tilesY = p.tilesX = 0;
  p.bz = p.by = p.bx = p.nbz = p.nby = p.nbx = 0;
  p.dbg = 0;
  p.split_nc = Cin / (nd == 3 ? 16 : 32); p.x_lo = x_lo; p.y_lo = y_lo; p.oscale = oscale; p.sat = sat;
  p.per_sample = 0; p.query_rows = nullptr;
  if (nd == 2) {
    // the cross-pair step (NP2, the compact operator of pack mode 6) for every 2-D launch: one summation order per layer whatever the grid
    if (iunet_conv3_v4_x2_pack_mode(2) == 6) return launch_v4<f16, 2, false, false, false, false, true, true>(p, stream);
    return launch_v4<f16, 2, false, false, false, false, false, true>(p, stream);
  }
  // the tile size follows the grid as in the 16-bit launch (the operator order -- the padded K16 one -- does not: a layer keeps one
  // summation order whatever the grid).  The compact order's split instantiation spills its fragment arrays (640 B per lane) and is not built.
  const long long big_tiles = (long long)N * ((D + 3) / 4) * ((H + 7) / 8) * ((W + 15) / 16);
  const bool small = big_tiles * (Cout / 32) < 128;
  return small ? launch_v4<f16, 3, false, true, false, false, false, true>(p, stream)
               : launch_v4<f16, 3, false, false, false, false, false, true>(p, stream);
}
