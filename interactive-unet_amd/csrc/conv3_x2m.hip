// 3x3x3 convolution of the split-precision forward with the CROSS TERMS ON THE fp8 MATRIX CORES ("x2m", 3-D only).
//
// fp16x2 (split16.hip, conv3_v4.hip flag SPL) evaluates (x_hi + x_lo)(w_hi + w_lo) as x_hi w_hi + x_lo w_hi + x_hi w_lo: three
// v_mfma_f32_16x16x32_f16 per product.  The two cross terms are 2^-11 of the product, so their operands need ~4 significant bits, not
// 11: here they run as ONE step of the K = 128 fp8 instruction (v_mfma_f32_16x16x128_f8f6f4, twice the multiply-adds per clock,
// conv3_f8k.hip) over 32 VIRTUAL channels per 16-channel chunk -- [x_lo8 | x_hi8] against [w_hi8 | w_lo8], all e4m3 -- into the SAME
// fp32 accumulators as the main term.  Per 16 input channels: one 16-bit step (120 K = 32 instructions) + one fp8 step (48 K = 128 +
// 24 K = 32 fp8 instructions) = 2 x 1 920 matrix cycles per wave instead of 3 x 1 920.  An e4m3 operand carries a relative rounding
// error of 2^-4 on a term that is 2^-12 (rms) of the product: 2^-16 per operand, four operands -- the logits land ~20x closer to the
// fp32 path than the fp16 mode's (tools/x2m_numerics.py: the same arithmetic on the CPU; tests/test_gpu_x2m.py on the device).
//
// Tensors.  An activation tensor of C channels (values v = act_scale * activation, fp32 in the producer's epilogue):
//   hi planes   C / 8  x [D][H][W][8] fp16         hi = f16(v)                         (the NHWC8c planes of every 16-bit kernel)
//   lo planes   C / 8  x [D][H][W][8] fp16         lo = f16(v - hi)                    (what the pool / transposed conv / head read; optional)
//   lo8 planes  C / 16 x [D][H][W][16 B] e4m3      lo8 = e4m3((v - hi) * 2^4) of the 16-channel chunk (parameter names x8 / y8 / "m8")
//   hi8 = e4m3(hi * 2^-8), the other half of the fp8 step's operand, is a function of the hi words: the loader waves make it in LDS from
//   the 16-bit halo image of the same pair while the consumers run its 16-bit step -- an activation costs 3 bytes in HBM, not 4
// and the operator of a (32 Cout, 16 Cin) block twice: w_hi in the padded K16 order of the 16-bit kernels (30 720 B, pack mode 2) and
// [w_hi8 = e4m3(w_hi * 2^-4) | w_lo8 = e4m3(w_lo * 2^8)] in the K128 order of conv3_f8k.hip (27 648 B).  The powers of two pair up
// (2^4 x 2^-4, 2^-8 x 2^8), so every term carries the scale of the main term and one accumulator scale per output channel undoes it.
// Ranges: |lo| <= 2^-11 |hi| <= 16 -> lo8 <= 256; hi <= 65504 -> hi8 <= 256; w_hi in [2^9, 2^10) per row -> w_hi8 < 64; w_lo8 <= 64.
//
// Structure: conv3_v4.hip's / conv3_f8k.hip's (one persistent workgroup per CU: 8 consumer waves read LDS and issue MFMAs, 4 loader
// waves move everything by LDS-DMA, one barrier per step, per-XCD bricks).  The two kinds of step alternate, so ONE set of buffers per
// kind is a double buffer: while the consumers work on the 16-bit step of chunk c the loaders fill the fp8 buffers of chunk c, and
// vice versa.  LDS: 34 816 (16-bit halo) + 30 720 (K16 operator) + 36 864 (e4m3 halo, z stride 192) + 27 648 (K128 operator) + 256.
#include "common.h"
#include "x2_prep_desc.h"
// tools/ab_build.sh conv3_x2m.hip -DX2M_ABLATE_NO_MFMA: timing only -- every matrix instruction becomes one multiply-add on the first words of its
// operands (the LDS reads and the loaders stay): what the kernel takes without its matrix work (what bounds the x2m kernel)
#ifdef X2M_ABLATE_NO_MFMA
template <typename A, typename B> __device__ __forceinline__ f32x4 x2m_ablate(const A a, const B b, f32x4 c) {
  c[0] = fmaf(__builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned long long, *(const unsigned long long*)&a)),
              __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned long long, *(const unsigned long long*)&b)), c[0]);
  return c;
}
#define X2M_MFMA16(a, b, c) x2m_ablate(a, b, c)
#define X2M_MFMA128(a, b, c) x2m_ablate(a, b, c)
#define X2M_MFMA32F8(a, b, c) x2m_ablate(a, b, c)
#else
#define X2M_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define X2M_MFMA128(a, b, c) __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0)
#define X2M_MFMA32F8(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, c, 0, 0, 0)
#endif
#include <cstdlib>
#include <type_traits>

namespace {

typedef long i64;
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef u32x2_t u32x2;

template <bool SMALL> struct XMTile { static constexpr int TZ = SMALL ? 2 : 4, TY = 8, TX = 16, NCW = SMALL ? 4 : 8; };
constexpr int XM_NLT = 256;                                    // loader threads (everything moves by LDS-DMA)

__device__ __attribute__((aligned(16))) unsigned int g_xm_zero16[4] = {0u, 0u, 0u, 0u};

struct ConvX2MParams {
  const void* x;  long long x_sstride;        // hi planes, elements
  const void* x8; long long x8_sstride;       // lo8 planes, bytes
  void* y;        long long y_sstride; int y_lo;      // hi planes (elements); lo planes y_lo planes further on, y_lo < 0: not written
  void* y8;       long long y8_sstride;       // lo8 planes of the output (bytes) or null
  const void* w16;                            // [cob][chunk16][column pair 5][dy][2][64][8] f16 (pack mode 2 of w_hi)
  const void* w8;                             // [cob][chunk16][F8K_WSTEP] e4m3 (K128 order of [w_hi8 | w_lo8])
  const float* oscale; const float* bias;
  int N, D, H, W, Cin, Cout;
  int tilesZ, tilesY, tilesX;
  int bz, by, bx;
  int nbz, nby, nbx;
  int epi;
  int* sat;                                   // optional: the largest |hi| bit pattern stored by this launch (atomicMax; range check)
  // fused 1x1 head (template HEAD = classes; the launch's 32 output channels are the head's input, never written): unet.py:63-69 +
  // predict.py:38 on the split words this conv would have stored -- the arithmetic of split16.hip's x2_head_kernel, bit for bit
  const float* head_w; const float* head_b;   // fp32 [ncls][32], [ncls]
  float inv_act;
  float* logits; float* probs; unsigned char* cls;
  long long oN, oC, oD, oH, oW;               // element strides of logits / probs
  float divisor; int accumulate;
  // 2^d max-pool on the way out (template POOL): besides y / y8 the launch writes the pooled tensor (hi + m8 planes of the half-size grid),
  // the words of x2m_maxpool_kernel on y / y8 (common.h: x2m_pool_take)
  void* pool_y;  long long pool_y_ss;         // hi planes, elements per sample
  void* pool_y8; long long pool_y8_ss;        // lo8 planes, bytes per sample
  // the network's FIRST conv computed on the way in (2-D, template FIRST): x / x8 are not read -- the loader waves make the 32-channel halo
  // tile from the caller's one-channel image with the first conv's own operator (split16.hip: x2_first_conv_kernel's arithmetic and words)
  const void* f_x; long long f_sN, f_sH, f_sW; int f_dtype;      // the caller's tensor (generic element strides; dtype code of x2_load_in)
  const void* f_w; const float* f_oscale; const float* f_bias;   // pack_first_conv order of [32][3][9], accumulator scale, folded bias
  float f_act;
};

// The head in the epilogue, in two parts.  (1) x2m_head_logits, per fragment: lane (q, l15) holds the 8 channels 8 q .. 8 q + 7 of voxel
// l15 (fp32 epilogue values r, scaled by act_scale).  Summation order of split16.hip's x2_head_kernel: one fmaf chain per 8-channel plane
// (= lane group), then (p0 + p1) + (p2 + p3) -- two butterfly exchanges between the lane groups; IEEE addition commutes, so all four
// groups end with the same bits -- + bias: the unfused kernel's logits, bit for bit, in every lane group.  (2) x2m_head_store, once per
// tile: a consumer wave owns FOUR fragments of 16 voxels and has four lane groups, so group q takes fragment q -- softmax, class map and
// stores run once per VOXEL per lane (computed per fragment they ran four times over: +70 us on the 2 x 128^3 launch).
// hw: LDS copy [NCLS][32] of the head weights + [NCLS] biases behind it.
template <int NCLS>
__device__ __forceinline__ void x2m_head_logits(const ConvX2MParams& p, const float* hw, const float (&r)[8], int q, bool ok, float (&l)[NCLS]) {
  f16x8 hi;
#pragma unroll
  for (int c = 0; c < NCLS; ++c) l[c] = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = fminf(fmaxf(r[j], -65504.f), 65504.f);
    const f16 h = (f16)v;
    const f16 lw = (f16)(v - (float)h);
    hi[j] = h;
    const float a = ((float)h + (float)lw) * p.inv_act;        // what the head would have read back: (hi + lo) / act_scale
#pragma unroll
    for (int c = 0; c < NCLS; ++c) l[c] = fmaf(a, hw[c * 32 + q * 8 + j], l[c]);
  }
  if (ok && p.sat != nullptr) x2_note_saturation(p.sat, hi);
#pragma unroll
  for (int c = 0; c < NCLS; ++c) {
    l[c] = __fadd_rn(l[c], __shfl_xor(l[c], 16));              // q0, q1: p0 + p1;  q2, q3: p2 + p3
    l[c] = __fadd_rn(l[c], __shfl_xor(l[c], 32));              // (p0 + p1) + (p2 + p3) in every group
    l[c] = __fadd_rn(l[c], hw[NCLS * 32 + c]);
  }
}
template <int NCLS>
__device__ __forceinline__ void x2m_head_store(const ConvX2MParams& p, const float (&l)[NCLS], bool ok, int n_img, long long nvox, long long vo,
                                               int gz, int gy, int gx) {
  if (!ok) return;
  const long long obase = n_img * p.oN + gz * p.oD + gy * p.oH + gx * p.oW;
  float mx = l[0];
#pragma unroll
  for (int c = 1; c < NCLS; ++c) mx = fmaxf(mx, l[c]);
  if (p.logits) {
#pragma unroll
    for (int c = 0; c < NCLS; ++c) p.logits[obase + c * p.oC] = l[c];
  }
  float e[NCLS], sum = 0.f;
#pragma unroll
  for (int c = 0; c < NCLS; ++c) { e[c] = expf(l[c] - mx); sum += e[c]; }
  float pr[NCLS];
  pr[0] = __fdiv_rn(e[0], sum);
  float pm = pr[0]; int am = 0;
#pragma unroll
  for (int c = 1; c < NCLS; ++c) { pr[c] = __fdiv_rn(e[c], sum); if (pr[c] > pm) { pm = pr[c]; am = c; } }
  if (p.cls) p.cls[n_img * nvox + vo] = (unsigned char)am;
  if (p.probs) {
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
      float* o = p.probs + obase + c * p.oC;
      float rr = p.accumulate ? __fadd_rn(*o, pr[c]) : pr[c];
      if (p.divisor != 1.0f) rr = __fdiv_rn(rr, p.divisor);
      *o = rr;
    }
  }
}

// POOL: the 2 x 2 x 2 max-pool of the output rides along.  A consumer wave owns 4 rows of ONE z slice: it pools x (lane pairs, DPP) and y
// (fragment pairs) in registers and leaves its 2 x 8 winners per 8-channel group in LDS (32 B each: eight 24-bit keys, common.h x2m_pool_keys; 2 KB per wave);
// the z pair lives in the wave two further on, so the LOADER waves -- idle between their LDS-DMA issue and the step's barrier -- combine the
// two slices after the next barrier and store the pooled hi / m8 words.  The consumers never wait for it.
template <bool SMALL, int HEAD = 0, bool POOL = false>
__global__ __launch_bounds__((XMTile<SMALL>::NCW * 64 + XM_NLT), 1) void conv3_x2m_kernel(ConvX2MParams p) {
  static_assert(!(POOL && HEAD > 0), "the pooled conv is an encoder conv: no head");
  using TL = XMTile<SMALL>;
  constexpr int NCW = TL::NCW, NLT = XM_NLT, NLW = NLT / 64;
  constexpr int TZ = TL::TZ, TY = TL::TY, TX = TL::TX;
  constexpr int NI = TZ * TY / NCW, NR = NI;                   // 4 tile rows (16 voxels each) per consumer wave
  constexpr int PZ = TZ + 2, PY = TY + 2, PX = TX + 2;
  constexpr int NPIX = PZ * PY * PX;                           // 1080 / 720
  // 16-bit halo image: two 8-channel planes [pixel][16 B], pixels dense
  constexpr int PLANE16 = ((NPIX * 16 + 255) / 256) * 256;
  constexpr int A16 = 2 * PLANE16;
  constexpr int NCMB = 5, W16 = NCMB * 3 * 2 * 1024;           // 30 720
  // e4m3 halo image: two 16-channel planes [voxel][16 B], z-plane stride padded 180 -> 192 voxels (conv3_f8k.hip)
  constexpr int ZS = 192;
  constexpr int PLANE8 = PZ * ZS * 16;
  constexpr int A8 = 2 * PLANE8;
  constexpr int W128 = 2 * 3 * 2 * 2 * 1024, W32 = 3 * 2 * 512, W8 = W128 + W32;      // 27 648
  constexpr int OFF_A16 = 0, OFF_W16 = A16, OFF_A8 = OFF_W16 + W16, OFF_W8 = OFF_A8 + A8, OFF_E = OFF_W8 + W8;
  constexpr int OFF_P = OFF_E + 256;                           // POOL: [wave][row pair 2][x 8][q 4] x 32 B
  static_assert(NI == 4 && NCW * NI == TZ * TY, "a consumer wave owns 4 rows of one z slice");

  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cob = blockIdx.y;
  const int nbricks = p.N * p.nbz * p.nby * p.nbx;
  const int xcd = (blockIdx.x + (nbricks < 8 ? blockIdx.y : 0)) & 7, slot = blockIdx.x >> 3;
  const int sx = slot % p.bx, sy = (slot / p.bx) % p.by, sz = slot / (p.bx * p.by);
  const int b_begin = (int)((long long)xcd * nbricks / 8), b_end = (int)((long long)(xcd + 1) * nbricks / 8);
  const int nchunk = p.Cin / 16;
  const int npairs = (b_end - b_begin) * nchunk;               // (16-bit step, fp8 step) pairs of this workgroup
  if (npairs <= 0) return;
  const long long plane_stride = (long long)p.D * p.H * p.W * 8;      // elements of a 16-bit plane = half the bytes of an m8 plane
  const long long plane16b = (long long)p.D * p.H * p.W * 16;         // bytes of either plane

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
  if (tid < 64) ((float*)(smem + OFF_E))[tid] = tid < 32 ? p.oscale[cob * 32 + tid] : (p.epi != 0 ? p.bias[cob * 32 + tid - 32] : 0.f);
  if constexpr (HEAD > 0) {      // head weights [HEAD][32] + biases behind the [oscale | bias] block; published by the first barrier
    for (int i = tid; i < HEAD * 33; i += (int)blockDim.x) ((float*)(smem + OFF_E + 256))[i] = i < HEAD * 32 ? p.head_w[i] : p.head_b[i - HEAD * 32];
  }

  if (wave >= NCW) {
    const unsigned* zero16 = iunet_opaque_ptr((const unsigned*)g_xm_zero16);      // (common.h: one address computation per kernel, not one per DMA piece)
    // ================================================================== loader waves: everything global -> LDS without registers
    const int lt = tid - NCW * 64;
    const int lw = __builtin_amdgcn_readfirstlane(lt >> 6);
    auto dma_piece = [&](const unsigned char* gsrc, unsigned dst) {      // one wave instruction: 64 x 16 B to LDS dst + lane * 16
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
    };
    auto dma_weights = [&](const unsigned char* ws, int off, int bytes) {      // lane-linear copy of a step's operator
      const int npiece = bytes / 1024;
      for (int piece = lw; piece < npiece; piece += NLW)
        dma_piece(ws + piece * 1024 + (lt & 63) * 16, __builtin_amdgcn_readfirstlane(lds0 + off + piece * 1024));
    };
    // 16-bit halo tile: 64 consecutive halo pixels per wave instruction (conv3_v4.hip: dma_acts)
    constexpr int AIT = (PLANE16 / 16 + NLT - 1) / NLT;
    int pcoord[AIT];
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      const int pix = min(lt + it * NLT, NPIX - 1);
      const int px = pix % PX, t2 = pix / PX;
      pcoord[it] = px | ((t2 % PY) << 8) | ((t2 / PY) << 16);
    }
    // e4m3 halo tile: 64 consecutive 16-byte slots of a plane per wave instruction (conv3_f8k.hip: dma_acts)
    constexpr int NSLOT = PZ * ZS;
    constexpr int DIT = (NSLOT + NLT - 1) / NLT;
    static_assert(NSLOT % 64 == 0, "an e4m3 plane is whole wave instructions");
    int dcoord[DIT];
#pragma unroll
    for (int it = 0; it < DIT; ++it) {
      const int s = lt + it * NLT;
      const int pz = s / ZS, rem = s - pz * ZS, py = rem / PX, px = rem - py * PX;
      dcoord[it] = (s < NSLOT && rem < PY * PX) ? (px | (py << 8) | (pz << 16)) : -1;
    }
    auto dma16 = [&](int k) {                                  // pair k: hi halo tile + K16 operator of its chunk
      const int tile = k / nchunk, chunk = k - tile * nchunk;
      int n_img, z0, y0, x0;
      tile_origin(tile, n_img, z0, y0, x0);
      dma_weights((const unsigned char*)p.w16 + ((long long)cob * nchunk + chunk) * W16, OFF_W16, W16);
      const f16* xc = (const f16*)p.x + (long long)n_img * p.x_sstride + (long long)chunk * 2 * plane_stride;
#pragma unroll
      for (int it = 0; it < AIT; ++it) {
        const int base = it * NLT + lw * 64;
        if (base < PLANE16 / 16) {
          const int pix = lt + it * NLT;
          const int px = pcoord[it] & 255, py = (pcoord[it] >> 8) & 255, pz = pcoord[it] >> 16;
          const int gz = z0 + pz - 1, gy = y0 + py - 1, gx = x0 + px - 1;
          const bool ok = pix < NPIX && (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
          const long long goff = (((long long)gz * p.H + gy) * p.W + gx) * 8;
          if (pix < PLANE16 / 16) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
              dma_piece(ok ? (const unsigned char*)(xc + e * plane_stride + goff) : (const unsigned char*)zero16,
                        __builtin_amdgcn_readfirstlane(lds0 + OFF_A16 + e * PLANE16 + base * 16));
          }
        }
      }
    };
    auto dma8 = [&](int k) {                                   // pair k: lo8 halo tile + K128 operator of its chunk
      const int tile = k / nchunk, chunk = k - tile * nchunk;
      int n_img, z0, y0, x0;
      tile_origin(tile, n_img, z0, y0, x0);
      dma_weights((const unsigned char*)p.w8 + ((long long)cob * nchunk + chunk) * W8, OFF_W8, W8);
      const unsigned char* xc = (const unsigned char*)p.x8 + (long long)n_img * p.x8_sstride + (long long)chunk * plane16b;
#pragma unroll
      for (int it = 0; it < DIT; ++it) {
        const int base = it * NLT + lw * 64;
        if (base < NSLOT) {
          const int c = dcoord[it];
          const int gz = z0 + (c >> 16) - 1, gy = y0 + ((c >> 8) & 255) - 1, gx = x0 + (c & 255) - 1;
          const bool ok = c >= 0 && (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
          const long long goff = (((long long)gz * p.H + gy) * p.W + gx) * 16;
          dma_piece(ok ? xc + goff : (const unsigned char*)zero16, __builtin_amdgcn_readfirstlane(lds0 + OFF_A8 + base * 16));
        }
      }
    };
    // the other half of the fp8 step's operand, hi8 = e4m3(hi * 2^-8): made here from the 16-bit halo image of the SAME pair (in LDS since
    // the last barrier, read by the consumers' 16-bit step at the same time) -- it is a function of the hi words, so it never travels
    auto make_hi8 = [&]() {                                    // every read first (one LDS round trip, not one per slot), then convert + write
      f16x8 va[DIT], vb[DIT];
#pragma unroll
      for (int it = 0; it < DIT; ++it) {
        const int c = dcoord[it];
        const int pix = c >= 0 ? ((c >> 16) * PY + ((c >> 8) & 255)) * PX + (c & 255) : 0;
        va[it] = *(const f16x8*)(smem + OFF_A16 + pix * 16);
        vb[it] = *(const f16x8*)(smem + OFF_A16 + PLANE16 + pix * 16);
      }
#pragma unroll
      for (int it = 0; it < DIT; ++it) {
        const int s = lt + it * NLT;
        if (s < NSLOT) {
          const u32x2 a = x2m_hi8(va[it]), b = x2m_hi8(vb[it]);
          *(u32x4*)(smem + OFF_A8 + PLANE8 + s * 16) = dcoord[it] >= 0 ? u32x4{a[0], a[1], b[0], b[1]} : u32x4{0u, 0u, 0u, 0u};
        }
      }
    };
#ifdef X2M_ABLATE_NO_WAIT      // (timing only: the step barriers do not wait for the copies to land)
    auto landed = [&]() {};
#else
    auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
#endif
    // POOL: thread = (pooled voxel of the tile, 8-channel group q): the z pair of its x-y winners sits in waves w and w + 2.  Two parts:
    // pool_words reads and decodes while the step's LDS-DMA is in flight; pool_store issues the three stores AFTER the wait for that
    // DMA -- a store issued before it would sit in the same vmcnt wait and hold the step's barrier until its write came back.
    [[maybe_unused]] f16x8 pw_hi; [[maybe_unused]] u32x2 pw_lo8;
    [[maybe_unused]] f16* pw_dst = nullptr; [[maybe_unused]] unsigned char* pw_dst8 = nullptr; [[maybe_unused]] long long pw_plane = 0;
    [[maybe_unused]] auto pool_words = [&](int tile) {
      pw_dst = nullptr;
      if (lt >= NCW * 32) return;
      const int q = lt & 3, x8 = (lt >> 2) & 7, py = (lt >> 5) & 3, pz = lt >> 7;
      const unsigned char* a = smem + OFF_P + (((((4 * pz + (py >> 1)) * 2 + (py & 1)) * 8 + x8) * 4 + q) * 32);
      const unsigned char* b = a + 2 * (2 * 8 * 4 * 32);
      const u32x4 a0 = *(const u32x4*)a, a1 = *(const u32x4*)(a + 16), b0 = *(const u32x4*)b, b1 = *(const u32x4*)(b + 16);
      unsigned kk[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { kk[j] = max(a0[j], b0[j]); kk[4 + j] = max(a1[j], b1[j]); }
      int n_img, z0, y0, x0;
      tile_origin(tile, n_img, z0, y0, x0);
      const int Do = p.D >> 1, Ho = p.H >> 1, Wo = p.W >> 1;
      const int oz = (z0 >> 1) + pz, oy = (y0 >> 1) + py, ox = (x0 >> 1) + x8;
      if (oz < Do && oy < Ho && ox < Wo) {
        const long long onvox = (long long)Do * Ho * Wo, ovo = ((long long)oz * Ho + oy) * Wo + ox;
        x2m_pool_unkeys(kk, pw_hi, pw_lo8);
        pw_dst = (f16*)p.pool_y + (long long)n_img * p.pool_y_ss + ((long long)(cob * 4 + q) * onvox + ovo) * 8;
        pw_dst8 = (unsigned char*)p.pool_y8 + (long long)n_img * p.pool_y8_ss + x2m_off(cob * 4 + q, ovo, onvox);
        pw_plane = onvox * 16;
      }
    };
    [[maybe_unused]] auto pool_store = [&]() {
      if (pw_dst != nullptr) {
        *(f16x8*)pw_dst = pw_hi;
        *(u32x2*)pw_dst8 = pw_lo8;
      }
    };
    dma16(0);
    landed();
    lds_barrier();
    for (int k = 0; k < npairs; ++k) {
#ifndef X2M_ABLATE_NO_COPIES   // (timing only: the loader waves keep their barriers and nothing else -- the consumers alone)
      dma8(k);                                                 // consumers: 16-bit step of pair k
#endif
#if !defined(X2M_ABLATE_NO_HI8) && !defined(X2M_ABLATE_NO_COPIES)      // (tools/ab_build.sh -DX2M_ABLATE_NO_HI8: timing only, the fp8 operand's hi8 half stays stale)
      make_hi8();
#endif
      landed();
      lds_barrier();
#ifndef X2M_ABLATE_NO_COPIES
      if (k + 1 < npairs) dma16(k + 1);                        // consumers: fp8 step of pair k
#endif
      const bool pool_now = POOL && k > 0 && k % nchunk == 0;  // the tile that ended with pair k - 1 left its x-y winners before this barrier
      if constexpr (POOL) { if (pool_now) pool_words(k / nchunk - 1); }
      landed();
      if constexpr (POOL) { if (pool_now) pool_store(); }
      lds_barrier();
    }
    if constexpr (POOL) {
      lds_barrier();                                           // the consumers' last epilogue
      pool_words(npairs / nchunk - 1);
      pool_store();
    }
    return;
  }

  // ==================================================================== consumer waves
  const int l15 = lane & 15, q = lane >> 4;
  const int row_first = wave * NI;                             // first tile row (z * TY + y) of this wave
  // ---- 16-bit step: lanes q & 1 take the 8-channel half, q >> 1 the column of a pair (conv3_v4.hip)
  int col_off[NCMB];
#pragma unroll
  for (int c = 0; c < NCMB; ++c) {
    const int col = min(2 * c + (q >> 1), 8);                  // the missing partner re-reads a valid column (zero weights)
    col_off[c] = ((col / 3) * PY * PX + (col % 3)) * 16;
  }
  const int rbase16 = (q & 1) * PLANE16 + ((((row_first / TY) * PY + (row_first % TY)) * PX) + l15) * 16;
  // ---- fp8 step: a lane group takes one filter column of a K = 128 group (conv3_f8k.hip)
  int coff[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int col = q == 0 ? f8k_col(g, 0) : q == 1 ? f8k_col(g, 1) : q == 2 ? f8k_col(g, 2) : f8k_col(g, 3);
    coff[g] = ((col / 3) * ZS + (col % 3)) * 16;
  }
  const int coff8 = (2 * ZS + 2) * 16 + (q >> 1) * PLANE8 + (q & 1) * 8;
  const int rbase8 = (((row_first / TY) * ZS + (row_first % TY) * PX) + l15) * 16;

  f32x4 acc[2][NI];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NI; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
  using I6 = std::integral_constant<int, 6>; using I7 = std::integral_constant<int, 7>; using I8 = std::integral_constant<int, 8>;

  // Operand registers of the two kinds of step.  The software pipeline runs ACROSS the steps: a step's one barrier sits in front of its
  // LAST group of matrix instructions -- every LDS read of the step has landed by then (its last group's operands are in registers), so
  // the loaders may overwrite its buffers, and the other kind's buffers are published -- and the first group of the NEXT step is read
  // behind that barrier, between the last group's matrix instructions (conv3_v4.hip's resident-weight variants, conv3_f8k.hip).
  f16x8 R16[2][NR + 2], A16f[2][3][2];                         // 16-bit step: row / operator fragments of the running and the next column pair
  i32x8 R8f[2][NR + 2], A8f[2][2];                             // fp8 step: K = 128 row fragments of the two groups, operator of the running / next (group, dy)
  i64 R8n[NR + 2], A8n[3][2];                                  // the ninth column's K = 32 operands

  // -------------------------------------------------------------- the 16-bit step: x_hi w_hi, 5 column pairs x 3 dy x 4 rows x 2 halves
  auto load_group = [&](int c, auto BUF) {
    constexpr int b = decltype(BUF)::value;
    const unsigned char* ab = smem + OFF_A16 + rbase16;
    const unsigned char* wl = smem + OFF_W16 + lane * 16;
#pragma unroll
    for (int r = 0; r < NR + 2; ++r) R16[b][r] = *(const f16x8*)(ab + r * PX * 16 + col_off[c]);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      A16f[b][dy][0] = *(const f16x8*)(wl + ((c * 3 + dy) * 2 + 0) * 1024);
      A16f[b][dy][1] = *(const f16x8*)(wl + ((c * 3 + dy) * 2 + 1) * 1024);
    }
  };
  auto group_mfmas = [&](auto BUF, bool reads_pending) {
    constexpr int b = decltype(BUF)::value;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int n = 0; n < NI; ++n) {
        acc[0][n] = X2M_MFMA16(A16f[b][dy][0], R16[b][n + dy], acc[0][n]);
        acc[1][n] = X2M_MFMA16(A16f[b][dy][1], R16[b][n + dy], acc[1][n]);
      }
    if (reads_pending) {
      constexpr int NRD = NR + 2 + 6, MPR = (3 * NI * 2) / NRD;      // 12 LDS reads spread over 24 MFMAs
#pragma unroll
      for (int i = 0; i < NRD; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // -------------------------------------------------------------- the fp8 step: [x_lo8 | x_hi8] [w_hi8 | w_lo8], 2 K = 128 groups + the ninth column
  auto rd128 = [&](const unsigned char* ptr, int second) -> i32x8 {
    const u32x4 lo = *(const u32x4*)ptr, hi = *(const u32x4*)(ptr + second);
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  auto load_sub = [&](auto SG) {                               // sub-group sg = 0 .. 8: (group 0, dy 0..2), (group 1, dy 0..2), (ninth column, dy 0..2)
    constexpr int sg = decltype(SG)::value, g = sg / 3, dy = sg % 3;
    const unsigned char* ab = smem + OFF_A8;
    const unsigned char* wl = smem + OFF_W8;
    if constexpr (g < 2) {
#pragma unroll
      for (int m = 0; m < 2; ++m) A8f[sg & 1][m] = rd128(wl + ((g * 3 + dy) * 2 + m) * 2048 + lane * 16, 1024);
      constexpr int r0 = dy == 0 ? 0 : NR - 1 + dy, r1 = dy == 0 ? NR : NR + dy;
#pragma unroll
      for (int r = r0; r < r1; ++r) R8f[g][r] = rd128(ab + rbase8 + coff[g] + r * PX * 16, PLANE8);
    } else if constexpr (dy == 0) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int m = 0; m < 2; ++m) A8n[d][m] = *(const i64*)(wl + W128 + (d * 2 + m) * 512 + lane * 8);
#pragma unroll
      for (int r = 0; r < NR + 2; ++r) R8n[r] = *(const i64*)(ab + rbase8 + coff8 + r * PX * 16);
    }
  };
  auto mfma_sub = [&](auto SG, auto NREADS) {
    constexpr int sg = decltype(SG)::value, g = sg / 3, dy = sg % 3, nreads = decltype(NREADS)::value;
#pragma unroll
    for (int n = 0; n < NI; ++n)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if constexpr (g < 2) acc[m][n] = X2M_MFMA128(A8f[sg & 1][m], R8f[g][n + dy], acc[m][n]);
        else acc[m][n] = X2M_MFMA32F8(A8n[dy][m], R8n[n + dy], acc[m][n]);
      }
    if constexpr (nreads >= 8) {
      constexpr int RPM = (nreads + 7) / 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
      }
    } else if constexpr (nreads > 0) {
      constexpr int MPR = 8 / nreads;
#pragma unroll
      for (int i = 0; i < nreads; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    if constexpr (sg < 6 || sg == 8) __builtin_amdgcn_sched_barrier(0);      // (the three K = 32 sub-groups are one scheduling region)
  };
  using N4 = std::integral_constant<int, 4>; using N6 = std::integral_constant<int, 6>; using N12 = std::integral_constant<int, 12>;

  auto tile_epilogue = [&](int tile) {
    int n_img, z0, y0, x0;
    tile_origin(tile, n_img, z0, y0, x0);
    f16* yout = (f16*)p.y + (long long)n_img * p.y_sstride;
    float bias_r[8], os_r[8];
    {
      const f32x4* ep = (const f32x4*)(smem + OFF_E) + 2 * q;
      const f32x4 w0 = ep[0], w1 = ep[1], b0 = ep[8], b1 = ep[9];
#pragma unroll
      for (int j = 0; j < 4; ++j) { os_r[j] = w0[j]; os_r[4 + j] = w1[j]; bias_r[j] = b0[j]; bias_r[4 + j] = b1[j]; }
    }
    const float relu_floor = p.epi == 2 ? 0.f : -65504.f;
    [[maybe_unused]] float hl[HEAD > 0 ? HEAD : 1];
    [[maybe_unused]] unsigned pool_k[8];                       // POOL: the keys of the first fragment of a y pair (common.h: x2m_pool_keys)
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int row = row_first + n;
      const int gz = z0 + row / TY, gy = y0 + row % TY, gx = x0 + l15;
      const bool ok = gz < p.D && gy < p.H && gx < p.W;
      const long long vo = ((long long)gz * p.H + gy) * p.W + gx;
      float r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        r[j] = fmaf(j < 4 ? acc[0][n][j] : acc[1][n][j - 4], os_r[j], bias_r[j]);
        if constexpr (HEAD > 0) { if (p.epi == 2) r[j] = fmaxf(r[j], 0.f); }      // (the head variants sit at their register cap)
        else r[j] = __builtin_amdgcn_fmed3f(r[j], relu_floor, 65504.f);           // ReLU AND the split's range clamp in one v_med3: floor = 0 or -65504 (NaN -> the floor, as fmax / fmin gave)
      }
      if constexpr (HEAD > 0) {
        float lf[HEAD];
        x2m_head_logits<HEAD>(p, (const float*)(smem + OFF_E + 256), r, q, ok, lf);
#pragma unroll
        for (int c = 0; c < HEAD; ++c) hl[c] = q == n ? lf[c] : hl[c];          // lane group q keeps fragment q's voxels
      } else {
      f16x8 hi, lo;
      u32x2 lo8, hi8;
      x2m_split8<true>(r, hi, lo, lo8, hi8);
      if (ok) {
        *(f16x8*)(yout + (long long)(cob * 4 + q) * plane_stride + vo * 8) = hi;
        if (p.y_lo >= 0) *(f16x8*)(yout + (long long)(p.y_lo + cob * 4 + q) * plane_stride + vo * 8) = lo;
        if (p.y8 != nullptr) {
          unsigned char* y8 = (unsigned char*)p.y8 + (long long)n_img * p.y8_sstride + x2m_off(cob * 4 + q, vo, plane16b / 16);
          *(u32x2*)y8 = lo8;
        }
        if (p.sat != nullptr) x2_note_saturation(p.sat, hi);
      }
      if constexpr (POOL) {
        unsigned kk[8];
        x2m_pool_keys(hi, lo8, kk);
        if ((n & 1) == 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) pool_k[j] = kk[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            kk[j] = max(kk[j], pool_k[j]);                     // rows n - 1, n (y pair)
            kk[j] = max(kk[j], lane_xor1(kk[j]));              // x pair
          }
          if (!(l15 & 1)) {
            unsigned char* dst = smem + OFF_P + ((((wave * 2 + (n >> 1)) * 8 + (l15 >> 1)) * 4 + q) * 32);
            *(u32x4*)dst = u32x4{kk[0], kk[1], kk[2], kk[3]};
            *(u32x4*)(dst + 16) = u32x4{kk[4], kk[5], kk[6], kk[7]};
          }
        }
      }
      }
      acc[0][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[1][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (HEAD > 0) {
      const int row = row_first + q;                                            // this lane group's fragment
      const int gz = z0 + row / TY, gy = y0 + row % TY, gx = x0 + l15;
      x2m_head_store<HEAD>(p, hl, gz < p.D && gy < p.H && gx < p.W, n_img, plane16b / 16, ((long long)gz * p.H + gy) * p.W + gx, gz, gy, gx);
    }
  };

  lds_barrier();                                               // the first 16-bit step is in LDS
  load_group(0, I0{});
  __builtin_amdgcn_sched_barrier(0);
  for (int k = 0; k < npairs; ++k) {
    // ---- 16-bit step of pair k (its first column pair is in fragment set 0 already)
    load_group(1, I1{}); group_mfmas(I0{}, true);
    load_group(2, I0{}); group_mfmas(I1{}, true);
    load_group(3, I1{}); group_mfmas(I0{}, true);
    load_group(4, I0{}); group_mfmas(I1{}, true);
    lds_barrier();                                             // the 16-bit buffers are read; the fp8 buffers of pair k are filled
    load_sub(I0{});                                            // 12 reads between the last column pair's 24 MFMAs
    group_mfmas(I0{}, true);
    // ---- fp8 step of pair k
    load_sub(I1{}); mfma_sub(I0{}, N6{});
    load_sub(I2{}); mfma_sub(I1{}, N6{});
    load_sub(I3{}); mfma_sub(I2{}, N12{});
    load_sub(I4{}); mfma_sub(I3{}, N6{});
    load_sub(I5{}); mfma_sub(I4{}, N6{});
    load_sub(I6{}); mfma_sub(I5{}, N12{});
    lds_barrier();                                             // the fp8 buffers are read; the 16-bit buffers of pair k + 1 are filled
    load_group(0, I0{});                                       // (after the last pair: a harmless re-read) 12 reads between the 24 K = 32 instructions
    mfma_sub(I6{}, N4{}); mfma_sub(I7{}, N4{}); mfma_sub(I8{}, N4{});
    const int tile = k / nchunk;
    if (k - tile * nchunk == nchunk - 1) tile_epilogue(tile);
  }
  if constexpr (POOL) lds_barrier();                           // the loaders finish the last tile's pool behind this one
}


// ==================================================================================================================== 2-D (3 x 3 filters)
// The same two kinds of step on conv3_v4.hip's 2-D geometry: tile 16 x 32 pixels, 8 consumer waves (2 rows x 2 x-halves each), a step =
// 32 input channels.  16-bit step: the cross-pair order (pack mode 6: sub-chunk 0 columns 0 / 1, sub-chunk 1 columns 0 / 1, column 2 of
// both: 9 taps in 9 k-slots, 72 K = 32 instructions).  fp8 step: the 32 channels are two virtual blocks b = [lo8 | hi8] of 16 channels; a
// lane of the K = 128 instruction holds one block at ONE tap, the four lane groups take four taps -- groups (b0: taps 0..3), (b0: 4..7),
// (b1: 0..3), (b1: 4..7) -- and tap 8 of both blocks goes to two K = 32 fp8 instructions: 32 K = 128 + 16 K = 32 instructions = 1 280
// matrix cycles per wave beside the 1 152 of the 16-bit step (2 432 against the 3 456 of fp16x2's three 16-bit steps).  Operator of a
// (32 Cout, 32 Cin) block: [group 4][m 2][half 2][64 lanes][16 B] + [block 2][m 2][64 lanes][8 B] = 18 432 B.  Small images (the deep
// levels of 128^2 slices) run several slot groups per XCD, as conv3_v4.hip does, so a batch of slices fills the chip.
constexpr int X2M2_W8 = 4 * 2 * 2 * 1024 + 2 * 2 * 512;         // 18 432
__host__ __device__ inline int x2m2_w8_offset(int tap, int m, int b, int e, int o, int row) {      // 8 channels 8 o.. of half e of block b at `tap`
  if (tap == 8) return 4 * 2 * 2 * 1024 + ((b * 2 + m) * 64 + (2 * e + o) * 16 + row) * 8;
  const int G = b * 2 + (tap >> 2), q = tap & 3;
  return (((G * 2 + m) * 2 + e) * 64 + q * 16 + row) * 16 + 8 * o;
}

// POOL: the 2 x 2 max-pool of the output rides along -- a consumer wave owns two whole rows of the tile, so the pool is in registers
// (y: fragment pairs, x: lane pairs by DPP); the winners' keys go to LDS (8 x 16 pooled pixels x 4 channel groups x 32 B) and the LOADER
// waves decode and store them after the next barrier, as in 3-D (consumers storing the pooled words themselves measured slower).
// FIRST: the launch is the SECOND conv of the first encoder stage and computes the first conv (one input channel -> 32, BatchNorm folded,
// ReLU) itself: its 32-channel input tensor never exists in HBM.  Each loader wave owns a band of the 18 halo rows: it stages the band's
// raw pixels (+ 1 ring) as split words [lo | hi | hi] in a private LDS patch, gathers the im2col operand per lane and runs
// x2_first_conv_kernel's two MFMAs per 16 halo pixels (K = 27 of 32), scales, splits (x2m_split8: the words that kernel would have
// stored) and writes the hi words into the 16-bit halo image during the consumers' fp8 step; the e4m3 words wait in registers for the
// barrier that frees the fp8 halo image.  Halo pixels outside the image are the second conv's zero padding, not conv values.
template <int HEAD = 0, bool POOL = false, bool FIRST = false>
__global__ __launch_bounds__(8 * 64 + XM_NLT, 1) void conv2_x2m_kernel(ConvX2MParams p) {
  static_assert(!(POOL && HEAD > 0), "the pooled conv is an encoder conv: no head");
  static_assert(!(FIRST && HEAD > 0), "the first stage's second conv has no head");
  constexpr int NCW = 8, NLT = XM_NLT, NLW = NLT / 64;
  constexpr int TY = 16, TX = 32, FX = 2, NI = 4, NR = 2;
  constexpr int PY = TY + 2, PX = TX + 2, NPIX = PY * PX;                     // 612
  constexpr int PLANE = ((NPIX * 16 + 255) / 256) * 256;                       // 9 984: one plane of either halo image
  constexpr int A16 = 4 * PLANE, W16 = 3 * 3 * 2 * 1024;                       // 32 channels = four 8-channel planes; 18 432
  constexpr int A8 = 4 * PLANE, W128 = 4 * 2 * 2 * 1024, W8 = X2M2_W8;          // [lo8 b0 | hi8 b0 | lo8 b1 | hi8 b1]
  constexpr int OFF_A16 = 0, OFF_W16 = A16, OFF_A8 = OFF_W16 + W16, OFF_W8 = OFF_A8 + A8, OFF_E = OFF_W8 + W8;
  constexpr int OFF_P = OFF_E + 256;                           // POOL: [pooled row 8][pooled x 16][q 4] x 32 B of keys
  constexpr int FPX = PX + 2, FPIX = 7 * FPX, FPATCH = 3 * FPIX * 2;      // FIRST: a wave's patch: 7 rows x 36 pixels x [lo | hi | hi] f16 (1 512 B)
  constexpr int OFF_F = OFF_P + (POOL ? 8 * 16 * 4 * 32 : 0);

  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cob = blockIdx.y;
  const int nbricks = p.N * p.nby * p.nbx;
  const int bslots = p.bx * p.by, ngrp = (int)gridDim.x / (8 * bslots);
  const int xcd = (blockIdx.x + (nbricks < 8 ? blockIdx.y : 0)) & 7, slot_all = blockIdx.x >> 3;
  const int grp = slot_all / bslots, slot = slot_all - grp * bslots;
  const int sx = slot % p.bx, sy = slot / p.bx;
  const int xb0 = (int)((long long)xcd * nbricks / 8), xb1 = (int)((long long)(xcd + 1) * nbricks / 8);
  const int b_begin = xb0 + (int)((long long)(xb1 - xb0) * grp / ngrp), b_end = xb0 + (int)((long long)(xb1 - xb0) * (grp + 1) / ngrp);
  const int nchunk = p.Cin / 32;
  const int npairs = (b_end - b_begin) * nchunk;
  if (npairs <= 0) return;
  const long long nvox = (long long)p.H * p.W;
  const long long plane_stride = nvox * 8;                    // elements of a 16-bit plane
  const long long plane16b = nvox * 16;                       // bytes of either plane

  auto tile_origin = [&](int k, int& n_img, int& y0, int& x0) -> bool {
    int b = b_begin + k;
    const int Bx = b % p.nbx; b /= p.nbx;
    const int By = b % p.nby; n_img = b / p.nby;
    const int ty = By * p.by + sy, tx = Bx * p.bx + sx;
    y0 = ty * TY; x0 = tx * TX;
    return ty < p.tilesY && tx < p.tilesX;
  };
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  if (tid < 64) ((float*)(smem + OFF_E))[tid] = tid < 32 ? p.oscale[cob * 32 + tid] : (p.epi != 0 ? p.bias[cob * 32 + tid - 32] : 0.f);
  if constexpr (HEAD > 0) {      // head weights [HEAD][32] + biases behind the [oscale | bias] block; published by the first barrier
    for (int i = tid; i < HEAD * 33; i += (int)blockDim.x) ((float*)(smem + OFF_E + 256))[i] = i < HEAD * 32 ? p.head_w[i] : p.head_b[i - HEAD * 32];
  }

  if (wave >= NCW) {
    const unsigned* zero16 = iunet_opaque_ptr((const unsigned*)g_xm_zero16);      // (common.h: one address computation per kernel, not one per DMA piece)
    // ================================================================== loader waves (LDS-DMA only)
    const int lt = tid - NCW * 64;
    const int lw = __builtin_amdgcn_readfirstlane(lt >> 6);
    auto dma_piece = [&](const unsigned char* gsrc, unsigned dst) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
    };
    auto dma_weights = [&](const unsigned char* ws, int off, int bytes) {
      const int npiece = bytes / 1024;
      for (int piece = lw; piece < npiece; piece += NLW)
        dma_piece(ws + piece * 1024 + (lt & 63) * 16, __builtin_amdgcn_readfirstlane(lds0 + off + piece * 1024));
    };
    constexpr int AIT = (PLANE / 16 + NLT - 1) / NLT;          // 3
    int pcoord[AIT];
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      const int pix = min(lt + it * NLT, NPIX - 1);
      pcoord[it] = (pix % PX) | ((pix / PX) << 8);
    }
    // one halo image: NPL source planes of 16 bytes per pixel (`src` = the first, consecutive ones plane16b apart) -> LDS planes off + e * STEP * PLANE
    auto dma_halo = [&](const unsigned char* src, int y0, int x0, int off, auto NPLc, auto STEPc) {
      constexpr int NPL = decltype(NPLc)::value, STEP = decltype(STEPc)::value;
#pragma unroll
      for (int it = 0; it < AIT; ++it) {
        const int base = it * NLT + lw * 64;
        if (base < PLANE / 16) {
          const int pix = lt + it * NLT;
          const int px = pcoord[it] & 255, py = pcoord[it] >> 8;
          const int gy = y0 + py - 1, gx = x0 + px - 1;
          const bool ok = pix < NPIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
          const long long goff = ((long long)gy * p.W + gx) * 16;
          if (pix < PLANE / 16) {
#pragma unroll
            for (int e = 0; e < NPL; ++e)
              dma_piece(ok ? src + e * plane16b + goff : (const unsigned char*)zero16,
                        __builtin_amdgcn_readfirstlane(lds0 + off + e * STEP * PLANE + base * 16));
          }
        }
      }
    };
    auto dma16 = [&](int k) {
      const int tile = k / nchunk, chunk = k - tile * nchunk;
      int n_img, y0, x0;
      tile_origin(tile, n_img, y0, x0);
      if (nchunk > 1 || k == 0)                                // (a 32-channel input: the one chunk's operators stay resident)
        dma_weights((const unsigned char*)p.w16 + ((long long)cob * nchunk + chunk) * W16, OFF_W16, W16);
      dma_halo((const unsigned char*)((const f16*)p.x + (long long)n_img * p.x_sstride + (long long)chunk * 4 * plane_stride), y0, x0, OFF_A16,
               std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{});
    };
    auto dma8 = [&](int k) {
      const int tile = k / nchunk, chunk = k - tile * nchunk;
      int n_img, y0, x0;
      tile_origin(tile, n_img, y0, x0);
      if (nchunk > 1 || k == 0)
        dma_weights((const unsigned char*)p.w8 + ((long long)cob * nchunk + chunk) * W8, OFF_W8, W8);
      dma_halo((const unsigned char*)p.x8 + (long long)n_img * p.x8_sstride + (long long)chunk * 2 * plane16b, y0, x0, OFF_A8,
               std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});      // the lo8 planes of the two 16-channel blocks -> LDS planes 0, 2
    };
    // hi8 planes (LDS planes 1, 3) of the fp8 halo image from the 16-bit halo image of the same pair (see conv3_x2m_kernel)
    auto make_hi8 = [&]() {                                    // every read first (one LDS round trip, not one per pixel), then convert + write
      f16x8 v[AIT][4];
#pragma unroll
      for (int it = 0; it < AIT; ++it) {
        const int pix = min(lt + it * NLT, PLANE / 16 - 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[it][e] = *(const f16x8*)(smem + OFF_A16 + e * PLANE + pix * 16);
      }
#pragma unroll
      for (int it = 0; it < AIT; ++it) {
        const int pix = lt + it * NLT;
        if (pix < PLANE / 16) {
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const u32x2 a0 = x2m_hi8(v[it][2 * b]), a1 = x2m_hi8(v[it][2 * b + 1]);
            *(u32x4*)(smem + OFF_A8 + (2 * b + 1) * PLANE + pix * 16) = u32x4{a0[0], a0[1], a1[0], a1[1]};
          }
        }
      }
    };
#ifdef X2M_ABLATE_NO_WAIT      // (timing only: the step barriers do not wait for the copies to land)
    auto landed = [&]() {};
#else
    auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
#endif
    // POOL: two (pooled pixel, channel group) items per loader thread.  pool_words reads and decodes the winners' keys while the step's
    // LDS-DMA is in flight; pool_store issues the stores AFTER the wait for that DMA (a store issued before it would sit in the same
    // vmcnt wait and hold the step's barrier until its write came back).
    [[maybe_unused]] f16x8 pw_hi[2]; [[maybe_unused]] u32x2 pw_lo8[2];
    [[maybe_unused]] f16* pw_dst[2] = {nullptr, nullptr}; [[maybe_unused]] unsigned char* pw_dst8[2] = {nullptr, nullptr};
    [[maybe_unused]] long long pw_plane = 0;
    [[maybe_unused]] auto pool_words = [&](int tile) {
      int n_img, y0, x0;
      tile_origin(tile, n_img, y0, x0);
      const int Ho = p.H >> 1, Wo = p.W >> 1;
      const long long onvox = (long long)Ho * Wo;
      pw_plane = onvox * 16;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int item = lt + it * NLT;
        const int q = item & 3, px = (item >> 2) & 15, py = item >> 6;
        const unsigned char* a = smem + OFF_P + item * 32;
        const u32x4 a0 = *(const u32x4*)a, a1 = *(const u32x4*)(a + 16);
        const unsigned kk[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const int oy = (y0 >> 1) + py, ox = (x0 >> 1) + px;
        pw_dst[it] = nullptr;
        if (oy < Ho && ox < Wo) {
          const long long ovo = (long long)oy * Wo + ox;
          x2m_pool_unkeys(kk, pw_hi[it], pw_lo8[it]);
          pw_dst[it] = (f16*)p.pool_y + (long long)n_img * p.pool_y_ss + ((long long)(cob * 4 + q) * onvox + ovo) * 8;
          pw_dst8[it] = (unsigned char*)p.pool_y8 + (long long)n_img * p.pool_y8_ss + x2m_off(cob * 4 + q, ovo, onvox);
        }
      }
    };
    [[maybe_unused]] auto pool_store = [&]() {
#pragma unroll
      for (int it = 0; it < 2; ++it)
        if (pw_dst[it] != nullptr) {
          *(f16x8*)pw_dst[it] = pw_hi[it];
          *(u32x2*)pw_dst8[it] = pw_lo8[it];
        }
    };
    if constexpr (FIRST) {
      // ---- the first conv on the way in (nchunk == 1: a pair is a tile; the operators of the one chunk stay resident)
      const int ll = lt & 63, l15 = ll & 15, q = ll >> 4;
      const int hr0 = lw == 0 ? 0 : lw == 1 ? 5 : lw == 2 ? 10 : 14, nrow = lw < 2 ? 5 : 4;      // this wave's halo rows [hr0, hr0 + nrow)
      const int npx = nrow * PX, nfr = (npx + 15) / 16;                   // 170 / 136 pixels: 11 / 9 fragments
      constexpr int MAXF = 11;
      f16* xs = (f16*)(smem + OFF_F + lw * FPATCH);
      const f16x8 wa0 = ((const f16x8*)p.f_w)[ll], wa1 = ((const f16x8*)p.f_w)[64 + ll];
      // the first conv's accumulator scales and biases: a copy per wave behind its patch (read per fragment: 16 registers less to hold)
      float* fsb = (float*)(smem + OFF_F + 4 * FPATCH + lw * 256);
      fsb[ll] = ll < 32 ? p.f_oscale[ll] : p.f_bias[ll - 32];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      int koff[8];                                                       // k = 8 q + j = tap * 3 + part (pack_first_conv_kernel); k >= 27: zero weight
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * q + j, kc = k < 27 ? k : 0, tap = kc / 3, part = kc % 3;
        koff[j] = part * FPIX + (tap / 3) * FPX + (tap % 3);
      }
      float raw[4];
      auto patch_load = [&](int tile) {                                  // raw pixels of the band's patch: image rows y0 + hr0 - 2 .., columns x0 - 2 ..
        int n_img, y0, x0;
        tile_origin(tile, n_img, y0, x0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int idx = ll + 64 * i, pr = idx / FPX, pc = idx - pr * FPX;
          const int gy = y0 + hr0 - 2 + pr, gx = x0 - 2 + pc;
          raw[i] = 0.f;
          if (idx < (nrow + 2) * FPX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
            raw[i] = x2_load_in(p.f_x, n_img * p.f_sN + gy * p.f_sH + gx * p.f_sW, p.f_dtype);
        }
      };
      auto patch_commit = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int idx = ll + 64 * i;
          if (idx < (nrow + 2) * FPX) {
            f16 hi, lo;
            split16<f16>(raw[i] * p.f_act, hi, lo);
            xs[idx] = lo; xs[FPIX + idx] = hi; xs[2 * FPIX + idx] = hi;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (the patch is this wave's own: no barrier)
      };
      u32x2 m_lo[MAXF];                                                  // (the hi8 words are a function of the hi words: m8_commit reads those back from the 16-bit image)
      auto first_conv = [&](int tile) {                                  // hi words -> the 16-bit halo image; e4m3 words -> m_lo / m_hi
        int n_img, y0, x0;
        tile_origin(tile, n_img, y0, x0);
#pragma unroll
        for (int f = 0; f < MAXF; ++f) {
          if (f < nfr) {
            const int lp = min(16 * f + l15, npx - 1);                   // the band's pixel of this lane (the last fragment is partial)
            const int pr = lp / PX, pc = lp - pr * PX;
            const int base = pr * FPX + pc;
            f16x8 b;
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = xs[base + koff[j]];
            const f32x4 a0 = mfma16<f16>(wa0, b, f32x4{0.f, 0.f, 0.f, 0.f}), a1 = mfma16<f16>(wa1, b, f32x4{0.f, 0.f, 0.f, 0.f});
            float rr[8];
            {
              const f32x4 s0 = *(const f32x4*)(fsb + 8 * q), s1 = *(const f32x4*)(fsb + 8 * q + 4);
              const f32x4 b0 = *(const f32x4*)(fsb + 32 + 8 * q), b1 = *(const f32x4*)(fsb + 32 + 8 * q + 4);
#pragma unroll
              for (int j = 0; j < 4; ++j) { rr[j] = fmaxf(fmaf(a0[j], s0[j], b0[j]), 0.f); rr[4 + j] = fmaxf(fmaf(a1[j], s1[j], b1[j]), 0.f); }
            }
            f16x8 hi, lo;
            u32x2 l8, h8;
            x2m_split8(rr, hi, lo, l8, h8);
            const int gy = y0 + hr0 + pr - 1, gx = x0 + pc - 1;
            const bool in = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            if (!in) { hi = f16x8{0, 0, 0, 0, 0, 0, 0, 0}; l8 = u32x2{0u, 0u}; h8 = u32x2{0u, 0u}; }      // the second conv's zero padding
            else if (p.sat != nullptr) x2_note_saturation(p.sat, hi);
            m_lo[f] = l8;
            if (16 * f + l15 < npx) *(f16x8*)(smem + OFF_A16 + q * PLANE + (hr0 * PX + lp) * 16) = hi;
          }
        }
      };
      auto m8_commit = [&]() {
#pragma unroll
        for (int f = 0; f < MAXF; ++f) {
          if (f < nfr && 16 * f + l15 < npx) {
            const int pix = hr0 * PX + 16 * f + l15;
            const f16x8 hi = *(const f16x8*)(smem + OFF_A16 + q * PLANE + pix * 16);
            unsigned char* d = smem + OFF_A8 + (2 * (q >> 1)) * PLANE + pix * 16 + (q & 1) * 8;
            *(u32x2*)d = m_lo[f];
            *(u32x2*)(d + PLANE) = x2m_hi8(hi);
          }
        }
      };
      // (A/B, 8 x 512^2, with the hi8 words and the scales still held in registers: the fragments kept apart by scheduling barriers 197 us, left
      // to the compiler 179 us, spread over both step windows with the first five fragments' hi words held back 201 us -- 400 B of scratch
      // per lane; as it is now 171-181 us; the two launches 53 + 140-154 us)
      patch_load(0);
      patch_commit();
      first_conv(0);
      dma_weights((const unsigned char*)p.w16 + (long long)cob * W16, OFF_W16, W16);
      dma_weights((const unsigned char*)p.w8 + (long long)cob * W8, OFF_W8, W8);
      landed();
      lds_barrier();
      for (int k = 0; k < npairs; ++k) {
        m8_commit();                                                     // consumers: 16-bit step of tile k
        if (k + 1 < npairs) patch_load(k + 1);
        landed();
        lds_barrier();
        const bool pool_now = POOL && k > 0;
        if (k + 1 < npairs) { patch_commit(); first_conv(k + 1); }       // consumers: fp8 step of tile k
        if constexpr (POOL) { if (pool_now) pool_words(k - 1); }
        landed();
        if constexpr (POOL) { if (pool_now) pool_store(); }
        lds_barrier();
      }
      if constexpr (POOL) {
        lds_barrier();
        pool_words(npairs - 1);
        pool_store();
      }
      return;
    }
    dma16(0);
    landed();
    lds_barrier();
    for (int k = 0; k < npairs; ++k) {
#ifndef X2M_ABLATE_NO_COPIES
      dma8(k);
#endif
#if !defined(X2M_ABLATE_NO_HI8) && !defined(X2M_ABLATE_NO_COPIES)      // (tools/ab_build.sh -DX2M_ABLATE_NO_HI8: timing only, the fp8 operand's hi8 half stays stale)
      make_hi8();
#endif
      landed();
      lds_barrier();
#ifndef X2M_ABLATE_NO_COPIES
      if (k + 1 < npairs) dma16(k + 1);
#endif
      const bool pool_now = POOL && k > 0 && k % nchunk == 0;  // the tile that ended with pair k - 1 left its keys before this barrier
      if constexpr (POOL) { if (pool_now) pool_words(k / nchunk - 1); }
      landed();
      if constexpr (POOL) { if (pool_now) pool_store(); }
      lds_barrier();
    }
    if constexpr (POOL) {
      lds_barrier();                                           // the consumers' last epilogue
      pool_words(npairs / nchunk - 1);
      pool_store();
    }
    return;
  }

  // ==================================================================== consumer waves
  const int l15 = lane & 15, q = lane >> 4;
  const int row_first = wave * NR;                            // first tile row of this wave
  // 16-bit step: q & 1 = 8-channel half, q >> 1 = column of the pair (groups 0, 1) / sub-chunk (cross group)
  const int rbase16 = (q & 1) * PLANE + (row_first * PX + l15) * 16;
  const int hoff16[3] = {(q >> 1) * 16, 2 * PLANE + (q >> 1) * 16, (q >> 1) * 2 * PLANE + 2 * 16};
  // fp8 step: lane group q of K = 128 group G reads block G >> 1 at tap 4 (G & 1) + q
  int toff[4];
#pragma unroll
  for (int G = 0; G < 4; ++G) {
    const int tap = (G & 1) * 4 + q;
    toff[G] = (G >> 1) * 2 * PLANE + ((tap / 3) * PX + (tap % 3)) * 16;
  }
  const int rbase8 = (row_first * PX + l15) * 16;
  const int toff8 = (q >> 1) * PLANE + (2 * PX + 2) * 16 + (q & 1) * 8;       // tap 8 for the K = 32 instruction (+ block * 2 PLANE)

  f32x4 acc[2][NI];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NI; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

  f16x8 R16[2][FX][NR + 2], A16f[2][3][2];
  i32x8 R8f[2][NI], A8f[2][2];
  i64 R8n[2][NI], A8n[2][2];
  auto load_group = [&](int g, auto BUF) {
    constexpr int b = decltype(BUF)::value;
    const unsigned char* ab = smem + OFF_A16 + rbase16 + hoff16[g];
    const unsigned char* wl = smem + OFF_W16 + lane * 16;
#pragma unroll
    for (int xh = 0; xh < FX; ++xh)
#pragma unroll
      for (int r = 0; r < NR + 2; ++r) R16[b][xh][r] = *(const f16x8*)(ab + (r * PX + xh * 16) * 16);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      A16f[b][dy][0] = *(const f16x8*)(wl + ((g * 3 + dy) * 2 + 0) * 1024);
      A16f[b][dy][1] = *(const f16x8*)(wl + ((g * 3 + dy) * 2 + 1) * 1024);
    }
  };
  auto group_mfmas = [&](auto BUF) {
    constexpr int b = decltype(BUF)::value;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int n = 0; n < NI; ++n) {
        acc[0][n] = X2M_MFMA16(A16f[b][dy][0], R16[b][n % FX][n / FX + dy], acc[0][n]);
        acc[1][n] = X2M_MFMA16(A16f[b][dy][1], R16[b][n % FX][n / FX + dy], acc[1][n]);
      }
#pragma unroll
    for (int i = 0; i < 12; ++i) {                            // the next group's (12 .. 14) LDS reads between the 24 MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto rd128 = [&](const unsigned char* ptr, int second) -> i32x8 {
    const u32x4 lo = *(const u32x4*)ptr, hi = *(const u32x4*)(ptr + second);
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  auto load_g8 = [&](int G, auto BUF) {
    constexpr int b = decltype(BUF)::value;
    const unsigned char* ab = smem + OFF_A8 + rbase8 + toff[G];
#pragma unroll
    for (int m = 0; m < 2; ++m) A8f[b][m] = rd128(smem + OFF_W8 + (G * 2 + m) * 2048 + lane * 16, 1024);
#pragma unroll
    for (int n = 0; n < NI; ++n) R8f[b][n] = rd128(ab + ((n / FX) * PX + (n % FX) * 16) * 16, PLANE);
  };
  auto load_tap8 = [&]() {
#pragma unroll
    for (int bl = 0; bl < 2; ++bl) {
#pragma unroll
      for (int m = 0; m < 2; ++m) A8n[bl][m] = *(const i64*)(smem + OFF_W8 + W128 + (bl * 2 + m) * 512 + lane * 8);
#pragma unroll
      for (int n = 0; n < NI; ++n) R8n[bl][n] = *(const i64*)(smem + OFF_A8 + rbase8 + bl * 2 * PLANE + toff8 + ((n / FX) * PX + (n % FX) * 16) * 16);
    }
  };
  auto mfma_g8 = [&](auto BUF) {
    constexpr int b = decltype(BUF)::value;
#pragma unroll
    for (int n = 0; n < NI; ++n)
#pragma unroll
      for (int m = 0; m < 2; ++m)
        acc[m][n] = X2M_MFMA128(A8f[b][m], R8f[b][n], acc[m][n]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {                             // the next group's 12 reads between the 8 K = 128 instructions
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfma_tap8 = [&]() {
#pragma unroll
    for (int bl = 0; bl < 2; ++bl)
#pragma unroll
      for (int n = 0; n < NI; ++n)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m][n] = X2M_MFMA32F8(A8n[bl][m], R8n[bl][n], acc[m][n]);
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto tile_epilogue = [&](int tile) {
    int n_img, y0, x0;
    tile_origin(tile, n_img, y0, x0);
    f16* yout = (f16*)p.y + (long long)n_img * p.y_sstride;
    float bias_r[8], os_r[8];
    {
      const f32x4* ep = (const f32x4*)(smem + OFF_E) + 2 * q;
      const f32x4 w0 = ep[0], w1 = ep[1], b0 = ep[8], b1 = ep[9];
#pragma unroll
      for (int j = 0; j < 4; ++j) { os_r[j] = w0[j]; os_r[4 + j] = w1[j]; bias_r[j] = b0[j]; bias_r[4 + j] = b1[j]; }
    }
    const float relu_floor = p.epi == 2 ? 0.f : -65504.f;
    [[maybe_unused]] float hl[HEAD > 0 ? HEAD : 1];
    [[maybe_unused]] unsigned pool_k[2][8];                    // POOL: the keys of row 0's two fragments (common.h: x2m_pool_keys)
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int gy = y0 + row_first + n / FX, gx = x0 + (n % FX) * 16 + l15;
      const bool ok = gy < p.H && gx < p.W;
      const long long vo = (long long)gy * p.W + gx;
      float r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        r[j] = fmaf(j < 4 ? acc[0][n][j] : acc[1][n][j - 4], os_r[j], bias_r[j]);
        if constexpr (HEAD > 0) { if (p.epi == 2) r[j] = fmaxf(r[j], 0.f); }      // (the head variants sit at their register cap)
        else r[j] = __builtin_amdgcn_fmed3f(r[j], relu_floor, 65504.f);           // ReLU AND the split's range clamp in one v_med3: floor = 0 or -65504 (NaN -> the floor, as fmax / fmin gave)
      }
      if constexpr (HEAD > 0) {
        float lf[HEAD];
        x2m_head_logits<HEAD>(p, (const float*)(smem + OFF_E + 256), r, q, ok, lf);
#pragma unroll
        for (int c = 0; c < HEAD; ++c) hl[c] = q == n ? lf[c] : hl[c];
      } else {
      f16x8 hi, lo;
      u32x2 lo8, hi8;
      x2m_split8<true>(r, hi, lo, lo8, hi8);
      if (ok) {
        *(f16x8*)(yout + (long long)(cob * 4 + q) * plane_stride + vo * 8) = hi;
        if (p.y_lo >= 0) *(f16x8*)(yout + (long long)(p.y_lo + cob * 4 + q) * plane_stride + vo * 8) = lo;
        if (p.y8 != nullptr) {
          unsigned char* y8 = (unsigned char*)p.y8 + (long long)n_img * p.y8_sstride + x2m_off(cob * 4 + q, vo, nvox);
          *(u32x2*)y8 = lo8;
        }
        if (p.sat != nullptr) x2_note_saturation(p.sat, hi);
      }
      if constexpr (POOL) {
        unsigned kk[8];
        x2m_pool_keys(hi, lo8, kk);
        if (n < FX) {
#pragma unroll
          for (int j = 0; j < 8; ++j) pool_k[n % FX][j] = kk[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            kk[j] = max(kk[j], pool_k[n % FX][j]);             // the wave's two rows (y pair)
            kk[j] = max(kk[j], lane_xor1(kk[j]));              // x pair
          }
          if (!(l15 & 1)) {                                    // pooled pixel (row `wave`, x (n % FX) * 8 + l15 / 2) of the tile
            unsigned char* dst = smem + OFF_P + (((wave * 16 + (n % FX) * 8 + (l15 >> 1)) * 4 + q) * 32);
            *(u32x4*)dst = u32x4{kk[0], kk[1], kk[2], kk[3]};
            *(u32x4*)(dst + 16) = u32x4{kk[4], kk[5], kk[6], kk[7]};
          }
        }
      }
      }
      acc[0][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[1][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (HEAD > 0) {
      const int gy = y0 + row_first + q / FX, gx = x0 + (q % FX) * 16 + l15;     // this lane group's fragment
      x2m_head_store<HEAD>(p, hl, gy < p.H && gx < p.W, n_img, nvox, (long long)gy * p.W + gx, 0, gy, gx);
    }
  };

  lds_barrier();                                               // the first 16-bit step is in LDS
  load_group(0, I0{});
  __builtin_amdgcn_sched_barrier(0);
  for (int k = 0; k < npairs; ++k) {
    // ---- 16-bit step (its first group is in fragment set 0)
    load_group(1, I1{}); group_mfmas(I0{});
    load_group(2, I0{}); group_mfmas(I1{});
    lds_barrier();                                             // the 16-bit buffers are read; the fp8 buffers of pair k are filled
    load_g8(0, I0{});
    group_mfmas(I0{});
    // ---- fp8 step
    load_g8(1, I1{}); mfma_g8(I0{});
    load_g8(2, I0{}); mfma_g8(I1{});
    load_g8(3, I1{}); mfma_g8(I0{});
    load_tap8(); mfma_g8(I1{});
    lds_barrier();                                             // the fp8 buffers are read; the 16-bit buffers of pair k + 1 are filled
    load_group(0, I0{});                                       // (after the last pair: a harmless re-read)
    mfma_tap8();
    const int tile = k / nchunk;
    if (k - tile * nchunk == nchunk - 1) tile_epilogue(tile);
  }
  if constexpr (POOL) lds_barrier();                           // the loaders store the last tile's pool behind this one
}

template <int HEAD, bool POOL = false, bool FIRST = false>
int launch_x2m_2d(ConvX2MParams p, hipStream_t stream) {
  constexpr int PLANE = ((18 * 34 * 16 + 255) / 256) * 256;
  const int lds = 8 * PLANE + 18432 + X2M2_W8 + 256 + (HEAD > 0 ? 2048 : 0) + (POOL ? 8 * 16 * 4 * 32 : 0) + (FIRST ? 4 * 3 * 7 * 36 * 2 + 4 * 256 : 0);
  IUNET_SET_MAX_LDS((conv2_x2m_kernel<HEAD, POOL, FIRST>), lds);
  p.tilesZ = 1; p.tilesY = (p.H + 15) / 16; p.tilesX = (p.W + 31) / 32;
  const int ncob = p.Cout / 32;
  iunet_brick_shape(2, ncob, 1, p.tilesY, p.tilesX, &p.bz, &p.by, &p.bx);
  p.nbz = 1; p.nby = (p.tilesY + p.by - 1) / p.by; p.nbx = (p.tilesX + p.bx - 1) / p.bx;
  // slot groups: a brick clamped to a small tile grid holds fewer slots than the XCD has workgroups for this Cout tile
  const long long nbricks = (long long)p.N * p.nby * p.nbx;
  const int table = 8 * (ncob == 1 ? 32 : ncob == 2 ? 16 : ncob <= 4 ? 8 : 4);      // (conv3_v4.hip: iunet_conv3_v4_stats_parts)
  int groups = table / 8 / (p.by * p.bx);
  while (groups > 1 && nbricks / 8 < groups) groups >>= 1;
  if (groups < 1) groups = 1;
  const int gx = 8 * p.by * p.bx * groups;
  hipLaunchKernelGGL((conv2_x2m_kernel<HEAD, POOL, FIRST>), dim3(gx, ncob), dim3(8 * 64 + XM_NLT), lds, stream, p);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

template <bool SMALL, int HEAD = 0, bool POOL = false>
int launch_x2m(ConvX2MParams p, hipStream_t stream) {
  using TL = XMTile<SMALL>;
  constexpr int PZ = TL::TZ + 2, NPIX = PZ * 10 * 18;
  constexpr int PLANE16 = ((NPIX * 16 + 255) / 256) * 256, PLANE8 = PZ * 192 * 16;
  const int lds = 2 * PLANE16 + 30720 + 2 * PLANE8 + F8K_WSTEP + 256 + (HEAD > 0 ? 2048 : 0) + (POOL ? TL::NCW * 64 * 32 : 0);
  IUNET_SET_MAX_LDS((conv3_x2m_kernel<SMALL, HEAD, POOL>), lds);
  p.tilesZ = (p.D + TL::TZ - 1) / TL::TZ; p.tilesY = (p.H + TL::TY - 1) / TL::TY; p.tilesX = (p.W + TL::TX - 1) / TL::TX;
  const int ncob = p.Cout / 32;
  iunet_brick_shape(3, ncob, p.tilesZ, p.tilesY, p.tilesX, &p.bz, &p.by, &p.bx);
  p.nbz = (p.tilesZ + p.bz - 1) / p.bz; p.nby = (p.tilesY + p.by - 1) / p.by; p.nbx = (p.tilesX + p.bx - 1) / p.bx;
  const int gx = 8 * p.bz * p.by * p.bx;
  hipLaunchKernelGGL((conv3_x2m_kernel<SMALL, HEAD, POOL>), dim3(gx, ncob), dim3(TL::NCW * 64 + XM_NLT), lds, stream, p);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

// ------------------------------------------------------------------ operator preparation
// One workgroup per output channel: BatchNorm fold (the oracle's fp32 operation order, as x2_prep_kernel), row scale s = 2^k with
// max |w'| s in [2^9, 2^10), hi = f16(w' s), res = w' s - hi (exact).  Writes w_hi as fp32 [Cout][Cin][27] (iunet_pack_conv3 mode 2 makes
// the K16 operator of it) and the K128 operator of the virtual channels [w_hi8 = e4m3(hi / 16) | w_lo8 = e4m3(res * 256)] per 16-channel chunk.
__device__ __forceinline__ unsigned xm_e4m3(float v) {
  const int a = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v, -448.0f, 448.0f), 0.f, 0, false);
  return (unsigned)a & 0xffu;
}
__device__ __forceinline__ void x2m_prep_row(const float* __restrict__ w, float* __restrict__ whi, unsigned char* __restrict__ w8,
                                             float* __restrict__ oscale, float* __restrict__ bias_out,
                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                             const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                             float act_in, float act_out, int Cout, int Cin, int taps, int co, float* red) {
#pragma clang fp contract(off)
  const int tid = threadIdx.x;
  float a = 1.0f;
  if (gamma) { const float s = var[co] + eps; a = gamma[co] / sqrtf(s); }
  const int n = Cin * taps;
  const float* wc = w + (long long)co * n;
  float m = 0.f;
  for (int i = tid; i < n; i += 256) m = fmaxf(m, fabsf(gamma ? wc[i] * a : wc[i]));
  red[tid] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
  m = red[0];
  float s = 1.0f;
  if (m > 0.f && m < INFINITY) {
    int e;
    (void)frexpf(m, &e);
    int k = 10 - e;
    k = k < -40 ? -40 : k > 40 ? 40 : k;
    s = ldexpf(1.0f, k);
  }
  const int cob = co >> 5, r32 = co & 31;
  const int mt = (r32 >> 2) & 1, row = (r32 >> 3) * 4 + (r32 & 3);       // co = cob * 32 + 8 (row >> 2) + 4 m + (row & 3)
  const int nchunk = Cin >> 4;
  // one thread = the 8 consecutive input channels of one tap: 8 fp32 words of w_hi + two 8-byte pieces of the K128 block
  for (int i = tid; i < (Cin >> 3) * taps; i += 256) {
    const int g8 = i / taps, tap = i - g8 * taps;
    const int chunk = g8 >> 1, o = g8 & 1;
    const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3, col = dz * 3 + dx;
    unsigned long long ph = 0, pl = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = g8 * 8 + j;
      const float v0 = wc[(long long)ci * taps + tap];
      float v = (gamma ? v0 * a : v0) * s;
      v = fminf(fmaxf(v, -65504.f), 65504.f);
      const f16 h = (f16)v;
      const float res = v - (float)h;
      whi[(long long)co * n + (long long)ci * taps + tap] = (float)h;
      ph |= (unsigned long long)xm_e4m3((float)h * 0.0625f) << (8 * j);
      pl |= (unsigned long long)xm_e4m3(res * 256.0f) << (8 * j);
    }
    if (taps == 9) {          // 2-D: blocks of 32 input channels = two virtual blocks (conv2_x2m_kernel)
      unsigned char* blk = w8 + ((long long)cob * (nchunk >> 1) + (chunk >> 1)) * X2M2_W8;
      *(unsigned long long*)(blk + x2m2_w8_offset(tap, mt, chunk & 1, 0, o, row)) = ph;
      *(unsigned long long*)(blk + x2m2_w8_offset(tap, mt, chunk & 1, 1, o, row)) = pl;
      continue;
    }
    unsigned char* blk = w8 + ((long long)cob * nchunk + chunk) * F8K_WSTEP;
    *(unsigned long long*)(blk + f8k_offset(col, dy, mt, 0, o, row)) = ph;      // virtual channels 0..15: against x_lo8
    *(unsigned long long*)(blk + f8k_offset(col, dy, mt, 1, o, row)) = pl;      // virtual channels 16..31: against x_hi8
  }
  if (tid == 0) {
    oscale[co] = act_out / (act_in * s);
    float b = 0.f;
    if (gamma) { const float t = mean[co] * a; b = beta[co] - t; }
    bias_out[co] = b * act_out;
  }
}
__global__ __launch_bounds__(256) void x2m_prep_kernel(const float* __restrict__ w, float* __restrict__ whi, unsigned char* __restrict__ w8,
                                                      float* __restrict__ oscale, float* __restrict__ bias_out,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                      float act_in, float act_out, int Cout, int Cin, int taps) {
  __shared__ float red[256];
  x2m_prep_row(w, whi, w8, oscale, bias_out, gamma, beta, mean, var, eps, act_in, act_out, Cout, Cin, taps, blockIdx.x, red);
}
// every x2m stage conv of a table (x2_prep_desc.h, kind 3) in one launch: workgroup -> (operator, output channel)
__global__ __launch_bounds__(256) void x2m_prep_batch_kernel(const X2PrepDesc* __restrict__ table, int n) {
  __shared__ float red[256];
  int i = 0;
  while (i + 1 < n && (int)blockIdx.x >= table[i + 1].row0) ++i;
  const X2PrepDesc d = table[i];
  const int co = (int)blockIdx.x - d.row0;
  if (co >= d.Cout) return;
  x2m_prep_row(d.w, d.out, d.w8, d.oscale, d.bias_out, d.gamma, d.beta, d.mean, d.var, d.eps, d.act_in, d.act_out, d.Cout, d.Cin, d.taps, co, red);
}

// ------------------------------------------------------------------ lo8 planes of a tensor that some other kernel wrote as hi + lo words
// (first conv, max-pool, transposed conv): one thread = one voxel of one 16-channel chunk
__global__ __launch_bounds__(256) void x2m_make8_kernel(const f16* __restrict__ x, long long x_ss, int x_lo, unsigned char* __restrict__ y8,
                                                       long long y8_ss, int chunks, long long vox) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= vox) return;
  const int c = blockIdx.y, n = blockIdx.z;
  const f16* xl = x + n * x_ss + ((long long)(2 * c + x_lo) * vox + i) * 8;
  u32x4 o_lo;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f16x8 vl = *(const f16x8*)(xl + (long long)h * vox * 8);
    float l4[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) l4[j] = (float)vl[j] * 16.0f;
    const u32x2 w = x2m_pack8(l4);
    o_lo[2 * h] = w[0]; o_lo[2 * h + 1] = w[1];
  }
  *(u32x4*)(y8 + n * y8_ss + ((long long)c * vox + i) * 16) = o_lo;
  (void)chunks;
}

// ------------------------------------------------------------------ max-pool 2^d on (hi, lo8): the larger hi + lo8 / 16 wins -- the order
// of common.h's x2m_pool_keys, shared with the conv epilogue that pools on the way out -- and its hi word and lo8 byte are copied: the
// pooled tensor holds exactly the values its source holds for a 3x3x3 consumer.  One thread = one output voxel of one 16-channel chunk: two
// hi planes (16 B each) and the chunk's whole lo8 granule per input voxel -- every access a full 16-byte item, consecutive threads on
// consecutive voxels (a thread per 8-channel plane read half granules: 1.8 TB/s at 128^3).
template <int ND>
__global__ __launch_bounds__(256) void x2m_maxpool_kernel(const f16* __restrict__ x, long long x_ss, const unsigned char* __restrict__ x8,
                                                         long long x8_ss, f16* __restrict__ y, long long y_ss, unsigned char* __restrict__ y8,
                                                         long long y8_ss, int chunks, int Do, int Ho, int Wo) {
  const long long ovox = (long long)Do * Ho * Wo;
  const long long total = ovox * chunks;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int n = blockIdx.y;
  const int c = (int)(i / ovox);
  const long long r = i - (long long)c * ovox;
  const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), oz = (int)(r / ((long long)Wo * Ho));
  const int Di = ND == 3 ? Do * 2 : 1, Hi = Ho * 2, Wi = Wo * 2;
  const long long ivox = (long long)Di * Hi * Wi;
  const f16* xh = x + n * x_ss + (long long)(2 * c) * ivox * 8;
  const unsigned char* xm = x8 + n * x8_ss + (long long)c * ivox * 16;
  unsigned k0[8], k1[8];                                       // the winners so far (common.h: x2m_pool_keys), channels 0..7 / 8..15
#pragma unroll
  for (int j = 0; j < 8; ++j) { k0[j] = 0u; k1[j] = 0u; }
#pragma unroll
  for (int a = 0; a < (ND == 3 ? 2 : 1); ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int z = ND == 3 ? oz * 2 + a : 0;
        const long long vi = ((long long)z * Hi + oy * 2 + b) * Wi + ox * 2 + cc;
        const f16x8 v0 = *(const f16x8*)(xh + vi * 8), v1 = *(const f16x8*)(xh + (ivox + vi) * 8);
        const u32x4 l8 = *(const u32x4*)(xm + vi * 16);       // (the hi8 granule is a function of the hi words: not read)
        unsigned c0[8], c1[8];
        x2m_pool_keys(v0, u32x2{l8[0], l8[1]}, c0);
        x2m_pool_keys(v1, u32x2{l8[2], l8[3]}, c1);
#pragma unroll
        for (int j = 0; j < 8; ++j) { k0[j] = max(k0[j], c0[j]); k1[j] = max(k1[j], c1[j]); }
      }
  f16x8 o0, o1;
  u32x2 l0, l1;
  x2m_pool_unkeys(k0, o0, l0);
  x2m_pool_unkeys(k1, o1, l1);
  f16* yo = y + n * y_ss + ((long long)(2 * c) * ovox + r) * 8;
  *(f16x8*)yo = o0;
  *(f16x8*)(yo + ovox * 8) = o1;
  *(u32x4*)(y8 + n * y8_ss + ((long long)c * ovox + r) * 16) = u32x4{l0[0], l0[1], l1[0], l1[1]};
}

}  // namespace

extern "C" {

/* bytes of the K128 operator of a 3x3x3 split conv with the cross terms on the fp8 matrix cores (iunet_x2m_prep) */
long long iunet_x2m_w8_bytes(int Cout, int Cin) { return (long long)(Cout / 32) * (Cin / 16) * F8K_WSTEP; }
/* the same for nd = 2 (3 x 3 filters: 18 432 bytes per 32 x 32 block) or 3 */
long long iunet_x2m_w8_bytes_nd(int nd, int Cout, int Cin) {
  return nd == 2 ? (long long)(Cout / 32) * (Cin / 32) * X2M2_W8 : iunet_x2m_w8_bytes(Cout, Cin);
}

/* operator of a 3x3x3 stage conv in the x2m form: whi = fp32 [Cout][Cin][27] holding w_hi (feed it to iunet_pack_conv3, dtype 0, mode 2),
 * w8 = iunet_x2m_w8_bytes bytes (the K128 operator of [w_hi8 | w_lo8]); oscale / bias_out as iunet_x2_prep */
int iunet_x2m_prep_nd(int nd, const void* w, void* whi, void* w8, void* oscale, void* bias_out, const void* gamma, const void* beta,
                      const void* mean, const void* var, float eps, float act_in, float act_out, int Cout, int Cin, void* stream);
int iunet_x2m_prep(const void* w, void* whi, void* w8, void* oscale, void* bias_out, const void* gamma, const void* beta, const void* mean,
                   const void* var, float eps, float act_in, float act_out, int Cout, int Cin, void* stream) {
  return iunet_x2m_prep_nd(3, w, whi, w8, oscale, bias_out, gamma, beta, mean, var, eps, act_in, act_out, Cout, Cin, stream);
}
/* nd = 2: w fp32 [Cout][Cin][9] -> whi (feed it to iunet_pack_conv3, dtype 0, mode 6: the cross-pair order) + the 2-D K128 operator */
int iunet_x2m_prep_nd(int nd, const void* w, void* whi, void* w8, void* oscale, void* bias_out, const void* gamma, const void* beta,
                      const void* mean, const void* var, float eps, float act_in, float act_out, int Cout, int Cin, void* stream) {
  IUNET_REQUIRE(nd == 2 || nd == 3, "x2m_prep: nd must be 2 or 3");
  IUNET_REQUIRE(w && whi && w8 && oscale && bias_out, "x2m_prep: null pointer");
  IUNET_REQUIRE(Cout > 0 && Cout % 32 == 0 && Cin > 0 && Cin % 32 == 0, "x2m_prep: channels must be positive multiples of 32 (%d, %d)", Cout, Cin);
  IUNET_REQUIRE(!gamma || (beta && mean && var), "x2m_prep: a BatchNorm fold needs gamma, beta, mean and var");
  int e1, e2;
  IUNET_REQUIRE(act_in > 0.f && act_out > 0.f && frexpf(act_in, &e1) == 0.5f && frexpf(act_out, &e2) == 0.5f,
                "x2m_prep: the activation scales must be powers of two (got %g, %g)", act_in, act_out);
  hipLaunchKernelGGL(x2m_prep_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, (const float*)w, (float*)whi, (unsigned char*)w8,
                     (float*)oscale, (float*)bias_out, (const float*)gamma, (const float*)beta, (const float*)mean, (const float*)var,
                     eps, act_in, act_out, Cout, Cin, nd == 3 ? 27 : 9);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

/* iunet_x2m_prep_nd for every stage conv of a device-resident table of n X2PrepDesc rows (kind 3; taps = 27 or 9) in ONE launch of `rows`
 * workgroups (the table's running sum of Cout): the same kernel body per row, the same bits */
int iunet_x2m_prep_batch(const void* table, int n, int rows, void* stream) {
  IUNET_REQUIRE(table != nullptr && n > 0 && rows > 0, "x2m_prep_batch: empty table");
  hipLaunchKernelGGL(x2m_prep_batch_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const X2PrepDesc*)table, n);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

/* m8 planes (x8: [2 C / 16][D][H][W][16 B] per sample, x8_ss bytes apart) of a split tensor of C channels (hi planes at x, lo planes
 * x_lo planes further on; x_ss elements) */
int iunet_x2m_make8(const void* x, long long x_ss, int x_lo, void* x8, long long x8_ss, int C, int N, int D, int H, int W, void* stream) {
  IUNET_REQUIRE(x && x8, "x2m_make8: null pointer");
  IUNET_REQUIRE(C > 0 && C % 16 == 0, "x2m_make8: C must be a multiple of 16 (got %d)", C);
  IUNET_REQUIRE_GRID("x2m_make8", N, D, H, W);
  const long long vox = (long long)D * H * W;
  dim3 grid((unsigned)((vox + 255) / 256), C / 16, N);
  hipLaunchKernelGGL(x2m_make8_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, x_ss, x_lo, (unsigned char*)x8, x8_ss, C / 16, vox);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

/* 2^d max-pool of a tensor in the x2m form: hi planes + m8 planes in (x_ss elements / x8_ss bytes per sample), the same out; Do, Ho, Wo =
 * output grid.  The winner is the larger hi + lo8 / 16 (what a 3x3x3 consumer would read); its hi word and m8 bytes are copied. */
int iunet_x2m_maxpool_fwd(int nd, const void* x, long long x_ss, const void* x8, long long x8_ss, void* y, long long y_ss, void* y8,
                          long long y8_ss, int C, int N, int Do, int Ho, int Wo, void* stream) {
  IUNET_REQUIRE(x && x8 && y && y8, "x2m_maxpool: null pointer");
  IUNET_REQUIRE(nd == 2 || nd == 3, "x2m_maxpool: nd must be 2 or 3");
  IUNET_REQUIRE(C > 0 && C % 16 == 0 && N > 0 && Do > 0 && Ho > 0 && Wo > 0, "x2m_maxpool: bad shape");
  const long long total = (long long)Do * Ho * Wo * (C / 16);
  dim3 grid((unsigned)((total + 255) / 256), N);
  if (nd == 3) hipLaunchKernelGGL((x2m_maxpool_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, x_ss, (const unsigned char*)x8, x8_ss,
                                  (f16*)y, y_ss, (unsigned char*)y8, y8_ss, C / 16, Do, Ho, Wo);
  else hipLaunchKernelGGL((x2m_maxpool_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, x_ss, (const unsigned char*)x8, x8_ss,
                          (f16*)y, y_ss, (unsigned char*)y8, y8_ss, C / 16, Do, Ho, Wo);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

/* 3x3x3 stage conv of the split-precision forward, cross terms on the fp8 matrix cores.  x: Cin / 8 hi planes (x_ss elements per sample),
 * x8: the m8 planes of the same tensor (x8_ss bytes per sample); y: Cout / 8 hi planes, the lo planes y_lo planes further on (y_lo < 0: no
 * lo planes -- a tensor only 3x3x3 convs read), y8: its m8 planes or null; w16 / w8 / oscale / bias from iunet_x2m_prep (+ iunet_pack_conv3);
 * epi as iunet_conv3_fwd; sat: optional device int, raised (atomicMax) to the bit pattern of a saturated hi word */
int iunet_x2m_conv_fwd(int nd, const void* x, long long x_ss, const void* x8, long long x8_ss, void* y, long long y_ss, int y_lo, void* y8,
                       long long y8_ss, const void* w16, const void* w8, const void* oscale, const void* bias, int N, int D, int H, int W,
                       int Cin, int Cout, int epi, void* sat, void* stream);
int iunet_x2m_conv3_fwd(const void* x, long long x_ss, const void* x8, long long x8_ss, void* y, long long y_ss, int y_lo, void* y8,
                        long long y8_ss, const void* w16, const void* w8, const void* oscale, const void* bias, int N, int D, int H, int W,
                        int Cin, int Cout, int epi, void* sat, void* stream) {
  return iunet_x2m_conv_fwd(3, x, x_ss, x8, x8_ss, y, y_ss, y_lo, y8, y8_ss, w16, w8, oscale, bias, N, D, H, W, Cin, Cout, epi, sat, stream);
}
static int x2m_conv_impl(const char* who, int nd, const void* x, long long x_ss, const void* x8, long long x8_ss, void* y, long long y_ss, int y_lo,
                         void* y8, long long y8_ss, void* py, long long py_ss, void* py8, long long py8_ss, const void* w16, const void* w8,
                         const void* oscale, const void* bias, int N, int D, int H, int W, int Cin, int Cout, int epi, void* sat, void* stream) {
  IUNET_REQUIRE(nd == 2 || nd == 3, "%s: nd must be 2 or 3", who);
  IUNET_REQUIRE(nd == 3 || D == 1, "%s: 2-D needs D == 1", who);
  IUNET_REQUIRE(x && x8 && y && w16 && w8 && oscale, "%s: null pointer", who);
  IUNET_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "%s: bad shape N %d, %d x %d x %d", who, N, D, H, W);
  IUNET_REQUIRE(Cin >= 32 && Cout >= 32 && Cin % 32 == 0 && Cout % 32 == 0, "%s: channels must be positive multiples of 32 (%d -> %d)", who, Cin, Cout);
  IUNET_REQUIRE(epi >= 0 && epi <= 2, "%s: bad epilogue %d", who, epi);
  IUNET_REQUIRE(epi == 0 || bias != nullptr, "%s: epilogue %d needs a bias", who, epi);
  const bool pool = py != nullptr;
  if (pool) {
    IUNET_REQUIRE(py8 != nullptr, "%s: the pooled tensor needs its m8 planes", who);
    IUNET_REQUIRE(H % 2 == 0 && W % 2 == 0 && (nd == 2 || D % 2 == 0), "%s: the pooled grid needs even sizes (%d, %d, %d)", who, D, H, W);
  }
  ConvX2MParams p;
  p.x = x; p.x_sstride = x_ss; p.x8 = x8; p.x8_sstride = x8_ss; p.y = y; p.y_sstride = y_ss; p.y_lo = y_lo; p.y8 = y8; p.y8_sstride = y8_ss;
  p.w16 = w16; p.w8 = w8; p.oscale = (const float*)oscale; p.bias = (const float*)bias;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.epi = epi; p.sat = (int*)sat;
  p.tilesZ = p.tilesY = p.tilesX = 0;
  p.bz = p.by = p.bx = p.nbz = p.nby = p.nbx = 0;
  p.head_w = p.head_b = nullptr; p.inv_act = 0.f; p.logits = p.probs = nullptr; p.cls = nullptr;
  p.oN = p.oC = p.oD = p.oH = p.oW = 0; p.divisor = 1.f; p.accumulate = 0;
  p.pool_y = py; p.pool_y_ss = py_ss; p.pool_y8 = py8; p.pool_y8_ss = py8_ss;
  p.f_x = p.f_w = nullptr; p.f_oscale = p.f_bias = nullptr; p.f_sN = p.f_sH = p.f_sW = 0; p.f_dtype = 0; p.f_act = 0.f;
  hipStream_t s = (hipStream_t)stream;
  if (nd == 2) return pool ? launch_x2m_2d<0, true>(p, s) : launch_x2m_2d<0>(p, s);
  // the tile size follows the grid as in the 16-bit launch; the summation order of a voxel does not depend on it
  const long long big_tiles = (long long)N * ((D + 3) / 4) * ((H + 7) / 8) * ((W + 15) / 16);
  const bool small = big_tiles * (Cout / 32) < 128;
  if (pool) return small ? launch_x2m<true, 0, true>(p, s) : launch_x2m<false, 0, true>(p, s);
  return small ? launch_x2m<true>(p, s) : launch_x2m<false>(p, s);
}

/* the same for nd = 2 (3 x 3 filters, D == 1; operators from iunet_x2m_prep_nd(2, ..) + iunet_pack_conv3 mode 6) or 3 */
int iunet_x2m_conv_fwd(int nd, const void* x, long long x_ss, const void* x8, long long x8_ss, void* y, long long y_ss, int y_lo, void* y8,
                       long long y8_ss, const void* w16, const void* w8, const void* oscale, const void* bias, int N, int D, int H, int W,
                       int Cin, int Cout, int epi, void* sat, void* stream) {
  return x2m_conv_impl("x2m_conv3", nd, x, x_ss, x8, x8_ss, y, y_ss, y_lo, y8, y8_ss, nullptr, 0, nullptr, 0, w16, w8, oscale, bias, N, D, H, W,
                       Cin, Cout, epi, sat, stream);
}

/* 1 where the callers (net.hip, engine_x2.py) let the pool ride in the conv with C channels: in 3-D (2 x 128^3, one box, 3 bytes per
 * element: 32->32 @ 128^3 441 us against 441 + 92 for conv + pool, 64->64 @ 64^3 190 against 184 + 19, 128->128 @ 32^3 91 against 89 + 5) and
 * in 2-D (8 x 512^2: 132 against 130 + 39, 64->64 @ 256^2 90 against 80 + 20, 79 against 77 + 14; at 4 bytes per element the 64-channel launch
 * -- two Cout tiles on two steps per tile -- read 107-113 against 79 + 22 and was excluded; the whole forward with it: 1.55-1.57 against
 * 1.56-1.57 ms at 8 x 512^2, 1.60-1.61 against 1.60-1.63 at 128 x 128^2).  IUNET_X2M_POOL=0: never, =3: 3-D only, =4: 2-D without the
 * 64-channel stage (A/B switches; unset or any other value: everywhere, the default). */
int iunet_x2m_pool_fusable(int nd, int C) {
  static const int mode = getenv("IUNET_X2M_POOL") ? atoi(getenv("IUNET_X2M_POOL")) : 1;
  if (mode == 0 || (nd != 2 && nd != 3)) return 0;
  if (nd == 3) return 1;
  return mode == 3 ? 0 : mode == 4 ? C != 64 : 1;
}

/* An encoder stage's second conv (unet.py:63-69: the skip tensor) WITH the stage's 2^d max-pool riding along: y / y8 as iunet_x2m_conv_fwd,
 * and py / py8 = the pooled tensor (hi planes, py_ss elements per sample; m8 planes, py8_ss bytes per sample) of the grid D/2 (nd = 3), H/2,
 * W/2 -- the words iunet_x2m_maxpool_fwd makes of y / y8, bit for bit, without reading them back */
int iunet_x2m_conv_pool_fwd(int nd, const void* x, long long x_ss, const void* x8, long long x8_ss, void* y, long long y_ss, int y_lo, void* y8,
                            long long y8_ss, void* py, long long py_ss, void* py8, long long py8_ss, const void* w16, const void* w8,
                            const void* oscale, const void* bias, int N, int D, int H, int W, int Cin, int Cout, int epi, void* sat, void* stream) {
  IUNET_REQUIRE(py && py8, "x2m_conv_pool: null pooled tensor");
  return x2m_conv_impl("x2m_conv_pool", nd, x, x_ss, x8, x8_ss, y, y_ss, y_lo, y8, y8_ss, py, py_ss, py8, py8_ss, w16, w8, oscale, bias, N, D, H, W,
                       Cin, Cout, epi, sat, stream);
}

/* 1 where the callers (net.hip, engine_x2.py) run the first encoder stage as ONE launch: 2-D, one input channel, 32 channels at level 0, a
 * batch of at least 2 048 tiles of 16 x 32 pixels (8 per CU) -- the loader waves' first conv is the longer side of a tile step, and the
 * first tile's has nothing to hide behind -- AND a stage whose max-pool does not ride in the second conv (iunet_x2m_pool_fusable): the
 * callers' stage 0 always feeds a pool, and at 3 bytes per element the launch with first conv AND pool on the loader waves loses to first
 * conv + pooled conv (8 x 512^2: 218 us against 47 + 140; 128 x 128^2: 219 against 49 + 119; without the pool in the launch 161 + 38 against
 * 47 + 128 + 38: tools/bench_first_stage.py).  IUNET_X2M_FIRST=0: never, =2: whatever the batch and the pool (A/B and test switch) */
int iunet_x2m_first_stage_fusable(int nd, int cin, int c0, int N, int H, int W) {
  static const int mode = getenv("IUNET_X2M_FIRST") ? atoi(getenv("IUNET_X2M_FIRST")) : 1;
  if (mode == 0 || nd != 2 || cin != 1 || c0 != 32 || N < 1 || H < 1 || W < 1) return 0;
  if (mode >= 2) return 1;
  return !iunet_x2m_pool_fusable(nd, c0) && (long long)N * ((H + 15) / 16) * ((W + 31) / 32) >= 2048;
}

/* The FIRST ENCODER STAGE of the 2-D network as one launch (unet.py:63-69: conv 1 -> 32 + BatchNorm + ReLU, conv 32 -> 32 + BatchNorm + ReLU):
 * the second conv (iunet_x2m_conv_fwd's operators w16 / w8 / oscale / bias, epilogue 2) whose loader waves compute the first conv on the way
 * in, from the caller's one-channel image (x, in_dtype / in_strides as iunet_x2m_first_conv_fwd) with that conv's own operator (fw / f_oscale
 * / f_bias / act_scale: what iunet_x2m_first_conv_fwd takes) -- the 32-channel tensor between the two convs is never written.  y / y8 (and,
 * with py != NULL, the 2 x 2 max-pool py / py8 as iunet_x2m_conv_pool_fwd) hold iunet_x2m_first_conv_fwd + iunet_x2m_conv_fwd (+ pool) bit
 * for bit. */
int iunet_x2m_first_stage_fwd(const void* x, int in_dtype, const long long* in_strides, const void* fw, const void* f_oscale, const void* f_bias,
                              float act_scale, void* y, long long y_ss, int y_lo, void* y8, long long y8_ss, void* py, long long py_ss, void* py8,
                              long long py8_ss, const void* w16, const void* w8, const void* oscale, const void* bias, int N, int H, int W,
                              void* sat, void* stream) {
  IUNET_REQUIRE(x && in_strides && fw && f_oscale && f_bias && y && w16 && w8 && oscale && bias, "x2m_first_stage: null pointer");
  IUNET_REQUIRE(in_dtype >= 0 && in_dtype <= 3, "x2m_first_stage: bad input dtype %d", in_dtype);
  IUNET_REQUIRE(N > 0 && H > 0 && W > 0, "x2m_first_stage: bad shape N %d, %d x %d", N, H, W);
  IUNET_REQUIRE((py == nullptr) == (py8 == nullptr), "x2m_first_stage: the pooled tensor is hi planes AND m8 planes");
  IUNET_REQUIRE(py == nullptr || (H % 2 == 0 && W % 2 == 0), "x2m_first_stage: the pooled grid needs even sizes (%d, %d)", H, W);
  int e1;
  IUNET_REQUIRE(act_scale > 0.f && frexpf(act_scale, &e1) == 0.5f, "x2m_first_stage: the activation scale must be a power of two (got %g)", act_scale);
  ConvX2MParams p;
  p.x = nullptr; p.x_sstride = 0; p.x8 = nullptr; p.x8_sstride = 0; p.y = y; p.y_sstride = y_ss; p.y_lo = y_lo; p.y8 = y8; p.y8_sstride = y8_ss;
  p.w16 = w16; p.w8 = w8; p.oscale = (const float*)oscale; p.bias = (const float*)bias;
  p.N = N; p.D = 1; p.H = H; p.W = W; p.Cin = 32; p.Cout = 32; p.epi = 2; p.sat = (int*)sat;
  p.tilesZ = p.tilesY = p.tilesX = 0;
  p.bz = p.by = p.bx = p.nbz = p.nby = p.nbx = 0;
  p.head_w = p.head_b = nullptr; p.inv_act = 0.f; p.logits = p.probs = nullptr; p.cls = nullptr;
  p.oN = p.oC = p.oD = p.oH = p.oW = 0; p.divisor = 1.f; p.accumulate = 0;
  p.pool_y = py; p.pool_y_ss = py_ss; p.pool_y8 = py8; p.pool_y8_ss = py8_ss;
  p.f_x = x; p.f_sN = in_strides[0]; p.f_sH = in_strides[3]; p.f_sW = in_strides[4]; p.f_dtype = in_dtype;
  p.f_w = fw; p.f_oscale = (const float*)f_oscale; p.f_bias = (const float*)f_bias; p.f_act = act_scale;
  hipStream_t s = (hipStream_t)stream;
  return py != nullptr ? launch_x2m_2d<0, true, true>(p, s) : launch_x2m_2d<0, false, true>(p, s);
}

/* 1 if iunet_x2m_conv_head_fwd takes this head (2 or 3 classes on 32 feature channels), else 0: the caller then runs the conv into hi + lo
 * planes and iunet_x2_head_fwd on them */
int iunet_x2m_head_fusable(int ncls, int C0) {
  static const int off = getenv("IUNET_X2M_HEAD") ? (atoi(getenv("IUNET_X2M_HEAD")) == 0) : 0;      // A/B switch: IUNET_X2M_HEAD=0 keeps the head its own launch
  return !off && ncls >= 2 && ncls <= 3 && C0 == 32;      // (4 classes: the epilogue's registers spill beside the step loop's)
}

/* The LAST stage conv of the network with the 1x1 head + softmax + class map in its epilogue (unet.py:63-69, predict.py:38): the conv's
 * 32 output channels are never written -- the head works on the split words the conv would have stored, in the unfused head's own fmaf
 * chain, so logits / probs / cls are iunet_x2m_conv_fwd + iunet_x2_head_fwd bit for bit.  Output contract of iunet_head_fwd
 * (out_strides n, c, d, h, w in elements; probs = ((accumulate ? probs : 0) + p) / divisor; cls uint8 [N][D*H*W]). */
int iunet_x2m_conv_head_fwd(int nd, const void* x, long long x_ss, const void* x8, long long x8_ss, const void* w16, const void* w8,
                            const void* oscale, const void* bias, const void* head_w, const void* head_b, float act_scale, int ncls,
                            void* logits, void* probs, void* cls, const long long* out_strides, float divisor, int accumulate, int N, int D,
                            int H, int W, int Cin, void* sat, void* stream) {
  IUNET_REQUIRE(nd == 2 || nd == 3, "x2m_conv_head: nd must be 2 or 3");
  IUNET_REQUIRE(nd == 3 || D == 1, "x2m_conv_head: 2-D needs D == 1");
  IUNET_REQUIRE(x && x8 && w16 && w8 && oscale && bias && head_w && head_b && out_strides, "x2m_conv_head: null pointer");
  IUNET_REQUIRE_GRID("x2m_conv_head", N, D, H, W);
  IUNET_REQUIRE(Cin >= 32 && Cin % 32 == 0, "x2m_conv_head: Cin must be a positive multiple of 32 (got %d)", Cin);
  IUNET_REQUIRE(ncls == 2 || ncls == 3, "x2m_conv_head: 2 or 3 classes (got %d)", ncls);
  int e1;
  IUNET_REQUIRE(act_scale > 0.f && frexpf(act_scale, &e1) == 0.5f, "x2m_conv_head: the activation scale must be a power of two (got %g)", act_scale);
  ConvX2MParams p;
  p.x = x; p.x_sstride = x_ss; p.x8 = x8; p.x8_sstride = x8_ss; p.y = nullptr; p.y_sstride = 0; p.y_lo = -1; p.y8 = nullptr; p.y8_sstride = 0;
  p.w16 = w16; p.w8 = w8; p.oscale = (const float*)oscale; p.bias = (const float*)bias;
  p.N = N; p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = 32; p.epi = 2; p.sat = (int*)sat;
  p.tilesZ = p.tilesY = p.tilesX = 0;
  p.bz = p.by = p.bx = p.nbz = p.nby = p.nbx = 0;
  p.head_w = (const float*)head_w; p.head_b = (const float*)head_b; p.inv_act = 1.0f / act_scale;
  p.logits = (float*)logits; p.probs = (float*)probs; p.cls = (unsigned char*)cls;
  p.oN = out_strides[0]; p.oC = out_strides[1]; p.oD = out_strides[2]; p.oH = out_strides[3]; p.oW = out_strides[4];
  p.divisor = divisor; p.accumulate = accumulate;
  p.pool_y = p.pool_y8 = nullptr; p.pool_y_ss = p.pool_y8_ss = 0;
  p.f_x = p.f_w = nullptr; p.f_oscale = p.f_bias = nullptr; p.f_sN = p.f_sH = p.f_sW = 0; p.f_dtype = 0; p.f_act = 0.f;
  hipStream_t s = (hipStream_t)stream;
  if (nd == 2) return ncls == 2 ? launch_x2m_2d<2>(p, s) : launch_x2m_2d<3>(p, s);
  // (the tile size follows the grid as in iunet_x2m_conv_fwd: one summation order per voxel either way)
  const long long big_tiles = (long long)N * ((D + 3) / 4) * ((H + 7) / 8) * ((W + 15) / 16);
  if (big_tiles < 128)
    return ncls == 2 ? launch_x2m<true, 2>(p, s) : launch_x2m<true, 3>(p, s);
  return ncls == 2 ? launch_x2m<false, 2>(p, s) : launch_x2m<false, 3>(p, s);
}

}  // extern "C"
