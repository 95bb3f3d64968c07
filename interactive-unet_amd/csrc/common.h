// Common device/host helpers for libiunet (gfx950 only).
//
// Activation layout in HBM ("NHWC8c", channel-blocked NHWC): a tensor with C channels
// (C % 8 == 0) over a D x H x W grid is stored as C/8 planes, each plane
// [D][H][W][8] with the 8 channels of a block innermost (16 bytes per voxel per
// plane).  One MFMA operand fragment (16 consecutive x voxels x 8 channels) is then
// 256 contiguous bytes, LDS images are lane-linear (bank-conflict free for
// ds_read_b128) and a channel concat is a concatenation of planes (no copy).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 f16;
typedef __bf16 bf16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// A tensor view: `planes` channel planes of [D][H][W][8] elements starting at `ptr`;
// consecutive samples are `sample_stride` elements apart (lets a conv read or write a
// slice of a wider concat buffer).
struct TView {
  void* ptr;
  long long sample_stride;   // elements
  int planes;                // C/8 of this view
};

template <typename T> struct Vec8;
template <> struct Vec8<f16> { typedef f16x8 type; };
template <> struct Vec8<bf16> { typedef bf16x8 type; };

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

// fp16x2 split precision: v ~ hi + lo with hi = T(v), lo = T(v - hi) -- for T = f16 22 significant bits (the lo word is subnormal, i.e.
// exact to 2^-24, once |v| < 2^-2: callers keep activations scaled up by a power of two).  Finite by construction: |v| is clamped
// to the largest T below the rounding-to-infinity threshold.
template <typename T>
__device__ __forceinline__ void split16(float v, T& hi, T& lo) {
  v = fminf(fmaxf(v, -65504.f), 65504.f);
  hi = from_f32<T>(v);
  lo = from_f32<T>(v - to_f32<T>(hi));
}

template <typename T>
__device__ __forceinline__ f32x4 mfma16(typename Vec8<T>::type a, typename Vec8<T>::type b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mfma16<f16>(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mfma16<bf16>(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// XCD-aware block remap (cdna_hip_programming.md T1, bijective form): blocks b and b+8
// share an XCD/L2, so give each XCD a contiguous run of tiles (neighbouring tiles share
// halo voxels).  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is a full fence: it also waits vmcnt(0), i.e. for
// every global load a loader wave has just put in flight for a later step and for every output store of a consumer
// wave -- that wait (1-2 us per step) is exactly what the wave-specialised kernels must not pay.  Use where the only
// cross-wave communication of the step is through LDS.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

#define IUNET_OK 0
#define IUNET_ERR_ARG (-1)
#define IUNET_ERR_HIP (-2)
#define IUNET_ERR_UNSUPPORTED (-3)
#define IUNET_ERR_WORKSPACE (-4)

void iunet_set_error(const char* fmt, ...);
#define IUNET_CHECK_HIP(expr)                                                        \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      iunet_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return IUNET_ERR_HIP;                                                          \
    }                                                                                \
  } while (0)
#define IUNET_REQUIRE_GRID(what, N, D, H, W)                                          \
  IUNET_REQUIRE((N) > 0 && (D) > 0 && (H) > 0 && (W) > 0, what ": bad shape N %d, %d x %d x %d", (N), (D), (H), (W))
#define IUNET_REQUIRE(cond, ...)                                                     \
  do {                                                                               \
    if (!(cond)) { iunet_set_error(__VA_ARGS__); return IUNET_ERR_ARG; }             \
  } while (0)

// Opt a kernel in to more than 64 KB of dynamic LDS: once per (kernel, device), thread-safe.  The reference calls into
// this path from threads (app.py:737-739, :774-778) and a process may run the model on several GPUs (Engine(device=...)):
// a plain `static bool` would race, and would skip the call on the second device.  State = one slot per device holding
// the largest size set so far; racing threads at worst both make the (idempotent) call.
#include <atomic>
struct IunetLdsOnce { std::atomic<int> set[32]; };
inline int iunet_set_max_lds(IunetLdsOnce& st, const void* fn, int lds) {
  int dev = 0;
  IUNET_CHECK_HIP(hipGetDevice(&dev));
  const bool tracked = dev >= 0 && dev < 32;
  if (tracked && st.set[dev].load(std::memory_order_acquire) >= lds) return IUNET_OK;
  IUNET_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  if (tracked) {
    int cur = st.set[dev].load(std::memory_order_relaxed);
    while (cur < lds && !st.set[dev].compare_exchange_weak(cur, lds, std::memory_order_release)) {}
  }
  return IUNET_OK;
}
#define IUNET_SET_MAX_LDS(fn, lds)                                              \
  do {                                                                          \
    static IunetLdsOnce once_;                                                  \
    const int rc_ = iunet_set_max_lds(once_, (const void*)(fn), (lds));         \
    if (rc_ != IUNET_OK) return rc_;                                            \
  } while (0)

// Brick shape of the wave-specialised convolutions (conv3_v4.hip, conv3_f8.hip): the 32 / ncob workgroups that an XCD gives
// one Cout tile work on one compact brick of bz x by x bx tiles at a time.  A brick wider than the tile grid would make its
// surplus slots run fully masked -- whole workgroups of wasted work on the small grids of the deep levels (16^3: half of them,
// 8^3: three quarters) -- so every extent is clamped to the grid and the lost factor handed to the other axes.
inline void iunet_brick_shape(int nd, int ncob, int tilesZ, int tilesY, int tilesX, int* bz, int* by, int* bx) {
  auto p2 = [](int v) { int r = 1; while (r * 2 <= v) r *= 2; return r; };
  int z, y, x;
  if (nd == 3) {
    if (ncob == 1)      { z = 2; y = 4; x = 4; }
    else if (ncob == 2) { z = 2; y = 4; x = 2; }
    else if (ncob <= 4) { z = 2; y = 2; x = 2; }
    else                { z = 1; y = 2; x = 2; }
  } else {
    z = 1;
    if (ncob == 1)      { y = 4; x = 8; }
    else if (ncob == 2) { y = 4; x = 4; }
    else if (ncob <= 4) { y = 2; x = 4; }
    else                { y = 2; x = 2; }
  }
  if (nd == 3 && ncob == 1) {                    // A/B switch (IUNET_BRICK3="z y x", product 32): brick shape of the one-Cout-tile 3-D launches
    static const char* e = getenv("IUNET_BRICK3");
    int ez, ey, ex;
    if (e && sscanf(e, "%d %d %d", &ez, &ey, &ex) == 3 && ez * ey * ex == 32) { z = ez; y = ey; x = ex; }
  }
  const int want = z * y * x;
  const int mz = nd == 3 ? p2(tilesZ) : 1, my = p2(tilesY), mx = p2(tilesX);
  z = z < mz ? z : mz; y = y < my ? y : my; x = x < mx ? x : mx;
  while (z * y * x < want) {
    if (y * 2 <= my) y *= 2;
    else if (x * 2 <= mx) x *= 2;
    else if (z * 2 <= mz) z *= 2;
    else break;                                       // the grid has fewer tiles than slots: the launch is simply smaller
  }
  *bz = z; *by = y; *bx = x;
}

// Global -> LDS copy of `nitems` 16-byte items by `nthreads` threads with up to 8 loads in flight per thread.  A rolled
// `for (i = tid; i < n; i += nthreads) lds[i] = src[i]` waits for each load before it issues the next (the LDS store needs the value):
// a 64 KB operator staged by 256 threads took 16 memory latencies that way.  The caller publishes the copy with its own barrier.
__device__ __forceinline__ void stage_to_lds(unsigned char* lds, const u32x4* __restrict__ src, int nitems, int tid, int nthreads) {
  for (int i0 = tid; i0 < nitems; i0 += 8 * nthreads) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[min(i0 + u * nthreads, nitems - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * nthreads < nitems) *(u32x4*)(lds + (long long)(i0 + u * nthreads) * 16) = v[u];
  }
}

// ---- e4m3 activation planes ("NHWC16c" bytes: C / 16 planes of [D][H][W][16 B]) -- what the K = 128 fp8 convolution reads by LDS-DMA.
// 8 values of type T -> 8 OCP e4m3 bytes (round to nearest even, saturating at +-448): the conversion of the fp8 convs' loader waves
template <typename T>
__device__ __forceinline__ void e4m3_pack8(const u32x4 v, unsigned& lo, unsigned& hi) {
  const typename Vec8<T>::type in = __builtin_bit_cast(typename Vec8<T>::type, v);
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = __builtin_amdgcn_fmed3f(to_f32<T>(in[j]), -448.0f, 448.0f);
  int a = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
  a = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], a, true);
  int b = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
  b = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], b, true);
  lo = (unsigned)a; hi = (unsigned)b;
}
// the 8 channels of 16-bit plane `pl8` at voxel `vox` of a tensor with `nvox` voxels per plane -> their half of an e4m3 granule
// (y = the sample's base, bytes)
template <typename T>
__device__ __forceinline__ void e4m3_store8(unsigned char* y, int pl8, long long vox, long long nvox, const u32x4 v) {
  unsigned lo, hi;
  e4m3_pack8<T>(v, lo, hi);
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  *(u32x2_t*)(y + ((long long)(pl8 >> 1) * nvox + vox) * 16 + (pl8 & 1) * 8) = u32x2_t{lo, hi};
}

// The address of a __device__ object, taken ONCE per kernel: left to itself the compiler re-materialises a global's address at every use --
// s_getpc + s_load from the GOT + s_waitcnt lgkmcnt(0) in front of EVERY LDS-DMA piece of the loaders' loops (the zero page of the
// out-of-image halo) -- because a symbol address is "free" to recompute.  The empty asm makes the value opaque: it stays in two SGPRs.
template <typename T> __device__ __forceinline__ const T* iunet_opaque_ptr(const T* p) {
  asm volatile("" : "+s"(p));
  return p;
}
// ---- "m8" planes of the split-precision forward with its cross terms on the fp8 matrix cores (conv3_x2m.hip): beside its fp16 hi
// planes a tensor carries 2 C / 16 e4m3 planes [D][H][W][16 B], plane 2c = e4m3((v - hi) * 2^4), plane 2c + 1 = e4m3(hi * 2^-8) of the
// 16-channel chunk c.  8 fp32 values -> 8 e4m3 bytes (round to nearest even, saturating at +-448)
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2_t x2m_pack8(const float (&f)[8]) {
  float c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_fmed3f(f[j], -448.0f, 448.0f);
  int a = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
  a = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], a, true);
  int b = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], 0, false);
  b = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], b, true);
  return u32x2_t{(unsigned)a, (unsigned)b};
}
// this lane's 8 values r[j] (fp32, scaled by act_scale) -> hi words, lo words and its two half-granules of the m8 planes
// (lo8 = e4m3(res * 16) by v_cvt_scalef32_pk_fp8_f32 with scale 2^-4, one instruction per pair: bit for bit res * 16 -> clamp ->
// v_cvt_pk_fp8_f32 on every f32 pattern with |res| <= 28 -- tools/micro/cvt_scale_fp8_f32.hip walks all 2^32 -- and |res| <= 16 here, half
// an ulp of the largest hi word.  hi8 (a function of the hi words: x2m_hi8 below) is kept for the callers that still take it.)
__device__ __forceinline__ u32x2_t x2m_hi8(const f16x8 hi);
// (CLAMPED: the caller's values already lie in [-65504, 65504] -- the conv epilogues clamp and apply the ReLU in one v_med3)
template <bool CLAMPED = false>
__device__ __forceinline__ void x2m_split8(const float (&r)[8], f16x8& hi, f16x8& lo, u32x2_t& lo8, u32x2_t& hi8) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  float res[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = CLAMPED ? r[j] : fminf(fmaxf(r[j], -65504.f), 65504.f);
    const f16 h = (f16)v;
    res[j] = v - (float)h;                                     // exact in fp32
    hi[j] = h; lo[j] = (f16)res[j];
  }
  s16x2 a = s16x2{0, 0}, b = s16x2{0, 0};
  a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(a, res[0], res[1], 0.0625f, false);
  a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(a, res[2], res[3], 0.0625f, true);
  b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(b, res[4], res[5], 0.0625f, false);
  b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(b, res[6], res[7], 0.0625f, true);
  lo8 = u32x2_t{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
  hi8 = x2m_hi8(hi);
}
// one element of the caller's input tensor (generic strides): 0 f32, 1 f16, 2 u8 (/ 255: predict.py:30, correctly rounded as torch's), 3 bf16
__device__ __forceinline__ float x2_load_in(const void* p, long long off, int dt) {
  switch (dt) {
    case 0: return ((const float*)p)[off];
    case 1: return (float)((const f16*)p)[off];
    case 2: return __fdiv_rn((float)((const unsigned char*)p)[off], 255.0f);
    default: return (float)((const bf16*)p)[off];
  }
}
// ---- 2^d max-pool on the split words of the x2m form.  A candidate is the pair (hi word, lo8 byte) of one channel; the value a 3x3x3
// consumer reads is hi + lo8 / 16.  |lo8 / 16| never exceeds half an ulp of hi (x2m_split8: lo is the fp16 rounding residual, and e4m3
// rounding cannot carry it past the power of two that bounds it), so the order of the VALUES is the lexicographic order of (hi, lo8) --
// they differ only where two values tie at a rounding midpoint, and there the larger hi is as good a winner.  That order is one unsigned
// compare on a 24-bit key: [sortable hi word | sortable lo8 byte] (sign-magnitude -> offset binary: x ^ (sign ? all ones : sign bit)).
// A total order on the words themselves: the winner does not depend on the order the candidates meet in, so the pool kernel
// (conv3_x2m.hip: x2m_maxpool_kernel) and the conv epilogue that pools on the way out (x, y in registers, z through LDS) agree bit for bit.
__device__ __forceinline__ void x2m_pool_keys(const f16x8 hi, const u32x2_t lo8, unsigned (&k)[8]) {
  const u32x4 hb = __builtin_bit_cast(u32x4, hi);
  unsigned hf[4], lf[2];
#pragma unroll
  for (int d = 0; d < 4; ++d) hf[d] = hb[d] ^ (((hb[d] >> 15) & 0x00010001u) * 0x7fffu + 0x80008000u);
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const unsigned t = (lo8[e] >> 7) & 0x01010101u;
    lf[e] = lo8[e] ^ (((t << 7) - t) + 0x80808080u);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)           // bytes [lo8 | hi lo | hi hi | 0]
    k[j] = __builtin_amdgcn_perm(hf[j >> 1], lf[j >> 2], 0x0c000000u | ((5u + 2u * (j & 1)) << 16) | ((4u + 2u * (j & 1)) << 8) | (unsigned)(j & 3));
}
// the words of 8 winning keys: hi, lo8
__device__ __forceinline__ void x2m_pool_unkeys(const unsigned (&k)[8], f16x8& hi, u32x2_t& lo8) {
  u32x4 hb;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const unsigned f = __builtin_amdgcn_perm(k[2 * d + 1], k[2 * d], 0x06050201u);
    hb[d] = f ^ (0xffffffffu - ((f >> 15) & 0x00010001u) * 0x7fffu);
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const unsigned t0 = __builtin_amdgcn_perm(k[4 * e + 1], k[4 * e], 0x0c0c0400u), t1 = __builtin_amdgcn_perm(k[4 * e + 3], k[4 * e + 2], 0x0c0c0400u);
    const unsigned f = __builtin_amdgcn_perm(t1, t0, 0x05040100u);
    const unsigned t = (f >> 7) & 0x01010101u;
    lo8[e] = f ^ (0xffffffffu - ((t << 7) - t));
  }
  hi = __builtin_bit_cast(f16x8, hb);
}
// the value of lane ^ 1 (DPP quad_perm [1, 0, 3, 2]: no LDS traffic)
__device__ __forceinline__ unsigned lane_xor1(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
}
// range flag of the split-precision modes: a stored hi word of +-65504 means act_scale x activation saturated (split16's clamp).  Only a
// saturated word ever touches memory (atomicMax of its bit pattern 0x7bff into the caller's int): no traffic, no synchronisation.
__device__ __forceinline__ void x2_note_saturation(int* sat, const f16x8 hi) {
  const u32x4 hb = __builtin_bit_cast(u32x4, hi);
  unsigned m = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const unsigned a = hb[d] & 0x7fffu, b = (hb[d] >> 16) & 0x7fffu;
    m = m > a ? m : a; m = m > b ? m : b;
  }
  if (m >= 0x7bffu) atomicMax(sat, (int)m);
}
// byte offset, inside the lo8 planes of one sample ([C / 16][voxels][16 B]: the e4m3 words e4m3((v - hi) * 2^4) of a 16-channel chunk per
// voxel), of the half-granule that holds the 8 channels of 8-channel plane `pl8` at voxel `vox`.  (The other half of the fp8 step's
// operand, hi8 = e4m3(hi * 2^-8), is a function of the hi words: the conv loaders make it in LDS from the 16-bit halo image.)
__device__ __forceinline__ long long x2m_off(int pl8, long long vox, long long nvox) {
  return ((long long)(pl8 >> 1) * nvox + vox) * 16 + (pl8 & 1) * 8;
}
// hi8 of 8 hi words
// (v_cvt_scalef32_pk_fp8_f16 with scale 2^8: e4m3(h / 256) of a packed pair in ONE instruction -- bit for bit the three-instruction path
// (float)h * 2^-8 -> clamp -> v_cvt_pk_fp8_f32 of x2m_pack8 on every finite f16 pattern: tools/micro/cvt_scale_fp8_f16.hip.  The conv
// loaders make the hi8 half of every fp8 operand with this, once per step: 4 instructions per 8 channels instead of ~60.)
__device__ __forceinline__ u32x2_t x2m_hi8(const f16x8 hi) {
  typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  s16x2 a = s16x2{0, 0}, b = s16x2{0, 0};
  a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(a, h16x2{hi[0], hi[1]}, 256.0f, false);
  a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(a, h16x2{hi[2], hi[3]}, 256.0f, true);
  b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(b, h16x2{hi[4], hi[5]}, 256.0f, false);
  b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(b, h16x2{hi[6], hi[7]}, 256.0f, true);
  return u32x2_t{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
}

// ---- K = 128 operator order of the fp8 convolution (conv3_f8k.hip; written by pack_batch.hip kind 5 and conv3_f8.hip: pack_f8_kernel)
// filter column (dz * 3 + dx) of K = 128 group g, lane group q: {0, 3, 1, 4 | 2, 5, 6, 7} -- the lane groups 0 / 1 and 2 / 3 of one
// LDS pass differ in dz only; column 8 goes to the K = 32 instruction
__host__ __device__ constexpr int f8k_col(int g, int q) {
  return g == 0 ? (q == 0 ? 0 : q == 1 ? 3 : q == 2 ? 1 : 4) : (q == 0 ? 2 : q == 1 ? 5 : q == 2 ? 6 : 7);
}
constexpr int F8K_WSTEP = 2 * 3 * 2 * 2 * 1024 + 3 * 2 * 512;      // bytes of one (32 Cout, 32 Cin) block: 27 648
// Byte offset, inside that block, of the 8 input channels 16 e + 8 o .. + 7 of filter column col, row shift dy, Cout half m,
// operator row `row` (conv3_f8.hip's Cout map: co = 8 (row >> 2) + 4 m + (row & 3))
__host__ __device__ inline int f8k_offset(int col, int dy, int m, int e, int o, int row) {
  if (col == 8) return 2 * 3 * 2 * 2 * 1024 + ((dy * 2 + m) * 64 + (2 * e + o) * 16 + row) * 8;
  int g = 0, q = 0;
  for (int gg = 0; gg < 2; ++gg)
    for (int qq = 0; qq < 4; ++qq)
      if (f8k_col(gg, qq) == col) { g = gg; q = qq; }
  return ((((g * 3 + dy) * 2 + m) * 2 + e) * 64 + q * 16 + row) * 16 + 8 * o;
}
int iunet_f8_k128(int taps, int Cin);        // conv3_f8k.hip: 1 = the K = 128 order (3-D, Cin % 32 == 0), 0 = the K16 order
