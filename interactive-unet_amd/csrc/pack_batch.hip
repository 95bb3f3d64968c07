// One launch packs every layer of the network (after each optimiser step, and when BatchNorm is folded for
// inference): the per-layer pack kernels are a few microseconds of work each behind ~5 us of launch latency,
// ~150 launches per step.  A descriptor table in device memory (built once by the host: all pointers are
// stable) drives one kernel; blockIdx.y selects the layer.  The element mappings are the ones of the per-layer
// kernels (conv3_mfma.hip, pointwise.hip, train_misc.hip) -- tests/test_gpu_kernels.py checks bit equality.
#include "common.h"
#include "pack_desc.h"

namespace {


// round |x| <= 448 to the nearest OCP e4m3 value (4 exponent bits, bias 7, 3 mantissa bits, subnormal step 2^-9), ties to even
__device__ __forceinline__ float round_e4m3(float x) {
  const float a = fabsf(x);
  int e;
  (void)frexpf(a, &e);                                     // a = m 2^e, m in [0.5, 1)
  const int fl = (a == 0.f || e - 1 < -6) ? -6 : e - 1;     // exponent of the binade (subnormals share -6)
  const float step = ldexpf(1.0f, fl - 3);
  return copysignf(rintf(a / step) * step, x);
}

// per-output-channel scale 2^k with max |w| / 2^k <= 448 (k minimal)
__device__ __forceinline__ float e4m3_scale(float amax) {
  if (!(amax > 0.f)) return 1.0f;
  int e;
  const float m = frexpf(amax / 448.0f, &e);
  return ldexpf(1.0f, m == 0.5f ? e - 1 : e);
}

// byte of an e4m3 value (conv3_f8.hip: f8_encode_e4m3): sign | 4 exponent bits (bias 7) | 3 mantissa bits
__device__ __forceinline__ unsigned char encode_e4m3(float v) {
  const float a = fabsf(v);
  const unsigned char s = v < 0.f || (v == 0.f && __builtin_signbit(v)) ? 0x80 : 0;
  if (a == 0.f) return s;
  int e;
  const float m = frexpf(a, &e);                                  // a = m 2^e, m in [0.5, 1)
  if (e - 1 < -6) return s | (unsigned char)(int)ldexpf(a, 9);    // subnormal: a / 2^-9
  return s | (unsigned char)(((e - 1 + 7) << 3) | ((int)ldexpf(m, 4) - 8));
}

// The fold is the IEEE fp32 formula of the host / oracle (oracle/unet_ref.py: fold_bn): every operation rounded
// on its own (no fma contraction; hipcc's default sqrt and divide are correctly rounded).
__device__ __forceinline__ float fold_scale(const PackDesc& d, int co) {
#pragma clang fp contract(off)
  if (!d.gamma) return 1.0f;
  const float s = d.var[co] + d.eps;
  return d.gamma[co] / sqrtf(s);
}
__device__ __forceinline__ float fold_bias(const PackDesc& d, int co) {
#pragma clang fp contract(off)
  const float t = d.mean[co] * fold_scale(d, co);
  return d.beta[co] - t;
}
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}

__device__ __forceinline__ float elem(const PackDesc& d, long long i, int& oc) {
  oc = 0;
  int r = (int)i;                                      // a layer has < 2^31 elements: 32-bit divisions
  const int j = r & 7; r >>= 3;
  const int lane = r & 63; r >>= 6;
  const int row = lane & 15, qq = lane >> 4;
  const float* w = d.w;
  if (d.kind == 0) {                                   // [cob][chunk32][tap][MI][64][8]
    const int CoutP = d.dgrad ? d.Cin : d.Cout, CinP = d.dgrad ? d.Cout : d.Cin;
    const int MI = (CoutP % 64 == 0) ? 4 : 2, nchunk = CinP >> 5;
    const int m = r % MI; r /= MI;
    const int tap = r % d.taps; r /= d.taps;
    const int chunk = r % nchunk, cob = r / nchunk;
    const int co = cob * 16 * MI + 32 * (m >> 1) + 8 * (row >> 2) + 4 * (m & 1) + (row & 3);
    const int ci = chunk * 32 + 8 * qq + j;
    oc = co;
    if (!d.dgrad) return mul_rn(w[((long long)co * d.Cin + ci) * d.taps + tap], fold_scale(d, co));
    return w[((long long)ci * d.Cin + co) * d.taps + (d.taps - 1 - tap)];
  }
  if (d.kind == 1) {                                   // [cob32][chunk16][column pair][dy][2][64][8]
    const int CinP = d.dgrad ? d.Cout : d.Cin;
    const int ncol = d.taps / 3, ncmb = (ncol + 1) / 2, nchunk = CinP >> 4;
    const int m = r & 1; r >>= 1;
    const int dy = r % 3; r /= 3;
    const int c = r % ncmb; r /= ncmb;
    const int chunk = r % nchunk, cob = r / nchunk;
    const int co = cob * 32 + 8 * (row >> 2) + 4 * m + (row & 3);
    const int ci = chunk * 16 + 8 * (qq & 1) + j;
    const int col = 2 * c + (qq >> 1);
    if (col >= ncol) return 0.f;
    const int tap = ((col / 3) * 3 + dy) * 3 + (col % 3);
    oc = co;
    if (!d.dgrad) return mul_rn(w[((long long)co * d.Cin + ci) * d.taps + tap], fold_scale(d, co));
    return w[((long long)ci * d.Cin + co) * d.taps + (d.taps - 1 - tap)];
  }
  if (d.kind == 2) {                                   // first conv: [cob32][kstep][2][64][8], k = tap * Cin + c
    const int KK = d.taps * d.Cin, KS = (KK + 31) / 32;
    const int t = r & 1; r >>= 1;
    const int ks = r % KS, cob = r / KS;
    const int co = cob * 32 + 8 * (row >> 2) + 4 * t + (row & 3);
    const int k = 32 * ks + 8 * qq + j;
    if (k >= KK) return 0.f;
    oc = co;
    return mul_rn(w[(co * d.Cin + k % d.Cin) * d.taps + k / d.Cin], fold_scale(d, co));
  }
  const int npos = d.taps;
  if (d.kind == 3) {                                   // convT fwd: [cob32][kstep][pos][t][64][8]
    const int nk = d.Cin >> 5;
    const int t = r & 1; r >>= 1;
    const int s = r % npos; r /= npos;
    const int ks = r % nk, cob = r / nk;
    const int co = cob * 32 + 8 * (row >> 2) + 4 * t + (row & 3);
    const int ci = ks * 32 + 8 * qq + j;
    oc = co;
    return w[((long long)ci * d.Cout + co) * npos + s];
  }
  {                                                    // convT dgrad: [cib32][pos][kc][t][64][8]
    const int nk = d.Cout >> 5;
    const int t = r & 1; r >>= 1;
    const int kc = r % nk; r /= nk;
    const int s = r % npos, cib = r / npos;
    const int ci = cib * 32 + 8 * (row >> 2) + 4 * t + (row & 3);
    const int co = kc * 32 + 8 * qq + j;
    return w[((long long)ci * d.Cout + co) * npos + s];
  }
}

// The 8 elements of one 16-byte granule (j = 0..7) differ in the input channel only: one index decode per granule, then 8 loads
// at a fixed stride.  Returns false for a zero granule (the padded partner column of the K16 order).  kind 2 (first conv: k runs
// over taps and channels) is decoded per element by elem().
__device__ __forceinline__ bool granule(const PackDesc& d, int r, const float*& src, long long& stride, float& fs, int& oc) {
  oc = 0; fs = 1.0f;
  const int lane = r & 63; r >>= 6;
  const int row = lane & 15, qq = lane >> 4;
  if (d.kind == 0) {
    const int CoutP = d.dgrad ? d.Cin : d.Cout, CinP = d.dgrad ? d.Cout : d.Cin;
    const int MI = (CoutP % 64 == 0) ? 4 : 2, nchunk = CinP >> 5;
    const int m = r % MI; r /= MI;
    const int tap = r % d.taps; r /= d.taps;
    const int chunk = r % nchunk, cob = r / nchunk;
    const int co = cob * 16 * MI + 32 * (m >> 1) + 8 * (row >> 2) + 4 * (m & 1) + (row & 3);
    const int ci = chunk * 32 + 8 * qq;
    oc = co;
    if (!d.dgrad) { src = d.w + ((long long)co * d.Cin + ci) * d.taps + tap; stride = d.taps; fs = fold_scale(d, co); }
    else { src = d.w + ((long long)ci * d.Cin + co) * d.taps + (d.taps - 1 - tap); stride = (long long)d.Cin * d.taps; }
    return true;
  }
  if (d.kind == 1) {
    const int CinP = d.dgrad ? d.Cout : d.Cin;
    const int ncol = d.taps / 3, ncmb = (ncol + 1) / 2, nchunk = CinP >> 4;
    const int m = r & 1; r >>= 1;
    const int dy = r % 3; r /= 3;
    const int c = r % ncmb; r /= ncmb;
    const int chunk = r % nchunk, cob = r / nchunk;
    const int co = cob * 32 + 8 * (row >> 2) + 4 * m + (row & 3);
    const int ci = chunk * 16 + 8 * (qq & 1);
    const int col = 2 * c + (qq >> 1);
    if (col >= ncol) return false;
    const int tap = ((col / 3) * 3 + dy) * 3 + (col % 3);
    oc = co;
    if (!d.dgrad) { src = d.w + ((long long)co * d.Cin + ci) * d.taps + tap; stride = d.taps; fs = fold_scale(d, co); }
    else { src = d.w + ((long long)ci * d.Cin + co) * d.taps + (d.taps - 1 - tap); stride = (long long)d.Cin * d.taps; }
    return true;
  }
  const int npos = d.taps;
  if (d.kind == 3) {
    const int nk = d.Cin >> 5;
    const int t = r & 1; r >>= 1;
    const int s = r % npos; r /= npos;
    const int ks = r % nk, cob = r / nk;
    const int co = cob * 32 + 8 * (row >> 2) + 4 * t + (row & 3);
    const int ci = ks * 32 + 8 * qq;
    oc = co;
    src = d.w + ((long long)ci * d.Cout + co) * npos + s; stride = (long long)d.Cout * npos;
    return true;
  }
  {
    const int nk = d.Cout >> 5;
    const int t = r & 1; r >>= 1;
    const int kc = r % nk; r /= nk;
    const int s = r % npos, cib = r / npos;
    const int ci = cib * 32 + 8 * (row >> 2) + 4 * t + (row & 3);
    const int co = kc * 32 + 8 * qq;
    src = d.w + ((long long)ci * d.Cout + co) * npos + s; stride = npos;
    return true;
  }
}

// K16 operators (every 3^d conv of the network: all but a few thousand of the elements a step packs) go through LDS: one
// workgroup per (32-output-channel tile, 16-channel chunk) block reads the block's source rows as whole 16-byte runs -- 32 rows of
// 16 x taps floats (forward: w[co][chunk]), 16 rows of 32 x taps floats (data gradient: w[chunk][co tile]) -- and writes the
// block's 30 KB (3-D) of packed operator from there.  The direct gather of pack_batch_kernel's other kinds touches a 128-byte line
// for 4 useful bytes at a time: on C5's 90 M parameters that was ~1 ms per step.
__device__ __forceinline__ void pack_k16_block(const PackDesc& d, int blk, float* tile, bool k128) {
  const int CoutP = d.dgrad ? d.Cin : d.Cout, CinP = d.dgrad ? d.Cout : d.Cin;
  const int taps = d.taps, ncol = taps / 3, ncmb = (ncol + 1) / 2, nchunk = CinP >> 4;
  const int chunk = blk % nchunk, cob = blk / nchunk;
  (void)CoutP;
  // ---- source rows -> LDS ----
  const int nrows = d.dgrad ? 16 : 32, rowlen = (d.dgrad ? 32 : 16) * taps;       // floats; rowlen % 4 == 0
  const int rl4 = rowlen >> 2;
  auto row_base = [&](int row) -> long long {
    return d.dgrad ? ((long long)(chunk * 16 + row) * d.Cin + cob * 32) * taps : ((long long)(cob * 32 + row) * d.Cin + chunk * 16) * taps;
  };
  if (((size_t)d.w & 15) == 0) {                       // every row starts on a 16-byte boundary when the tensor does
    // seven loads in flight per thread (two rounds per 3-D block) (a rolled load -> LDS-store loop waited for every one of its 14 loads in turn: 27 us per block,
    // 700 us per step on C5's 13 000 blocks)
    for (int i0 = threadIdx.x; i0 < nrows * rl4; i0 += 7 * 256) {
      f32x4 v[7];
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        const int i = min(i0 + u * 256, nrows * rl4 - 1);
        const int row = i / rl4, o4 = i - row * rl4;
        v[u] = *(const f32x4*)(d.w + row_base(row) + o4 * 4);
      }
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        const int i = i0 + u * 256;
        if (i < nrows * rl4) {
          const int row = i / rl4, o4 = i - row * rl4;
          *(f32x4*)(tile + row * rowlen + o4 * 4) = v[u];
        }
      }
    }
  } else {                                             // a parameter at an odd offset of the flat vector
    for (int i = threadIdx.x; i < nrows * rowlen; i += 256) {
      const int row = i / rowlen, o = i - row * rowlen;
      tile[row * rowlen + o] = d.w[row_base(row) + o];
    }
  }
  __syncthreads();
  // ---- granules of this block: [column pair][dy][m][lane] x 8 channels ----
  const int ngran = ncmb * 3 * 2 * 64;
  const long long out0 = (long long)blk * ngran;                                 // granule index of the block's first granule
  for (int r0 = threadIdx.x; r0 < ngran; r0 += 256) {
    int r = r0;
    const int lane = r & 63; r >>= 6;
    const int row = lane & 15, qq = lane >> 4;
    const int m = r & 1; r >>= 1;
    const int dy = r % 3, c = r / 3;
    const int col = 2 * c + (qq >> 1);
    const int co_l = 8 * (row >> 2) + 4 * m + (row & 3), ci_l = 8 * (qq & 1);
    const int oc = cob * 32 + co_l;
    float v[8];
    if (col < ncol) {
      const int tap = ((col / 3) * 3 + dy) * 3 + (col % 3);
      if (!d.dgrad) {
        const float fs = fold_scale(d, oc);
        const float* src = tile + co_l * rowlen + ci_l * taps + tap;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = mul_rn(src[j * taps], fs);
      } else {
        const float* src = tile + ci_l * rowlen + co_l * taps + (taps - 1 - tap);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[j * rowlen];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    if (d.kind == 5) {                                 // e4m3 bytes of w * fold / scale (the arithmetic of conv3_f8.hip: pack_f8_kernel)
      const float sc = d.qscale[oc];
      unsigned long long pk = 0;
      if (col < ncol) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pk |= (unsigned long long)encode_e4m3(round_e4m3(v[j] / sc)) << (8 * j);
      }
      if (!k128) *(unsigned long long*)((unsigned char*)d.dst + (out0 + r0) * 8) = pk;
      else if (col < ncol)                           // conv3_f8k.hip's order: blocks of 32 input channels, no padded column
        *(unsigned long long*)((unsigned char*)d.dst + ((long long)cob * (nchunk >> 1) + (chunk >> 1)) * F8K_WSTEP +
                               f8k_offset(col, dy, m, chunk & 1, qq & 1, row)) = pk;
      continue;
    }
    if (d.qscale) {
      const float sc = d.qscale[col < ncol ? oc : 0];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = sc * round_e4m3(v[j] / sc);
    }
    if (d.dtype == 0) {
      typename Vec8<f16>::type o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = from_f32<f16>(v[j]);
      *(typename Vec8<f16>::type*)((f16*)d.dst + (out0 + r0) * 8) = o;
    } else {
      typename Vec8<bf16>::type o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = from_f32<bf16>(v[j]);
      *(typename Vec8<bf16>::type*)((bf16*)d.dst + (out0 + r0) * 8) = o;
    }
  }
  __syncthreads();                                     // the tile is reused by this workgroup's next block
}

// kind 6: the compact K16 order (per Cout tile and chunk PAIR: [even chunk 24 KB | odd chunk 24 KB | cross pair 6 KB]; 3^2 filters:
// [6 KB | 6 KB | 6 KB] -- one regular column pair per chunk and the third column in the cross pair).  One
// workgroup per (Cout tile, 16-channel chunk) as above: the chunk's regular column pairs, and its half of the cross fragments
// (the lanes q >> 1 = chunk parity: the last column of this chunk).
__device__ __forceinline__ void pack_k16c_block(const PackDesc& d, int blk, float* tile) {
  const int CinP = d.dgrad ? d.Cout : d.Cin;
  const int taps = d.taps, ncol = taps / 3;
  const int NREG = (ncol / 2) * 384;                   // granules of a chunk's regular pairs: 4 x (3 dy x 2 m x 64 lanes) / 1 x
  const int nchunk = CinP >> 4, chunk = blk % nchunk, cob = blk / nchunk;
  const int nrows = d.dgrad ? 16 : 32, rowlen = (d.dgrad ? 32 : 16) * taps, rl4 = rowlen >> 2;
  auto row_base = [&](int row) -> long long {
    return d.dgrad ? ((long long)(chunk * 16 + row) * d.Cin + cob * 32) * taps : ((long long)(cob * 32 + row) * d.Cin + chunk * 16) * taps;
  };
  if (((size_t)d.w & 15) == 0) {
    for (int i0 = threadIdx.x; i0 < nrows * rl4; i0 += 7 * 256) {      // seven loads in flight per thread (two rounds per 3-D block), as in pack_k16_block
      f32x4 v[7];
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        const int i = min(i0 + u * 256, nrows * rl4 - 1);
        const int row = i / rl4, o4 = i - row * rl4;
        v[u] = *(const f32x4*)(d.w + row_base(row) + o4 * 4);
      }
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        const int i = i0 + u * 256;
        if (i < nrows * rl4) {
          const int row = i / rl4, o4 = i - row * rl4;
          *(f32x4*)(tile + row * rowlen + o4 * 4) = v[u];
        }
      }
    }
  } else {
    for (int i = threadIdx.x; i < nrows * rowlen; i += 256) {
      const int row = i / rowlen, o = i - row * rowlen;
      tile[row * rowlen + o] = d.w[row_base(row) + o];
    }
  }
  __syncthreads();
  const int par = chunk & 1;
  const long long pair_base = ((long long)cob * (nchunk >> 1) + (chunk >> 1)) * (2 * NREG + 384);      // granules (16 B)
  for (int t = threadIdx.x; t < NREG + 192; t += 256) {
    int frag, lane, col;
    long long og;
    if (t < NREG) { frag = t >> 6; lane = t & 63; col = 2 * (frag / 6) + (lane >> 5); frag %= 6; og = pair_base + par * NREG + t; }
    else {
      const int u = t - NREG, l32 = u & 31;
      frag = u >> 5; lane = (par * 2 + (l32 >> 4)) * 16 + (l32 & 15); col = ncol - 1; og = pair_base + 2 * NREG + frag * 64 + lane;
    }
    const int dy = frag >> 1, m = frag & 1;
    const int row = lane & 15, qq = lane >> 4;
    const int co_l = 8 * (row >> 2) + 4 * m + (row & 3), ci_l = 8 * (qq & 1);
    const int oc = cob * 32 + co_l;
    const int tap = ((col / 3) * 3 + dy) * 3 + (col % 3);
    float v[8];
    if (!d.dgrad) {
      const float fs = fold_scale(d, oc);
      const float* src = tile + co_l * rowlen + ci_l * taps + tap;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = mul_rn(src[j * taps], fs);
    } else {
      const float* src = tile + ci_l * rowlen + co_l * taps + (taps - 1 - tap);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[j * rowlen];
    }
    if (d.qscale) {
      const float sc = d.qscale[oc];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = sc * round_e4m3(v[j] / sc);
    }
    if (d.dtype == 0) {
      typename Vec8<f16>::type o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = from_f32<f16>(v[j]);
      *(typename Vec8<f16>::type*)((f16*)d.dst + og * 8) = o;
    } else {
      typename Vec8<bf16>::type o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = from_f32<bf16>(v[j]);
      *(typename Vec8<bf16>::type*)((bf16*)d.dst + og * 8) = o;
    }
  }
  __syncthreads();
}

// Tasks of one layer: a K16 block each for the 3^d convs, 1 024 granules (of 8 elements) each for the other kinds.
__device__ __forceinline__ int pack_tasks(const PackDesc& d) {
  if (d.kind == 1 || d.kind == 5 || d.kind == 6) return ((d.dgrad ? d.Cin : d.Cout) >> 5) * ((d.dgrad ? d.Cout : d.Cin) >> 4);
  const long long gran = (d.total + 7) / 8;
  return (int)((gran + 1023) / 1024);
}
constexpr int PACK_MAX_LAYERS = 512;

// The layers' tasks form ONE list walked grid-stride by the launch's workgroups (each rebuilds the prefix of the task counts in
// LDS: n <= 512 descriptors).  The first form launched 1 024 workgroups per layer and let the surplus exit: 40 000 workgroups with
// 54 KB of LDS each for the 800 blocks of the C3 net -- 70 us per step of workgroup dispatch (573 us on C5's 13 000 blocks).
__global__ __launch_bounds__(256) void pack_batch_kernel(const PackDesc* __restrict__ descs, int n, int f8_k128) {
  __shared__ __attribute__((aligned(16))) float tile[32 * 16 * 27];               // one K16 block of source weights (54 KB)
  __shared__ int first[PACK_MAX_LAYERS + 1];
  for (int i = threadIdx.x; i < n; i += 256) first[i + 1] = pack_tasks(descs[i]);
  if (threadIdx.x == 0) first[0] = 0;
  __syncthreads();
  if (threadIdx.x < 64) {                     // inclusive scan by one wave, 64 entries at a time
    int carry = 0;
    for (int base = 0; base < n; base += 64) {
      const int i = base + threadIdx.x;
      int v = i < n ? first[i + 1] : 0;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if ((int)threadIdx.x >= o) v += t; }
      if (i < n) first[i + 1] = v + carry;
      carry += __shfl(v, 63);
    }
  }
  __syncthreads();
  const int ntask = first[n];
  for (int task = blockIdx.x; task < ntask; task += gridDim.x) {
    int lo = 0, hi = n - 1;                   // the layer of this task: last i with first[i] <= task (uniform over the workgroup)
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (first[mid] <= task) lo = mid; else hi = mid - 1; }
    lo = __builtin_amdgcn_readfirstlane(lo);            // uniform by construction: the descriptor then loads into scalar registers
    const PackDesc d = descs[lo];                        // (left in vector registers, every index computation below ran on the vector ALU)
    const int local = __builtin_amdgcn_readfirstlane(task - first[lo]);
    if (d.bias_out && local == 0) {
      for (int co = threadIdx.x; co < d.Cout; co += 256) d.bias_out[co] = fold_bias(d, co);      // separately rounded, as the host formula beta - mean * scale
    }
    if (d.kind == 1 || d.kind == 5 || d.kind == 6) {
      const bool k128 = f8_k128 && d.kind == 5 && d.taps == 27 && d.Cin % 32 == 0;      // iunet_f8_k128(taps, Cin)
      if (d.kind == 6) pack_k16c_block(d, local, tile); else pack_k16_block(d, local, tile, k128);
      __syncthreads();                        // the tile is read to the end before the next task overwrites it
      continue;
    }
    const bool fold = d.gamma != nullptr && !d.dgrad && d.kind <= 1;     // the kinds whose elem() multiplies by the folded scale
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
      const long long g = (long long)local * 1024 + k * 256 + threadIdx.x;
      if (g * 8 >= d.total) break;
      float v[8];
      int oc = 0;
      if (d.kind == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = elem(d, g * 8 + j, oc);
      } else {
        const float* src; long long stride; float fs;
        if (granule(d, (int)g, src, stride, fs, oc)) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = src[j * stride];
          if (fold) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = mul_rn(v[j], fs);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
      }
      if (d.qscale) {
        const float sc = d.qscale[oc];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = sc * round_e4m3(v[j] / sc);
      }
      if (d.dtype == 0) {
        typename Vec8<f16>::type o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = from_f32<f16>(v[j]);
        *(typename Vec8<f16>::type*)((f16*)d.dst + g * 8) = o;
      } else {
        typename Vec8<bf16>::type o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = from_f32<bf16>(v[j]);
        *(typename Vec8<bf16>::type*)((bf16*)d.dst + g * 8) = o;
      }
    }
  }
}

// one block per (layer, output channel): scale of the e4m3 quantisation from max |folded weight|
__global__ __launch_bounds__(256) void pack_qscale_kernel(const PackDesc* __restrict__ descs) {
  const PackDesc d = descs[blockIdx.y];
  const int co = blockIdx.x;
  if (!d.qscale || d.dgrad || d.kind == 4 || co >= d.Cout) return;
  __shared__ float red[256];
  float m = 0.f;
  if (d.kind == 3) {                                     // convT: w[ci][co][pos]
    for (int i = threadIdx.x; i < d.Cin * d.taps; i += 256)
      m = fmaxf(m, fabsf(d.w[((long long)(i / d.taps) * d.Cout + co) * d.taps + i % d.taps]));
  } else {                                               // conv: w[co][ci][tap], folded
    const float fs = fold_scale(d, co);
    const float* w = d.w + (long long)co * d.Cin * d.taps;
    for (int i = threadIdx.x; i < d.Cin * d.taps; i += 256) m = fmaxf(m, fabsf(mul_rn(w[i], fs)));
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
  if (threadIdx.x == 0) d.qscale[co] = e4m3_scale(red[0]);
}

}  // namespace

extern "C" {

int iunet_pack_desc_bytes(void) { return (int)sizeof(PackDesc); }

// descs: device array of `n` descriptors (layout: see PackDesc above / _native.PackDesc)
// quant_max_cout > 0: some descriptors carry a qscale buffer (e4m3 weight quantisation); the scales are computed first
int iunet_pack_batch(const void* descs, int n, int quant_max_cout, void* stream) {
  IUNET_REQUIRE(descs && n > 0, "pack_batch: empty descriptor table");
  if (quant_max_cout > 0)
    hipLaunchKernelGGL(pack_qscale_kernel, dim3(quant_max_cout, n), dim3(256), 0, (hipStream_t)stream, (const PackDesc*)descs);
  IUNET_REQUIRE(n <= PACK_MAX_LAYERS, "pack_batch: %d descriptors (at most %d per launch)", n, PACK_MAX_LAYERS);
  // one task list over all layers, 2 048 workgroups walking it (two resident per CU by their LDS, four rounds of them: the largest
  // operators -- 28 M elements in C5 -- need that many to hide their gather latency; a small net's surplus exits after the prefix)
  hipLaunchKernelGGL(pack_batch_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const PackDesc*)descs, n,
                     iunet_f8_k128(27, 32));
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

}  // extern "C"
