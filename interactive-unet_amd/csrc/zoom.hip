// Nearest-neighbour zoom of uint8 volumes: the multiscale pyramid of the reference's volumes
//   interactive_unet/utils.py:29-48  resize_volume   (per block: scipy.ndimage.zoom(block, scale, order=0))
//   interactive_unet/utils.py:50-77  add_multiscales (level i + 1 = resize_volume(level i, 0.5), predict.py:261)
// scipy.ndimage.zoom with order 0, mode 'constant', grid_mode False (scipy 1.15 ni_interpolation.c NI_ZoomShift; the
// dependency is not under /root/reference): per axis, output index o reads input index floor(o * z + 0.5) with
// z = (n_in - 1) / (n_out - 1) in double, n_out = round-half-even(n_in * zoom) -- and 0 (cval) where o * z > n_in - 1, which
// rounding makes true for the LAST sample of some sizes (28, 30, 32, 48, 56, ... at zoom 0.5).  That quirk is kept.
//
// HBM-bound byte gather: every axis has an index table (-1 = constant 0), four consecutive output bytes per lane, one
// dword store; a wave's loads walk consecutive cache lines of one input row.
#include "common.h"
#include <cmath>

namespace {

struct ZoomParams {
  const unsigned char* src;
  unsigned char* dst;
  long long s0, s1, s2, s3;        // source strides (bytes)
  long long q0, q1;                // destination strides of the two outer axes (bytes); the (d2, d3) plane is contiguous
  int d0, d1, d2, d3;              // destination extents
  const int* t0; const int* t1; const int* t2; const int* t3;
};

constexpr int ZOOM_ROWS = 8;       // destination rows (axis 1) per thread: the inner-axis table entries are loaded once for all of them

// block = tx x ty threads: tx lanes walk the contiguous (d2, d3) plane, four bytes each; ty groups x ZOOM_ROWS rows of axis 1
__global__ __launch_bounds__(256) void zoom_nearest_kernel(ZoomParams p, int tx_log2) {
  const int tx = threadIdx.x & ((1 << tx_log2) - 1), ty = threadIdx.x >> tx_log2;
  const int o0 = blockIdx.z;
  const int plane = p.d2 * p.d3;
  const int e0 = ((blockIdx.x << tx_log2) + tx) * 4;
  if (e0 >= plane) return;
  const int i0 = p.t0[o0];
  long long off[4];
  bool okc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = min(e0 + j, plane - 1);
    int o2, o3;
    if (p.d3 == 1) { o2 = e; o3 = 0; } else { o2 = e / p.d3; o3 = e - o2 * p.d3; }
    const int i2 = p.t2[o2], i3 = p.t3[o3];
    okc[j] = i0 >= 0 && i2 >= 0 && i3 >= 0;
    off[j] = okc[j] ? (long long)i0 * p.s0 + (long long)i2 * p.s2 + (long long)i3 * p.s3 : 0;
  }
  const int rows_per_block = (256 >> tx_log2) * ZOOM_ROWS;
  const int r0 = blockIdx.y * rows_per_block + ty;
#pragma unroll
  for (int r = 0; r < ZOOM_ROWS; ++r) {
    const int o1 = r0 + r * (256 >> tx_log2);
    if (o1 >= p.d1) break;
    const int i1 = p.t1[o1];
    const unsigned char* row = p.src + (long long)(i1 >= 0 ? i1 : 0) * p.s1;
    unsigned v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {          // (three aligned dword loads instead of these four byte loads measured the same)
      const unsigned char b = row[off[j]];
      v[j] = (okc[j] && i1 >= 0) ? b : 0u;
    }
    unsigned char* out = p.dst + (long long)o0 * p.q0 + (long long)o1 * p.q1 + e0;
    if ((((size_t)out) & 3) == 0 && e0 + 3 < plane) {
      *(unsigned*)out = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (e0 + j < plane) out[j] = (unsigned char)v[j];
    }
  }
}

// Python's round(): half to even (scipy computes the output shape with it)
long long round_half_even(double x) {
  const double f = std::floor(x), d = x - f;
  if (d > 0.5) return (long long)f + 1;
  if (d < 0.5) return (long long)f;
  return ((long long)f % 2 == 0) ? (long long)f : (long long)f + 1;
}

}  // namespace

extern "C" {

int iunet_zoom_nearest_len(int n_in, double zoom) {
  if (n_in < 1 || !(zoom > 0.0)) return 0;
  const long long n = round_half_even((double)n_in * zoom);
  return n < 0 ? 0 : (n > 2147483647LL ? 0 : (int)n);
}

int iunet_zoom_nearest_table(int n_in, double zoom, int* table, int n_out) {
  IUNET_REQUIRE(n_in >= 1 && zoom > 0.0, "zoom_nearest_table: n_in %d, zoom %g", n_in, zoom);
  IUNET_REQUIRE(table != nullptr && n_out == iunet_zoom_nearest_len(n_in, zoom), "zoom_nearest_table: n_out %d != round(%d * %g)", n_out,
                n_in, zoom);
  const double z = n_out - 1 > 0 ? (double)(n_in - 1) / (double)(n_out - 1) : 1.0;
  for (int o = 0; o < n_out; ++o) {
    const double cc = (double)o * z;
    table[o] = (cc < 0.0 || cc > (double)(n_in - 1)) ? -1 : (int)std::floor(cc + 0.5);
  }
  return IUNET_OK;
}

int iunet_zoom_nearest_u8(const void* src, const int* src_dims, const long long* src_strides, void* dst,
                          const long long* dst_strides, const int* dst_dims, const int* tables, void* stream) {
  IUNET_REQUIRE(src != nullptr && dst != nullptr && src_dims != nullptr && src_strides != nullptr && dst_strides != nullptr && dst_dims != nullptr &&
                tables != nullptr,
                "zoom_nearest_u8: null argument");
  ZoomParams p;
  p.src = (const unsigned char*)src; p.dst = (unsigned char*)dst;
  p.s0 = src_strides[0]; p.s1 = src_strides[1]; p.s2 = src_strides[2]; p.s3 = src_strides[3];
  p.d0 = dst_dims[0]; p.d1 = dst_dims[1]; p.d2 = dst_dims[2]; p.d3 = dst_dims[3];
  p.q0 = dst_strides[0]; p.q1 = dst_strides[1];
  IUNET_REQUIRE(src_dims[0] > 0 && src_dims[1] > 0 && src_dims[2] > 0 && src_dims[3] > 0, "zoom_nearest_u8: source %d x %d x %d x %d",
                src_dims[0], src_dims[1], src_dims[2], src_dims[3]);
  if (p.d0 == 0 || p.d1 == 0 || p.d2 == 0 || p.d3 == 0) return IUNET_OK;       // an empty level (the reference's 4-D quirk)
  IUNET_REQUIRE(p.d0 > 0 && p.d1 > 0 && p.d2 > 0 && p.d3 > 0 && p.d0 <= 65535 && p.d1 <= 65535 &&
                (long long)p.d2 * p.d3 <= 2147483647LL, "zoom_nearest_u8: destination %d x %d x %d x %d", p.d0, p.d1, p.d2, p.d3);
  p.t0 = tables; p.t1 = p.t0 + p.d0; p.t2 = p.t1 + p.d1; p.t3 = p.t2 + p.d2;
  IUNET_REQUIRE(dst_strides[3] == 1 && dst_strides[2] == p.d3, "zoom_nearest_u8: the two inner destination axes must be contiguous "
                "(strides %lld, %lld for extents %d, %d)", dst_strides[2], dst_strides[3], p.d2, p.d3);
  const int plane = p.d2 * p.d3;
  int tx_log2 = 0;
  while (tx_log2 < 8 && (4 << tx_log2) < plane) ++tx_log2;                 // tx = min(256, pow2 >= plane / 4)
  const int tx = 1 << tx_log2, rows_per_block = (256 / tx) * ZOOM_ROWS;
  const long long gy = (p.d1 + rows_per_block - 1) / rows_per_block;
  dim3 grid((unsigned)((plane + 4 * tx - 1) / (4 * tx)), (unsigned)gy, (unsigned)p.d0);
  hipLaunchKernelGGL(zoom_nearest_kernel, grid, dim3(256), 0, (hipStream_t)stream, p, tx_log2);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

}  // extern "C"
