// Oblique-slice extraction on the device (slicer.py:94-115 coordinates, :196-228 crop + map_coordinates).
// The plane point of output pixel (i, j) is origin + a * r_i + b * r_j (float64, every operation rounded on its
// own exactly as numpy evaluates `a[:, None, None] * r[None, :, None] + b[:, None, None] * r[None, None, :] + o`);
// it is sampled from the bounding-box crop [lo, lo + len) of the uint8 volume the way scipy.ndimage.map_coordinates
// (mode='constant', cval=0) does it for spline orders 0 and 1:
//   * a coordinate outside [0, len - 1] along any axis gives 0 (no interpolation against the padding);
//   * order 0: voxel floor(c + 0.5);
//   * order 1: weights w0 = 1 - frac, w1 = 1 - w0, neighbour indices mirrored at the crop edge (weight 0 there),
//     t = sum over (dz, dy, dx) of ((v * wz) * wy) * wx in that order, result trunc(t + 0.5) clamped to 255.
// HBM-bound gather: one thread per output pixel, 1 or 8 byte reads.
#include "common.h"

namespace {

struct SliceParams {
  const unsigned char* vol;
  int Z, Y, X;
  double a[3], b[3], o[3];
  int lo[3], len[3];
  int sw, start, order;
  unsigned char* out;
};

__device__ __forceinline__ int mirror_idx(int idx, int len) {
  if (len <= 1) return 0;
  const int s2 = 2 * len - 2;
  if (idx < 0) { idx = s2 * (-idx / s2) + idx; return idx <= 1 - len ? idx + s2 : -idx; }
  if (idx >= len) { idx -= s2 * (idx / s2); if (idx >= len) idx = s2 - idx; }
  return idx;
}

__global__ __launch_bounds__(256) void slice_gather_kernel(SliceParams p) {
#pragma clang fp contract(off)
  const int j = blockIdx.x * 16 + (threadIdx.x & 15), i = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (i >= p.sw || j >= p.sw) return;
  const double ri = (double)(p.start + i), rj = (double)(p.start + j);
  double c[3];
  bool inside = true;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double t1 = p.a[k] * ri, t2 = p.b[k] * rj;
    const double s = t1 + t2;
    const double g = s + p.o[k];
    c[k] = g - (double)p.lo[k];
    inside = inside && c[k] >= 0.0 && c[k] <= (double)(p.len[k] - 1);
  }
  unsigned char res = 0;
  if (inside) {
    const long long sy = p.X, sz = (long long)p.Y * p.X;
    const unsigned char* base = p.vol + (long long)p.lo[0] * sz + (long long)p.lo[1] * sy + p.lo[2];
    if (p.order == 0) {
      const int z = (int)floor(c[0] + 0.5), y = (int)floor(c[1] + 0.5), x = (int)floor(c[2] + 0.5);
      res = base[z * sz + y * sy + x];
    } else {
      int st[3];
      double w[3][2];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double f = floor(c[k]);
        st[k] = (int)f;
        const double x = c[k] - f;
        w[k][0] = 1.0 - x;
        w[k][1] = 1.0 - w[k][0];
      }
      double t = 0.0;
#pragma unroll
      for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const int iz = mirror_idx(st[0] + dz, p.len[0]), iy = mirror_idx(st[1] + dy, p.len[1]),
                      ix = mirror_idx(st[2] + dx, p.len[2]);
            double coeff = (double)base[iz * sz + iy * sy + ix];
            coeff = coeff * w[0][dz];
            coeff = coeff * w[1][dy];
            coeff = coeff * w[2][dx];
            t = t + coeff;
          }
      t = t > 0.0 ? t + 0.5 : 0.0;
      t = t > 255.0 ? 255.0 : t;
      res = (unsigned char)t;
    }
  }
  p.out[(long long)i * p.sw + j] = res;
}

// ---- update_volume (slicer.py:230-257): volume[round(coords), clipped] = data ------------------------------------------
// numpy's fancy-index assignment writes the pixels in row-major order, so where several pixels land on one voxel (clipping
// at the volume edge, or two neighbours of an oblique plane rounding to the same voxel) the LAST pixel wins.  To keep that
// deterministic on the device every voxel hit gets an owner first: an open-addressing table keyed by the voxel index holds
// the largest pixel index (atomicMax on (voxel + 1) << 21 | pixel); the second kernel lets only the owner write.
struct ScatterParams {
  unsigned char* vol;
  int Z, Y, X, C;
  double a[3], b[3], o[3];
  int sw, start;
  const unsigned char* data;
  unsigned long long* table;
  int log2t;
};

__device__ __forceinline__ long long scatter_voxel(const ScatterParams& p, int i, int j) {
#pragma clang fp contract(off)
  const double ri = (double)(p.start + i), rj = (double)(p.start + j);
  const int dims[3] = {p.Z, p.Y, p.X};
  long long idx[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double t1 = p.a[k] * ri, t2 = p.b[k] * rj;
    const double s = t1 + t2;
    const double g = s + p.o[k];
    const long long r = (long long)rint(g);                       // np.round: half to even
    idx[k] = r < 0 ? 0 : (r > dims[k] - 1 ? dims[k] - 1 : r);
  }
  return (idx[0] * p.Y + idx[1]) * p.X + idx[2];
}

__device__ __forceinline__ unsigned scatter_slot(unsigned long long v, int log2t) {
  return (unsigned)((v * 0x9E3779B97F4A7C15ull) >> (64 - log2t));
}

__global__ __launch_bounds__(256) void slice_scatter_claim_kernel(ScatterParams p) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= p.sw * p.sw) return;
  const unsigned long long v = (unsigned long long)scatter_voxel(p, pix / p.sw, pix % p.sw);
  const unsigned long long packed = ((v + 1) << 21) | (unsigned long long)pix;
  const unsigned mask = (1u << p.log2t) - 1;
  unsigned slot = scatter_slot(v, p.log2t);
  for (unsigned probe = 0; probe <= mask; ++probe) {          // (bounded: the table has twice as many slots as pixels)
    unsigned long long cur = p.table[slot];
    if (cur == 0ull) {
      cur = atomicCAS(&p.table[slot], 0ull, packed);
      if (cur == 0ull) return;
    }
    if ((cur >> 21) == v + 1) { atomicMax(&p.table[slot], packed); return; }
    slot = (slot + 1) & mask;
  }
}

__global__ __launch_bounds__(256) void slice_scatter_write_kernel(ScatterParams p) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= p.sw * p.sw) return;
  const unsigned long long v = (unsigned long long)scatter_voxel(p, pix / p.sw, pix % p.sw);
  const unsigned mask = (1u << p.log2t) - 1;
  unsigned slot = scatter_slot(v, p.log2t);
  unsigned probe = 0;
  while ((p.table[slot] >> 21) != v + 1 && probe++ <= mask) slot = (slot + 1) & mask;      // the claim kernel put this key in
  if ((p.table[slot] >> 21) != v + 1 || (int)(p.table[slot] & ((1ull << 21) - 1)) != pix) return;
  for (int c = 0; c < p.C; ++c) p.vol[(long long)v * p.C + c] = p.data[(long long)pix * p.C + c];
}

int scatter_log2t(int sw) {
  int l = 4;
  while ((1ll << l) < 2ll * sw * sw) ++l;
  return l;
}

}  // namespace

extern "C" {

long long iunet_slice_scatter_workspace_bytes(int sw) {
  if (sw < 1 || sw > 1024) return 0;
  return (long long)sizeof(unsigned long long) << scatter_log2t(sw);
}

// vol: uint8 [Z][Y][X][C] on the device (C = 1 for a plain volume); data: uint8 [sw][sw][C] on the device; geom as for
// iunet_slice_gather; workspace: iunet_slice_scatter_workspace_bytes(sw) bytes of device memory (contents ignored).
int iunet_slice_scatter(void* vol, int Z, int Y, int X, int C, const double* geom, int sw, int start, const void* data,
                        void* workspace, void* stream) {
  IUNET_REQUIRE(vol && geom && data && workspace, "slice_scatter: null pointer");
  IUNET_REQUIRE(sw >= 1 && sw <= 1024, "slice_scatter: slice width %d outside 1..1024", sw);
  IUNET_REQUIRE(Z > 0 && Y > 0 && X > 0 && C > 0 && (long long)Z * Y * X < (1ll << 42), "slice_scatter: volume %d x %d x %d x %d", Z, Y, X, C);
  ScatterParams p;
  p.vol = (unsigned char*)vol; p.Z = Z; p.Y = Y; p.X = X; p.C = C;
  for (int k = 0; k < 3; ++k) { p.a[k] = geom[k]; p.b[k] = geom[3 + k]; p.o[k] = geom[6 + k]; }
  p.sw = sw; p.start = start; p.data = (const unsigned char*)data;
  p.table = (unsigned long long*)workspace; p.log2t = scatter_log2t(sw);
  IUNET_CHECK_HIP(hipMemsetAsync(workspace, 0, (size_t)iunet_slice_scatter_workspace_bytes(sw), (hipStream_t)stream));
  const int nb = (sw * sw + 255) / 256;
  hipLaunchKernelGGL(slice_scatter_claim_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, p);
  hipLaunchKernelGGL(slice_scatter_write_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, p);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}


// vol: uint8 [Z][Y][X] on the device; geom: 9 host doubles (a[3], b[3], origin[3]); lo / len: the crop box in voxels
// (host ints, 0 <= lo, lo + len <= shape); out: uint8 [sw][sw] on the device.
int iunet_slice_gather(const void* vol, int Z, int Y, int X, const double* geom, const int* lo, const int* len, int sw,
                       int start, int order, void* out, void* stream) {
  IUNET_REQUIRE(vol && geom && lo && len && out, "slice_gather: null pointer");
  IUNET_REQUIRE(order == 0 || order == 1, "slice_gather: spline order must be 0 or 1 (got %d)", order);
  IUNET_REQUIRE(sw > 0, "slice_gather: bad slice width %d", sw);
  SliceParams p;
  p.vol = (const unsigned char*)vol; p.Z = Z; p.Y = Y; p.X = X;
  const int dims[3] = {Z, Y, X};
  for (int k = 0; k < 3; ++k) {
    p.a[k] = geom[k]; p.b[k] = geom[3 + k]; p.o[k] = geom[6 + k];
    p.lo[k] = lo[k]; p.len[k] = len[k];
    IUNET_REQUIRE(lo[k] >= 0 && len[k] >= 0 && lo[k] + len[k] <= dims[k], "slice_gather: crop box outside the volume (axis %d)", k);
  }
  p.sw = sw; p.start = start; p.order = order; p.out = (unsigned char*)out;
  if (len[0] == 0 || len[1] == 0 || len[2] == 0) {        // empty crop: everything is outside
    IUNET_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)sw * sw, (hipStream_t)stream));
    return IUNET_OK;
  }
  hipLaunchKernelGGL(slice_gather_kernel, dim3((sw + 15) / 16, (sw + 15) / 16), dim3(256), 0, (hipStream_t)stream, p);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

}  // extern "C"
