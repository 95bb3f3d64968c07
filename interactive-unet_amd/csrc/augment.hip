// Batch producer on the device: the reference's loader contract as ONE gather launch per batch.
//   interactive_unet/loader.py:32-42    image / mask / weight uint8 -> value / 255 (float64 -> float32), weight repeated over the
//                                       classes, mask and weight zeroed where image[0] == 0
//   interactive_unet/loader.py:125-133  RandomHorizontalFlip, RandomVerticalFlip, RandomRotation(NEAREST), RandomResizedCrop(NEAREST)
//   interactive_unet/loader.py:138-154  the same parameters for the three tensors; float16 out, channels first
// All four transforms are nearest-neighbour index maps, so their composition is one source pixel (or none: the rotation's
// zero padding) per output pixel.  The index arithmetic is torchvision's (not under /root/reference; restated in
// oracle/loader_ref.py, which checks it against torch's own grid_sample / interpolate): crop + interpolate(mode='nearest')
// -> pixel of the rotated image: min(floor(o * (h / out)), h - 1) in float32; rotation: the affine grid g = fma(y, r1, x * r0)
// + r2 (the order torch's CPU bmm accumulates in), ((g + 1) * size - 1) / 2, rint, zero outside; flips last.
// HBM-bound: per output pixel ch + C + 1 source bytes read, (ch + 2 C) halves written.
#include "common.h"

namespace {

struct AugDesc {                 // one sample of the batch (mirrored by interactive_unet/loader.py: AugDesc)
  const unsigned char* image;    // uint8 [H][W][ch]
  const unsigned char* mask;     // uint8 [H][W][C]
  const unsigned char* weight;   // uint8 [H][W]
  const float* xg;               // torch.linspace((1 - W) / 2, (W - 1) / 2, W)
  const float* yg;               // torch.linspace((1 - H) / 2, (H - 1) / 2, H)
  int H, W;
  int hflip, vflip;
  int kind;                      // 0 affine grid, 1 identity, 2 rot90 k=2, 3 rot90 k=1, 4 rot90 k=3 (torchvision rotate fast paths)
  int ci, cj, ch, cw;            // crop of the rotated image: top, left, height, width
  float r[6];                    // theta^T / (W/2, H/2): r00 r10 r20 (x) r01 r11 r21 (y)
  int keep_dark;                 // 1: mask / weight are NOT zeroed where the image is 0 (the Suggestor's tensors)
};

struct AugParams {
  const AugDesc* descs;
  int B, ch, C, OH, OW;
  const f16* lut;                // fp16(float32(v / 255)) for v = 0..255
  f16* X; f16* y; f16* w;        // [B][ch][OH][OW], [B][C][OH][OW], [B][C][OH][OW]
};

// one lane per output pixel (four pixels per lane with 8-byte stores measured slower: 38 vs 25 us per batch of 8 -- the rows are
// only 512 pixels wide and the rotated gather loses its coalescing)
__global__ __launch_bounds__(256) void augment_batch_kernel(AugParams p) {
#pragma clang fp contract(off)
  const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y, b = blockIdx.z;
  if (ox >= p.OW) return;
  const AugDesc d = p.descs[b];
  // resized crop: pixel (cy, cx) of the rotated image
  const float sy = (float)d.ch / (float)p.OH, sx = (float)d.cw / (float)p.OW;
  const int cy = d.ci + min((int)floorf((float)oy * sy), d.ch - 1);
  const int cx = d.cj + min((int)floorf((float)ox * sx), d.cw - 1);
  int ry, rx;
  bool ok = true;
  if (d.kind == 1) { ry = cy; rx = cx; }
  else if (d.kind == 2) { ry = d.H - 1 - cy; rx = d.W - 1 - cx; }
  else if (d.kind == 3) { ry = cx; rx = d.W - 1 - cy; }
  else if (d.kind == 4) { ry = d.H - 1 - cx; rx = cy; }
  else {
    const float X = d.xg[cx], Y = d.yg[cy];
    const float gx = __fmaf_rn(Y, d.r[1], X * d.r[0]) + d.r[2];
    const float gy = __fmaf_rn(Y, d.r[4], X * d.r[3]) + d.r[5];
    const float ix = ((gx + 1.f) * (float)d.W - 1.f) / 2.f;
    const float iy = ((gy + 1.f) * (float)d.H - 1.f) / 2.f;
    const float fx = rintf(ix), fy = rintf(iy);
    ok = fx >= 0.f && fx <= (float)(d.W - 1) && fy >= 0.f && fy <= (float)(d.H - 1);
    rx = ok ? (int)fx : 0; ry = ok ? (int)fy : 0;
  }
  if (d.vflip) ry = d.H - 1 - ry;
  if (d.hflip) rx = d.W - 1 - rx;
  const long long src = (long long)ry * d.W + rx;
  const long long plane = (long long)p.OH * p.OW, o = (long long)oy * p.OW + ox;
  const f16 zero = p.lut[0];
  bool lit = false;                                   // image[0] != 0 at the source pixel
  for (int c = 0; c < p.ch; ++c) {
    const unsigned char v = ok ? d.image[src * p.ch + c] : (unsigned char)0;
    if (c == 0) lit = ok && (v != 0 || d.keep_dark);
    p.X[((long long)b * p.ch + c) * plane + o] = p.lut[v];
  }
  const f16 wv = lit ? p.lut[d.weight[src]] : zero;
  for (int c = 0; c < p.C; ++c) {
    p.y[((long long)b * p.C + c) * plane + o] = lit ? p.lut[d.mask[src * p.C + c]] : zero;
    p.w[((long long)b * p.C + c) * plane + o] = wv;
  }
}

}  // namespace

extern "C" {

int iunet_augment_desc_bytes(void) { return (int)sizeof(AugDesc); }

int iunet_augment_batch(const void* descs, int B, int ch, int C, int OH, int OW, const void* lut_f16, void* X, void* y, void* w,
                        void* stream) {
  IUNET_REQUIRE(descs && lut_f16 && X && y && w, "augment_batch: null pointer");
  IUNET_REQUIRE(B >= 1 && B <= 65535 && ch >= 1 && ch <= 4 && C >= 1 && C <= 16 && OH >= 1 && OH <= 65535 && OW >= 1,
                "augment_batch: B %d, channels %d, classes %d, output %d x %d", B, ch, C, OH, OW);
  AugParams p;
  p.descs = (const AugDesc*)descs; p.B = B; p.ch = ch; p.C = C; p.OH = OH; p.OW = OW;
  p.lut = (const f16*)lut_f16; p.X = (f16*)X; p.y = (f16*)y; p.w = (f16*)w;
  hipLaunchKernelGGL(augment_batch_kernel, dim3((OW + 255) / 256, OH, B), dim3(256), 0, (hipStream_t)stream, p);
  IUNET_CHECK_HIP(hipGetLastError());
  return IUNET_OK;
}

}  // extern "C"
