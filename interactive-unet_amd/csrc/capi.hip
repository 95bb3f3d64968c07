// extern "C" surface of libiunet.so (see include/iunet.h).  Plain pointers and sizes
// only; the caller (PyTorch-ROCm tensors on the host side) owns every buffer and passes
// the stream it wants the work ordered on.  No allocation, no synchronisation here.
#include "common.h"
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";

void iunet_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// internal launchers (conv3_mfma.hip, pointwise.hip)
int iunet_conv3_launch(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride,
                       const void* wpk, const float* bias, float* stats, int N, int D, int H, int W, int Cin,
                       int Cout, int epi, int layout, hipStream_t stream, const float* in_scale = nullptr,
                       const float* in_shift = nullptr, const void* bw_y = nullptr, long long bw_y_ss = 0,
                       const float* const* bw_par = nullptr);
int iunet_conv3_pick(int nd, int N, int D, int H, int W, int Cin, int Cout);
int iunet_conv3_v4_launch(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride, const void* wpk,
                          const float* bias, float* stats, int N, int D, int H, int W, int Cin, int Cout, int epi,
                          const float* in_scale, const float* in_shift, hipStream_t stream, const void* bw_y, long long bw_y_ss,
                          const float* const* bw_par, int compact, int per_sample, int* query_rows);
int iunet_conv3_v4_stats_parts(int nd, int Cout);
int iunet_conv3_v4_pairs(int nd, int N, int D, int H, int W, int Cin, int Cout, int bw);
int iunet_conv3_tiles(int nd, int N, int D, int H, int W);
long long iunet_pack_conv3_size(int Cout, int Cin, int taps, int mode);
int iunet_pack_conv3_launch(int dtype, const float* w, const float* scale, void* dst, int Cout, int Cin, int taps,
                            int mode, hipStream_t stream);
int iunet_first_conv_launch(int dtype, int nd, const void* x, int in_dtype, long long sN, long long sC, long long sD,
                            long long sH, long long sW, void* y, long long y_sstride, const void* w,
                            const float* bias, float* stats, int N, int D, int H, int W, int Cin, int Cout, int relu,
                            hipStream_t stream, int out8 = 0);
int iunet_pack_first_conv_launch(int dtype, const float* w, const float* scale, void* dst, int Cout, int Cin, int taps,
                                 hipStream_t stream);
int iunet_maxpool_launch(int dtype, int nd, const void* x, long long x_ss, void* y, long long y_ss, int planes, int N,
                         int Do, int Ho, int Wo, hipStream_t stream);
int iunet_convT_launch(int dtype, int nd, const void* x, long long x_ss, void* y, long long y_ss, const void* wpk,
                       const float* bias, int N, int D, int H, int W, int Cin, int Cout, hipStream_t stream, int out8 = 0);
int iunet_maxpool_q_launch(int nd, const void* x, long long x_ss, void* y, long long y_ss, int planes16, int N, int Do, int Ho, int Wo,
                           hipStream_t stream);
int iunet_conv3_f8_launch_q(int dtype, int nd, const void* x, long long x_sstride, int x_fmt, void* y, long long y_sstride, int y_fmt,
                            const void* wpk, const float* wscale, const float* bias, int N, int D, int H, int W, int Cin, int Cout,
                            int epi, float* workspace, hipStream_t stream);
int iunet_pack_convT_launch(int dtype, const float* w, void* dst, int Cin, int Cout, int npos, hipStream_t stream);
int iunet_head_launch(int dtype, const void* x, long long x_ss, int C0, const float* w, const float* bias, int ncls,
                      float* logits, float* probs, unsigned char* cls, long long oN, long long oC, long long oD,
                      long long oH, long long oW, float divisor, int accumulate, int N, int D, int H, int W,
                      hipStream_t stream);

// fp32 parity mode (precise_f32.hip)
long long iunet_f32_pack_size(int Cout, int Cin, int taps);
int iunet_f32_pack_launch(const float* w, float* dst, float* bias_out, const float* gamma, const float* beta,
                          const float* mean, const float* var, float eps, int Cout, int Cin, int taps, int transposed,
                          hipStream_t stream);
int iunet_f32_conv_launch(int nd, const void* x, int in_dtype, const long long* st, float* y, long long y_ss, const float* wpk,
                          const float* bias, int N, int D, int H, int W, int Cin, int Cout, int relu, int transposed,
                          hipStream_t stream);
int iunet_f32_maxpool_launch(int nd, const float* x, long long x_ss, float* y, long long y_ss, int C, int N, int Do, int Ho,
                             int Wo, hipStream_t stream);
int iunet_f32_head_launch(const float* x, long long x_ss, int C0, const float* w, const float* bias, int ncls, float* logits,
                          float* probs, unsigned char* cls, const long long* os, float divisor, int accumulate, int N, int D,
                          int H, int W, hipStream_t stream);

// fp8 matrix-core convolution (conv3_f8.hip)
long long iunet_f8_pack_bytes(int Cout, int Cin, int taps);
int iunet_f8_pack_launch(const float* w, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                         void* dst, float* wscale, float* bias_out, int Cout, int Cin, int taps, hipStream_t stream);
int iunet_conv3_f8_launch(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride, const void* wpk,
                          const float* wscale, const float* bias, int N, int D, int H, int W, int Cin, int Cout, int epi,
                          float* workspace, hipStream_t stream);
long long iunet_conv3_f8_workspace_floats(int nd, int N, int D, int H, int W, int Cin, int Cout);

#define DT_OK(dt) IUNET_REQUIRE((dt) == 0 || (dt) == 1, "dtype must be 0 (f16) or 1 (bf16), got %d", (dt))

extern "C" {

const char* iunet_last_error(void) { return g_err; }
int iunet_abi_version(void) { return 1; }

int iunet_conv3_num_tiles(int nd, int N, int D, int H, int W) { return iunet_conv3_tiles(nd, N, D, H, W); }

// rows of the statistics buffer a conv3_fwd launch with this layout writes (and bn_finalize must read)
int iunet_conv3_stats_parts(int nd, int N, int D, int H, int W, int Cout, int layout) {
  return layout >= 2 ? iunet_conv3_v4_stats_parts(nd, Cout) : iunet_conv3_tiles(nd, N, D, H, W);
}

long long iunet_pack_conv3_elems(int Cout, int Cin, int taps, int mode) { return iunet_pack_conv3_size(Cout, Cin, taps, mode); }

int iunet_pack_conv3(int dtype, const void* w, const void* scale, void* dst, int Cout, int Cin, int taps, int mode,
                     void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(w && dst, "pack_conv3: null pointer");
  IUNET_REQUIRE(taps == 9 || taps == 27, "pack_conv3: taps must be 9 or 27 (got %d)", taps);
  return iunet_pack_conv3_launch(dtype, (const float*)w, (const float*)scale, dst, Cout, Cin, taps, mode, (hipStream_t)stream);
}

int iunet_pack_first_conv(int dtype, const void* w, const void* scale, void* dst, int Cout, int Cin, int taps,
                          void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(w && dst, "pack_first_conv: null pointer");
  IUNET_REQUIRE(Cout % 32 == 0 && Cin >= 1 && Cin <= 12, "pack_first_conv: Cout multiple of 32, Cin 1..4 (3..12: the virtual operator of iunet_x2_prep)");
  return iunet_pack_first_conv_launch(dtype, (const float*)w, (const float*)scale, dst, Cout, Cin, taps, (hipStream_t)stream);
}

long long iunet_pack_first_conv_elems(int Cout, int Cin, int taps) { return (long long)Cout * (((taps * Cin + 31) / 32) * 32); }

int iunet_pack_convT(int dtype, const void* w, void* dst, int Cin, int Cout, int npos, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(w && dst, "pack_convT: null pointer");
  IUNET_REQUIRE(npos == 4 || npos == 8, "pack_convT: npos must be 4 or 8");
  IUNET_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, "pack_convT: channels must be multiples of 32");
  return iunet_pack_convT_launch(dtype, (const float*)w, dst, Cin, Cout, npos, (hipStream_t)stream);
}

int iunet_conv3_pick_layout(int nd, int N, int D, int H, int W, int Cin, int Cout) {
  return iunet_conv3_pick(nd, N, D, H, W, Cin, Cout);
}

// can this launch run on layout 3 (compact operator, padding-free step: conv3_v4.hip NP)?  Fused BatchNorm-backward sums in 2-D on the
// resident-weights variant only (Cin <= 64).  3-D:
// streamed weights (Cin > 32), a fused input activation up to 192 input channels (LDS).  2-D (the cross-pair step): every channel
// count, a fused input activation on the resident-weights variant only (Cin <= 64: the layers the training forward fuses).
// IUNET_NO_COMPACT2D=1: A/B switch back to layouts 0 / 1 / 2 in 2-D.
int iunet_conv3_compact_ok(int nd, int N, int D, int H, int W, int Cin, int Cout, int act, int bw) {
  static const bool off2d = getenv("IUNET_NO_COMPACT2D") != nullptr;
  static const bool no_bw2d = getenv("IUNET_NO_COMPACT2D_BW") != nullptr;      // A/B: the fused-sums data gradient back on layout 2
  if ((nd != 2 && nd != 3) || N < 1 || D < 1 || H < 1 || W < 1 || Cin < 32 || Cout < 32 || Cin % 32 || Cout % 32 || (bw && nd != 2)) return 0;
  if (nd == 2) return !off2d && D == 1 && !((act || bw) && Cin > 64) && !(bw && no_bw2d);
  if (Cin <= 32) return 0;
  return !(act && Cin > 192);        // independent of the grid: a layer keeps one summation order whatever the launch size
}

int iunet_conv3_tile_pairs(int nd, int N, int D, int H, int W, int Cin, int Cout) {
  if ((nd != 2 && nd != 3) || N < 1 || D < 1 || H < 1 || W < 1 || Cin < 32 || Cout < 32 || Cin % 32 || Cout % 32) return 0;
  return iunet_conv3_v4_pairs(nd, N, D, H, W, Cin, Cout, 0);
}

int iunet_conv3_fwd(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride,
                    const void* wpk, const void* bias, void* stats, int N, int D, int H, int W, int Cin, int Cout,
                    int epi, int layout, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(x && y && wpk, "conv3: null pointer");
  IUNET_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "conv3: bad shape %d %d %d %d", N, D, H, W);
  IUNET_REQUIRE(epi >= 0 && epi <= 2, "conv3: bad epilogue %d", epi);
  IUNET_REQUIRE(layout >= 0 && layout <= 3, "conv3: layout must be 0, 1, 2 or 3 (got %d)", layout);
  return iunet_conv3_launch(dtype, nd, x, x_sstride, y, y_sstride, wpk, (const float*)bias, (float*)stats, N, D, H, W,
                            Cin, Cout, epi, layout, (hipStream_t)stream);
}

// iunet_conv3_fwd whose input is relu(in_scale[c] * x + in_shift[c]) (training: the BatchNorm + ReLU of the previous
// conv, applied by the loader waves instead of a separate pass); layout 2 only.
int iunet_conv3_fwd_act(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride,
                        const void* wpk, const void* bias, void* stats, const void* in_scale, const void* in_shift,
                        int N, int D, int H, int W, int Cin, int Cout, int epi, int layout, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(x && y && wpk && in_scale && in_shift, "conv3_act: null pointer");
  IUNET_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "conv3: bad shape %d %d %d %d", N, D, H, W);
  IUNET_REQUIRE(epi >= 0 && epi <= 2, "conv3: bad epilogue %d", epi);
  IUNET_REQUIRE(layout == 2 || layout == 3, "conv3_act: the fused input activation exists in layouts 2 and 3 only (got %d)", layout);
  return iunet_conv3_launch(dtype, nd, x, x_sstride, y, y_sstride, wpk, (const float*)bias, (float*)stats, N, D, H, W,
                            Cin, Cout, epi, layout, (hipStream_t)stream, (const float*)in_scale, (const float*)in_shift);
}

// GroupNorm: the conv's statistics epilogue per SAMPLE.  iunet_conv3_sample_stats_rows: rows per sample that
// iunet_conv3_fwd_sample_stats writes on this grid and layout (2 or 3), or 0 when the launch has no per-sample form (layouts 0 / 1; a
// grid with fewer than 8 bricks per sample: the caller runs its own statistics pass, iunet_gn_relu_fwd).  iunet_conv3_fwd_sample_stats =
// iunet_conv3_fwd (epi 0, no bias) writing stats [N][rows][Cout][2] = (sum, sum of squares) of the fp32 accumulators: the slab of
// iunet_gn_relu_fwd_rows / iunet_gn_relu_pool_fwd_rows.  The brick schedule is one sample's, walked once per sample.
int iunet_conv3_sample_stats_rows(int dtype, int nd, int N, int D, int H, int W, int Cin, int Cout, int layout) {
  DT_OK(dtype);
  if (layout < 2 || (nd != 2 && nd != 3) || N < 1 || D < 1 || H < 1 || W < 1 || Cin < 32 || Cout < 32 || Cin % 32 || Cout % 32) return 0;
  if (layout == 3 && !iunet_conv3_compact_ok(nd, N, D, H, W, Cin, Cout, 0, 0)) return 0;
  int rows = 0;
  if (iunet_conv3_v4_launch(dtype, nd, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, N, D, H, W, Cin, Cout, 0, nullptr, nullptr, nullptr,
                            nullptr, 0, nullptr, layout == 3, 1, &rows) != IUNET_OK) return 0;
  return rows;
}

int iunet_conv3_fwd_sample_stats(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride, const void* wpk,
                                 void* stats, int N, int D, int H, int W, int Cin, int Cout, int layout, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(x && y && wpk && stats, "conv3_fwd_sample_stats: null pointer");
  IUNET_REQUIRE_GRID("conv3_fwd_sample_stats", N, D, H, W);
  IUNET_REQUIRE(layout == 2 || layout == 3, "conv3_fwd_sample_stats: layout 2 or 3 (got %d)", layout);
  IUNET_REQUIRE(nd == 2 || nd == 3, "conv3: nd must be 2 or 3 (got %d)", nd);
  IUNET_REQUIRE(nd == 3 || D == 1, "conv3: 2-D conv needs D == 1");
  return iunet_conv3_v4_launch(dtype, nd, x, x_sstride, y, y_sstride, wpk, nullptr, (float*)stats, N, D, H, W, Cin, Cout, 0, nullptr, nullptr,
                               (hipStream_t)stream, nullptr, 0, nullptr, layout == 3, 1, nullptr);
}

// iunet_conv3_dgrad_bnstats_lay per SAMPLE (GroupNorm: mean / invstd / scale / shift are [N][Cout] rows, the sums are wanted per sample):
// stats [N][rows][Cout][2], rows = iunet_conv3_sample_stats_rows(dtype, nd, N, D, H, W, Cin, Cout, layout) > 0 -- the slab of
// iunet_gn_relu_bwd_rows.
int iunet_conv3_dgrad_sample_bnstats(int dtype, int nd, const void* dy, long long dy_sstride, void* dz, long long dz_sstride, const void* wpk,
                                     void* stats, const void* yp, long long yp_sstride, const void* mean, const void* invstd, const void* scale,
                                     const void* shift, int N, int D, int H, int W, int Cin, int Cout, int layout, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(dy && dz && wpk && stats && yp && mean && invstd && scale && shift, "conv3_dgrad_sample_bnstats: null pointer");
  IUNET_REQUIRE_GRID("conv3_dgrad_sample_bnstats", N, D, H, W);
  IUNET_REQUIRE(layout == 2 || (layout == 3 && nd == 2 && Cin <= 64), "conv3_dgrad_sample_bnstats: layout 2, or 3 in 2-D up to 64 input channels (got %d)", layout);
  IUNET_REQUIRE(nd == 2 || nd == 3, "conv3: nd must be 2 or 3 (got %d)", nd);
  IUNET_REQUIRE(nd == 3 || D == 1, "conv3: 2-D conv needs D == 1");
  const float* par[4] = {(const float*)mean, (const float*)invstd, (const float*)scale, (const float*)shift};
  return iunet_conv3_v4_launch(dtype, nd, dy, dy_sstride, dz, dz_sstride, wpk, nullptr, (float*)stats, N, D, H, W, Cin, Cout, 0, nullptr, nullptr,
                               (hipStream_t)stream, yp, yp_sstride, par, layout == 3, 1, nullptr);
}

// iunet_conv3_fwd used as the data gradient of a conv whose INPUT was z = relu(bn(yp)): besides dz (its output) it accumulates
// the BatchNorm-backward sums of that producer layer -- sum dz', sum dz' * xhat with dz' = dz where z > 0 -- in its epilogue
// (the reduction pass of iunet_bn_relu_bwd over dz and yp goes away); stats: [iunet_conv3_stats_parts(.., layout 2)][Cout][2].
int iunet_conv3_dgrad_bnstats_lay(int dtype, int nd, const void* dy, long long dy_sstride, void* dz, long long dz_sstride,
                                  const void* wpk, void* stats, const void* yp, long long yp_sstride, const void* mean,
                                  const void* invstd, const void* scale, const void* shift, int N, int D, int H, int W, int Cin,
                                  int Cout, int layout, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(dy && dz && wpk && stats && yp && mean && invstd && scale && shift, "conv3_dgrad_bnstats: null pointer");
  IUNET_REQUIRE_GRID("conv3_dgrad_bnstats", N, D, H, W);
  IUNET_REQUIRE(layout == 2 || layout == 3, "conv3_dgrad_bnstats: layout 2, or 3 (the compact operator: 2-D, Cin <= 64; iunet_conv3_compact_ok(.., bw = 1)) -- got %d", layout);
  const float* par[4] = {(const float*)mean, (const float*)invstd, (const float*)scale, (const float*)shift};
  return iunet_conv3_launch(dtype, nd, dy, dy_sstride, dz, dz_sstride, wpk, nullptr, (float*)stats, N, D, H, W, Cin, Cout, 0, layout,
                            (hipStream_t)stream, nullptr, nullptr, yp, yp_sstride, par);
}

// (the layout-2 form, kept for callers of the first ABI)
int iunet_conv3_dgrad_bnstats(int dtype, int nd, const void* dy, long long dy_sstride, void* dz, long long dz_sstride,
                              const void* wpk, void* stats, const void* yp, long long yp_sstride, const void* mean,
                              const void* invstd, const void* scale, const void* shift, int N, int D, int H, int W, int Cin,
                              int Cout, void* stream) {
  return iunet_conv3_dgrad_bnstats_lay(dtype, nd, dy, dy_sstride, dz, dz_sstride, wpk, stats, yp, yp_sstride, mean, invstd, scale, shift,
                                       N, D, H, W, Cin, Cout, 2, stream);
}

int iunet_first_conv_fwd(int dtype, int nd, const void* x, int in_dtype, const long long* in_strides, void* y,
                         long long y_sstride, const void* w, const void* bias, void* stats, int N, int D, int H, int W,
                         int Cin, int Cout, int relu, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(x && y && w && in_strides, "first_conv: null pointer");
  IUNET_REQUIRE(in_dtype >= 0 && in_dtype <= 3, "first_conv: bad input dtype %d", in_dtype);
  IUNET_REQUIRE(nd == 2 || nd == 3, "first_conv: nd must be 2 or 3");
  IUNET_REQUIRE(nd == 3 || D == 1, "first_conv: 2-D needs D == 1");
  return iunet_first_conv_launch(dtype, nd, x, in_dtype, in_strides[0], in_strides[1], in_strides[2], in_strides[3],
                                 in_strides[4], y, y_sstride, w, (const float*)bias, (float*)stats, N, D,
                                 H, W, Cin, Cout, relu, (hipStream_t)stream);
}

int iunet_maxpool_fwd(int dtype, int nd, const void* x, long long x_ss, void* y, long long y_ss, int C, int N, int Do,
                      int Ho, int Wo, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(x && y, "maxpool: null pointer");
  IUNET_REQUIRE(C % 8 == 0, "maxpool: C must be a multiple of 8");
  return iunet_maxpool_launch(dtype, nd, x, x_ss, y, y_ss, C / 8, N, Do, Ho, Wo, (hipStream_t)stream);
}

int iunet_convT_fwd(int dtype, int nd, const void* x, long long x_ss, void* y, long long y_ss, const void* wpk,
                    const void* bias, int N, int D, int H, int W, int Cin, int Cout, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(x && y && wpk, "convT: null pointer");
  return iunet_convT_launch(dtype, nd, x, x_ss, y, y_ss, wpk, (const float*)bias, N, D, H, W, Cin, Cout, (hipStream_t)stream);
}

int iunet_head_fwd(int dtype, const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls,
                   void* logits, void* probs, void* cls, const long long* out_strides, float divisor, int accumulate,
                   int N, int D, int H, int W, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(x && w && bias && out_strides, "head: null pointer");
  IUNET_REQUIRE(C0 % 8 == 0, "head: C0 must be a multiple of 8");
  return iunet_head_launch(dtype, x, x_ss, C0, (const float*)w, (const float*)bias, ncls, (float*)logits, (float*)probs,
                           (unsigned char*)cls, out_strides[0], out_strides[1], out_strides[2], out_strides[3],
                           out_strides[4], divisor, accumulate, N, D, H, W, (hipStream_t)stream);
}

/* ---- fp32 parity mode (precise_f32.hip): planar fp32 activations, v_mfma_f32_16x16x4_f32 ------------------- */
long long iunet_f32_pack_conv_elems(int Cout, int Cin, int taps) {
  if (Cout <= 0 || Cout % 32 || Cin <= 0 || taps <= 0) return 0;
  return iunet_f32_pack_size(Cout, Cin, taps);
}

int iunet_f32_pack_conv(const void* w, void* dst, void* bias_out, const void* gamma, const void* beta, const void* mean,
                        const void* var, float eps, int Cout, int Cin, int taps, int transposed, void* stream) {
  IUNET_REQUIRE(w && dst, "f32_pack_conv: null pointer");
  IUNET_REQUIRE(Cout > 0 && Cout % 32 == 0 && Cin > 0, "f32_pack_conv: Cout must be a positive multiple of 32, Cin > 0 (got %d, %d)", Cout, Cin);
  IUNET_REQUIRE(transposed ? (taps == 4 || taps == 8) : (taps == 9 || taps == 27 || taps == 1),
                "f32_pack_conv: taps must be 9 / 27 (conv), 1 (pointwise) or 4 / 8 (transposed), got %d", taps);
  IUNET_REQUIRE(!gamma || (beta && mean && var && bias_out), "f32_pack_conv: a BatchNorm fold needs gamma, beta, mean, var and bias_out");
  return iunet_f32_pack_launch((const float*)w, (float*)dst, (float*)bias_out, (const float*)gamma, (const float*)beta,
                               (const float*)mean, (const float*)var, eps, Cout, Cin, taps, transposed, (hipStream_t)stream);
}

int iunet_f32_conv_fwd(int nd, const void* x, int in_dtype, const long long* in_strides, void* y, long long y_ss,
                       const void* wpk, const void* bias, int N, int D, int H, int W, int Cin, int Cout, int relu,
                       int transposed, void* stream) {
  IUNET_REQUIRE(x && y && wpk && in_strides, "f32_conv: null pointer");
  IUNET_REQUIRE(nd == 2 || nd == 3, "f32_conv: nd must be 2 or 3");
  IUNET_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && (nd == 3 || D == 1), "f32_conv: bad shape %d %d %d %d", N, D, H, W);
  IUNET_REQUIRE(in_dtype >= 0 && in_dtype <= 3, "f32_conv: bad input dtype %d", in_dtype);
  IUNET_REQUIRE(Cout > 0 && Cout % 32 == 0 && Cin > 0, "f32_conv: Cout must be a positive multiple of 32, Cin > 0 (got %d, %d)", Cout, Cin);
  return iunet_f32_conv_launch(nd, x, in_dtype, in_strides, (float*)y, y_ss, (const float*)wpk, (const float*)bias, N, D, H, W,
                               Cin, Cout, relu, transposed, (hipStream_t)stream);
}

int iunet_f32_maxpool_fwd(int nd, const void* x, long long x_ss, void* y, long long y_ss, int C, int N, int Do, int Ho, int Wo,
                          void* stream) {
  IUNET_REQUIRE(x && y, "f32_maxpool: null pointer");
  IUNET_REQUIRE(nd == 2 || nd == 3, "f32_maxpool: nd must be 2 or 3");
  IUNET_REQUIRE(C > 0 && N > 0 && Do > 0 && Ho > 0 && Wo > 0, "f32_maxpool: bad shape");
  return iunet_f32_maxpool_launch(nd, (const float*)x, x_ss, (float*)y, y_ss, C, N, Do, Ho, Wo, (hipStream_t)stream);
}

int iunet_f32_head_fwd(const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls, void* logits,
                       void* probs, void* cls, const long long* out_strides, float divisor, int accumulate, int N, int D, int H,
                       int W, void* stream) {
  IUNET_REQUIRE(x && w && bias && out_strides, "f32_head: null pointer");
  IUNET_REQUIRE(ncls >= 2 && ncls <= 10, "f32_head: num_classes must be 2..10 (got %d)", ncls);
  return iunet_f32_head_launch((const float*)x, x_ss, C0, (const float*)w, (const float*)bias, ncls, (float*)logits,
                               (float*)probs, (unsigned char*)cls, out_strides, divisor, accumulate, N, D, H, W,
                               (hipStream_t)stream);
}

/* ---- fp8 matrix cores (conv3_f8.hip): BASELINE config C5 ---------------------------------------------------- */
int iunet_f8_pack_order(int taps, int Cin) { return iunet_f8_k128(taps, Cin); }

long long iunet_f8_pack_conv3_bytes(int Cout, int Cin, int taps) {
  if (Cout <= 0 || Cout % 32 || Cin <= 0 || Cin % 32 || (taps != 9 && taps != 27)) return 0;
  return iunet_f8_pack_bytes(Cout, Cin, taps);
}

int iunet_f8_pack_conv3(const void* w, const void* gamma, const void* beta, const void* mean, const void* var, float eps,
                        void* dst, void* wscale, void* bias_out, int Cout, int Cin, int taps, void* stream) {
  IUNET_REQUIRE(w && dst && wscale, "f8_pack_conv3: null pointer");
  IUNET_REQUIRE(Cout > 0 && Cout % 32 == 0 && Cin > 0 && Cin % 32 == 0, "f8_pack_conv3: channels must be positive multiples of 32 (%d, %d)", Cout, Cin);
  IUNET_REQUIRE(taps == 9 || taps == 27, "f8_pack_conv3: taps must be 9 or 27 (got %d)", taps);
  IUNET_REQUIRE(!gamma || (beta && mean && var && bias_out), "f8_pack_conv3: a BatchNorm fold needs gamma, beta, mean, var and bias_out");
  return iunet_f8_pack_launch((const float*)w, (const float*)gamma, (const float*)beta, (const float*)mean, (const float*)var, eps,
                              dst, (float*)wscale, (float*)bias_out, Cout, Cin, taps, (hipStream_t)stream);
}

long long iunet_conv3_f8_workspace_elems(int nd, int N, int D, int H, int W, int Cin, int Cout) {
  if ((nd != 2 && nd != 3) || N < 1 || D < 1 || H < 1 || W < 1 || Cin < 32 || Cout < 32 || Cin % 32 || Cout % 32) return 0;
  return iunet_conv3_f8_workspace_floats(nd, N, D, H, W, Cin, Cout);
}

int iunet_conv3_f8_fwd(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride, const void* wpk,
                       const void* wscale, const void* bias, int N, int D, int H, int W, int Cin, int Cout, int epi,
                       void* workspace, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(x && y && wpk && wscale, "conv3_f8: null pointer");
  IUNET_REQUIRE(nd == 2 || nd == 3, "conv3_f8: nd must be 2 or 3");
  IUNET_REQUIRE_GRID("conv3_f8", N, D, H, W);
  IUNET_REQUIRE(nd == 3 || D == 1, "conv3_f8: 2-D needs D == 1");
  IUNET_REQUIRE(Cin >= 32 && Cout >= 32 && Cin % 32 == 0 && Cout % 32 == 0, "conv3_f8: channels must be positive multiples of 32 (%d, %d)", Cin, Cout);
  IUNET_REQUIRE(epi >= 0 && epi <= 2, "conv3_f8: bad epilogue %d", epi);
  IUNET_REQUIRE(epi == 0 || bias != nullptr, "conv3_f8: epilogue %d needs a bias", epi);
  return iunet_conv3_f8_launch(dtype, nd, x, x_sstride, y, y_sstride, wpk, (const float*)wscale, (const float*)bias, N, D, H, W,
                               Cin, Cout, epi, (float*)workspace, (hipStream_t)stream);
}

/* e4m3 activation planes (format 1: C / 16 planes of [D][H][W][16 bytes], sample strides in BYTES) between the layers of the fp8
 * network: what the K = 128 convolution reads by LDS-DMA.  Producers round their 16-bit result once more to e4m3 -- the rounding
 * the consumer conv's loader would have applied -- so the network's values do not depend on the format of the tensors in between. */
int iunet_conv3_f8_fwd_q(int dtype, int nd, const void* x, long long x_sstride, int x_fmt, void* y, long long y_sstride, int y_fmt,
                         const void* wpk, const void* wscale, const void* bias, int N, int D, int H, int W, int Cin, int Cout, int epi,
                         void* workspace, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(x && y && wpk && wscale, "conv3_f8_q: null pointer");
  IUNET_REQUIRE(nd == 2 || nd == 3, "conv3_f8_q: nd must be 2 or 3");
  IUNET_REQUIRE((x_fmt == 0 || x_fmt == 1) && (y_fmt == 0 || y_fmt == 1), "conv3_f8_q: formats are 0 (16-bit) or 1 (e4m3 planes)");
  IUNET_REQUIRE_GRID("conv3_f8_q", N, D, H, W);
  IUNET_REQUIRE(nd == 3 || D == 1, "conv3_f8_q: 2-D needs D == 1");
  IUNET_REQUIRE(Cin >= 32 && Cout >= 32 && Cin % 32 == 0 && Cout % 32 == 0, "conv3_f8_q: channels must be positive multiples of 32 (%d, %d)", Cin, Cout);
  IUNET_REQUIRE(epi >= 0 && epi <= 2, "conv3_f8_q: bad epilogue %d", epi);
  IUNET_REQUIRE(epi == 0 || bias != nullptr, "conv3_f8_q: epilogue %d needs a bias", epi);
  return iunet_conv3_f8_launch_q(dtype, nd, x, x_sstride, x_fmt, y, y_sstride, y_fmt, wpk, (const float*)wscale, (const float*)bias,
                                 N, D, H, W, Cin, Cout, epi, (float*)workspace, (hipStream_t)stream);
}

int iunet_first_conv_fwd_q(int dtype, int nd, const void* x, int in_dtype, const long long* in_strides, void* y,
                           long long y_sstride_bytes, const void* w, const void* bias, int N, int D, int H, int W,
                           int Cin, int Cout, int relu, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(x && y && w && in_strides, "first_conv_q: null pointer");
  IUNET_REQUIRE(in_dtype >= 0 && in_dtype <= 3, "first_conv_q: bad input dtype %d", in_dtype);
  IUNET_REQUIRE(nd == 2 || nd == 3, "first_conv_q: nd must be 2 or 3");
  IUNET_REQUIRE(nd == 3 || D == 1, "first_conv_q: 2-D needs D == 1");
  IUNET_REQUIRE_GRID("first_conv_q", N, D, H, W);
  return iunet_first_conv_launch(dtype, nd, x, in_dtype, in_strides[0], in_strides[1], in_strides[2], in_strides[3],
                                 in_strides[4], y, y_sstride_bytes, w, (const float*)bias, nullptr, N, D, H, W, Cin, Cout, relu,
                                 (hipStream_t)stream, 1);
}

int iunet_maxpool_q_fwd(int nd, const void* x, long long x_ss_bytes, void* y, long long y_ss_bytes, int C, int N, int Do, int Ho, int Wo,
                        void* stream) {
  IUNET_REQUIRE(x && y, "maxpool_q: null pointer");
  IUNET_REQUIRE(nd == 2 || nd == 3, "maxpool_q: nd must be 2 or 3");
  IUNET_REQUIRE(C > 0 && C % 16 == 0, "maxpool_q: C must be a positive multiple of 16 (got %d)", C);
  IUNET_REQUIRE_GRID("maxpool_q", N, Do, Ho, Wo);
  return iunet_maxpool_q_launch(nd, x, x_ss_bytes, y, y_ss_bytes, C / 16, N, Do, Ho, Wo, (hipStream_t)stream);
}

int iunet_convT_fwd_q(int dtype, int nd, const void* x, long long x_ss, void* y, long long y_ss_bytes, const void* wpk,
                      const void* bias, int N, int D, int H, int W, int Cin, int Cout, void* stream) {
  DT_OK(dtype);
  IUNET_REQUIRE(x && y && wpk, "convT_q: null pointer");
  IUNET_REQUIRE(nd == 2 || nd == 3, "convT_q: nd must be 2 or 3");
  IUNET_REQUIRE_GRID("convT_q", N, D, H, W);
  return iunet_convT_launch(dtype, nd, x, x_ss, y, y_ss_bytes, wpk, (const float*)bias, N, D, H, W, Cin, Cout, (hipStream_t)stream, 1);
}

}  // extern "C"
