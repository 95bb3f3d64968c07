/* libiunet.so -- MI355X (gfx950) native kernels for the U-Net train + predict hot path
 * of laprade117/interactive-unet.  C ABI: plain pointers and sizes, no torch types.
 *
 * The reference has NO native / FFI boundary for this path: its device
 * math is whatever torch + segmentation_models_pytorch run under these Python sites
 *   interactive_unet/unet.py:65-69     UNet.forward  = softmax(model(x))
 *   interactive_unet/unet.py:88-102    training_step = forward + loss (metrics.py) + autograd
 *   interactive_unet/unet.py:71-73     AdamW
 *   interactive_unet/predict.py:79-112 predict_block (2.5-D)
 *   interactive_unet/predict.py:244-256 blend-accumulate, normalise, quantise
 *   interactive_unet/predict.py:291-316 get_padded_block (reflect)
 * Each entry point below names the reference lines it replaces.  The Python host side
 * (interactive-unet_amd/interactive_unet/) binds them with ctypes; INTEGRATION.md shows
 * the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; iunet_last_error() (thread
 *     local) holds the message.  No exceptions, no allocation, no device-wide sync.
 *   - the caller owns every buffer; `stream` is a hipStream_t the work is ordered on.
 *   - dtype: 0 = fp16, 1 = bf16 activations / packed weights (fp32 accumulation always).
 *   - activation tensors are channel-blocked NHWC ("NHWC8c"): C/8 planes of
 *     [D][H][W][8]; `*_sstride` = elements between consecutive samples (a view may be a
 *     slice of planes of a wider concat buffer).  2-D tensors have D = 1.
 */
#ifndef IUNET_H
#define IUNET_H
#ifdef __cplusplus
extern "C" {
#endif

const char* iunet_last_error(void);
int iunet_abi_version(void);

/* ---- weight packing (host fp32 master weights -> MFMA fragment order) ---------------- */
/* conv weights fp32 [Cout][Cin][taps] (torch Conv{2,3}d layout); optional per-cout scale
 * folds an eval-mode BatchNorm.  mode 0: forward operator; mode 1: data-gradient operator
 * (channels transposed, taps mirrored).  dst: Cout*Cin*taps elements of `dtype`. */
int iunet_pack_conv3(int dtype, const void* w, const void* scale, void* dst, int Cout, int Cin, int taps, int mode,
                     void* stream);
/* first conv (Cin <= 4): dst fp32 [taps][Cin][Cout], values rounded through `dtype`. */
int iunet_pack_first_conv(int dtype, const void* w, const void* scale, void* dst, int Cout, int Cin, int taps,
                          void* stream);
/* ConvTranspose k2 s2 weights fp32 [Cin][Cout][2^d]. */
int iunet_pack_convT(int dtype, const void* w, void* dst, int Cin, int Cout, int npos, void* stream);

/* ---- forward kernels (replace the smp conv stack under unet.py:65-69) ----------------- */
/* 3^d conv, stride 1, pad 1, implicit GEMM on MFMA.  epi: 0 raw, 1 +bias, 2 +bias+ReLU.
 * stats (optional): fp32 [iunet_conv3_num_tiles][Cout][2] partial sum / sum of squares of
 * the raw output (BatchNorm batch statistics), reduced by the caller. */
int iunet_conv3_fwd(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride,
                    const void* wpk, const void* bias, void* stats, int N, int D, int H, int W, int Cin, int Cout,
                    int epi, void* stream);
int iunet_conv3_num_tiles(int nd, int N, int D, int H, int W);
/* first conv reads the caller's tensor directly: in_dtype 0 f32, 1 f16, 2 u8 (x/255,
 * predict.py:30), 3 bf16; in_strides = element strides (n, c, d, h, w). */
int iunet_first_conv_fwd(int dtype, int nd, const void* x, int in_dtype, const long long* in_strides, void* y,
                         long long y_sstride, const void* w, const void* bias, void* stats, int N, int D, int H, int W,
                         int Cin, int Cout, int relu, void* stream);
int iunet_first_conv_num_blocks(int N, int D, int H, int W);
int iunet_maxpool_fwd(int dtype, int nd, const void* x, long long x_ss, void* y, long long y_ss, int C, int N, int Do,
                      int Ho, int Wo, void* stream);
int iunet_convT_fwd(int dtype, int nd, const void* x, long long x_ss, void* y, long long y_ss, const void* wpk,
                    const void* bias, int N, int D, int H, int W, int Cin, int Cout, void* stream);
/* 1x1 head + softmax (unet.py:63-69) + class map = argmax of the probabilities
 * (predict.py:38).  logits / probs fp32 with element strides out_strides (n, c, d, h, w):
 * probs = ((accumulate ? probs : 0) + p) / divisor -- the 2.5-D accumulation of
 * predict.py:101-110 with the slice axis folded into the strides.  cls uint8 [N][D*H*W]. */
int iunet_head_fwd(int dtype, const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls,
                   void* logits, void* probs, void* cls, const long long* out_strides, float divisor, int accumulate,
                   int N, int D, int H, int W, void* stream);

/* ---- whole-volume prediction (predict.py:201-256) -------------------------------------- */
/* get_padded_block (predict.py:291-316): reflect-padded S^3 uint8 block of a device volume. */
int iunet_gather_block(const void* vol, int Vz, int Vy, int Vx, int i0, int j0, int k0, int S, void* out, void* stream);
/* pred[blk] += P[local] * window[local]; weight[blk] += window[local] (predict.py:244-245). */
int iunet_blend_accumulate(void* pred, void* weight, const void* P, const void* window, int Vz, int Vy, int Vx, int C,
                           int S, const int* block, const int* local, void* stream);
/* uint8(255 * pred / max(weight, eps)), truncating (predict.py:255). */
int iunet_normalize_quantize(const void* pred, const void* weight, void* out_u8, long long nvox, int C, float eps,
                             void* stream);
int iunet_div_f32(void* p, long long n, float d, void* stream);

#ifdef __cplusplus
}
#endif
#endif
