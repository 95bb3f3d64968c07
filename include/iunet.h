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
 * folds an eval-mode BatchNorm.  mode bit 0: data-gradient operator (channels transposed, taps
 * mirrored); mode bit 1: K16 fragment order for weight layout 1 (see iunet_conv3_pick_layout); mode bit 2: the compact K16 order (no padded filter column: layout 3 for taps = 27; for taps = 9 the cross-pair step of the 2-D split-precision conv).  dst: iunet_pack_conv3_elems(...) elements of `dtype`. */
long long iunet_pack_conv3_elems(int Cout, int Cin, int taps, int mode);
int iunet_pack_conv3(int dtype, const void* w, const void* scale, void* dst, int Cout, int Cin, int taps, int mode,
                     void* stream);
/* first conv (Cin <= 4) runs on MFMA with K = taps*Cin padded to 32: dst = iunet_pack_first_conv_elems
 * elements of `dtype` in fragment order. */
long long iunet_pack_first_conv_elems(int Cout, int Cin, int taps);
int iunet_pack_first_conv(int dtype, const void* w, const void* scale, void* dst, int Cout, int Cin, int taps,
                          void* stream);
/* ConvTranspose k2 s2 weights fp32 [Cin][Cout][2^d]. */
int iunet_pack_convT(int dtype, const void* w, void* dst, int Cin, int Cout, int npos, void* stream);

/* All layers in one launch: `descs` = device array of n descriptors (struct layout: csrc/pack_batch.hip PackDesc,
 * mirrored by interactive_unet/_native.py; iunet_pack_desc_bytes() = its size).  kind 0/1 conv3 layout 0/1, 2 first
 * conv, 3 convT, 4 convT data-gradient; a non-NULL gamma folds BatchNorm (scale = gamma / sqrt(var + eps)) and writes
 * the folded bias.  Element mappings are identical to the per-layer entry points above.  A descriptor with a qscale
 * buffer (fp32 [Cout]) has its folded weights quantised to OCP e4m3 values x a per-output-channel power-of-two scale
 * (BASELINE config C5: fp8 weights, 16-bit activations; the products are exact in f16 / bf16, so the MFMA path is the
 * same); pass quant_max_cout = the largest Cout of such descriptors, 0 if there are none. */
int iunet_pack_desc_bytes(void);
int iunet_pack_batch(const void* descs, int n, int quant_max_cout, void* stream);

/* ---- forward kernels (replace the smp conv stack under unet.py:65-69) ----------------- */
/* 3^d conv, stride 1, pad 1, implicit GEMM on MFMA.  epi: 0 raw, 1 +bias, 2 +bias+ReLU.
 * stats (optional): fp32 [iunet_conv3_stats_parts(..., layout)][Cout][2] partial sum / sum of squares of
 * the raw output (BatchNorm batch statistics), reduced by the caller. */
int iunet_conv3_fwd(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride,
                    const void* wpk, const void* bias, void* stats, int N, int D, int H, int W, int Cin, int Cout,
                    int epi, int layout, void* stream);
/* which kernel structure / weight layout serves this launch best: 0 = 32-channel chunks, weights through
 * registers (wpk packed with mode bit 1 clear); 1 = persistent LDS-fed Cout-32 structure (mode bit 1 set);
 * 2 = weight-stationary variant of 1 (same mode-bit-1 operator; 3-D, Cin <= 64: all weights of a Cout tile stay
 * in LDS for the whole launch).  Layout 1 or 2 is mandatory when Cout is not a multiple of 64. */
int iunet_conv3_pick_layout(int nd, int N, int D, int H, int W, int Cin, int Cout);
/* 1 if a layout-2 launch of this shape walks its tiles in pairs (one weight stream per two tiles: 3-D, streamed weights, an even
 * number of tiles per workgroup), else 0.  Speed only: the results are the same bits either way.  Exposed for the tests. */
int iunet_conv3_tile_pairs(int nd, int N, int D, int H, int W, int Cin, int Cout);
/* layout 3 of iunet_conv3_fwd / _fwd_act: the layout-2 kernel on the COMPACT K16 operator (iunet_pack_conv3 mode bit 2; 3^3 filters:
 * Cout * Cin * 27 elements, the ninth filter column of two consecutive 16-channel chunks shares one k-slot) -- 10 % fewer matrix
 * instructions and weight bytes.  1 if this launch qualifies (3-D: Cin > 32, no fused BatchNorm-backward sums (bw), a fused
 * input activation (act) only up to 192 input channels; 2-D, 3^2 filters on the cross-pair step: every channel count, act or bw only
 * up to 64 input channels), else 0: use layout 2.  The answer does not depend on the grid: layouts
 * 2 and 3 add the taps in different orders, and a layer keeps one order whatever the number of blocks in a launch. */
int iunet_conv3_compact_ok(int nd, int N, int D, int H, int W, int Cin, int Cout, int act, int bw);
/* iunet_conv3_fwd whose input is relu(in_scale[c] * x + in_shift[c]) (fp32 [Cin] each): in training the BatchNorm + ReLU of
 * the previous conv is applied by the loader waves instead of a separate pass over HBM.  Layout 2 only. */
int iunet_conv3_fwd_act(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride,
                        const void* wpk, const void* bias, void* stats, const void* in_scale, const void* in_shift,
                        int N, int D, int H, int W, int Cin, int Cout, int epi, int layout, void* stream);
int iunet_conv3_num_tiles(int nd, int N, int D, int H, int W);
/* rows of partial sums a conv3_fwd launch writes: one per tile (layouts 0, 1) or one per workgroup (layout 2) */
int iunet_conv3_stats_parts(int nd, int N, int D, int H, int W, int Cout, int layout);
/* first conv reads the caller's tensor directly: in_dtype 0 f32, 1 f16, 2 u8 (x/255,
 * predict.py:30), 3 bf16; in_strides = element strides (n, c, d, h, w).  stats (optional): partial BatchNorm
 * sums [iunet_conv3_num_tiles][Cout][2], as for iunet_conv3_fwd. */
int iunet_first_conv_fwd(int dtype, int nd, const void* x, int in_dtype, const long long* in_strides, void* y,
                         long long y_sstride, const void* w, const void* bias, void* stats, int N, int D, int H, int W,
                         int Cin, int Cout, int relu, void* stream);
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

/* ---- fp32 parity mode -------------------------------------------------------------------------------------------------
 * BASELINE.json north_star: "outputs match the reference CPU PyTorch path within 1e-3 on logits (integer-exact on argmax
 * class map)".  The 16-bit paths above round every activation in HBM and cannot reach that (5e-3 .. 7e-2 measured); these
 * entry points run the same forward graph (unet.py:65-69 over the canonical network) with fp32 activations, fp32 weights
 * and the f32-input matrix instruction (exact fp32 products and sums), 1/16 of the bf16 rate.  Layout of THIS mode: planar
 * fp32, C planes of [D][H][W]; `*_ss` = elements between consecutive samples. */
/* operator packing: conv w fp32 [Cout][Cin][taps] (taps 9 / 27; 1 = pointwise) or, transposed != 0, ConvTranspose k2 s2 w fp32
 * [Cin][Cout][taps] (taps 4 / 8); a non-NULL gamma folds an eval-mode BatchNorm (w * gamma / sqrt(var + eps); bias_out =
 * beta - mean * that scale; every operation separately rounded, as the CPU oracle's fold).  dst:
 * iunet_f32_pack_conv_elems floats. */
long long iunet_f32_pack_conv_elems(int Cout, int Cin, int taps);
int iunet_f32_pack_conv(const void* w, void* dst, void* bias_out, const void* gamma, const void* beta, const void* mean,
                        const void* var, float eps, int Cout, int Cin, int taps, int transposed, void* stream);
/* 3^d conv pad 1 (transposed == 0), ConvTranspose k2 s2 (transposed == 1; D, H, W = input grid, output 2x) or 1x1 conv
 * (transposed == 2; operator packed with taps == 1) + bias + optional ReLU.  The input is read through element strides in_strides (n, c, d, h, w) and in_dtype (0 f32, 1 f16, 2 u8 / 255, 3
 * bf16), so the first conv takes the caller's tensor or a 2.5-D view of a block directly; y: planar fp32. */
int iunet_f32_conv_fwd(int nd, const void* x, int in_dtype, const long long* in_strides, void* y, long long y_ss,
                       const void* wpk, const void* bias, int N, int D, int H, int W, int Cin, int Cout, int relu,
                       int transposed, void* stream);
int iunet_f32_maxpool_fwd(int nd, const void* x, long long x_ss, void* y, long long y_ss, int C, int N, int Do, int Ho, int Wo,
                          void* stream);
/* iunet_head_fwd on planar fp32 features: w fp32 [ncls][C0]; same output contract. */
int iunet_f32_head_fwd(const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls, void* logits,
                       void* probs, void* cls, const long long* out_strides, float divisor, int accumulate, int N, int D, int H,
                       int W, void* stream);

/* ---- fp32 parity form of the TRAINING step (csrc/train_f32.hip; interactive_unet/train_engine_f32.py sequences it) -------------
 * unet.py:88-102 + autograd + AdamW with planar fp32 tensors and fp32 arithmetic throughout: convolutions, transposed convolutions
 * and every data gradient run on iunet_f32_conv_fwd (a conv's data gradient is the conv with the flipped, transposed operator; a
 * transposed conv's is the 1x1 conv -- transposed == 2, operator packed with taps == 1 -- over the space-to-depth view of dy); the
 * entry points below are the rest of the step.  A whole step differs from CPU autograd only by the order of the sums. */
/* BatchNorm batch statistics of y [N][C][vox]: mean, std = sqrt(biased var + eps) (two passes, double accumulation); run_mean /
 * run_var (or NULL both): running statistics updated with `momentum` and the unbiased variance */
int iunet_f32_bn_stats(const void* y, long long y_ss, int C, int N, long long vox, float eps, float momentum, void* mean, void* stdv,
                       void* run_mean, void* run_var, void* stream);
/* z = relu(((y - mean) / std) * gamma + beta) */
int iunet_f32_bn_relu_fwd(const void* y, long long y_ss, void* z, long long z_ss, const void* mean, const void* stdv, const void* gamma,
                          const void* beta, int C, int N, long long vox, void* stream);
/* backward of the pair: dz -> dy (through the ReLU mask and the batch-statistics BatchNorm), dgamma [C], dbeta [C] */
int iunet_f32_bn_relu_bwd(const void* dz, long long dz_ss, const void* y, long long y_ss, void* dy, long long dy_ss, const void* mean,
                          const void* stdv, const void* gamma, const void* beta, void* dgamma, void* dbeta, int C, int N, long long vox,
                          void* stream);
/* dz (+)= dpool at the first maximum of every 2^d window of z (Do, Ho, Wo = pooled grid) */
int iunet_f32_maxpool_bwd(int nd, const void* z, long long z_ss, const void* dpool, long long dp_ss, void* dz, long long dz_ss, int C,
                          int N, int Do, int Ho, int Wo, int accumulate, void* stream);
/* weight gradient on the f32-input MFMA: slab [iunet_f32_wgrad_splits][Cout][Cin][taps] partial sums of dy (x) tap-shifted x (taps 3^nd,
 * or 1: pointwise -- transposed-conv and head weight gradients); iunet_reduce_slab sums the rows in a fixed order */
int iunet_f32_wgrad_splits(int nd, int N, int D, int H, int W, int Cin, int Cout);
int iunet_f32_wgrad(int nd, const void* x, long long x_ss, const void* dy, long long dy_ss, void* slab, int N, int D, int H, int W,
                    int Cin, int Cout, int taps, void* stream);
/* iunet_head_loss_fwd / _bwd on planar fp32 features: out4 = [loss, dice, iou, mcc (rounded tensors, unet.py:75-86)], coef [ncls][3];
 * backward: dlogits [N][dl_ss / vox planes][vox] (the first ncls written; the head's dW = iunet_f32_wgrad taps 1, db =
 * iunet_f32_channel_sum) and dx [N][C0][vox] */
int iunet_f32_head_loss_num_parts(int N, long long vox);
int iunet_f32_head_loss_fwd(const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls, const void* target,
                            const void* weight, int tdtype, int kind, void* slab, void* out4, void* coef, int N, long long vox,
                            void* stream);
int iunet_f32_head_loss_bwd(const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls, const void* target,
                            const void* weight, int tdtype, const void* coef, void* dlogits, long long dl_ss, void* dx, long long dx_ss,
                            int N, long long vox, void* stream);
int iunet_f32_channel_sum(const void* t, long long t_ss, void* out, int C, int N, long long vox, void* stream);

/* ---- fp16x2 split precision: the tolerance-meeting forward on the 16-bit matrix cores ------------------------------------
 * BASELINE.json north_star's 1e-3 on logits / integer-exact class map against the fp32 reference predict (predict.py:30-35,
 * unet.py:65-69), at a third of the 16-bit matrix rate instead of the sixteenth of the f32-input instruction: every activation
 * and operator entry is two fp16 words, hi = f16(v) and lo = f16(v - hi) (22 bits); a product is x_hi w_hi + x_lo w_hi +
 * x_hi w_lo on v_mfma_f32_16x16x32_f16 into one fp32 accumulator.  Tensor layout of THIS mode: C/8 hi planes [D][H][W][8] of
 * fp16 + C/8 lo planes `*_lo` PLANES further on (halves of a concat buffer stay views); `*_ss` = fp16 elements between samples.
 * Activations are stored multiplied by a power of two act_scale; csrc/split16.hip states the scaling. */
/* operator preparation: w fp32 [Cout][Cin][taps] (transposed == 0; taps 9 / 27) or ConvTranspose w [Cin][Cout][taps]
 * (transposed != 0; taps 4 / 8), optional eval-mode BatchNorm fold (gamma..var, as iunet_f32_pack_conv) or the layer's own
 * bias_in -> wv: transposed 0: the VIRTUAL fp32 operator over 3 Cin input channels, [w_hi | w_hi | w_lo] per chunk of `chunk`
 * channels -- the step of the kernel that consumes it: 16 (3-D stage conv), 32 (2-D), Cin (first conv) -- (3 Cout Cin taps
 * floats; feed it to iunet_pack_conv3 mode iunet_x2_pack_mode(nd) / iunet_pack_first_conv with "Cin" = 3 Cin, dtype 0); transposed 2 (what
 * iunet_x2_convT_fwd takes; Cin % 32 == 0): both words once over 2 Cin channels, in chunks of iunet_x2_convT_kc(Cin) k-steps of 32
 * channels [chunk][hi | lo][k-step][32] (feed it to iunet_pack_convT with "Cin" = 2 Cin); oscale [Cout] = the power of two the
 * accumulator is multiplied by (act_out / (act_in * row scale)), bias_out [Cout] = act_out * bias. */
int iunet_x2_convT_kc(int Cin);
int iunet_x2_prep(const void* w, void* wv, void* oscale, void* bias_out, const void* gamma, const void* beta, const void* mean,
                  const void* var, const void* bias_in, float eps, float act_in, float act_out, int Cout, int Cin, int taps,
                  int transposed, int chunk, void* stream);
/* The same preparation for EVERY operator of a network in one launch per kernel (a prediction engine that follows a training loop re-prepares
 * ~17 operators per optimiser step: predict.py:30-35 behind trainer.py:56-63): `table` = n rows of X2PrepDesc in device memory
 * (csrc/x2_prep_desc.h: the arguments above per row + `row0`, the running sum of Cout), `rows` = the sum of Cout.  iunet_x2_prep_batch takes
 * the rows of iunet_x2_prep (kind = transposed, kc = chunk or iunet_x2_convT_kc), iunet_x2m_prep_batch those of iunet_x2m_prep_nd (kind 3,
 * taps 27 / 9); feed the `out` operators to iunet_pack_batch.  Same bits as the per-layer calls. */
int iunet_x2_prep_desc_bytes(void);
int iunet_x2_prep_batch(const void* table, int n, int rows, void* stream);
int iunet_x2m_prep_batch(const void* table, int n, int rows, void* stream);
/* first conv: the caller's tensor (strides / dtype as iunet_first_conv_fwd; u8 is x / 255 correctly rounded, predict.py:30),
 * multiplied by act_scale and split -> y = split(relu?(acc * oscale + bias)) */
int iunet_x2_first_conv_fwd(int nd, const void* x, int in_dtype, const long long* in_strides, void* y, long long y_sstride, int y_lo,
                            const void* w, const void* oscale, const void* bias, float act_scale, int N, int D, int H, int W, int Cin,
                            int Cout, int relu, void* stream);
/* iunet_pack_conv3 mode of the stage convs' virtual operator: 2 (padded K16 order) in 3-D; 6 (compact order) in 2-D, where the third
 * filter column of the two 16-channel halves of a 32-channel step shares one k-group (9 taps in 9 k-slots instead of 12) */
int iunet_x2_pack_mode(int nd);
/* stage conv 3^d pad 1 (epi as iunet_conv3_fwd); Cin = real input channels; wpk packed with mode iunet_x2_pack_mode(nd) */
int iunet_x2_conv3_fwd(int nd, const void* x, long long x_sstride, int x_lo, void* y, long long y_sstride, int y_lo, const void* wpk,
                       const void* oscale, const void* bias, int N, int D, int H, int W, int Cin, int Cout, int epi, void* stream);
/* the same stage conv with a range flag: sat = optional device int that a launch raises (atomicMax, no synchronisation) to 0x7bff when a
 * stored hi word saturated at +-65504 -- act_scale x |activation| left the fp16 range and the result is no longer within tolerance;
 * iunet_x2m_first_conv_fwd / iunet_x2m_convT_fwd take the same flag */
int iunet_x2_conv3_fwd_flag(int nd, const void* x, long long x_sstride, int x_lo, void* y, long long y_sstride, int y_lo, const void* wpk,
                            const void* oscale, const void* bias, int N, int D, int H, int W, int Cin, int Cout, int epi, void* sat, void* stream);
/* 2^d max-pool on hi + lo sums (the winner's word pair is copied: no rounding) */
int iunet_x2_maxpool_fwd(int nd, const void* x, long long x_ss, int x_lo, void* y, long long y_ss, int y_lo, int C, int N, int Do,
                         int Ho, int Wo, void* stream);
/* ConvTranspose k2 s2 (+ bias); D, H, W = input grid */
int iunet_x2_convT_fwd(int nd, const void* x, long long x_ss, int x_lo, void* y, long long y_ss, int y_lo, const void* wpk,
                       const void* oscale, const void* bias, int N, int D, int H, int W, int Cin, int Cout, void* stream);
/* iunet_head_fwd on split features ((hi + lo) / act_scale, exact), fp32 arithmetic of iunet_f32_head_fwd; same output contract */
int iunet_x2_head_fwd(const void* x, long long x_ss, int x_lo, int C0, const void* w, const void* bias, float act_scale, int ncls,
                      void* logits, void* probs, void* cls, const long long* out_strides, float divisor, int accumulate, int N, int D,
                      int H, int W, void* stream);

/* ---- fp16x2 with the cross terms on the fp8 matrix cores ("x2m", the stage convs; csrc/conv3_x2m.hip) ---------------------------
 * The two cross terms of a split product (x_lo w_hi, x_hi w_lo) are 2^-11 of it: they run as one K = 128 step of
 * v_mfma_f32_16x16x128_f8f6f4 over the virtual channels [x_lo8 | x_hi8] x [w_hi8 | w_lo8] (e4m3) into the accumulator of the main
 * term x_hi w_hi -- two matrix-step units per 16 input channels instead of three.  Beside its hi (and optional lo) planes a tensor
 * carries lo8 planes (the "m8" / x8 / y8 arguments): C / 16 planes [D][H][W][16 B] of e4m3((v - hi) * 2^4) per 16-channel chunk (v =
 * act_scale * activation): 3 bytes per element in HBM.  The other half of the fp8 step's operand, e4m3(hi * 2^-8), is a function of the hi
 * words: the conv's loader waves make it in LDS.  Same graph, same reference semantics (unet.py:65-69, predict.py:30-35). */
long long iunet_x2m_w8_bytes(int Cout, int Cin);
/* w fp32 [Cout][Cin][27] (+ optional BatchNorm fold) -> whi fp32 [Cout][Cin][27] = w_hi (feed it to iunet_pack_conv3, dtype 0, mode 2),
 * w8 = iunet_x2m_w8_bytes bytes (K128 order of [e4m3(w_hi 2^-4) | e4m3(w_lo 2^8)] per 16-channel chunk), oscale / bias_out as iunet_x2_prep */
int iunet_x2m_prep(const void* w, void* whi, void* w8, void* oscale, void* bias_out, const void* gamma, const void* beta, const void* mean,
                   const void* var, float eps, float act_in, float act_out, int Cout, int Cin, void* stream);
/* nd-generic forms (nd = 2: 3 x 3 filters on 16 x 32-pixel tiles, a step = 32 channels = two virtual blocks, four taps per K = 128
 * instruction + tap 8 on the K = 32 fp8 instruction; whi goes to iunet_pack_conv3 mode 6, the cross-pair order; nd = 3: as above) */
long long iunet_x2m_w8_bytes_nd(int nd, int Cout, int Cin);
int iunet_x2m_prep_nd(int nd, const void* w, void* whi, void* w8, void* oscale, void* bias_out, const void* gamma, const void* beta,
                      const void* mean, const void* var, float eps, float act_in, float act_out, int Cout, int Cin, void* stream);
int iunet_x2m_conv_fwd(int nd, const void* x, long long x_ss, const void* x8, long long x8_ss, void* y, long long y_ss, int y_lo, void* y8,
                       long long y8_ss, const void* w16, const void* w8, const void* oscale, const void* bias, int N, int D, int H, int W,
                       int Cin, int Cout, int epi, void* sat, void* stream);
/* an ENCODER stage's second conv (unet.py:63-69: the skip tensor) with the stage's 2^d max-pool riding in its epilogue: y / y8 as
 * iunet_x2m_conv_fwd, and py / py8 = hi planes (py_ss elements per sample) / lo8 planes (py8_ss bytes per sample) of the pooled tensor on the
 * grid D/2 (nd = 3), H/2, W/2 -- the words iunet_x2m_maxpool_fwd makes of y / y8, bit for bit, without reading them back (2-D: pooled in the
 * consumer waves' registers; 3-D: x and y in registers, the z pair through LDS by the loader waves).  iunet_x2m_pool_fusable: 1 where the
 * library's own callers fuse a conv of C channels (3-D and 2-D; IUNET_X2M_POOL=0: nowhere, =3: 3-D only, =4: 2-D without the 64-channel
 * stage, the policy of the 4-byte format). */
int iunet_x2m_pool_fusable(int nd, int C);
int iunet_x2m_conv_pool_fwd(int nd, const void* x, long long x_ss, const void* x8, long long x8_ss, void* y, long long y_ss, int y_lo, void* y8,
                            long long y8_ss, void* py, long long py_ss, void* py8, long long py8_ss, const void* w16, const void* w8,
                            const void* oscale, const void* bias, int N, int D, int H, int W, int Cin, int Cout, int epi, void* sat, void* stream);
/* The FIRST ENCODER STAGE of the 2-D network as one launch (unet.py:63-69: conv 1 -> 32 + BatchNorm + ReLU, conv 32 -> 32 + BatchNorm + ReLU): the
 * second conv (operators w16 / w8 / oscale / bias of iunet_x2m_conv_fwd) whose loader waves compute the first conv on the way in from the
 * caller's one-channel image (x, in_dtype, in_strides; fw / f_oscale / f_bias / act_scale: what iunet_x2m_first_conv_fwd takes) -- the
 * 32-channel tensor between the two convs never exists in HBM.  y / y8 (and, with py != NULL, the pooled py / py8 of
 * iunet_x2m_conv_pool_fwd) hold iunet_x2m_first_conv_fwd + iunet_x2m_conv_fwd (+ pool) bit for bit.  iunet_x2m_first_stage_fusable: 1 where
 * the library's own callers use it (2-D, one input channel, 32 channels at level 0, a batch of >= 2 048 tiles of 16 x 32 pixels: the loader
 * waves' first conv is the longer side of a tile step and the first tile's hides behind nothing -- and a stage whose pool does NOT ride in
 * the second conv, iunet_x2m_pool_fusable: with first conv AND pool on the loader waves the launch measured slower than first conv + pooled
 * conv; IUNET_X2M_FIRST=0: never, =2: always). */
int iunet_x2m_first_stage_fusable(int nd, int cin, int c0, int N, int H, int W);
int iunet_x2m_first_stage_fwd(const void* x, int in_dtype, const long long* in_strides, const void* fw, const void* f_oscale, const void* f_bias,
                              float act_scale, void* y, long long y_ss, int y_lo, void* y8, long long y8_ss, void* py, long long py_ss, void* py8,
                              long long py8_ss, const void* w16, const void* w8, const void* oscale, const void* bias, int N, int H, int W,
                              void* sat, void* stream);
/* the LAST stage conv of the network with the 1x1 head + softmax + class map (unet.py:63-69, predict.py:38) in its epilogue: its 32 output
 * channels are never written; logits / probs / cls are iunet_x2m_conv_fwd + iunet_x2_head_fwd bit for bit (the head's fmaf chain is walked
 * through the lane groups in channel order).  iunet_x2m_head_fusable: 1 for the heads it takes (2 or 3 classes on 32 channels). */
int iunet_x2m_head_fusable(int ncls, int C0);
int iunet_x2m_conv_head_fwd(int nd, const void* x, long long x_ss, const void* x8, long long x8_ss, const void* w16, const void* w8,
                            const void* oscale, const void* bias, const void* head_w, const void* head_b, float act_scale, int ncls,
                            void* logits, void* probs, void* cls, const long long* out_strides, float divisor, int accumulate, int N, int D,
                            int H, int W, int Cin, void* sat, void* stream);
/* lo8 planes (x8_ss bytes per sample) of a split tensor that another kernel wrote as hi + lo words (x_ss fp16 elements per sample) */
int iunet_x2m_make8(const void* x, long long x_ss, int x_lo, void* x8, long long x8_ss, int C, int N, int D, int H, int W, void* stream);
/* producers of the x2m form: iunet_x2_first_conv_fwd / iunet_x2_convT_fwd writing, beside the hi planes, the lo8 planes of their output
 * (y8, bytes per sample; null: none) and the lo planes only when y_lo >= 0; the max-pool on (hi, lo8): the larger hi + lo8 / 16 wins, its hi
 * word and lo8 byte are copied (Do, Ho, Wo = output grid) */
int iunet_x2m_first_conv_fwd(int nd, const void* x, int in_dtype, const long long* in_strides, void* y, long long y_sstride, int y_lo,
                             void* y8, long long y8_sstride, const void* w, const void* oscale, const void* bias, float act_scale, int N,
                             int D, int H, int W, int Cin, int Cout, int relu, void* sat, void* stream);
int iunet_x2m_convT_fwd(int nd, const void* x, long long x_ss, int x_lo, void* y, long long y_ss, int y_lo, void* y8, long long y8_ss,
                        const void* wpk, const void* oscale, const void* bias, int N, int D, int H, int W, int Cin, int Cout, void* sat, void* stream);
int iunet_x2m_maxpool_fwd(int nd, const void* x, long long x_ss, const void* x8, long long x8_ss, void* y, long long y_ss, void* y8,
                          long long y8_ss, int C, int N, int Do, int Ho, int Wo, void* stream);
/* the stage conv: x = Cin / 8 hi planes + x8 = its lo8 planes; y = Cout / 8 hi planes (+ lo planes y_lo planes further on unless
 * y_lo < 0) + y8 = its lo8 planes (or null); sat: optional device int raised to the bit pattern of a saturated (|v| >= 65504) hi word */
int iunet_x2m_conv3_fwd(const void* x, long long x_ss, const void* x8, long long x8_ss, void* y, long long y_ss, int y_lo, void* y8,
                        long long y8_ss, const void* w16, const void* w8, const void* oscale, const void* bias, int N, int D, int H, int W,
                        int Cin, int Cout, int epi, void* sat, void* stream);

/* ---- handle level: the whole forward as one call (csrc/net.hip) ---------------------------------------------------------
 * For a caller that is not Python: the launch graph interactive_unet/engine.py / engine_x2.py sequence (unet.py:65-69 over the
 * canonical network) sequenced in C++.  The handle is host memory; every device buffer is the caller's.
 *   iunet_net* net; iunet_net_create(2, 4, 32, 1, 2, 2, 0.f, &net);             // 2-D, 4 levels, base 32, 1 -> 2 classes, fp16x2
 *   flat = device floats [iunet_net_num_params(net)] filled tensor by tensor (iunet_net_param gives name / offset / count)
 *   packed = device bytes [iunet_net_packed_bytes(net)];  iunet_net_load(net, flat, packed, stream);
 *   ws = device bytes [iunet_net_workspace_bytes(net, N, 1, H, W)];  iunet_net_forward_argmax(net, x_u8, cls_u8, N, 1, H, W, ws, stream);
 * mode: 0 fp16, 1 bf16 (16-bit activations), 2 fp16x2 (split precision: logits within 1e-3 of the fp32 predict), 3 fp16x2 with
 * the cross terms of the stage convs on the fp8 matrix cores (the x2m entry points above: two matrix-step units per 16 input channels
 * instead of three); modes 2 / 3: the first 4 bytes of the workspace are an int the forward raises to 0x7bff when a stored activation word
 * saturates at 65504 -- zero it once, read it when convenient; act_scale: a power of two, modes 2 / 3 only (0 = default 64). */
typedef struct iunet_net iunet_net;
int iunet_net_create(int dim, int levels, int base, int cin, int ncls, int mode, float act_scale, iunet_net** out);
/* the same with the normalisation named: norm 0 = BatchNorm (eval-mode statistics folded into the operators), 1 = GroupNorm(groups) + ReLU
 * after every stage conv (north star "GroupNorm/BN") -- mode 2 only, the split-precision form GroupNorm networks predict in (nothing
 * folds: a stage conv writes its raw output, iunet_x2_gn_relu_fwd normalises it with per-(sample, group) statistics taken in double; the
 * bn{j}.weight / .bias tensors of the flat vector are the affine pair, the running statistics are unused) */
int iunet_net_create_ex(int dim, int levels, int base, int cin, int ncls, int mode, float act_scale, int norm, int groups, iunet_net** out);
void iunet_net_destroy(iunet_net* net);
/* the flat fp32 parameter vector: trainable tensors and BatchNorm running statistics in the canonical order (the state_dict keys of
 * interactive_unet/unet.py: enc{l}.conv{j}.weight [Cout][Cin][3^d], enc{l}.bn{j}.{weight,bias,running_mean,running_var},
 * dec{l}.up.weight [Cin][Cout][2^d], dec{l}.up.bias, ..., head.weight [ncls][base], head.bias) */
long long iunet_net_num_params(const iunet_net* net);
int iunet_net_num_tensors(const iunet_net* net);
int iunet_net_param(const iunet_net* net, int index, char* name, int name_cap, long long* offset, long long* numel);
long long iunet_net_packed_bytes(const iunet_net* net);
/* fold eval-mode BatchNorm, scale / split (mode 2), reorder: flat_params -> packed; both stay the caller's and must outlive the forwards */
int iunet_net_load(iunet_net* net, const void* flat_params, void* packed, void* stream);
long long iunet_net_workspace_bytes(const iunet_net* net, int N, int D, int H, int W);      /* 0 = bad shape */
/* x: the caller's tensor (in_dtype 0 f32, 1 f16, 2 u8 / 255, 3 bf16; element strides n, c, d, h, w) -> logits / probs (fp32, element
 * strides out_strides; probs = ((accumulate ? probs : 0) + p) / divisor, predict.py:101-110) and / or cls uint8 [N][D*H*W] */
int iunet_net_forward(iunet_net* net, const void* x, int in_dtype, const long long* in_strides, int N, int D, int H, int W, void* workspace,
                      void* logits, void* probs, void* cls, const long long* out_strides, float divisor, int accumulate, void* stream);
/* predict.py:30-38: uint8 [N][cin][D][H][W] -> class map uint8 [N][D*H*W] */
int iunet_net_forward_argmax(iunet_net* net, const void* x_u8, void* cls_u8, int N, int D, int H, int W, void* workspace, void* stream);
/* validation_step (unet.py:104-116) as one call, 16-bit modes (0 / 1): eval-mode forward (running statistics folded by iunet_net_load) +
 * fused head + softmax + loss / metrics on target / weight [N][ncls][D*H*W] (tdtype 0 f32, 1 f16; weight may be NULL; loss_kind 0 ce, 1
 * dice, 2 iou, 3 mcc, 4 dice_ce, 5 iou_ce, 6 mcc_ce) -> out4 = [Loss, Dice, IoU, MCC] (4 device floats).  scratch:
 * iunet_net_eval_scratch_bytes device bytes. */
long long iunet_net_eval_scratch_bytes(const iunet_net* net, int N, int D, int H, int W);
int iunet_net_eval_step(iunet_net* net, const void* x, int in_dtype, const long long* in_strides, const void* target, const void* weight,
                        int tdtype, int loss_kind, int N, int D, int H, int W, void* workspace, void* scratch, void* out4, void* stream);

/* ---- fp8 matrix cores: BASELINE config C5 ("fp8 weights / bf16 activations on CDNA4 fp8 MFMA") ------------------------
 * The stage convolutions of the forward pass (unet.py:65-69 over the canonical network) with the operator stored as OCP
 * e4m3 BYTES (half the weight bytes in HBM and LDS) and the product on v_mfma_f32_16x16x32_fp8_fp8; the 16-bit activations
 * in HBM are rounded to e4m3 (saturating at 448) on their way into LDS -- gfx950 has no mixed fp8 x bf16 MFMA. */
/* w fp32 [Cout][Cin][taps] (x an optional eval-mode BatchNorm fold, as iunet_f32_pack_conv) -> dst: iunet_f8_pack_conv3_bytes
 * bytes (K16 fragment order of iunet_conv3_pick_layout's layouts 1 / 2, one byte per element), wscale fp32 [Cout]: the
 * power-of-two scale 2^k, k minimal with max |w'| / 2^k <= 448, each weight = e4m3(w' / scale); bias_out fp32 [Cout]. */
/* Operator order of a layer's e4m3 bytes: 0 = K16 ([cob32][chunk16][column pair][dy][2][64 lanes][8 B], v_mfma_f32_16x16x32_fp8_fp8),
 * 1 = K128 (3-D layers with Cin % 32 == 0: per (32 Cout, 32 Cin) block [group 2][dy 3][m 2][half 2][64 lanes][16 B] +
 * [dy 3][m 2][64][8 B] for the ninth filter column, 27 648 B; v_mfma_f32_16x16x128_f8f6f4, csrc/conv3_f8k.hip).  A function of the
 * layer only; both pack entry points and iunet_conv3_f8_fwd follow it. */
int iunet_f8_pack_order(int taps, int Cin);
long long iunet_f8_pack_conv3_bytes(int Cout, int Cin, int taps);
int iunet_f8_pack_conv3(const void* w, const void* gamma, const void* beta, const void* mean, const void* var, float eps,
                        void* dst, void* wscale, void* bias_out, int Cout, int Cin, int taps, void* stream);
/* y = epilogue(wscale[c] * sum e4m3(x) * w8 + bias[c]); x, y: NHWC8c of `dtype`; epi as iunet_conv3_fwd.
 * workspace: iunet_conv3_f8_workspace_elems floats of scratch or NULL.  With a workspace, launches whose grid would leave most
 * of the chip idle behind a long loop over the input channels (the 16^3 / 8^3 levels of C5) are split along Cin: each share
 * is a workgroup of its own writing fp32 partial sums there, and a fixed-order reduction applies scale, bias and ReLU. */
long long iunet_conv3_f8_workspace_elems(int nd, int N, int D, int H, int W, int Cin, int Cout);
int iunet_conv3_f8_fwd(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride, const void* wpk,
                       const void* wscale, const void* bias, int N, int D, int H, int W, int Cin, int Cout, int epi,
                       void* workspace, void* stream);
/* e4m3 ACTIVATION PLANES between the layers of the fp8 network (format 1: C / 16 planes of [D][H][W][16 bytes]; sample strides in
 * BYTES; format 0 = the 16-bit NHWC8c planes of every other entry point).  The K = 128 convolution (iunet_f8_pack_order == 1) reads
 * them by LDS-DMA: half the bytes of the 16-bit tensor cross the L2 -> CU fabric, no conversion work in the loader waves.  Every
 * producer rounds its 16-bit result once more to e4m3 -- the rounding the consumer conv's loader applies to a 16-bit tensor -- so the
 * network's values do not depend on the format of the tensors in between (tests/test_gpu_f8.py).  Formats other than 0 are refused
 * (IUNET_ERR_UNSUPPORTED) for layers on the K16 path.  max-pool: rounding is monotonic, the pooled bytes are the rounded maximum. */
int iunet_conv3_f8_fwd_q(int dtype, int nd, const void* x, long long x_sstride, int x_fmt, void* y, long long y_sstride, int y_fmt,
                         const void* wpk, const void* wscale, const void* bias, int N, int D, int H, int W, int Cin, int Cout, int epi,
                         void* workspace, void* stream);
int iunet_first_conv_fwd_q(int dtype, int nd, const void* x, int in_dtype, const long long* in_strides, void* y,
                           long long y_sstride_bytes, const void* w, const void* bias, int N, int D, int H, int W,
                           int Cin, int Cout, int relu, void* stream);
int iunet_maxpool_q_fwd(int nd, const void* x, long long x_ss_bytes, void* y, long long y_ss_bytes, int C, int N, int Do, int Ho, int Wo,
                        void* stream);
int iunet_convT_fwd_q(int dtype, int nd, const void* x, long long x_ss, void* y, long long y_ss_bytes, const void* wpk,
                      const void* bias, int N, int D, int H, int W, int Cin, int Cout, void* stream);

/* ---- GroupNorm(groups) + ReLU of the tolerance-meeting prediction modes (csrc/gn_precise.hip; north star "GroupNorm/BN": the reference's
 * normalisation layers live in smp behind unet.py:33-61).  Statistics per (sample, group) in double, nothing folds into the operators: a
 * stage conv writes its raw output, then y = relu((x - mean) * rstd * gamma + beta).  slab: iunet_gn_precise_slab_bytes(N, C, vox) bytes
 * of scratch; scale / shift: fp32 [N][C] scratch (receive the per-sample affine pair). */
long long iunet_gn_precise_slab_bytes(int N, int C, long long vox);
/* fp32 mode: planar fp32 [N][C][vox], sample strides in elements (y may be a half of a concat buffer). */
int iunet_f32_gn_relu_fwd(const void* x, long long x_ss, void* y, long long y_ss, const void* gamma, const void* beta, int groups, float eps,
                          void* slab, void* scale, void* shift, int C, int N, long long vox, void* stream);
/* split precision (fp16x2): C / 8 hi planes + lo planes x_lo / y_lo planes further on, values act_scale x activation; sat: optional
 * range flag (an int raised to 0x7bff when a stored word saturates). */
int iunet_x2_gn_relu_fwd(const void* x, long long x_ss, int x_lo, void* y, long long y_ss, int y_lo, const void* gamma, const void* beta,
                         int groups, float eps, float act_scale, void* slab, void* scale, void* shift, int C, int N, long long vox,
                         void* sat, void* stream);
/* the same with the output in the x2m form (hi planes at y, lo8 planes at y8 -- y8_ss BYTES per sample --, lo planes only where y_lo >= 0):
 * the GroupNorm network in the default prediction mode's fast form (the stage convs: iunet_x2m_conv_fwd with epilogue 0 into hi + lo planes) */
int iunet_x2m_gn_relu_fwd(const void* x, long long x_ss, int x_lo, void* y, long long y_ss, int y_lo, void* y8, long long y8_ss, const void* gamma,
                          const void* beta, int groups, float eps, float act_scale, void* slab, void* scale, void* shift, int C, int N, long long vox,
                          void* sat, void* stream);

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
/* class map -> colours (predict.py:41-45 + utils.py:351-357): out_rgb uint8 [n][3] = palette[cls[i]], palette uint8
 * [ncls][3] on the device, classes >= ncls black. */
int iunet_colorize(const void* cls, long long n, const void* palette, int ncls, void* out_rgb, void* stream);
/* Calibration of the default prediction mode (the reference predicts in fp32, predict.py:30-35; north star: logits within 1e-3):
 * out2[0] = max |a - b|, out2[1] = max |a| of two fp32 logit tensors of n elements (device floats, overwritten; NaN if either
 * holds one).  interactive_unet/engine_auto.py runs one tile through the x2m and the fp16x2 forward and keeps x2m iff
 * out2[0] <= its threshold (4e-4). */
int iunet_logit_diff(const void* a, const void* b, long long n, void* out2, void* stream);

/* ---- oblique slices (slicer.py:94-115, :196-228; SURVEY 8f "Slicer on device") ----------- */
/* out[i][j] (uint8 [sw][sw]) = map_coordinates(vol[lo : lo + len], origin + a * r_i + b * r_j - lo, order, mode
 * 'constant'), r_k = start + k; bit-exact with scipy for orders 0 and 1.  geom: 9 host doubles a[3], b[3], origin[3];
 * lo, len: host ints [3] (the reference's bounding-box crop). */
int iunet_slice_gather(const void* vol, int Z, int Y, int X, const double* geom, const int* lo, const int* len, int sw,
                       int start, int order, void* out, void* stream);
/* Slicer.update_volume (slicer.py:230-257): vol[round-half-even(origin + a * r_i + b * r_j), clipped to the volume] =
 * data[i][j] for the sw x sw pixels in row-major order -- where several pixels land on one voxel the last one wins, as in
 * numpy's fancy-index assignment (resolved on the device through an owner table, so the result is deterministic).
 * vol: uint8 [Z][Y][X][C] (C = 1: plain volume), data: uint8 [sw][sw][C], both on the device; geom as above; workspace:
 * iunet_slice_scatter_workspace_bytes(sw) bytes of device memory; sw <= 1024. */
long long iunet_slice_scatter_workspace_bytes(int sw);
int iunet_slice_scatter(void* vol, int Z, int Y, int X, int C, const double* geom, int sw, int start, const void* data,
                        void* workspace, void* stream);

/* ---- multiscale pyramid (utils.py:29-77 resize_volume / add_multiscales; SURVEY 8f "Zarr block I/O ... 0.5x nearest
 * multiscale pyramid on device") ----------------------------------------------------------------------------------- */
/* scipy.ndimage.zoom(x, zoom, order=0) index arithmetic for one axis (host only, no GPU): n_out = round-half-even(n_in *
 * zoom); table[o] = floor(o * z + 0.5), z = (n_in - 1) / (n_out - 1) in double (1 if n_out == 1), or -1 where scipy's
 * mode='constant' yields 0 (o * z > n_in - 1: the last sample of some sizes, e.g. 32 or 384 at zoom 0.5). */
int iunet_zoom_nearest_len(int n_in, double zoom);
int iunet_zoom_nearest_table(int n_in, double zoom, int* table, int n_out);
/* dst uint8 [d0][d1][d2][d3] = src[t0[o0]][t1[o1]][t2[o2]][t3[o3]] (0 where any index is -1); src_dims, src_strides,
 * dst_strides, dst_dims: 4 host values each, strides in bytes (source and destination may be blocks of larger volumes;
 * the two inner destination axes must be contiguous); tables: device int32, the four tables concatenated (d0 + d1 + d2 +
 * d3 entries, every entry -1 or < src_dims of its axis).  3-D volumes pass 1 for the fourth axis.  One call = one
 * `dst[block] = ndimage.zoom(src[block], zoom, order=0)` of resize_volume (utils.py:46), or -- with tables that
 * concatenate the blocks' tables, which is what the host side does -- the whole block loop of utils.py:33-48. */
int iunet_zoom_nearest_u8(const void* src, const int* src_dims, const long long* src_strides, void* dst,
                          const long long* dst_strides, const int* dst_dims, const int* tables, void* stream);

/* ---- batch producer (loader.py:28-44, :125-154; SURVEY 8f "Batch producer") ------------------------------------------ */
/* One launch = one batch of UNetDataset.__getitem__ results: for sample b, X[b] = image, y[b] = mask, w[b] = weight of the
 * annotation after hflip, vflip, rotation (nearest, zero padding), crop + nearest resize to OH x OW, as fp16 of
 * float32(uint8 / 255), mask and weight zero where image channel 0 is 0 (loader.py:40-42), weight repeated over the C
 * classes.  descs: device array of B descriptors (struct AugDesc in csrc/augment.hip, mirrored by
 * interactive_unet/loader.py; iunet_augment_desc_bytes() = its size): source pointers (uint8 [H][W][ch], [H][W][C],
 * [H][W]), the rotation's base grids and rescaled matrix, flip flags, crop box.  lut_f16: 256 fp16 values on the device.
 * X [B][ch][OH][OW], y and w [B][C][OH][OW] fp16.  The index arithmetic is torchvision's (v2 rotate / resized_crop on
 * CPU tensors), see oracle/loader_ref.py. */
int iunet_augment_desc_bytes(void);
int iunet_augment_batch(const void* descs, int B, int ch, int C, int OH, int OW, const void* lut_f16, void* X, void* y, void* w,
                        void* stream);

/* ---- training step (replaces autograd + AMP + AdamW under unet.py:71-102, trainer.py:56-63) -- */
/* BatchNorm batch statistics: slab = partial (sum, sumsq) [nparts][C][2] written by the conv
 * epilogues -> per-channel scale/shift (gamma*invstd, beta-mean*scale), mean, invstd; updates
 * running_mean / running_var (momentum, unbiased variance) when they are not NULL. */
int iunet_bn_finalize(const void* slab, int nparts, int C, double count, const void* gamma, const void* beta,
                      void* running_mean, void* running_var, float momentum, float eps, void* scale, void* shift,
                      void* mean, void* invstd, void* stream);
int iunet_bn_relu_fwd(int dtype, const void* y, long long y_ss, void* z, long long z_ss, const void* scale,
                      const void* shift, int C, int N, long long vox, void* stream);
/* the same plus the 2^d max-pool of z in one pass (encoder stages); (Do, Ho, Wo) is the pooled grid. */
int iunet_bn_relu_pool_fwd(int dtype, int nd, const void* y, long long y_ss, void* z, long long z_ss, void* pooled,
                           long long p_ss, const void* scale, const void* shift, int C, int N, int Do, int Ho, int Wo,
                           void* stream);
/* backward of z = relu(bn(y)): dy, dgamma, dbeta from dz, y (and z; z may be NULL: the ReLU mask is then
 * recomputed from y with scale / shift, one tensor read less per pass).  slab: iunet_bn_bwd_num_parts*C*2
 * floats, coef: 3*C floats of scratch. */
int iunet_bn_bwd_num_parts(int N, long long vox);
int iunet_bn_relu_bwd(int dtype, const void* dz, long long dz_ss, const void* z, long long z_ss, const void* y,
                      long long y_ss, void* dy, long long dy_ss, const void* mean, const void* invstd, const void* gamma,
                      const void* scale, const void* shift, void* dgamma, void* dbeta, void* slab, void* coef, int C, int N,
                      long long vox, void* stream);
/* The first pass of iunet_bn_relu_bwd folded into the data-gradient launch that produces dz.  iunet_conv3_dgrad_bnstats =
 * iunet_conv3_fwd (layout 2, epi 0) on the data-gradient operator, whose epilogue also reads yp (the raw output of the layer the
 * gradient flows into, z = relu(bn(yp))) and writes per-workgroup rows of (sum dz', sum dz' * xhat) to stats
 * [iunet_conv3_stats_parts(nd, N, D, H, W, Cout, 2)][Cout][2]; iunet_bn_relu_bwd_apply finalizes those rows (dgamma, dbeta,
 * coefficients) and runs pass 2 (dy NULL: coefficients only, for iunet_first_conv_wgrad_bn). */
int iunet_conv3_dgrad_bnstats(int dtype, int nd, const void* dy, long long dy_sstride, void* dz, long long dz_sstride,
                              const void* wpk, void* stats, const void* yp, long long yp_sstride, const void* mean,
                              const void* invstd, const void* scale, const void* shift, int N, int D, int H, int W, int Cin,
                              int Cout, void* stream);
/* ... on a named layout: 2 (the K16 operator, as above) or 3 (the compact operator: 2-D, up to 64 input channels --
 * iunet_conv3_compact_ok(.., act 0, bw 1) says whether a launch has it); the rows and their order are the layout-2 ones. */
int iunet_conv3_dgrad_bnstats_lay(int dtype, int nd, const void* dy, long long dy_sstride, void* dz, long long dz_sstride,
                                  const void* wpk, void* stats, const void* yp, long long yp_sstride, const void* mean,
                                  const void* invstd, const void* scale, const void* shift, int N, int D, int H, int W, int Cin,
                                  int Cout, int layout, void* stream);
int iunet_bn_relu_bwd_apply(int dtype, const void* dz, long long dz_ss, const void* y, long long y_ss, void* dy, long long dy_ss,
                            const void* mean, const void* invstd, const void* gamma, const void* scale, const void* shift,
                            void* dgamma, void* dbeta, const void* slab, int nparts, void* coef, int C, int N, long long vox,
                            void* stream);
/* GroupNorm + ReLU (north_star "GroupNorm/BN"; the GroupNorm(8) variant of SURVEY 8d's canonical stage): statistics per
 * (sample, group), identical at training and inference -- nothing folds into the conv.  z = relu(group_norm(y)); slab:
 * iunet_gn_num_parts(N, vox) * C * 2 floats of scratch; scale / shift / mean / invstd: fp32 [N][C], written by the forward
 * and read by the backward; coef: N * C * 3 floats of scratch; C <= 1024 in the backward. */
int iunet_gn_num_parts(int N, long long vox);
int iunet_gn_relu_fwd(int dtype, const void* y, long long y_ss, void* z, long long z_ss, const void* gamma, const void* beta,
                      int groups, float eps, void* slab, void* scale, void* shift, void* mean, void* invstd, int C, int N,
                      long long vox, void* stream);
/* GroupNorm statistics of ONE sample from the statistics epilogue of the convolutions (stats = [nparts][C][2] partial (sum, sum of
 * squares) of a launch with N = 1: iunet_conv3_fwd / iunet_first_conv_fwd, nparts = iunet_conv3_stats_parts / iunet_conv3_num_tiles)
 * -> scale / shift / mean / invstd [C] of that sample, applied by iunet_conv3_fwd_act (in the next conv's loader waves) or by
 * iunet_bn_relu_fwd / iunet_bn_relu_pool_fwd: the fused form of the inference GroupNorm (no statistics pass over the tensor). */
int iunet_gn_finalize(const void* stats, int nparts, int C, int groups, long long vox, const void* gamma, const void* beta, float eps,
                      void* scale, void* shift, void* mean, void* invstd, void* stream);
int iunet_gn_relu_bwd(int dtype, const void* dz, long long dz_ss, const void* y, long long y_ss, void* dy, long long dy_ss,
                      const void* gamma, int groups, const void* scale, const void* shift, const void* mean, const void* invstd,
                      void* dgamma, void* dbeta, void* slab, void* coef, int C, int N, long long vox, void* stream);
/* the same with the 2^d max-pool of an encoder stage folded in: the forward's normalise pass writes z AND its max-pool; the backward forms
 * dz = dskip + route(dpool) on the fly in both of its passes (the GroupNorm forms of iunet_bn_relu_pool_fwd / _bwd) */
int iunet_gn_relu_pool_fwd(int dtype, int nd, const void* y, long long y_ss, void* z, long long z_ss, void* pooled, long long p_ss,
                           const void* gamma, const void* beta, int groups, float eps, void* slab, void* scale, void* shift, void* mean,
                           void* invstd, int C, int N, int Do, int Ho, int Wo, void* stream);
/* GroupNorm training without the statistics pass: the conv that produces y writes per-SAMPLE partial sums from its epilogue.
 * iunet_conv3_sample_stats_rows: rows per sample of such a launch on this grid and layout (2 or 3), 0 = not available (layouts 0 / 1,
 * fewer than 8 bricks per sample): run iunet_gn_relu_fwd.  iunet_conv3_fwd_sample_stats = iunet_conv3_fwd (epi 0, no bias) + stats
 * [N][rows][Cout][2] = (sum, sum of squares) of the fp32 accumulators; its brick schedule is one sample's, walked once per sample.
 * iunet_gn_relu_fwd_rows / _pool_fwd_rows: iunet_gn_relu_fwd / _pool_fwd on that slab (rows > 0; rows = 0: their own statistics pass). */
int iunet_conv3_sample_stats_rows(int dtype, int nd, int N, int D, int H, int W, int Cin, int Cout, int layout);
int iunet_conv3_fwd_sample_stats(int dtype, int nd, const void* x, long long x_sstride, void* y, long long y_sstride, const void* wpk,
                                 void* stats, int N, int D, int H, int W, int Cin, int Cout, int layout, void* stream);
int iunet_gn_relu_fwd_rows(int dtype, const void* y, long long y_ss, void* z, long long z_ss, const void* gamma, const void* beta,
                           int groups, float eps, void* slab, int rows, void* scale, void* shift, void* mean, void* invstd, int C, int N,
                           long long vox, void* stream);
int iunet_gn_relu_pool_fwd_rows(int dtype, int nd, const void* y, long long y_ss, void* z, long long z_ss, void* pooled, long long p_ss,
                                const void* gamma, const void* beta, int groups, float eps, void* slab, int rows, void* scale, void* shift,
                                void* mean, void* invstd, int C, int N, int Do, int Ho, int Wo, void* stream);
/* ... and the backward's first pass the same way: iunet_conv3_dgrad_sample_bnstats = iunet_conv3_dgrad_bnstats_lay with per-sample
 * parameter rows ([N][Cout] each) and per-sample sums, stats [N][rows][Cout][2] (rows from iunet_conv3_sample_stats_rows on the launch's
 * own Cin / Cout); iunet_gn_relu_bwd_rows = iunet_gn_relu_bwd on that slab (rows > 0: no reduction pass over dz and y). */
int iunet_conv3_dgrad_sample_bnstats(int dtype, int nd, const void* dy, long long dy_sstride, void* dz, long long dz_sstride, const void* wpk,
                                     void* stats, const void* yp, long long yp_sstride, const void* mean, const void* invstd, const void* scale,
                                     const void* shift, int N, int D, int H, int W, int Cin, int Cout, int layout, void* stream);
int iunet_gn_relu_bwd_rows(int dtype, const void* dz, long long dz_ss, const void* y, long long y_ss, void* dy, long long dy_ss,
                           const void* gamma, int groups, const void* scale, const void* shift, const void* mean, const void* invstd,
                           void* dgamma, void* dbeta, void* slab, int rows, void* coef, int C, int N, long long vox, void* stream);
int iunet_gn_relu_pool_bwd(int dtype, int nd, const void* dskip, long long ds_ss, const void* dpool, long long dp_ss, const void* y,
                           long long y_ss, void* dy, long long dy_ss, const void* gamma, int groups, const void* scale, const void* shift,
                           const void* mean, const void* invstd, void* dgamma, void* dbeta, void* slab, void* coef, int C, int N, int Do,
                           int Ho, int Wo, void* stream);
/* iunet_maxpool_bwd (add_skip) + iunet_bn_relu_bwd of an encoder stage's second conv without materialising the gradient of
 * the stage output: dz = dskip + route(dpool), routed to the first maximum of each 2^d window of relu(bn(y)) (recomputed).
 * (Do, Ho, Wo) = pooled grid; dskip, y, dy on the 2x grid; slab as for iunet_bn_relu_bwd on the 2x grid. */
int iunet_bn_relu_pool_bwd(int dtype, int nd, const void* dskip, long long ds_ss, const void* dpool, long long dp_ss,
                           const void* y, long long y_ss, void* dy, long long dy_ss, const void* mean, const void* invstd,
                           const void* gamma, const void* scale, const void* shift, void* dgamma, void* dbeta, void* slab,
                           void* coef, int C, int N, int Do, int Ho, int Wo, void* stream);
/* dz = (add_skip ? dz : 0) + max-pool backward of dpool (first maximum gets the gradient), in place. */
int iunet_maxpool_bwd(int dtype, int nd, const void* z, long long z_ss, const void* dpool, long long dp_ss, void* dz,
                      long long dz_ss, int add_skip, int C, int N, int Do, int Ho, int Wo, void* stream);
/* fused 1x1 head + softmax + weighted soft-confusion loss of metrics.py:3-187 with axes = batch + spatial
 * (unet.py:98).  kind: 0 CE, 1 Dice, 2 IoU, 3 MCC, 4 Dice+CE, 5 IoU+CE, 6 MCC+CE (utils.py:458-475).
 * target / weight: [N][ncls][vox], tdtype 0 f32 / 1 f16; weight may be NULL.
 * out4 = {loss, Dice, IoU, MCC on rounded tensors (unet.py:75-86)}; coef = [ncls][3] gradient coefficients
 * consumed by iunet_head_loss_bwd.  slab: iunet_head_loss_num_parts*ncls*8 floats. */
int iunet_head_loss_num_parts(int N, long long vox);
int iunet_head_loss_fwd(int dtype, const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls,
                        const void* target, const void* weight, int tdtype, int kind, void* slab, void* out4, void* coef,
                        int N, long long vox, void* stream);
/* backward through loss, softmax and head: dx (NHWC8c), dW/db partial slab
 * [iunet_head_loss_bwd_num_parts][ncls*(C0+1)] (per row: [C0/8][ncls][8] weight partials, then [ncls] bias partials). */
int iunet_head_loss_bwd_num_parts(int N, long long vox, int ncls, int C0);
int iunet_head_loss_bwd(int dtype, const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls,
                        const void* target, const void* weight, int tdtype, const void* coef, float loss_scale, void* dx,
                        long long dx_ss, void* dwslab, int N, long long vox, void* stream);
/* both with the head input given as relu(in_scale[c] * x + in_shift[c]) rounded to the activation dtype (fp32 [C0] each; see
 * iunet_conv3_fwd_act): in training the BatchNorm + ReLU of the last stage conv is applied while loading and its output
 * tensor is never written (bit-identical to running on the tensor iunet_bn_relu_fwd would store).  dx of the backward is the
 * gradient of that ACTIVATION, i.e. what iunet_bn_relu_bwd takes as dz. */
int iunet_head_loss_fwd_act(int dtype, const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls,
                            const void* target, const void* weight, int tdtype, int kind, void* slab, void* out4, void* coef,
                            const void* in_scale, const void* in_shift, int N, long long vox, void* stream);
int iunet_head_loss_bwd_act(int dtype, const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls,
                            const void* target, const void* weight, int tdtype, const void* coef, float loss_scale, void* dx,
                            long long dx_ss, void* dwslab, const void* in_scale, const void* in_shift, int N, long long vox,
                            void* stream);
/* out[i] = alpha * sum_p slab[p][i] (+ out[i]); fixed summation order.  The slab is scratch:
 * wide slabs are folded in place first. */
int iunet_reduce_slab(void* slab, int nparts, long long n, void* out, float alpha, int accumulate, void* stream);
/* weight gradient of the 3^d conv on MFMA (transposing LDS reads): dW fp32 [Cout][Cin][taps]. */
int iunet_conv3_wgrad_blocks(int nd, int N, int D, int H, int W, int Cin, int Cout);
long long iunet_conv3_wgrad_slab_floats(int nd, int N, int D, int H, int W, int Cin, int Cout);
int iunet_conv3_wgrad(int dtype, int nd, const void* x, long long x_ss, const void* dy, long long dy_ss, void* slab,
                      void* dW, float alpha, int N, int D, int H, int W, int Cin, int Cout, void* stream);
/* the same with the conv input given as relu(x_scale[c] * x + x_shift[c]) (see iunet_conv3_fwd_act). */
int iunet_conv3_wgrad_act(int dtype, int nd, const void* x, long long x_ss, const void* dy, long long dy_ss, void* slab,
                          void* dW, float alpha, const void* x_scale, const void* x_shift, int N, int D, int H, int W,
                          int Cin, int Cout, void* stream);
/* transposed conv backward (N, D, H, W, Cin, Cout describe the forward op). */
int iunet_pack_convT_dgrad(int dtype, const void* w, void* dst, int Cin, int Cout, int npos, void* stream);
int iunet_convT_dgrad(int dtype, int nd, const void* dy, long long dy_ss, void* dx, long long dx_ss, const void* wpk,
                      int N, int D, int H, int W, int Cin, int Cout, void* stream);
/* dW fp32 [Cin][Cout][2^d] and db fp32 [Cout] on MFMA; wslab: blocks*Cin*Cout*2^d floats, bslab: blocks*Cout floats. */
int iunet_convT_wgrad_blocks(int nd, int N, int D, int H, int W, int Cin, int Cout);
int iunet_convT_wgrad(int dtype, int nd, const void* x, long long x_ss, const void* dy, long long dy_ss, void* wslab,
                      void* bslab, void* dW, void* db, int N, int D, int H, int W, int Cin, int Cout, void* stream);
/* first conv weight gradient on MFMA: dW fp32 [Cout][Cin][taps]; slab: blocks * Cout * 112 floats of scratch. */
int iunet_first_conv_wgrad_blocks(int nd, int N, int D, int H, int W);
int iunet_first_conv_wgrad(int dtype, int nd, const void* x, int in_dtype, const long long* in_strides, const void* dy,
                           long long dy_ss, void* slab, void* dW, int N, int D, int H, int W, int Cin, int Cout,
                           void* stream);
/* the same with the second BatchNorm-backward pass folded in: dz = gradient of the first conv's activation, y = its raw
 * output, coef from iunet_bn_relu_bwd called with dy = NULL (sums + coefficients only); the gradient of the raw output is
 * never written. */
int iunet_first_conv_wgrad_bn(int dtype, int nd, const void* x, int in_dtype, const long long* in_strides, const void* dz,
                              long long dz_ss, const void* y, long long y_ss, const void* mean, const void* invstd,
                              const void* coef, const void* scale, const void* shift, void* slab, void* dW, int N, int D, int H,
                              int W, int Cin, int Cout, void* stream);
/* AdamW with torch defaults (unet.py:71-73); grads are multiplied by grad_scale_inv (loss scaling, 1/world);
 * if *skip_flag != 0 (set by iunet_check_finite) the step is skipped. */
int iunet_check_finite(const void* g, long long n, void* flag, void* stream);
int iunet_adamw_step(void* p, const void* g, void* m, void* v, long long n, float lr, float b1, float b2, float eps,
                     float wd, int step, float grad_scale_inv, const void* skip_flag, void* stream);

/* ---- training state on the device (csrc/train_pointwise.hip): 8 x 4 bytes -- [0] float loss scale, [1] int completed optimiser steps,
 * [2] int good steps since the scale last changed, [3] int overflow flag of the last step, [4..6] the step's AdamW coefficients, [7] int
 * dynamic scaling.  The reference trains under precision='16-mixed' (trainer.py:59): GradScaler's semantics -- an overflowing fp16 step is
 * skipped, halves the scale and is not counted; 2000 good steps double it -- without a host read per step. */
int iunet_train_state_init(void* state, float loss_scale, int dynamic, void* stream);
/* iunet_head_loss_bwd / _bwd_act (in_scale / in_shift non-null) with the loss scale read from state[0] */
int iunet_head_loss_bwd_dev(int dtype, const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls,
                            const void* target, const void* weight, int tdtype, const void* coef, const void* state, void* dx,
                            long long dx_ss, void* dwslab, const void* in_scale, const void* in_shift, int N, long long vox,
                            void* stream);
/* [r5] The head's backward and the BatchNorm + ReLU backward of the last stage conv (training with the head reading that conv's RAW
 * output y: iunet_head_loss_fwd_act) in two passes over y -- the head's input gradient is never written; replaces iunet_head_loss_bwd_dev
 * + iunet_bn_relu_bwd for that layer (y read twice and dy written once instead of y three times, dz written once and read twice).
 * iunet_head_bn_bwd_ok: 32 or 64 head input channels, 2..4 classes.  dwslab / the reduction of its rows as iunet_head_loss_bwd; dgamma, dbeta,
 * bncoef (3 * C0 floats of scratch) as iunet_bn_relu_bwd; bnslab: iunet_bn_bwd_num_parts(N, vox) * 2 C0 floats; dl_scratch: N * vox * ncls
 * floats (the logit gradients, written by the first pass and read by the second); dy: the gradient of y.
 * Loss scale: state[0] when state is not NULL (the training handle's device state), else loss_scale. */
int iunet_head_bn_bwd_ok(int C0, int ncls);
int iunet_head_bn_bwd(int dtype, const void* y, long long y_ss, int C0, const void* w, const void* bias, int ncls, const void* target,
                      const void* weight, int tdtype, const void* coef, float loss_scale, const void* state, const void* scale,
                      const void* shift, const void* mean, const void* invstd, const void* gamma, void* dgamma, void* dbeta, void* dy,
                      long long dy_ss, void* dwslab, void* bnslab, void* bncoef, void* dl_scratch, int N, long long vox, void* stream);
/* ... for GroupNorm networks: scale / shift / mean / invstd are [N][C0] rows, bncoef N * C0 * 3 floats (as iunet_gn_relu_bwd), the head's
 * forward reads the last conv's raw output with per-sample rows too (iunet_head_loss_fwd_act_ps, per_sample = 1; iunet_gn_relu_fwd_rows with
 * z = NULL leaves only scale / shift / mean / invstd). */
int iunet_head_gn_bwd(int dtype, const void* y, long long y_ss, int C0, const void* w, const void* bias, int ncls, const void* target,
                      const void* weight, int tdtype, const void* coef, float loss_scale, const void* state, const void* scale,
                      const void* shift, const void* mean, const void* invstd, const void* gamma, int groups, void* dgamma, void* dbeta, void* dy,
                      long long dy_ss, void* dwslab, void* bnslab, void* bncoef, void* dl_scratch, int N, long long vox, void* stream);
int iunet_head_loss_fwd_act_ps(int dtype, const void* x, long long x_ss, int C0, const void* w, const void* bias, int ncls,
                               const void* target, const void* weight, int tdtype, int kind, void* slab, void* out4, void* coef,
                               const void* in_scale, const void* in_shift, int per_sample, int N, long long vox, void* stream);
/* dW [ncls][C0], db [ncls] of the head from the reduced row of iunet_head_loss_bwd's slab ([C0 / 8][ncls][8] weight sums, then [ncls]) */
int iunet_head_grad_scatter(const void* row, void* dw, void* db, int ncls, int C0, void* stream);
/* the optimiser step on the device state: overflow check of the flat gradient (check != 0: fp16), AdamW (unet.py:71-73; skipped on
 * overflow) with its bias corrections from state[1] + 1, then the scale / step-count update; world = ranks the gradient was summed over */
int iunet_adamw_step_dev(void* p, const void* g, void* m, void* v, long long n, float lr, float b1, float b2, float eps, float wd,
                         void* state, int check, float world, void* stream);

/* ---- handle level: the TRAINING step as one call (csrc/train_net.hip) -----------------------------------------------------------
 * unet.py:88-102 (forward with BatchNorm batch statistics, the reference's loss on the softmax probabilities, metrics.py) + backward +
 * AdamW (unet.py:71-73) + the re-pack of the updated operators, sequenced in C++ -- the launches interactive_unet/train_engine.py
 * sequences from Python, bit for bit.  Every device buffer is the caller's:
 *   iunet_train* t; iunet_train_create(2, 4, 32, 1, 2, 0, 6, &t);          // 2-D, 4 levels, base 32, 1 -> 2 classes, fp16, mcc_ce
 *   flat, grad, m, v = device floats [iunet_train_num_params(t)] (iunet_train_param: name / offset / count; grad, m, v zeroed)
 *   running[2 * iunet_train_num_bn(t)] = device pointers: running_mean, running_var ([Cout] floats) of every BatchNorm, canonical order
 *   state = 32 device bytes, iunet_train_state_init(state, 1024.f, 1, stream);   packed = device bytes [iunet_train_packed_bytes(t)]
 *   iunet_train_bind(t, flat, grad, m, v, running, packed, state, stream);
 *   ws = device bytes [iunet_train_workspace_bytes(t, N, 1, H, W)];
 *   iunet_train_step(t, x, 2, x_strides, target, weight, 1, N, 1, H, W, ws, 1e-4f, 0.9f, 0.999f, 1e-8f, 1e-2f, out4, stream);
 * dtype: 0 fp16 (dynamic loss scale), 1 bf16; loss_kind: 0 ce, 1 dice, 2 iou, 3 mcc, 4 dice_ce, 5 iou_ce, 6 mcc_ce (utils.py:458-475).
 * Data parallel callers run iunet_train_forward_backward, all-reduce `grad`, then iunet_train_update(world = ranks). */
typedef struct iunet_train iunet_train;
int iunet_train_create(int dim, int levels, int base, int cin, int ncls, int dtype, int loss_kind, iunet_train** out);
/* the same with the normalisation named: norm 0 = BatchNorm, 1 = GroupNorm(groups) after every stage conv (north star "GroupNorm/BN";
 * statistics per (sample, group), the same at training and inference: the running-statistics pointers of iunet_train_bind are accepted and
 * left alone; train_engine.TrainEngine on a UNet(norm='group') sequences the same launches) */
int iunet_train_create_ex(int dim, int levels, int base, int cin, int ncls, int dtype, int loss_kind, int norm, int groups, iunet_train** out);
void iunet_train_destroy(iunet_train* t);
long long iunet_train_num_params(const iunet_train* t);
int iunet_train_num_tensors(const iunet_train* t);
int iunet_train_param(const iunet_train* t, int index, char* name, int name_cap, long long* offset, long long* numel);
int iunet_train_num_bn(const iunet_train* t);
long long iunet_train_packed_bytes(const iunet_train* t);
long long iunet_train_workspace_bytes(const iunet_train* t, int N, int D, int H, int W);
int iunet_train_bind(iunet_train* t, void* flat, void* grad, void* m, void* v, void* const* running, void* packed, void* state, void* stream);
int iunet_train_repack(iunet_train* t, void* stream);
int iunet_train_forward_backward(iunet_train* t, const void* x, int in_dtype, const long long* in_strides, const void* target,
                                 const void* weight, int tdtype, int N, int D, int H, int W, void* workspace, void* out4, void* stream);
/* Data parallelism with the all-reduce overlapped (trainer.py:56-63 on N GPUs, one process per GPU): the same call with a HOST callback
 * between the backward's launches -- hook(ctx, 0) once every launch writing the decoder + head gradients (the tail of `grad`, from the
 * first decoder parameter on) is enqueued, hook(ctx, 1) once the bottom encoder level's are; the caller starts that bucket's all-reduce
 * there, behind an event on `stream` (interactive_unet/dp.py: GradBuckets).  hook may be NULL. */
typedef void (*iunet_train_hook)(void* ctx, int stage);
int iunet_train_forward_backward_hooks(iunet_train* t, const void* x, int in_dtype, const long long* in_strides, const void* target,
                                       const void* weight, int tdtype, int N, int D, int H, int W, void* workspace, void* out4,
                                       iunet_train_hook hook, void* hook_ctx, void* stream);
int iunet_train_update(iunet_train* t, float lr, float b1, float b2, float eps, float wd, float world, void* stream);
int iunet_train_step(iunet_train* t, const void* x, int in_dtype, const long long* in_strides, const void* target, const void* weight,
                     int tdtype, int N, int D, int H, int W, void* workspace, float lr, float b1, float b2, float eps, float wd, void* out4,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif
