// One layer of iunet_pack_batch (pack_batch.hip); built by interactive_unet/_native.py (ctypes mirror) and by train_net.hip.
#pragma once
struct PackDesc {                // mirrored by interactive_unet/_native.py: PackDesc (ctypes)
  const float* w;                // fp32 master weights
  const float* gamma;            // BatchNorm fold (all four or none): scale = gamma / sqrt(var + eps)
  const float* beta;
  const float* mean;
  const float* var;
  float* bias_out;               // folded bias [Cout] = beta - mean * scale (or null)
  void* dst;                     // packed operator
  long long total;               // elements of dst
  int Cout, Cin, taps;           // original operator dims (convT: Cin, Cout, npos in `taps`)
  int kind;                      // 0 conv3 layout 0, 1 conv3 K16 (layout 1), 2 first conv, 3 convT fwd, 4 convT dgrad,
                                 // 5 conv3 K16 as OCP e4m3 bytes (conv3_f8.hip's operator; qscale = its per-channel scales, required),
                                 // 6 conv3 compact K16 (3^3 only; conv3_v4.hip layout 3, conv3_mfma.hip: pack_conv3_k16c_kernel)
  int dgrad;                     // conv3 only: data-gradient operator
  int dtype;                     // 0 f16, 1 bf16
  float eps;
  int pad_;
  float* qscale;                 // [output channels] or null.  Non-null: weights are quantised to OCP e4m3 values times a
                                 // per-output-channel power-of-two scale (exact in f16 / bf16) -- BASELINE config C5
};
