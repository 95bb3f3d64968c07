// One operator of iunet_x2_prep_batch (split16.hip: kinds 0..2) / iunet_x2m_prep_batch (conv3_x2m.hip: kind 3): the arguments of
// iunet_x2_prep / iunet_x2m_prep_nd as a row of a device-resident table, so that a prediction engine re-prepares every operator of the
// network after an optimiser step in one launch per kernel instead of one per layer (interactive_unet/engine_x2.py: load_eval).
#pragma once
struct X2PrepDesc {              // mirrored by interactive_unet/_native.py: X2PrepDesc (ctypes)
  const float* w;                // fp32 master weights: [Cout][Cin][taps] (kinds 0, 3) or [Cin][Cout][npos] (kinds 1, 2)
  float* out;                    // iunet_x2_prep's wv / iunet_x2m_prep_nd's whi (fp32, fed to iunet_pack_batch)
  unsigned char* w8;             // kind 3: the K = 128 operator bytes; else null
  float* oscale;                 // [Cout]
  float* bias_out;               // [Cout]
  const float* gamma;            // eval-mode BatchNorm fold (all four or none)
  const float* beta;
  const float* mean;
  const float* var;
  const float* bias_in;          // the layer's own bias (transposed convs) or null
  float eps, act_in, act_out;
  int Cout, Cin, taps;
  int kind;                      // 0..2: iunet_x2_prep's `transposed`; 3: iunet_x2m_prep_nd
  int kc;                        // kind 0: channels per chunk; kinds 1, 2: iunet_x2_convT_kc(Cin)
  int row0;                      // first workgroup of this operator in the launch (running sum of Cout over the table)
};
