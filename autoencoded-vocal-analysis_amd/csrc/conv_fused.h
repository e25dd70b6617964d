// Fused backward (data gradient + BatchNorm-backward sums + weight/bias-gradient partials) of one conv layer.
#pragma once
#include "conv_common.h"

struct FusedArgs {
  const float* x;        // raw layer input [B,Hi,Wi,CI]; BatchNorm scale/shift xa, xb (x_n = xa*x + xb)
  const float* xa;
  const float* xb;
  const float* dy;       // upstream gradient [B,Ho,Wo,CO]; with dy2 (saved activation) and da, db, dc: PRO_BWD
  const float* dy2;
  const float* da;
  const float* db;
  const float* dc;
  const float* Gb;       // backward-data weights in gather layout (pack kinds 3..6)
  float* dx;             // [B,Hi,Wi,CI] gradient w.r.t. the BatchNorm output (may be null for the 1 -> 8 layer: sums only)
  const float* mean;     // batch statistics of x (BatchNorm-backward sums)
  const float* invstd;
  float* bn_partials;    // [grid][2*CI]
  float* wg_partials;    // [grid][9*CI*CO + CO]
  int B, Hi, Wi, Ho, Wo;
  int tiles_y, tiles_x, ntiles;
  long long* acc_out;    // != null: sum dx, sum dx*xhat are accumulated here (bn_acc.h) instead of bn_partials rows
  BnFin fin;             // fin.acc != null: da / db / dc are derived from the accumulated sums of the layer above
  int act_bf16;          // x and dy2 (activations) are stored as bfloat16; dy and dx (gradients) are always fp32
  RecompArgs rcd;        // rcd.G1 != null: `dy` is a 1-channel tensor and the 8-channel upstream gradient is its 3x3 gather with the
                         // weights G1 [9][1][8], formed in the staging waves (conv_recomp.h: convt7's data gradient inside convt6's backward)
  int skip_dx;           // thin 8 -> 1 backward: only the weight gradient + BatchNorm sums (its data gradient is formed by the consumer)
  RecompArgs rc;         // rc.G1 != null: `x` is the raw spectrogram batch; the layer input y1 is recomputed from it (conv_recomp.h)
  int sweep;             // thin kernels: workgroups sweep the tile list together instead of per-XCD chunks
  int dbg;               // lab build: phase ablation bits of the wave-specialised kernel (AVA_FDBG; timing only)
#ifdef AVA_LAB
  unsigned long long* stamps;   // lab: 16 s_memrealtime stamps of this launch (workgroup 0), tools/lab/conv_stamps.py
#endif
};
#ifdef AVA_LAB
unsigned long long* ava_lab_next_stamps_f();
#endif

// 0 when (Cin, Cout, mode, size) has no fused instantiation
int ava_conv_fused_grid_for(int B, int Hi, int Wi, int Cin, int Cout, int mode);
int ava_conv3x3_bwd_fused_launch(const FusedArgs& a, int Cin, int Cout, int mode, int dy_pro, hipStream_t st);

// conv1 (1 -> 8) and convt7 (8 -> 1) at 128 x 128: VALU kernels in conv_thin.hip
int ava_thin_fused_grid(int B, int Hi, int Wi, int Cin, int Cout, int mode);
int ava_thin_bwd_fused_launch(const FusedArgs& a, int grid, int Cin, int dy_pro, hipStream_t st);

// the same backward with both products on bf16 limb MFMA (conv_fused_limb.hip); AVA_EINVAL: no limb instantiation
bool ava_conv_fused_limb_has(int Cin, int Cout, int mode);
int ava_conv_fused_limb_cap(int Cin, int Cout, int mode, int* tw, int* th);
int ava_conv3x3_bwd_fused_limb_launch(const FusedArgs& a, int grid, int Cin, int Cout, int mode, int dy_pro, hipStream_t st);
