// The thin-convolution launches that only the per-operation C entry points (ava_conv3x3 / ava_conv3x3_wgrad: include/ava_hip.h)
// reach -- the separate data gradients of conv1 / convt7 and the separate weight gradients.  The model's own backward runs the
// fused kernels of conv_thin.hip; these instantiations live in their own translation unit so that their register-heavy
// variants stay out of the hot file's code object and ISA scans.  Kernel templates: conv_thin_kernels.h.
#include "conv_thin_kernels.h"

static inline bool thin_width_ok(int W) { return W == 128 || W == 256; }
template <typename K>
static int thin_set_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes) == hipSuccess ? AVA_OK : AVA_ELAUNCH;
}

// a: as prepared by conv_thin.hip's dispatcher (ntiles, part_rows, clamped grid); EPI_BWD forms only
template <int W>
static int thin_perop_w(const ConvArgs& a, int grid, int Cin, int pro, hipStream_t st) {
  const size_t kThin8Lds = (size_t)(THIN_IR * THIN_IC * 8 + 96 + 8) * sizeof(float);
  const dim3 block(2 * W);
  if (Cin == 1) {
    if (pro == PRO_ID) hipLaunchKernelGGL((thin_1to8_kernel<W, PRO_ID, EPI_BWD>), dim3(grid), block, 0, st, a);
    else if (pro == PRO_BWD) hipLaunchKernelGGL((thin_1to8_kernel<W, PRO_BWD, EPI_BWD>), dim3(grid), block, 0, st, a);
    else return AVA_EINVAL;
  } else if (Cin == 8) {
    static bool attr = false;
    if (!attr) {
      if (thin_set_lds(&thin_8to1_kernel<W, PRO_BWD, EPI_BWD>, kThin8Lds) != AVA_OK ||
          thin_set_lds(&thin_8to1_kernel<W, PRO_ID, EPI_BWD>, kThin8Lds) != AVA_OK)
        return AVA_ELAUNCH;
      attr = true;
    }
    if (pro == PRO_BWD) hipLaunchKernelGGL((thin_8to1_kernel<W, PRO_BWD, EPI_BWD>), dim3(grid), block, kThin8Lds, st, a);
    else if (pro == PRO_ID) hipLaunchKernelGGL((thin_8to1_kernel<W, PRO_ID, EPI_BWD>), dim3(grid), block, kThin8Lds, st, a);
    else return AVA_EINVAL;
  } else {
    return AVA_EINVAL;
  }
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

int ava_conv3x3_thin_perop(const ConvArgs& a, int grid, int W, int Cin, int pro, hipStream_t st) {
  if (W == 128) return thin_perop_w<128>(a, grid, Cin, pro, st);
  if (W == 256) return thin_perop_w<256>(a, grid, Cin, pro, st);
  return AVA_EINVAL;
}

template <int W>
static int conv3x3_wgrad_thin_w(const WgradArgs& a0, int grid, int Cin, int Cout, int dy_pro, hipStream_t st) {
  const size_t kThin8Lds = (size_t)(THIN_IR * THIN_IC * 8 + 96 + 8) * sizeof(float);
  const dim3 block(2 * W);
  WgradArgs a = a0;
  if (a.act_bf16) return AVA_EINVAL;                     // per-op kernels: fp32 activations only
  a.ntiles = a.B * (a.Ho / THIN_TH);
  if (Cin == 1 && Cout == 8) {
    if (W == 256 && grid > 256) grid = 256;              // 200-register kernel: one 512-thread workgroup per CU
    if (dy_pro == PRO_BWD) hipLaunchKernelGGL((thin_wgrad_1to8_kernel<W, PRO_BWD>), dim3(grid), block, 0, st, a);
    else if (dy_pro == PRO_ID) hipLaunchKernelGGL((thin_wgrad_1to8_kernel<W, PRO_ID>), dim3(grid), block, 0, st, a);
    else return AVA_EINVAL;
  } else if (Cin == 8 && Cout == 1) {
    static bool attr = false;
    if (!attr) {
      if (thin_set_lds(&thin_wgrad_8to1_kernel<W, PRO_BWD>, kThin8Lds) != AVA_OK ||
          thin_set_lds(&thin_wgrad_8to1_kernel<W, PRO_ID>, kThin8Lds) != AVA_OK)
        return AVA_ELAUNCH;
      attr = true;
    }
    if (W == 256 && grid > 256) grid = 256;
    if (dy_pro == PRO_BWD) hipLaunchKernelGGL((thin_wgrad_8to1_kernel<W, PRO_BWD>), dim3(grid), block, kThin8Lds, st, a);
    else if (dy_pro == PRO_ID) hipLaunchKernelGGL((thin_wgrad_8to1_kernel<W, PRO_ID>), dim3(grid), block, kThin8Lds, st, a);
    else return AVA_EINVAL;
  } else {
    return AVA_EINVAL;
  }
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

// workgroups (= partial rows) ava_conv3x3_wgrad_thin launches for this shape; 0: not a thin shape
int ava_conv3x3_wgrad_thin_rows(const WgradArgs& a, int grid, int Cin, int Cout, int mode) {
  if (mode != MODE_S1 || !thin_width_ok(a.Wo) || a.Ho % THIN_TH != 0) return 0;
  if (!((Cin == 1 && Cout == 8) || (Cin == 8 && Cout == 1))) return 0;
  return (a.Wo == 256 && grid > 256) ? 256 : grid;
}

int ava_conv3x3_wgrad_thin(const WgradArgs& a0, int grid, int Cin, int Cout, int mode, int dy_pro, hipStream_t st) {
  if (mode != MODE_S1 || !thin_width_ok(a0.Wo) || a0.Ho % THIN_TH != 0) return AVA_EINVAL;
  if (a0.Wo == 128) return conv3x3_wgrad_thin_w<128>(a0, grid, Cin, Cout, dy_pro, st);
  return conv3x3_wgrad_thin_w<256>(a0, grid, Cin, Cout, dy_pro, st);
}

