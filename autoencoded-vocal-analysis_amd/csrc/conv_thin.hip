// The "thin" convolutions at full resolution (128x128, stride 1) -- conv1 (1 -> 8 channels) and convt7 (8 -> 1)
// forward, their backward-data counterparts and both weight gradients -- plus convt6's forward (8 -> 8, stride-2
// transposed).  With 1 or 8 channels per side the matrix cores idle most of the time (15/16 resp. 1/2 of every MFMA
// is padding); these are HBM-bound VALU kernels built on packed FMAs (v_pk_fma_f32).
//
// The kernels on the model's path are in the DIRECT form: lane pairs share a pixel column, thread (x, h) owns 4 of
// the 8 channels of its column, the 8-channel tensor is read from / written to global memory by the thread that owns
// the pixel (every 16-byte access of a wave is part of 1 KB of full lines), and only 1-channel windows or a few
// partial sums go through LDS:
//   thin_1to8_kernel                      conv1 forward, convt7 data gradient (1-channel window in LDS)
//   thin_8to1_direct_kernel               convt7 forward + SSE epilogue (24 partial sums per thread through LDS)
//   thin_bwd_fused_1to8_kernel            conv1 backward (weight gradient + BatchNorm sums; no data gradient needed)
//   thin_wgrad_stats_8to1_direct_kernel   convt7 weight gradient + BatchNorm sums
//   up88_direct_kernel                    convt6 forward
// The earlier LDS-staged forms (thin_8to1_kernel, thin_8to1_ws_kernel, thin_wgrad_*_kernel, thin_wgrad_stats_8to1_kernel:
// a thread owns a vertical strip of 4 pixels, the 8-channel window is staged in LDS with the prologue applied) remain
// as alternatives behind environment switches (tools/README.md) and for the shapes the tests exercise through
// ava_conv3x3 / ava_conv3x3_wgrad.  Same ConvArgs / WgradArgs / partial-row conventions as conv.hip.
#include "conv_thin_kernels.h"

#ifndef AVA_THIN_RECY
#define AVA_THIN_RECY 1     // conv1's backward recomputes y1 from x (thin_bwd_fused_1to8_kernel<.., RECY>); 0: reads the saved tensor
#endif

// launchers of the kernel forms that only the per-operation entry points reach (conv_thin_perop.hip): the separate data gradients
// of conv1 / convt7 (the model's backward runs the fused kernels instead) and the separate weight gradients
int ava_conv3x3_thin_perop(const ConvArgs& a, int grid, int W, int Cin, int pro, hipStream_t st);

// convt6 forward in the direct form; AVA_EINVAL: not this layer / switched off (AVA_UP88_DIRECT=0)
int ava_conv3x3_up88_direct(const ConvArgs& a0, int grid, int Cin, int Cout, int mode, int pro, int epi, hipStream_t st) {
  static const int on = [] { const char* e = ava_env("AVA_UP88_DIRECT"); return e ? atoi(e) : 1; }();
  if (!on || Cin != 8 || Cout != 8 || mode != MODE_UP || pro != PRO_BN || epi != EPI_FWD || !a0.relu || a0.out2 != nullptr ||
      a0.Wo != 128 || a0.Wi != 64 || a0.Ho % 8 != 0 || a0.out == nullptr)
    return AVA_EINVAL;
  ConvArgs a = a0;
  a.ntiles = a.B * (a.Ho / 8);
  a.part_rows = grid;
  static const int resident = ava_resident_grid(&up88_direct_kernel<float>, 0);
  int g = grid < ava_scale_grid(resident) ? grid : ava_scale_grid(resident);
  { const char* e = ava_env("AVA_UP88_GRID"); if (e && atoi(e) >= 8 && atoi(e) < g) g = atoi(e); }
  if (g > a.ntiles) g = a.ntiles;
  if (a.act_bf16) hipLaunchKernelGGL(up88_direct_kernel<ava_bf16>, dim3(g), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(up88_direct_kernel<float>, dim3(g), dim3(256), 0, st, a);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// host side.  The kernels exist for W = 128 (BASELINE configs 1-4) and W = 256 (config 5); a workgroup has 2*W threads.
// ---------------------------------------------------------------------------------------------------------
static inline bool thin_width_ok(int W) { return W == 128 || W == 256; }

// resident workgroups of a 2*W-thread kernel (occupancy x CUs)
template <typename K>
static int thin_resident(K kernel, int W, size_t lds) { return ava_resident_grid(kernel, lds, 2 * W); }

#ifdef AVA_LAB
static int thin_ws_mode() {
  static const int ws = [] { const char* e = ava_env("AVA_THIN_WS"); return e ? atoi(e) : 1; }();
  return ws;
}
#endif

template <int W>
static int thin_fused_grid_w(int nt, int Cin) {
  if (Cin == 1) {
    // conv1's backward: 768 workgroups of 256 threads (3 per CU) measured best at W = 128; one resident wave at W = 256
    static const int cap1 = [] {
      const char* e = ava_env("AVA_THIN_GRID1");
      if (e && atoi(e) >= 8) return atoi(e);
      if (AVA_THIN_RECY) {   // 238 VGPRs: two 256-thread workgroups per CU.  The grid is also the partial-row count, so it must not
                             // depend on the activation type: the smaller residency of the two instantiations serves both
        const int rf = thin_resident(&thin_bwd_fused_1to8_kernel<W, PRO_BWD, float, true>, W, 0);
        const int rb = thin_resident(&thin_bwd_fused_1to8_kernel<W, PRO_BWD, ava_bf16, true>, W, 0);
        return rf < rb ? rf : rb;
      }
      if (W == 128) return 768;
      return thin_resident(&thin_bwd_fused_1to8_kernel<W, PRO_BWD>, W, 0);
    }();
    return nt < ava_scale_grid(cap1) ? nt : ava_scale_grid(cap1);
  }
  static const int cap8 = [] {
    const char* e = ava_env("AVA_THIN_GRID8");
    if (e && atoi(e) >= 8) return atoi(e);
#ifdef AVA_LAB
    const char* d = ava_env("AVA_THIN_STATS_DIRECT");
    if (d && atoi(d) == 0) return 512;                    // LDS-staged form: two workgroups per CU
#endif
    // direct form: one resident wave (3 per CU at 138 VGPRs; in-step A/B 512 / 768 / 1024 -> 48.0 / 42.4 / 55.7 us)
    return thin_resident(&thin_wgrad_stats_8to1_direct_kernel<W, PRO_ID>, W, 0);
  }();
  return nt < ava_scale_grid(cap8) ? nt : ava_scale_grid(cap8);
}

int ava_thin_fused_grid(int B, int Hi, int Wi, int Cin, int Cout, int mode) {
  if (mode != MODE_S1 || !thin_width_ok(Wi) || Hi % THIN_TH != 0) return 0;
  if (!((Cin == 1 && Cout == 8) || (Cin == 8 && Cout == 1))) return 0;
  const int nt = B * (Hi / THIN_TH);
  return Wi == 128 ? thin_fused_grid_w<128>(nt, Cin) : thin_fused_grid_w<256>(nt, Cin);
}

template <int W>
static int thin_bwd_fused_launch_w(const FusedArgs& a0, int grid, int Cin, int dy_pro, hipStream_t st) {
  FusedArgs a = a0;
  a.ntiles = a.B * (a.Ho / THIN_TH);
  // the 8-channel tensors of these kernels have no halo: sweeping the tile list together beats per-XCD chunks
  // (in-step A/B: conv1 backward 68.0 -> 59.0 us, convt7 weight gradient 44.5 -> 42.9 us)
  { static const int sw = [] { const char* e = ava_env("AVA_THIN_SWEEP"); return e ? atoi(e) : 1; }(); a.sweep = sw; }
  if (dy_pro != PRO_BWD && dy_pro != PRO_ID) return AVA_EINVAL;
  const dim3 block(2 * W);
  if (Cin == 1) {
    if (a.dx != nullptr) return AVA_EINVAL;              // this layer's data gradient is never formed
    const bool recy = AVA_THIN_RECY && dy_pro == PRO_BWD && a.rc.G1 != nullptr && a.rc.bias1 != nullptr && a.rc.pa1 != nullptr && a.rc.pb1 != nullptr;
    if (recy && a.act_bf16) hipLaunchKernelGGL((thin_bwd_fused_1to8_kernel<W, PRO_BWD, ava_bf16, true>), dim3(grid), block, 0, st, a);
    else if (recy) hipLaunchKernelGGL((thin_bwd_fused_1to8_kernel<W, PRO_BWD, float, true>), dim3(grid), block, 0, st, a);
    else if (dy_pro == PRO_BWD && a.act_bf16) hipLaunchKernelGGL((thin_bwd_fused_1to8_kernel<W, PRO_BWD, ava_bf16>), dim3(grid), block, 0, st, a);
    else if (dy_pro == PRO_BWD) hipLaunchKernelGGL((thin_bwd_fused_1to8_kernel<W, PRO_BWD>), dim3(grid), block, 0, st, a);
    else hipLaunchKernelGGL((thin_bwd_fused_1to8_kernel<W, PRO_ID>), dim3(grid), block, 0, st, a);      // no activation read
    AVA_CHECK_LAUNCH();
    return AVA_OK;
  }
  // 8 -> 1: dx by the 1 -> 8 gather kernel (store only), then weight gradient + BatchNorm sums in one pass over x
  if (a.dx == nullptr && !a.skip_dx) return AVA_EINVAL;
  ConvArgs c = {};
  c.in = a.dy; c.in2 = a.dy2; c.pa = a.da; c.pb = a.db; c.pc = a.dc; c.G = a.Gb; c.out = a.dx;
  c.B = a.B; c.Hi = a.Ho; c.Wi = a.Wo; c.Ho = a.Hi; c.Wo = a.Wi; c.ntiles = a.ntiles;
  // no partial rows here, so the grid is free: one resident wave of workgroups (6 per CU at 76 VGPRs) is the fastest
  // (in-step rocprof A/B: 768 / 1024 / 1536 / 2048 / 4096 workgroups -> 33.5 / 33.1 / 30.7 / 35.1 / 39.8 us)
  static const int dcap = [] {
    const char* e = ava_env("AVA_THIN_DGRID");
    if (e && atoi(e) >= 8) return atoi(e);
    return thin_resident(&thin_1to8_kernel<W, PRO_ID, EPI_NONE>, W, 0);
  }();
  const int dgrid = a.ntiles < ava_scale_grid(dcap) ? a.ntiles : ava_scale_grid(dcap);
  if (!a.skip_dx) {                       // (skipped when the consumer forms this gradient itself: conv_recomp.h)
    if (dy_pro == PRO_BWD) hipLaunchKernelGGL((thin_1to8_kernel<W, PRO_BWD, EPI_NONE>), dim3(dgrid), block, 0, st, c);
    else hipLaunchKernelGGL((thin_1to8_kernel<W, PRO_ID, EPI_NONE>), dim3(dgrid), block, 0, st, c);
    AVA_CHECK_LAUNCH();
  }
#ifdef AVA_LAB
  if constexpr (W == 128) {
    static const int direct = [] { const char* e = ava_env("AVA_THIN_STATS_DIRECT"); return e ? atoi(e) : 1; }();
    if (direct == 0) {
      const size_t lds = (size_t)(THIN_IR * 130 * 8 + 96 + 16) * sizeof(float);
      static bool attr = false;
      if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_wgrad_stats_8to1_kernel<PRO_BWD>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_wgrad_stats_8to1_kernel<PRO_ID>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
          return AVA_ELAUNCH;
        attr = true;
      }
      if (dy_pro == PRO_BWD) hipLaunchKernelGGL((thin_wgrad_stats_8to1_kernel<PRO_BWD>), dim3(grid), dim3(256), lds, st, a);
      else hipLaunchKernelGGL((thin_wgrad_stats_8to1_kernel<PRO_ID>), dim3(grid), dim3(256), lds, st, a);
      AVA_CHECK_LAUNCH();
      return AVA_OK;
    }
  }
#endif
  if (a.act_bf16) {
    if (dy_pro == PRO_BWD) hipLaunchKernelGGL((thin_wgrad_stats_8to1_direct_kernel<W, PRO_BWD, ava_bf16>), dim3(grid), block, 0, st, a);
    else hipLaunchKernelGGL((thin_wgrad_stats_8to1_direct_kernel<W, PRO_ID, ava_bf16>), dim3(grid), block, 0, st, a);
  } else {
    if (dy_pro == PRO_BWD) hipLaunchKernelGGL((thin_wgrad_stats_8to1_direct_kernel<W, PRO_BWD>), dim3(grid), block, 0, st, a);
    else hipLaunchKernelGGL((thin_wgrad_stats_8to1_direct_kernel<W, PRO_ID>), dim3(grid), block, 0, st, a);
  }
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

int ava_thin_bwd_fused_launch(const FusedArgs& a0, int grid, int Cin, int dy_pro, hipStream_t st) {
  if (a0.Wi == 128) return thin_bwd_fused_launch_w<128>(a0, grid, Cin, dy_pro, st);
  if (a0.Wi == 256) return thin_bwd_fused_launch_w<256>(a0, grid, Cin, dy_pro, st);
  return AVA_EINVAL;
}

// ---------------------------------------------------------------------------------------------------------
// dispatch (called from conv_dispatch.hip); AVA_EINVAL = shape not handled here
// ---------------------------------------------------------------------------------------------------------
template <typename K>
static int thin_set_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes) == hipSuccess ? AVA_OK : AVA_ELAUNCH;
}

template <int W>
static int conv3x3_thin_w(const ConvArgs& a0, int grid, int Cin, int Cout, int pro, int epi, hipStream_t st) {
  const dim3 block(2 * W);
  ConvArgs a = a0;
  a.ntiles = a.B * (a.Ho / THIN_TH);                     // workgroups beyond ntiles still write their (zero) partial row
  a.part_rows = grid;                                    // rows the caller sized; one resident wave is launched
  if (Cin == 8 && grid > ava_scale_grid(512)) grid = ava_scale_grid(512);                // 8 -> 1: two workgroups per CU are resident (measured -3.5 us)
  { const char* e = ava_env("AVA_THIN_GRID"); if (e && atoi(e) > 0 && atoi(e) < grid) grid = atoi(e); }
  if (Cin == 1 && Cout == 8) {
    if (grid > ava_scale_grid(1024)) grid = ava_scale_grid(1024);      // W = 128: four 256-thread workgroups per CU
    if (W == 256) {                                      // 512-thread workgroups: at most one resident wave
      static const int res = thin_resident(&thin_1to8_kernel<W, PRO_BN, EPI_FWD>, W, 0);
      if (grid > ava_scale_grid(res)) grid = ava_scale_grid(res);
    }
    if (a.act_bf16 && !(pro == PRO_BN && epi == EPI_FWD)) return AVA_EINVAL;   // bf16 activations: only the model's launches
    if (pro == PRO_BN && epi == EPI_FWD && a.act_bf16) hipLaunchKernelGGL((thin_1to8_kernel<W, PRO_BN, EPI_FWD, ava_bf16>), dim3(grid), block, 0, st, a);
    else if (pro == PRO_BN && epi == EPI_FWD) hipLaunchKernelGGL((thin_1to8_kernel<W, PRO_BN, EPI_FWD>), dim3(grid), block, 0, st, a);
    else if (epi == EPI_BWD) return ava_conv3x3_thin_perop(a, grid, W, Cin, pro, st);     // data-gradient forms: conv_thin_perop.hip
    else return AVA_EINVAL;
  } else if (Cin == 8 && Cout == 1) {
    if (W == 256 && grid > 256) grid = 256;              // 82 KB of LDS per workgroup: one per CU
#ifdef AVA_LAB
    static const int direct = [] { const char* e = ava_env("AVA_THIN_FWD_DIRECT"); return e ? atoi(e) : 1; }();
    if (W == 128 && direct == 0 && pro == PRO_BN && epi == EPI_SSE) {
      if constexpr (W == 128) {
        if (thin_ws_mode() != 0) {
          const size_t ws_lds = (size_t)(2 * THIN_IR * 130 * 8 + 96 + 8) * sizeof(float);
          static bool attr_ws = false;
          if (!attr_ws) {
            if (thin_set_lds(&thin_8to1_ws_kernel<PRO_BN, EPI_SSE, 256>, ws_lds) != AVA_OK) return AVA_ELAUNCH;
            attr_ws = true;
          }
          const int g = grid < 256 ? grid : 256;          // one workgroup per CU
          hipLaunchKernelGGL((thin_8to1_ws_kernel<PRO_BN, EPI_SSE, 256>), dim3(g), dim3(512), ws_lds, st, a);
          AVA_CHECK_LAUNCH();
          return AVA_OK;
        }
      }
    }
#endif
    if (pro == PRO_BN && epi == EPI_SSE) {
      static const int resident = thin_resident(&thin_8to1_direct_kernel<W, PRO_BN, EPI_SSE>, W, 0);
      int g = a.part_rows < resident ? a.part_rows : resident;   // at most one resident wave; rows beyond the grid are zero-filled
      if (g > ava_scale_grid(resident)) g = ava_scale_grid(resident);
      if (g > ava_scale_grid(512)) g = ava_scale_grid(512);                               // in-step A/B: 512 / 768 / 1023 workgroups -> 34.1 / 34.7 / 37.0 us
      { const char* e = ava_env("AVA_THIN_FWD_GRID"); if (e && atoi(e) >= 8 && atoi(e) < a.part_rows) g = atoi(e); }
      if (g > a.ntiles) g = a.ntiles;
      if (a.fold.wg_partials != nullptr) {
        // the training forward that also leaves convt7's weight-gradient partials and BatchNorm-backward sums behind (FOLD)
        if (a.epi_x == nullptr || a.out2 == nullptr || (a.fin.acc == nullptr && (a.fold.mean == nullptr || a.fold.invstd == nullptr)) ||
            (a.fold.acc_out == nullptr && a.fold.bn_partials == nullptr))
          return AVA_EINVAL;
        // one resident wave of THIS kernel (more registers and LDS than the plain forward: two workgroups per CU at W = 128,
        // one at W = 256; the smaller of the two activation types' answers so that both partition the tiles alike)
        static const int fold_resident = [] {
          const int rf = thin_resident(&thin_8to1_direct_fold_kernel<W, float>, W, 0);
          const int rb = thin_resident(&thin_8to1_direct_fold_kernel<W, ava_bf16>, W, 0);
          return rf < rb ? rf : rb;
        }();
        if (g > ava_scale_grid(fold_resident)) g = ava_scale_grid(fold_resident);
        if (a.act_bf16) hipLaunchKernelGGL((thin_8to1_direct_fold_kernel<W, ava_bf16>), dim3(g), block, 0, st, a);
        else hipLaunchKernelGGL((thin_8to1_direct_fold_kernel<W, float>), dim3(g), block, 0, st, a);
      }
      else if (a.act_bf16) hipLaunchKernelGGL((thin_8to1_direct_kernel<W, PRO_BN, EPI_SSE, ava_bf16>), dim3(g), block, 0, st, a);
      else hipLaunchKernelGGL((thin_8to1_direct_kernel<W, PRO_BN, EPI_SSE>), dim3(g), block, 0, st, a);
    }
    else if (a.act_bf16) return AVA_EINVAL;
    else if (epi == EPI_BWD) return ava_conv3x3_thin_perop(a, grid, W, Cin, pro, st);     // data-gradient forms: conv_thin_perop.hip
    else return AVA_EINVAL;
  } else {
    return AVA_EINVAL;
  }
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

int ava_conv3x3_thin(const ConvArgs& a0, int grid, int Cin, int Cout, int mode, int pro, int epi, hipStream_t st) {
  if (mode != MODE_S1 || !thin_width_ok(a0.Wo) || a0.Ho % THIN_TH != 0) return AVA_EINVAL;
  if (a0.Wo == 128) return conv3x3_thin_w<128>(a0, grid, Cin, Cout, pro, epi, st);
  return conv3x3_thin_w<256>(a0, grid, Cin, Cout, pro, epi, st);
}

