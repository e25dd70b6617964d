// Matrix-core version of the 3x3 gather convolutions (forward and backward-data of every conv / convT
// layer whose channel counts are multiples of 8), gfx950, exact fp32 (v_mfma_f32_16x16x4_f32).
//
// Implicit GEMM per output tile:  D[cout][pixel] += A[cout][k] * B[k][pixel],  k = (tap, cin).
//   * B comes from the NHWC input tile staged in LDS (prologue already applied, zero padding in the
//     post-BatchNorm domain).  One ds_read_b128 per lane fetches 4 consecutive input channels of
//     "its" pixel for "its" tap; element j of that read is the K-slot of MFMA j, so one LDS read feeds
//     4 MFMAs (the K order inside a 16-wide chunk is permuted identically for A and B).
//   * A (the weights, <= 6912 floats per layer) lives in registers for the whole kernel: lane
//     (m = lane&15, kg = lane>>4) keeps G[k = 16c+4kg+j][cout = 16mt+m] for every chunk c, j, mt.
//   * D: lane holds 4 consecutive output channels of one pixel -> one 16-byte NHWC store per lane,
//     1 KiB contiguous per wave instruction; bias/ReLU/BN statistics fused in registers.
// A workgroup (4 waves) walks a list of output tiles; each wave owns groups of 16 consecutive output
// pixels of a row (for the x2-upsampling pattern: of one output-parity class, so the tap set is uniform).
#include <stdlib.h>
#include <type_traits>
#include "conv_mfma.h"

// MSPLIT (layers with two cout tiles): waves 0,2 compute cout tile 0 and waves 1,3 tile 1, each for half of the pixel
// groups -- half the weight registers per wave (112 -> 56 for 24 -> 24 channels), so two workgroups fit on a CU.
// PAIR (stride 1, 8 output channels): two output rows share one MFMA tile (PairFrag) -- a third fewer MFMAs.
// ACT: storage type of the activations read (forward input / saved activation in2 / raw x of EPI_BWD).  This kernel's
// OUTPUT is always fp32: in the library it only runs conv7's forward, whose output feeds the fully connected layers.
template <int CIN, int COUT, int MODE, int PRO, int EPI, int TW, int TH, bool MSPLIT, bool PAIR, typename ACT>
__global__ __launch_bounds__(256, (MSPLIT ? 2 : 1)) void conv3x3_mfma_kernel(const ConvArgs a) {
  using TIN = typename std::conditional<PRO == PRO_BN, ACT, float>::type;
  static_assert(!PAIR || (MODE == MODE_S1 && COUT == 8 && !MSPLIT && TH % 2 == 0), "PAIR: stride 1, 8 output channels");
  using G = Geom<MODE, TW, TH>;
  constexpr int IR = G::IR, IC = G::IC;
  constexpr int MTA = (COUT + 15) / 16;             // cout tiles of the layer
  constexpr int MT = MSPLIT ? 1 : MTA;              // cout tiles of this wave
  static_assert(!MSPLIT || MTA == 2, "MSPLIT deals exactly two cout tiles to the wave pairs");
  constexpr int NCLS = n_classes<MODE>();
  extern __shared__ __align__(16) float smem[];
  float* tile = smem;                       // [IR*IC*CIN]
  float* coef = smem + IR * IC * CIN;       // [3][32]
  float* red = coef + 96;                   // [4][2*16*MTA]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = lane & 15, kg = lane >> 4;
  const int mtb = MSPLIT ? (wave & 1) : 0;  // first cout tile of this wave
  const int wp = wave >> 1;                 // MSPLIT: which half of the pixel groups
  __shared__ double accvals[64];            // consumer prologue scratch (bn_coef_from_acc)
  if (a.fin.acc != nullptr) {
    bn_coef_from_acc(coef, accvals, a.fin);   // BatchNorm finalised here from the producer's accumulated sums
  } else if (t < 96) {
    const int which = t >> 5, c = t & 31;
    const float* src = which == 0 ? a.pa : (which == 1 ? a.pb : a.pc);
    coef[t] = (src != nullptr && c < CIN) ? src[c] : 0.f;
  }

  typename std::conditional<PAIR, PairFrag<CIN, IC>, ClassFrag<CIN, COUT, MODE, 0, IC, MT>>::type f0;
  ClassFrag<CIN, COUT, MODE, (NCLS > 1 ? 1 : 0), IC, MT> f1;
  ClassFrag<CIN, COUT, MODE, (NCLS > 1 ? 2 : 0), IC, MT> f2;
  ClassFrag<CIN, COUT, MODE, (NCLS > 1 ? 3 : 0), IC, MT> f3;
  constexpr int SP = MODE == MODE_DOWN ? 2 : 1;            // input pixels per output pixel along x
  // output offset (floats) of this lane inside a 16-pixel group: pixel n (every 2nd pixel for UP), channels 4kg..
  // (PAIR: rows 4kg.. of the tile are output row kg >> 1, channels 4 (kg & 1)..)
  const int lane_out = PAIR ? ((kg >> 1) * a.Wo + n) * COUT + 4 * (kg & 1) : (MODE == MODE_UP ? 2 * n : n) * COUT + 4 * kg;
  const int cq = PAIR ? 4 * (kg & 1) : 4 * kg;    // first output channel of this lane inside its cout tile

  // tile origin (image, output row/col, input row/col) of tile `tl`
  auto origin = [&](int tl, int& b, int& oy0, int& ox0, int& gy0, int& gx0) {
    b = tl / (a.tiles_y * a.tiles_x);
    const int rem = tl - b * (a.tiles_y * a.tiles_x);
    oy0 = (rem / a.tiles_x) * TH;
    ox0 = (rem % a.tiles_x) * TW;
    if (MODE == MODE_S1) { gy0 = oy0 - 1; gx0 = ox0 - 1; }
    else if (MODE == MODE_DOWN) { gy0 = 2 * oy0 - 1; gx0 = 2 * ox0 - 1; }
    else { gy0 = oy0 / 2; gx0 = ox0 / 2; }
  };
  constexpr int GROUPS = PAIR ? (TH / 2) * (TW / 16) : ((MODE == MODE_UP) ? TH * TW / 16 : TH * (TW / 16));
  constexpr int GPW = MSPLIT ? GROUPS / 2 : GROUPS / 4;     // groups per wave
  static_assert(GROUPS % 4 == 0 && (!MSPLIT || MODE != MODE_UP || GROUPS % 8 == 0),
                "tile must give every wave the same number of pixel groups");
  // gi-th group of this wave.  UP pattern: the parity class (g & 3) must stay a compile-time function of gi.
  auto group_of = [&](int gi) -> int {
    if (!MSPLIT) return wave * GPW + gi;
    if (MODE == MODE_UP) return 8 * (gi >> 2) + 4 * wp + (gi & 3);
    return wp + 2 * gi;
  };
  // scalar offset (floats) of pixel group g's first pixel relative to the tile's first output pixel
  //   S1/DOWN: group g = (row g / GPR, 16 columns from 16*(g % GPR));  UP: g = (row pair g >> 2, parity class g & 3)
  auto group_out = [&](int g) -> int {
    if (MODE == MODE_UP) return (((2 * (g >> 2)) + ((g & 3) >> 1)) * a.Wo + (g & 1)) * COUT;
    constexpr int GPR = TW / 16;
    return (((PAIR ? 2 : 1) * (g / GPR)) * a.Wo + 16 * (g % GPR)) * COUT;
  };
  avaf4 exn[EPI == EPI_BWD ? GPW * MT : 1];
  auto load_ex = [&](int b, int oy0, int ox0) {
    const ACT* __restrict__ xb = ava_as<ACT>(a.epi_x) + (((size_t)b * a.Ho + oy0) * a.Wo + ox0) * COUT;
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int cb = 16 * (mtb + mt) + cq;
        // lanes whose 4-channel slot lies beyond COUT (COUT = 8 or 24) re-read slot 0: stays in bounds
        exn[gi * MT + mt] = ava_ld4<ACT>(xb + group_out(group_of(gi)) + (cb < COUT ? lane_out + 16 * (mtb + mt) : lane_out - 4 * kg));
      }
  };
  TileStager<CIN, PRO, IR, IC, false, 256, TIN, ACT> stg;
  stg.init();
  TileWalk walk(a.ntiles);
  if (walk.valid()) {
    int b, oy0, ox0, gy0, gx0;
    origin(walk.cur, b, oy0, ox0, gy0, gx0);
    stg.load(a.in, a.in2, b, a.Hi, a.Wi, gy0, gx0);
    if (EPI == EPI_BWD) load_ex(b, oy0, ox0);
  }
  // Weights and epilogue constants are fetched AFTER the first tile's loads were issued: both round trips to
  // memory overlap instead of following each other at the start of every workgroup.
  constexpr bool BF16M = std::is_same<ACT, ava_bf16>::value;     // bf16 arithmetic: weights rounded to bfloat16 (TileStager rounds the BatchNorm output)
  if (!AVA_DBG_BIT(a, 8)) f0.init(a.G, lane, SP * n * CIN, mtb, BF16M);
  if (NCLS > 1) { f1.init(a.G, lane, n * CIN, mtb, BF16M); f2.init(a.G, lane, n * CIN, mtb, BF16M); f3.init(a.G, lane, n * CIN, mtb, BF16M); }
  // epilogue constants for this lane's 4 output channels per cout tile
  float bias[MT][4];
  float s1[MT][4], s2[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = 16 * (mtb + mt) + cq + r;
      bias[mt][r] = (EPI == EPI_FWD && co < COUT) ? a.bias[co] : 0.f;
      s1[mt][r] = s2[mt][r] = 0.f;
      // retire these loop-invariant loads before the tile loop (see ClassFrag::init)
      asm volatile("" ::"v"(bias[mt][r]));
    }

  for (; walk.valid(); walk.advance()) {
    int b, oy0, ox0, gy0, gx0;
    origin(walk.cur, b, oy0, ox0, gy0, gx0);
    __syncthreads();                       // previous tile fully consumed (and coef[] visible on the first pass)
    if (!AVA_DBG_BIT(a, 2)) stg.store(tile, coef);                 // s_waitcnt vmcnt(0): retires the prefetch (and exn) of this tile
    if (EPI == EPI_BWD) ava_wait_vm0(exn);
    __syncthreads();
    // uniform (scalar) part of the addresses of this tile
    const size_t tile_pix = ((size_t)b * a.Ho + oy0) * a.Wo + ox0;
    float* __restrict__ obase = a.out != nullptr ? a.out + tile_pix * COUT : nullptr;
    // EPI_BWD: the x values for the BatchNorm-backward sums of the NEXT tile travel with its prefetch
    // (asynchronous loads, retired by the same wait); `ex` holds the current tile's, moved over here.
    avaf4 ex[GPW * MT];
    if (EPI == EPI_BWD) {
#pragma unroll
      for (int i = 0; i < GPW * MT; ++i) ex[i] = exn[i];
    }
    if (walk.has_next() && !AVA_DBG_BIT(a, 2)) {  // next tile's loads stay in flight during the MFMAs below
      int nb, noy0, nox0, ngy0, ngx0;
      origin(walk.next(), nb, noy0, nox0, ngy0, ngx0);
      stg.load(a.in, a.in2, nb, a.Hi, a.Wi, ngy0, ngx0);
      if (EPI == EPI_BWD) load_ex(nb, noy0, nox0);
    }

#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {   // fully unrolled: hipcc drains vmcnt(0) in front of a loop that stores
      const int g = group_of(gi);
      f32x4 acc[2][MT];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[h][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (AVA_DBG_BIT(a, 1)) {
      } else if (MODE == MODE_UP) {
        const int cls = gi & 3, r = g >> 2;       // == g & 3 (both group mappings keep the class in the low bits of gi)
        const float* px = tile + r * IC * CIN;
        if (cls == 0) f0.run(px, acc);
        else if (cls == 1) f1.run(px, acc);
        else if (cls == 2) f2.run(px, acc);
        else f3.run(px, acc);
      } else {
        constexpr int GPR = TW / 16;
        constexpr int S = (MODE == MODE_S1 && !PAIR) ? 1 : 2;        // PAIR: a group is a pair of rows
        constexpr int SX = MODE == MODE_DOWN ? 2 : 1;
        f0.run(tile + (S * (g / GPR) * IC + SX * 16 * (g % GPR)) * CIN, acc);
      }
      const int gout = group_out(g) + lane_out;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int cb = 16 * (mtb + mt) + cq;
        if (cb < COUT) {
          f32x4 v = acc[0][mt] + acc[1][mt];
          if (EPI == EPI_FWD) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float x = fmaxf(v[r] + bias[mt][r], 0.f);      // every matrix-core forward layer has a ReLU
              v[r] = x;
              s1[mt][r] += x;
              s2[mt][r] = fmaf(x, x, s2[mt][r]);
            }
          } else {   // EPI_BWD
            const avaf4 xr = ex[gi * MT + mt];
            const float xv[4] = {xr[0], xr[1], xr[2], xr[3]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              s1[mt][r] += v[r];
              s2[mt][r] = fmaf(v[r], xv[r], s2[mt][r]);      // raw x: centred after the loop
            }
          }
          if (obase != nullptr && !AVA_DBG_BIT(a, 4))
            *reinterpret_cast<float4*>(obase + gout + 16 * (mtb + mt)) = make_float4(v[0], v[1], v[2], v[3]);
          if (EPI == EPI_FWD && MODE == MODE_S1 && !PAIR && a.out2 != nullptr) {
            // second copy in NCHW order (the flatten order of the fully connected layer that follows, vae.py:224):
            // 16 lanes = 16 consecutive pixels of a row -> 64 contiguous bytes per channel
            constexpr int GPR = TW / 16;
            const int oy = oy0 + g / GPR, ox = ox0 + 16 * (g % GPR) + n;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              a.out2[(((size_t)b * COUT + cb + r) * a.Ho + oy) * a.Wo + ox] = v[r];
          }
        }
      }
    }
  }

  if (AVA_DBG_BIT(a, 16)) return;
  // ---- per-workgroup partial statistics: reduce over the 16 pixel lanes, then over the 4 waves ----
  __syncthreads();
  if (MSPLIT) {                              // a wave only fills its own cout tile: the other slots must read as 0
    if (t < 4 * 32 * MTA / 2) { red[t] = 0.f; red[t + 4 * 32 * MTA / 2] = 0.f; }
    __syncthreads();
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v1 = s1[mt][r], v2 = s2[mt][r];
      if (EPI == EPI_BWD) {
        // the hot loop accumulates sum g*x on RAW x; centred and scaled once per lane here: sum g*xhat = invstd * (sum g*x - mean * sum g)
        const int cc = 16 * (mtb + mt) + cq + r;
        const float mu = cc < COUT ? a.epi_mean[cc] : 0.f, is = cc < COUT ? a.epi_invstd[cc] : 0.f;
        v2 = fmaf(-mu, v1, v2) * is;
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { v1 += __shfl_xor(v1, o, 64); v2 += __shfl_xor(v2, o, 64); }
      if (PAIR) { v1 += __shfl_xor(v1, 32, 64); v2 += __shfl_xor(v2, 32, 64); }   // the two rows of a pair
      if (n == 0 && (!PAIR || kg < 2)) {
        const int co = 16 * (mtb + mt) + cq + r;
        red[wave * 32 * MTA + co] = v1;
        red[wave * 32 * MTA + 16 * MTA + co] = v2;
      }
    }
  __syncthreads();
  if (t < 2 * COUT && a.acc_out != nullptr) {
    const int which = t / COUT, co = t - which * COUT;
    const int idx = which * 16 * MTA + co;
    bn_acc_add(a.acc_out, which * 32 + co, (red[idx] + red[32 * MTA + idx]) + (red[64 * MTA + idx] + red[96 * MTA + idx]));
  } else if (t < 2 * COUT && a.partials != nullptr) {
    const int which = t / COUT, co = t - which * COUT;
    const int idx = which * 16 * MTA + co;
    a.partials[(size_t)blockIdx.x * 2 * COUT + t] =
        (red[idx] + red[32 * MTA + idx]) + (red[64 * MTA + idx] + red[96 * MTA + idx]);
    for (int r = gridDim.x + blockIdx.x; r < a.part_rows; r += gridDim.x)       // rows of workgroups not launched
      a.partials[(size_t)r * 2 * COUT + t] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------------
template <int CIN, int COUT, int MODE, int PRO, int EPI, int TW, int TH, typename ACT>
static int launch_mfma_t(const ConvArgs& a, int grid, hipStream_t st) {
  using G = Geom<MODE, TW, TH>;
  constexpr int MT = (COUT + 15) / 16;
  constexpr bool MSPLIT = MT == 2 && CIN >= 16;     // the register-bound shapes
  constexpr bool PAIR = MODE == MODE_S1 && COUT == 8;
  const size_t lds = (size_t)(G::IR * G::IC * CIN + 96 + 4 * 32 * MT) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<CIN, COUT, MODE, PRO, EPI, TW, TH, MSPLIT, PAIR, ACT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return AVA_ELAUNCH;
    attr_set = true;
  }
  ConvArgs b = a;
  b.tiles_y = a.Ho / TH;
  b.tiles_x = a.Wo / TW;
  b.ntiles = a.B * b.tiles_y * b.tiles_x;
  // one resident wave of workgroups; the partial rows of the workgroups not launched are zero-filled by the kernel
  static const int resident = ava_resident_grid(&conv3x3_mfma_kernel<CIN, COUT, MODE, PRO, EPI, TW, TH, MSPLIT, PAIR, ACT>, lds);
  b.part_rows = grid;
  if (grid > b.ntiles) grid = b.ntiles;
  if (grid > ava_scale_grid(resident)) grid = ava_scale_grid(resident);
  { const char* e = ava_env("AVA_GRID"); if (e) grid = atoi(e); if (grid > b.ntiles) grid = b.ntiles; if (grid > b.part_rows) grid = b.part_rows; }
  hipLaunchKernelGGL((conv3x3_mfma_kernel<CIN, COUT, MODE, PRO, EPI, TW, TH, MSPLIT, PAIR, ACT>), dim3(grid), dim3(256), lds, st, b);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

template <int CIN, int COUT, int MODE, int PRO, int EPI, int TW, int TH>
static int launch_mfma(const ConvArgs& a, int grid, hipStream_t st) {
  if (a.act_bf16) {
    // bf16 activations: this kernel writes fp32, so it may only stand in where the consumer reads fp32 (conv7)
    if (EPI == EPI_FWD && a.out2 == nullptr) return AVA_EINVAL;
    return launch_mfma_t<CIN, COUT, MODE, PRO, EPI, TW, TH, ava_bf16>(a, grid, st);
  }
  return launch_mfma_t<CIN, COUT, MODE, PRO, EPI, TW, TH, float>(a, grid, st);
}

#ifdef AVA_LAB
template <int CIN, int COUT, int MODE, int TW, int TH>
static int launch_mfma_pe(const ConvArgs& a, int grid, int pro, int epi, hipStream_t st) {
  if (pro == PRO_BN && epi == EPI_FWD) return launch_mfma<CIN, COUT, MODE, PRO_BN, EPI_FWD, TW, TH>(a, grid, st);
  if (pro == PRO_BWD && epi == EPI_BWD) return launch_mfma<CIN, COUT, MODE, PRO_BWD, EPI_BWD, TW, TH>(a, grid, st);
  if (pro == PRO_ID && epi == EPI_BWD) return launch_mfma<CIN, COUT, MODE, PRO_ID, EPI_BWD, TW, TH>(a, grid, st);
  return AVA_EINVAL;
}
#endif

// returns AVA_EINVAL when the shape has no matrix-core instantiation (caller falls back to the VALU kernel)
// `grid` = number of workgroups = number of partial rows (the caller's ava_conv_grid value)
int ava_conv3x3_mfma_ws(const ConvArgs& a, int grid, int Cin, int Cout, int mode, int pro, int epi, hipStream_t st);   // conv_ws.hip

int ava_conv3x3_up88_direct(const ConvArgs& a, int grid, int Cin, int Cout, int mode, int pro, int epi, hipStream_t st);

int ava_conv3x3_mfma(const ConvArgs& a, int grid, int Cin, int Cout, int mode, int pro, int epi, hipStream_t st) {
  if (epi == EPI_FWD && !a.relu) return AVA_EINVAL;        // the fused epilogue always applies the ReLU
  {                                                        // convt6 forward: direct packed-FMA kernel (conv_thin.hip)
    const int rc = ava_conv3x3_up88_direct(a, grid, Cin, Cout, mode, pro, epi, st);
    if (rc != AVA_EINVAL) return rc;
  }
  // forward layers run the wave-specialised kernel (conv_ws.hip; -27 us/step); AVA_CONV_WS=0 selects the plain one
  static const bool ws = [] { const char* e = ava_env("AVA_CONV_WS"); return e == nullptr || atoi(e) != 0; }();
  static const bool ws_bwd = [] { const char* e = ava_env("AVA_CONV_WS_BWD"); return e == nullptr || atoi(e) != 0; }();
  if (ws && a.out2 == nullptr && (epi == EPI_FWD || (epi == EPI_BWD && ws_bwd))) {
    const int rc = ava_conv3x3_mfma_ws(a, grid, Cin, Cout, mode, pro, epi, st);
    if (rc != AVA_EINVAL) return rc;
  }
#ifndef AVA_LAB
  // Everything else runs the wave-specialised kernel above.  The plain 256-thread kernel is compiled for the one
  // launch that needs its second output: conv7's forward, which also writes the NCHW-flatten copy fc1 reads.
  if (a.out2 != nullptr && Cin == 24 && Cout == 32 && mode == MODE_S1 && a.Wo % 16 == 0 && a.Ho % 8 == 0 && pro == PRO_BN && epi == EPI_FWD)
    return launch_mfma<24, 32, MODE_S1, PRO_BN, EPI_FWD, 16, 8>(a, grid, st);
  return AVA_EINVAL;
#else
#define AVA_MFMA_CASE(ci, co, md, tww, thh) \
  if (Cin == ci && Cout == co && mode == md && a.Wo % tww == 0 && a.Ho % thh == 0) return launch_mfma_pe<ci, co, md, tww, thh>(a, grid, pro, epi, st);
  AVA_MFMA_CASE(8, 8, MODE_DOWN, 32, 4)
  AVA_MFMA_CASE(8, 16, MODE_S1, 32, 8)
  AVA_MFMA_CASE(16, 16, MODE_DOWN, 32, 4)
  AVA_MFMA_CASE(16, 24, MODE_S1, 32, 4)
  AVA_MFMA_CASE(24, 24, MODE_DOWN, 16, 4)
  AVA_MFMA_CASE(24, 32, MODE_S1, 16, 8)
  AVA_MFMA_CASE(32, 24, MODE_S1, 16, 8)
  AVA_MFMA_CASE(24, 24, MODE_UP, 32, 8)
  AVA_MFMA_CASE(24, 16, MODE_S1, 32, 4)
  AVA_MFMA_CASE(16, 16, MODE_UP, 32, 8)
  AVA_MFMA_CASE(16, 8, MODE_S1, 32, 8)
  AVA_MFMA_CASE(8, 8, MODE_UP, 32, 8)
#undef AVA_MFMA_CASE
  return AVA_EINVAL;
#endif
}

// ================================================================================================
// weight / bias gradient on the matrix cores:  dG[(tap,ci)][co] = sum_pixels xhat(in-pos)[ci] * dU(pixel)[co]
//   M = (tap, ci) rows (A operand, read from the staged input tile), N = co (B operand, read from the
//   staged dU tile), K = output pixels, 4 consecutive pixels of a row per MFMA.
// Every wave sweeps a quarter of each tile's pixels into a full set of accumulators that lives in registers
// across ALL tiles of the workgroup; at the end the four waves are summed through LDS in a fixed order
// and the workgroup writes one partial row [9*CIN*COUT + COUT] (same format as the VALU kernel).
// The bias gradient is the column sum of the B fragments (one VALU add per LDS read).
// ================================================================================================
template <int CIN, int COUT, int MODE, int DYPRO, int TW, int TH, typename ACT>
__global__ __launch_bounds__(256) void conv3x3_wgrad_mfma_kernel(const WgradArgs a) {
  using G = Geom<MODE, TW, TH>;
  constexpr int IR = G::IR, IC = G::IC;
  constexpr int NT = (COUT + 15) / 16;
  constexpr int NCLS = n_classes<MODE>();
  constexpr int NW = 9 * CIN * COUT;
  extern __shared__ __align__(16) float smem[];
  float* xt = smem;                               // [IR*IC*CIN]
  float* dyt = xt + IR * IC * CIN;                // [TH*TW*COUT] (+16 pad: padded cout columns read past the end)
  float* cx = dyt + TH * TW * COUT + 16;          // [3][32]
  float* cd = cx + 96;                            // [3][32]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = lane & 15, kg = lane >> 4;
  if (t < 96) {
    const int which = t >> 5, c = t & 31;
    const float* sx = which == 0 ? a.xa : (which == 1 ? a.xb : nullptr);
    const float* sd = which == 0 ? a.da : (which == 1 ? a.db : a.dc);
    cx[t] = (sx != nullptr && c < CIN) ? sx[c] : 0.f;
    cd[t] = (sd != nullptr && c < COUT) ? sd[c] : 0.f;
  }
  if (t < 16) dyt[TH * TW * COUT + t] = 0.f;

  WClass<CIN, COUT, MODE, 0, IC> w0;
  WClass<CIN, COUT, MODE, (NCLS > 1 ? 1 : 0), IC> w1;
  WClass<CIN, COUT, MODE, (NCLS > 1 ? 2 : 0), IC> w2;
  WClass<CIN, COUT, MODE, (NCLS > 1 ? 3 : 0), IC> w3;
  w0.init(lane);
  if (NCLS > 1) { w1.init(lane); w2.init(lane); w3.init(lane); }
  float bsum[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bsum[nt] = 0.f;

  auto origin = [&](int tl, int& b, int& oy0, int& ox0, int& gy0, int& gx0) {
    b = tl / (a.tiles_y * a.tiles_x);
    const int rem = tl - b * (a.tiles_y * a.tiles_x);
    oy0 = (rem / a.tiles_x) * TH;
    ox0 = (rem % a.tiles_x) * TW;
    if (MODE == MODE_S1) { gy0 = oy0 - 1; gx0 = ox0 - 1; }
    else if (MODE == MODE_DOWN) { gy0 = 2 * oy0 - 1; gx0 = 2 * ox0 - 1; }
    else { gy0 = oy0 / 2; gx0 = ox0 / 2; }
  };
  TileStager<CIN, PRO_BN, IR, IC, false, 256, ACT, ACT> sx;       // x: activation
  TileStager<COUT, DYPRO, TH, TW, false, 256, float, ACT> sd;      // dy: fp32 gradient, dy2: saved activation
  sx.init();
  sd.init();
  TileWalk walk(a.ntiles);
  if (walk.valid()) {
    int b, oy0, ox0, gy0, gx0;
    origin(walk.cur, b, oy0, ox0, gy0, gx0);
    sx.load(a.x, nullptr, b, a.Hi, a.Wi, gy0, gx0);
    sd.load(a.dy, a.dy2, b, a.Ho, a.Wo, oy0, ox0);
  }
  for (; walk.valid(); walk.advance()) {
    __syncthreads();
    sx.store(xt, cx);
    sd.store(dyt, cd);
    __syncthreads();
    if (walk.has_next()) {
      int b, oy0, ox0, gy0, gx0;
      origin(walk.next(), b, oy0, ox0, gy0, gx0);
      sx.load(a.x, nullptr, b, a.Hi, a.Wi, gy0, gx0);
      sd.load(a.dy, a.dy2, b, a.Ho, a.Wo, oy0, ox0);
    }

    if (MODE == MODE_UP) {
      // wave <-> class row r (TH/2 == 4 rows); per class 16 columns c = 4*s + kg
      const int r = wave;
#pragma unroll 1
      for (int s = 0; s < TW / 8; ++s) {
        const int c = 4 * s + kg;
        const float* xa = xt + (r * IC + c) * CIN;
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
          const int py = cls >> 1, px = cls & 1;
          const float* bp = dyt + ((2 * r + py) * TW + 2 * c + px) * COUT + n;
          float bf[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) { bf[nt] = bp[16 * nt]; bsum[nt] += (16 * nt + n < COUT) ? bf[nt] : 0.f; }
          if (cls == 0) w0.step(xa, bf);
          else if (cls == 1) w1.step(xa, bf);
          else if (cls == 2) w2.step(xa, bf);
          else w3.step(xa, bf);
        }
      }
    } else {
      constexpr int S = MODE == MODE_S1 ? 1 : 2;
      constexpr int RPW = TH / 4;                 // rows per wave
#pragma unroll 1
      for (int rr = 0; rr < RPW; ++rr) {
        const int ty = wave * RPW + rr;
#pragma unroll 1
        for (int s = 0; s < TW / 4; ++s) {
          const int x = 4 * s + kg;
          const float* bp = dyt + (ty * TW + x) * COUT + n;
          float bf[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) { bf[nt] = bp[16 * nt]; bsum[nt] += (16 * nt + n < COUT) ? bf[nt] : 0.f; }
          w0.step(xt + ((S * ty) * IC + S * x) * CIN, bf);
        }
      }
    }
  }

  // ---- fixed-order reduction of the four waves through LDS, one partial row per workgroup ----------
  __syncthreads();
  float* wacc = smem;                             // [NW + COUT], aliases the tiles (all reads are done)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    bsum[nt] += __shfl_xor(bsum[nt], 16, 64);
    bsum[nt] += __shfl_xor(bsum[nt], 32, 64);
  }
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      w0.flush(wacc, lane, w == 0);
      if (NCLS > 1) { w1.flush(wacc, lane, w == 0); w2.flush(wacc, lane, w == 0); w3.flush(wacc, lane, w == 0); }
      if (kg == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int co = 16 * nt + n;
          if (co < COUT) wacc[NW + co] = (w == 0) ? bsum[nt] : wacc[NW + co] + bsum[nt];
        }
      }
    }
    __syncthreads();
  }
  float* prow = a.partials + (size_t)blockIdx.x * (NW + COUT);
  for (int e = t; e < NW + COUT; e += 256) prow[e] = wacc[e];
}

// the kernel's body as a device function of (workgroup index, workgroups of this layer): conv3x3_wgrad_split_kernel runs it
// for its whole grid, conv3x3_wgrad_pair_kernel runs two layers' bodies in one launch
template <int CIN, int COUT, int MODE, int DYPRO, int TW, int TH, typename ACT>
__device__ __forceinline__ void wgrad_split_body(const WgradArgs& a, const int wg, const int nwg) {
  using G = Geom<MODE, TW, TH>;
  constexpr int IR = G::IR, IC = G::IC;
  constexpr int NT = (COUT + 15) / 16;
  constexpr int NCLS = n_classes<MODE>();
  constexpr int NW = 9 * CIN * COUT;
  extern __shared__ __align__(16) float smem[];
  float* xt = smem;                               // [IR*IC*CIN]
  float* dyt = xt + IR * IC * CIN;                // [TH*TW*COUT] (+16 pad: padded cout columns read past the end)
  float* cx = dyt + TH * TW * COUT + 16;          // [3][32]
  float* cd = cx + 96;                            // [3][32]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = lane & 15, kg = lane >> 4;
  if (t < 16) dyt[TH * TW * COUT + t] = 0.f;

  WSplit<CIN, COUT, MODE, 0, IC> w0;
  WSplit<CIN, COUT, MODE, (NCLS > 1 ? 1 : 0), IC> w1;
  WSplit<CIN, COUT, MODE, (NCLS > 1 ? 2 : 0), IC> w2;
  WSplit<CIN, COUT, MODE, (NCLS > 1 ? 3 : 0), IC> w3;
  w0.init(lane, wave);
  if (NCLS > 1) { w1.init(lane, wave); w2.init(lane, wave); w3.init(lane, wave); }
  float bsum[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bsum[nt] = 0.f;

  auto origin = [&](int tl, int& b, int& oy0, int& ox0, int& gy0, int& gx0) {
    b = tl / (a.tiles_y * a.tiles_x);
    const int rem = tl - b * (a.tiles_y * a.tiles_x);
    oy0 = (rem / a.tiles_x) * TH;
    ox0 = (rem % a.tiles_x) * TW;
    if (MODE == MODE_S1) { gy0 = oy0 - 1; gx0 = ox0 - 1; }
    else if (MODE == MODE_DOWN) { gy0 = 2 * oy0 - 1; gx0 = 2 * ox0 - 1; }
    else { gy0 = oy0 / 2; gx0 = ox0 / 2; }
  };
  TileStager<CIN, PRO_BN, IR, IC, false, 256, ACT, ACT> sx;       // x: activation
  TileStager<COUT, DYPRO, TH, TW, false, 256, float, ACT> sd;      // dy: fp32 gradient, dy2: saved activation
  sx.init();
  sd.init();
  TileWalk walk(a.ntiles, true, wg, nwg);
  if (walk.valid()) {
    int b, oy0, ox0, gy0, gx0;
    origin(walk.cur, b, oy0, ox0, gy0, gx0);
    sx.load(a.x, nullptr, b, a.Hi, a.Wi, gy0, gx0);
    sd.load(a.dy, a.dy2, b, a.Ho, a.Wo, oy0, ox0);
  }
  // the prologue coefficients are requested BEHIND the first tile (in front of it they were a serial memory latency of
  // their own; now both are in flight together and the loop's first barrier covers the LDS writes)
  if (t < 96) {
    const int which = t >> 5, c = t & 31;
    const float* sx_ = which == 0 ? a.xa : (which == 1 ? a.xb : nullptr);
    const float* sd_ = which == 0 ? a.da : (which == 1 ? a.db : a.dc);
    cx[t] = (sx_ != nullptr && c < CIN) ? sx_[c] : 0.f;
    cd[t] = (sd_ != nullptr && c < COUT) ? sd_[c] : 0.f;
  }
  for (; walk.valid(); walk.advance()) {
    __syncthreads();
    sx.store(xt, cx);
    sd.store(dyt, cd);
    __syncthreads();
    if (walk.has_next()) {
      int b, oy0, ox0, gy0, gx0;
      origin(walk.next(), b, oy0, ox0, gy0, gx0);
      sx.load(a.x, nullptr, b, a.Hi, a.Wi, gy0, gx0);
      sd.load(a.dy, a.dy2, b, a.Ho, a.Wo, oy0, ox0);
    }

    if (MODE == MODE_UP) {
      // every wave sweeps all TH/2 x-space rows; per class 16 columns c = 4*s + kg
#pragma unroll 1
      for (int r = 0; r < TH / 2; ++r) {
        const bool mine = (r & 3) == wave;            // bias gradient: each dU pixel is counted by one wave
#pragma unroll 1
        for (int s = 0; s < TW / 8; ++s) {
          const int c = 4 * s + kg;
          const float* xa = xt + (r * IC + c) * CIN;
#pragma unroll
          for (int cls = 0; cls < 4; ++cls) {
            const int py = cls >> 1, px = cls & 1;
            const float* bp = dyt + ((2 * r + py) * TW + 2 * c + px) * COUT + n;
            float bf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { bf[nt] = bp[16 * nt]; bsum[nt] += (mine && 16 * nt + n < COUT) ? bf[nt] : 0.f; }
            if (cls == 0) w0.step(xa, bf);
            else if (cls == 1) w1.step(xa, bf);
            else if (cls == 2) w2.step(xa, bf);
            else w3.step(xa, bf);
          }
        }
      }
    } else {
      constexpr int S = MODE == MODE_S1 ? 1 : 2;
#pragma unroll 1
      for (int ty = 0; ty < TH; ++ty) {
        const bool mine = (ty & 3) == wave;
#pragma unroll 1
        for (int s = 0; s < TW / 4; ++s) {
          const int x = 4 * s + kg;
          const float* bp = dyt + (ty * TW + x) * COUT + n;
          float bf[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) { bf[nt] = bp[16 * nt]; bsum[nt] += (mine && 16 * nt + n < COUT) ? bf[nt] : 0.f; }
          w0.step(xt + ((S * ty) * IC + S * x) * CIN, bf);
        }
      }
    }
  }

  // ---- every wave's rows are final: gather them in LDS, add the bias sums of the four waves, one partial row ----
  __syncthreads();
  float* wacc = smem;                             // [NW + 4*COUT], aliases the tiles (all reads are done)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    bsum[nt] += __shfl_xor(bsum[nt], 16, 64);
    bsum[nt] += __shfl_xor(bsum[nt], 32, 64);
  }
  w0.flush(wacc, lane);
  if (NCLS > 1) { w1.flush(wacc, lane); w2.flush(wacc, lane); w3.flush(wacc, lane); }
  if (kg == 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = 16 * nt + n;
      if (co < COUT) wacc[NW + wave * COUT + co] = bsum[nt];
    }
  }
  __syncthreads();
  float* prow = a.partials + (size_t)wg * (NW + COUT);
  for (int e = t; e < NW; e += 256) prow[e] = wacc[e];
  if (t < COUT) prow[NW + t] = (wacc[NW + t] + wacc[NW + COUT + t]) + (wacc[NW + 2 * COUT + t] + wacc[NW + 3 * COUT + t]);
}

template <int CIN, int COUT, int MODE, int DYPRO, int TW, int TH, typename ACT>
__global__ __launch_bounds__(256) void conv3x3_wgrad_split_kernel(const WgradArgs a) {
  wgrad_split_body<CIN, COUT, MODE, DYPRO, TW, TH, ACT>(a, (int)blockIdx.x, (int)gridDim.x);
}

// Two layers' weight-gradient kernels in ONE launch: workgroups [0, grid_a) run layer A's body, the rest layer B's.  The
// four 16 x 16 layers' weight gradients are independent of the data-gradient chain once their dU exists, and each is a
// latency-bound launch of one or two tiles per workgroup; issued as pairs (conv7 + conv6, convt2 + convt1) behind both
// data-gradient kernels they overlap each other.  Every workgroup computes exactly what it computed in its own launch
// (same tiles, same partial row), so the gradients are bit-identical.
template <int CA, int OA, int MA, int PA, int TWA, int THA, int CB, int OB, int MB, int PB, int TWB, int THB, typename ACT>
__global__ __launch_bounds__(256) void conv3x3_wgrad_pair_kernel(const WgradArgs a, const WgradArgs b, const int grid_a) {
  if ((int)blockIdx.x < grid_a) wgrad_split_body<CA, OA, MA, PA, TWA, THA, ACT>(a, (int)blockIdx.x, grid_a);
  else wgrad_split_body<CB, OB, MB, PB, TWB, THB, ACT>(b, (int)blockIdx.x - grid_a, (int)gridDim.x - grid_a);
}

// tile geometry, workgroups (= partial rows written) and LDS bytes of one layer's weight-gradient kernel
template <int CIN, int COUT, int MODE, int DYPRO, int TW, int TH, typename ACT>
static int wgrad_plan(const WgradArgs& a, int grid, WgradArgs* b, size_t* lds_out) {
  using G = Geom<MODE, TW, TH>;
  // layers with many (tap, cin) rows deal the M tiles out to the waves (fewer registers, no cross-wave reduction)
  constexpr bool SPLIT = CIN >= 24 && COUT > 16;   // measured: 24->24, 24->32, 32->24 gain 10-55 %, 16-channel sides lose
  const auto kernel = SPLIT ? &conv3x3_wgrad_split_kernel<CIN, COUT, MODE, DYPRO, TW, TH, ACT>
                            : &conv3x3_wgrad_mfma_kernel<CIN, COUT, MODE, DYPRO, TW, TH, ACT>;
  const size_t tiles_f = (size_t)G::IR * G::IC * CIN + TH * TW * COUT + 16 + 192;
  const size_t red_f = (size_t)9 * CIN * COUT + 4 * COUT;
  const size_t lds = (tiles_f > red_f ? tiles_f : red_f) * sizeof(float);
  *lds_out = lds;
  *b = a;
  b->tiles_y = a.Ho / TH;
  b->tiles_x = a.Wo / TW;
  b->ntiles = a.B * b->tiles_y * b->tiles_x;
  if (grid > b->ntiles) grid = b->ntiles;
  static const int resident = ava_resident_grid(kernel, lds);
  if (grid > ava_scale_grid(resident)) grid = ava_scale_grid(resident);          // one resident wave of workgroups = partial rows written
  // the split kernels (the 16 x 16 layers): one workgroup per CU.  Each workgroup ends with a 5-7 k-float partial row,
  // which at one or two tiles per workgroup costs more than the tiles (same-box A/B of the step: 512 -> 256 workgroups
  // -13 us, 384 +-0, 128 +13 us)
  if (SPLIT && grid > ava_scale_grid(256)) grid = ava_scale_grid(256);
  { const char* e = ava_env("AVA_WGRID"); if (e && atoi(e) > 0 && atoi(e) < grid) grid = atoi(e); }
  return grid;
}

template <int CIN, int COUT, int MODE, int DYPRO, int TW, int TH, typename ACT>
static int launch_wgrad_mfma_t(const WgradArgs& a, int grid, hipStream_t st) {
  constexpr bool SPLIT = CIN >= 24 && COUT > 16;
  const auto kernel = SPLIT ? &conv3x3_wgrad_split_kernel<CIN, COUT, MODE, DYPRO, TW, TH, ACT>
                            : &conv3x3_wgrad_mfma_kernel<CIN, COUT, MODE, DYPRO, TW, TH, ACT>;
  WgradArgs b;
  size_t lds;
  grid = wgrad_plan<CIN, COUT, MODE, DYPRO, TW, TH, ACT>(a, grid, &b, &lds);
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return AVA_ELAUNCH;
    attr_set = true;
  }
  if (a.partials == nullptr) return grid;        // row-count query (ava_conv_wgrad_rows)
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, st, b);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

// the pair launch: each layer keeps the grid (tile partition, partial rows) of its own launch
template <int CA, int OA, int MA, int PA, int TWA, int THA, int CB, int OB, int MB, int PB, int TWB, int THB, typename ACT>
static int launch_wgrad_pair_t(const WgradArgs& a, int grid_a, const WgradArgs& b, int grid_b, hipStream_t st) {
  static_assert(CA >= 24 && OA > 16 && CB >= 24 && OB > 16, "pairs of split kernels only");
  const auto kernel = &conv3x3_wgrad_pair_kernel<CA, OA, MA, PA, TWA, THA, CB, OB, MB, PB, TWB, THB, ACT>;
  WgradArgs pa, pb;
  size_t lds_a, lds_b;
  grid_a = wgrad_plan<CA, OA, MA, PA, TWA, THA, ACT>(a, grid_a, &pa, &lds_a);
  grid_b = wgrad_plan<CB, OB, MB, PB, TWB, THB, ACT>(b, grid_b, &pb, &lds_b);
  const size_t lds = lds_a > lds_b ? lds_a : lds_b;
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return AVA_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kernel, dim3(grid_a + grid_b), dim3(256), lds, st, pa, pb, grid_a);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

int ava_conv3x3_wgrad_mfma_pair(const WgradArgs& a, int grid_a, const WgradCall& ca, const WgradArgs& b, int grid_b,
                                const WgradCall& cb, hipStream_t st) {
  auto is = [](const WgradCall& c, const WgradArgs& w, int ci, int co, int md, int pro, int tw, int th) {
    return c.Cin == ci && c.Cout == co && c.mode == md && c.dy_pro == pro && w.Wo % tw == 0 && w.Ho % th == 0;
  };
  // encoder: conv7 (24 -> 32, given dU) + conv6 (24 -> 24, stride 2)
  if (is(ca, a, 24, 32, MODE_S1, PRO_ID, 16, 8) && is(cb, b, 24, 24, MODE_DOWN, PRO_BWD, 16, 4)) {
    if (a.act_bf16) return launch_wgrad_pair_t<24, 32, MODE_S1, PRO_ID, 16, 8, 24, 24, MODE_DOWN, PRO_BWD, 16, 4, ava_bf16>(a, grid_a, b, grid_b, st);
    return launch_wgrad_pair_t<24, 32, MODE_S1, PRO_ID, 16, 8, 24, 24, MODE_DOWN, PRO_BWD, 16, 4, float>(a, grid_a, b, grid_b, st);
  }
  // decoder: convt2 (24 -> 24, x2) + convt1 (32 -> 24)
  if (is(ca, a, 24, 24, MODE_UP, PRO_BWD, 32, 8) && is(cb, b, 32, 24, MODE_S1, PRO_BWD, 16, 8)) {
    if (a.act_bf16) return launch_wgrad_pair_t<24, 24, MODE_UP, PRO_BWD, 32, 8, 32, 24, MODE_S1, PRO_BWD, 16, 8, ava_bf16>(a, grid_a, b, grid_b, st);
    return launch_wgrad_pair_t<24, 24, MODE_UP, PRO_BWD, 32, 8, 32, 24, MODE_S1, PRO_BWD, 16, 8, float>(a, grid_a, b, grid_b, st);
  }
  return AVA_EINVAL;
}

template <int CIN, int COUT, int MODE, int DYPRO, int TW, int TH>
static int launch_wgrad_mfma(const WgradArgs& a, int grid, hipStream_t st) {
  if (a.act_bf16) return launch_wgrad_mfma_t<CIN, COUT, MODE, DYPRO, TW, TH, ava_bf16>(a, grid, st);
  return launch_wgrad_mfma_t<CIN, COUT, MODE, DYPRO, TW, TH, float>(a, grid, st);
}

int ava_conv3x3_wgrad_mfma(const WgradArgs& a, int grid, int Cin, int Cout, int mode, int dy_pro, hipStream_t st) {
#define AVA_WGM_CASE(ci, co, md, tww, thh)                                                     \
  if (Cin == ci && Cout == co && mode == md && a.Wo % tww == 0 && a.Ho % thh == 0) {                                    \
    if (dy_pro == PRO_BWD) return launch_wgrad_mfma<ci, co, md, PRO_BWD, tww, thh>(a, grid, st); \
    if (dy_pro == PRO_ID) return launch_wgrad_mfma<ci, co, md, PRO_ID, tww, thh>(a, grid, st);   \
    return AVA_EINVAL;                                                                         \
  }
  AVA_WGM_CASE(8, 8, MODE_DOWN, 32, 4)
  AVA_WGM_CASE(8, 16, MODE_S1, 32, 8)
  AVA_WGM_CASE(16, 16, MODE_DOWN, 32, 4)
  AVA_WGM_CASE(16, 24, MODE_S1, 32, 4)
  AVA_WGM_CASE(24, 24, MODE_DOWN, 16, 4)
  AVA_WGM_CASE(24, 32, MODE_S1, 16, 8)
  AVA_WGM_CASE(32, 24, MODE_S1, 16, 8)
  AVA_WGM_CASE(24, 24, MODE_UP, 32, 8)
  AVA_WGM_CASE(24, 16, MODE_S1, 32, 4)
  AVA_WGM_CASE(16, 16, MODE_UP, 32, 8)
  AVA_WGM_CASE(16, 8, MODE_S1, 32, 8)
  AVA_WGM_CASE(8, 8, MODE_UP, 32, 8)
#undef AVA_WGM_CASE
  return AVA_EINVAL;
}
