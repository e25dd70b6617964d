// Version-0 VALU kernels of the 3x3 gather convolutions -- LAB BUILD ONLY (make lab, -DAVA_LAB).
// Superseded on the model's path by the matrix-core kernels (conv_mfma.hip, conv_ws.hip, conv_fused.hip) and the
// packed-FMA kernels of the 1/8-channel layers (conv_thin.hip); kept as the A/B baseline of the first rows of
// DESIGN.md section 5 (AVA_CONV_IMPL=valu).  The product library does not contain them.
//
// Version-0 kernels (VALU): one workgroup of 256 threads walks a list of output tiles; the input
// tile (with halo) is staged through registers into LDS with the prologue applied, every thread
// owns one output pixel x all output channels; the weights are wave-uniform and are fetched with
// scalar loads.  Per-channel statistics are accumulated in registers across all tiles of the
// workgroup and written once as a deterministic partial row (no float atomics).
#include <stdlib.h>
#include <string.h>
#include "../conv_common.h"


// acc[co] += sum_ci px[ci] * Gt[ci][co]   (Gt wave-uniform -> scalar loads)
template <int CIN, int COUT>
__device__ __forceinline__ void tap_fma(float (&acc)[COUT], const float* __restrict__ px,
                                        const float* __restrict__ Gt) {
  if constexpr (CIN % 4 == 0) {
#pragma unroll
    for (int c4 = 0; c4 < CIN; c4 += 4) {
      const float4 v = *reinterpret_cast<const float4*>(px + c4);
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = fmaf(vv[j], Gt[(c4 + j) * COUT + co], acc[co]);
    }
  } else {
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      const float v = px[ci];
#pragma unroll
      for (int co = 0; co < COUT; ++co) acc[co] = fmaf(v, Gt[ci * COUT + co], acc[co]);
    }
  }
}

template <int CIN, int COUT, int MODE, int PRO, int EPI, int TW>
__global__ __launch_bounds__(256) void conv3x3_kernel(const ConvArgs a) {
  using G = Geom<MODE, TW>;
  constexpr int TH = G::TH, IR = G::IR, IC = G::IC;
  extern __shared__ __align__(16) float smem[];
  float* tile = smem;                       // [IR*IC*CIN]
  float* coef = smem + IR * IC * CIN;       // [3][32] prologue coefficients
  float* red = coef + 96;                   // [4][2*COUT] cross-wave reduction

  const int t = threadIdx.x;
  if (t < 96) {
    const int which = t >> 5, c = t & 31;
    const float* src = which == 0 ? a.pa : (which == 1 ? a.pb : a.pc);
    coef[t] = (src != nullptr && c < CIN) ? src[c] : 0.f;
  }
  float s1[COUT], s2[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) s1[co] = s2[co] = 0.f;

  // thread -> output pixel inside the tile
  int ty, tx;
  if (MODE == MODE_UP) {
    const int w = t >> 6, l = t & 63;
    const int py = w >> 1, px = w & 1;
    ty = 2 * (l / (TW / 2)) + py;
    tx = 2 * (l % (TW / 2)) + px;
  } else {
    ty = t / TW;
    tx = t % TW;
  }
  const float* __restrict__ Gw = a.G;

  for (TileWalk walk(a.ntiles); walk.valid(); walk.advance()) {
    const int tl = walk.cur;
    const int b = tl / (a.tiles_y * a.tiles_x);
    const int rem = tl - b * (a.tiles_y * a.tiles_x);
    const int oy0 = (rem / a.tiles_x) * TH, ox0 = (rem % a.tiles_x) * TW;
    int gy0, gx0;
    if (MODE == MODE_S1) { gy0 = oy0 - 1; gx0 = ox0 - 1; }
    else if (MODE == MODE_DOWN) { gy0 = 2 * oy0 - 1; gx0 = 2 * ox0 - 1; }
    else { gy0 = oy0 / 2; gx0 = ox0 / 2; }
    __syncthreads();   // previous tile fully consumed (also orders the coef[] fill on the first pass)
    stage_tile<CIN, PRO, IR, IC>(tile, a.in, a.in2, coef, b, a.Hi, a.Wi, gy0, gx0);
    __syncthreads();

    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;

    if (MODE == MODE_S1 || MODE == MODE_DOWN) {
      constexpr int S = MODE == MODE_S1 ? 1 : 2;
      const float* base = tile + ((S * ty) * IC + S * tx) * CIN;
#pragma unroll 1
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll 1
        for (int kx = 0; kx < 3; ++kx)
          tap_fma<CIN, COUT>(acc, base + (ky * IC + kx) * CIN, Gw + (ky * 3 + kx) * CIN * COUT);
    } else {
      // oy = 2r+py: py==0 -> ky=1 (row r); py==1 -> ky=0 (row r+1), ky=2 (row r)
      const int py = ty & 1, px = tx & 1, r = ty >> 1, c = tx >> 1;
      const int nky = py ? 2 : 1, nkx = px ? 2 : 1;     // wave-uniform
#pragma unroll 1
      for (int iy = 0; iy < nky; ++iy) {
        const int ky = py ? 2 * iy : 1;
        const int lr = r + ((py && iy == 0) ? 1 : 0);
#pragma unroll 1
        for (int ix = 0; ix < nkx; ++ix) {
          const int kx = px ? 2 * ix : 1;
          const int lc = c + ((px && ix == 0) ? 1 : 0);
          tap_fma<CIN, COUT>(acc, tile + (lr * IC + lc) * CIN, Gw + (ky * 3 + kx) * CIN * COUT);
        }
      }
    }

    // ---- epilogue -------------------------------------------------------------------------
    const int oy = oy0 + ty, ox = ox0 + tx;
    const size_t opix = ((size_t)b * a.Ho + oy) * a.Wo + ox;
    if (EPI == EPI_FWD) {
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        float v = acc[co] + a.bias[co];
        if (a.relu) v = fmaxf(v, 0.f);
        acc[co] = v;
        s1[co] += v;
        s2[co] = fmaf(v, v, s2[co]);
      }
    } else if (EPI == EPI_BWD) {
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        const float xh = (a.epi_x[opix * COUT + co] - a.epi_mean[co]) * a.epi_invstd[co];
        s1[co] += acc[co];
        s2[co] = fmaf(acc[co], xh, s2[co]);
      }
    } else {   // EPI_SSE, COUT == 1
      const float v = acc[0] + a.bias[0];
      acc[0] = v;
      if (a.epi_x != nullptr) {             // decode-only calls have no target
        const float r = v - a.epi_x[opix];
        a.out2[opix] = a.prec * r;
        s1[0] = fmaf(r, r, s1[0]);
      }
    }
    if (a.out != nullptr) {
      if constexpr (COUT % 4 == 0) {
#pragma unroll
        for (int co = 0; co < COUT; co += 4)
          *reinterpret_cast<float4*>(a.out + opix * COUT + co) = make_float4(acc[co], acc[co + 1], acc[co + 2], acc[co + 3]);
      } else {
#pragma unroll
        for (int co = 0; co < COUT; ++co) a.out[opix * COUT + co] = acc[co];
      }
    }
  }

  // ---- deterministic per-workgroup partial statistics ------------------------------------------
  __syncthreads();
  const int w = t >> 6, l = t & 63;
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
    const float r1 = wave_sum(s1[co]);
    const float r2 = wave_sum(s2[co]);
    if (l == 0) { red[w * 2 * COUT + co] = r1; red[w * 2 * COUT + COUT + co] = r2; }
  }
  __syncthreads();
  if (t < 2 * COUT && a.partials != nullptr)
    a.partials[(size_t)blockIdx.x * 2 * COUT + t] =
        (red[t] + red[2 * COUT + t]) + (red[4 * COUT + t] + red[6 * COUT + t]);
}

// ------------------------------------------------------------------------------------------------
// weight / bias gradient: dG[t][ci][co] = sum_pixels xhat(in-pos(t)) * dU(out-pos), db[co] = sum dU
// ------------------------------------------------------------------------------------------------

template <int CIN, int COUT, int MODE, int DYPRO, int TW>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const WgradArgs a) {
  using G = Geom<MODE, TW>;
  constexpr int TH = G::TH, IR = G::IR, IC = G::IC;
  constexpr int BI = CIN < 4 ? CIN : 4, BO = COUT < 4 ? COUT : 4;
  constexpr int NI = CIN / BI, NO = COUT / BO;
  constexpr int NBLK = 9 * NI * NO;
  constexpr int NBT = (NBLK + 255) / 256;                    // blocks per thread
  constexpr int SL = NBLK >= 256 ? 1 : 256 / NBLK;           // pixel slices
  constexpr int NW = 9 * CIN * COUT;
  extern __shared__ __align__(16) float smem[];
  float* xt = smem;                         // [IR*IC*CIN]
  float* dyt = xt + IR * IC * CIN;          // [256*COUT]
  float* cx = dyt + 256 * COUT;             // [3][32] coefficients for x
  float* cd = cx + 96;                      // [3][32] coefficients for dy
  const int t = threadIdx.x;
  if (t < 96) {
    const int which = t >> 5, c = t & 31;
    const float* sx = which == 0 ? a.xa : (which == 1 ? a.xb : nullptr);
    const float* sd = which == 0 ? a.da : (which == 1 ? a.db : a.dc);
    cx[t] = (sx != nullptr && c < CIN) ? sx[c] : 0.f;
    cd[t] = (sd != nullptr && c < COUT) ? sd[c] : 0.f;
  }
  float acc[NBT][BI][BO];
#pragma unroll
  for (int n = 0; n < NBT; ++n)
#pragma unroll
    for (int i = 0; i < BI; ++i)
#pragma unroll
      for (int o = 0; o < BO; ++o) acc[n][i][o] = 0.f;
  float bacc = 0.f;                                           // bias: thread -> (co = t % COUT, part = t / COUT)
  constexpr int BPARTS = 256 / COUT;

  const int slice = SL > 1 ? t / NBLK : 0;
  const bool active = SL > 1 ? (t < NBLK * SL) : true;

  for (TileWalk walk(a.ntiles); walk.valid(); walk.advance()) {
    const int tl = walk.cur;
    const int b = tl / (a.tiles_y * a.tiles_x);
    const int rem = tl - b * (a.tiles_y * a.tiles_x);
    const int oy0 = (rem / a.tiles_x) * TH, ox0 = (rem % a.tiles_x) * TW;
    int gy0, gx0;
    if (MODE == MODE_S1) { gy0 = oy0 - 1; gx0 = ox0 - 1; }
    else if (MODE == MODE_DOWN) { gy0 = 2 * oy0 - 1; gx0 = 2 * ox0 - 1; }
    else { gy0 = oy0 / 2; gx0 = ox0 / 2; }
    __syncthreads();
    stage_tile<CIN, PRO_BN, IR, IC>(xt, a.x, nullptr, cx, b, a.Hi, a.Wi, gy0, gx0);
    stage_tile<COUT, DYPRO, TH, TW>(dyt, a.dy, a.dy2, cd, b, a.Ho, a.Wo, oy0, ox0);
    __syncthreads();

    // bias gradient
    if (t < BPARTS * COUT) {
      const int co = t % COUT, part = t / COUT;
      for (int p = part; p < 256; p += BPARTS) bacc += dyt[p * COUT + co];
    }
    if (active) {
#pragma unroll
      for (int n = 0; n < NBT; ++n) {
        const int blk = (SL > 1 ? t % NBLK : t) + n * 256;
        if (blk < NBLK) {
          const int tap = blk / (NI * NO), r2 = blk - tap * (NI * NO);
          const int ib = r2 / NO, ob = r2 - ib * NO;
          const int ky = tap / 3, kx = tap - 3 * ky;
          if (MODE == MODE_UP) {
            // tap (ky,kx) touches output pixels of parity py = (ky!=1), px = (kx!=1)
            const int py = ky != 1, px = kx != 1;
            const int dr = ky == 0 ? 1 : 0, dc = kx == 0 ? 1 : 0;
            constexpr int NP = (TH / 2) * (TW / 2);
            for (int p = slice; p < NP; p += SL) {
              const int r = p / (TW / 2), c = p - r * (TW / 2);
              const float* xp = xt + ((r + dr) * IC + (c + dc)) * CIN + ib * BI;
              const float* dp = dyt + ((2 * r + py) * TW + 2 * c + px) * COUT + ob * BO;
              float xv[BI], dv[BO];
#pragma unroll
              for (int i = 0; i < BI; ++i) xv[i] = xp[i];
#pragma unroll
              for (int o = 0; o < BO; ++o) dv[o] = dp[o];
#pragma unroll
              for (int i = 0; i < BI; ++i)
#pragma unroll
                for (int o = 0; o < BO; ++o) acc[n][i][o] = fmaf(xv[i], dv[o], acc[n][i][o]);
            }
          } else {
            constexpr int S = MODE == MODE_S1 ? 1 : 2;
            for (int p = slice; p < 256; p += SL) {
              const int r = p / TW, c = p - r * TW;
              const float* xp = xt + ((S * r + ky) * IC + S * c + kx) * CIN + ib * BI;
              const float* dp = dyt + p * COUT + ob * BO;
              float xv[BI], dv[BO];
#pragma unroll
              for (int i = 0; i < BI; ++i) xv[i] = xp[i];
#pragma unroll
              for (int o = 0; o < BO; ++o) dv[o] = dp[o];
#pragma unroll
              for (int i = 0; i < BI; ++i)
#pragma unroll
                for (int o = 0; o < BO; ++o) acc[n][i][o] = fmaf(xv[i], dv[o], acc[n][i][o]);
            }
          }
        }
      }
    }
  }

  // ---- cross-slice reduction through LDS, then one partial row per workgroup ----------------------
  __syncthreads();
  float* red = smem;                                          // reuse: [SL][NW] then [BPARTS][COUT]
  if (active) {
#pragma unroll
    for (int n = 0; n < NBT; ++n) {
      const int blk = (SL > 1 ? t % NBLK : t) + n * 256;
      if (blk < NBLK) {
        const int tap = blk / (NI * NO), r2 = blk - tap * (NI * NO);
        const int ib = r2 / NO, ob = r2 - ib * NO;
#pragma unroll
        for (int i = 0; i < BI; ++i)
#pragma unroll
          for (int o = 0; o < BO; ++o)
            red[slice * NW + (tap * CIN + ib * BI + i) * COUT + ob * BO + o] = acc[n][i][o];
      }
    }
  }
  float* bred = red + SL * NW;
  if (t < BPARTS * COUT) bred[t] = bacc;                       // index = part*COUT + co
  __syncthreads();
  float* prow = a.partials + (size_t)blockIdx.x * (NW + COUT);
  for (int e = t; e < NW; e += 256) {
    float s = 0.f;
#pragma unroll 1
    for (int sl = 0; sl < SL; ++sl) s += red[sl * NW + e];
    prow[e] = s;
  }
  if (t < COUT) {
    float s = 0.f;
    for (int p = 0; p < BPARTS; ++p) s += bred[p * COUT + t];
    prow[NW + t] = s;
  }
}


template <int CIN, int COUT, int MODE, int PRO, int EPI, int TW>
static int launch_conv(const ConvArgs& a, int grid, hipStream_t st) {
  using G = Geom<MODE, TW>;
  const size_t lds = (size_t)(G::IR * G::IC * CIN + 96 + 8 * COUT) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<CIN, COUT, MODE, PRO, EPI, TW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return AVA_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_kernel<CIN, COUT, MODE, PRO, EPI, TW>), dim3(grid), dim3(256), lds, st, a);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

template <int CIN, int COUT, int MODE, int TW>
static int launch_conv_pe(const ConvArgs& a, int grid, int pro, int epi, hipStream_t st) {
  if (epi == EPI_SSE) {
    if constexpr (COUT == 1) { if (pro == PRO_BN) return launch_conv<CIN, COUT, MODE, PRO_BN, EPI_SSE, TW>(a, grid, st); }
    return AVA_EINVAL;
  }
  if (pro == PRO_BN && epi == EPI_FWD) return launch_conv<CIN, COUT, MODE, PRO_BN, EPI_FWD, TW>(a, grid, st);
  if (pro == PRO_BWD && epi == EPI_BWD) return launch_conv<CIN, COUT, MODE, PRO_BWD, EPI_BWD, TW>(a, grid, st);
  if (pro == PRO_ID && epi == EPI_BWD) return launch_conv<CIN, COUT, MODE, PRO_ID, EPI_BWD, TW>(a, grid, st);
  return AVA_EINVAL;
}


int ava_conv3x3_valu(const ConvArgs& a, int grid, int Cin, int Cout, int mode, int pro, int epi, int tw, hipStream_t st) {
#define AVA_CONV_CASE(ci, co, md, tww) \
  if (Cin == ci && Cout == co && mode == md && tw == tww) return launch_conv_pe<ci, co, md, tww>(a, grid, pro, epi, st);
  // encoder forward / decoder backward-data shapes
  AVA_CONV_CASE(1, 8, MODE_S1, 32)
  AVA_CONV_CASE(8, 8, MODE_DOWN, 32)
  AVA_CONV_CASE(8, 16, MODE_S1, 32)
  AVA_CONV_CASE(16, 16, MODE_DOWN, 32)
  AVA_CONV_CASE(16, 24, MODE_S1, 32)
  AVA_CONV_CASE(24, 24, MODE_DOWN, 16)
  AVA_CONV_CASE(24, 32, MODE_S1, 16)
  // decoder forward / encoder backward-data shapes
  AVA_CONV_CASE(32, 24, MODE_S1, 16)
  AVA_CONV_CASE(24, 24, MODE_UP, 32)
  AVA_CONV_CASE(24, 16, MODE_S1, 32)
  AVA_CONV_CASE(16, 16, MODE_UP, 32)
  AVA_CONV_CASE(16, 8, MODE_S1, 32)
  AVA_CONV_CASE(8, 8, MODE_UP, 32)
  AVA_CONV_CASE(8, 1, MODE_S1, 32)
#undef AVA_CONV_CASE
  return AVA_EINVAL;
}

template <int CIN, int COUT, int MODE, int DYPRO, int TW>
static int launch_wgrad(const WgradArgs& a, int grid, hipStream_t st) {
  using G = Geom<MODE, TW>;
  constexpr int BI = CIN < 4 ? CIN : 4, BO = COUT < 4 ? COUT : 4;
  constexpr int NBLK = 9 * (CIN / BI) * (COUT / BO);
  constexpr int SL = NBLK >= 256 ? 1 : 256 / NBLK;
  size_t main_f = (size_t)G::IR * G::IC * CIN + 256 * COUT + 192;
  size_t red_f = (size_t)SL * 9 * CIN * COUT + 256;
  const size_t lds = (main_f > red_f ? main_f : red_f) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wgrad_kernel<CIN, COUT, MODE, DYPRO, TW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return AVA_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_wgrad_kernel<CIN, COUT, MODE, DYPRO, TW>), dim3(grid), dim3(256), lds, st, a);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}


int ava_conv3x3_wgrad_valu(const WgradArgs& a, int grid, int Cin, int Cout, int mode, int dy_pro, int tw, hipStream_t st) {
#define AVA_WG_CASE(ci, co, md, tww)                                                          \
  if (Cin == ci && Cout == co && mode == md && tw == tww) {                                   \
    if (dy_pro == PRO_BWD) return launch_wgrad<ci, co, md, PRO_BWD, tww>(a, grid, st);        \
    if (dy_pro == PRO_ID) return launch_wgrad<ci, co, md, PRO_ID, tww>(a, grid, st);          \
    return AVA_EINVAL;                                                                        \
  }
  AVA_WG_CASE(1, 8, MODE_S1, 32)
  AVA_WG_CASE(8, 8, MODE_DOWN, 32)
  AVA_WG_CASE(8, 16, MODE_S1, 32)
  AVA_WG_CASE(16, 16, MODE_DOWN, 32)
  AVA_WG_CASE(16, 24, MODE_S1, 32)
  AVA_WG_CASE(24, 24, MODE_DOWN, 16)
  AVA_WG_CASE(24, 32, MODE_S1, 16)
  AVA_WG_CASE(32, 24, MODE_S1, 16)
  AVA_WG_CASE(24, 24, MODE_UP, 32)
  AVA_WG_CASE(24, 16, MODE_S1, 32)
  AVA_WG_CASE(16, 16, MODE_UP, 32)
  AVA_WG_CASE(16, 8, MODE_S1, 32)
  AVA_WG_CASE(8, 8, MODE_UP, 32)
  AVA_WG_CASE(8, 1, MODE_S1, 32)
#undef AVA_WG_CASE
  return AVA_EINVAL;
}
