// Fused backward of one conv / convT layer on the matrix cores (gfx950, exact fp32 v_mfma_f32_16x16x4_f32).
//
// The separate kernels of conv_mfma.hip read the upstream gradient twice: the backward-data kernel stages
// dU = relu'(y) * BatchNorm-backward(g) to convolve it with the flipped weights, and the weight-gradient kernel
// stages the same dU again to correlate it with the layer input.  Here ONE workgroup stages, per tile,
//   * the dU window (with the halo the data gradient needs; prologue PRO_BWD or PRO_ID applied on the way in), and
//   * the layer-input window x_n = BatchNorm(x) (with the halo the weight gradient needs, zero padded),
// and runs both implicit GEMMs off those two LDS tiles:
//   dx[p][ci]        = sum_{tap,co} dU[p (+) tap][co] * Gb[tap][co][ci]      (+ sum dx, sum dx*xhat for BatchNorm l)
//   dG[tap][ci][co] += sum_{p in tile interior} x_n[p (+) tap][ci] * dU[p][co],   db[co] += sum_p dU[p][co]
// so g, y and x are fetched from HBM once per layer instead of twice (reference: autograd's conv backward,
// ava/models/vae.py:347-353 loss.backward()).  Tiles are indexed in the LOW-resolution space of the layer
// (S1: both tensors; stride-2 conv: the dU side; stride-2 convT: the x side).
#include <stdlib.h>
#include <type_traits>
#include "conv_mfma.h"
#include "conv_fused.h"
#include "conv_recomp.h"

// Round 5: the fp32-MFMA kernels of this file are LAB-ONLY (make lab).  Every layer shape has a limb instantiation
// (conv_fused_limb.hip) whose tiles divide every supported image size (W in {128, 256}, H a multiple of 128: low-resolution
// sides >= 16 against tiles of at most 32 x 8), so no supported size reached them any more; a size they would have served now
// reports "no fused kernel" (grid 0) and the caller runs the separate data-gradient and weight-gradient kernels.  What the
// product library keeps of this file is the dispatch: thin (1 <-> 8-channel) kernels, then the limb kernels.
#ifdef AVA_LAB
template <int LMODE, int TW, int TH>
struct FGeom {
  // x window [XR x XC], dU window [DR x DC], dU interior [DIH x DIW] at offset (DOFF, DOFF), dx region [OH x OW]
  static constexpr int XR = LMODE == MODE_S1 ? TH + 2 : (LMODE == MODE_DOWN ? 2 * TH + 1 : TH + 1);
  static constexpr int XC = LMODE == MODE_S1 ? TW + 2 : (LMODE == MODE_DOWN ? 2 * TW + 1 : TW + 1);
  static constexpr int DR = LMODE == MODE_S1 ? TH + 2 : (LMODE == MODE_DOWN ? TH + 1 : 2 * TH + 1);
  static constexpr int DC = LMODE == MODE_S1 ? TW + 2 : (LMODE == MODE_DOWN ? TW + 1 : 2 * TW + 1);
  static constexpr int DOFF = LMODE == MODE_DOWN ? 0 : 1;
  static constexpr int OH = LMODE == MODE_DOWN ? 2 * TH : TH;
  static constexpr int OW = LMODE == MODE_DOWN ? 2 * TW : TW;
};

// ACT: storage type of the activations x (layer input) and dy2 (saved output); dy and dx are fp32 gradients.
template <int CI, int CO, int LMODE, int DYPRO, int TW, int TH, int MINW, typename ACT>
__global__ __launch_bounds__(256, MINW) void conv3x3_bwd_fused_kernel(const FusedArgs a) {
  using FG = FGeom<LMODE, TW, TH>;
  constexpr int XR = FG::XR, XC = FG::XC, DR = FG::DR, DC = FG::DC, DOFF = FG::DOFF;
  constexpr int BMODE = LMODE == MODE_S1 ? MODE_S1 : (LMODE == MODE_DOWN ? MODE_UP : MODE_DOWN);   // gather pattern of dx
  constexpr int MT = (CI + 15) / 16;        // dx channel tiles
  constexpr int NT = (CO + 15) / 16;        // dG column tiles
  constexpr int NW = 9 * CI * CO;
  constexpr int BCLS = n_classes<BMODE>(), WCLS = n_classes<LMODE>();
  // stride-1 layers with 8 input channels: the data gradient has 8 output channels -> two dx rows per MFMA tile
  constexpr bool PAIR = LMODE == MODE_S1 && CI == 8 && TH % 2 == 0;
  extern __shared__ __align__(16) float smem[];
  float* xt = smem;                          // [XR*XC*CI]   BatchNorm(x), zero padded
  float* dut = xt + XR * XC * CI;            // [DR*DC*CO]   dU, zero outside the image (+16 floats of zero pad)
  float* cx = dut + DR * DC * CO + 16;       // [3][32]
  float* cd = cx + 96;                       // [3][32]
  float* red = cd + 96;                      // [4][32*MT]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = lane & 15, kg = lane >> 4;
  if (t < 96) {
    const int which = t >> 5, c = t & 31;
    const float* sx = which == 0 ? a.xa : (which == 1 ? a.xb : nullptr);
    const float* sd = which == 0 ? a.da : (which == 1 ? a.db : a.dc);
    cx[t] = (sx != nullptr && c < CI) ? sx[c] : 0.f;
    cd[t] = (sd != nullptr && c < CO) ? sd[c] : 0.f;
  }
  if (t < 16) dut[DR * DC * CO + t] = 0.f;

  // ---- backward-data fragments: the flipped/packed weights stay in registers for the whole kernel ----
  constexpr int SPB = BMODE == MODE_DOWN ? 2 : 1;
  typename std::conditional<PAIR, PairFrag<CO, DC>, ClassFrag<CO, CI, BMODE, 0, DC>>::type f0;
  ClassFrag<CO, CI, BMODE, (BCLS > 1 ? 1 : 0), DC> f1;
  ClassFrag<CO, CI, BMODE, (BCLS > 1 ? 2 : 0), DC> f2;
  ClassFrag<CO, CI, BMODE, (BCLS > 1 ? 3 : 0), DC> f3;
  constexpr bool BF16M = std::is_same<ACT, ava_bf16>::value;     // bf16 arithmetic: weights rounded to bfloat16
  f0.init(a.Gb, lane, SPB * n * CO, 0, BF16M);
  if (BCLS > 1) { f1.init(a.Gb, lane, n * CO, 0, BF16M); f2.init(a.Gb, lane, n * CO, 0, BF16M); f3.init(a.Gb, lane, n * CO, 0, BF16M); }
  const int lane_out = PAIR ? ((kg >> 1) * a.Wi + n) * CI + 4 * (kg & 1) : (BMODE == MODE_UP ? 2 * n : n) * CI + 4 * kg;
  const int cq = PAIR ? 4 * (kg & 1) : 4 * kg;    // first dx channel of this lane inside its channel tile
  float s1[MT][4], s2[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s1[mt][r] = s2[mt][r] = 0.f;
    }

  // ---- weight-gradient accumulators (persist over all tiles of this workgroup) ----
  WClass<CI, CO, LMODE, 0, XC> w0;
  WClass<CI, CO, LMODE, (WCLS > 1 ? 1 : 0), XC> w1;
  WClass<CI, CO, LMODE, (WCLS > 1 ? 2 : 0), XC> w2;
  WClass<CI, CO, LMODE, (WCLS > 1 ? 3 : 0), XC> w3;
  w0.init(lane);
  if (WCLS > 1) { w1.init(lane); w2.init(lane); w3.init(lane); }
  float bsum[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bsum[nt] = 0.f;

  // tile -> image, low-resolution origin, window origins
  auto origin = [&](int tl, int& b, int& y0, int& x0) {
    b = tl / (a.tiles_y * a.tiles_x);
    const int rem = tl - b * (a.tiles_y * a.tiles_x);
    y0 = (rem / a.tiles_x) * TH;
    x0 = (rem % a.tiles_x) * TW;
  };
  auto x_origin = [&](int y0, int x0, int& gy, int& gx) {
    if (LMODE == MODE_S1) { gy = y0 - 1; gx = x0 - 1; }
    else if (LMODE == MODE_DOWN) { gy = 2 * y0 - 1; gx = 2 * x0 - 1; }
    else { gy = y0; gx = x0; }
  };
  auto d_origin = [&](int y0, int x0, int& gy, int& gx) {
    if (LMODE == MODE_S1) { gy = y0 - 1; gx = x0 - 1; }
    else if (LMODE == MODE_DOWN) { gy = y0; gx = x0; }
    else { gy = 2 * y0 - 1; gx = 2 * x0 - 1; }
  };
  TileStager<CI, PRO_BN, XR, XC, false, 256, ACT, ACT> sx;
  TileStager<CO, DYPRO, DR, DC, false, 256, float, ACT> sd;
  sx.init();
  sd.init();
  auto prefetch = [&](int tl) {
    int b, y0, x0, gy, gx;
    origin(tl, b, y0, x0);
    x_origin(y0, x0, gy, gx);
    sx.load(a.x, nullptr, b, a.Hi, a.Wi, gy, gx);
    d_origin(y0, x0, gy, gx);
    sd.load(a.dy, a.dy2, b, a.Ho, a.Wo, gy, gx);
  };

  // pixel groups of the dx region: 16 consecutive pixels of a row (UP pattern: of one parity class)
  constexpr int CB = FG::OW / (BMODE == MODE_UP ? 32 : 16);          // column blocks
  constexpr int GROUPS = PAIR ? (FG::OH / 2) * CB : (BMODE == MODE_UP ? 4 * (FG::OH / 2) * CB : FG::OH * CB);
  constexpr int GPW = GROUPS / 4;
  static_assert(GROUPS % 4 == 0 && (BMODE != MODE_UP || GPW % 4 == 0), "tile must split evenly over the 4 waves");
  // offset (floats, relative to the dx region's first pixel) and LDS pixel base of group g
  auto group_out = [&](int g) -> int {
    if (BMODE == MODE_UP) {
      const int cls = g & 3, rest = g >> 2, r = rest / CB, cb = rest % CB;
      return ((2 * r + (cls >> 1)) * a.Wi + 32 * cb + (cls & 1)) * CI;
    }
    return (((PAIR ? 2 : 1) * (g / CB)) * a.Wi + 16 * (g % CB)) * CI;
  };

  // raw x at this lane's dx pixels (BatchNorm-backward sums).  Loaded one tile ahead, AFTER the tile's data-gradient
  // phase has consumed the previous values: the lines were requested by the window prefetch a moment earlier, so
  // this hits L2, and the loads ride under the weight-gradient phase.
  avaf4 ex[GPW * MT];
  auto load_ex = [&](int tl) {
    int b, y0, x0;
    origin(tl, b, y0, x0);
    const int oy0 = LMODE == MODE_DOWN ? 2 * y0 : y0, ox0 = LMODE == MODE_DOWN ? 2 * x0 : x0;
    const ACT* __restrict__ xb = ava_as<ACT>(a.x) + (((size_t)b * a.Hi + oy0) * a.Wi + ox0) * CI;
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int cb4 = 16 * mt + cq;
        // lanes whose 4-channel slot lies beyond CI re-read slot 0: stays in bounds
        ex[gi * MT + mt] = ava_ld4<ACT>(
            xb + group_out(wave * GPW + gi) + (cb4 < CI ? lane_out + 16 * mt : lane_out - 4 * kg));
      }
  };

  TileWalk walk(a.ntiles);
  if (walk.valid()) { prefetch(walk.cur); load_ex(walk.cur); }
  for (; walk.valid(); walk.advance()) {
    int b, y0, x0;
    origin(walk.cur, b, y0, x0);
    __syncthreads();                       // previous tile fully consumed (and cx/cd visible on the first pass)
    sx.store(xt, cx);
    sd.store(dut, cd);
    __syncthreads();
    const int oy0 = LMODE == MODE_DOWN ? 2 * y0 : y0, ox0 = LMODE == MODE_DOWN ? 2 * x0 : x0;
    const size_t tile_pix = ((size_t)b * a.Hi + oy0) * a.Wi + ox0;
    float* __restrict__ obase = a.dx + tile_pix * CI;
    if (walk.has_next()) prefetch(walk.next());      // stays in flight during both matrix-core phases below

    // ---- phase 1: data gradient of the tile + BatchNorm-backward sums ----
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      const int g = wave * GPW + gi;
      f32x4 acc[2][MT];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[h][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (BMODE == MODE_UP) {
        const int cls = gi & 3, rest = g >> 2, r = rest / CB, cb = rest % CB;   // GPW % 4 == 0: cls is compile-time
        const float* px = dut + (r * DC + 16 * cb) * CO;
        if (cls == 0) f0.run(px, acc);
        else if (cls == 1) f1.run(px, acc);
        else if (cls == 2) f2.run(px, acc);
        else f3.run(px, acc);
      } else {
        constexpr int S = (BMODE == MODE_S1 && !PAIR) ? 1 : 2;        // PAIR: a group is a pair of dx rows
        constexpr int SX = BMODE == MODE_DOWN ? 2 : 1;
        f0.run(dut + (S * (g / CB) * DC + SX * 16 * (g % CB)) * CO, acc);
      }
      const int gout = group_out(g) + lane_out;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int cb4 = 16 * mt + cq;
        if (cb4 < CI) {
          const f32x4 v = acc[0][mt] + acc[1][mt];
          const avaf4 xr = ex[gi * MT + mt];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s1[mt][r] += v[r];
            s2[mt][r] = fmaf(v[r], xr[r], s2[mt][r]);          // raw x: centred after the loop
          }
          *reinterpret_cast<float4*>(obase + gout + 16 * mt) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }

    if (walk.has_next()) load_ex(walk.next());

    // ---- phase 2: weight / bias gradient over the tile's interior dU pixels ----
    if (LMODE == MODE_UP) {
      // x-space rows r = wave + 4 rr, columns c = 4 s + kg; the four output-parity classes of each x pixel
#pragma unroll 1
      for (int rr = 0; rr < TH / 4; ++rr) {
        const int r = wave + 4 * rr;
#pragma unroll 1
        for (int s = 0; s < TW / 4; ++s) {
          const int c = 4 * s + kg;
          const float* xa = xt + (r * XC + c) * CI;
#pragma unroll
          for (int cls = 0; cls < 4; ++cls) {
            const int py = cls >> 1, px = cls & 1;
            const float* bp = dut + ((2 * r + py + DOFF) * DC + 2 * c + px + DOFF) * CO + n;
            float bf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { bf[nt] = bp[16 * nt]; bsum[nt] += (16 * nt + n < CO) ? bf[nt] : 0.f; }
            if (cls == 0) w0.step(xa, bf);
            else if (cls == 1) w1.step(xa, bf);
            else if (cls == 2) w2.step(xa, bf);
            else w3.step(xa, bf);
          }
        }
      }
    } else {
      constexpr int S = LMODE == MODE_S1 ? 1 : 2;
      constexpr int RPW = TH / 4;                 // dU rows per wave
#pragma unroll 1
      for (int rr = 0; rr < RPW; ++rr) {
        const int ty = wave * RPW + rr;
#pragma unroll 1
        for (int s = 0; s < TW / 4; ++s) {
          const int x = 4 * s + kg;
          const float* bp = dut + ((ty + DOFF) * DC + x + DOFF) * CO + n;
          float bf[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) { bf[nt] = bp[16 * nt]; bsum[nt] += (16 * nt + n < CO) ? bf[nt] : 0.f; }
          w0.step(xt + ((S * ty) * XC + S * x) * CI, bf);
        }
      }
    }
  }

  // ---- BatchNorm-backward partial sums: over the 16 pixel lanes, then over the 4 waves (fixed order) ----
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v1 = s1[mt][r], v2 = s2[mt][r];
      {
        // the hot loop accumulates sum g*x on RAW x; centred and scaled once per lane here: sum g*xhat = invstd * (sum g*x - mean * sum g)
        const int cc = 16 * mt + cq + r;
        const float mu = cc < CI ? a.mean[cc] : 0.f, is = cc < CI ? a.invstd[cc] : 0.f;
        v2 = fmaf(-mu, v1, v2) * is;
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { v1 += __shfl_xor(v1, o, 64); v2 += __shfl_xor(v2, o, 64); }
      if (PAIR) { v1 += __shfl_xor(v1, 32, 64); v2 += __shfl_xor(v2, 32, 64); }   // the two rows of a pair
      if (n == 0 && (!PAIR || kg < 2)) {
        const int ci = 16 * mt + cq + r;
        red[wave * 32 * MT + ci] = v1;
        red[wave * 32 * MT + 16 * MT + ci] = v2;
      }
    }
  __syncthreads();
  if (t < 2 * CI) {
    const int which = t / CI, ci = t - which * CI;
    const int idx = which * 16 * MT + ci;
    a.bn_partials[(size_t)blockIdx.x * 2 * CI + t] =
        (red[idx] + red[32 * MT + idx]) + (red[64 * MT + idx] + red[96 * MT + idx]);
  }
  __syncthreads();

  // ---- weight-gradient partial row: the four waves summed through LDS in a fixed order ----
  float* wacc = smem;                             // [NW + CO], aliases the tiles (all reads are done)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    bsum[nt] += __shfl_xor(bsum[nt], 16, 64);
    bsum[nt] += __shfl_xor(bsum[nt], 32, 64);
  }
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      w0.flush(wacc, lane, w == 0);
      if (WCLS > 1) { w1.flush(wacc, lane, w == 0); w2.flush(wacc, lane, w == 0); w3.flush(wacc, lane, w == 0); }
      if (kg == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int co = 16 * nt + n;
          if (co < CO) wacc[NW + co] = (w == 0) ? bsum[nt] : wacc[NW + co] + bsum[nt];
        }
      }
    }
    __syncthreads();
  }
  float* prow = a.wg_partials + (size_t)blockIdx.x * (NW + CO);
  for (int e = t; e < NW + CO; e += 256) prow[e] = wacc[e];
}

// lab build: phase ablation (AVA_FDBG bits: 1 no data-gradient MFMA, 2 no weight-gradient MFMA, 4 no prologue / LDS write of
// the staging waves, 8 no tile loads, 16 no dx stores, 32 no raw-x loads of the matrix-core waves); timing only
#ifdef AVA_LAB
#define AVA_FABL(bit) ((a.dbg & (bit)) != 0)
#else
#define AVA_FABL(bit) false
#endif

// DUREC (convt6's backward): the upstream gradient dy (8 channels, full resolution) does not exist in memory -- it is
// convt7's data gradient, a 3x3 gather of the 1-channel seed a.dy, formed on the matrix cores by the staging waves as
// they build the dU tile (conv_recomp.h: DU1to8Stager).
template <int CI, int CO, int LMODE, int DYPRO, int TW, int TH, int MINW, typename ACT, bool DUREC = false>
__global__ __launch_bounds__(512, (CI * CO > 256 || (CI == 16 && CO == 16 && LMODE == MODE_DOWN)) ? 2 : 4) void conv3x3_bwd_fused_ws_kernel(const FusedArgs a) {
  using FG = FGeom<LMODE, TW, TH>;
  constexpr int XR = FG::XR, XC = FG::XC, DR = FG::DR, DC = FG::DC, DOFF = FG::DOFF;
  constexpr int BMODE = LMODE == MODE_S1 ? MODE_S1 : (LMODE == MODE_DOWN ? MODE_UP : MODE_DOWN);   // gather pattern of dx
  constexpr int MT = (CI + 15) / 16;        // dx channel tiles
  constexpr int NT = (CO + 15) / 16;        // dG column tiles
  constexpr int NW = 9 * CI * CO;
  constexpr int BCLS = n_classes<BMODE>(), WCLS = n_classes<LMODE>();
  // stride-1 layers with 8 input channels: the data gradient has 8 output channels -> two dx rows per MFMA tile
  constexpr bool PAIR = LMODE == MODE_S1 && CI == 8 && TH % 2 == 0;
  extern __shared__ __align__(16) float smem[];
  // two buffers of {x tile [XR*XC*CI], dU tile [DR*DC*CO] + 16 floats of zero pad}
  constexpr int XF = XR * XC * CI, BUF_F = XF + DR * DC * CO + 16;
  float* cx = smem + 2 * BUF_F;              // [3][32]
  float* cd = cx + 96;                       // [3][32]
  float* red = cd + 96;                      // [4][32*MT]
  float* xs = red + 4 * 32 * MT;             // DUREC: the staging waves' private seed windows
  static_assert(!DUREC || (CO == 8 && DYPRO == PRO_BWD && DR == 9), "the 1 -> 8 gather feeds a 9-row dU window of 8 channels");

  const int t = threadIdx.x, lane = t & 63, wave8 = t >> 6;
  const bool stager = wave8 < 4;             // waves 0-3 stage tiles, waves 4-7 run the two matrix-core phases
  const int wave = wave8 & 3;
  const int n = lane & 15, kg = lane >> 4;
  if (t < 32) smem[(t >> 4) * BUF_F + XF + DR * DC * CO + (t & 15)] = 0.f;

  // tile -> image, low-resolution origin, window origins
  auto origin = [&](int tl, int& b, int& y0, int& x0) {
    b = tl / (a.tiles_y * a.tiles_x);
    const int rem = tl - b * (a.tiles_y * a.tiles_x);
    y0 = (rem / a.tiles_x) * TH;
    x0 = (rem % a.tiles_x) * TW;
  };
  auto x_origin = [&](int y0, int x0, int& gy, int& gx) {
    if (LMODE == MODE_S1) { gy = y0 - 1; gx = x0 - 1; }
    else if (LMODE == MODE_DOWN) { gy = 2 * y0 - 1; gx = 2 * x0 - 1; }
    else { gy = y0; gx = x0; }
  };
  auto d_origin = [&](int y0, int x0, int& gy, int& gx) {
    if (LMODE == MODE_S1) { gy = y0 - 1; gx = x0 - 1; }
    else if (LMODE == MODE_DOWN) { gy = y0; gx = x0; }
    else { gy = 2 * y0 - 1; gx = 2 * x0 - 1; }
  };
  TileWalk walk(a.ntiles);
  TileStager<CI, PRO_BN, XR, XC, false, 256, ACT, ACT> sx;      // staging waves only (threadIdx.x 0..255)
  typename std::conditional<DUREC, DU1to8Stager<DC, ACT>, TileStager<CO, DYPRO, DR, DC, false, 256, float, ACT>>::type sd;
  auto sd_store = [&](float* dst) __attribute__((always_inline)) {
    if constexpr (DUREC) sd.store(dst, cd, xs); else sd.store(dst, cd);
  };
  auto prefetch = [&](int tl) {
    int b, y0, x0, gy, gx;
    origin(tl, b, y0, x0);
    if (AVA_FABL(8)) return;
    x_origin(y0, x0, gy, gx);
    sx.load(a.x, nullptr, b, a.Hi, a.Wi, gy, gx);
    d_origin(y0, x0, gy, gx);
    sd.load(a.dy, a.dy2, b, a.Ho, a.Wo, gy, gx);
  };
  if (stager) {
    sx.init();
    if constexpr (DUREC) sd.init(a.rcd, xs); else sd.init();
    if (walk.valid()) prefetch(walk.cur);    // tile 0 goes in flight BEFORE the coefficient prologue
  }
  __shared__ double accvals[64];            // consumer prologue scratch (bn_coef_from_acc)
  // Everything the prologue and the epilogue read from global memory is requested HERE, in front of the coefficient
  // finalisation and its barrier: x's scale / shift, the packed data-gradient weights (touched: the fragment loads behind the
  // barrier then hit) and the statistics the final reduction centres with.  Requested where they are used, each was one
  // more exposed memory latency of a kernel whose skeleton is 15-27 us.
  float cxv = 0.f;
  if (t < 96) {
    const int which = t >> 5, c = t & 31;
    const float* sx = which == 0 ? a.xa : (which == 1 ? a.xb : nullptr);
    cxv = (sx != nullptr && c < CI) ? sx[c] : 0.f;
  }
  // (one float per 128-byte line and thread: 256 threads cover 32 KB, the largest weight array is 27 KB)
  static_assert(9 * CI * CO * 4 <= 256 * 128, "one touch per thread covers the packed weights");
  float wpf = 0.f;
  __shared__ float ems[64];                   // mean [0..31], invstd [32..63] of x's BatchNorm for the final reduction
  if (!stager) {
    wpf = a.Gb[min(32 * (t - 256), 9 * CI * CO - 1)];
    const int e = t - 320;                    // the SECOND matrix-core wave: the first one finalises the coefficients
    if (e >= 0 && e < 64) {
      const int c = e & 31;
      ems[e] = c < CI ? (e < 32 ? a.mean[c] : a.invstd[c]) : 0.f;
    }
  }
  if (a.fin.acc != nullptr) {
    // A, Bc, Cc of the BatchNorm above finalised here from the accumulated sums of the kernel that ran before
    // (by the first matrix-core wave, under the staging waves' first tile load)
    bn_coef_from_acc(cd, accvals, a.fin, 256);
    if (t < 96) cx[t] = cxv;
  } else if (t < 96) {
    const int which = t >> 5, c = t & 31;
    const float* sd = which == 0 ? a.da : (which == 1 ? a.db : a.dc);
    cx[t] = cxv;
    cd[t] = (sd != nullptr && c < CO) ? sd[c] : 0.f;
  }
  __syncthreads();                           // cx / cd / zero pads visible
  if (!stager) asm volatile("" ::"v"(wpf));

  if (stager) {
    // ---------------- staging waves ----------------
    // issue priority over the matrix-core waves of the same SIMD: their loads and LDS stores are the longer pipeline
    // (DESIGN.md section 3, item 16); same-box A/B of the step: 1.9141 / 1.9131 ms without, 1.9087 / 1.8979 ms with (the
    // reverse, matrix-core waves first, costs +26 us)
    __builtin_amdgcn_s_setprio(3);
    if (walk.valid()) {
      if (!AVA_FABL(4)) { sx.store(smem, cx); sd_store(smem + XF); }
      if (walk.has_next()) prefetch(walk.next());
    }
    __syncthreads();                                            // (A) tile 0 ready
    int it = 0;
    for (; walk.valid(); walk.advance(), ++it) {
      if (walk.has_next()) {                                    // tile it+1 -> the other buffer, tile it+2 in flight
        float* nb = smem + ((it + 1) & 1) * BUF_F;
        if (!AVA_FABL(4)) { sx.store(nb, cx); sd_store(nb + XF); }
        const int nn = walk.next() + walk.step;
        if (nn < walk.end) prefetch(nn);
      }
      __syncthreads();                                          // (B)
    }
    for (int i = 0; i < 7; ++i) __syncthreads();                // the seven barriers of the reductions below
    return;
  }

  // ---------------- matrix-core waves ----------------
  // ---- backward-data fragments: the flipped/packed weights stay in registers for the whole kernel ----
  constexpr int SPB = BMODE == MODE_DOWN ? 2 : 1;
  typename std::conditional<PAIR, PairFrag<CO, DC>, ClassFrag<CO, CI, BMODE, 0, DC>>::type f0;
  ClassFrag<CO, CI, BMODE, (BCLS > 1 ? 1 : 0), DC> f1;
  ClassFrag<CO, CI, BMODE, (BCLS > 1 ? 2 : 0), DC> f2;
  ClassFrag<CO, CI, BMODE, (BCLS > 1 ? 3 : 0), DC> f3;
  constexpr bool BF16M = std::is_same<ACT, ava_bf16>::value;     // bf16 arithmetic: weights rounded to bfloat16
  f0.init(a.Gb, lane, SPB * n * CO, 0, BF16M);
  if (BCLS > 1) { f1.init(a.Gb, lane, n * CO, 0, BF16M); f2.init(a.Gb, lane, n * CO, 0, BF16M); f3.init(a.Gb, lane, n * CO, 0, BF16M); }
  const int lane_out = PAIR ? ((kg >> 1) * a.Wi + n) * CI + 4 * (kg & 1) : (BMODE == MODE_UP ? 2 * n : n) * CI + 4 * kg;
  const int cq = PAIR ? 4 * (kg & 1) : 4 * kg;    // first dx channel of this lane inside its channel tile
  float s1[MT][4], s2[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s1[mt][r] = s2[mt][r] = 0.f;
    }

  // ---- weight-gradient accumulators (persist over all tiles of this workgroup) ----
  WClass<CI, CO, LMODE, 0, XC> w0;
  WClass<CI, CO, LMODE, (WCLS > 1 ? 1 : 0), XC> w1;
  WClass<CI, CO, LMODE, (WCLS > 1 ? 2 : 0), XC> w2;
  WClass<CI, CO, LMODE, (WCLS > 1 ? 3 : 0), XC> w3;
  w0.init(lane);
  if (WCLS > 1) { w1.init(lane); w2.init(lane); w3.init(lane); }
  float bsum[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bsum[nt] = 0.f;

  // pixel groups of the dx region: 16 consecutive pixels of a row (UP pattern: of one parity class)
  constexpr int CB = FG::OW / (BMODE == MODE_UP ? 32 : 16);          // column blocks
  constexpr int GROUPS = PAIR ? (FG::OH / 2) * CB : (BMODE == MODE_UP ? 4 * (FG::OH / 2) * CB : FG::OH * CB);
  constexpr int GPW = GROUPS / 4;
  static_assert(GROUPS % 4 == 0 && (BMODE != MODE_UP || GPW % 4 == 0), "tile must split evenly over the 4 waves");
  // offset (floats, relative to the dx region's first pixel) and LDS pixel base of group g
  auto group_out = [&](int g) -> int {
    if (BMODE == MODE_UP) {
      const int cls = g & 3, rest = g >> 2, r = rest / CB, cb = rest % CB;
      return ((2 * r + (cls >> 1)) * a.Wi + 32 * cb + (cls & 1)) * CI;
    }
    return (((PAIR ? 2 : 1) * (g / CB)) * a.Wi + 16 * (g % CB)) * CI;
  };

  // raw x at this lane's dx pixels (BatchNorm-backward sums).  Loaded one tile ahead, AFTER the tile's data-gradient
  // phase has consumed the previous values: the lines were requested by the window prefetch a moment earlier, so
  // this hits L2, and the loads ride under the weight-gradient phase.
  avaf4 ex[GPW * MT];
  auto load_ex = [&](int tl) {
    int b, y0, x0;
    origin(tl, b, y0, x0);
    const int oy0 = LMODE == MODE_DOWN ? 2 * y0 : y0, ox0 = LMODE == MODE_DOWN ? 2 * x0 : x0;
    const ACT* __restrict__ xb = ava_as<ACT>(a.x) + (((size_t)b * a.Hi + oy0) * a.Wi + ox0) * CI;
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int cb4 = 16 * mt + cq;
        // lanes whose 4-channel slot lies beyond CI re-read slot 0: stays in bounds
        ex[gi * MT + mt] = ava_ld4<ACT>(
            xb + group_out(wave * GPW + gi) + (cb4 < CI ? lane_out + 16 * mt : lane_out - 4 * kg));
      }
  };

  if (walk.valid() && !AVA_FABL(32)) load_ex(walk.cur);
  __syncthreads();                                              // (A)
  int it = 0;
  for (; walk.valid(); walk.advance(), ++it) {
    int b, y0, x0;
    origin(walk.cur, b, y0, x0);
    const float* xt = smem + (it & 1) * BUF_F;
    const float* dut = xt + XF;
    const int oy0 = LMODE == MODE_DOWN ? 2 * y0 : y0, ox0 = LMODE == MODE_DOWN ? 2 * x0 : x0;
    const size_t tile_pix = ((size_t)b * a.Hi + oy0) * a.Wi + ox0;
    float* __restrict__ obase = a.dx + tile_pix * CI;

    // ---- phase 1: data gradient of the tile + BatchNorm-backward sums ----
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      const int g = wave * GPW + gi;
      f32x4 acc[2][MT];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[h][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (AVA_FABL(1)) {
      } else if (BMODE == MODE_UP) {
        const int cls = gi & 3, rest = g >> 2, r = rest / CB, cb = rest % CB;   // GPW % 4 == 0: cls is compile-time
        const float* px = dut + (r * DC + 16 * cb) * CO;
        if (cls == 0) f0.run(px, acc);
        else if (cls == 1) f1.run(px, acc);
        else if (cls == 2) f2.run(px, acc);
        else f3.run(px, acc);
      } else {
        constexpr int S = (BMODE == MODE_S1 && !PAIR) ? 1 : 2;        // PAIR: a group is a pair of dx rows
        constexpr int SX = BMODE == MODE_DOWN ? 2 : 1;
        f0.run(dut + (S * (g / CB) * DC + SX * 16 * (g % CB)) * CO, acc);
      }
      const int gout = group_out(g) + lane_out;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int cb4 = 16 * mt + cq;
        if (cb4 < CI) {
          const f32x4 v = acc[0][mt] + acc[1][mt];
          const avaf4 xr = ex[gi * MT + mt];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s1[mt][r] += v[r];
            s2[mt][r] = fmaf(v[r], xr[r], s2[mt][r]);          // raw x: centred after the loop
          }
          if (!AVA_FABL(16)) *reinterpret_cast<float4*>(obase + gout + 16 * mt) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }

    if (walk.has_next() && !AVA_FABL(32)) load_ex(walk.next());

    // ---- phase 2: weight / bias gradient over the tile's interior dU pixels ----
    if (AVA_FABL(2)) {
    } else if (LMODE == MODE_UP) {
      // x-space rows r = wave + 4 rr, columns c = 4 s + kg; the four output-parity classes of each x pixel
#pragma unroll 1
      for (int rr = 0; rr < TH / 4; ++rr) {
        const int r = wave + 4 * rr;
#pragma unroll 1
        for (int s = 0; s < TW / 4; ++s) {
          const int c = 4 * s + kg;
          const float* xa = xt + (r * XC + c) * CI;
#pragma unroll
          for (int cls = 0; cls < 4; ++cls) {
            const int py = cls >> 1, px = cls & 1;
            const float* bp = dut + ((2 * r + py + DOFF) * DC + 2 * c + px + DOFF) * CO + n;
            float bf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { bf[nt] = bp[16 * nt]; bsum[nt] += (16 * nt + n < CO) ? bf[nt] : 0.f; }
            if (cls == 0) w0.step(xa, bf);
            else if (cls == 1) w1.step(xa, bf);
            else if (cls == 2) w2.step(xa, bf);
            else w3.step(xa, bf);
          }
        }
      }
    } else {
      constexpr int S = LMODE == MODE_S1 ? 1 : 2;
      constexpr int RPW = TH / 4;                 // dU rows per wave
#pragma unroll 1
      for (int rr = 0; rr < RPW; ++rr) {
        const int ty = wave * RPW + rr;
#pragma unroll 1
        for (int s = 0; s < TW / 4; ++s) {
          const int x = 4 * s + kg;
          const float* bp = dut + ((ty + DOFF) * DC + x + DOFF) * CO + n;
          float bf[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) { bf[nt] = bp[16 * nt]; bsum[nt] += (16 * nt + n < CO) ? bf[nt] : 0.f; }
          w0.step(xt + ((S * ty) * XC + S * x) * CI, bf);
        }
      }
    }
    __syncthreads();                                            // (B)
  }

  // ---- BatchNorm-backward partial sums: over the 16 pixel lanes, then over the 4 waves (fixed order) ----
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v1 = s1[mt][r], v2 = s2[mt][r];
      {
        // the hot loop accumulates sum g*x on RAW x; centred and scaled once per lane here: sum g*xhat = invstd * (sum g*x - mean * sum g)
        const int cc = 16 * mt + cq + r;
        const float mu = ems[cc & 31], is = ems[32 + (cc & 31)];  // requested in the prologue (zero beyond CI)
        v2 = fmaf(-mu, v1, v2) * is;
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { v1 += __shfl_xor(v1, o, 64); v2 += __shfl_xor(v2, o, 64); }
      if (PAIR) { v1 += __shfl_xor(v1, 32, 64); v2 += __shfl_xor(v2, 32, 64); }   // the two rows of a pair
      if (n == 0 && (!PAIR || kg < 2)) {
        const int ci = 16 * mt + cq + r;
        red[wave * 32 * MT + ci] = v1;
        red[wave * 32 * MT + 16 * MT + ci] = v2;
      }
    }
  __syncthreads();
  const int tc = t - 256;
  if (tc < 2 * CI) {
    const int which = tc / CI, ci = tc - which * CI;
    const int idx = which * 16 * MT + ci;
    const float tot = (red[idx] + red[32 * MT + idx]) + (red[64 * MT + idx] + red[96 * MT + idx]);
    if (a.acc_out != nullptr) bn_acc_add(a.acc_out, which * 32 + ci, tot);
    else a.bn_partials[(size_t)blockIdx.x * 2 * CI + tc] = tot;
  }
  __syncthreads();

  // ---- weight-gradient partial row: the four waves summed through LDS in a fixed order ----
  float* wacc = smem;                             // [NW + CO], aliases the tiles (all reads are done)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    bsum[nt] += __shfl_xor(bsum[nt], 16, 64);
    bsum[nt] += __shfl_xor(bsum[nt], 32, 64);
  }
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      w0.flush(wacc, lane, w == 0);
      if (WCLS > 1) { w1.flush(wacc, lane, w == 0); w2.flush(wacc, lane, w == 0); w3.flush(wacc, lane, w == 0); }
      if (kg == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int co = 16 * nt + n;
          if (co < CO) wacc[NW + co] = (w == 0) ? bsum[nt] : wacc[NW + co] + bsum[nt];
        }
      }
    }
    __syncthreads();
  }
  float* prow = a.wg_partials + (size_t)blockIdx.x * (NW + CO);
  for (int e = tc; e < NW + CO; e += 256) prow[e] = wacc[e];
}

// ------------------------------------------------------------------------------------------------
static void fused_defaults(int Cin, int Cout, int mode, bool* ws, int* cap);

template <int CI, int CO, int LMODE, int DYPRO, int TW, int TH, int MINW, typename ACT, bool DUREC = false>
static int launch_fused_t(const FusedArgs& a, int grid, hipStream_t st) {
  using FG = FGeom<LMODE, TW, TH>;
  constexpr int MT = (CI + 15) / 16;
  // wave-specialised variant (staging waves beside matrix-core waves, two tile buffers) where it wins: fused_defaults
  bool ws; int cap_unused;
  fused_defaults(CI, CO, LMODE, &ws, &cap_unused);
#ifndef AVA_LAB
  if (!ws) return AVA_EINVAL;           // every shape of the library runs the wave-specialised kernel
#endif
  const size_t buf_f = (size_t)FG::XR * FG::XC * CI + FG::DR * FG::DC * CO + 16;
  const size_t tiles_f = (ws ? 2 : 1) * buf_f + 192 + 4 * 32 * MT + (DUREC ? DU1to8Stager<FG::DC, ACT>::LDS_FLOATS : 0);
  const size_t red_f = (size_t)9 * CI * CO + CO;
  const size_t lds = (tiles_f > red_f ? tiles_f : red_f) * sizeof(float);
#ifdef AVA_LAB
  const void* kfn = ws ? reinterpret_cast<const void*>(&conv3x3_bwd_fused_ws_kernel<CI, CO, LMODE, DYPRO, TW, TH, MINW, ACT, DUREC>)
                       : reinterpret_cast<const void*>(&conv3x3_bwd_fused_kernel<CI, CO, LMODE, DYPRO, TW, TH, MINW, ACT>);
#else
  const void* kfn = reinterpret_cast<const void*>(&conv3x3_bwd_fused_ws_kernel<CI, CO, LMODE, DYPRO, TW, TH, MINW, ACT, DUREC>);
#endif
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return AVA_ELAUNCH;
    attr_set = true;
  }
  FusedArgs b = a;
  const int hl = LMODE == MODE_DOWN ? a.Ho : a.Hi, wl = LMODE == MODE_DOWN ? a.Wo : a.Wi;   // low-resolution side
  if (hl % TH != 0 || wl % TW != 0) return AVA_EINVAL;
  b.tiles_y = hl / TH;
  b.tiles_x = wl / TW;
  b.ntiles = a.B * b.tiles_y * b.tiles_x;
  if (grid < 1 || grid > b.ntiles) return AVA_EINVAL;
#ifdef AVA_LAB
  if (!ws) hipLaunchKernelGGL((conv3x3_bwd_fused_kernel<CI, CO, LMODE, DYPRO, TW, TH, MINW, ACT>), dim3(grid), dim3(256), lds, st, b);
  else
#endif
  hipLaunchKernelGGL((conv3x3_bwd_fused_ws_kernel<CI, CO, LMODE, DYPRO, TW, TH, MINW, ACT, DUREC>), dim3(grid), dim3(512), lds, st, b);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

template <int CI, int CO, int LMODE, int DYPRO, int TW, int TH, int MINW>
static int launch_fused(const FusedArgs& a, int grid, hipStream_t st) {
  if constexpr (CI == 8 && CO == 8 && LMODE == MODE_UP && DYPRO == PRO_BWD && TH == 4) {
    if (a.rcd.G1 != nullptr) {            // convt6's backward with convt7's data gradient formed in the staging waves
      if (a.act_bf16) return launch_fused_t<CI, CO, LMODE, DYPRO, TW, TH, MINW, ava_bf16, true>(a, grid, st);
      return launch_fused_t<CI, CO, LMODE, DYPRO, TW, TH, MINW, float, true>(a, grid, st);
    }
  }
  if (a.rcd.G1 != nullptr) return AVA_EINVAL;
  if (a.act_bf16) return launch_fused_t<CI, CO, LMODE, DYPRO, TW, TH, MINW, ava_bf16>(a, grid, st);
  return launch_fused_t<CI, CO, LMODE, DYPRO, TW, TH, MINW, float>(a, grid, st);
}

// shapes with a fused instantiation: (cin, cout, mode, variant) -> low-resolution tile, occupancy hint.
// Variant 0 is what the library runs.  The lab build (make lab) also compiles the tile shapes that lost the
// tools/fused_bench.py comparison (AVA_FUSED_VAR=n), several of which spill.
#define AVA_FUSED_SHAPES_DEFAULT(X) \
  X(8, 8, MODE_DOWN, 0, 16, 4, 2)   \
  X(8, 16, MODE_S1, 0, 32, 4, 2)    \
  X(16, 16, MODE_DOWN, 0, 16, 4, 2) \
  X(16, 16, MODE_UP, 0, 16, 4, 2)   \
  X(16, 8, MODE_S1, 0, 32, 4, 2)    \
  X(8, 8, MODE_UP, 0, 32, 4, 2)     \
  X(16, 24, MODE_S1, 0, 32, 4, 1)   \
  X(24, 16, MODE_S1, 0, 32, 4, 1)
#ifdef AVA_LAB
#define AVA_FUSED_SHAPES(X)         \
  AVA_FUSED_SHAPES_DEFAULT(X)       \
  X(8, 8, MODE_DOWN, 1, 32, 4, 2)   \
  X(8, 8, MODE_DOWN, 2, 16, 4, 3)   \
  X(8, 16, MODE_S1, 1, 16, 8, 2)    \
  X(8, 16, MODE_S1, 2, 32, 8, 2)    \
  X(16, 16, MODE_DOWN, 1, 32, 4, 2) \
  X(16, 16, MODE_DOWN, 2, 16, 8, 2) \
  X(16, 16, MODE_UP, 1, 32, 4, 2)   \
  X(16, 16, MODE_UP, 2, 16, 4, 3)   \
  X(16, 8, MODE_S1, 1, 16, 8, 2)    \
  X(16, 8, MODE_S1, 2, 32, 8, 2)    \
  X(16, 24, MODE_S1, 1, 16, 4, 1)   \
  X(24, 16, MODE_S1, 1, 16, 4, 1)   \
  X(8, 8, MODE_UP, 1, 16, 4, 3)     \
  X(8, 8, MODE_UP, 2, 32, 4, 3)
#else
#define AVA_FUSED_SHAPES(X) AVA_FUSED_SHAPES_DEFAULT(X)
#endif

// Which shapes run the wave-specialised kernel, and with how many workgroups (= partial rows).  Measured at batch 256
// (tools/fused_bench.py): every shape gains 4-25 % once its 512-thread workgroup fits 128 VGPRs, so that two are resident per
// CU (launch bounds (512, 4); conv4's 16->16 DOWN needs 200 and runs one per CU).  AVA_FUSED_WS=0 / 1 forces all off / on.
static void fused_defaults(int Cin, int Cout, int mode, bool* ws, int* cap) {
  *ws = false; *cap = 512;
  if (Cin == 8 && Cout == 8 && mode == MODE_DOWN) { *ws = true; *cap = 512; }    // 16x4 tiles: 126 VGPRs, 2 workgroups / CU
  if (Cin == 8 && Cout == 8 && mode == MODE_UP) { *ws = true; *cap = 512; }
  if (Cin == 16 && Cout == 16 && mode == MODE_DOWN) { *ws = true; *cap = 256; }
  if (Cin == 16 && Cout == 16 && mode == MODE_UP) { *ws = true; *cap = 512; }    // 16x4 tiles: 128 VGPRs
  if (mode == MODE_S1) { *ws = true; *cap = 512; }                                // 32x4 tiles: 126 / 128 VGPRs (12 B spill for 16->8)
  if (mode == MODE_S1 && Cin * Cout > 256) { *ws = true; *cap = 256; }            // conv5 / convt3: one workgroup per CU
  static const int force = [] { const char* e = ava_env("AVA_FUSED_WS"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
  if (force == 0) { *ws = false; *cap = 512; }
  if (force == 1) *ws = true;
  static const int gcap = [] { const char* e = ava_env("AVA_FUSED_GRID"); return (e && atoi(e) >= 8) ? atoi(e) : 0; }();
  if (gcap > 0) *cap = gcap;
}


static int fused_variant() {
  static const int v = [] { const char* e = ava_env("AVA_FUSED_VAR"); return e ? atoi(e) : 0; }();
  return v;
}

static bool fused_tile(int Cin, int Cout, int mode, int* tw, int* th) {
  const int var = fused_variant();
#define X(ci, co, md, vr, tww, thh, mw) if (Cin == ci && Cout == co && mode == md && var == vr) { *tw = tww; *th = thh; return true; }
  AVA_FUSED_SHAPES(X)
#undef X
  return false;
}
#endif   // AVA_LAB

// both products on bf16 limb MFMA where the shape has that kernel (conv_fused_limb.hip; lab: AVA_FUSED_LIMB=0 keeps fp32 MFMA)
static bool fused_limb_on() {
  static const bool limb = [] { const char* e = ava_env("AVA_FUSED_LIMB"); return e == nullptr || atoi(e) != 0; }();
  return limb;
}

// number of workgroups (= partial rows of both outputs) of the fused kernel, 0 when the shape has none
int ava_conv_fused_grid_for(int B, int Hi, int Wi, int Cin, int Cout, int mode) {
  const int thin = ava_thin_fused_grid(B, Hi, Wi, Cin, Cout, mode);
  if (thin > 0) return thin;
  int tw, th;
  const int hl = mode == MODE_DOWN ? Hi / 2 : Hi, wl = mode == MODE_DOWN ? Wi / 2 : Wi;
  if (fused_limb_on()) {                        // the limb kernel's own tile and resident-wave size (conv_fused_limb.hip)
    const int lcap = ava_conv_fused_limb_cap(Cin, Cout, mode, &tw, &th);
    if (lcap > 0 && hl % th == 0 && wl % tw == 0) {
      const int nt = B * (hl / th) * (wl / tw);
      return nt < ava_scale_grid(lcap) ? nt : ava_scale_grid(lcap);
    }
  }
#ifdef AVA_LAB
  if (!fused_tile(Cin, Cout, mode, &tw, &th)) return 0;
  if (hl % th != 0 || wl % tw != 0) return 0;
  const int nt = B * (hl / th) * (wl / tw);
  bool ws; int cap;
  fused_defaults(Cin, Cout, mode, &ws, &cap);   // 512 = two resident 256-thread workgroups per CU (384 / 768 / 1024 are slower)
  return nt < ava_scale_grid(cap) ? nt : ava_scale_grid(cap);
#else
  return 0;
#endif
}

int ava_conv3x3_bwd_fused_launch(const FusedArgs& a_, int Cin, int Cout, int mode, int dy_pro, hipStream_t st) {
  FusedArgs a = a_;
  { static const int dbg = [] { const char* e = ava_env("AVA_FDBG"); return e ? atoi(e) : 0; }(); a.dbg = dbg; }
  const int grid = ava_conv_fused_grid_for(a.B, a.Hi, a.Wi, Cin, Cout, mode);
  if (grid <= 0) return AVA_EINVAL;
  if (Cin == 1 || Cout == 1) return ava_thin_bwd_fused_launch(a, grid, Cin, dy_pro, st);
  if (a.dx == nullptr) return AVA_EINVAL;
  if (fused_limb_on() && ava_conv_fused_limb_has(Cin, Cout, mode)) {
    const int rc = ava_conv3x3_bwd_fused_limb_launch(a, grid, Cin, Cout, mode, dy_pro, st);
    if (rc != AVA_EINVAL) return rc;            // AVA_EINVAL: the image does not divide into the limb kernel's tiles
  }
#ifdef AVA_LAB
  const int var = fused_variant();
#define X(ci, co, md, vr, tww, thh, mw)                                                            \
  if (Cin == ci && Cout == co && mode == md && var == vr) {                                         \
    if (dy_pro == PRO_BWD) return launch_fused<ci, co, md, PRO_BWD, tww, thh, mw>(a, grid, st);     \
    if (dy_pro == PRO_ID) return launch_fused<ci, co, md, PRO_ID, tww, thh, mw>(a, grid, st);       \
    return AVA_EINVAL;                                                                              \
  }
  AVA_FUSED_SHAPES(X)
#undef X
#endif
  return AVA_EINVAL;
}

extern "C" int ava_conv_fused_grid(int B, int Hi, int Wi, int Cin, int Cout, int mode) {
  return ava_conv_fused_grid_for(B, Hi, Wi, Cin, Cout, mode);
}

extern "C" int ava_conv3x3_bwd_fused(const float* x, const float* xa, const float* xb, const float* dy, const float* dy2,
                                     const float* da, const float* db, const float* dc, const float* Gb, float* dx,
                                     const float* mean, const float* invstd, float* bn_partials, float* wg_partials,
                                     int B, int Hi, int Wi, int Cin, int Cout, int mode, int dy_pro, ava_stream_t s) {
  if (x == nullptr || xa == nullptr || xb == nullptr || dy == nullptr || Gb == nullptr || (dx == nullptr && Cin != 1) ||
      mean == nullptr || invstd == nullptr || bn_partials == nullptr || wg_partials == nullptr || B < 1)
    return AVA_EINVAL;
  if (dy_pro == PRO_BWD && (dy2 == nullptr || da == nullptr || db == nullptr || dc == nullptr)) return AVA_EINVAL;
  FusedArgs a = {};
  a.x = x; a.xa = xa; a.xb = xb; a.dy = dy; a.dy2 = dy2; a.da = da; a.db = db; a.dc = dc; a.Gb = Gb; a.dx = dx;
  a.mean = mean; a.invstd = invstd; a.bn_partials = bn_partials; a.wg_partials = wg_partials;
  a.act_bf16 = 0;
  a.acc_out = nullptr;
  a.fin = BnFin{};
  a.B = B; a.Hi = Hi; a.Wi = Wi;
  a.Ho = mode == MODE_S1 ? Hi : (mode == MODE_DOWN ? Hi / 2 : Hi * 2);
  a.Wo = mode == MODE_S1 ? Wi : (mode == MODE_DOWN ? Wi / 2 : Wi * 2);
  a.tiles_y = a.tiles_x = a.ntiles = 0;
  return ava_conv3x3_bwd_fused_launch(a, Cin, Cout, mode, dy_pro, to_stream(s));
}
