// y1 = relu(conv1(bn1(x))) (ava/models/vae.py:217) recomputed where it is consumed instead of being stored.
//
// y1 and its gradient are the two largest tensors of the step (8 channels at full resolution: 128 MiB each at batch
// 256) while conv1 itself is 1.18 MMAC per sample on a 16 MiB input.  The forward therefore takes y1's BatchNorm
// statistics in a store-free pass (thin_1to8_kernel with out == nullptr) and conv2's kernels -- forward and fused
// backward -- build the y1 window they need from the x window, in the staging waves, with EXACTLY the arithmetic of
// thin_1to8_kernel (same fma order: kx outer, ky inner, accumulator from 0, then + bias, ReLU, storage rounding), so the
// values are bit-identical to those whose statistics were taken and the ReLU masks of forward and backward agree.
//
// Y1Stager has TileStager's interface (init / load / store) for an [R x C] window of the 8-channel tensor.  Work is
// split by ROWS over the four staging waves so that a wave only needs x rows it has loaded itself: every wave keeps a
// private [(rows + 2) x (C + 2)] window of bn1(x) (zero outside the image: conv1's padding) in LDS and no barrier is
// needed between writing it and reading it back (LDS operations of one wave execute in order).
#pragma once
#include "conv_common.h"
#include "conv_mfma.h"

// RAWI: also keep the raw (un-normalised) y1 of the window's interior [RI x CI at (ROFF, COFF)] in a second LDS tile
// (the fused backward's BatchNorm sums need raw x at the dx pixels)
template <int R, int C, typename ACT, bool RAWI = false, int RI = 0, int CI = 0, int ROFF = 0, int COFF = 0>
struct Y1Stager {
  static constexpr int XC = C + 2;
  static constexpr int BASE = R / 4, REM = R % 4, MAXR = BASE + (REM > 0 ? 1 : 0);
  static constexpr int XS_F = (MAXR + 2) * XC;                  // floats of one wave's private x window
  static constexpr int NLX = (XS_F + 63) / 64;                  // x loads per lane and tile
  static constexpr int NU = (MAXR * C * 2 + 63) / 64;           // (pixel, channel quad) units per lane and tile
  static_assert(NLX <= 16 && NU <= 16, "masks are 16 bits wide");
  static constexpr int LDS_FLOATS = 4 * XS_F;                   // all four waves' windows

  float xr[NLX];
  float w[9][4], bias[4];
  float ca1, cb1;
  int r0, nr;             // this wave's rows [r0, r0 + nr) of the y1 window
  unsigned xin;           // bit i: x element i of the current registers is inside the image
  unsigned yin;           // bit i: unit i's pixel is inside the image (conv2 pads bn2(y1) with zeros outside)
  int q4;                 // 4 * channel quad of this lane (lane & 1)
  int lane, wave;

  __device__ __forceinline__ void init(const RecompArgs& rc, int tid = (int)threadIdx.x) {
    lane = tid & 63; wave = tid >> 6;
    nr = BASE + (wave < REM ? 1 : 0);
    r0 = wave * BASE + (wave < REM ? wave : REM);
    q4 = 4 * (lane & 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = rc.G1[tap * 8 + q4 + j];
        asm volatile("" : "+v"(v));            // read once, before the tile loop (see ClassFrag::init)
        w[tap][j] = v;
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) { float v = rc.bias1[q4 + j]; asm volatile("" : "+v"(v)); bias[j] = v; }
    ca1 = rc.pa1[0]; cb1 = rc.pb1[0];
    xin = yin = 0u;
  }

  // (gy0, gx0): image coordinates of the y1 window's first pixel
  __device__ __forceinline__ void load(const float* __restrict__ x, const float* /*in2*/, int b, int H, int W, int gy0, int gx0) {
    const float* __restrict__ base = x + (size_t)b * H * W;
    xin = 0u;
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int e = lane + 64 * i, er = e / XC, ec = e - er * XC;
      const int gy = gy0 - 1 + r0 + er, gx = gx0 - 1 + ec;
      const bool ok = er < nr + 2 && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
      xr[i] = base[cy * W + cx];
      xin |= ok ? (1u << i) : 0u;
    }
    yin = 0u;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int pu = (lane + 64 * i) >> 1, rr = pu / C, cc = pu - rr * C;
      const int gy = gy0 + r0 + rr, gx = gx0 + cc;
      yin |= (gy >= 0 && gy < H && gx >= 0 && gx < W) ? (1u << i) : 0u;
    }
  }

  // lds: the [R x C x 8] tile; coef: bn2 scale [0..7] / shift [32..39]; xs_all: LDS_FLOATS of scratch; raw: RAWI tile
  __device__ __forceinline__ void store(float* __restrict__ lds, const float* __restrict__ coef, float* __restrict__ xs_all,
                                        float* __restrict__ raw = nullptr) {
    float* __restrict__ xs = xs_all + wave * XS_F;
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int e = lane + 64 * i;
      const float v = ((xin >> i) & 1u) ? fmaf(xr[i], ca1, cb1) : 0.f;       // prologue<PRO_BN>, zero outside the image
      if (e < XS_F) xs[e] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float a2[4], b2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a2[j] = coef[q4 + j]; b2[j] = coef[32 + q4 + j]; }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int pu = (lane + 64 * i) >> 1, rr = pu / C, cc = pu - rr * C;
      if (rr < nr) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const float xv = xs[(rr + ky) * XC + cc + kx];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(xv, w[ky * 3 + kx][j], acc[j]);
          }
        avaf4 y, o;
        const bool in = (yin >> i) & 1u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          y[j] = ava_stored<ACT>(fmaxf(acc[j] + bias[j], 0.f));
          o[j] = in ? fmaf(y[j], a2[j], b2[j]) : 0.f;
        }
        *reinterpret_cast<avaf4*>(lds + ((r0 + rr) * C + cc) * 8 + q4) = o;
        if constexpr (RAWI) {
          const int ri = r0 + rr - ROFF, ci = cc - COFF;
          if (ri >= 0 && ri < RI && ci >= 0 && ci < CI) *reinterpret_cast<avaf4*>(raw + (ri * CI + ci) * 8 + q4) = y;
        }
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// The same idea on the MATRIX cores, for a 1 -> 8-channel 3x3 gather whose 8-channel result is only an intermediate:
// convt7's data gradient dd6 = gather(seed r, W7) (vae.py:269 backward), consumed by convt6's fused backward as the `dy`
// operand of its ReLU / BatchNorm-backward prologue.  The VALU form above costs ~16 vector instructions per output
// value; here two output rows of 16 pixels x 8 channels are ONE 16 x 16 tile of v_mfma_f32_16x16x4_f32 (M = 2 rows x 8
// channels, N = 16 pixels, K = 3 taps kx x 4 input rows: 3 MFMAs, the PairFrag arrangement of conv_mfma.h), and a lane ends
// up with four channels of one pixel -- exactly the quad it writes to the LDS tile.  ~30 vector instructions per 256 values.
// The k order is (kx outer, input row inner), i.e. the fma chain of thin_1to8_kernel with a zero term per kx.
//
// Window [9 x C] (the stride-2 kernels' 2 TH + 1 rows at TH = 4).  Wave w of the four staging waves produces row pair
// (2w, 2w + 1) for every 16-column group, and the groups g = w, w + 4, .. of the last row (8).  It reads only input rows
// it loaded itself: a private LDS window of 8 rows x (16 NG + 2) columns -- rows 0-3: input rows of its pair, rows 4-6:
// input rows of row 8, row 7 and the pad columns: zeros (written once), so that a zero weight never meets a non-finite
// pad value.
template <int C>
struct Conv1to8Core {
  static constexpr int NG = (C + 15) / 16;
  static constexpr int XCP = 16 * NG + 2;
  static constexpr int XS_F = 8 * XCP;                          // floats of one wave's private window
  static constexpr int XE = 7 * (C + 2);                        // elements loaded per tile and wave
  static constexpr int NLX = (XE + 63) / 64;
  static constexpr int NUW = NG + (NG + 3) / 4;                 // units per wave (upper bound)
  static constexpr int LDS_FLOATS = 4 * XS_F;
  static_assert(NLX <= 16 && NUW <= 16, "masks are 16 bits wide");
  float xr[NLX];
  float wA[3];
  unsigned xin;
  int lane, wave;

  __device__ __forceinline__ void init(const float* __restrict__ G1, float* __restrict__ xs_all, int tid) {
    lane = tid & 63; wave = tid >> 6;
    const int m = lane & 15, kgA = lane >> 4, half = m >> 3, co = m & 7, ky = kgA - half;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = (ky >= 0 && ky <= 2) ? G1[(ky * 3 + c) * 8 + co] : 0.f;
      asm volatile("" : "+v"(v));
      wA[c] = v;
    }
    float* xs = xs_all + wave * XS_F;
    for (int e = lane; e < XS_F; e += 64) xs[e] = 0.f;          // pad columns and row 7 stay zero for the whole kernel
    xin = 0u;
  }
  // window row (0..10, relative to image row gy0 - 1) of private row pr
  __device__ __forceinline__ int win_row(int pr) const { return pr < 4 ? 2 * wave + pr : 4 + pr; }

  __device__ __forceinline__ void load(const float* __restrict__ src, int b, int H, int W, int gy0, int gx0) {
    const float* __restrict__ base = src + (size_t)b * H * W;
    xin = 0u;
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int e = lane + 64 * i, pr = e / (C + 2), pc = e - pr * (C + 2);
      const int gy = gy0 - 1 + win_row(pr), gx = gx0 - 1 + pc;
      const bool ok = e < XE && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
      xr[i] = base[cy * W + cx];
      xin |= ok ? (1u << i) : 0u;
    }
  }
  // private window <- the loaded values through v -> pro(v) (zero outside the image)
  template <class PRO>
  __device__ __forceinline__ void stage(float* __restrict__ xs_all, PRO pro) {
    float* __restrict__ xs = xs_all + wave * XS_F;
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int e = lane + 64 * i, pr = e / (C + 2), pc = e - pr * (C + 2);
      if (e < XE) xs[pr * XCP + pc] = ((xin >> i) & 1u) ? pro(xr[i]) : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // unit u of this wave -> (valid, first window row of the tile's M rows 0-7, column group); lane's output pixel / quad
  __device__ __forceinline__ bool unit(int u, int& row, int& col, int& quad, int& pb, int& g) const {
    const int n = lane & 15, kg = lane >> 4;
    const bool own = u < NG;
    g = own ? u : wave + 4 * (u - NG);
    pb = own ? 0 : 4;
    row = (own ? 2 * wave : 8) + (kg >> 1);
    col = 16 * g + n;
    quad = kg & 1;
    return g < NG && col < C && (own || (kg >> 1) == 0);
  }
  __device__ __forceinline__ f32x4 mma(const float* __restrict__ xs_all, int pb, int g) const {
    const float* __restrict__ xs = xs_all + wave * XS_F;
    const int n = lane & 15, kg = lane >> 4;
    const float* p = xs + (pb + kg) * XCP + 16 * g + n;
    const float b0 = p[0], b1 = p[1], b2 = p[2];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[0], b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[1], b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[2], b2, acc, 0, 0, 0);
    return acc;
  }
};

// dU tile [9 x C x 8] of convt6's fused backward, with dy = convt7's data gradient formed on the fly from the 1-channel
// seed:  dU = y > 0 ? A * dy + Bc * y + Cc : 0  (prologue<PRO_BWD>), y = the saved activation d6 (a.dy2), zero outside
// the image.  Interface of TileStager (init / load / store).
template <int C, typename ACT>
struct DU1to8Stager {
  using Core = Conv1to8Core<C>;
  static constexpr int LDS_FLOATS = Core::LDS_FLOATS;
  Core core;
  avaf4 y2[Core::NUW];
  unsigned yin;                      // bit u: unit u's pixel exists in the window and lies inside the image
  __device__ __forceinline__ void init(const RecompArgs& rc, float* xs_all, int tid = (int)threadIdx.x) {
    core.init(rc.G1, xs_all, tid);
    yin = 0u;
  }
  // in: the 1-channel seed [B,H,W]; in2: the saved 8-channel activation [B,H,W,8]; (gy0, gx0): window origin
  __device__ __forceinline__ void load(const float* __restrict__ in, const float* __restrict__ in2, int b, int H, int W,
                                       int gy0, int gx0) {
    core.load(in, b, H, W, gy0, gx0);
    const ACT* __restrict__ base2 = ava_as<ACT>(in2) + (size_t)b * H * W * 8;
    yin = 0u;
#pragma unroll
    for (int u = 0; u < Core::NUW; ++u) {
      int row, col, quad, pb, g;
      const bool v = core.unit(u, row, col, quad, pb, g);
      const int gy = gy0 + row, gx = gx0 + col;
      const bool ok = v && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
      y2[u] = ava_ld4<ACT>(base2 + (cy * W + cx) * 8 + 4 * quad);
      yin |= ok ? (1u << u) : 0u;
    }
  }
  // lds: the [9 x C x 8] fp32 tile; coef: A [0..7], Bc [32..39], Cc [64..71]
  __device__ __forceinline__ void store(float* __restrict__ lds, const float* __restrict__ coef, float* __restrict__ xs_all) {
    core.stage(xs_all, [](float v) { return v; });               // PRO_ID on the seed
    // the lane's channel quad is the same for every unit: its coefficients are read once, unconditionally (inside the
    // `in && y > 0 ? .. : 0` select the compiler may not speculate the LDS reads and serialises them per value)
    const float* ca = coef + 4 * ((core.lane >> 4) & 1);
    const avaf4 kA = {ca[0], ca[1], ca[2], ca[3]}, kB = {ca[32], ca[33], ca[34], ca[35]}, kC = {ca[64], ca[65], ca[66], ca[67]};
#pragma unroll
    for (int u = 0; u < Core::NUW; ++u) {
      int row, col, quad, pb, g;
      const bool v = core.unit(u, row, col, quad, pb, g);
      if (u >= Core::NG && core.wave + 4 * (u - Core::NG) >= Core::NG) continue;      // wave-uniform: no such group
      const f32x4 acc = core.mma(xs_all, pb, g);
      const bool in = (yin >> u) & 1u;
      avaf4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float y = y2[u][r];
        const float du = fmaf(kA[r], acc[r], fmaf(kB[r], y, kC[r]));
        o[r] = (in && y > 0.f) ? du : 0.f;
      }
      if (v) *reinterpret_cast<avaf4*>(lds + (row * C + col) * 8 + 4 * quad) = o;
    }
  }

  // the same tile as three bf16 limb planes ([limb][9 * C pixels][8 channels]: TileStagerL's layout for 8 channels) for the
  // limb form of the fused backward (conv_fused_limb.hip); same values, split exactly (x = x0 + x1 + x2)
  __device__ __forceinline__ void store_limb(unsigned char* __restrict__ lds, const float* __restrict__ coef, float* __restrict__ xs_all) {
    constexpr int PLANE_BYTES = ava_plane_pix(9 * C) * 16;      // one octet: the limb-plane stride of TileStagerL<8, .., 9, C>
    core.stage(xs_all, [](float v) { return v; });               // PRO_ID on the seed
    const float* ca = coef + 4 * ((core.lane >> 4) & 1);
    const avaf4 kA = {ca[0], ca[1], ca[2], ca[3]}, kB = {ca[32], ca[33], ca[34], ca[35]}, kC = {ca[64], ca[65], ca[66], ca[67]};
#pragma unroll
    for (int u = 0; u < Core::NUW; ++u) {
      int row, col, quad, pb, g;
      const bool v = core.unit(u, row, col, quad, pb, g);
      if (u >= Core::NG && core.wave + 4 * (u - Core::NG) >= Core::NG) continue;      // wave-uniform: no such group
      const f32x4 acc = core.mma(xs_all, pb, g);
      const bool in = (yin >> u) & 1u;
      avaf4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float y = y2[u][r];
        const float du = fmaf(kA[r], acc[r], fmaf(kB[r], y, kC[r]));
        o[r] = (in && y > 0.f) ? du : 0.f;
      }
      ava_u32x2 p0, p1, p2;
      uint32_t a, b, c;
      ava_limb_split2(o[0], o[1], a, b, c); p0[0] = a; p1[0] = b; p2[0] = c;
      ava_limb_split2(o[2], o[3], a, b, c); p0[1] = a; p1[1] = b; p2[1] = c;
      unsigned char* d = lds + (row * C + col) * 16 + quad * 8;
      if (v) {
        *reinterpret_cast<ava_u32x2*>(d) = p0;
        *reinterpret_cast<ava_u32x2*>(d + PLANE_BYTES) = p1;
        *reinterpret_cast<ava_u32x2*>(d + 2 * PLANE_BYTES) = p2;
      }
    }
  }
};

// y1 = relu(conv1(bn1 x)) windows on the matrix cores (Conv1to8Core): the form of Y1Stager that costs a third of the vector
// work.  [9 x C x 8] tile of bn2(y1) (zero outside the image) for conv2's forward / backward; RAWI: also the raw y1 of the
// interior [RI x CI at (ROFF, COFF)] (the fused backward's BatchNorm sums need raw x at its dx pixels).
template <int C, typename ACT, bool RAWI = false, int RI = 0, int CI = 0, int ROFF = 0, int COFF = 0>
struct Y1MfmaStager {
  using Core = Conv1to8Core<C>;
  static constexpr int LDS_FLOATS = Core::LDS_FLOATS;
  Core core;
  float ca1, cb1;
  float bias[4];                     // conv1's bias for this lane's channel quad (lane >> 4 & 1)
  int gy0s, gx0s, Hs, Ws;            // origin / image size of the tile whose x window is in the registers
  __device__ __forceinline__ void init(const RecompArgs& rc, float* xs_all, int tid = (int)threadIdx.x) {
    core.init(rc.G1, xs_all, tid);
    const int quad = ((tid & 63) >> 4) & 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) { float v = rc.bias1[4 * quad + j]; asm volatile("" : "+v"(v)); bias[j] = v; }
    ca1 = rc.pa1[0]; cb1 = rc.pb1[0];
    gy0s = gx0s = 0; Hs = Ws = 1;
  }
  __device__ __forceinline__ void load(const float* __restrict__ x, const float* /*in2*/, int b, int H, int W, int gy0, int gx0) {
    core.load(x, b, H, W, gy0, gx0);
    gy0s = gy0; gx0s = gx0; Hs = H; Ws = W;
  }
  __device__ __forceinline__ void store(float* __restrict__ lds, const float* __restrict__ coef, float* __restrict__ xs_all,
                                        float* __restrict__ raw = nullptr) {
    const float a1 = ca1, b1 = cb1;
    core.stage(xs_all, [a1, b1](float v) { return fmaf(v, a1, b1); });        // prologue<PRO_BN> of bn1
#pragma unroll
    for (int u = 0; u < Core::NUW; ++u) {
      int row, col, quad, pb, g;
      const bool v = core.unit(u, row, col, quad, pb, g);
      if (u >= Core::NG && core.wave + 4 * (u - Core::NG) >= Core::NG) continue;      // wave-uniform: no such group
      const f32x4 acc = core.mma(xs_all, pb, g);
      const float* ca = coef + 4 * quad;
      const int gy = gy0s + row, gx = gx0s + col;
      const bool in = gy >= 0 && gy < Hs && gx >= 0 && gx < Ws;
      avaf4 y, o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        y[r] = ava_stored<ACT>(fmaxf(acc[r] + bias[r], 0.f));
        o[r] = in ? fmaf(y[r], ca[r], ca[32 + r]) : 0.f;
      }
      if (v) {
        *reinterpret_cast<avaf4*>(lds + (row * C + col) * 8 + 4 * quad) = o;
        if constexpr (RAWI) {
          const int ri = row - ROFF, ci = col - COFF;
          if (ri >= 0 && ri < RI && ci >= 0 && ci < CI) *reinterpret_cast<avaf4*>(raw + (ri * CI + ci) * 8 + 4 * quad) = y;
        }
      }
    }
  }
};
