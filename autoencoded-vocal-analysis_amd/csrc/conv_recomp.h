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
