// Shared pieces of the VALU (conv.hip) and matrix-core (conv_mfma.hip) 3x3 gather convolutions.
#pragma once
#include "common.h"

enum { MODE_S1 = 0, MODE_DOWN = 1, MODE_UP = 2 };
enum { PRO_BN = 0, PRO_BWD = 1, PRO_ID = 2 };
enum { EPI_FWD = 0, EPI_BWD = 1, EPI_SSE = 2 };

struct ConvArgs {
  const float* in;
  const float* in2;
  const float* pa;
  const float* pb;
  const float* pc;
  const float* G;
  const float* bias;
  float* out;
  float* out2;
  const float* epi_x;
  const float* epi_mean;
  const float* epi_invstd;
  float* partials;
  int B, Hi, Wi, Ho, Wo;
  int relu;
  float prec;
  int tiles_y, tiles_x, ntiles;
};

template <int MODE, int TW, int TH_ = 256 / TW>
struct Geom {
  static constexpr int TH = TH_;
  static constexpr int IR = MODE == MODE_S1 ? TH + 2 : (MODE == MODE_DOWN ? 2 * TH + 1 : TH / 2 + 1);
  static constexpr int IC = MODE == MODE_S1 ? TW + 2 : (MODE == MODE_DOWN ? 2 * TW + 1 : TW / 2 + 1);
};

// prologue on one value of channel c
template <int PRO>
__device__ __forceinline__ float prologue(float v, float v2, float a, float b, float c) {
  if (PRO == PRO_BN) return fmaf(v, a, b);
  if (PRO == PRO_BWD) return v2 > 0.f ? fmaf(a, v, fmaf(b, v2, c)) : 0.f;
  return v;
}

// stage a [R x C x CIN] window of `in` (origin gy0,gx0; out-of-bounds -> 0 AFTER the prologue) into LDS
template <int CIN, int PRO, int R, int C>
__device__ __forceinline__ void stage_tile(float* __restrict__ lds, const float* __restrict__ in,
                                           const float* __restrict__ in2, const float* coef, int b, int Hi, int Wi,
                                           int gy0, int gx0) {
  const int t = threadIdx.x;
  if constexpr (CIN % 4 == 0) {
    constexpr int Q = CIN / 4;
    constexpr int NV = R * C * Q;
    for (int v = t; v < NV; v += 256) {
      const int pix = v / Q, q = v - pix * Q;
      const int r = pix / C, c = pix - r * C;
      const int gy = gy0 + r, gx = gx0 + c;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gy >= 0 && gy < Hi && gx >= 0 && gx < Wi) {
        const size_t off = (((size_t)b * Hi + gy) * Wi + gx) * CIN + 4 * q;
        const float4 x = *reinterpret_cast<const float4*>(in + off);
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
        if (PRO == PRO_BWD) y = *reinterpret_cast<const float4*>(in2 + off);
        const float* ca = coef + 4 * q;
        o.x = prologue<PRO>(x.x, y.x, ca[0], ca[32 + 0], ca[64 + 0]);
        o.y = prologue<PRO>(x.y, y.y, ca[1], ca[32 + 1], ca[64 + 1]);
        o.z = prologue<PRO>(x.z, y.z, ca[2], ca[32 + 2], ca[64 + 2]);
        o.w = prologue<PRO>(x.w, y.w, ca[3], ca[32 + 3], ca[64 + 3]);
      }
      *reinterpret_cast<float4*>(lds + (size_t)pix * CIN + 4 * q) = o;
    }
  } else {
    constexpr int NV = R * C * CIN;
    for (int v = t; v < NV; v += 256) {
      const int pix = v / CIN, ch = v - pix * CIN;
      const int r = pix / C, c = pix - r * C;
      const int gy = gy0 + r, gx = gx0 + c;
      float o = 0.f;
      if (gy >= 0 && gy < Hi && gx >= 0 && gx < Wi) {
        const size_t off = (((size_t)b * Hi + gy) * Wi + gx) * CIN + ch;
        const float x = in[off];
        const float y = PRO == PRO_BWD ? in2[off] : 0.f;
        o = prologue<PRO>(x, y, coef[ch], coef[32 + ch], coef[64 + ch]);
      }
      lds[v] = o;
    }
  }
}

struct WgradArgs {
  const float* x;      // raw layer input [B,Hi,Wi,CIN]; prologue 0 with xa, xb
  const float* xa;
  const float* xb;
  const float* dy;     // [B,Ho,Wo,COUT]
  const float* dy2;
  const float* da;
  const float* db;
  const float* dc;
  float* partials;     // [grid][9*CIN*COUT + COUT]
  int B, Hi, Wi, Ho, Wo;
  int tiles_y, tiles_x, ntiles;
};

// all 14 weight-gradient reductions in one launch (model.hip)
struct WgradReduceEntry {
  const float* partials;
  float* dw;
  float* dbias;
  int nparts, cin, cout, kind;
  int block0;            // first block of this entry
};
struct WgradReduceTable {
  WgradReduceEntry e[14];
  int n;
};
int ava_conv_wgrad_reduce_all(const WgradReduceTable& tab, int total_blocks, hipStream_t st);
