// Building blocks shared by the matrix-core convolution kernels (conv_mfma.hip: forward / backward-data and
// weight-gradient kernels; conv_fused.hip: the fused backward kernel): tap sets of the three gather patterns,
// the per-lane weight fragments of the implicit GEMM (ClassFrag) and the weight-gradient accumulators (WClass).
#pragma once
#include "conv_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- tap sets ------------------------------------------------------------------------------------
// S1 / DOWN: one class with all 9 taps.  UP: class = (oy&1)*2 + (ox&1); a tap (ky,kx) exists for
// oy parity py iff (py ? ky != 1 : ky == 1), same for kx.
template <int MODE> __host__ __device__ constexpr int n_classes() { return MODE == MODE_UP ? 4 : 1; }
template <int MODE> __host__ __device__ constexpr int n_taps(int cls) {
  return MODE != MODE_UP ? 9 : (cls == 0 ? 1 : (cls == 3 ? 4 : 2));
}
// t-th tap of a class -> (ky, kx)
template <int MODE> __host__ __device__ constexpr int tap_ky(int cls, int t) {
  if (MODE != MODE_UP) return t / 3;
  const int py = cls >> 1, px = cls & 1;
  const int nkx = px ? 2 : 1;
  const int iy = t / nkx;
  return py ? 2 * iy : 1;
}
template <int MODE> __host__ __device__ constexpr int tap_kx(int cls, int t) {
  if (MODE != MODE_UP) return t % 3;
  const int px = cls & 1;
  const int nkx = px ? 2 : 1;
  const int ix = t % nkx;
  return px ? 2 * ix : 1;
}

// per-lane, per-class constants: LDS offsets of the chunk reads and the A (weight) fragments
// MTO > 0: the fragment covers only MTO cout tiles starting at tile `mtb` (init argument) -- used when the cout tiles
// of a layer are dealt out to different waves.
template <int CIN, int COUT, int MODE, int CLS, int IC, int MTO = 0>
struct ClassFrag {
  static constexpr int KTOT = n_taps<MODE>(CLS) * CIN;
  static constexpr int NCH = (KTOT + 15) / 16;
  static constexpr int MT = MTO > 0 ? MTO : (COUT + 15) / 16;
  int off[NCH];
  float w[NCH][4][MT];

  // lane_base: LDS offset (floats) of this lane's pixel inside a 16-pixel group (n * CIN * stride)
  // q: bf16 arithmetic (act_dtype = bfloat16): the weights are rounded to bfloat16 once, here
  __device__ __forceinline__ void init(const float* __restrict__ G, int lane, int lane_base, int mtb = 0, bool q = false) {
    const int m = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int k = 16 * c + 4 * kg;
      const bool valid = k < KTOT;
      const int t = valid ? k / CIN : 0;
      const int ci = valid ? k - t * CIN : 0;
      int ky = 0, kx = 0;
      // (ky,kx) of the t-th tap of this class; t is lane dependent only through kg (<= 4 values)
#pragma unroll
      for (int tt = 0; tt < n_taps<MODE>(CLS); ++tt)
        if (tt == t) { ky = tap_ky<MODE>(CLS, tt); kx = tap_kx<MODE>(CLS, tt); }
      int dr, dc;
      if (MODE == MODE_UP) { dr = ky == 0 ? 1 : 0; dc = kx == 0 ? 1 : 0; }
      else { dr = ky; dc = kx; }
      off[c] = lane_base + (dr * IC + dc) * CIN + ci;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int co = 16 * (mtb + mt) + m;
          const float wv = (valid && co < COUT) ? G[((ky * 3 + kx) * CIN + ci + j) * COUT + co] : 0.f;
          w[c][j][mt] = q ? ava_stored<ava_bf16>(wv) : wv;
        }
    }
    // Retire the weight loads HERE.  Left pending, their first use sits inside the tile loop and hipcc's
    // conservative loop handling turns it into s_waitcnt vmcnt(0) there, which also drains the next tile's
    // prefetch (vmcnt retires in order) and serialises memory against the matrix cores.
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(w[c][j][mt]));
  }

  // acc[2][MT] += over all chunks; px = LDS address of this lane's pixel (tap (0,0), channel 0)
  __device__ __forceinline__ void run(const float* __restrict__ px, f32x4 (&acc)[2][MT]) const {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float4 b = *reinterpret_cast<const float4*>(px + off[c]);
      const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[j & 1][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c][j][mt], bv[j], acc[j & 1][mt], 0, 0, 0);
    }
  }
};

// ---- weight-gradient accumulators: M = (tap, ci) rows, N = cout, K = pixels (see conv_mfma.hip) ----
template <int CIN, int COUT, int MODE, int CLS, int IC>
struct WClass {
  static constexpr int KROWS = n_taps<MODE>(CLS) * CIN;
  static constexpr int MTK = (KROWS + 15) / 16;
  static constexpr int NT = (COUT + 15) / 16;
  int offA[MTK];
  f32x4 acc[MTK][NT];

  __device__ __forceinline__ static void tap_of_row(int mm, int& tapg, int& ci, int& dr, int& dc) {
    const int t = mm / CIN;
    ci = mm - t * CIN;
    int ky = 0, kx = 0;
#pragma unroll
    for (int tt = 0; tt < n_taps<MODE>(CLS); ++tt)
      if (tt == t) { ky = tap_ky<MODE>(CLS, tt); kx = tap_kx<MODE>(CLS, tt); }
    tapg = ky * 3 + kx;
    if (MODE == MODE_UP) { dr = ky == 0 ? 1 : 0; dc = kx == 0 ? 1 : 0; }
    else { dr = ky; dc = kx; }
  }

  __device__ __forceinline__ void init(int lane) {
    const int m = lane & 15;
#pragma unroll
    for (int mt = 0; mt < MTK; ++mt) {
      const int mm = 16 * mt + m;
      int tapg, ci, dr, dc;
      tap_of_row(mm < KROWS ? mm : 0, tapg, ci, dr, dc);
      offA[mt] = (dr * IC + dc) * CIN + ci;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }

  __device__ __forceinline__ void step(const float* __restrict__ xa, const float (&bf)[NT]) {
    float af[MTK];
#pragma unroll
    for (int mt = 0; mt < MTK; ++mt) af[mt] = xa[offA[mt]];
#pragma unroll
    for (int mt = 0; mt < MTK; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
  }

  // add (or store, first == true) this wave's accumulators into the LDS row in gather layout
  __device__ __forceinline__ void flush(float* __restrict__ wacc, int lane, bool first) const {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MTK; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mm = 16 * mt + 4 * kg + r;
        if (mm < KROWS) {
          int tapg, ci, dr, dc;
          tap_of_row(mm, tapg, ci, dr, dc);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int co = 16 * nt + n;
            if (co < COUT) {
              float* p = wacc + (tapg * CIN + ci) * COUT + co;
              *p = first ? acc[mt][nt][r] : *p + acc[mt][nt][r];
            }
          }
        }
      }
  }
};


// ---- the same accumulators with the M tiles dealt out to the four waves ---------------------------------------
// Unit u = (tiles of the classes before CLS) + tile index belongs to wave u % 4.  A wave then keeps a quarter of the
// accumulators (14 tiles x 2 x 4 = 112 VGPRs -> 32 for the 24-channel layers), sweeps ALL pixels of a tile instead
// of a quarter of them, and its rows of the partial result are final: no cross-wave reduction at the end.
template <int MODE> __host__ __device__ constexpr int wsplit_base(int cls, int cin) {
  int b = 0;
  for (int c = 0; c < cls; ++c) b += (n_taps<MODE>(c) * cin + 15) / 16;
  return b;
}
template <int CIN, int COUT, int MODE, int CLS, int IC>
struct WSplit {
  static constexpr int KROWS = n_taps<MODE>(CLS) * CIN;
  static constexpr int MTK = (KROWS + 15) / 16;
  static constexpr int NT = (COUT + 15) / 16;
  static constexpr int BASE = wsplit_base<MODE>(CLS, CIN);
  static constexpr int MTL = (MTK + 3) / 4;          // most tiles any wave owns
  int first;                                         // first owned tile; the wave owns first, first+4, ...
  int offA[MTL];
  f32x4 acc[MTL][NT];

  __device__ __forceinline__ void init(int lane, int wave) {
    const int m = lane & 15;
    first = (wave - BASE % 4 + 4) & 3;
#pragma unroll
    for (int i = 0; i < MTL; ++i) {
      const int mm = 16 * (first + 4 * i) + m;
      int tapg, ci, dr, dc;
      WClass<CIN, COUT, MODE, CLS, IC>::tap_of_row(mm < KROWS ? mm : 0, tapg, ci, dr, dc);
      offA[i] = (dr * IC + dc) * CIN + ci;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[i][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  __device__ __forceinline__ bool owns_any() const { return first < MTK; }

  __device__ __forceinline__ void step(const float* __restrict__ xa, const float (&bf)[NT]) {
#pragma unroll
    for (int i = 0; i < MTL; ++i)
      if (first + 4 * i < MTK) {                      // wave-uniform
        const float af = xa[offA[i]];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[nt], acc[i][nt], 0, 0, 0);
      }
  }

  // this wave's rows of the partial result, gather layout, plain stores (rows are disjoint between waves)
  __device__ __forceinline__ void flush(float* __restrict__ wacc, int lane) const {
    const int n = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int i = 0; i < MTL; ++i)
      if (first + 4 * i < MTK) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mm = 16 * (first + 4 * i) + 4 * kg + r;
          if (mm < KROWS) {
            int tapg, ci, dr, dc;
            WClass<CIN, COUT, MODE, CLS, IC>::tap_of_row(mm, tapg, ci, dr, dc);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const int co = 16 * nt + n;
              if (co < COUT) wacc[(tapg * CIN + ci) * COUT + co] = acc[i][nt][r];
            }
          }
        }
      }
  }
};

// ---- two output rows in one 16-row MFMA tile (stride-1 layers with 8 output channels) ----------------------------
// With Cout = 8 half of the 16 M rows of v_mfma_f32_16x16x4 are padding.  PairFrag puts output row y in M rows 0..7
// and output row y+1 in rows 8..15: both read the same input pixels (B operand), so K walks the FOUR input rows
// y-1 .. y+2 (x 3 taps x Cin) and the A operand holds W[ky = dr - half] where that tap exists, zero otherwise:
// 12 Cin/4 MFMAs for two rows of 16 pixels instead of 2 x 9 Cin/4 -- a third fewer, and no idle lanes in the epilogue.
template <int CIN, int IC>
struct PairFrag {
  static constexpr int KTOT = 12 * CIN;
  static constexpr int NCH = (KTOT + 15) / 16;
  static constexpr int MT = 1;
  int off[NCH];
  float w[NCH][4][1];

  // G: gather weights [9][CIN][8]; lane_base: LDS offset of this lane's pixel inside the 16-pixel group
  __device__ __forceinline__ void init(const float* __restrict__ G, int lane, int lane_base, int /*mtb*/ = 0, bool q = false) {
    const int m = lane & 15, kg = lane >> 4;
    const int half = m >> 3, co = m & 7;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int k = 16 * c + 4 * kg;
      const bool valid = k < KTOT;
      const int kk = valid ? k : 0;
      const int dr = kk / (3 * CIN), rem = kk - dr * 3 * CIN;
      const int kx = rem / CIN, ci = rem - kx * CIN;
      const int ky = dr - half;
      off[c] = lane_base + (dr * IC + kx) * CIN + ci;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float wv = (valid && ky >= 0 && ky <= 2) ? G[((ky * 3 + kx) * CIN + ci + j) * 8 + co] : 0.f;
        w[c][j][0] = q ? ava_stored<ava_bf16>(wv) : wv;
        asm volatile("" ::"v"(w[c][j][0]));          // retire before the tile loop (see ClassFrag::init)
      }
    }
  }

  __device__ __forceinline__ void run(const float* __restrict__ px, f32x4 (&acc)[2][1]) const {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float4 b = *reinterpret_cast<const float4*>(px + off[c]);
      const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j & 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c][j][0], bv[j], acc[j & 1][0], 0, 0, 0);
    }
  }
};

// ---- the same fragments on the bf16 matrix cores with three-limb operands (conv_common.h) --------------------------
// K = (tap, channel octet): lane (m / n, kg) of v_mfma_f32_16x16x32_bf16 holds 8 consecutive k = the 8 channels of one
// octet at one tap; a chunk of 4 k-groups is one MFMA per limb pair.  Weights: three bf16x8 limbs per (chunk, cout tile)
// in registers; activations: one 16-byte LDS read per limb and chunk (TileStagerL planes).
// W2L: the weights' THIRD limb (used by one of the six products) lives in an LDS table instead of 4 registers per chunk and
// tile: one more ds_read_b128 per chunk buys NCH x MT x 4 VGPRs (the role-split fused backward, conv_fused_limb.hip).  A
// lane reads back exactly the slot it wrote itself, so waves that build the same fragments may share one table.
// NLW / NLB: limbs of the weights / of the LDS operand.  3 / 3: fp32-faithful (six products).  1 / 1: bf16 arithmetic, forward
// (weights rounded to bfloat16, one plane of rounded BatchNorm outputs: one product).  1 / 3: bf16 arithmetic, data gradient
// (rounded weights against an fp32 gradient in three planes: three products -- the exact derivative of the rounded forward).
template <int CIN, int COUT, int MODE, int CLS, int IC, int NPIX, int MTO = 0, bool W2L = false, int NLW = 3, int NLB = 3>
struct ClassFragL {
  static_assert(CIN % 8 == 0, "channel octets");
  static_assert((NLW == 3 && NLB == 3) || (NLW == 1 && !W2L && (NLB == 1 || NLB == 3)), "limb sets: 3 x 3, 1 x 1, 1 x 3");
  static constexpr int Q8 = CIN / 8;
  static constexpr int KG = n_taps<MODE>(CLS) * Q8;
  static constexpr int NCH = (KG + 3) / 4;
  static constexpr int MT = MTO > 0 ? MTO : (COUT + 15) / 16;
  static constexpr int PLANE_BYTES = Q8 * NPIX * 16;
  static constexpr int W2_BYTES = NCH * MT * 1024;      // W2L: size of the third-limb table
  int off[NCH];                         // byte offset of this lane's 16-byte slot relative to the group's first pixel, plane 0
  ava_bf16x8 w[NCH][NLW == 1 ? 1 : (W2L ? 2 : 3)][MT];
  const unsigned char* w2p;             // W2L: this lane's slot of (chunk 0, tile 0); slot (c, mt) is (c * MT + mt) KB further

  // lane_pix: pixel offset of this lane's pixel inside a 16-pixel group (n * stride)
  __device__ __forceinline__ void init(const float* __restrict__ G, int lane, int lane_pix, int mtb = 0, unsigned char* w2tab = nullptr) {
    w2p = w2tab + lane * 16;
    const int m = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int g = 4 * c + kg;
      const bool valid = g < KG;
      const int t = valid ? g / Q8 : 0;
      const int c8 = valid ? g - t * Q8 : 0;
      int ky = 0, kx = 0;
#pragma unroll
      for (int tt = 0; tt < n_taps<MODE>(CLS); ++tt)
        if (tt == t) { ky = tap_ky<MODE>(CLS, tt); kx = tap_kx<MODE>(CLS, tt); }
      int dr, dc;
      if (MODE == MODE_UP) { dr = ky == 0 ? 1 : 0; dc = kx == 0 ? 1 : 0; }
      else { dr = ky; dc = kx; }
      off[c] = (c8 * NPIX + lane_pix + dr * IC + dc) * 16;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int co = 16 * (mtb + mt) + m;
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          wv[j] = (valid && co < COUT) ? G[((ky * 3 + kx) * CIN + c8 * 8 + j) * COUT + co] : 0.f;
        ava_u32x4 p0, p1, p2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint32_t a, b, d;
          ava_limb_split2(wv[2 * j], wv[2 * j + 1], a, b, d);
          p0[j] = a; p1[j] = b; p2[j] = d;
        }
        asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2));      // weights final before the tile loop (see ClassFrag::init)
        w[c][0][mt] = __builtin_bit_cast(ava_bf16x8, p0);      // (limb 0 alone IS the weight rounded to bfloat16: NLW == 1)
        if constexpr (NLW == 3) {
          w[c][1][mt] = __builtin_bit_cast(ava_bf16x8, p1);
          if constexpr (W2L) *reinterpret_cast<ava_u32x4*>(w2tab + ((c * MT + mt) * 64 + lane) * 16) = p2;
          else w[c][2][mt] = __builtin_bit_cast(ava_bf16x8, p2);
        }
      }
    }
  }
  __device__ __forceinline__ ava_bf16x8 w2(int c, int mt) const {
    if constexpr (W2L) return *reinterpret_cast<const ava_bf16x8*>(w2p + (c * MT + mt) * 1024);
    else return w[c][NLW == 3 ? 2 : 0][mt];
  }

  // px: LDS byte address of the group's first pixel (channel octet 0, limb plane 0)
  __device__ __forceinline__ void run(const unsigned char* __restrict__ px, f32x4 (&acc)[2][MT]) const {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const ava_bf16x8 b0 = *reinterpret_cast<const ava_bf16x8*>(px + off[c]);
      if constexpr (NLB == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[c & 1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][0][mt], b0, acc[c & 1][mt], 0, 0, 0);
      } else {
        const ava_bf16x8 b1 = *reinterpret_cast<const ava_bf16x8*>(px + off[c] + PLANE_BYTES);
        const ava_bf16x8 b2 = *reinterpret_cast<const ava_bf16x8*>(px + off[c] + 2 * PLANE_BYTES);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          f32x4 a = acc[c & 1][mt];                // smallest terms first
          if constexpr (NLW == 1) {
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][0][mt], b2, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][0][mt], b1, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][0][mt], b0, a, 0, 0, 0);
          } else {
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2(c, mt), b0, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][0][mt], b2, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][1][mt], b1, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][1][mt], b0, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][0][mt], b1, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][0][mt], b0, a, 0, 0, 0);
          }
          acc[c & 1][mt] = a;
        }
      }
    }
  }
};

// two output rows in one tile (stride 1, 8 output channels), limb form: K walks 4 input rows x 3 taps x channel octets
template <int CIN, int IC, int NPIX, bool W2L = false, int NLW = 3, int NLB = 3>
struct PairFragL {
  static_assert(CIN % 8 == 0, "channel octets");
  static_assert((NLW == 3 && NLB == 3) || (NLW == 1 && !W2L && (NLB == 1 || NLB == 3)), "limb sets: 3 x 3, 1 x 1, 1 x 3");
  static constexpr int Q8 = CIN / 8;
  static constexpr int KG = 12 * Q8;
  static constexpr int NCH = (KG + 3) / 4;
  static constexpr int MT = 1;
  static constexpr int PLANE_BYTES = Q8 * NPIX * 16;
  static constexpr int W2_BYTES = NCH * 1024;           // W2L: size of the third-limb table (see ClassFragL)
  int off[NCH];
  ava_bf16x8 w[NCH][NLW == 1 ? 1 : (W2L ? 2 : 3)][1];
  const unsigned char* w2p;

  __device__ __forceinline__ void init(const float* __restrict__ G, int lane, int lane_pix, int /*mtb*/ = 0, unsigned char* w2tab = nullptr) {
    w2p = w2tab + lane * 16;
    const int m = lane & 15, kg = lane >> 4;
    const int half = m >> 3, co = m & 7;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int g = 4 * c + kg;
      const bool valid = g < KG;
      const int gg = valid ? g : 0;
      const int dr = gg / (3 * Q8), rem = gg - dr * 3 * Q8;
      const int kx = rem / Q8, c8 = rem - kx * Q8;
      const int ky = dr - half;
      off[c] = (c8 * NPIX + lane_pix + dr * IC + kx) * 16;
      float wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        wv[j] = (valid && ky >= 0 && ky <= 2) ? G[((ky * 3 + kx) * CIN + c8 * 8 + j) * 8 + co] : 0.f;
      ava_u32x4 p0, p1, p2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t a, b, d;
        ava_limb_split2(wv[2 * j], wv[2 * j + 1], a, b, d);
        p0[j] = a; p1[j] = b; p2[j] = d;
      }
      asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2));
      w[c][0][0] = __builtin_bit_cast(ava_bf16x8, p0);
      if constexpr (NLW == 3) {
        w[c][1][0] = __builtin_bit_cast(ava_bf16x8, p1);
        if constexpr (W2L) *reinterpret_cast<ava_u32x4*>(w2tab + (c * 64 + lane) * 16) = p2;
        else w[c][2][0] = __builtin_bit_cast(ava_bf16x8, p2);
      }
    }
  }
  __device__ __forceinline__ ava_bf16x8 w2(int c) const {
    if constexpr (W2L) return *reinterpret_cast<const ava_bf16x8*>(w2p + c * 1024);
    else return w[c][NLW == 3 ? 2 : 0][0];
  }

  __device__ __forceinline__ void run(const unsigned char* __restrict__ px, f32x4 (&acc)[2][1]) const {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const ava_bf16x8 b0 = *reinterpret_cast<const ava_bf16x8*>(px + off[c]);
      f32x4 a = acc[c & 1][0];
      if constexpr (NLB == 1) {
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][0][0], b0, a, 0, 0, 0);
      } else {
        const ava_bf16x8 b1 = *reinterpret_cast<const ava_bf16x8*>(px + off[c] + PLANE_BYTES);
        const ava_bf16x8 b2 = *reinterpret_cast<const ava_bf16x8*>(px + off[c] + 2 * PLANE_BYTES);
        if constexpr (NLW == 1) {
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][0][0], b2, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][0][0], b1, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][0][0], b0, a, 0, 0, 0);
        } else {
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2(c), b0, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][0][0], b2, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][1][0], b1, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][1][0], b0, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][0][0], b1, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c][0][0], b0, a, 0, 0, 0);
        }
      }
      acc[c & 1][0] = a;
    }
  }
};
