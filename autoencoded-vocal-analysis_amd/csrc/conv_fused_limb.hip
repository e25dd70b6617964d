// Fused backward of one conv / convT layer with BOTH products on the bf16 matrix cores (gfx950, v_mfma_f32_16x16x32_bf16),
// fp32 operands as three bf16 limbs (x = x0 + x1 + x2 exactly; six limb products with i + j <= 2, fp32 accumulate: as close
// to fp64 as the fp32 MFMA it replaces -- conv_common.h, gemm_limb.hip) at 6/16 of the fp32 MFMA's matrix time.
// Same contract as conv3x3_bwd_fused_ws_kernel (conv_fused.hip): one staging of the dU window and of the layer-input window
// feeds the data gradient (+ BatchNorm-backward sums) and the weight / bias gradient (reference: autograd's convolution
// backward behind loss.backward(), ava/models/vae.py:352).
//
// What is different from the fp32 kernel:
//  * LDS tiles are limb planes ([limb][channel octet][pixel][8] bf16, written by TileStagerL): a pixel's octet is one 16-byte
//    slot.  The data-gradient phase reads them row-wise (ds_read_b128: 8 channels of one pixel = 8 consecutive k of the
//    implicit GEMM, ClassFragL / PairFragL).  The weight-gradient phase sums over PIXELS (K = 32 consecutive dU pixels per
//    MFMA), i.e. it needs 8 consecutive pixels of one channel per lane -- the transpose of the same image -- and reads it with
//    ds_read_b64_tr_b16 (a 4-pixel x 16-channel block delivered column-major): no second image, no shuffles.
//      A operand: M = 16 rows of (tap, ci), lane group g / lane 4q+p supplies the address of pixel k = 8g + 4h + q (+ tap
//      offset of rows 4p..4p+3), channels ci(4p)..+3 -- 8 contiguous bytes of that pixel's slot; B operand: N = 16 output
//      channels, same pixels, channels 4p..4p+3 of the dU slot.  D = dG rows (tap, ci) x co, the layout WClass::flush writes.
//    The bias gradient is one more M row whose A operand is the constant 1 (limb 0 only): three MFMAs per K step.
//  * The matrix-core waves are split by role: ND data-gradient waves (limb weights of the flipped kernel in registers, dx
//    stores, BatchNorm-backward sums) and 4 - ND weight-gradient waves (accumulators persist over all tiles of the workgroup).
//    The weight-gradient waves deal the M tiles among themselves (each sweeps ALL pixels of a tile for its rows), so a wave's
//    rows of the partial result are final: no cross-wave reduction, the accumulators go to the partial row straight from
//    registers.  Neither role's registers are live in the other's waves -- three limbs of weights (up to 84 VGPRs) and the
//    accumulators (up to 80) do not fit one 128-register wave together.
#include <stdlib.h>
#include <type_traits>
#include "conv_mfma.h"
#include "conv_fused.h"
#include "conv_recomp.h"

typedef short ava_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char ava_lds_u8;
typedef __attribute__((address_space(3))) ava_s16x4 ava_lds_s16x4;

// 8 bf16 of one MFMA operand fragment by two transposed LDS reads: k sub-blocks 0..3 at p, 4..7 at p + HB
template <int HB>
__device__ __forceinline__ ava_bf16x8 ava_lds_tr8(ava_lds_u8* p) {
  const ava_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<ava_lds_s16x4*>(p));
  const ava_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<ava_lds_s16x4*>(p + HB));
  return __builtin_bit_cast(ava_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// six limb products of one 16 x 16 x 32 step, smallest terms first (the order of ClassFragL::run)
__device__ __forceinline__ f32x4 ava_limb_mfma6(const ava_bf16x8 (&a)[3], const ava_bf16x8 (&b)[3], f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
  return c;
}

template <int LMODE, int TW, int TH>
struct FGeomL {
  // x window [XR x XC], dU window [DR x DC], dU interior at offset (DOFF, DOFF), dx region [OH x OW]  (= conv_fused.hip: FGeom)
  static constexpr int XR = LMODE == MODE_S1 ? TH + 2 : (LMODE == MODE_DOWN ? 2 * TH + 1 : TH + 1);
  static constexpr int XC = LMODE == MODE_S1 ? TW + 2 : (LMODE == MODE_DOWN ? 2 * TW + 1 : TW + 1);
  static constexpr int DR = LMODE == MODE_S1 ? TH + 2 : (LMODE == MODE_DOWN ? TH + 1 : 2 * TH + 1);
  static constexpr int DC = LMODE == MODE_S1 ? TW + 2 : (LMODE == MODE_DOWN ? TW + 1 : 2 * TW + 1);
  static constexpr int DOFF = LMODE == MODE_DOWN ? 0 : 1;
  static constexpr int OH = LMODE == MODE_DOWN ? 2 * TH : TH;
  static constexpr int OW = LMODE == MODE_DOWN ? 2 * TW : TW;
};

template <int N> using ava_ic = std::integral_constant<int, N>;

#ifndef AVA_FL_RAWPRE
#define AVA_FL_RAWPRE 0                    // lab: unrolled group loops request the raw x of every group at the top of the tile
#endif
#ifndef AVA_FL_NTSTORE
#define AVA_FL_NTSTORE 0                   // lab: dx written with non-temporal stores
#endif
#ifndef AVA_FL_W2L
#define AVA_FL_W2L 1                       // third weight limb of the single-class data gradients in an LDS table: 1 always, 0 never, 2 where the raw-x ring still takes registers
#endif
#ifndef AVA_FL_DUNROLL
#define AVA_FL_DUNROLL 1                   // unroll factor of the data-gradient waves' rolled pixel-group loop
#endif
#ifndef AVA_FL_RAWX
#define AVA_FL_RAWX 1                      // lab: 0 = the data-gradient waves load the raw x of their dx pixels from global memory
#endif
// Dynamic LDS of one workgroup (kernel and launcher agree on it here).  RAWX: each tile buffer also holds the raw x of the dx
// region as fp32 [OH][OW][CI] (TileStagerL::store_tight_raw) wherever the resident workgroups still fit the CU's 160 KB.
template <int CI, int CO, int LMODE, int TW, int TH, int NS, int ND, int NWV, int WPS, typename ACT, bool DUREC, bool DEEP>
struct FLds {
  using FG = FGeomL<LMODE, TW, TH>;
  static constexpr int MT = (CI + 15) / 16;
  static constexpr bool PAIRL = LMODE == MODE_S1 && CI == 8 && TH % 2 == 0;
  // bf16 arithmetic (ACT = bfloat16): the layer input's planes hold the ROUNDED BatchNorm output -- one limb -- and the weights are
  // rounded to bfloat16 (no third-limb table); the gradient dU keeps three limbs
  static constexpr bool BF16M = std::is_same<ACT, ava_bf16>::value;
  static constexpr int NLX = BF16M ? 1 : 3;
  static constexpr int W2_ALL = (LMODE == MODE_DOWN || BF16M) ? 0 : (((PAIRL ? 12 : 9) * (CO / 8) + 3) / 4) * 1024 * MT;   // the third-limb table
  static constexpr size_t planes = (size_t)16 * (NLX * (CI / 8) * ava_plane_pix(FG::XR * FG::XC) + 3 * (CO / 8) * ava_plane_pix(FG::DR * FG::DC));
  static constexpr size_t raw = (size_t)FG::OH * FG::OW * CI * sizeof(float);
  static constexpr size_t rest = (192 + ND * 32 * MT + (DUREC ? DU1to8Stager<FG::DC, ACT>::LDS_FLOATS : 0)) * sizeof(float) + W2_ALL;
  static constexpr int WG_PER_CU = WPS * 4 / (NS + ND + NWV);
  // measured per layer (same box, us): conv2 78.3 -> 70.3, conv4 46.6 -> 40.7, convt3 43.5 -> 37.2, conv6 34.7 -> 31.9, convt5 61.2 -> 59.4,
  // convt4 / convt2 / convt6 +-1; the stride-1 layers with MORE output than input channels lose (conv3 53.5 -> 60.1, conv5 38.8 -> 45.2:
  // few x channels to fetch, and their staging waves are the longer role already)
  static constexpr bool RAWX = AVA_FL_RAWX && (!DEEP || DUREC) && !(LMODE == MODE_S1 && CI < CO && AVA_FL_RAWX < 2) &&
                               (2 * (planes + raw) + rest + 1024) * WG_PER_CU <= 160 * 1024;
  static constexpr size_t buf = planes + (RAWX ? raw : 0);
  static constexpr size_t lds = 2 * buf + rest;
};

// lab: compile a wave role out (register-pressure / ablation experiments; results are wrong): 1 staging, 2 data gradient, 4 weight gradient
#ifndef AVA_FL_DCUT
#define AVA_FL_DCUT 0                      // lab, timing only: 1 no raw-x ring loads, 2 no dx stores, 4 no fragment reads / MFMAs
#endif
#ifndef AVA_FL_CUT
#define AVA_FL_CUT 0
#endif

// number of M tiles (units) of the weight gradient over all tap classes
template <int LMODE> __host__ __device__ constexpr int wl_units(int cin) { return wsplit_base<LMODE>(n_classes<LMODE>(), cin); }

// ACT: storage type of the activations x (layer input) and dy2 (saved output); dy and dx are fp32 gradients.
// NS / ND / NWV: staging, data-gradient and weight-gradient waves of the workgroup (64 * (NS + ND + NWV) threads; waves are dealt
// to the four SIMDs in turn, so with NS, ND and NWV multiples of 4 every SIMD hosts the same mix of roles).
// WPS: minimum waves per SIMD for the register allocator (workgroups per CU x waves of a workgroup / 4)
// DUREC (convt6's backward): the upstream gradient dy (8 channels, full resolution) does not exist in memory -- it is convt7's
// data gradient, a 3x3 gather of the 1-channel seed a.dy, formed on the matrix cores by the (four) staging waves as they build
// the dU tile (conv_recomp.h: DU1to8Stager; its limb-plane store below).
// DEEP: the staging waves hold TWO tiles in registers (tile it+2 and it+3 in flight while tile it+1 is converted): per tile the
// matrix-core waves have ~1 us of work, a tile's loads take 2-3 us under load, so one tile in flight leaves both roles waiting
// The three roles stay INLINED in one kernel.  Compiled as three noinline functions (tools/lab/noinline_roles.patch) each role gets
// its own register allocation and the in-loop spills disappear -- but every kernel then carries 156-256 B of scratch per lane (the
// callee-saved registers of the roles), and on this stack a kernel's time grows with its scratch size: the 16 x 16 layers, spill-free
// either way, went 24.0 -> 30.6 us, the whole family +90 us (profiles/r04/ab_roles_*.csv).
template <int CI, int CO, int LMODE, int DYPRO, int TW, int TH, int NS, int ND, int NWV, int WPS, typename ACT, bool DUREC = false,
          bool DEEP = false>
__global__ __launch_bounds__(64 * (NS + ND + NWV), WPS) void conv3x3_bwd_fused_limb_kernel(const FusedArgs a) {
  using FG = FGeomL<LMODE, TW, TH>;
  constexpr int XR = FG::XR, XC = FG::XC, DR = FG::DR, DC = FG::DC, DOFF = FG::DOFF;
  constexpr int BMODE = LMODE == MODE_S1 ? MODE_S1 : (LMODE == MODE_DOWN ? MODE_UP : MODE_DOWN);   // gather pattern of dx
  constexpr int MT = (CI + 15) / 16;        // dx channel tiles
  constexpr int NT = (CO + 15) / 16;        // dG column tiles
  constexpr int NW_ = 9 * CI * CO;
  constexpr int BCLS = n_classes<BMODE>(), WCLS = n_classes<LMODE>();
  constexpr int NST = 64 * NS;              // staging threads
  static_assert(NS >= 1 && ND >= 1 && NWV >= 1 && NWV <= 4 && ND + NWV >= 4, "roles");
  static_assert(CI % 8 == 0 && CO % 8 == 0, "limb planes are made of channel octets");
  // stride-1 layers with 8 input channels: the data gradient has 8 output channels -> two dx rows per MFMA tile
  constexpr bool PAIR = LMODE == MODE_S1 && CI == 8 && TH % 2 == 0;
  // two dx channel tiles: the data-gradient waves are dealt one tile each (half the limb weights), odd and even waves
  // sharing the pixel groups
  // (not where the four waves each own a parity class of a stride-2 gather -- CSPLIT below: those hold both tiles)
  constexpr bool DSPLIT = MT == 2 && ND % 2 == 0 && !(BMODE == MODE_UP && ND == 4);
  constexpr int NDG = DSPLIT ? ND / 2 : ND;     // waves that share the pixel groups of a tile
  constexpr int MTD = DSPLIT ? 1 : MT;
  constexpr int XNPIX = ava_plane_pix(XR * XC), DNPIX = ava_plane_pix(DR * DC);     // octet-plane strides (pixels) of the two limb images
  constexpr int XPLANE = (CI / 8) * XNPIX * 16, DPLANE = (CO / 8) * DNPIX * 16;      // bytes
  using FL = FLds<CI, CO, LMODE, TW, TH, NS, ND, NWV, WPS, ACT, DUREC, DEEP>;
  constexpr bool RAWX = FL::RAWX;           // raw x of the dx region behind the planes of each tile buffer
  // bf16 arithmetic: one limb of the (rounded) layer input and of the (rounded) weights; three of the gradient.  Both products
  // are then three MFMAs instead of six, and they are the exact derivative of the forward that rounded the same operands
  constexpr bool BF16M = FL::BF16M;
  constexpr int NLX = FL::NLX, NLW = BF16M ? 1 : 3;
  constexpr int XBYTES = NLX * XPLANE, RAWOFF = XBYTES + 3 * DPLANE, BUF = (int)FL::buf;
  static_assert(FL::planes == (size_t)RAWOFF, "kernel and launcher agree on the tile buffer");
  extern __shared__ __align__(16) unsigned char smem_b[];
  float* cx = reinterpret_cast<float*>(smem_b + 2 * BUF);     // [3][32]
  float* cd = cx + 96;                                         // [3][32]
  float* red = cd + 96;                                        // [ND][32 * MT]: per data-gradient wave {sum g [16 MT], sum g x [16 MT]}
  // the data-gradient weights' third limb as an LDS table (ClassFragL: W2L) where one wave holds ALL of a class's chunks
  // (single-class gathers): the role then fits 128 VGPRs with room to spare
  constexpr bool W2L = !BF16M && BCLS == 1 && (AVA_FL_W2L == 1 || (AVA_FL_W2L == 2 && !RAWX));
  constexpr int DKG = PAIR ? 12 * (CO / 8) : 9 * (CO / 8);     // k-groups of the (single) data-gradient class
  constexpr int W2_TILE = W2L ? ((DKG + 3) / 4) * 1024 : 0;    // bytes per dx channel tile
  constexpr int W2_ALL = W2_TILE * MT;
  unsigned char* w2tab = reinterpret_cast<unsigned char*>(red + ND * 32 * MT);
  float* xs = reinterpret_cast<float*>(w2tab + W2_ALL);        // DUREC: the staging waves' private seed windows
  static_assert(!DUREC || (CO == 8 && DYPRO == PRO_BWD && DR == 9 && NS == 4), "the 1 -> 8 gather feeds a 9-row dU window of 8 channels from four staging waves");
  __shared__ double accvals[64];            // consumer prologue scratch (bn_coef_from_acc)
  __shared__ float ems[64];                 // mean [0..31], invstd [32..63] of x's BatchNorm for the final reduction

  const int t = threadIdx.x, lane = t & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(t >> 6);
  AVA_STAMP(0, t == 0);
  const bool stager = wave8 < NS;            // waves 0 .. NS-1 stage tiles, the next ND form the data gradient, the rest the weight gradient
  const int n = lane & 15, kg = lane >> 4;

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
  // (coefficient sets in registers only where a thread's elements share ONE channel quad: three sets of the 24-channel
  // tensors are 24-36 VGPRs of a role that has to fit beside two others)
  TileStagerL<CI, PRO_BN, XR, XC, NST, ACT, ACT, 1> sx;       // staging waves only (threadIdx.x 0 .. NST-1)
  typename std::conditional<DUREC, DU1to8Stager<DC, ACT>, TileStagerL<CO, DYPRO, DR, DC, NST, float, ACT, 1>>::type sd;
  auto sd_store = [&](unsigned char* dst) __attribute__((always_inline)) {
    if constexpr (DUREC) sd.store_limb(dst, cd, xs); else sd.store_tight(dst, cd);
  };
  auto prefetch = [&](int tl) {
    int b, y0, x0, gy, gx;
    origin(tl, b, y0, x0);
    x_origin(y0, x0, gy, gx);
    sx.load(a.x, nullptr, b, a.Hi, a.Wi, gy, gx);
    d_origin(y0, x0, gy, gx);
    sd.load(a.dy, a.dy2, b, a.Ho, a.Wo, gy, gx);
  };
  // (DUREC + DEEP: both register sets of the seed gather pass through the SAME private LDS window of their wave -- it is only
  // live inside one store_limb call, and a wave's two stores follow each other)
  decltype(sx) sx2;                                            // DEEP: the second register set (odd tiles)
  decltype(sd) sd2;
  auto sd2_store = [&](unsigned char* dst) __attribute__((always_inline)) {
    if constexpr (DUREC) sd2.store_limb(dst, cd, xs); else sd2.store_tight(dst, cd);
  };
  auto prefetch2 = [&](int tl) {
    if constexpr (DEEP) {
      int b, y0, x0, gy, gx;
      origin(tl, b, y0, x0);
      x_origin(y0, x0, gy, gx);
      sx2.load(a.x, nullptr, b, a.Hi, a.Wi, gy, gx);
      d_origin(y0, x0, gy, gx);
      sd2.load(a.dy, a.dy2, b, a.Ho, a.Wo, gy, gx);
    }
  };
  if (stager) {
    sx.init();
    if constexpr (DEEP) { sx2.init(); if constexpr (DUREC) sd2.init(a.rcd, xs); else sd2.init(); }
    if constexpr (DUREC) sd.init(a.rcd, xs); else sd.init();
    if (walk.valid()) prefetch(walk.cur);    // tile 0 goes in flight BEFORE the coefficient prologue
  }
  // everything the prologue and the epilogue read from global memory is requested in front of the coefficient finalisation
  // and its barrier (conv_fused.hip)
  float cxv = 0.f;
  if (t < 96) {
    const int which = t >> 5, c = t & 31;
    const float* sxp = which == 0 ? a.xa : (which == 1 ? a.xb : nullptr);
    cxv = (sxp != nullptr && c < CI) ? sxp[c] : 0.f;
  }
  static_assert(9 * CI * CO * 4 <= 64 * (ND + NWV) * 128, "one touch per thread covers the packed weights");
  float wpf = 0.f;
  if (!stager) {
    wpf = a.Gb[min(32 * (t - NST), 9 * CI * CO - 1)];
    const int e = t - NST - 64;               // the SECOND matrix-core wave: the first one finalises the coefficients
    if (e >= 0 && e < 64) {
      const int c = e & 31;
      ems[e] = c < CI ? (e < 32 ? a.mean[c] : a.invstd[c]) : 0.f;
    }
  }
  if (a.fin.acc != nullptr) {
    bn_coef_from_acc(cd, accvals, a.fin, NST);
    if (t < 96) cx[t] = cxv;
  } else if (t < 96) {
    const int which = t >> 5, c = t & 31;
    const float* sdp = which == 0 ? a.da : (which == 1 ? a.db : a.dc);
    cx[t] = cxv;
    cd[t] = (sdp != nullptr && c < CO) ? sdp[c] : 0.f;
  }
  __syncthreads();                           // cx / cd visible
  if (!stager) asm volatile("" ::"v"(wpf));
  AVA_STAMP(1, t == 0);

  if (stager && (AVA_FL_CUT & 1)) return;
  if (stager) {
    // ---------------- staging waves ----------------
    __builtin_amdgcn_s_setprio(3);           // issue priority over the matrix-core waves of the same SIMD (conv_fused.hip)
    constexpr int XOFF = LMODE == MODE_UP ? 0 : 1;              // the dx region inside the x window
    auto sx_store_of = [&](auto& stg, unsigned char* dst) __attribute__((always_inline)) {
      if constexpr (RAWX) stg.template store_tight_raw<XOFF, XOFF, FG::OH, FG::OW>(dst, cx, reinterpret_cast<float*>(dst + RAWOFF));
      else stg.store_tight(dst, cx);
    };
    if constexpr (DEEP) {
      // tile k lives in register set k & 1 and goes to LDS buffer k & 1
      auto tile_k = [&](int k) { return walk.cur + k * walk.step; };
      if (walk.valid()) {
        if (tile_k(1) < walk.end) prefetch2(tile_k(1));
        sx_store_of(sx, smem_b);
        sd_store(smem_b + XBYTES);
        if (tile_k(2) < walk.end) prefetch(tile_k(2));
      }
      __syncthreads();                                          // (A) tile 0 ready
      int it = 0;
      for (; walk.valid(); walk.advance(), ++it) {              // (tile_k is relative to the advancing walk.cur)
        if (walk.has_next()) {
          if ((it & 1) == 0) {
            sx_store_of(sx2, smem_b + BUF);
            sd2_store(smem_b + BUF + XBYTES);
            if (tile_k(3) < walk.end) prefetch2(tile_k(3));
          } else {
            sx_store_of(sx, smem_b);
            sd_store(smem_b + XBYTES);
            if (tile_k(3) < walk.end) prefetch(tile_k(3));
          }
        }
        __syncthreads();                                        // (B)
      }
      __syncthreads();                                          // (E)
      return;
    }
    auto sx_store = [&](unsigned char* dst) __attribute__((always_inline)) { sx_store_of(sx, dst); };
    if (walk.valid()) {
      sx_store(smem_b);
      sd_store(smem_b + XBYTES);
      AVA_STAMP(2, t == 0);
      if (walk.has_next()) prefetch(walk.next());
    }
    __syncthreads();                                            // (A) tile 0 ready
    int it = 0;
    for (; walk.valid(); walk.advance(), ++it) {
      if (walk.has_next()) {                                    // tile it+1 -> the other buffer, tile it+2 in flight
        unsigned char* nb = smem_b + ((it + 1) & 1) * BUF;
        sx_store(nb);
        sd_store(nb + XBYTES);
        const int nn = walk.next() + walk.step;
        if (nn < walk.end) prefetch(nn);
      }
      __syncthreads();                                          // (B)
    }
    __syncthreads();                                            // (E) the reduction barrier of the matrix-core waves
    return;
  }

  if (wave8 < NS + ND && (AVA_FL_CUT & 2)) return;
  if (wave8 < NS + ND) {
    // ---------------- data-gradient waves ----------------
    // Stride-2 conv layers (dx gathered in four output-parity classes): with four data-gradient waves each wave takes ONE
    // class for all of the tile's pixel groups and holds only that class's limb weights (1-2 chunks instead of 5: with all
    // four classes in every wave the role needed > 128 VGPRs, and spill reloads inside a latency-bound loop cost 4x the tile).
    constexpr bool CSPLIT = BMODE == MODE_UP && ND == 4;
    const int dw = wave8 - NS;
    auto d_role = [&](auto dcls_c) __attribute__((always_inline)) {
      constexpr int DCLS = decltype(dcls_c)::value;               // CSPLIT: this wave's class; otherwise 0
      const int mtb = DSPLIT ? (dw & 1) : 0;                      // first dx channel tile of this wave
      const int dgi = CSPLIT ? 0 : (DSPLIT ? (dw >> 1) : dw);     // which share of the pixel groups
      constexpr int SPB = BMODE == MODE_DOWN ? 2 : 1;
      typename std::conditional<PAIR, PairFragL<CO, DC, DNPIX, W2L, NLW, 3>, ClassFragL<CO, CI, BMODE, (CSPLIT ? DCLS : 0), DC, DNPIX, MTD, W2L, NLW, 3>>::type f0;
      ClassFragL<CO, CI, BMODE, (BCLS > 1 && !CSPLIT ? 1 : 0), DC, DNPIX, MTD, false, NLW, 3> f1;
      ClassFragL<CO, CI, BMODE, (BCLS > 1 && !CSPLIT ? 2 : 0), DC, DNPIX, MTD, false, NLW, 3> f2;
      ClassFragL<CO, CI, BMODE, (BCLS > 1 && !CSPLIT ? 3 : 0), DC, DNPIX, MTD, false, NLW, 3> f3;
      f0.init(a.Gb, lane, SPB * n, mtb, w2tab + mtb * W2_TILE);
      if (BCLS > 1 && !CSPLIT) { f1.init(a.Gb, lane, n, mtb); f2.init(a.Gb, lane, n, mtb); f3.init(a.Gb, lane, n, mtb); }
      const int lane_out = PAIR ? ((kg >> 1) * a.Wi + n) * CI + 4 * (kg & 1) : (BMODE == MODE_UP ? 2 * n : n) * CI + 4 * kg;
      const int cq = PAIR ? 4 * (kg & 1) : 4 * kg;    // first dx channel of this lane inside its channel tile
      float s1[MTD][4], s2[MTD][4];
#pragma unroll
      for (int mt = 0; mt < MTD; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[mt][r] = s2[mt][r] = 0.f;

      // pixel groups of the dx region: 16 consecutive pixels of a row (UP pattern: of one parity class)
      constexpr int CB = FG::OW / (BMODE == MODE_UP ? 32 : 16);          // column blocks
      constexpr int GROUPS = PAIR ? (FG::OH / 2) * CB : (BMODE == MODE_UP ? 4 * (FG::OH / 2) * CB : FG::OH * CB);
      constexpr int GPW = CSPLIT ? GROUPS / 4 : GROUPS / NDG;
      static_assert(GROUPS % NDG == 0 && (BMODE != MODE_UP || CSPLIT || GPW % 4 == 0), "tile must split evenly over the data-gradient waves");
      // the gi-th group of this wave
      auto group_of = [&](int gi) -> int { return CSPLIT ? 4 * gi + DCLS : dgi * GPW + gi; };
      auto group_out = [&](int g) -> int {
        if (BMODE == MODE_UP) {
          const int cls = g & 3, rest = g >> 2, r = rest / CB, cb = rest % CB;
          return ((2 * r + (cls >> 1)) * a.Wi + 32 * cb + (cls & 1)) * CI;
        }
        return (((PAIR ? 2 : 1) * (g / CB)) * a.Wi + 16 * (g % CB)) * CI;
      };
      // RAWX: the raw x of this lane's dx pixels comes from the tile buffer (same index arithmetic with the region's width)
      const int lane_raw = PAIR ? ((kg >> 1) * FG::OW + n) * CI + 4 * (kg & 1) : (BMODE == MODE_UP ? 2 * n : n) * CI + 4 * kg;
      auto group_raw = [&](int g) -> int {
        if (BMODE == MODE_UP) {
          const int cls = g & 3, rest = g >> 2, r = rest / CB, cb = rest % CB;
          return ((2 * r + (cls >> 1)) * FG::OW + 32 * cb + (cls & 1)) * CI;
        }
        return (((PAIR ? 2 : 1) * (g / CB)) * FG::OW + 16 * (g % CB)) * CI;
      };
      // raw x at this lane's dx pixels (BatchNorm-backward sums): a ring of LA groups in flight -- group gi's values are
      // requested LA groups ahead of their use (the lines were fetched by the staging waves a tile or two earlier: L2 hits),
      // running on into the first groups of the next tile.
      constexpr int LA = GPW < 3 ? GPW : 3;
      // the group loop stays rolled where a wave has many groups (unrolled 8 times the register allocator spilled 200
      // values inside the loop); the class of a group must be a compile-time constant unless the wave owns one class.
      // Four groups: measured per shape -- rolled wins where the wave also carries the runtime channel-tile offset of
      // DSPLIT (convt3 49.4 -> 42.6 us), unrolled elsewhere (conv4 45.5 vs 55.7, convt5 61.7 vs 69.2 us)
      constexpr bool ROLLED = (GPW > 4 || (GPW > 2 && DSPLIT)) && (BMODE != MODE_UP || CSPLIT);
      avaf4 ring[LA][MTD];
      auto load_ex_group = [&](const ACT* __restrict__ xb, int gi, avaf4 (&dst)[MTD]) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MTD; ++mt) {
          const int cb4 = 16 * (mtb + mt) + cq;
          // lanes whose 4-channel slot lies beyond CI re-read slot 0: stays in bounds
          dst[mt] = ava_ld4<ACT>(xb + group_out(group_of(gi)) + (cb4 < CI ? lane_out + 16 * (mtb + mt) : lane_out - 4 * kg));
        }
      };
      auto ex_base = [&](int tl) -> const ACT* {
        int b, y0, x0;
        origin(tl, b, y0, x0);
        const int oy0 = LMODE == MODE_DOWN ? 2 * y0 : y0, ox0 = LMODE == MODE_DOWN ? 2 * x0 : x0;
        return ava_as<ACT>(a.x) + (((size_t)b * a.Hi + oy0) * a.Wi + ox0) * CI;
      };
      if (!RAWX && walk.valid()) {
        const ACT* __restrict__ xb = ex_base(walk.cur);
#pragma unroll
        for (int gi = 0; gi < LA; ++gi) load_ex_group(xb, gi, ring[gi]);
      }
      if (AVA_FL_DCUT & 1) {
#pragma unroll
        for (int gi = 0; gi < LA; ++gi)
#pragma unroll
          for (int mt = 0; mt < MTD; ++mt) ring[gi][mt] = (avaf4){1.f, 1.f, 1.f, 1.f};
      }
      AVA_STAMP(3, t == NST);
      __syncthreads();                                              // (A)
      AVA_STAMP(4, t == NST);
      int it = 0;
      for (; walk.valid(); walk.advance(), ++it) {
        int b, y0, x0;
        origin(walk.cur, b, y0, x0);
        const unsigned char* dut = smem_b + (it & 1) * BUF + XBYTES;
        const int oy0 = LMODE == MODE_DOWN ? 2 * y0 : y0, ox0 = LMODE == MODE_DOWN ? 2 * x0 : x0;
        const size_t tile_pix = ((size_t)b * a.Hi + oy0) * a.Wi + ox0;
        float* __restrict__ obase = a.dx + tile_pix * CI;
        const ACT* __restrict__ xcur = ava_as<ACT>(a.x) + tile_pix * CI;
        const ACT* __restrict__ xnext = (!RAWX && walk.has_next()) ? ex_base(walk.next()) : xcur;     // (last tile: harmless re-reads)
        const float* __restrict__ rawt = reinterpret_cast<const float*>(smem_b + (it & 1) * BUF + RAWOFF);
        auto raw_of = [&](int gi, int mt) -> avaf4 {
          const int cb4 = 16 * (mtb + mt) + cq;                      // lanes beyond CI re-read slot 0 (their sums are dropped)
          return *reinterpret_cast<const avaf4*>(rawt + group_raw(group_of(gi)) + (cb4 < CI ? lane_raw + 16 * (mtb + mt) : lane_raw - 4 * kg));
        };
        // unrolled group loop: the raw x of all of the tile's groups is requested in front of the first fragment read
        constexpr bool RAWPRE = RAWX && !ROLLED && AVA_FL_RAWPRE;
        avaf4 rawall[RAWPRE ? GPW : 1][MTD];
        if constexpr (RAWPRE) {
#pragma unroll
          for (int gi = 0; gi < GPW; ++gi)
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) rawall[gi][mt] = raw_of(gi, mt);
        }
        auto do_group = [&](int gi) __attribute__((always_inline)) {
          const int g = group_of(gi);
          avaf4 exv[MTD];
          if constexpr (RAWPRE) {
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) exv[mt] = rawall[gi][mt];
          } else if constexpr (RAWX) {
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) exv[mt] = raw_of(gi, mt);
          } else {
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) exv[mt] = ring[0][mt];
#pragma unroll
            for (int k = 0; k + 1 < LA; ++k)
#pragma unroll
              for (int mt = 0; mt < MTD; ++mt) ring[k][mt] = ring[k + 1][mt];
            const int gn = gi + LA;                                  // the group LA ahead: of this tile or of the next one
            if (!(AVA_FL_DCUT & 1)) load_ex_group(gn < GPW ? xcur : xnext, gn < GPW ? gn : gn - GPW, ring[LA - 1]);
          }
          f32x4 acc[2][MTD];
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) acc[h][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (AVA_FL_DCUT & 4) {
#pragma unroll
            for (int mt = 0; mt < MTD; ++mt) acc[0][mt] = (f32x4){exv[mt][0], exv[mt][1], exv[mt][2], exv[mt][3]};
          } else if (BMODE == MODE_UP) {
            const int rest = g >> 2, r = rest / CB, cb = rest % CB;
            const unsigned char* px = dut + (r * DC + 16 * cb) * 16;
            if constexpr (CSPLIT) {
              f0.run(px, acc);
            } else {
              const int cls = gi & 3;                                // GPW % 4 == 0 and the loop is unrolled: compile-time
              if (cls == 0) f0.run(px, acc);
              else if (cls == 1) f1.run(px, acc);
              else if (cls == 2) f2.run(px, acc);
              else f3.run(px, acc);
            }
          } else {
            constexpr int S = (BMODE == MODE_S1 && !PAIR) ? 1 : 2;        // PAIR: a group is a pair of dx rows
            constexpr int SX = BMODE == MODE_DOWN ? 2 : 1;
            f0.run(dut + (S * (g / CB) * DC + SX * 16 * (g % CB)) * 16, acc);
          }
          const int gout = group_out(g) + lane_out;
#pragma unroll
          for (int mt = 0; mt < MTD; ++mt) {
            const int cb4 = 16 * (mtb + mt) + cq;
            if (cb4 < CI) {
              const f32x4 v = acc[0][mt] + acc[1][mt];
              const avaf4 xr = exv[mt];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                s1[mt][r] += v[r];
                s2[mt][r] = fmaf(v[r], xr[r], s2[mt][r]);          // raw x: centred after the loop
              }
              if (!(AVA_FL_DCUT & 2)) {
                if (AVA_FL_NTSTORE) __builtin_nontemporal_store((avaf4){v[0], v[1], v[2], v[3]}, reinterpret_cast<avaf4*>(obase + gout + 16 * (mtb + mt)));
                else ava_st4_wt<float>(obase + gout + 16 * (mtb + mt), avaf4{v[0], v[1], v[2], v[3]});
              }
            }
          }
        };
        if constexpr (ROLLED) {
#pragma unroll AVA_FL_DUNROLL
          for (int gi = 0; gi < GPW; ++gi) do_group(gi);
        } else {
#pragma unroll
          for (int gi = 0; gi < GPW; ++gi) do_group(gi);
        }
        AVA_STAMP(5 + (it < 4 ? it : 4), t == NST);
        __syncthreads();                                            // (B)
      }
      AVA_STAMP(10, t == NST);
      // ---- BatchNorm-backward partial sums: over the 16 pixel lanes, then over the data-gradient waves (fixed order) ----
#pragma unroll
      for (int mt = 0; mt < MTD; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v1 = s1[mt][r], v2 = s2[mt][r];
          {
            // the hot loop accumulates sum g*x on RAW x; centred and scaled once per lane here
            const int cc = 16 * (mtb + mt) + cq + r;
            const float mu = ems[cc & 31], is = ems[32 + (cc & 31)];  // requested in the prologue (zero beyond CI)
            v2 = fmaf(-mu, v1, v2) * is;
          }
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) { v1 += __shfl_xor(v1, o, 64); v2 += __shfl_xor(v2, o, 64); }
          if (PAIR) { v1 += __shfl_xor(v1, 32, 64); v2 += __shfl_xor(v2, 32, 64); }   // the two rows of a pair
          if (n == 0 && (!PAIR || kg < 2)) {
            const int ci = 16 * (mtb + mt) + cq + r;
            red[dw * 32 * MT + ci] = v1;
            red[dw * 32 * MT + 16 * MT + ci] = v2;
          }
        }
    };
    if constexpr (CSPLIT) {
      if (dw == 0) d_role(ava_ic<0>{});
      else if (dw == 1) d_role(ava_ic<1>{});
      else if (dw == 2) d_role(ava_ic<2>{});
      else d_role(ava_ic<3>{});
    } else {
      d_role(ava_ic<0>{});
    }
    __syncthreads();                                              // (E)
    const int tc = t - NST;                                       // first data-gradient wave: 2 * CI <= 64 lanes
    if (tc < 2 * CI) {
      const int which = tc / CI, ci = tc - which * CI;
      const int idx = which * 16 * MT + ci;
      // fixed order over the waves that hold this channel (DSPLIT: waves of the channel's tile parity)
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < ND; ++w)
        if (!DSPLIT || (w & 1) == (ci >> 4)) tot += red[w * 32 * MT + idx];
      if (a.acc_out != nullptr) bn_acc_add(a.acc_out, which * 32 + ci, tot);
      else a.bn_partials[(size_t)blockIdx.x * 2 * CI + tc] = tot;
    }
    AVA_STAMP(12, t == NST);
    return;
  }

  // ---------------- weight-gradient waves ----------------
  if (AVA_FL_CUT & 4) return;
  // Units: the M tiles (16 rows of (tap, ci)) of every tap class, numbered class by class, plus the bias row as unit NU.
  // Unit u belongs to wave u % NWV, slot u / NWV.  K = 32 pixels per step: KW consecutive pixels of 32 / KW rows.
  constexpr int NU = wl_units<LMODE>(CI);
  constexpr int KW = TW >= 32 ? 32 : TW;                     // pixels of one row inside a K step
  static_assert(KW == 32 || KW == 16, "K steps are 32 pixels: one row of 32 or two rows of 16");
  constexpr int KSTEPS = TH * TW / 32;
  static_assert((TH * TW) % 32 == 0 && (KW == 32 || TH % 2 == 0), "tile must be made of whole K steps");
  constexpr int SA = LMODE == MODE_DOWN ? 2 : 1;             // x-window pixels per step pixel
  constexpr int SB = LMODE == MODE_UP ? 2 : 1;               // dU-window pixels per step pixel
  const int ww = wave8 - NS - ND;                            // wave-uniform
  ava_lds_u8* const lbase = (ava_lds_u8*)smem_b;
  // lane's pixel inside a K step (k sub-block 0): lane group g = lane >> 4 holds k = 8 g + {0..7}; lane 4 q + p of the group
  // supplies the address of k = 8 g + q (second read: + 4), columns 4 p .. 4 p + 3
  const int q = n >> 2, p = n & 3;
  const int k0 = 8 * kg + q, kr = k0 / KW, kc = k0 % KW;

  auto w_role = [&](auto ww_c) __attribute__((always_inline)) {
    constexpr int WW = decltype(ww_c)::value;
    constexpr int NOWN = (NU + 1 - WW + NWV - 1) / NWV;       // units WW, WW + NWV, ...
    constexpr bool OWNS_BIAS = NU % NWV == WW;
    int offA[NOWN > 0 ? NOWN : 1];
    f32x4 acc[NOWN > 0 ? NOWN : 1][NT];
    int offB[NT];
    // ---- per-lane LDS byte offsets (plane 0, k sub-block 0, step origin 0) ----
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      int co4 = 16 * nt + 4 * p;
      if (co4 >= CO) co4 -= 8;                                 // padding columns re-read valid channels (results unused)
      offB[nt] = ((co4 >> 3) * DNPIX + (SB * kr + DOFF) * DC + SB * kc + DOFF) * 16 + (co4 & 7) * 2;
    }
    auto for_units = [&](auto&& fn) __attribute__((always_inline)) {
      // fn(class, tile in class, unit) for every M tile, classes in order
      auto one = [&](auto cls_c) __attribute__((always_inline)) {
        constexpr int CLS = decltype(cls_c)::value;
        constexpr int MTK = (n_taps<LMODE>(CLS) * CI + 15) / 16;
        constexpr int UB = wsplit_base<LMODE>(CLS, CI);
#pragma unroll
        for (int mt = 0; mt < MTK; ++mt) fn(cls_c, mt, UB + mt);
      };
      one(ava_ic<0>{});
      if constexpr (WCLS > 1) { one(ava_ic<1>{}); one(ava_ic<2>{}); one(ava_ic<3>{}); }
    };
    for_units([&](auto cls_c, int mt, int u) __attribute__((always_inline)) {
      constexpr int CLS = decltype(cls_c)::value;
      constexpr int KROWS = n_taps<LMODE>(CLS) * CI;
      if (u % NWV == WW) {
        const int mm = 16 * mt + 4 * p;
        int tapg, ci, dr, dc;
        WClass<CI, CO, LMODE, CLS, XC>::tap_of_row(mm < KROWS ? mm : 0, tapg, ci, dr, dc);
        offA[u / NWV] = ((ci >> 3) * XNPIX + (SA * kr + dr) * XC + SA * kc + dc) * 16 + (ci & 7) * 2;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[u / NWV][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    });
    if constexpr (OWNS_BIAS) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[NU / NWV][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // the bias row's A operand: 1 in limb 0 of M row 0 (lanes 0, 16, 32, 48: all 8 k), 0 elsewhere
    ava_bf16x8 ones;
    {
      const uint32_t v = n == 0 ? 0x3f803f80u : 0u;
      const ava_u32x4 vv = {v, v, v, v};
      ones = __builtin_bit_cast(ava_bf16x8, vv);
    }

    AVA_STAMP(13, t == 64 * (NS + ND));
    __syncthreads();                                              // (A)
    int it = 0;
    for (; walk.valid(); walk.advance(), ++it) {
      ava_lds_u8* const xl = lbase + (it & 1) * BUF;
      ava_lds_u8* const dl = xl + XBYTES;
#pragma unroll 1
      for (int ks = 0; ks < KSTEPS; ++ks) {
        // step origin in step-pixel units (dU interior pixels; LMODE_UP: x pixels)
        constexpr int SPR = TW >= 32 ? TW / 32 : 1;          // K steps per tile row
        const int sy = KW == 32 ? ks / SPR : 2 * ks, sx0 = KW == 32 ? 32 * (ks % SPR) : 0;
        ava_lds_u8* const xs = xl + (SA * sy * XC + SA * sx0) * 16;
        ava_lds_u8* const ds = dl + (SB * sy * DC + SB * sx0) * 16;
        auto one = [&](auto cls_c) __attribute__((always_inline)) {
          constexpr int CLS = decltype(cls_c)::value;
          constexpr int MTK = (n_taps<LMODE>(CLS) * CI + 15) / 16;
          constexpr int UB = wsplit_base<LMODE>(CLS, CI);
          constexpr int CLSOFF = LMODE == MODE_UP ? ((CLS >> 1) * DC + (CLS & 1)) * 16 : 0;    // output parity (py, px)
          ava_bf16x8 bfr[NT][3];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int L = 0; L < 3; ++L) bfr[nt][L] = ava_lds_tr8<4 * SB * 16>(ds + offB[nt] + L * DPLANE + CLSOFF);
          if constexpr (OWNS_BIAS) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              f32x4 c = acc[NU / NWV][nt];
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bfr[nt][2], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bfr[nt][1], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bfr[nt][0], c, 0, 0, 0);
              acc[NU / NWV][nt] = c;
            }
          }
#pragma unroll
          for (int mt = 0; mt < MTK; ++mt) {
            if ((UB + mt) % NWV == WW) {
              const int sl = (UB + mt) / NWV;
              ava_bf16x8 afr[NLX];
#pragma unroll
              for (int L = 0; L < NLX; ++L) afr[L] = ava_lds_tr8<4 * SA * 16>(xs + offA[sl] + L * XPLANE);
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                if constexpr (NLX == 3) acc[sl][nt] = ava_limb_mfma6(afr, bfr[nt], acc[sl][nt]);
                else {                                           // bf16 arithmetic: the one limb of x against the three of dU
                  f32x4 c = acc[sl][nt];
                  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[0], bfr[nt][2], c, 0, 0, 0);
                  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[0], bfr[nt][1], c, 0, 0, 0);
                  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[0], bfr[nt][0], c, 0, 0, 0);
                  acc[sl][nt] = c;
                }
              }
            }
          }
        };
        one(ava_ic<0>{});
        if constexpr (WCLS > 1) { one(ava_ic<1>{}); one(ava_ic<2>{}); one(ava_ic<3>{}); }
      }
      __syncthreads();                                            // (B)
    }
    __syncthreads();                                              // (E)
    // ---- this wave's rows of the partial result, straight from the accumulators (gather layout [tap][ci][co] + bias) ----
    float* __restrict__ prow = a.wg_partials + (size_t)blockIdx.x * (NW_ + CO);
    for_units([&](auto cls_c, int mt, int u) __attribute__((always_inline)) {
      constexpr int CLS = decltype(cls_c)::value;
      constexpr int KROWS = n_taps<LMODE>(CLS) * CI;
      if (u % NWV == WW) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mm = 16 * mt + 4 * kg + r;
          if (mm < KROWS) {
            int tapg, ci, dr, dc;
            WClass<CI, CO, LMODE, CLS, XC>::tap_of_row(mm, tapg, ci, dr, dc);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const int co = 16 * nt + n;
              if (co < CO) prow[(tapg * CI + ci) * CO + co] = acc[u / NWV][nt][r];
            }
          }
        }
      }
    });
    if constexpr (OWNS_BIAS) {
      if (kg == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int co = 16 * nt + n;
          if (co < CO) prow[NW_ + co] = acc[NU / NWV][nt][0];
        }
      }
    }
  };
  if (ww == 0) w_role(ava_ic<0>{});
  else if (NWV > 1 && ww == 1) w_role(ava_ic<(NWV > 1 ? 1 : 0)>{});
  else if (NWV > 2 && ww == 2) w_role(ava_ic<(NWV > 2 ? 2 : 0)>{});
  else if (NWV > 3) w_role(ava_ic<(NWV > 3 ? 3 : 0)>{});
}

// ------------------------------------------------------------------------------------------------
template <int CI, int CO, int LMODE, int DYPRO, int TW, int TH, int NS, int ND, int NWV, int WPS, typename ACT, bool DUREC = false,
          bool DEEP = false>
static int launch_fused_limb_t(const FusedArgs& a, int grid, hipStream_t st) {
  using FL = FLds<CI, CO, LMODE, TW, TH, NS, ND, NWV, WPS, ACT, DUREC, DEEP>;
  constexpr size_t lds = FL::lds;
  constexpr int WG_PER_CU = FL::WG_PER_CU;
  static_assert(WG_PER_CU >= 1 && (lds + 1024) * WG_PER_CU <= 160 * 1024, "the resident workgroups' tile buffers must fit 160 KB of LDS");
  const void* kfn = reinterpret_cast<const void*>(&conv3x3_bwd_fused_limb_kernel<CI, CO, LMODE, DYPRO, TW, TH, NS, ND, NWV, WPS, ACT, DUREC, DEEP>);
  // (the kernel also carries < 1 KB of static LDS -- accvals, ems: the same allowance as the static_assert above; one attempt
  // per instantiation, thread-safe, its result remembered)
  static const bool attr_ok = (lds + 1024 <= 64 * 1024) ||
                              hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
  if (!attr_ok) return AVA_ELAUNCH;
  FusedArgs b = a;
#ifdef AVA_LAB
  b.stamps = ava_lab_next_stamps_f();
#endif
  const int hl = LMODE == MODE_DOWN ? a.Ho : a.Hi, wl = LMODE == MODE_DOWN ? a.Wo : a.Wi;   // low-resolution side
  if (hl % TH != 0 || wl % TW != 0) return AVA_EINVAL;
  b.tiles_y = hl / TH;
  b.tiles_x = wl / TW;
  b.ntiles = a.B * b.tiles_y * b.tiles_x;
  if (grid < 1 || grid > b.ntiles) return AVA_EINVAL;
  hipLaunchKernelGGL((conv3x3_bwd_fused_limb_kernel<CI, CO, LMODE, DYPRO, TW, TH, NS, ND, NWV, WPS, ACT, DUREC, DEEP>), dim3(grid), dim3(64 * (NS + ND + NWV)), lds, st, b);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

#ifndef AVA_FL_DEEP_MIN_NS
#define AVA_FL_DEEP_MIN_NS 99         // measured: two tiles in flight LOSE everywhere (same box, one -> two: conv3 61.1 -> 64.3 us,
                                      // convt4 41.7 -> 45.3 without a spill; convt5 61.5 -> 91, conv5 40.5 -> 62 with the spills the
                                      // second register set causes); lab: -DAVA_FL_DEEP_MIN_NS=8 builds it for the 8-stager shapes
#endif
#ifndef AVA_FL_T6_DEEP
#define AVA_FL_T6_DEEP 0              // convt6's backward (seed gather in the staging waves): two tiles in the staging registers
#endif
template <int CI, int CO, int LMODE, int TW, int TH, int NS, int ND, int NWV, int WPS>
static int launch_fused_limb(const FusedArgs& a, int grid, int dy_pro, hipStream_t st) {
  constexpr bool DEEP = NS >= AVA_FL_DEEP_MIN_NS;
  if constexpr (CI == 8 && CO == 8 && LMODE == MODE_UP && TH == 4 && NS == 4) {
    if (a.rcd.G1 != nullptr) {            // convt6's backward with convt7's data gradient formed in the staging waves
      if (dy_pro != PRO_BWD) return AVA_EINVAL;
      if (a.act_bf16) return launch_fused_limb_t<CI, CO, LMODE, PRO_BWD, TW, TH, NS, ND, NWV, WPS, ava_bf16, true, AVA_FL_T6_DEEP != 0>(a, grid, st);
      return launch_fused_limb_t<CI, CO, LMODE, PRO_BWD, TW, TH, NS, ND, NWV, WPS, float, true, AVA_FL_T6_DEEP != 0>(a, grid, st);
    }
  }
  if (a.rcd.G1 != nullptr) return AVA_EINVAL;
  if (dy_pro == PRO_BWD) {
    if (a.act_bf16) return launch_fused_limb_t<CI, CO, LMODE, PRO_BWD, TW, TH, NS, ND, NWV, WPS, ava_bf16, false, DEEP>(a, grid, st);
    return launch_fused_limb_t<CI, CO, LMODE, PRO_BWD, TW, TH, NS, ND, NWV, WPS, float, false, DEEP>(a, grid, st);
  }
  if (dy_pro == PRO_ID) {
    if (a.act_bf16) return launch_fused_limb_t<CI, CO, LMODE, PRO_ID, TW, TH, NS, ND, NWV, WPS, ava_bf16, false, DEEP>(a, grid, st);
    return launch_fused_limb_t<CI, CO, LMODE, PRO_ID, TW, TH, NS, ND, NWV, WPS, float, false, DEEP>(a, grid, st);
  }
  return AVA_EINVAL;
}

// Shapes with a limb instantiation: (cin, cout, mode) -> low-resolution tile, staging / data-gradient / weight-gradient waves,
// waves per SIMD.  Measured in the step at batch 256 (tools/run_r04.sh variants; fp32 kernel -> this one):
//   conv3  8 -> 16 S1   82.5 -> 52.4 us   1024 threads, 32 x 8 tiles (32 x 4: 59.3; the halo rows are 33 % instead of 59 % extra)
//   convt5 16 -> 8 S1   92.3 -> 62.4 us   768 threads (168 VGPRs), 32 x 8 tiles (1024 threads, 32 x 4: 68.6)
//   conv5  16 -> 24 S1  65.7 -> 43.8 us   768 threads: the data-gradient waves hold 84 VGPRs of limb weights
//   convt3 24 -> 16 S1  63.8 -> 46.2 us   the two dx channel tiles dealt to odd / even data-gradient waves
//   convt4 16 -> 16 UP  43.8 -> 41.3 us
//   conv4  16 -> 16 DOWN 49.4 -> 45.5 us  (16 x 4 tiles; each data-gradient wave owns one output-parity class)
//   conv2 / convt6: below
// (lab experiments: -DAVA_FL_CFG="th, ns, nd, nwv, wps" overrides conv3's row: tools/lab/build_variant.sh)
#ifndef AVA_FL_CFG
#define AVA_FL_CFG 32, 8, 8, 4, 4, 4          // conv3
#endif
#ifndef AVA_FL_CFG2
#define AVA_FL_CFG2 32, 8, 8, 4, 4, 4         // convt5
#endif
#ifndef AVA_FL_T3
#define AVA_FL_T3 32, 4, 8, 4, 4, 4           // convt3
#endif
#ifndef AVA_FL_T4
#define AVA_FL_T4 16, 4, 10, 4, 2, 4          // convt4 (8+4+4: 41.3 us, 10+4+2: 38.6 us)
#endif
#ifndef AVA_FL_T6
#define AVA_FL_T6 16, 4, 4, 2, 2, 4           // convt6
#endif
#ifndef AVA_FL_C2
#define AVA_FL_C2 16, 4, 3, 4, 1, 4           // conv2: one dx parity class per data-gradient wave (99.8 -> 78.3 us; 2+4+2: 80.1)
#endif
#ifndef AVA_FL_C4
#define AVA_FL_C4 16, 4, 8, 4, 4, 4           // conv4
#endif
#ifndef AVA_FL_C5
#define AVA_FL_C5 32, 4, 9, 4, 3, 4           // conv5 (8+4+4: 39.7 us, 9+4+3: 38.5 us)
#endif
// The four layers at 16 x 16 (conv6, conv7, convt1, convt2; until round 4 a data-gradient launch plus half a weight-gradient
// pair launch each).  With 24 / 32 channels on both sides the three roles together spilled 100-350 registers at 128 or 168
// VGPRs although each role fits alone (-DAVA_FL_CUT), and a spill reload inside these latency-bound loops is a dependent
// memory round trip: fused, they LOST (same box, fused against data-gradient launch + half a pair launch: conv7 34.4 vs 33.9 us,
// convt1 30.9 vs 33.6, conv6 74.3 vs 45.5, convt2 51.5 vs 45.4 at 768 threads; 33.7 / 41.5 / 84.4 / 65.3 at 1024).  ONE
// 512-thread workgroup per CU (4 staging + 2 + 2 matrix-core waves, 256 VGPRs, no spill) wins: conv6 34.6, convt2 28.8,
// conv7 24.6, convt1 23.8 us -- 111.8 us for the four against 158.4, and six launches fewer.
#ifndef AVA_FL_C7
#define AVA_FL_C7 16, 8, 4, 2, 2, 2
#endif
#ifndef AVA_FL_T1
#define AVA_FL_T1 16, 8, 4, 2, 2, 2
#endif
#ifndef AVA_FL_C6
#define AVA_FL_C6 16, 4, 4, 2, 2, 2
#endif
#ifndef AVA_FL_T2
#define AVA_FL_T2 16, 4, 4, 2, 2, 2
#endif
#define AVA_FL_ROW(X, ci, co, md, ...) X(ci, co, md, __VA_ARGS__)
#ifndef AVA_FL_NO16
#define AVA_FL_16(X) AVA_FL_ROW(X, 24, 32, MODE_S1, AVA_FL_C7) AVA_FL_ROW(X, 32, 24, MODE_S1, AVA_FL_T1) AVA_FL_ROW(X, 24, 24, MODE_DOWN, AVA_FL_C6) AVA_FL_ROW(X, 24, 24, MODE_UP, AVA_FL_T2)
#else
#define AVA_FL_16(X)
#endif
// The 8 <-> 8 stride-2 layers at full resolution (conv2, and convt6 with convt7's data gradient gathered in its staging
// waves) run 512-thread workgroups, two per CU like the fp32 kernel they replace (4 staging + 2 + 2 matrix-core waves: every
// role fits 128 VGPRs there): same box conv2 106.0 -> 99.5 us, convt6 90.1 -> 87.1 us.  As ONE workgroup per CU they lost
// (1024 threads: 112 .. 124 / 105 us; 768 threads: 99 / 124 us): a CU then has one small tile in flight instead of two.
#define AVA_FL_88(X) AVA_FL_ROW(X, 8, 8, MODE_UP, AVA_FL_T6) AVA_FL_ROW(X, 8, 8, MODE_DOWN, AVA_FL_C2)
#ifdef AVA_FL_ONLY16                            // lab: only the four 16 x 16 layers
#define AVA_FUSED_LIMB_SHAPES(X)                \
  AVA_FL_ROW(X, 24, 32, MODE_S1, AVA_FL_C7)     \
  AVA_FL_ROW(X, 32, 24, MODE_S1, AVA_FL_T1)     \
  AVA_FL_ROW(X, 24, 24, MODE_DOWN, AVA_FL_C6)   \
  AVA_FL_ROW(X, 24, 24, MODE_UP, AVA_FL_T2)
#elif defined(AVA_FL_EXP_ONLY)                  // lab: only the rows under experiment (fast variant builds)
#ifdef AVA_FL_NONE
#define AVA_FUSED_LIMB_SHAPES(X)
#else
#define AVA_FUSED_LIMB_SHAPES(X)                \
  AVA_FL_ROW(X, 8, 8, MODE_UP, AVA_FL_T6)       \
  AVA_FL_ROW(X, 8, 8, MODE_DOWN, AVA_FL_C2)
#endif
#else
#define AVA_FUSED_LIMB_SHAPES(X)                \
  AVA_FL_ROW(X, 8, 16, MODE_S1, AVA_FL_CFG)     \
  AVA_FL_ROW(X, 16, 8, MODE_S1, AVA_FL_CFG2)    \
  AVA_FL_ROW(X, 16, 16, MODE_UP, AVA_FL_T4)     \
  AVA_FL_ROW(X, 24, 16, MODE_S1, AVA_FL_T3)     \
  AVA_FL_ROW(X, 16, 24, MODE_S1, AVA_FL_C5)     \
  AVA_FL_ROW(X, 16, 16, MODE_DOWN, AVA_FL_C4)   \
  AVA_FL_16(X)                                  \
  AVA_FL_88(X)
#endif

bool ava_conv_fused_limb_has(int Cin, int Cout, int mode) {
#define X(ci, co, md, tww, thh, ns, nd, nwv, wps) if (Cin == ci && Cout == co && mode == md) return true;
  AVA_FUSED_LIMB_SHAPES(X)
#undef X
  return false;
}

// workgroups of one resident wave of the limb kernel (= its grid cap = partial rows) and its low-resolution tile;
// 0: no limb instantiation
int ava_conv_fused_limb_cap(int Cin, int Cout, int mode, int* tw, int* th) {
#define X(ci, co, md, tww, thh, ns, nd, nwv, wps) \
  if (Cin == ci && Cout == co && mode == md) { *tw = tww; *th = thh; return 256 * (wps * 4 / (ns + nd + nwv)); }
  AVA_FUSED_LIMB_SHAPES(X)
#undef X
  return 0;
}

// AVA_EINVAL when the shape has no limb instantiation (the caller then runs the fp32 kernel)
int ava_conv3x3_bwd_fused_limb_launch(const FusedArgs& a, int grid, int Cin, int Cout, int mode, int dy_pro, hipStream_t st) {
  if (a.rc.G1 != nullptr || a.dx == nullptr) return AVA_EINVAL;
#define X(ci, co, md, tww, thh, ns, nd, nwv, wps) \
  if (Cin == ci && Cout == co && mode == md) return launch_fused_limb<ci, co, md, tww, thh, ns, nd, nwv, wps>(a, grid, dy_pro, st);
  AVA_FUSED_LIMB_SHAPES(X)
#undef X
  return AVA_EINVAL;
}
