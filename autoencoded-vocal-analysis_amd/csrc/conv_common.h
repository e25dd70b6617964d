// Shared pieces of the VALU (conv.hip) and matrix-core (conv_mfma.hip) 3x3 gather convolutions.
#pragma once
#include <type_traits>
#include "common.h"
#include "bn_fuse.h"
#include "bn_acc.h"

enum { MODE_S1 = 0, MODE_DOWN = 1, MODE_UP = 2 };
enum { PRO_BN = 0, PRO_BWD = 1, PRO_ID = 2 };
enum { EPI_FWD = 0, EPI_BWD = 1, EPI_SSE = 2, EPI_NONE = 3 };   // EPI_NONE: store only (internal)

// conv1 recomputed inside the consumer of its output (conv_recomp.h): when G1 != null the launch's 8-channel input
// tensor does not exist in memory; `in` / `x` is the raw 1-channel spectrogram batch instead
struct RecompArgs {
  const float* G1 = nullptr;       // conv1 weights, gather layout [9][1][8]; null: no recomputation
  const float* bias1 = nullptr;    // [8]
  const float* pa1 = nullptr;      // bn1 scale / shift (one channel)
  const float* pb1 = nullptr;
};

// convt7's forward also forms convt7's weight / bias gradient partials and the BatchNorm-backward sums of its input (they need
// only the input windows and the seed prec * (xhat - x), both in registers there): conv_thin_kernels.h, FOLD
struct ThinFold {
  float* wg_partials = nullptr;    // [part_rows][73]; null: no fold
  long long* acc_out = nullptr;    // BatchNorm-backward sums {sum dx, sum dx * xhat} of the layer input (bn_acc.h slot); null: bn_partials
  float* bn_partials = nullptr;    // [part_rows][16] when acc_out is null
  const float* mean = nullptr;     // batch statistics of the layer input when the prologue reads coefficient arrays (fin.acc null)
  const float* invstd = nullptr;
};

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
  int part_rows; // rows of `partials` the caller sized ([ava_conv_grid]); rows beyond the launched grid are zero-filled
  int act_bf16;  // 1: activations (layer inputs / saved outputs) are stored as bfloat16 (see ava_bf16 below)
  long long* acc_out;  // != null: the per-channel sums of the epilogue are accumulated here (bn_acc.h) instead of partial rows
  BnFin fin;     // fin.acc != null: the prologue coefficients are derived from accumulated sums instead of pa / pb / pc
  RecompArgs rc; // rc.G1 != null: `in` is the raw spectrogram batch x and the kernel recomputes y1 = relu(conv1(bn1 x)) from it
  ThinFold fold; // fold.wg_partials != null (convt7's training forward): the launch also leaves the layer's weight-gradient partials behind
  int dbg;   // AVA_DBG ablation bits (diagnostic builds of the experiments in DESIGN.md): 1 skip MFMA, 2 skip staging, 4 skip stores
#ifdef AVA_LAB
  unsigned long long* stamps;   // lab: 16 s_memrealtime stamps of this launch (workgroup 0), tools/lab/conv_stamps.py
#endif
};
#ifdef AVA_LAB
#define AVA_STAMP(i, cond) do { if (a.stamps != nullptr && blockIdx.x == 0 && (cond)) a.stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define AVA_STAMP(i, cond) do { } while (0)
#endif

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

// Register-staged tile prefetch: `load` issues the global loads of a [R x C x CIN] window (raw values,
// out-of-bounds lanes remembered in a bit mask), `store` applies the prologue and writes LDS.  Splitting
// the two lets a workgroup keep the NEXT tile's loads in flight while it multiplies the current one.
// Everything that does not depend on the tile (which window element a thread owns, its offset relative
// to the window origin) is computed once in `init`; per tile a load costs a handful of integer ops.
typedef float avaf4 __attribute__((ext_vector_type(4)));

// Plain (compiler-visible) 16-byte load.  An inline-asm "asynchronous" variant that hides the load from
// hipcc's waitcnt pass was tried and rejected: the compiler is then free to copy/spill the destination
// registers before the data has landed (observed: stale BatchNorm-backward sums in one kernel), and the
// ablation runs show the matrix-core phase, not the memory phase, bounds these kernels anyway.
__device__ __forceinline__ avaf4 ava_load_f4_async(const float* p) {
  return *reinterpret_cast<const avaf4*>(p);
}
template <int N>
__device__ __forceinline__ void ava_wait_vm0(avaf4 (&r)[N]) {}

// ---- activation storage type (BASELINE configs[4]: "bf16 conv + fp32 ELBO") ------------------------------------
// Activations between the convolutions (the tensors kept for the backward pass: the bulk of the step's HBM traffic)
// can be stored as bfloat16: a kernel template parameter ACT selects how they are loaded / stored, everything is
// computed in fp32 (matrix cores accumulate fp32; BatchNorm statistics, gradients, ELBO, Adam stay fp32).  A bf16
// value is the upper half of its fp32 pattern; stores round to nearest even (v_cvt_pk_bf16_f32).
typedef unsigned short ava_bf16;

template <typename T> __device__ __forceinline__ avaf4 ava_ld4(const T* p);
template <> __device__ __forceinline__ avaf4 ava_ld4<float>(const float* p) { return *reinterpret_cast<const avaf4*>(p); }
template <> __device__ __forceinline__ avaf4 ava_ld4<ava_bf16>(const ava_bf16* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return avaf4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
               __uint_as_float(u.y & 0xffff0000u)};
}
// the value as it will read back from storage of type T
template <typename T> __device__ __forceinline__ float ava_stored(float v);
template <> __device__ __forceinline__ float ava_stored<float>(float v) { return v; }
template <> __device__ __forceinline__ float ava_stored<ava_bf16>(float v) { return (float)(__bf16)v; }
template <typename T> __device__ __forceinline__ void ava_st4(T* p, avaf4 v);
template <> __device__ __forceinline__ void ava_st4<float>(float* p, avaf4 v) { *reinterpret_cast<avaf4*>(p) = v; }
template <> __device__ __forceinline__ void ava_st4<ava_bf16>(ava_bf16* p, avaf4 v) {
  typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
  const bf4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  *reinterpret_cast<bf4*>(p) = b;
}
// WRITE-THROUGH form of the 16-byte activation store (sc1): the line goes to memory as it is written instead of staying dirty
// in the XCD's L2 until the end-of-kernel release writes everything back at once (a dependent kernel boundary costs + dirty
// bytes / 6 TB/s: MI355X_MICROARCH.md, row "boundary").  Only for stores whose wave instruction writes whole 128-byte lines.
// lab A/B: -DAVA_WT_STORES=1
#ifndef AVA_WT_STORES
#define AVA_WT_STORES 0
#endif
template <typename T> __device__ __forceinline__ void ava_st4_wt(T* p, avaf4 v) {
#if AVA_WT_STORES
  if constexpr (std::is_same<T, float>::value) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    return;
  }
#endif
  ava_st4<T>(p, v);
}
// typed view of an untyped (float*) argument: offsets are in ELEMENTS either way
template <typename T> __device__ __forceinline__ const T* ava_as(const float* p) { return reinterpret_cast<const T*>(p); }
template <typename T> __device__ __forceinline__ T* ava_as(float* p) { return reinterpret_cast<T*>(p); }

// TIN / TIN2: storage types of `in` and of `in2` (the saved activation of prologue PRO_BWD)
// MAXQP: most coefficient sets a thread keeps in registers for a whole call (see QP below); beyond that the sets are read
// from LDS element by element
template <int CIN, int PRO, int R, int C, bool PLANES = false, int NT = 256, typename TIN = float, typename TIN2 = float, int MAXQP = 4>
struct TileStager {
  static_assert(CIN % 4 == 0, "vector staging needs a multiple of 4 channels");
  static constexpr int Q = CIN / 4;
  static constexpr int NV = R * C * Q;
  static constexpr int NPF = (NV + NT - 1) / NT;
  static_assert(NPF <= 32, "element masks are 32 bits wide");
  // a BatchNorm prologue on a bfloat16-stored input: the layer computes in bf16 arithmetic (see value_one)
  static constexpr bool BF16_MATH = PRO == PRO_BN && std::is_same<TIN, ava_bf16>::value;
  avaf4 v[NPF];
  avaf4 v2[PRO == PRO_BWD ? NPF : 1];
  int rc[NPF];          // (r << 16) | c of the owned window element (clamped duplicate for idle lanes)
  int q4[NPF];          // 4 * channel quad
  unsigned live;        // bit i: element i exists (idx < NV)
  unsigned inb;         // bit i: element i of the CURRENT register contents is inside the image
  int tid;              // index of this thread among the NT staging threads

  __device__ __forceinline__ void init(int tid_ = (int)threadIdx.x) {
    live = 0u;
    tid = tid_;
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      int idx = tid + NT * i;
      if (idx < NV) live |= 1u << i;
      else idx = NV - 1;
      const int pix = idx / Q, q = idx - pix * Q;
      const int r = pix / C, c = pix - r * C;
      rc[i] = (r << 16) | c;
      q4[i] = 4 * q;
    }
  }

  // Branch-free: every lane always loads (out-of-range coordinates are clamped to a valid address and
  // masked at store time).  The loads are asynchronous (see ava_load_f4_async).
  __device__ __forceinline__ void load(const float* __restrict__ in, const float* __restrict__ in2, int b, int Hi,
                                       int Wi, int gy0, int gx0) {
    unsigned nb = 0u;
    const TIN* __restrict__ base = ava_as<TIN>(in) + (size_t)b * Hi * Wi * CIN;
    const TIN2* __restrict__ base2 = PRO == PRO_BWD ? ava_as<TIN2>(in2) + (size_t)b * Hi * Wi * CIN : nullptr;
#pragma unroll
    for (int i = 0; i < NPF; ++i) load_one(i, base, base2, Hi, Wi, gy0, gx0, nb);
    inb = nb;
  }

  // element i of the tile at (b, gy0, gx0) into register i; its inside-the-image bit into `nb`
  __device__ __forceinline__ void load_one(int i, const TIN* __restrict__ base, const TIN2* __restrict__ base2, int Hi, int Wi,
                                           int gy0, int gx0, unsigned& nb) {
    const int gy = gy0 + (rc[i] >> 16), gx = gx0 + (rc[i] & 0xffff);
    const bool ok = ((live >> i) & 1u) && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
    const int cy = min(max(gy, 0), Hi - 1), cx = min(max(gx, 0), Wi - 1);
    const int off = (cy * Wi + cx) * CIN + q4[i];
    v[i] = ava_ld4<TIN>(base + off);
    if (PRO == PRO_BWD) v2[i] = ava_ld4<TIN2>(base2 + off);
    nb |= ok ? (1u << i) : 0u;
  }

  // Prologue coefficients of element i's channel quad, read from LDS as vectors and UNCONDITIONALLY: written as
  // `ok ? prologue(x, coef[..]) : 0` the compiler may not speculate the LDS reads and emits, per element, four exec-masked
  // blocks of { ds_read2_b32; s_waitcnt lgkmcnt(0); fma } -- four serialised LDS round trips (measured: 1.9 us of a
  // 2.9 us tile step of conv4's forward).  Where NT is a multiple of the quads per pixel every element of a
  // thread has the same quad and the three vectors are fetched once per call instead of once per element.
  // In general a thread's elements cycle through QP = Q / gcd(NT mod Q, Q) quads (24 channels: 6 quads, 256 threads -> 3):
  // the QP coefficient sets are fetched once per call (Coefs) and element i takes set i mod QP.
  static constexpr int gcd_(int a, int b) { return b == 0 ? a : gcd_(b, a % b); }
  static constexpr int QP = (NT % Q == 0) ? 1 : Q / gcd_(NT % Q, Q);
  static constexpr bool SAMEQ = QP <= MAXQP;        // coefficient sets held in registers for the whole call
  static constexpr int NKQ = SAMEQ ? (QP < NPF ? QP : NPF) : 1;
  struct Coef { avaf4 a, b, c; };
  struct Coefs { Coef k[NKQ]; };
  __device__ __forceinline__ Coefs coefs_of(const float* __restrict__ coef) const {
    Coefs r;
#pragma unroll
    for (int j = 0; j < NKQ; ++j) r.k[j] = coef_of(j, coef);
    return r;
  }
  __device__ __forceinline__ Coef coef_of(int i, const float* __restrict__ coef) const {
    Coef k;
    const float* ca = coef + q4[i];
    k.a = k.b = k.c = avaf4{0.f, 0.f, 0.f, 0.f};
    if (PRO != PRO_ID) {
      k.a = avaf4{ca[0], ca[1], ca[2], ca[3]};
      k.b = avaf4{ca[32], ca[33], ca[34], ca[35]};
    }
    if (PRO == PRO_BWD) k.c = avaf4{ca[64], ca[65], ca[66], ca[67]};
    return k;
  }
  // prologue of element i (zero outside the image)
  __device__ __forceinline__ avaf4 value_one(int i, const Coef& k) const {
    const avaf4 x = v[i];
    const avaf4 y = PRO == PRO_BWD ? v2[i] : x;
    const bool ok = (inb >> i) & 1u;
    avaf4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float p = prologue<PRO>(x[e], y[e], k.a[e], k.b[e], k.c[e]);
      // bf16 ARITHMETIC of the convolutions (act_dtype = bfloat16, BASELINE configs[4] "bf16 conv"): the BatchNorm output --
      // the operand of the layer's products, forward and weight gradient -- is rounded to bfloat16 (nearest even)
      if (BF16_MATH) p = ava_stored<ava_bf16>(p);
      o[e] = ok ? p : 0.f;
    }
    return o;
  }

  __device__ __forceinline__ void store_one(int i, float* __restrict__ lds, const float* __restrict__ coef, const Coefs& kq) {
    const int idx = tid + NT * i;
    const avaf4 o = SAMEQ ? value_one(i, kq.k[i % NKQ]) : value_one(i, coef_of(i, coef));
    // PLANES: channel quad q of every pixel in its own [R*C][4] plane, so lanes that walk along x read 16-byte
    // slots 16 bytes apart (conflict-free ds_read_b128) instead of CIN*4 bytes apart
    const int dst = PLANES ? ((idx % Q) * (R * C) + idx / Q) : idx;
    if ((live >> i) & 1u) *reinterpret_cast<avaf4*>(lds + 4 * dst) = o;
  }

  // wait for the loads, apply the prologue, write LDS
  __device__ __forceinline__ void store(float* __restrict__ lds, const float* __restrict__ coef) {
    ava_wait_vm0(v);
    if (PRO == PRO_BWD) ava_wait_vm0(v2);
    const Coefs kq = coefs_of(coef);
#pragma unroll
    for (int i = 0; i < NPF; ++i) store_one(i, lds, coef, kq);
  }

  // store() of the tile in the registers and load() of the tile at (b, gy0, gx0), register by register: element i of the
  // next tile is requested as soon as element i of this one has been converted, so the next tile's loads travel under
  // this tile's conversion (a wave's loads return in order: the wait in front of element i leaves the NPF - 1 younger
  // requests in flight).  With store() followed by load() the whole tile was requested only after the whole conversion
  // and the staging waves -- which have nothing else to do -- sat out its full latency (measured additive: DESIGN.md 3).
  __device__ __forceinline__ void store_load(float* __restrict__ lds, const float* __restrict__ coef,
                                             const float* __restrict__ in, const float* __restrict__ in2, int b, int Hi,
                                             int Wi, int gy0, int gx0) {
    unsigned nb = 0u;
    const TIN* __restrict__ base = ava_as<TIN>(in) + (size_t)b * Hi * Wi * CIN;
    const TIN2* __restrict__ base2 = PRO == PRO_BWD ? ava_as<TIN2>(in2) + (size_t)b * Hi * Wi * CIN : nullptr;
    const Coefs kq = coefs_of(coef);
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      store_one(i, lds, coef, kq);
      load_one(i, base, base2, Hi, Wi, gy0, gx0, nb);
      __builtin_amdgcn_sched_barrier(0);
    }
    inb = nb;
  }
};

// ---- fp32 values as three bf16 limbs (matrix-core kernels on v_mfma_f32_16x16x32_bf16) -----------------------------
// x = x0 + x1 + x2 exactly (round-to-nearest at each cut; gemm_limb.hip has the error analysis): a product keeps the six
// limb pairs with i + j <= 2 and is fp32-faithful at 6/16 of the fp32 MFMA's matrix time.
typedef __bf16 ava_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ava_bf16x2v __attribute__((ext_vector_type(2)));
typedef float ava_f32x2v __attribute__((ext_vector_type(2)));
typedef uint32_t ava_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t ava_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float ava_limb_sub(float a, float b) {      // one v_sub_f32 (never paired into v_pk_add_f32)
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void ava_limb_split2(float x, float y, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
  p0 = __builtin_bit_cast(uint32_t, __builtin_convertvector((ava_f32x2v){x, y}, ava_bf16x2v));
  float rx = ava_limb_sub(x, __uint_as_float(p0 << 16)), ry = ava_limb_sub(y, __uint_as_float(p0 & 0xffff0000u));
  p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector((ava_f32x2v){rx, ry}, ava_bf16x2v));
  rx = ava_limb_sub(rx, __uint_as_float(p1 << 16));
  ry = ava_limb_sub(ry, __uint_as_float(p1 & 0xffff0000u));
  p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector((ava_f32x2v){rx, ry}, ava_bf16x2v));
}

// Octet-plane stride of the limb images, in pixels (16-byte slots).  ds_read_b128 is served in lane groups that pair 8 lanes
// of one k-group with 8 lanes of the next ({0-3,12-15 | 20-27}, {4-11 | 16-19,28-31}, ...: MI355X_MICROARCH.md, LDS): the
// two halves cover all 64 banks exactly once iff the k-groups' addresses differ by a multiple of 256 bytes.  Where a
// chunk's neighbouring k-groups are two channel octets of the same tap (16 / 32 input channels always, 24 two times out of
// three) that difference is the octet-plane stride; -DAVA_PLANE_PAD=1 rounds it up to 16 pixels.  Measured in round 5 (same
// box, whole step, twice each, alternating): tight stride 1440.5 / 1444.0 / 1446.0 / 1447.2 us of kernels per step, padded
// 1445.0 / 1451.4 / 1453.3 / 1447.5 -- the 6-16 % of LDS cycles the counters attribute to bank conflicts in these kernels
// (profiles/r05/pmc_sq.json) are hidden behind the waves' other waits; the tight stride (less LDS) stays.
#ifndef AVA_PLANE_PAD
#define AVA_PLANE_PAD 0
#endif
__host__ __device__ constexpr int ava_plane_pix(int npix) { return AVA_PLANE_PAD ? ((npix + 15) & ~15) : npix; }

// TileStager for the limb kernels: same loads, prologue and masks; the tile lands in LDS as three limb planes, each
// [CIN / 8 channel octets][R * C pixels][8 channels] bf16 -- a pixel's octet is one 16-byte slot and the 16 pixels of a
// matrix-core group are 256 contiguous bytes (conflict-free ds_read_b128 fragments).
// NL: limb planes written.  3: fp32 values, split exactly.  1: the values are bfloat16 already (the rounded BatchNorm output of
// the bf16-arithmetic mode, TileStager::BF16_MATH): one plane, no split.
template <typename TIN, int PRO> constexpr int ava_stager_limbs() { return (PRO == PRO_BN && std::is_same<TIN, ava_bf16>::value) ? 1 : 3; }
template <int CIN, int PRO, int R, int C, int NT = 256, typename TIN = float, typename TIN2 = float, int MAXQP = 4>
struct TileStagerL : TileStager<CIN, PRO, R, C, false, NT, TIN, TIN2, MAXQP> {
  using Base = TileStager<CIN, PRO, R, C, false, NT, TIN, TIN2, MAXQP>;
  static_assert(CIN % 8 == 0, "limb planes are made of channel octets");
  static constexpr int NL = ava_stager_limbs<TIN, PRO>();
  static constexpr int NPIX = ava_plane_pix(R * C), Q8 = CIN / 8;      // octet-plane stride in pixels
  static constexpr int PLANE_BYTES = Q8 * NPIX * 16;
  static constexpr int TILE_BYTES = NL * PLANE_BYTES;
  using Coef = typename Base::Coef;
  using Coefs = typename Base::Coefs;
  __device__ __forceinline__ void store_one(int i, unsigned char* __restrict__ lds, const float* __restrict__ coef, const Coefs& kq) {
    const int idx = this->tid + NT * i;
    const avaf4 o = Base::SAMEQ ? this->value_one(i, kq.k[i % Base::NKQ]) : this->value_one(i, this->coef_of(i, coef));
    const int q = idx % Base::Q, pix = idx / Base::Q;
    unsigned char* d = lds + ((q >> 1) * NPIX + pix) * 16 + (q & 1) * 8;
    if constexpr (NL == 1) {
      ava_u32x2 p0;                                    // exact: value_one has rounded the values to bfloat16
      p0[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector((ava_f32x2v){o[0], o[1]}, ava_bf16x2v));
      p0[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector((ava_f32x2v){o[2], o[3]}, ava_bf16x2v));
      if ((this->live >> i) & 1u) *reinterpret_cast<ava_u32x2*>(d) = p0;
    } else {
      ava_u32x2 p0, p1, p2;
      uint32_t a, b, c;
      ava_limb_split2(o[0], o[1], a, b, c); p0[0] = a; p1[0] = b; p2[0] = c;
      ava_limb_split2(o[2], o[3], a, b, c); p0[1] = a; p1[1] = b; p2[1] = c;
      if ((this->live >> i) & 1u) {
        *reinterpret_cast<ava_u32x2*>(d) = p0;
        *reinterpret_cast<ava_u32x2*>(d + PLANE_BYTES) = p1;
        *reinterpret_cast<ava_u32x2*>(d + 2 * PLANE_BYTES) = p2;
      }
    }
  }
  __device__ __forceinline__ void store(unsigned char* __restrict__ lds, const float* __restrict__ coef) {
    const Coefs kq = this->coefs_of(coef);
#pragma unroll
    for (int i = 0; i < Base::NPF; ++i) store_one(i, lds, coef, kq);
  }
  // store() with the elements kept apart in the schedule: interleaving the conversions of all NPF elements costs a role-split
  // kernel's staging waves registers they do not have (conv_fused_limb.hip)
  __device__ __forceinline__ void store_tight(unsigned char* __restrict__ lds, const float* __restrict__ coef) {
    const Coefs kq = this->coefs_of(coef);
#pragma unroll
    for (int i = 0; i < Base::NPF; ++i) {
      store_one(i, lds, coef, kq);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // store_tight() that also leaves the RAW values (before the prologue) of the window's interior [RI x CIr] at offset
  // (ROFF, COFF) in `raw` as fp32 [RI][CIr][CIN]: the fused backward's data-gradient waves take the x of their dx pixels from
  // there for the BatchNorm-backward sums instead of loading it from global memory a second time (conv_fused_limb.hip)
  template <int ROFF, int COFF, int RI, int CIr>
  __device__ __forceinline__ void store_tight_raw(unsigned char* __restrict__ lds, const float* __restrict__ coef, float* __restrict__ raw) {
    const Coefs kq = this->coefs_of(coef);
#pragma unroll
    for (int i = 0; i < Base::NPF; ++i) {
      store_one(i, lds, coef, kq);
      const int r = (this->rc[i] >> 16) - ROFF, c = (this->rc[i] & 0xffff) - COFF;
      if (((this->live >> i) & 1u) && (unsigned)r < (unsigned)RI && (unsigned)c < (unsigned)CIr)
        *reinterpret_cast<avaf4*>(raw + (r * CIr + c) * CIN + this->q4[i]) = this->v[i];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // see TileStager::store_load
  __device__ __forceinline__ void store_load(unsigned char* __restrict__ lds, const float* __restrict__ coef,
                                             const float* __restrict__ in, const float* __restrict__ in2, int b, int Hi,
                                             int Wi, int gy0, int gx0) {
    unsigned nb = 0u;
    const TIN* __restrict__ base = ava_as<TIN>(in) + (size_t)b * Hi * Wi * CIN;
    const TIN2* __restrict__ base2 = PRO == PRO_BWD ? ava_as<TIN2>(in2) + (size_t)b * Hi * Wi * CIN : nullptr;
    const Coefs kq = this->coefs_of(coef);
#pragma unroll
    for (int i = 0; i < Base::NPF; ++i) {
      store_one(i, lds, coef, kq);
      this->load_one(i, base, base2, Hi, Wi, gy0, gx0, nb);
      __builtin_amdgcn_sched_barrier(0);
    }
    this->inb = nb;
  }
};

// Workgroups of one resident wave of a persistent kernel: occupancy (workgroups per CU) x CUs.  A persistent grid
// larger than that runs in rounds whose last one is partly empty, and pays the per-workgroup set-up (weights into
// registers, pipeline fill) once per round.
template <typename K>
static inline int ava_resident_grid(K kernel, size_t lds_bytes, int block_threads = 256) {
  int per_cu = 0, dev = 0, cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kernel), block_threads, lds_bytes) != hipSuccess ||
      per_cu < 1)
    per_cu = 1;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    cus = prop.multiProcessorCount;
  return per_cu * cus;
}

// XCD-aware walk over a persistent workgroup's tile list.  Workgroups are dispatched round-robin over the 8 XCDs
// (workgroup w runs on XCD w % 8), each with its own L2; handing every XCD one contiguous eighth of the tile list
// keeps the tiles that share halo rows (neighbours inside one image) in the same L2 instead of fetching the halo
// from HBM once per XCD.  Falls back to the plain strided walk when the grid or tile count is not a multiple of 8.
struct TileWalk {
  int cur, end, step;
  // chunked: each XCD (workgroup index mod 8) walks its own contiguous eighth of the tile list, so that tiles sharing
  // halo rows share an L2; otherwise the workgroups sweep the list together (better for halo-free, store-heavy kernels)
  // (w, g): this workgroup's index and the number of workgroups that share the tile list; default: the launch grid
  __device__ __forceinline__ explicit TileWalk(int ntiles, bool chunked = true, int w_ = -1, int g_ = 0) {
    const int g = w_ >= 0 ? g_ : (int)gridDim.x, w = w_ >= 0 ? w_ : (int)blockIdx.x;
    if (chunked && ((g | ntiles) & 7) == 0) {
      const int chunk = ntiles >> 3, xcd = w & 7;
      cur = xcd * chunk + (w >> 3);
      end = (xcd + 1) * chunk;
      step = g >> 3;
    } else {
      cur = w; end = ntiles; step = g;
    }
  }
  __device__ __forceinline__ bool valid() const { return cur < end; }
  __device__ __forceinline__ bool has_next() const { return cur + step < end; }
  __device__ __forceinline__ int next() const { return cur + step; }
  __device__ __forceinline__ void advance() { cur += step; }
};

// optional accumulator hook-up of one conv launch (model driver): see bn_acc.h
struct ConvAcc {
  long long* acc_out;    // producer side (null: partial rows)
  BnFin fin;             // consumer side (fin.acc null: coefficient arrays)
  RecompArgs rc;         // conv2's forward: y1 recomputed from x (conv_recomp.h)
  ThinFold fold;         // convt7's training forward: weight-gradient partials + BatchNorm-backward sums from the same launch
};

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
  int act_bf16;        // x and dy2 (activations) are stored as bfloat16
};

// one weight-gradient launch as data (model.hip defers the 16 x 16 layers' calls and issues them in pairs)
struct WgradCall {
  const float *x, *xa, *xb, *dy, *dy2, *da, *db, *dc;
  float* partials;
  int Hi, Wi, Cin, Cout, mode, dy_pro;
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
  BnFin fin0;            // fin0.acc != null: block 0 also finalises bn1's own gradient (d gamma, d beta) from the sums
};                       // conv1's backward accumulated (bn_acc.h) -- the last BatchNorm of the backward pass has no consumer kernel
int ava_conv_wgrad_reduce_all(const WgradReduceTable& tab, int total_blocks, hipStream_t st);
