// fp32-faithful GEMM on the gfx950 bf16 matrix cores for the large products of the fully connected layers (fc1 / fc8
// forward, dX, dW: ava/models/vae.py:142,153,225,261 and their autograd products).
//
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate (MI355X_MICROARCH.md, Matrix cores).  An fp32 value is EXACTLY
// the sum of three bf16 "limbs" (8 + 8 + 8 significand bits, round-to-nearest at each cut, so |a1| <= 2^-8 |a|,
// |a2| <= 2^-17 |a|):  a = a0 + a1 + a2,  b = b0 + b1 + b2.  The product keeps the six limb pairs with i + j <= 2,
//     a b  ~  a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0 ,
// each on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; the three dropped pairs are <= 2^-24 |a b| together, i.e.
// below one fp32 rounding of the product.  Six bf16 MFMAs replace sixteen fp32 MFMAs' worth of matrix time.
//
// Tiling: BM x BN = 128 x (128 | 64) output tile, K step 32, 256 threads = 2 x 2 waves, wave tile 64 x (64 | 32) as
// 16 x 16 MFMA tiles.  Operands are split while they are staged: global fp32 (16-byte loads, either storage order)
// -> registers -> three limb planes in LDS, K contiguous ([limb][row][32 k] bf16, 64-byte rows), so that one
// ds_read_b128 is one MFMA operand fragment and a wave reads 1 KB of contiguous LDS (conflict free).  Operands stored
// with K strided (the dW products, W in the dX products) are transposed in registers (4 k x 4 columns per thread);
// their rows sit in the image with bits 0 and 2 of the row index swapped, which makes those 8-byte writes conflict
// free as well.  Double-buffered LDS, one barrier per K step, tile k+2 in flight in registers while tile k is
// multiplied (same pipeline as gemm.hip).  Split-K writes fp32 slabs that gemm.hip's fixed-order reduce kernel sums.
#include <type_traits>
#include "gemm.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define LBK 32

// (x, y) -> three packed bf16 pairs; x = x0 + x1 + x2 exactly (the two subtractions are exact in fp32, the last
// remainder has <= 8 significant bits)
// The subtractions are written as single v_sub_f32: left to itself hipcc pairs them into v_pk_add_f32, which issues at
// a third of the rate of two scalar ops beside MFMAs (MI355X_MICROARCH.md, "price of one filler beside MFMAs").
__device__ __forceinline__ float limb_sub(float a, float b) {
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void limb_split2(float x, float y, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
  p0 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){x, y}, bf16x2));
  float rx = limb_sub(x, __uint_as_float(p0 << 16)), ry = limb_sub(y, __uint_as_float(p0 & 0xffff0000u));
  p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){rx, ry}, bf16x2));
  rx = limb_sub(rx, __uint_as_float(p1 << 16));
  ry = limb_sub(ry, __uint_as_float(p1 & 0xffff0000u));
  p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){rx, ry}, bf16x2));
}

// position of logical row r (0..15) inside its 16-row block of the LDS image
template <bool PERM>
__device__ __forceinline__ int pos16(int r) {
  return PERM ? ((r & ~5) | ((r & 1) << 2) | ((r >> 2) & 1)) : r;
}
// 16-byte chunk (0..3) of a row's 64 bytes that holds k octet `oct` of logical row r.  With 64-byte rows the rows r and
// r + 4 start on the same bank, and ds_read_b128 serves lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... together
// (MI355X_MICROARCH.md, LDS): each such group holds four rows of every bank class, two per k octet.  XOR-ing the chunk
// index with a function of the row spreads them over the four chunks: conflict-free fragment reads (checked lane group
// by lane group for both images), and the writes keep their contiguity (a row's four chunks are only permuted).
template <bool PERM>
__device__ __forceinline__ int chunk_of(int r, int oct) {
  return PERM ? (oct ^ ((r & 1) << 1)) : (oct ^ ((4 - ((r >> 2) & 3)) & 3));
}

// The staging waves keep RING steps of operand tiles in registers: the loads of step s + RING - 1 are issued while step s
// is being split into LDS, so a tile has RING - 2 whole K steps (plus the current one) to arrive from L2 / HBM.

// ---- K-contiguous source: element (row, k) at base[row * ld + k].  Unit = 8 consecutive k of one row. -------------
template <int BT, int T, int LIMB_RING>
struct LimbLoaderK {
  static constexpr int UNITS = BT * 4, NU = (UNITS + T - 1) / T;
  float4 r[LIMB_RING][NU][2];
  unsigned ok[LIMB_RING];      // bit i: unit i of that slot lies inside the matrix and inside its item's K range
  size_t off[NU];              // load-side state of the item being loaded
  unsigned okrow;
  __device__ __forceinline__ void init(int t0, int extent, int ld) {
    okrow = 0u;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int u = threadIdx.x + T * i, row = u >> 2, oct = u & 3, g = t0 + row;
      if (u < UNITS && g < extent) okrow |= 1u << i;
      off[i] = (size_t)min(g, extent - 1) * ld + oct * 8;
    }
  }
  template <int SLOT>
  __device__ __forceinline__ void load(const float* __restrict__ base, int /*ld*/, int k0, int kend, int K) {
    unsigned okk = 0u;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int oct = (threadIdx.x + T * i) & 3, gk = k0 + oct * 8;
      const float* p = base + off[i] + min(k0, K - 8 - oct * 8);        // K % 8 == 0: an octet is inside or outside as a whole
      r[SLOT][i][0] = *reinterpret_cast<const float4*>(p);
      r[SLOT][i][1] = *reinterpret_cast<const float4*>(p + 4);
      if (gk < kend) okk |= 1u << i;
    }
    ok[SLOT] = okk & okrow;
  }
  template <int SLOT, bool COLSUM>
  __device__ __forceinline__ void store(unsigned char* __restrict__ S) {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int u = threadIdx.x + T * i, row = u >> 2, oct = u & 3;
      if (UNITS % T != 0 && u >= UNITS) continue;
      float4 a = r[SLOT][i][0], b = r[SLOT][i][1];
      if (!((ok[SLOT] >> i) & 1u)) { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
      u32x4 p0, p1, p2;
      uint32_t x, y, z;
      limb_split2(a.x, a.y, x, y, z); p0[0] = x; p1[0] = y; p2[0] = z;
      limb_split2(a.z, a.w, x, y, z); p0[1] = x; p1[1] = y; p2[1] = z;
      limb_split2(b.x, b.y, x, y, z); p0[2] = x; p1[2] = y; p2[2] = z;
      limb_split2(b.z, b.w, x, y, z); p0[3] = x; p1[3] = y; p2[3] = z;
      unsigned char* d = S + row * 64 + chunk_of<false>(row & 15, oct) * 16;
      *reinterpret_cast<u32x4*>(d) = p0;
      *reinterpret_cast<u32x4*>(d + BT * 64) = p1;
      *reinterpret_cast<u32x4*>(d + 2 * BT * 64) = p2;
    }
  }
};

// ---- K-strided source: element (k, col) at base[k * ld + col].  Unit = KPU k x 4 columns, transposed in registers
// (KPU = 4, or 2 where 4 would leave half of the staging threads without a unit: the 64-wide operand).
// Thread -> (kq = u % KQ, cq = u / KQ), KQ = 32 / KPU: a wave's load covers KQ rows x (64 / KQ) x 16 contiguous bytes. ----
template <int BT, int T, int LIMB_RING>
struct LimbLoaderN {
  static constexpr int KPU = (2 * BT >= T) ? 4 : 2;
  static constexpr int KQ = LBK / KPU;
  static constexpr int UNITS = (BT / 4) * KQ, NU = (UNITS + T - 1) / T;
  float4 r[LIMB_RING][NU][KPU];
  unsigned ok[LIMB_RING];      // bit KPU i + j: row j of unit i of that slot is inside the matrix / the item's K range
  int coff[NU];
  unsigned okcol;
  float cs[NU][4];             // running column sums of the item being stored (bias gradient of the dW products)
  __device__ __forceinline__ LimbLoaderN() {
#pragma unroll
    for (int i = 0; i < NU; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) cs[i][c] = 0.f;
  }
  __device__ __forceinline__ void init(int t0, int extent, int /*ld*/) {
    okcol = 0u;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int u = threadIdx.x + T * i, cq = u / KQ, g = t0 + 4 * cq;
      if (u < UNITS && g < extent) okcol |= ((1u << KPU) - 1u) << (KPU * i);   // extent % 4 == 0: a column quad is inside or outside as a whole
      coff[i] = min(g, extent - 4);
    }
  }
  template <int SLOT>
  __device__ __forceinline__ void load(const float* __restrict__ base, int ld, int k0, int kend, int K) {
    unsigned okk = 0u;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int kq = (threadIdx.x + T * i) % KQ;
#pragma unroll
      for (int j = 0; j < KPU; ++j) {
        const int gk = k0 + KPU * kq + j;
        r[SLOT][i][j] = *reinterpret_cast<const float4*>(base + (size_t)min(gk, K - 1) * ld + coff[i]);
        if (gk < kend) okk |= 1u << (KPU * i + j);
      }
    }
    ok[SLOT] = okk & okcol;
  }
  template <int SLOT, bool COLSUM>
  __device__ __forceinline__ void store(unsigned char* __restrict__ S) {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int u = threadIdx.x + T * i, kq = u % KQ, cq = u / KQ;
      if (UNITS % T != 0 && u >= UNITS) continue;
      float v[KPU][4];
#pragma unroll
      for (int j = 0; j < KPU; ++j) {
        const bool okj = (ok[SLOT] >> (KPU * i + j)) & 1u;
        v[j][0] = okj ? r[SLOT][i][j].x : 0.f; v[j][1] = okj ? r[SLOT][i][j].y : 0.f;
        v[j][2] = okj ? r[SLOT][i][j].z : 0.f; v[j][3] = okj ? r[SLOT][i][j].w : 0.f;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int row = 4 * cq + c;
        unsigned char* d = S + ((row & ~15) + pos16<true>(row & 15)) * 64;
        uint32_t x, y, z;
        if constexpr (KPU == 4) {
          if (COLSUM) cs[i][c] += (v[0][c] + v[1][c]) + (v[2][c] + v[3][c]);
          u32x2 p0, p1, p2;
          limb_split2(v[0][c], v[1][c], x, y, z); p0[0] = x; p1[0] = y; p2[0] = z;
          limb_split2(v[2][c], v[3][c], x, y, z); p0[1] = x; p1[1] = y; p2[1] = z;
          d += chunk_of<true>(row & 15, kq >> 1) * 16 + (kq & 1) * 8;
          *reinterpret_cast<u32x2*>(d) = p0;
          *reinterpret_cast<u32x2*>(d + BT * 64) = p1;
          *reinterpret_cast<u32x2*>(d + 2 * BT * 64) = p2;
        } else {
          if (COLSUM) cs[i][c] += v[0][c] + v[1][c];
          limb_split2(v[0][c], v[1][c], x, y, z);
          d += chunk_of<true>(row & 15, kq >> 2) * 16 + (kq & 3) * 4;
          *reinterpret_cast<uint32_t*>(d) = x;
          *reinterpret_cast<uint32_t*>(d + BT * 64) = y;
          *reinterpret_cast<uint32_t*>(d + 2 * BT * 64) = z;
        }
      }
    }
  }
};

template <int BT, int T, int RING, bool KMAJ>
struct LimbLoader;
template <int BT, int T, int RING>
struct LimbLoader<BT, T, RING, true> : LimbLoaderK<BT, T, RING> {};
template <int BT, int T, int RING>
struct LimbLoader<BT, T, RING, false> : LimbLoaderN<BT, T, RING> {};

// f(integral_constant<int, J>) for J = 0 .. R-1, unrolled at compile time (static ring-slot indices)
template <int J, int R, class F>
__device__ __forceinline__ void limb_for_stage(F&& f) {
  f(std::integral_constant<int, J>{});
  if constexpr (J + 1 < R) limb_for_stage<J + 1, R>(f);
}

// One work item = (split, tile row, tile column).  A workgroup walks its items as ONE continuous sequence of K steps:
// the staging waves run up to one step ahead across item boundaries, so the first loads of the next item are in flight
// while the matrix-core waves finish the current one and write it out (no LDS in the epilogue: the MFMA operands are
// swapped, D' = B^T A^T, so a lane's four accumulator registers are four consecutive columns of one output row).
struct LimbItem { int m0, n0, split, kbeg, kend, bx; };

// RING: operand tiles the staging waves keep in registers (loads RING - 1 steps ahead).  MINW: waves per SIMD the
// register allocation must allow (4 = two 512-thread workgroups per CU: the staging waves of one workgroup then run
// beside the matrix-core waves of the other, which matters because a staging wave alone is stalled half of the time).
// NSW: staging waves (4, or 8: the split is the long pole of a K step, section 3 item 17 of DESIGN.md -- two staging waves
// per SIMD share it beside one matrix-core wave); the workgroup has NSW + 4 waves.
template <int BN, bool A_KMAJ, bool B_KMAJ, int RING, int MINW, int NSW = 4>
__global__ __launch_bounds__(64 * (NSW + 4), MINW) void gemm_limb_kernel(const GemmArgs g, const int tiles_m, const int tiles_n,
                                                                          const int nitems) {
  constexpr int BM = 128, T = 64 * NSW, WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
  constexpr int A_BYTES = 3 * BM * 64, B_BYTES = 3 * BN * 64, BUF = A_BYTES + B_BYTES;
  __shared__ __align__(16) unsigned char smem[2 * BUF];

  const int t = threadIdx.x, lane = t & 63, wave8 = t >> 6;
  const bool stager = wave8 < NSW;
  // item list: workgroups on one XCD (workgroup index mod 8) take a contiguous eighth of the list, so that items which
  // share an operand panel share an L2 (speed only)
  int it, it_end, it_step;
  {
    const int gsz = (int)gridDim.x, w = (int)blockIdx.x;
    if (((gsz | nitems) & 7) == 0) { const int chunk = nitems >> 3, x = w & 7; it = x * chunk + (w >> 3); it_end = (x + 1) * chunk; it_step = gsz >> 3; }
    else { it = w; it_end = nitems; it_step = gsz; }
  }
  auto decode = [&](int item) __attribute__((always_inline)) {
    LimbItem r;
    const int per = tiles_m * tiles_n;
    r.split = item / per;
    const int rem = item - r.split * per;
    const int by = rem / tiles_n;
    r.bx = rem - by * tiles_n;
    r.m0 = by * BM; r.n0 = r.bx * BN;
    r.kbeg = r.split * g.klen;
    r.kend = min(g.K, r.kbeg + g.klen);
    return r;
  };
  if (it >= it_end) return;

  if (stager) {
    // ------------------------------------------------ staging waves ------------------------------------------------
    __builtin_amdgcn_s_setprio(3);
    LimbLoader<BM, T, RING, A_KMAJ> la;
    LimbLoader<BN, T, RING, B_KMAJ> lb;
    // total K steps of this workgroup's item list: both roles pass exactly one barrier per step (plus the initial one)
    int S = 0;
    for (int i = it; i < it_end; i += it_step) { const LimbItem q = decode(i); S += (q.kend - q.kbeg + LBK - 1) / LBK; }
    // load cursor: the next step to request;  store cursor (RING - 1 steps behind at most): the next step to split
    int l_it = it, s_it = it;
    LimbItem li = decode(l_it), si = li;
    int l_k = li.kbeg, s_k = si.kbeg;
    // The load-side state of the loaders (row offsets, bounds) is set right before the first load of an item; what a
    // later store needs (the in-range masks) is captured per ring slot at load time.
    auto load_step = [&](auto slot) __attribute__((always_inline)) {
      constexpr int SLOT = decltype(slot)::value;
      if (l_k == li.kbeg) { la.init(li.m0, g.M, g.lda); lb.init(li.n0, g.N, g.ldb); }
      if (!AVA_DBG_BIT(g, 4)) {
        la.template load<SLOT>(g.A, g.lda, l_k, li.kend, g.K);
        lb.template load<SLOT>(g.B, g.ldb, l_k, li.kend, g.K);
      }
      l_k += LBK;
      if (l_k >= li.kend) {
        l_it += it_step;
        if (l_it < it_end) { li = decode(l_it); l_k = li.kbeg; }
      }
    };
    auto store_step = [&](auto slot, int buf) __attribute__((always_inline)) {
      constexpr int SLOT = decltype(slot)::value;
      unsigned char* S_ = smem + buf * BUF;
      if (!AVA_DBG_BIT(g, 2)) {
        la.template store<SLOT, !A_KMAJ>(S_);
        lb.template store<SLOT, false>(S_ + A_BYTES);
      }
      s_k += LBK;
      if (s_k >= si.kend) {             // last step of an item: its column sums (bias gradient) are complete
        if constexpr (!A_KMAJ) {
          if (g.colsum != nullptr && si.bx == 0) {
            const bool fin = g.splits == 1;
#pragma unroll
            for (int i = 0; i < LimbLoaderN<BM, T, RING>::NU; ++i) {
              constexpr int KQA = LimbLoaderN<BM, T, RING>::KQ;      // threads (consecutive lanes) that own the same 4 columns
              const int u = t + T * i, cq = u / KQA;
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                float v = la.cs[i][c];     // the KQA threads that own the same 4 columns, fixed order
#pragma unroll
                for (int o = 1; o < KQA; o <<= 1) v += __shfl_xor(v, o, 64);
                const int gm = si.m0 + 4 * cq + c;
                if ((u % KQA) == 0 && u < LimbLoaderN<BM, T, RING>::UNITS && gm < g.M) {
                  if (fin) g.colsum[gm] = v;
                  else g.C[(size_t)g.splits * g.M * g.N + (size_t)si.split * g.M + gm] = v;   // partial, behind the slabs
                }
              }
            }
          }
#pragma unroll
          for (int i = 0; i < LimbLoaderN<BM, T, RING>::NU; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) la.cs[i][c] = 0.f;
        }
        s_it += it_step;
        if (s_it < it_end) { si = decode(s_it); s_k = si.kbeg; }
      }
    };
    // step j lives in ring slot j % RING.  Prologue: steps 0 .. RING-2 requested, step 0 split into buffer 0.
    limb_for_stage<0, RING - 1>([&](auto J) __attribute__((always_inline)) { if (decltype(J)::value < S) load_step(J); });
    store_step(std::integral_constant<int, 0>{}, 0);
    __syncthreads();                      // (A) step 0 ready
    // stage s (the matrix-core waves multiply step s): request step s + RING - 1 into the slot step s has left, split
    // step s + 1 into the other LDS buffer.
    // Steady state WITHOUT conditionals around the loads: hipcc's waitcnt pass merges the two sides of an `if (more) load`
    // conservatively (it must assume the loads were not issued), which turns the store's wait into vmcnt(0) -- i.e. into
    // a wait for the loads issued a moment ago, and the ring into a one-deep pipeline.
    int s0 = 0;
    for (; s0 + 2 * RING - 2 < S; s0 += RING)
      limb_for_stage<0, RING>([&](auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        load_step(std::integral_constant<int, (j + RING - 1) % RING>{});
        store_step(std::integral_constant<int, (j + 1) % RING>{}, (s0 + j + 1) & 1);
        __syncthreads();                  // (B) step consumed, next ready
      });
    for (; s0 < S; s0 += RING)            // the last (up to 2 RING - 2) steps
      limb_for_stage<0, RING>([&](auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        if (s0 + j < S) {
          if (s0 + j + RING - 1 < S) load_step(std::integral_constant<int, (j + RING - 1) % RING>{});
          if (s0 + j + 1 < S) store_step(std::integral_constant<int, (j + 1) % RING>{}, (s0 + j + 1) & 1);
          __syncthreads();                // (B)
        }
      });
    return;
  }

  // -------------------------------------------------- matrix-core waves --------------------------------------------------
  const int wave = wave8 - NSW, wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, kg = lane >> 4;
  const int a_lane = (wm * WM) * 64 + pos16<!A_KMAJ>(fr) * 64 + chunk_of<!A_KMAJ>(fr, kg) * 16;
  const int b_lane = A_BYTES + (wn * WN) * 64 + pos16<!B_KMAJ>(fr) * 64 + chunk_of<!B_KMAJ>(fr, kg) * 16;
  f32x4 acc[TM][TN];
  __syncthreads();                       // (A)
  int s = 0;
  for (; it < it_end; it += it_step) {
    const LimbItem ci = decode(it);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = ci.kbeg; k0 < ci.kend; k0 += LBK, ++s) {
      const unsigned char* As = smem + (s & 1) * BUF + a_lane;
      const unsigned char* Bs = smem + (s & 1) * BUF + b_lane;
      bf16x8 af[TM][3];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int l = 0; l < 3; ++l) af[i][l] = *reinterpret_cast<const bf16x8*>(As + l * BM * 64 + i * 1024);
      if (!AVA_DBG_BIT(g, 1))
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bf16x8 bf[3];
#pragma unroll
        for (int l = 0; l < 3; ++l) bf[l] = *reinterpret_cast<const bf16x8*>(Bs + l * BN * 64 + j * 1024);
        // D' = B^T A^T (rows = n, columns = m); smallest terms first; consecutive MFMAs go to different accumulators
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0], af[i][2], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[2], af[i][0], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[1], af[i][1], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0], af[i][1], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[1], af[i][0], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0], af[i][0], acc[i][j], 0, 0, 0);
      }
      __syncthreads();                   // (B)
    }
    // ---- epilogue, straight from the accumulators: lane (fr, kg) holds C[m0 + .. + fr][n0 + .. + 4 kg + 0..3] ----
    const bool fin = g.splits == 1;
    const int ldo = fin ? g.ldc : g.N;
    float* __restrict__ obase = fin ? g.C : g.C + (size_t)ci.split * g.M * g.N;
    // dW products and split-K slabs: nothing to add or mask -- keep the per-element activation switch out of the store loop
    const bool plain = !fin || (g.bias == nullptr && g.mask == nullptr && g.act == ACT_NONE);
    const bool relu = g.act == ACT_RELU;         // (ava_gemm_limb_ok admits ACT_NONE and ACT_RELU only)
    if (!AVA_DBG_BIT(g, 8) && plain) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int gm = ci.m0 + wm * WM + i * 16 + fr;
        if (gm >= g.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int gn = ci.n0 + wn * WN + j * 16 + 4 * kg;
          if (gn >= g.N) continue;         // N % 4 == 0: a quad is inside or outside as a whole
          *reinterpret_cast<f32x4*>(obase + (size_t)gm * ldo + gn) = acc[i][j];
        }
      }
    } else if (!AVA_DBG_BIT(g, 8)) {
      // row tiles outermost: consecutive store instructions of a wave then write neighbouring 64-byte pieces of the SAME
      // 16 output rows (4 kg quads x 16 bytes per row and instruction), which the L2 merges into full lines
      float4 bq[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int gn = ci.n0 + wn * WN + j * 16 + 4 * kg;
        bq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias != nullptr && gn < g.N) bq[j] = *reinterpret_cast<const float4*>(g.bias + gn);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int gm = ci.m0 + wm * WM + i * 16 + fr;
        if (gm >= g.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int gn = ci.n0 + wn * WN + j * 16 + 4 * kg;
          if (gn >= g.N) continue;
          float cv[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
          cv[0] += bq[j].x; cv[1] += bq[j].y; cv[2] += bq[j].z; cv[3] += bq[j].w;
          if (relu) { cv[0] = fmaxf(cv[0], 0.f); cv[1] = fmaxf(cv[1], 0.f); cv[2] = fmaxf(cv[2], 0.f); cv[3] = fmaxf(cv[3], 0.f); }
          if (g.mask != nullptr) {
            const float4 mk = *reinterpret_cast<const float4*>(g.mask + (size_t)gm * g.ldc + gn);
            if (!(mk.x > 0.f)) cv[0] = 0.f;
            if (!(mk.y > 0.f)) cv[1] = 0.f;
            if (!(mk.z > 0.f)) cv[2] = 0.f;
            if (!(mk.w > 0.f)) cv[3] = 0.f;
          }
          *reinterpret_cast<float4*>(obase + (size_t)gm * ldo + gn) = make_float4(cv[0], cv[1], cv[2], cv[3]);
        }
      }
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------
// Shapes the limb kernel takes: the products whose two long sides are those of fc1 / fc8 (1024 x F); the third is the
// batch, of any size.  The rule looks at the shape only, so a layer runs the same arithmetic at every batch size.
static bool limb_shape(int M, int N, int K) {
  const int mn = M < N ? M : N, mnk = mn < K ? mn : K;
  const size_t pair = (size_t)M * N * K / (size_t)mnk;
  return pair >= ((size_t)1 << 22);
}

bool ava_gemm_limb_ok(const GemmArgs& g, int a_kmajor, int b_kmajor) {
  static const bool on = [] { const char* e = ava_env("AVA_GEMM_LIMB"); return e == nullptr || atoi(e) != 0; }();
  if (!on || !limb_shape(g.M, g.N, g.K) || !g.vec_a || !g.vec_b) return false;
  if ((a_kmajor || b_kmajor) && g.K % 8 != 0) return false;
  if (!a_kmajor && g.M % 4 != 0) return false;
  if (!b_kmajor && g.N % 4 != 0) return false;
  if (g.colsum != nullptr && a_kmajor) return false;
  if (g.act != ACT_NONE && g.act != ACT_RELU) return false;
  // the epilogue stores (and reads bias / mask) as 16-byte quads of one output row
  if (g.N % 4 != 0 || g.ldc % 4 != 0 || (reinterpret_cast<uintptr_t>(g.C) & 15) != 0) return false;
  if (g.bias != nullptr && (reinterpret_cast<uintptr_t>(g.bias) & 15) != 0) return false;
  if (g.mask != nullptr && (reinterpret_cast<uintptr_t>(g.mask) & 15) != 0) return false;
  return true;
}

void ava_gemm_limb_plan(int M, int N, int K, int a_kmajor, int* bn, int* splits, int* klen) {
  // Forward / dX products (A = the batch-sized activation, K contiguous): 128 x 64 tiles (74 KB of LDS, <= 128 VGPRs), two
  // 512-thread workgroups per CU, so 512 work items fill the chip once.  dW products (both operands K-strided: their
  // in-register transposition needs more registers than that occupancy leaves): 128 x 128 tiles, one workgroup per CU.
  // K is split until the chip is full (fp32 slabs + fixed-order reduce kernel), at least two K steps per item.
  int b = a_kmajor ? 64 : 128;
  { const char* e = ava_env("AVA_GEMM_LIMB_BN"); if (e && a_kmajor) b = atoi(e) == 128 ? 128 : 64; }
  const int tiles = ceil_div(M, 128) * ceil_div(N, b);
  const int want = b == 64 ? 512 : 256;
  int s = 1;
  if (tiles < want * 3 / 4) {
    s = ceil_div(want, tiles);
    const int max_s = K / 64 > 0 ? K / 64 : 1;
    if (s > max_s) s = max_s;
  }
  { const char* e = ava_env("AVA_GEMM_LIMB_SPLITS"); if (e) { s = atoi(e); if (s < 1) s = 1; if (s > ceil_div(K, LBK)) s = ceil_div(K, LBK); } }
  int kl = ceil_div(ceil_div(K, s), LBK) * LBK;
  s = ceil_div(K, kl);
  *bn = b; *splits = s; *klen = kl;
}

template <int BN, int RING, int MINW, int NSW = 4>
static void launch_limb(const GemmArgs& g, int a_k, int b_k, int tm, int tn, int nitems, int grid, hipStream_t st) {
  const dim3 blk(64 * (NSW + 4));
  if (a_k && b_k) hipLaunchKernelGGL((gemm_limb_kernel<BN, true, true, RING, MINW, NSW>), dim3(grid), blk, 0, st, g, tm, tn, nitems);
  else if (a_k && !b_k) hipLaunchKernelGGL((gemm_limb_kernel<BN, true, false, RING, MINW, NSW>), dim3(grid), blk, 0, st, g, tm, tn, nitems);
  else if (!a_k && b_k) hipLaunchKernelGGL((gemm_limb_kernel<BN, false, true, RING, MINW, NSW>), dim3(grid), blk, 0, st, g, tm, tn, nitems);
  else hipLaunchKernelGGL((gemm_limb_kernel<BN, false, false, RING, MINW, NSW>), dim3(grid), blk, 0, st, g, tm, tn, nitems);
}

int ava_gemm_limb_launch(const GemmArgs& g0, int a_kmajor, int b_kmajor, int bn, hipStream_t st) {
  GemmArgs g = g0;
  { const char* e = ava_env("AVA_GEMM_LIMB_DBG"); g.dbg = e ? atoi(e) : 0; }
  const int tm = ceil_div(g.M, 128), tn = ceil_div(g.N, bn), nitems = tm * tn * g.splits;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  const int resident = ava_scale_grid((bn == 64 ? 2 : 1) * cus);     // persistent workgroups: one resident wave of them
  int grid = nitems < resident ? nitems : resident;
  { const char* e = ava_env("AVA_GEMM_LIMB_GRID"); if (e) { grid = atoi(e); if (grid > nitems) grid = nitems; if (grid < 1) grid = 1; } }
#ifdef AVA_LAB
  { static const int nsw = [] { const char* e = ava_env("AVA_GEMM_LIMB_NSW"); return e ? atoi(e) : 4; }();
    if (nsw == 8 && bn == 128) { launch_limb<128, 4, 3, 8>(g, a_kmajor, b_kmajor, tm, tn, nitems, grid, st); AVA_CHECK_LAUNCH(); return AVA_OK; } }
#endif
  if (bn == 64 && a_kmajor) {
    if (b_kmajor) hipLaunchKernelGGL((gemm_limb_kernel<64, true, true, 3, 4>), dim3(grid), dim3(512), 0, st, g, tm, tn, nitems);
    else hipLaunchKernelGGL((gemm_limb_kernel<64, true, false, 3, 4>), dim3(grid), dim3(512), 0, st, g, tm, tn, nitems);
  } else if (bn == 128 && !a_kmajor) {
    if (b_kmajor) hipLaunchKernelGGL((gemm_limb_kernel<128, false, true, 4, 2>), dim3(grid), dim3(512), 0, st, g, tm, tn, nitems);
    else hipLaunchKernelGGL((gemm_limb_kernel<128, false, false, 4, 2>), dim3(grid), dim3(512), 0, st, g, tm, tn, nitems);
  }
#ifdef AVA_LAB
  else if (bn == 128) launch_limb<128, 4, 2>(g, a_kmajor, b_kmajor, tm, tn, nitems, grid, st);
#endif
  else return AVA_EINVAL;
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}
