// Kernel templates of the "thin" convolutions (conv_thin.hip has the description and the launchers of the model's path,
// conv_thin_perop.hip the launchers of the forms only the per-operation entry points ava_conv3x3 / ava_conv3x3_wgrad reach).
#pragma once
#include <type_traits>
#include "conv_common.h"
#include "conv_fused.h"

// Every kernel is a template on the image width W (128 or 256: BASELINE configs 1-4 resp. 5): a workgroup of 2*W
// threads owns a tile of 8 full-width rows -- lane pair (2x, 2x+1) shares pixel column x -- so no kernel needs a
// horizontal halo from another workgroup.
#define THIN_TH 8           // tile: 8 rows x W columns
#define THIN_IR (THIN_TH + 2)
#define THIN_IC (W + 2)
#define THIN_NT (2 * W)      // threads per workgroup
#define THIN_NW (2 * W / 64) // waves per workgroup

// sum over the workgroup's waves in a fixed order: f(w) = wave w's value.  ((0+1)+(2+3)) for 4 waves (W = 128, the
// order the 128-wide kernels have always used), the same tree one level deeper for 8 (W = 256).
template <int NW, typename F>
__device__ __forceinline__ float thin_sum_waves(F f) {
  if constexpr (NW == 4) return (f(0) + f(1)) + (f(2) + f(3));
  else return ((f(0) + f(1)) + (f(2) + f(3))) + ((f(4) + f(5)) + (f(6) + f(7)));
}
#ifndef THIN_PLANES
#define THIN_PLANES 0     // 1: two [10][130][4] channel planes instead of [10][130][8]
#endif

// sum N per-thread values over the workgroup: wave shuffles, then one LDS exchange (lds: [NW][N] floats);
// result i is written to out[i] by thread i.  Fixed order -> deterministic.
template <int N, int NW>
__device__ __forceinline__ void thin_block_reduce(const float (&v)[N], float* lds, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float r = wave_sum(v[i]);
    if (lane == 0) lds[wave * N + i] = r;
  }
  __syncthreads();
  if (threadIdx.x < N && out != nullptr)
    out[threadIdx.x] = thin_sum_waves<NW>([&](int w) { return lds[w * N + threadIdx.x]; });
}

// partial rows of workgroups that were not launched (grid < part_rows) are zero-filled, N floats per row
template <int N>
__device__ __forceinline__ void thin_zero_rows(float* __restrict__ partials, int part_rows) {
  if (partials == nullptr || threadIdx.x >= N) return;
  for (int r = gridDim.x + blockIdx.x; r < part_rows; r += gridDim.x) partials[(size_t)r * N + threadIdx.x] = 0.f;
}

// The 72 wave-uniform weights of a thin layer, fetched ONCE into scalar registers.  Left as G[...] reads inside
// the tile loop they cannot be hoisted (hipcc must assume the output stores alias them) and turn into 18
// vector loads per tile and thread plus 72 VGPRs.
__device__ __forceinline__ float ava_uniform(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
struct ThinWeights {
  float w[72];
  __device__ __forceinline__ explicit ThinWeights(const float* __restrict__ G) {
#pragma unroll
    for (int i = 0; i < 72; ++i) w[i] = ava_uniform(G[i]);
  }
};

// Packed-FMA helpers: two fp32 lanes per instruction (v_pk_fma_f32).  ThinPairWeights keeps the 72 weights of an
// 8 -> 1 layer as 36 channel pairs in VECTOR registers (every lane the same value): as scalar-register pairs they do
// not fit beside the kernel's other scalars and hipcc spills them to VGPR lanes (125 v_readlane + 89 v_writelane per tile
// in the first version); the empty asm keeps hipcc from re-reading them from memory inside the tile loop.
typedef float avaf2 __attribute__((ext_vector_type(2)));
// A scalar that is BROADCAST into both halves of a packed operand must not reach the instruction as "src1, high register for both
// lanes": `v_pk_fma_f32 vD, vA, vB, vC op_sel:[0,1,0]` (and v_pk_mul_f32 ... op_sel:[0,1]) -- what hipcc emits when the scalar is the
// second dword of a ds_read2_b32 or element 1 / 3 of a 16-byte load and it sits in the src1 slot.  On gfx950 that form returns a
// WRONG low half in lanes 48..63 while a wave of another kernel that issues v_mfma_f32_16x16x32_bf16 shares the SIMD -- a second
// stream or a second process; never within one stream.  The same select on src0 or src2, every op_sel_hi form and the swapped-halves
// form are executed correctly (tools/lab/op_sel_forms.hip: an 80-line probe; tools/lab/two_proc_repro.hip; profiles/NOTES.md
// item 44).  Passing the scalar through an empty asm makes it a register of its own, and the broadcast then reads the LOW register
// of an aligned pair (op_sel_hi:[1,0,1]) -- at the price of the odd partner register and a move (convt7's forward: + 1.9 us).  The
// two kernels in which hipcc produced the bad form (convt7's forward + fold and its weight-gradient kernel: the seed of the thread's
// own column, second dword of a ds_read2_b32) now pass the broadcast as the FIRST multiplicand instead: hipcc keeps the order, the
// select lands on src0, no register and no instruction is spent, results are bit-identical (a * b is commutative).  The compiler is
// free to change its mind, so tools/lab/op_sel_scan.py scans the BUILT code objects for the bad form and tests/test_cpu_boundary.py
// runs the scan; AVA_PIN_MASK (one bit per site below) is the fallback that does not depend on operand order.
#ifndef AVA_PIN_MASK
#define AVA_PIN_MASK 0
#endif
template <int SITE>
__device__ __forceinline__ float ava_pin(float v) {
  if constexpr ((AVA_PIN_MASK >> SITE) & 1) asm volatile("" : "+v"(v));
  return v;
}
struct ThinPairWeights {
  avaf2 w[9][4];                                        // [tap][channel pair]
  __device__ __forceinline__ explicit ThinPairWeights(const float* __restrict__ G) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        avaf2 v = {G[tap * 8 + 2 * q], G[tap * 8 + 2 * q + 1]};
        asm volatile("" : "+v"(v));
        w[tap][q] = v;
      }
  }
};

// stage a 1-channel [10 x 130] window (origin row gy0, column -1) with the prologue applied
template <int W, int PRO>
__device__ __forceinline__ void thin_stage1(float* __restrict__ lds, const float* __restrict__ in,
                                            const float* __restrict__ in2, float ca, float cb, float cc, int b,
                                            int H, int gy0) {
  for (int v = threadIdx.x; v < THIN_IR * THIN_IC; v += THIN_NT) {
    const int r = v / THIN_IC, c = v - r * THIN_IC;
    const int gy = gy0 + r, gx = c - 1;
    float o = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      const size_t off = ((size_t)b * H + gy) * W + gx;
      o = prologue<PRO>(in[off], PRO == PRO_BWD ? in2[off] : 0.f, ca, cb, cc);
    }
    lds[v] = o;
  }
}

// The same window, register staged: `load` issues the global loads of a tile's window (clamped addresses, in-image bits
// remembered), `store` applies the prologue and writes LDS.  The kernels below load tile k+1's window while tile k is
// being multiplied, so the window's HBM latency is no longer paid between the two barriers of every tile (the direct
// thin_stage1 waits for its loads right where it issues them).
template <int W, int PRO>
struct ThinWindow {
  static constexpr int NV = (THIN_IR * THIN_IC + THIN_NT - 1) / THIN_NT;
  float v[NV], v2[PRO == PRO_BWD ? NV : 1];
  unsigned inb;
  __device__ __forceinline__ void load(const float* __restrict__ in, const float* __restrict__ in2, int b, int H, int gy0) {
    inb = 0u;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = threadIdx.x + THIN_NT * i, r = e / THIN_IC, c = e - r * THIN_IC;
      const int gy = gy0 + r, gx = c - 1;
      const bool ok = e < THIN_IR * THIN_IC && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const size_t off = ((size_t)b * H + min(max(gy, 0), H - 1)) * W + min(max(gx, 0), W - 1);
      v[i] = in[off];
      if (PRO == PRO_BWD) v2[i] = in2[off];
      inb |= ok ? (1u << i) : 0u;
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ lds, float ca, float cb, float cc) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = threadIdx.x + THIN_NT * i;
      const float o = ((inb >> i) & 1u) ? prologue<PRO>(v[i], PRO == PRO_BWD ? v2[i] : 0.f, ca, cb, cc) : 0.f;
      if (e < THIN_IR * THIN_IC) lds[e] = o;
    }
  }
};

// ---------------------------------------------------------------------------------------------------------
// 1 -> 8 channels: conv1 forward (PRO_BN, EPI_FWD), convt7 backward-data (PRO_ID, EPI_BWD)
// ---------------------------------------------------------------------------------------------------------
// Thread mapping: lane pair (2x, 2x+1) shares pixel column x of the 8-row tile; thread (x, h) owns output channels
// 4h..4h+3 of all 8 rows.  Every store instruction of a wave is then 64 contiguous 16-byte slots (1 KB, full
// lines); with one thread per pixel and two 16-byte stores per pixel each instruction wrote every other 16-byte
// slot and the layer ran at 3.8 TB/s of a possible ~5.  Weights depend on h, so they live in vector registers
// (36 per thread) as channel pairs for v_pk_fma_f32.
// ACT: storage type of the 8-channel ACTIVATION this launch touches (EPI_FWD: the output; EPI_BWD: epi_x); the
// 8-channel output of the data-gradient forms (EPI_BWD / EPI_NONE) is an fp32 gradient.
template <int W, int PRO, int EPI, typename ACT = float>
__global__ __launch_bounds__(2 * W, W == 128 ? 4 : 2) void thin_1to8_kernel(const ConvArgs a) {
  using TOUT = typename std::conditional<EPI == EPI_FWD, ACT, float>::type;
  __shared__ float tile[THIN_IR * THIN_IC];
  __shared__ float red[THIN_NW][2][8];
  const int t = threadIdx.x, h = t & 1, x = t >> 1, lane = t & 63, wave = t >> 6;
  float ca = a.pa ? a.pa[0] : 0.f, cb = a.pb ? a.pb[0] : 0.f;
  const float cc = a.pc ? a.pc[0] : 0.f;
  if (PRO == PRO_BN && a.fin.acc != nullptr) {        // bn1 (conv1's forward): the input sums of the pack launch, finalised here
    __shared__ float coef[96];
    __shared__ double accvals[64];
    bn_coef_from_acc(coef, accvals, a.fin, 0);
    ca = coef[0];
    cb = coef[32];
  }
  avaf2 w2[9][2];                                    // [tap][channel pair of this half]
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      avaf2 v = {a.G[tap * 8 + 4 * h + 2 * q], a.G[tap * 8 + 4 * h + 2 * q + 1]};
      asm volatile("" : "+v"(v));                    // read once: stores below may alias G as far as hipcc knows
      w2[tap][q] = v;
    }
  avaf2 bias2[2], s1[2], s2[2];
  float emean[4], einv[4];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    bias2[q] = EPI == EPI_FWD ? avaf2{a.bias[4 * h + 2 * q], a.bias[4 * h + 2 * q + 1]} : avaf2{0.f, 0.f};
    s1[q] = s2[q] = avaf2{0.f, 0.f};
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    emean[c] = EPI == EPI_BWD ? a.epi_mean[4 * h + c] : 0.f;
    einv[c] = EPI == EPI_BWD ? a.epi_invstd[4 * h + c] : 0.f;
  }
  const bool relu = a.relu != 0;
  const int tiles_y = a.Ho / THIN_TH;
  // sweeping walk: store-heavy, the only shared input is a 1-channel halo row (in-step A/B: conv1 forward 32.2 -> 29.3 us,
  // convt7 data gradient 35.4 -> 33.1 us against the per-XCD chunked walk)
  ThinWindow<W, PRO> win;
  TileWalk walk(a.ntiles, false);
  if (walk.valid()) { const int tl = walk.cur, b0 = tl / tiles_y; win.load(a.in, a.in2, b0, a.Hi, (tl - b0 * tiles_y) * THIN_TH - 1); }
  for (; walk.valid(); walk.advance()) {
    const int tl = walk.cur;
    const int b = tl / tiles_y, oy0 = (tl - b * tiles_y) * THIN_TH;
    __syncthreads();
    win.store(tile, ca, cb, cc);
    __syncthreads();
    if (walk.has_next()) { const int tn = walk.next(), bn = tn / tiles_y; win.load(a.in, a.in2, bn, a.Hi, (tn - bn * tiles_y) * THIN_TH - 1); }
    avaf2 acc[THIN_TH][2];
#pragma unroll
    for (int p = 0; p < THIN_TH; ++p) acc[p][0] = acc[p][1] = avaf2{0.f, 0.f};
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float in[THIN_IR];
#pragma unroll
      for (int j = 0; j < THIN_IR; ++j) in[j] = ava_pin<0>(tile[j * THIN_IC + x + kx]);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int p = 0; p < THIN_TH; ++p) {
          const avaf2 iv = {in[p + ky], in[p + ky]};
#pragma unroll
          for (int q = 0; q < 2; ++q) acc[p][q] = __builtin_elementwise_fma(iv, w2[ky * 3 + kx][q], acc[p][q]);
        }
    }
    const size_t o0 = (((size_t)b * a.Ho + oy0) * W + x) * 8 + 4 * h;
#pragma unroll
    for (int p = 0; p < THIN_TH; ++p) {
      const size_t off = o0 + (size_t)p * W * 8;
      avaf2 v0 = acc[p][0], v1 = acc[p][1];
      if (EPI == EPI_FWD) {
        v0 += bias2[0]; v1 += bias2[1];
        if (relu) { v0 = avaf2{fmaxf(v0[0], 0.f), fmaxf(v0[1], 0.f)}; v1 = avaf2{fmaxf(v1[0], 0.f), fmaxf(v1[1], 0.f)}; }
        v0 = avaf2{ava_stored<TOUT>(v0[0]), ava_stored<TOUT>(v0[1])};          // statistics of what is stored
        v1 = avaf2{ava_stored<TOUT>(v1[0]), ava_stored<TOUT>(v1[1])};
        s1[0] += v0; s1[1] += v1;
        s2[0] = __builtin_elementwise_fma(v0, v0, s2[0]);
        s2[1] = __builtin_elementwise_fma(v1, v1, s2[1]);
      } else if (EPI == EPI_BWD) {
        const avaf4 xr = ava_ld4<ACT>(ava_as<ACT>(a.epi_x) + off);
        const avaf2 xh0 = {(xr[0] - emean[0]) * einv[0], (xr[1] - emean[1]) * einv[1]};
        const avaf2 xh1 = {(xr[2] - emean[2]) * einv[2], (xr[3] - emean[3]) * einv[3]};
        s1[0] += v0; s1[1] += v1;
        s2[0] = __builtin_elementwise_fma(v0, xh0, s2[0]);
        s2[1] = __builtin_elementwise_fma(v1, xh1, s2[1]);
      }
      if (a.out != nullptr) ava_st4_wt<TOUT>(ava_as<TOUT>(a.out) + off, avaf4{v0[0], v0[1], v1[0], v1[1]});
    }
  }
  if (EPI == EPI_NONE) return;
  // ---- per-channel sums: lanes of equal parity hold the same 4 channels; waves, then workgroup, fixed order ----
  float sv[8] = {s1[0][0], s1[0][1], s1[1][0], s1[1][1], s2[0][0], s2[0][1], s2[1][0], s2[1][1]};
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float v = sv[i];
#pragma unroll
    for (int o = 32; o > 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane < 2) red[wave][lane][i] = v;              // lane = h
  }
  __syncthreads();
  if (t < 16 && a.acc_out != nullptr) {                    // sums accumulated for the consumer's prologue (bn_acc.h)
    const int which = t >> 3, co = t & 7, hh = co >> 2, i = which * 4 + (co & 3);
    bn_acc_add(a.acc_out, which * 32 + co, thin_sum_waves<THIN_NW>([&](int w) { return red[w][hh][i]; }));
    return;
  }
  if (a.acc_out != nullptr) return;
  if (t < 16 && a.partials != nullptr) {
    const int which = t >> 3, co = t & 7, hh = co >> 2, i = which * 4 + (co & 3);
    a.partials[(size_t)blockIdx.x * 16 + t] = thin_sum_waves<THIN_NW>([&](int w) { return red[w][hh][i]; });
  }
  thin_zero_rows<16>(a.partials, a.part_rows);
}

// ---------------------------------------------------------------------------------------------------------
// 8 -> 1 channels: convt7 forward (PRO_BN, EPI_SSE), conv1 backward-data (PRO_BWD / PRO_ID, EPI_BWD)
// ---------------------------------------------------------------------------------------------------------
template <int W, int PRO, int EPI>
__global__ __launch_bounds__(2 * W, W == 128 ? 2 : 1) void thin_8to1_kernel(const ConvArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* tile = smem;                                   // [10][130][8]
  float* coef = smem + THIN_IR * THIN_IC * 8;           // [3][32]
  float* red = coef + 96;                               // [4][2]
  const int t = threadIdx.x, ty0 = (t / W) * 4, x = t % W;
  if (t < 96) {
    const int which = t >> 5, c = t & 31;
    const float* src = which == 0 ? a.pa : (which == 1 ? a.pb : a.pc);
    coef[t] = (src != nullptr && c < 8) ? src[c] : 0.f;
  }
  const ThinWeights Wt(a.G);                            // [9][8][1]
  const float bias0 = EPI == EPI_SSE ? ava_uniform(a.bias[0]) : 0.f;
  const float em0 = EPI == EPI_BWD ? ava_uniform(a.epi_mean[0]) : 0.f, ei0 = EPI == EPI_BWD ? ava_uniform(a.epi_invstd[0]) : 0.f;
  float s1 = 0.f, s2 = 0.f;
  TileStager<8, PRO, THIN_IR, THIN_IC, (THIN_PLANES != 0), THIN_NT> stg;
  stg.init();
  const int tiles_y = a.Ho / THIN_TH;
  for (TileWalk walk(a.ntiles); walk.valid(); walk.advance()) {
    const int tl = walk.cur;
    const int b = tl / tiles_y, oy0 = (tl - b * tiles_y) * THIN_TH;
    __syncthreads();
    stg.load(a.in, a.in2, b, a.Hi, a.Wi, oy0 - 1, -1);
    stg.store(tile, coef);
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      // one tap column at a time: without the fence hipcc hoists all 36 LDS vectors of the tile (144 VGPRs)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float* px = tile + ((ty0 + j) * THIN_IC + x + kx) * (THIN_PLANES ? 4 : 8);
        const float4 u = *reinterpret_cast<const float4*>(px);
        const float4 w4 = *reinterpret_cast<const float4*>(px + (THIN_PLANES ? THIN_IR * THIN_IC * 4 : 4));
        const float in[8] = {u.x, u.y, u.z, u.w, w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int p = j - ky;                         // output row fed by input row j through tap ky
          if (p >= 0 && p < 4) {
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) acc[p] = fmaf(in[ci], Wt.w[(ky * 3 + kx) * 8 + ci], acc[p]);
          }
        }
      }
    }
    const size_t opix0 = ((size_t)b * a.Ho + oy0 + ty0) * W + x;
    if (EPI == EPI_SSE) {
      const float bias = bias0;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const size_t opix = opix0 + (size_t)p * W;
        const float v = acc[p] + bias;
        if (a.epi_x != nullptr) {
          const float r = v - a.epi_x[opix];
          a.out2[opix] = a.prec * r;
          s1 = fmaf(r, r, s1);
        }
        if (a.out != nullptr) a.out[opix] = v;
      }
    } else {
      const float m = em0, is = ei0;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const size_t opix = opix0 + (size_t)p * W;
        s1 += acc[p];
        s2 = fmaf(acc[p], (a.epi_x[opix] - m) * is, s2);
        if (a.out != nullptr) a.out[opix] = acc[p];
      }
    }
  }
  const float sv[2] = {s1, s2};
  thin_block_reduce<2, THIN_NW>(sv, red, a.partials != nullptr ? a.partials + (size_t)blockIdx.x * 2 : nullptr);
  thin_zero_rows<2>(a.partials, a.part_rows);
}

#ifdef AVA_LAB
#define W 128   /* lab-only kernels: 128-wide images only */
// Wave-specialised variant (512 threads, one workgroup per CU): waves 0-3 stage tile k+1 (global -> registers ->
// prologue -> LDS buffer (k+1)&1) while waves 4-7 multiply tile k out of buffer k&1; one barrier per tile.  At the
// barrier of tile k the staging waves have filled buffer k&1 and the compute waves have left buffer (k-1)&1, which
// is the one the staging waves write next.  The next window's loads are issued right after the LDS writes, so they
// are in flight for the whole period of a tile instead of being waited for back to back.
template <int PRO, int EPI, int NS>
__global__ __launch_bounds__(NS + 256) void thin_8to1_ws_kernel(const ConvArgs a) {
  extern __shared__ __align__(16) float smem[];
  constexpr int TILE_F = THIN_IR * THIN_IC * 8;
  float* coef = smem + 2 * TILE_F;                      // [3][32]
  float* red = coef + 96;                               // [4][2]
  const int t = threadIdx.x;
  if (t < 96) {
    const int which = t >> 5, c = t & 31;
    const float* src = which == 0 ? a.pa : (which == 1 ? a.pb : a.pc);
    coef[t] = (src != nullptr && c < 8) ? src[c] : 0.f;
  }
  const int tiles_y = a.Ho / THIN_TH;
  float s1 = 0.f, s2 = 0.f;
  if (t < NS) {
    // ---- staging waves ----
    TileStager<8, PRO, THIN_IR, THIN_IC, false, NS> stg;
    stg.init(t);
    TileWalk walk(a.ntiles);
    if (walk.valid()) {
      const int b = walk.cur / tiles_y, oy0 = (walk.cur - b * tiles_y) * THIN_TH;
      stg.load(a.in, a.in2, b, a.Hi, a.Wi, oy0 - 1, -1);
    }
    __syncthreads();                                    // coefficients visible
    for (int k = 0; walk.valid(); walk.advance(), k ^= 1) {
      stg.store(smem + k * TILE_F, coef);
      if (walk.has_next()) {
        const int tn = walk.next();
        const int b = tn / tiles_y, oy0 = (tn - b * tiles_y) * THIN_TH;
        stg.load(a.in, a.in2, b, a.Hi, a.Wi, oy0 - 1, -1);
      }
      __syncthreads();                                  // buffer k full
    }
  } else {
    // ---- compute waves ----
    const int tc = t - NS, ty0 = (tc >> 7) * 4, x = tc & 127;
    const ThinPairWeights Wp(a.G);                       // [9][8][1] as channel pairs
    const float bias0 = EPI == EPI_SSE ? ava_uniform(a.bias[0]) : 0.f;
    const float em0 = EPI == EPI_BWD ? ava_uniform(a.epi_mean[0]) : 0.f, ei0 = EPI == EPI_BWD ? ava_uniform(a.epi_invstd[0]) : 0.f;
    __syncthreads();
    int k = 0;
    for (TileWalk walk(a.ntiles); walk.valid(); walk.advance(), k ^= 1) {
      const int tl = walk.cur;
      const int b = tl / tiles_y, oy0 = (tl - b * tiles_y) * THIN_TH;
      const size_t opix0 = ((size_t)b * a.Ho + oy0 + ty0) * W + x;
      float ex[4] = {0.f, 0.f, 0.f, 0.f};               // epilogue operand, requested before the wait for the tile
      if (a.epi_x != nullptr) {
#pragma unroll
        for (int p = 0; p < 4; ++p) ex[p] = a.epi_x[opix0 + (size_t)p * W];
      }
      __syncthreads();                                  // buffer k full
      const float* tile = smem + k * TILE_F;
      avaf2 acc2[4];                                    // even / odd input channels of the 4 output pixels
#pragma unroll
      for (int p = 0; p < 4; ++p) acc2[p] = avaf2{0.f, 0.f};
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        __builtin_amdgcn_sched_barrier(0);              // one tap column at a time (register pressure)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const float* px = tile + ((ty0 + j) * THIN_IC + x + kx) * 8;
          const avaf4 u = *reinterpret_cast<const avaf4*>(px);
          const avaf4 v = *reinterpret_cast<const avaf4*>(px + 4);
          const avaf2 in2[4] = {avaf2{u[0], u[1]}, avaf2{u[2], u[3]}, avaf2{v[0], v[1]}, avaf2{v[2], v[3]}};
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int p = j - ky;                       // output row fed by input row j through tap ky
            if (p >= 0 && p < 4) {
#pragma unroll
              for (int q = 0; q < 4; ++q) acc2[p] = __builtin_elementwise_fma(in2[q], Wp.w[ky * 3 + kx][q], acc2[p]);
            }
          }
        }
      }
      float acc[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) acc[p] = acc2[p][0] + acc2[p][1];
      if (EPI == EPI_SSE) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const size_t opix = opix0 + (size_t)p * W;
          const float v = acc[p] + bias0;
          if (a.epi_x != nullptr) {
            const float r = v - ex[p];
            a.out2[opix] = a.prec * r;
            s1 = fmaf(r, r, s1);
          }
          if (a.out != nullptr) a.out[opix] = v;
        }
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const size_t opix = opix0 + (size_t)p * W;
          s1 += acc[p];
          s2 = fmaf(acc[p], (ex[p] - em0) * ei0, s2);
          if (a.out != nullptr) a.out[opix] = acc[p];
        }
      }
    }
  }
  // ---- workgroup sums (compute waves 4..7 hold them), fixed order ----
  __syncthreads();
  const int lane = t & 63, wave = t >> 6;
  const float r1 = wave_sum(s1), r2 = wave_sum(s2);
  constexpr int W0 = NS / 64;                            // first compute wave
  if (wave >= W0 && lane == 0) { red[(wave - W0) * 2] = r1; red[(wave - W0) * 2 + 1] = r2; }
  __syncthreads();
  if (t < 2 && a.partials != nullptr)
    a.partials[(size_t)blockIdx.x * 2 + t] = (red[t] + red[2 + t]) + (red[4 + t] + red[6 + t]);
  thin_zero_rows<2>(a.partials, a.part_rows);
}

#undef W
#endif  // AVA_LAB

// "Direct" form of the 8 -> 1 forward (convt7 + SSE epilogue): the 8-channel input needs no LDS window.  Thread (x, h)
// of a lane pair reads channels 4h..4h+3 of its pixel column straight from global memory (10 rows: the 8 rows of the
// tile and one halo row on each side), applies BatchNorm (zero outside the image) and reduces over its channels and
// over ky in registers:   u[r][kx] = sum_ky sum_c W[ky][kx][c] * x_n[r + ky - 1][c].
// Only these 24 partial sums per thread cross lanes, through a 25 KB LDS array with zero border columns:
//   y[r][x] = bias + sum_h sum_kx u_h[r][kx] at column x + kx - 1.
// No staging role, three workgroups per CU resident, ten independent 16-byte loads per thread in flight; the
// LDS-staged wave-specialised form kept one 41.6 KB window per workgroup in flight.
// FOLD (convt7's TRAINING forward, a.fold): the same launch also leaves behind what convt7's backward needs besides the data
// gradient -- the weight / bias gradient partials and the BatchNorm-backward sums of the layer input (reference: autograd's
// convolution backward behind loss.backward(), ava/models/vae.py:352).  Both are correlations of the input windows with the
// seed dU = prec * (xhat - x), and the thread that owns an input pixel column has its ten input rows in registers the moment
// the tile's seed exists; the seed takes one more trip through a 4 KB LDS tile (zero border columns):
//     dG'[ky][kx][c] += xhat0[r + ky][c] (own column x) * dU[r][x - kx + 1]          r = 0..7: the tile's own output rows
// (xhat0 = (v - mean) * invstd, zero outside the image), indexed by the OUTPUT pixel, so no seed row of another tile is needed.
// The rest is thin_wgrad_stats_8to1_direct_kernel's algebra (S[tap] from the nine border sums of dU; dG = gamma dG' + beta S;
// sum dx = sum_tap G S, sum dx xhat = sum_tap G dG').  That kernel -- a 34 us launch that re-read the 134 MB input and the seed
// -- is gone from the training step; 144 more packed FMAs per thread and tile in a kernel whose VALU was idle.  The partials
// are those of loss scale 1: a backward with another scale runs the separate kernel (model.hip).
template <int W, typename ACT, bool FOLD>
__device__ __forceinline__ void thin_8to1_direct_body(const ConvArgs& a) {
  __shared__ float U[2][3][THIN_TH][THIN_IC];           // [half][kx][row][column + 1]; columns 0 and 129 stay zero
  __shared__ float red[THIN_NW][2];
  __shared__ float coef[96];
  __shared__ double accvals[64];
  __shared__ float ems[64];                             // FOLD: mean [0..31], invstd [32..63] of the input's BatchNorm
  __shared__ float dUt[FOLD ? THIN_TH * THIN_IC : 1];   // FOLD: the tile's seed, [row][column + 1], zero border columns
  __shared__ float fred[FOLD ? THIN_NW : 1][2][36];     // FOLD: per wave, per channel half: dG' [9][4]
  __shared__ float fsc[FOLD ? THIN_NW : 1][9];          // per wave: T, Rt, Rb, Cl, Cr, Ktl, Ktr, Kbl, Kbr
  __shared__ float ftot[FOLD ? 2 : 1][36];
  __shared__ float fstot[9];
  __shared__ float fscratch[FOLD ? 2 : 1][72];
  const int t = threadIdx.x, h = t & 1, x = t >> 1, lane = t & 63, wave = t >> 6;
  if (t < 2 * 3 * THIN_TH) {                            // zero borders, once
    float* row = &U[0][0][0][0] + t * THIN_IC;
    row[0] = 0.f;
    row[THIN_IC - 1] = 0.f;
  }
  if constexpr (FOLD) {
    if (t < THIN_TH) { dUt[t * THIN_IC] = 0.f; dUt[t * THIN_IC + THIN_IC - 1] = 0.f; }
  }
  float ca[4], cb[4];
  if (a.fin.acc != nullptr) {               // BatchNorm of the input: sums accumulated by the producer (bn_acc.h), finalised here
    bn_coef_from_acc(coef, accvals, a.fin, 0, FOLD ? ems : nullptr);
#pragma unroll
    for (int c = 0; c < 4; ++c) { ca[c] = coef[4 * h + c]; cb[c] = coef[32 + 4 * h + c]; }
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) { ca[c] = a.pa[4 * h + c]; cb[c] = a.pb[4 * h + c]; }
    if (FOLD) {
      if (t < 64) { const int c = t & 31; ems[t] = c < 8 ? (t < 32 ? a.fold.mean[c] : a.fold.invstd[c]) : 0.f; }
      if (t < 16) coef[(t >> 3) * 32 + (t & 7)] = t < 8 ? a.pa[t] : a.pb[t & 7];
      __syncthreads();
    }
  }
  float ha[4], hb[4];                                   // FOLD: xhat = ha * v + hb for this half's channels
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float is = FOLD ? ems[32 + 4 * h + c] : 0.f;
    ha[c] = is;
    hb[c] = FOLD ? -ems[4 * h + c] * is : 0.f;
  }
  avaf2 w2[9][2];                                       // [tap][channel pair of this half], G is [9][8][1]
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      avaf2 v = {a.G[tap * 8 + 4 * h + 2 * q], a.G[tap * 8 + 4 * h + 2 * q + 1]};
      asm volatile("" : "+v"(v));
      w2[tap][q] = v;
    }
  const float bias0 = a.bias[0];
  const int xo = t % W, r0 = (t / W) * 4;            // phase 2: output pixels (r0 + p, xo)
  float s1 = 0.f;
  avaf2 facc[FOLD ? 9 : 1][2];
#pragma unroll
  for (int k = 0; k < (FOLD ? 9 : 1); ++k) facc[k][0] = facc[k][1] = avaf2{0.f, 0.f};
  float T = 0.f, Rt = 0.f, Rb = 0.f, Cl = 0.f, Cr = 0.f, Ktl = 0.f, Ktr = 0.f, Kbl = 0.f, Kbr = 0.f;
  const float own = h == 0 ? 1.f : 0.f;                 // the scalar sums of dU are taken by one lane of each pair
  const int tiles_y = a.Ho / THIN_TH;
  for (TileWalk walk(a.ntiles); walk.valid(); walk.advance()) {
    const int tl = walk.cur;
    const int b = tl / tiles_y, oy0 = (tl - b * tiles_y) * THIN_TH;
    // ---- phase 1: own pixel column, 10 rows ----
    // (FOLD, measured: requesting the NEXT tile's rows here, to travel under phases 2 and 3, costs 256 VGPRs + spills and makes
    // the kernel slower, 52.7 -> 60.6 us; the second resident workgroup already covers the loads)
    const ACT* __restrict__ xin = ava_as<ACT>(a.in) + ((size_t)b * a.Hi * W + x) * 8 + 4 * h;
    avaf2 xn[THIN_IR][2];
    avaf2 xh[FOLD ? THIN_IR : 1][2];
#pragma unroll
    for (int j = 0; j < THIN_IR; ++j) {
      const int gy = oy0 - 1 + j;
      const bool ok = gy >= 0 && gy < a.Hi;             // wave-uniform
      const avaf4 v = ava_ld4<ACT>(xin + (size_t)min(max(gy, 0), a.Hi - 1) * W * 8);
      xn[j][0] = ok ? avaf2{fmaf(ca[0], v[0], cb[0]), fmaf(ca[1], v[1], cb[1])} : avaf2{0.f, 0.f};
      xn[j][1] = ok ? avaf2{fmaf(ca[2], v[2], cb[2]), fmaf(ca[3], v[3], cb[3])} : avaf2{0.f, 0.f};
      if constexpr (FOLD) {
        xh[j][0] = ok ? avaf2{fmaf(ha[0], v[0], hb[0]), fmaf(ha[1], v[1], hb[1])} : avaf2{0.f, 0.f};
        xh[j][1] = ok ? avaf2{fmaf(ha[2], v[2], hb[2]), fmaf(ha[3], v[3], hb[3])} : avaf2{0.f, 0.f};
      }
    }
    const size_t opix0 = ((size_t)b * a.Ho + oy0 + r0) * W + xo;
    float ex[4] = {0.f, 0.f, 0.f, 0.f};                 // epilogue operand of this thread's output pixels
    if (a.epi_x != nullptr) {
#pragma unroll
      for (int p = 0; p < 4; ++p) ex[p] = a.epi_x[opix0 + (size_t)p * W];
    }
    float u[THIN_TH][3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int r = 0; r < THIN_TH; ++r) {
        avaf2 sacc = {0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int q = 0; q < 2; ++q) sacc = __builtin_elementwise_fma(xn[r + ky][q], w2[ky * 3 + kx][q], sacc);
        u[r][kx] = sacc[0] + sacc[1];
      }
    __syncthreads();                                    // the previous tile's phase 2 has read U (FOLD: its phase 3 has read dUt)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int r = 0; r < THIN_TH; ++r) U[h][kx][r][x + 1] = u[r][kx];
    __syncthreads();
    // ---- phase 2: 4 output pixels per thread ----
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float v = bias0;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) v += U[0][kx][r0 + p][xo + kx] + U[1][kx][r0 + p][xo + kx];
      const size_t opix = opix0 + (size_t)p * W;
      if (a.epi_x != nullptr) {
        const float r = v - ex[p];
        const float sd = a.prec * r;
        a.out2[opix] = sd;
        if constexpr (FOLD) dUt[(r0 + p) * THIN_IC + xo + 1] = sd;
        s1 = fmaf(r, r, s1);
      }
      if (a.out != nullptr) a.out[opix] = v;
    }
    if constexpr (FOLD) {
      // ---- phase 3: correlations of the own input column with the tile's seed ----
      __syncthreads();
      {
        float col = 0.f;
#pragma unroll
        for (int r = 0; r < THIN_TH; ++r) col += dUt[r * THIN_IC + x + 1];
        const float top = oy0 == 0 ? dUt[x + 1] : 0.f;
        const float bot = oy0 + THIN_TH == a.Ho ? dUt[(THIN_TH - 1) * THIN_IC + x + 1] : 0.f;
        T += own * col;
        Rt += own * top;
        Rb += own * bot;
        if (x == 0) { Cl += own * col; Ktl += own * top; Kbl += own * bot; }
        if (x == W - 1) { Cr += own * col; Ktr += own * top; Kbr += own * bot; }
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        float d[THIN_TH];
#pragma unroll
        for (int r = 0; r < THIN_TH; ++r) d[r] = ava_pin<1>(dUt[r * THIN_IC + x + 2 - kx]);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int r = 0; r < THIN_TH; ++r) {
            const avaf2 dv = {d[r], d[r]};
#pragma unroll
            for (int q = 0; q < 2; ++q) facc[ky * 3 + kx][q] = __builtin_elementwise_fma(dv, xh[r + ky][q], facc[ky * 3 + kx][q]);
          }
      }
    }
  }
  const float r1 = wave_sum(s1);
  if (lane == 0) { red[wave][0] = r1; red[wave][1] = 0.f; }
  if constexpr (FOLD) {
    // ---- workgroup totals: dG' per channel half over lanes of equal parity, the nine scalar sums over all lanes ----
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float v = facc[k][q][e];
#pragma unroll
          for (int o = 32; o > 1; o >>= 1) v += __shfl_xor(v, o, 64);
          if (lane < 2) fred[wave][lane][k * 4 + 2 * q + e] = v;
        }
    const float sv[9] = {T, Rt, Rb, Cl, Cr, Ktl, Ktr, Kbl, Kbr};
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const float r = wave_sum(sv[i]);
      if (lane == 0) fsc[wave][i] = r;
    }
  }
  __syncthreads();
  if (t < 2 && a.partials != nullptr)
    a.partials[(size_t)blockIdx.x * 2 + t] = thin_sum_waves<THIN_NW>([&](int w) { return red[w][t]; });
  thin_zero_rows<2>(a.partials, a.part_rows);
  if constexpr (FOLD) {
    if (t < 72) {
      const int hh = t / 36, i = t - 36 * hh;
      ftot[hh][i] = thin_sum_waves<THIN_NW>([&](int w) { return fred[w][hh][i]; });
    } else if (t < 81) {
      const int i = t - 72;
      fstot[i] = thin_sum_waves<THIN_NW>([&](int w) { return fsc[w][i]; });
    }
    __syncthreads();
    if (t < 72) {
      const int tap = t >> 3, ci = t & 7, hh = ci >> 2, c = ci & 3, ky = tap / 3, kx = tap - 3 * ky;
      float S = fstot[0];
      if (ky == 0) S -= fstot[1];
      if (ky == 2) S -= fstot[2];
      if (kx == 0) S -= fstot[3];
      if (kx == 2) S -= fstot[4];
      if (ky == 0 && kx == 0) S += fstot[5];
      if (ky == 0 && kx == 2) S += fstot[6];
      if (ky == 2 && kx == 0) S += fstot[7];
      if (ky == 2 && kx == 2) S += fstot[8];
      const float xa = coef[ci], xb = coef[32 + ci], mean = ems[ci], invstd = ems[32 + ci];
      const float gamma = xa / invstd, beta = fmaf(mean, xa, xb);
      const float dgp = ftot[hh][tap * 4 + c];
      a.fold.wg_partials[(size_t)blockIdx.x * 73 + t] = fmaf(gamma, dgp, beta * S);
      const float w = a.G[tap * 8 + ci];                // forward gather weight G[tap][ci]
      fscratch[0][t] = w * S;
      fscratch[1][t] = w * dgp;
    } else if (t == 72) {
      a.fold.wg_partials[(size_t)blockIdx.x * 73 + 72] = fstot[0];                    // bias gradient = T
    }
    __syncthreads();
    if (t < 16) {
      const int which = t >> 3, ci = t & 7;
      float s = 0.f;
      for (int tap = 0; tap < 9; ++tap) s += fscratch[which][tap * 8 + ci];
      if (a.fold.acc_out != nullptr) bn_acc_add(a.fold.acc_out, which * 32 + ci, s);     // accumulated for the consumer's prologue
      else a.fold.bn_partials[(size_t)blockIdx.x * 16 + t] = s;
    }
    thin_zero_rows<73>(a.fold.wg_partials, a.part_rows);
    if (a.fold.acc_out == nullptr) thin_zero_rows<16>(a.fold.bn_partials, a.part_rows);
  }
}

template <int W, int PRO, int EPI, typename ACT = float>
__global__ __launch_bounds__(2 * W) void thin_8to1_direct_kernel(const ConvArgs a) {
  static_assert(PRO == PRO_BN && EPI == EPI_SSE, "only convt7's forward uses this form");
  thin_8to1_direct_body<W, ACT, false>(a);
}
// the launch is two workgroups per CU at W = 128, one at W = 256 (conv_thin.hip): 256 VGPRs are there to use
template <int W, typename ACT = float>
__global__ __launch_bounds__(2 * W, 2) void thin_8to1_direct_fold_kernel(const ConvArgs a) {
  thin_8to1_direct_body<W, ACT, true>(a);
}

// ---------------------------------------------------------------------------------------------------------
// weight gradients.  dy-side prologue is applied on the fly to each thread's own 4 pixels (no LDS needed for dy)
// ---------------------------------------------------------------------------------------------------------
// conv1: CIN = 1, COUT = 8.  dG[9][8], db[8]
template <int W, int DYPRO>
__global__ __launch_bounds__(2 * W) void thin_wgrad_1to8_kernel(const WgradArgs a) {
  __shared__ float tile[THIN_IR * THIN_IC];
  __shared__ float red[THIN_NW * 80];
  const int t = threadIdx.x, ty0 = (t / W) * 4, x = t % W;
  const float xa = a.xa[0], xb = a.xb[0];
  float acc[9][8], bacc[8];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int co = 0; co < 8; ++co) acc[k][co] = 0.f;
#pragma unroll
  for (int co = 0; co < 8; ++co) bacc[co] = 0.f;
  float da[8], db[8], dc[8];
#pragma unroll
  for (int co = 0; co < 8; ++co) {
    da[co] = DYPRO == PRO_BWD ? a.da[co] : 0.f;
    db[co] = DYPRO == PRO_BWD ? a.db[co] : 0.f;
    dc[co] = DYPRO == PRO_BWD ? a.dc[co] : 0.f;
  }
  const int tiles_y = a.Ho / THIN_TH;
  for (TileWalk walk(a.ntiles); walk.valid(); walk.advance()) {
    const int tl = walk.cur;
    const int b = tl / tiles_y, oy0 = (tl - b * tiles_y) * THIN_TH;
    __syncthreads();
    thin_stage1<W, PRO_BN>(tile, a.x, nullptr, xa, xb, 0.f, b, a.Hi, oy0 - 1);
    __syncthreads();
    const size_t opix0 = ((size_t)b * a.Ho + oy0 + ty0) * W + x;
    float du[4][8];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const size_t opix = opix0 + (size_t)p * W - p;
      const float4 g0 = *reinterpret_cast<const float4*>(a.dy + (opix + p) * 8);
      const float4 g1 = *reinterpret_cast<const float4*>(a.dy + (opix + p) * 8 + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      float y[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (DYPRO == PRO_BWD) {
        const float4 y0 = *reinterpret_cast<const float4*>(a.dy2 + (opix + p) * 8);
        const float4 y1 = *reinterpret_cast<const float4*>(a.dy2 + (opix + p) * 8 + 4);
        y[0] = y0.x; y[1] = y0.y; y[2] = y0.z; y[3] = y0.w; y[4] = y1.x; y[5] = y1.y; y[6] = y1.z; y[7] = y1.w;
      }
#pragma unroll
      for (int co = 0; co < 8; ++co) {
        du[p][co] = prologue<DYPRO>(g[co], y[co], da[co], db[co], dc[co]);
        bacc[co] += du[p][co];
      }
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float in[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) in[j] = ava_pin<7>(tile[(ty0 + j) * THIN_IC + x + kx]);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int co = 0; co < 8; ++co) acc[ky * 3 + kx][co] = fmaf(in[p + ky], du[p][co], acc[ky * 3 + kx][co]);
    }
  }
  float sv[80];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int co = 0; co < 8; ++co) sv[k * 8 + co] = acc[k][co];
#pragma unroll
  for (int co = 0; co < 8; ++co) sv[72 + co] = bacc[co];
  thin_block_reduce<80, THIN_NW>(sv, red, a.partials + (size_t)blockIdx.x * 80);
}

// convt7: CIN = 8, COUT = 1.  dG[9][8][1], db[1]
template <int W, int DYPRO>
__global__ __launch_bounds__(2 * W, W == 128 ? 2 : 1) void thin_wgrad_8to1_kernel(const WgradArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* tile = smem;
  float* coef = smem + THIN_IR * THIN_IC * 8;
  const int t = threadIdx.x, ty0 = (t / W) * 4, x = t % W;
  if (t < 96) {
    const int which = t >> 5, c = t & 31;
    const float* src = which == 0 ? a.xa : (which == 1 ? a.xb : nullptr);
    coef[t] = (src != nullptr && c < 8) ? src[c] : 0.f;
  }
  const float da = DYPRO == PRO_BWD ? a.da[0] : 0.f, db = DYPRO == PRO_BWD ? a.db[0] : 0.f,
              dc = DYPRO == PRO_BWD ? a.dc[0] : 0.f;
  float acc[9][8], bacc = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) acc[k][ci] = 0.f;
  TileStager<8, PRO_BN, THIN_IR, THIN_IC, (THIN_PLANES != 0), THIN_NT> stg;
  stg.init();
  const int tiles_y = a.Ho / THIN_TH;
  for (TileWalk walk(a.ntiles); walk.valid(); walk.advance()) {
    const int tl = walk.cur;
    const int b = tl / tiles_y, oy0 = (tl - b * tiles_y) * THIN_TH;
    __syncthreads();
    stg.load(a.x, nullptr, b, a.Hi, a.Wi, oy0 - 1, -1);
    stg.store(tile, coef);
    __syncthreads();
    const size_t opix0 = ((size_t)b * a.Ho + oy0 + ty0) * W + x;
    float du[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const size_t opix = opix0 + (size_t)p * W;
      du[p] = prologue<DYPRO>(a.dy[opix], DYPRO == PRO_BWD ? a.dy2[opix] : 0.f, da, db, dc);
    }
    bacc += (du[0] + du[1]) + (du[2] + du[3]);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float* px = tile + ((ty0 + j) * THIN_IC + x + kx) * (THIN_PLANES ? 4 : 8);
        const float4 u = *reinterpret_cast<const float4*>(px);
        const float4 w4 = *reinterpret_cast<const float4*>(px + (THIN_PLANES ? THIN_IR * THIN_IC * 4 : 4));
        const float in[8] = {u.x, u.y, u.z, u.w, w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int p = j - ky;
          if (p >= 0 && p < 4) {
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) acc[ky * 3 + kx][ci] = fmaf(in[ci], du[p], acc[ky * 3 + kx][ci]);
          }
        }
      }
  }
  float sv[73];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) sv[k * 8 + ci] = acc[k][ci];
  sv[72] = bacc;
  thin_block_reduce<73, THIN_NW>(sv, smem, a.partials + (size_t)blockIdx.x * 73);     // tiles are dead: reuse their LDS
}

// ---------------------------------------------------------------------------------------------------------
// conv1 backward in one pass (CIN = 1, COUT = 8).  The first layer needs no data gradient, only the two
// BatchNorm-backward sums of it -- and with a single input channel those follow from the weight-gradient
// correlations, so dU is read ONCE, straight from global memory, and never convolved:
//     dG'[tap][co] = sum_q xhat0[q + tap] * dU[q][co]        xhat0 = (x - mean) * invstd, zero outside the image
//     S[tap][co]   = sum_q [q + tap inside the image] dU[q][co]   = T - border rows/columns (+ corners)
//     dG[tap][co]  = gamma * dG' + beta * S                  (x_n = gamma * xhat + beta inside the image, 0 outside)
//     sum_p dx[p]        = sum_{tap,co} W[tap][co] * S[tap][co]
//     sum_p dx[p]*xhat[p] = sum_{tap,co} W[tap][co] * dG'[tap][co]
// with gamma = xa / invstd, beta = xb + mean * xa recovered from the BatchNorm scale/shift (invstd > 0).
// All of it is linear in dU, so every workgroup emits ordinary partial rows (bn [2], weight gradient [80]).
// Replaces thin_8to1_kernel<.., EPI_BWD> + thin_wgrad_1to8_kernel on the model's path.
// ---------------------------------------------------------------------------------------------------------
// Thread mapping: lane pair (2x, 2x+1) shares pixel column x; thread (x, h) owns channels 4h..4h+3 of the 8 rows
// of the tile, so every g / y load is one 16-byte slot per lane, contiguous across the wave.
// RECY: the saved activation y1 = relu(conv1(bn1 x)) (a.dy2: the ReLU mask and the y term of bn2's backward) is NOT read -- it
// is recomputed from the x window the kernel stages anyway, with exactly thin_1to8_kernel's arithmetic (bn1 as one fma, zero
// outside the image; accumulator from 0, kx outer, ky inner; + bias; ReLU; storage rounding): bit-identical values for 72 more
// fmas per pixel instead of 32 bytes per pixel from HBM (a.rc: conv1's gather weights, bias, bn1 scale / shift).
template <int W, int DYPRO, typename ACT = float, bool RECY = false>
__global__ __launch_bounds__(2 * W) void thin_bwd_fused_1to8_kernel(const FusedArgs a) {
  static_assert(!RECY || DYPRO == PRO_BWD, "only the ReLU / BatchNorm-backward prologue reads the activation");
  __shared__ float tile[THIN_IR * THIN_IC];             // xhat0 window
  __shared__ float tile_n[RECY ? THIN_IR * THIN_IC : 1];       // RECY: bn1(x) window (conv1's input, zero padded)
  __shared__ float red[THIN_NW][2][44];                       // per wave, per channel half: dG' [9][4], T [4], border sums
  __shared__ float tot[2][44];
  __shared__ float ccol[2][2][4], kcor[2][2][2][4];     // [left/right][h], [left/right][top/bottom][h]
  const int t = threadIdx.x, h = t & 1, x = t >> 1, wave = t >> 6, lane = t & 63;
  const float mean = ava_uniform(a.mean[0]), invstd = ava_uniform(a.invstd[0]);
  const float ha = invstd, hb = -mean * invstd;         // xhat = ha * x + hb
  float acc[9][4], T[4], Rt[4], Rb[4], Cc[4], Kt[4], Kb[4];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[k][c] = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) T[c] = Rt[c] = Rb[c] = Cc[c] = Kt[c] = Kb[c] = 0.f;
  float da[4], db[4], dc[4];
  if (DYPRO == PRO_BWD && a.fin.acc != nullptr) {   // bn2's A, Bc, Cc from the sums conv2's backward accumulated (bn_acc.h)
    __shared__ float coef[96];
    __shared__ double accvals[64];
    bn_coef_from_acc(coef, accvals, a.fin, 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) { da[c] = coef[4 * h + c]; db[c] = coef[32 + 4 * h + c]; dc[c] = coef[64 + 4 * h + c]; }
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      da[c] = DYPRO == PRO_BWD ? a.da[4 * h + c] : 0.f;
      db[c] = DYPRO == PRO_BWD ? a.db[4 * h + c] : 0.f;
      dc[c] = DYPRO == PRO_BWD ? a.dc[4 * h + c] : 0.f;
    }
  }
  const bool edge_col = x == 0 || x == W - 1;
  const int tiles_y = a.Ho / THIN_TH;
  __shared__ float w1s[RECY ? 80 : 1];                  // RECY: conv1's gather weights [9][8] and bias [8] (read per use: 36 VGPRs otherwise)
  float ca1 = 0.f, cb1 = 0.f;
  if constexpr (RECY) {
    if (t < 72) w1s[t] = a.rc.G1[t];
    else if (t < 80) w1s[t] = a.rc.bias1[t - 72];
    ca1 = ava_uniform(a.rc.pa1[0]); cb1 = ava_uniform(a.rc.pb1[0]);
  }
  ThinWindow<W, PRO_BN> win;
  TileWalk walk(a.ntiles, a.sweep == 0);
  if (walk.valid()) { const int tl = walk.cur, b0 = tl / tiles_y; win.load(a.x, nullptr, b0, a.Hi, (tl - b0 * tiles_y) * THIN_TH - 1); }
  for (; walk.valid(); walk.advance()) {
    const int tl = walk.cur;
    const int b = tl / tiles_y, oy0 = (tl - b * tiles_y) * THIN_TH;
    __syncthreads();
    win.store(tile, ha, hb, 0.f);
    if constexpr (RECY) win.store(tile_n, ca1, cb1, 0.f);
    const size_t o0 = (((size_t)b * a.Ho + oy0) * W + x) * 8 + 4 * h;
    float du[THIN_TH][4];
    float yrec[RECY ? THIN_TH : 1][4];
    if constexpr (RECY) {
      float4 gq[THIN_TH];
#pragma unroll
      for (int p = 0; p < THIN_TH; ++p) gq[p] = *reinterpret_cast<const float4*>(a.dy + o0 + (size_t)p * W * 8);     // in flight over the barrier
      __syncthreads();                                  // tile_n (and tile) written
      int hoff = 4 * h;
      asm volatile("" : "+v"(hoff));                    // keeps the weight reads inside the tile loop (hoisted they cost 36 VGPRs and a resident workgroup)
#pragma unroll
      for (int p = 0; p < THIN_TH; ++p)
#pragma unroll
        for (int c = 0; c < 4; ++c) yrec[p][c] = 0.f;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        float inn[THIN_IR];
#pragma unroll
        for (int j = 0; j < THIN_IR; ++j) inn[j] = ava_pin<2>(tile_n[j * THIN_IC + x + kx]);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const avaf4 wq = *reinterpret_cast<const avaf4*>(w1s + (ky * 3 + kx) * 8 + hoff);
#pragma unroll
          for (int p = 0; p < THIN_TH; ++p)
#pragma unroll
            for (int c = 0; c < 4; ++c) yrec[p][c] = fmaf(inn[p + ky], wq[c], yrec[p][c]);
        }
      }
      const avaf4 bq = *reinterpret_cast<const avaf4*>(w1s + 72 + hoff);
#pragma unroll
      for (int p = 0; p < THIN_TH; ++p) {
        float yv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) yv[c] = ava_stored<ACT>(fmaxf(yrec[p][c] + bq[c], 0.f));
        const float gv[4] = {gq[p].x, gq[p].y, gq[p].z, gq[p].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          du[p][c] = prologue<DYPRO>(gv[c], yv[c], da[c], db[c], dc[c]);
          T[c] += du[p][c];
        }
      }
    }
#pragma unroll
    for (int p = 0; p < (RECY ? 0 : THIN_TH); ++p) {
      const float4 g = *reinterpret_cast<const float4*>(a.dy + o0 + (size_t)p * W * 8);
      float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
      if (DYPRO == PRO_BWD) {
        const avaf4 yv = ava_ld4<ACT>(ava_as<ACT>(a.dy2) + o0 + (size_t)p * W * 8);
        y = make_float4(yv[0], yv[1], yv[2], yv[3]);
      }
      du[p][0] = prologue<DYPRO>(g.x, y.x, da[0], db[0], dc[0]);
      du[p][1] = prologue<DYPRO>(g.y, y.y, da[1], db[1], dc[1]);
      du[p][2] = prologue<DYPRO>(g.z, y.z, da[2], db[2], dc[2]);
      du[p][3] = prologue<DYPRO>(g.w, y.w, da[3], db[3], dc[3]);
#pragma unroll
      for (int c = 0; c < 4; ++c) T[c] += du[p][c];
    }
    // border sums: image row 0 is row 0 of an image's first tile, row H-1 is row 7 of its last tile (wave-uniform)
    if (oy0 == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) { Rt[c] += du[0][c]; Kt[c] += edge_col ? du[0][c] : 0.f; }
    }
    if (oy0 + THIN_TH == a.Ho) {
#pragma unroll
      for (int c = 0; c < 4; ++c) { Rb[c] += du[THIN_TH - 1][c]; Kb[c] += edge_col ? du[THIN_TH - 1][c] : 0.f; }
    }
    if (edge_col) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < THIN_TH; ++p) s += du[p][c];
        Cc[c] += s;
      }
    }
    if constexpr (!RECY) __syncthreads();
    if (walk.has_next()) { const int tn = walk.next(), bn = tn / tiles_y; win.load(a.x, nullptr, bn, a.Hi, (tn - bn * tiles_y) * THIN_TH - 1); }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float in[THIN_IR];
#pragma unroll
      for (int j = 0; j < THIN_IR; ++j) in[j] = ava_pin<3>(tile[j * THIN_IC + x + kx]);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int p = 0; p < THIN_TH; ++p)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[ky * 3 + kx][c] = fmaf(in[p + ky], du[p][c], acc[ky * 3 + kx][c]);
    }
  }
  // ---- workgroup totals per channel half (lanes of equal parity), fixed order ----
  float sv[44];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) sv[k * 4 + c] = acc[k][c];
#pragma unroll
  for (int c = 0; c < 4; ++c) { sv[36 + c] = T[c]; sv[40 + c] = Rt[c]; }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 44; ++i) {
    float v = sv[i];
#pragma unroll
    for (int o = 32; o > 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane < 2) red[wave][lane][i] = v;
  }
  float rb[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float v = Rb[c];
#pragma unroll
    for (int o = 32; o > 1; o >>= 1) v += __shfl_xor(v, o, 64);
    rb[c] = v;
  }
  __shared__ float rbot[THIN_NW][2][4];
  if (lane < 2) {
#pragma unroll
    for (int c = 0; c < 4; ++c) rbot[wave][lane][c] = rb[c];
  }
  if (edge_col) {
    const int side = x == 0 ? 0 : 1;
#pragma unroll
    for (int c = 0; c < 4; ++c) { ccol[side][h][c] = Cc[c]; kcor[side][0][h][c] = Kt[c]; kcor[side][1][h][c] = Kb[c]; }
  }
  __syncthreads();
  if (t < 88) {
    const int hh = t / 44, i = t - 44 * hh;
    tot[hh][i] = thin_sum_waves<THIN_NW>([&](int w) { return red[w][hh][i]; });
  }
  __syncthreads();
  float* scratch = &red[0][0][0];                        // [2][72] products for the two BatchNorm sums (red is dead)
  if (t < 72) {
    const int tap = t >> 3, co = t & 7, hh = co >> 2, c = co & 3, ky = tap / 3, kx = tap - 3 * ky;
    float S = tot[hh][36 + c];
    if (ky == 0) S -= tot[hh][40 + c];
    if (ky == 2) S -= thin_sum_waves<THIN_NW>([&](int w) { return rbot[w][hh][c]; });
    if (kx == 0) S -= ccol[0][hh][c];
    if (kx == 2) S -= ccol[1][hh][c];
    if (ky == 0 && kx == 0) S += kcor[0][0][hh][c];
    if (ky == 0 && kx == 2) S += kcor[1][0][hh][c];
    if (ky == 2 && kx == 0) S += kcor[0][1][hh][c];
    if (ky == 2 && kx == 2) S += kcor[1][1][hh][c];
    const float xa = a.xa[0], xb = a.xb[0];
    const float gamma = xa / invstd, beta = fmaf(mean, xa, xb);
    const float dgp = tot[hh][tap * 4 + c];
    a.wg_partials[(size_t)blockIdx.x * 80 + t] = fmaf(gamma, dgp, beta * S);
    const float w = a.Gb[(8 - tap) * 8 + co];            // forward weight W[tap][co] out of the flipped backward pack
    scratch[t] = w * S;
    scratch[72 + t] = w * dgp;
  } else {
    if (t < 80) a.wg_partials[(size_t)blockIdx.x * 80 + t] = tot[(t - 72) >> 2][36 + ((t - 72) & 3)];   // bias gradient = T
  }
  __syncthreads();
  if (t < 2) {
    float s = 0.f;
    for (int i = 0; i < 72; ++i) s += scratch[72 * t + i];
    if (a.acc_out != nullptr) bn_acc_add(a.acc_out, 32 * t, s);       // one channel: values 0 (sum dx) and 32 (sum dx * xhat)
    else a.bn_partials[(size_t)blockIdx.x * 2 + t] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------
// convt7 backward (CIN = 8, COUT = 1; dU = the 1-channel seed gradient or PRO_BWD of it).  Same identity as conv1:
// with xhat0 = (x - mean) * invstd zero padded,  dG'[tap][ci] = sum_p xhat0[p + tap][ci] * dU[p]  gives
//   dG[tap][ci] = gamma_ci dG' + beta_ci S[tap],   sum_q dx[q][ci] = sum_tap G[tap][ci] S[tap],
//   sum_q dx[q][ci] xhat[q][ci] = sum_tap G[tap][ci] dG'[tap][ci]
// (S[tap] = sum of dU over the pixels whose tap lands inside the image), so the data-gradient kernel
// (thin_1to8_kernel<.., EPI_NONE>) only has to write dx and never reads x; this kernel reads x once.
// ---------------------------------------------------------------------------------------------------------
#ifdef AVA_LAB
#define W 128   /* lab-only kernels: 128-wide images only */   // the LDS-staged form of this kernel (AVA_THIN_STATS_DIRECT=0); the library runs the direct form below
template <int DYPRO>
__global__ __launch_bounds__(256, 2) void thin_wgrad_stats_8to1_kernel(const FusedArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* tile = smem;                                   // [10][130][8] xhat0
  float* coef = smem + THIN_IR * THIN_IC * 8;           // [3][32]
  float* aux = coef + 96;                               // [4] row sums per wave, [4] column sums, [4] corners
  const int t = threadIdx.x, ty0 = (t >> 7) * 4, x = t & 127, wave = t >> 6, lane = t & 63;
  if (t < 96) {
    const int which = t >> 5, c = t & 31;
    float v = 0.f;
    if (c < 8) v = which == 0 ? a.invstd[c] : (which == 1 ? -a.mean[c] * a.invstd[c] : 0.f);
    coef[t] = v;
  }
  const float da = DYPRO == PRO_BWD ? a.da[0] : 0.f, db = DYPRO == PRO_BWD ? a.db[0] : 0.f,
              dc = DYPRO == PRO_BWD ? a.dc[0] : 0.f;
  avaf2 acc2[9][4];                                     // channel pairs: the 288 FMAs per tile issue as 144 v_pk_fma_f32
  float T = 0.f, R = 0.f, Cc = 0.f, K = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc2[k][q] = avaf2{0.f, 0.f};
  const bool edge_col = x == 0 || x == W - 1;
  TileStager<8, PRO_BN, THIN_IR, THIN_IC, (THIN_PLANES != 0)> stg;
  stg.init();
  const int tiles_y = a.Ho / THIN_TH;
  for (TileWalk walk(a.ntiles); walk.valid(); walk.advance()) {
    const int tl = walk.cur;
    const int b = tl / tiles_y, oy0 = (tl - b * tiles_y) * THIN_TH;
    __syncthreads();
    stg.load(a.x, nullptr, b, a.Hi, a.Wi, oy0 - 1, -1);
    stg.store(tile, coef);
    __syncthreads();
    const size_t opix0 = ((size_t)b * a.Ho + oy0 + ty0) * W + x;
    float du[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const size_t opix = opix0 + (size_t)p * W;
      du[p] = prologue<DYPRO>(a.dy[opix], DYPRO == PRO_BWD ? a.dy2[opix] : 0.f, da, db, dc);
    }
    const float strip = (du[0] + du[1]) + (du[2] + du[3]);
    T += strip;
    // border sums; a thread's role is fixed by (ty0, x) -- see thin_bwd_fused_1to8_kernel
    const bool top = oy0 + ty0 == 0, bottom = oy0 + ty0 + 3 == a.Ho - 1;
    if (top || bottom) {
      const float v = top ? du[0] : du[3];
      R += v;
      K += edge_col ? v : 0.f;
    }
    Cc += edge_col ? strip : 0.f;
    avaf2 dd[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) { const float dp = ava_pin<4>(du[p]); dd[p] = avaf2{dp, dp}; }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float* px = tile + ((ty0 + j) * THIN_IC + x + kx) * (THIN_PLANES ? 4 : 8);
        const avaf4 u = *reinterpret_cast<const avaf4*>(px);
        const avaf4 v = *reinterpret_cast<const avaf4*>(px + (THIN_PLANES ? THIN_IR * THIN_IC * 4 : 4));
        const avaf2 in2[4] = {avaf2{u[0], u[1]}, avaf2{u[2], u[3]}, avaf2{v[0], v[1]}, avaf2{v[2], v[3]}};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int p = j - ky;
          if (p >= 0 && p < 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              acc2[ky * 3 + kx][q] = __builtin_elementwise_fma(in2[q], dd[p], acc2[ky * 3 + kx][q]);
          }
        }
      }
  }
  float sv[73];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) sv[k * 8 + ci] = acc2[k][ci >> 1][ci & 1];
  sv[72] = T;
  float* tot = smem + 4 * 73;                           // [73] behind the reduction scratch (tiles are dead)
  thin_block_reduce<73, 4>(sv, smem, tot);
  const float rw = wave_sum(R);
  if (lane == 0) aux[wave] = rw;                        // waves 0,1: image row 0;  waves 2,3: image row H-1
  if (edge_col) {
    const int role = (t >> 7) * 2 + (x == 0 ? 0 : 1);   // 0: x=0 top half, 1: x=127 top half, 2: x=0 bottom, 3: x=127 bottom
    aux[4 + role] = Cc;
    aux[8 + role] = K;
  }
  __syncthreads();
  float* scratch = smem + 512;                          // [2][72]
  if (t < 72) {
    const int tap = t >> 3, ci = t & 7, ky = tap / 3, kx = tap - 3 * ky;
    float S = tot[72];
    if (ky == 0) S -= aux[0] + aux[1];
    if (ky == 2) S -= aux[2] + aux[3];
    if (kx == 0) S -= aux[4] + aux[6];
    if (kx == 2) S -= aux[5] + aux[7];
    if (ky == 0 && kx == 0) S += aux[8];
    if (ky == 0 && kx == 2) S += aux[9];
    if (ky == 2 && kx == 0) S += aux[10];
    if (ky == 2 && kx == 2) S += aux[11];
    const float xa = a.xa[ci], xb = a.xb[ci], mean = a.mean[ci], invstd = a.invstd[ci];
    const float gamma = xa / invstd, beta = fmaf(mean, xa, xb);
    const float dgp = tot[t];
    a.wg_partials[(size_t)blockIdx.x * 73 + t] = fmaf(gamma, dgp, beta * S);
    const float w = a.Gb[(8 - tap) * 8 + ci];            // forward gather weight G[tap][ci] out of the backward pack
    scratch[t] = w * S;
    scratch[72 + t] = w * dgp;
  } else if (t == 72) {
    a.wg_partials[(size_t)blockIdx.x * 73 + 72] = tot[72];                    // bias gradient = T
  }
  __syncthreads();
  if (t < 16) {
    const int which = t >> 3, ci = t & 7;
    float s = 0.f;
    for (int tap = 0; tap < 9; ++tap) s += scratch[72 * which + tap * 8 + ci];
    a.bn_partials[(size_t)blockIdx.x * 16 + t] = s;
  }
}

#undef W
#endif  // AVA_LAB

// "Direct" form of thin_wgrad_stats_8to1_kernel: the same correlation, indexed by the INPUT pixel q instead of the
// output pixel p,   dG'[tap][ci] = sum_q xhat[q][ci] * dU[q - tap]   (dU zero outside the image),
// so the 8-channel tensor x needs no neighbourhood and is read straight from global memory into registers by the
// thread that owns the pixel (lane pairs share a pixel column, thread (x, h) owns channels 4h..4h+3 of the 8 rows of
// the tile, exactly like thin_bwd_fused_1to8_kernel), and only the 1-channel dU window goes through LDS (5 KB).
// The LDS-staged form keeps one 41.6 KB window per workgroup in flight and runs at 2.7 TB/s; this one has no staging
// role at all, more resident workgroups and eight independent 16-byte loads per thread in flight.
template <int W, int DYPRO, typename ACT = float>
__global__ __launch_bounds__(2 * W) void thin_wgrad_stats_8to1_direct_kernel(const FusedArgs a) {
  __shared__ float tile[THIN_IR * THIN_IC];             // dU window (prologue applied, zero outside the image)
  __shared__ float red[THIN_NW][2][36];                       // per wave, per channel half: dG' [9][4]
  __shared__ float sc[THIN_NW][9];                            // per wave: T, Rt, Rb, Cl, Cr, Ktl, Ktr, Kbl, Kbr
  __shared__ float tot[2][36];
  __shared__ float stot[9];
  __shared__ float scratch[2][72];
  const int t = threadIdx.x, h = t & 1, x = t >> 1, wave = t >> 6, lane = t & 63;
  float ha[4], hb[4];                                   // xhat = ha * x + hb for this half's channels
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float is = a.invstd[4 * h + c];
    ha[c] = is;
    hb[c] = -a.mean[4 * h + c] * is;
  }
  const float da = DYPRO == PRO_BWD ? a.da[0] : 0.f, db = DYPRO == PRO_BWD ? a.db[0] : 0.f,
              dc = DYPRO == PRO_BWD ? a.dc[0] : 0.f;
  avaf2 acc[9][2];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k][0] = acc[k][1] = avaf2{0.f, 0.f};
  float T = 0.f, Rt = 0.f, Rb = 0.f, Cl = 0.f, Cr = 0.f, Ktl = 0.f, Ktr = 0.f, Kbl = 0.f, Kbr = 0.f;
  const float own = h == 0 ? 1.f : 0.f;                 // the scalar sums of dU are taken by one lane of each pair
  const int tiles_y = a.Hi / THIN_TH;
  ThinWindow<W, DYPRO> win;
  TileWalk walk(a.ntiles, a.sweep == 0);
  if (walk.valid()) { const int tl = walk.cur, b0 = tl / tiles_y; win.load(a.dy, a.dy2, b0, a.Ho, (tl - b0 * tiles_y) * THIN_TH - 1); }
  for (; walk.valid(); walk.advance()) {
    const int tl = walk.cur;
    const int b = tl / tiles_y, oy0 = (tl - b * tiles_y) * THIN_TH;
    __syncthreads();
    win.store(tile, da, db, dc);
    const size_t o0 = (((size_t)b * a.Hi + oy0) * W + x) * 8 + 4 * h;
    avaf2 xh[THIN_TH][2];
#pragma unroll
    for (int r = 0; r < THIN_TH; ++r) {
      const avaf4 v = ava_ld4<ACT>(ava_as<ACT>(a.x) + o0 + (size_t)r * W * 8);
      xh[r][0] = avaf2{fmaf(ha[0], v[0], hb[0]), fmaf(ha[1], v[1], hb[1])};
      xh[r][1] = avaf2{fmaf(ha[2], v[2], hb[2]), fmaf(ha[3], v[3], hb[3])};
    }
    __syncthreads();
    if (walk.has_next()) { const int tn = walk.next(), bn = tn / tiles_y; win.load(a.dy, a.dy2, bn, a.Ho, (tn - bn * tiles_y) * THIN_TH - 1); }
    // sums of dU over this thread's column of the tile and the image-border rows / columns / corners
    {
      float col = 0.f;
#pragma unroll
      for (int r = 0; r < THIN_TH; ++r) col += tile[(r + 1) * THIN_IC + x + 1];
      const float top = oy0 == 0 ? tile[1 * THIN_IC + x + 1] : 0.f;
      const float bot = oy0 + THIN_TH == a.Ho ? tile[THIN_TH * THIN_IC + x + 1] : 0.f;
      T += own * col;
      Rt += own * top;
      Rb += own * bot;
      if (x == 0) { Cl += own * col; Ktl += own * top; Kbl += own * bot; }
      if (x == W - 1) { Cr += own * col; Ktr += own * top; Kbr += own * bot; }
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float d[THIN_IR];
#pragma unroll
      for (int j = 0; j < THIN_IR; ++j) d[j] = ava_pin<5>(tile[j * THIN_IC + x + 2 - kx]);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int r = 0; r < THIN_TH; ++r) {
          const avaf2 dv = {d[r + 2 - ky], d[r + 2 - ky]};
#pragma unroll
          for (int q = 0; q < 2; ++q) acc[ky * 3 + kx][q] = __builtin_elementwise_fma(dv, xh[r][q], acc[ky * 3 + kx][q]);
        }
    }
  }
  // ---- workgroup totals: dG' per channel half over lanes of equal parity, the nine scalar sums over all lanes ----
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float v = acc[k][q][e];
#pragma unroll
        for (int o = 32; o > 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane < 2) red[wave][lane][k * 4 + 2 * q + e] = v;
      }
  {
    const float sv[9] = {T, Rt, Rb, Cl, Cr, Ktl, Ktr, Kbl, Kbr};
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const float r = wave_sum(sv[i]);
      if (lane == 0) sc[wave][i] = r;
    }
  }
  __syncthreads();
  if (t < 72) {
    const int hh = t / 36, i = t - 36 * hh;
    tot[hh][i] = thin_sum_waves<THIN_NW>([&](int w) { return red[w][hh][i]; });
  } else if (t < 81) {
    const int i = t - 72;
    stot[i] = thin_sum_waves<THIN_NW>([&](int w) { return sc[w][i]; });
  }
  __syncthreads();
  if (t < 72) {
    const int tap = t >> 3, ci = t & 7, hh = ci >> 2, c = ci & 3, ky = tap / 3, kx = tap - 3 * ky;
    float S = stot[0];
    if (ky == 0) S -= stot[1];
    if (ky == 2) S -= stot[2];
    if (kx == 0) S -= stot[3];
    if (kx == 2) S -= stot[4];
    if (ky == 0 && kx == 0) S += stot[5];
    if (ky == 0 && kx == 2) S += stot[6];
    if (ky == 2 && kx == 0) S += stot[7];
    if (ky == 2 && kx == 2) S += stot[8];
    const float xa = a.xa[ci], xb = a.xb[ci], mean = a.mean[ci], invstd = a.invstd[ci];
    const float gamma = xa / invstd, beta = fmaf(mean, xa, xb);
    const float dgp = tot[hh][tap * 4 + c];
    a.wg_partials[(size_t)blockIdx.x * 73 + t] = fmaf(gamma, dgp, beta * S);
    const float w = a.Gb[(8 - tap) * 8 + ci];            // forward gather weight G[tap][ci] out of the backward pack
    scratch[0][t] = w * S;
    scratch[1][t] = w * dgp;
  } else if (t == 72) {
    a.wg_partials[(size_t)blockIdx.x * 73 + 72] = stot[0];                    // bias gradient = T
  }
  __syncthreads();
  if (t < 16) {
    const int which = t >> 3, ci = t & 7;
    float s = 0.f;
    for (int tap = 0; tap < 9; ++tap) s += scratch[which][tap * 8 + ci];
    if (a.acc_out != nullptr) bn_acc_add(a.acc_out, which * 32 + ci, s);     // accumulated for the consumer's prologue
    else a.bn_partials[(size_t)blockIdx.x * 16 + t] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------
// convt6 forward (8 -> 8 channels, stride-2 transposed conv, 64x64 -> 128x128) in the direct form of the thin
// kernels.  On the matrix cores half of every MFMA of this layer is padding (8 of 16 rows) and a staging role is
// needed; its 0.6 GMAC fit the packed-FMA rate many times over, so the layer is HBM-bound if nothing else is in
// the way.  Thread (xo, h) of a lane pair owns output column xo (low-resolution column c = xo >> 1, parity px) and
// output channels 4h..4h+3 of the 8 output rows of a tile; it reads the 5 x 2 low-resolution input pixels it needs
// straight from global memory (BatchNorm applied, zero beyond the image), and the per-lane weight set -- which taps
// exist depends on px, which channels on h -- comes from a 3 KB LDS table as 16-byte reads (4 distinct addresses per
// wave).  out(2r+py, 2c+px) = bias + sum over ky with (ky != 1) == py, kx with (kx != 1) == px of
// G[ky][kx] . x_n(r + (ky == 0), c + (kx == 0));  every store instruction writes 1 KB of full lines.
// ---------------------------------------------------------------------------------------------------------
#define UP88_WSTRIDE 49     // float4 per (px, h) weight set: 48 used, padded so the four sets start in different banks
template <typename ACT>
__global__ __launch_bounds__(256, 3) void up88_direct_kernel(const ConvArgs a) {
  __shared__ __align__(16) float wt[4 * UP88_WSTRIDE * 4];   // [px][h][ky][slot][ci] x 4 output channels
  __shared__ float red[4][2][8];
  const int t = threadIdx.x, h = t & 1, xo = t >> 1, c = xo >> 1, px = xo & 1, lane = t & 63, wave = t >> 6;
  for (int i = t; i < 4 * 48 * 4; i += 256) {
    const int co4 = i & 3, e = i >> 2, set = e / 48, r = e - 48 * set;       // r = (ky*2 + slot)*8 + ci
    const int spx = set >> 1, sh = set & 1, ky = r / 16, slot = (r >> 3) & 1, ci = r & 7;
    const int kx = slot == 0 ? (spx ? 2 : 1) : 0;                              // slot 0: column c, slot 1: column c + 1
    const bool exists = slot == 0 || spx == 1;
    // (bf16 arithmetic, ACT = bfloat16: weights and BatchNorm outputs rounded to bfloat16, exact products, fp32 accumulation --
    // the semantics of the matrix-core layers' one-limb form; ava_stored<float> is the identity)
    wt[(set * UP88_WSTRIDE + r) * 4 + co4] = exists ? ava_stored<ACT>(a.G[((ky * 3 + kx) * 8 + ci) * 8 + 4 * sh + co4]) : 0.f;
  }
  float sca[8], shf[8];
  if (a.fin.acc != nullptr) {               // BatchNorm of the input: sums accumulated by the producer (bn_acc.h), finalised here
    __shared__ float coef[96];
    __shared__ double accvals[64];
    bn_coef_from_acc(coef, accvals, a.fin, 0);
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) { sca[ci] = ava_uniform(coef[ci]); shf[ci] = ava_uniform(coef[32 + ci]); }
  } else {
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) { sca[ci] = ava_uniform(a.pa[ci]); shf[ci] = ava_uniform(a.pb[ci]); }   // scalar registers
  }
  avaf2 bias2[2] = {avaf2{a.bias[4 * h], a.bias[4 * h + 1]}, avaf2{a.bias[4 * h + 2], a.bias[4 * h + 3]}};
  avaf2 s1[2] = {avaf2{0.f, 0.f}, avaf2{0.f, 0.f}}, s2[2] = {avaf2{0.f, 0.f}, avaf2{0.f, 0.f}};
  const float* wl = wt + ((px * 2 + h) * UP88_WSTRIDE) * 4;
  const bool colB = c + 1 < a.Wi;
  const int tiles_y = a.Ho / 8;
  __syncthreads();
  for (TileWalk walk(a.ntiles, false); walk.valid(); walk.advance()) {     // sweeping walk: 40.6 vs 42.3 us chunked
    const int tl = walk.cur;
    const int b = tl / tiles_y, oy0 = (tl - b * tiles_y) * 8, r0 = oy0 >> 1;
    avaf2 acc[8][2];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j][0] = acc[j][1] = avaf2{0.f, 0.f};
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
      // ---- input pixels of this column slot: rows r0 .. r0+4, column c + slot, all 8 channels ----
      float xn[5][8];
      const bool colok = slot == 0 || colB;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int gy = r0 + j;
        const bool ok = colok && gy < a.Hi;
        const ACT* __restrict__ pp = ava_as<ACT>(a.in) + (((size_t)b * a.Hi + min(gy, a.Hi - 1)) * a.Wi + min(c + slot, a.Wi - 1)) * 8;
        const avaf4 v0 = ava_ld4<ACT>(pp), v1 = ava_ld4<ACT>(pp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xn[j][e] = ok ? ava_stored<ACT>(fmaf(sca[e], v0[e], shf[e])) : 0.f;
          xn[j][4 + e] = ok ? ava_stored<ACT>(fmaf(sca[4 + e], v1[e], shf[4 + e])) : 0.f;
        }
      }
#pragma unroll
      for (int ci = 0; ci < 8; ++ci) {
        __builtin_amdgcn_sched_barrier(0);              // one (slot, ci) at a time: hoisting all 48 weight reads costs 192 VGPRs
        const avaf4 w0 = *reinterpret_cast<const avaf4*>(wl + ((0 * 2 + slot) * 8 + ci) * 4);   // ky = 0: row r + 1, odd output rows
        const avaf4 w1 = *reinterpret_cast<const avaf4*>(wl + ((1 * 2 + slot) * 8 + ci) * 4);   // ky = 1: row r, even output rows
        const avaf4 w2 = *reinterpret_cast<const avaf4*>(wl + ((2 * 2 + slot) * 8 + ci) * 4);   // ky = 2: row r, odd output rows
        const avaf2 w0a = {w0[0], w0[1]}, w0b = {w0[2], w0[3]}, w1a = {w1[0], w1[1]}, w1b = {w1[2], w1[3]},
                    w2a = {w2[0], w2[1]}, w2b = {w2[2], w2[3]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float xr0 = ava_pin<6>(xn[r][ci]), xr1 = ava_pin<6>(xn[r + 1][ci]);      // (see ava_pin)
          const avaf2 x0 = {xr0, xr0}, x1 = {xr1, xr1};
          acc[2 * r][0] = __builtin_elementwise_fma(x0, w1a, acc[2 * r][0]);
          acc[2 * r][1] = __builtin_elementwise_fma(x0, w1b, acc[2 * r][1]);
          acc[2 * r + 1][0] = __builtin_elementwise_fma(x0, w2a, acc[2 * r + 1][0]);
          acc[2 * r + 1][1] = __builtin_elementwise_fma(x0, w2b, acc[2 * r + 1][1]);
          acc[2 * r + 1][0] = __builtin_elementwise_fma(x1, w0a, acc[2 * r + 1][0]);
          acc[2 * r + 1][1] = __builtin_elementwise_fma(x1, w0b, acc[2 * r + 1][1]);
        }
      }
    }
    const size_t o0 = (((size_t)b * a.Ho + oy0) * a.Wo + xo) * 8 + 4 * h;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      avaf2 v0 = acc[j][0] + bias2[0], v1 = acc[j][1] + bias2[1];
      v0 = avaf2{fmaxf(v0[0], 0.f), fmaxf(v0[1], 0.f)};
      v1 = avaf2{fmaxf(v1[0], 0.f), fmaxf(v1[1], 0.f)};
      v0 = avaf2{ava_stored<ACT>(v0[0]), ava_stored<ACT>(v0[1])};               // statistics of what is stored
      v1 = avaf2{ava_stored<ACT>(v1[0]), ava_stored<ACT>(v1[1])};
      s1[0] += v0; s1[1] += v1;
      s2[0] = __builtin_elementwise_fma(v0, v0, s2[0]);
      s2[1] = __builtin_elementwise_fma(v1, v1, s2[1]);
      ava_st4_wt<ACT>(ava_as<ACT>(a.out) + o0 + (size_t)j * a.Wo * 8, avaf4{v0[0], v0[1], v1[0], v1[1]});
    }
  }
  // ---- per-channel sums: lanes of equal parity hold the same 4 channels; waves, then workgroup, fixed order ----
  float sv[8] = {s1[0][0], s1[0][1], s1[1][0], s1[1][1], s2[0][0], s2[0][1], s2[1][0], s2[1][1]};
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float v = sv[i];
#pragma unroll
    for (int o = 32; o > 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane < 2) red[wave][lane][i] = v;               // lane = h
  }
  __syncthreads();
  if (t < 16 && a.acc_out != nullptr) {                    // sums accumulated for the consumer's prologue (bn_acc.h)
    const int which = t >> 3, co = t & 7, hh = co >> 2, i = which * 4 + (co & 3);
    bn_acc_add(a.acc_out, which * 32 + co, (red[0][hh][i] + red[1][hh][i]) + (red[2][hh][i] + red[3][hh][i]));
    return;
  }
  if (a.acc_out != nullptr) return;
  if (t < 16 && a.partials != nullptr) {
    const int which = t >> 3, co = t & 7, hh = co >> 2, i = which * 4 + (co & 3);
    a.partials[(size_t)blockIdx.x * 16 + t] = (red[0][hh][i] + red[1][hh][i]) + (red[2][hh][i] + red[3][hh][i]);
  }
  thin_zero_rows<16>(a.partials, a.part_rows);
}

