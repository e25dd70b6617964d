// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain) for
// the twelve nn.Linear layers of the VAE (ava/models/vae.py:142-154,225-232,258-261) and their
// backward products.  One kernel serves the three products of a Linear layer:
//   forward   Y  = X  W^T + b      A = X  [M,K] (k-major)   B = W  [N,K] (k-major)
//   backward  dX = dY W            A = dY [M,K] (k-major)   B = W  [K,N] (n-major)
//   backward  dW = dY^T X          A = dY [K,M] (m-major)   B = X  [K,N] (n-major),  db = column sums of dY
//
// Tiling: 256 threads = 4 waves in a 2x2 grid over a BMxBN tile (128x128 or 64x64), K step 16.
// LDS holds both operands k-major ([16][BM+pad]): an MFMA operand fragment is then 32 consecutive
// floats of one LDS row (ds_read_b32, conflict free).  k-major sources are transposed by the LDS
// write (pad chosen so the 4 strided writes of a lane group hit distinct banks); m-major sources
// are copied with 16-byte writes.  The next K tile is prefetched into registers while the current
// one is multiplied.  Small-MN / large-K products are split along K into partial slabs that a
// second kernel sums (fixed order: deterministic) and finishes with bias + activation.
#include <stdlib.h>
#include "gemm.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define AVA_OPAQUE4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))

// One operand tile (BT rows/cols x BK) global -> registers.  Straight-line code: every lane always loads from a
// clamped (valid) address and out-of-range elements are zeroed with selects when they are written to LDS -- with the loads under exec-masked
// branches hipcc serialises them behind s_waitcnt vmcnt(N) (1334 instructions and 51 waits per K step in the
// first version of this kernel).  VEC: 16-byte loads legal (leading dimension, base pointer and the relevant
// extent are multiples of 4); otherwise four scalar loads with their own clamps.
template <int BT, int BK, bool KMAJ, bool VEC>
struct OperandLoader {
  static constexpr int N = BT * BK / 4 / 256;
  float4 r[N];
  int off[N];       // element offset of this thread's vector inside the matrix for k0 = 0 (row/col clamped)
  unsigned okrow;   // bit i: row/col of vector i is inside the matrix
  unsigned okk;     // bit i: K position of vector i (current registers) is inside this split's K range (VEC path)
  int kk[N];        // KMAJ: k offset (4*kq) of vector i ; MMAJ: k row of vector i

  __device__ __forceinline__ void init(int t0, int extent, int ld) {
    okrow = 0u;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int v = threadIdx.x + 256 * i;
      if (KMAJ) {
        const int row = v / (BK / 4), kq = v % (BK / 4);
        const int g = t0 + row;
        if (g < extent) okrow |= 1u << i;
        off[i] = min(g, extent - 1) * ld;
        kk[i] = 4 * kq;
      } else {
        const int k = v / (BT / 4), m4 = v % (BT / 4);
        const int g = t0 + 4 * m4;
        if (g < extent) okrow |= 1u << i;           // VEC: extent % 4 == 0, so the whole vector is in or out
        off[i] = VEC ? min(g, extent - 4) : g;
        kk[i] = k;
      }
    }
  }

  __device__ __forceinline__ void load(const float* __restrict__ base, int ld, int extent, int k0, int kend, int K) {
    okk = VEC ? 0u : 0xffffffffu;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float4 x;
      if (KMAJ) {
        const int gk = k0 + kk[i];
        if (VEC) {
          x = *reinterpret_cast<const float4*>(base + off[i] + min(gk, K - 4));
          if (gk < kend) okk |= 1u << i;              // zeroing happens in store(): see the struct comment
        } else {
          const float* p = base + off[i];
          const bool ok = (okrow >> i) & 1u;
          x.x = (ok && gk + 0 < kend) ? p[min(gk + 0, K - 1)] : 0.f;
          x.y = (ok && gk + 1 < kend) ? p[min(gk + 1, K - 1)] : 0.f;
          x.z = (ok && gk + 2 < kend) ? p[min(gk + 2, K - 1)] : 0.f;
          x.w = (ok && gk + 3 < kend) ? p[min(gk + 3, K - 1)] : 0.f;
        }
      } else {
        const int gk = k0 + kk[i];
        const float* p = base + (size_t)min(gk, K - 1) * ld;
        if (VEC) {
          x = *reinterpret_cast<const float4*>(p + off[i]);
          if (gk < kend) okk |= 1u << i;
        } else {
          const int g = off[i];
          const bool okk = gk < kend;
          x.x = (okk && g + 0 < extent) ? p[min(g + 0, extent - 1)] : 0.f;
          x.y = (okk && g + 1 < extent) ? p[min(g + 1, extent - 1)] : 0.f;
          x.z = (okk && g + 2 < extent) ? p[min(g + 2, extent - 1)] : 0.f;
          x.w = (okk && g + 3 < extent) ? p[min(g + 3, extent - 1)] : 0.f;
        }
      }
      r[i] = x;
    }
  }

  // registers -> LDS tile [BK][LD] (k-major in LDS; a k-major source is transposed here)
  template <int LD>
  __device__ __forceinline__ void store(float* __restrict__ S) const {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int v = threadIdx.x + 256 * i;
      float4 x = r[i];
      if (VEC && !((okrow & okk) >> i & 1u)) x = make_float4(0.f, 0.f, 0.f, 0.f);   // the select lives here, far from
      if (KMAJ) {                                                                   // the load, so the load stays unconditional
        const int row = v / (BK / 4), kq = v % (BK / 4);
        S[(4 * kq + 0) * LD + row] = x.x;
        S[(4 * kq + 1) * LD + row] = x.y;
        S[(4 * kq + 2) * LD + row] = x.z;
        S[(4 * kq + 3) * LD + row] = x.w;
      } else {
        const int k = v / (BT / 4), m4 = v % (BT / 4);
        *reinterpret_cast<float4*>(&S[k * LD + 4 * m4]) = x;
      }
    }
  }
};

template <int BM, int BN, bool A_KMAJ, bool B_KMAJ, int BK, bool VEC>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const int bx, const int by, const int split) {
  constexpr int LDA_S = BM + (A_KMAJ ? 2 : 4);
  constexpr int LDB_S = BN + (B_KMAJ ? 2 : 4);
  constexpr int WM = BM / 2, WN = BN / 2;        // wave tile
  constexpr int TM = WM / 32, TN = WN / 32;      // 32x32 MFMA tiles per wave
  // one LDS block: the two double-buffered operand tiles during the K loop, the C tile [BM][BN + 4] in the epilogue
  constexpr int LDC_S = BN + 4;
  constexpr int OPER_F = 2 * BK * LDA_S + 2 * BK * LDB_S, CT_F = BM * LDC_S;
  __shared__ __align__(16) float smem_all[OPER_F > CT_F ? OPER_F : CT_F];
  float (*As2)[BK * LDA_S] = reinterpret_cast<float (*)[BK * LDA_S]>(smem_all);
  float (*Bs2)[BK * LDB_S] = reinterpret_cast<float (*)[BK * LDB_S]>(smem_all + 2 * BK * LDA_S);

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = split * g.klen;
  const int kend = min(g.K, kbeg + g.klen);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  OperandLoader<BM, BK, A_KMAJ, VEC> la;
  OperandLoader<BN, BK, B_KMAJ, VEC> lb;
  la.init(m0, g.M, g.lda);
  lb.init(n0, g.N, g.ldb);

  float csum = 0.f;   // column sum of A (bias gradient), thread t < BM owns column t
  const bool do_colsum = g.colsum != nullptr && bx == 0;

  // software pipeline: while K step k is multiplied out of LDS buffer (k & 1), tile k+1 (requested during step k-1)
  // is written to the other buffer in the MIDDLE of the step's MFMA stream and the loads of tile k+2 are issued
  // right behind it, so (i) the LDS writes and their wait for the loads sit under matrix-core work instead of
  // between the last MFMA and the barrier, and (ii) a tile has 1.5 K steps (~2.5 us) to arrive.  Loads are
  // unconditional (clamped addresses; tiles past the end are zeroed by the K mask at store time), which keeps
  // hipcc's vmcnt accounting exact.  One barrier per K step.
  const int kq = lane >> 5, li = lane & 31;
  if (kbeg < kend) {
    la.load(g.A, g.lda, g.M, kbeg, kend, g.K);
    lb.load(g.B, g.ldb, g.N, kbeg, kend, g.K);
    la.template store<LDA_S>(As2[0]);
    lb.template store<LDB_S>(Bs2[0]);
    la.load(g.A, g.lda, g.M, kbeg + BK, kend, g.K);
    lb.load(g.B, g.ldb, g.N, kbeg + BK, kend, g.K);
  }
  __syncthreads();
  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const float* As = As2[cur];
    const float* Bs = Bs2[cur];
    if (do_colsum && t < BM) {
#pragma unroll
      for (int k = 0; k < BK; ++k) csum += As[k * LDA_S + t];
    }
    // fragments of k-pair kk+2 are fetched from LDS before the MFMAs of k-pair kk are issued
    float af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = As[kq * LDA_S + wm * WM + i * 32 + li];
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = Bs[kq * LDB_S + wn * WN + j * 32 + li];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int c = (kk >> 1) & 1, nx = c ^ 1;
      if (kk + 2 < BK) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[nx][i] = As[(kk + 2 + kq) * LDA_S + wm * WM + i * 32 + li];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[nx][j] = Bs[(kk + 2 + kq) * LDB_S + wn * WN + j * 32 + li];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i], bf[c][j], acc[i][j], 0, 0, 0);
      if (kk == BK / 2 - 2) {
        // tile k+1 -> the other buffer (nobody reads it during this step), then request tile k+2
        la.template store<LDA_S>(As2[cur ^ 1]);
        lb.template store<LDB_S>(Bs2[cur ^ 1]);
        la.load(g.A, g.lda, g.M, k0 + 2 * BK, kend, g.K);
        lb.load(g.B, g.ldb, g.N, k0 + 2 * BK, kend, g.K);
      }
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: D[row][col], col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  The accumulators go through LDS
  // so that the tile leaves as full rows of 16-byte stores (BN*4 contiguous bytes per 32 lanes) instead of 64 scalar
  // stores per thread; the last K step's barrier has already retired every read of the operand buffers. ----
  float* Ct = smem_all;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        Ct[row * LDC_S + wn * WN + j * 32 + (lane & 31)] = acc[i][j][r];
      }
  __syncthreads();
  float* Cout = g.C + (size_t)split * g.M * g.N;   // partial slab (splits > 1 only)
  const bool fin = g.splits == 1;
  const int ldo = fin ? g.ldc : g.N;
  float* __restrict__ obase = fin ? g.C : Cout;
  constexpr int QPR = BN / 4;                       // float4 per tile row
  const bool vec_out = (ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(obase) & 15) == 0) && (n0 + BN <= g.N) &&
                       (!fin || g.mask == nullptr || ((reinterpret_cast<uintptr_t>(g.mask) & 15) == 0));
  for (int v = t; v < BM * QPR; v += 256) {
    const int row = v / QPR, q = v - row * QPR;
    const int gm = m0 + row, gn = n0 + 4 * q;
    if (gm >= g.M) continue;
    const float4 c4 = *reinterpret_cast<const float4*>(&Ct[row * LDC_S + 4 * q]);
    float cv[4] = {c4.x, c4.y, c4.z, c4.w};
    if (fin) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (gn + e < g.N) {
          float x = cv[e] + (g.bias != nullptr ? g.bias[gn + e] : 0.f);
          x = apply_act(x, g.act);
          if (g.mask != nullptr && !(g.mask[(size_t)gm * g.ldc + gn + e] > 0.f)) x = 0.f;
          cv[e] = x;
        }
      }
    }
    float* dst = obase + (size_t)gm * ldo + gn;
    if (vec_out) {
      *reinterpret_cast<float4*>(dst) = make_float4(cv[0], cv[1], cv[2], cv[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (gn + e < g.N) dst[e] = cv[e];
    }
  }
  if (do_colsum && t < BM && m0 + t < g.M) {
    if (fin) g.colsum[m0 + t] = csum;
    else g.C[(size_t)g.splits * g.M * g.N + (size_t)split * g.M + m0 + t] = csum;   // partial, behind the slabs
  }
}

template <int BM, int BN, bool A_KMAJ, bool B_KMAJ, int BK, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs g) {
  gemm_body<BM, BN, A_KMAJ, B_KMAJ, BK, VEC>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// several independent small products (same operand layouts, no split-K) in ONE launch: blockIdx.z selects the
// problem, workgroups outside a problem's tile range exit at once.  Used for the three 64->z heads and for all
// the small weight-gradient products of the fully connected layers (8 launches -> 1).
#define AVA_GEMM_GROUP_MAX 8
struct GemmGroup {
  GemmArgs g[AVA_GEMM_GROUP_MAX];
  int tile0[AVA_GEMM_GROUP_MAX + 1];     // skinny grouped launch: first 16x16 tile of each problem in a flat grid
  int n;
};
template <int BM, bool A_KMAJ, bool B_KMAJ, bool VEC>
__global__ __launch_bounds__(256) void gemm_grouped_kernel(const GemmGroup grp) {
  const GemmArgs& g = grp.g[blockIdx.z];
  if ((int)blockIdx.x * BM >= g.N || (int)blockIdx.y * BM >= g.M) return;
  gemm_body<BM, BM, A_KMAJ, B_KMAJ, 16, VEC>(g, blockIdx.x, blockIdx.y, 0);
}

// sum the split-K slabs in a fixed order, add bias, activation
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                     const float* __restrict__ mask, float* __restrict__ C,
                                     float* __restrict__ colsum, int M, int N, int ldc, int splits, int act) {
  const size_t mn = (size_t)M * N;
  if (colsum != nullptr && blockIdx.x == 0) {
    const float* cs = ws + (size_t)splits * mn;
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
      float s = 0.f;
      for (int p = 0; p < splits; ++p) s += cs[(size_t)p * M + m];
      colsum[m] = s;
    }
  }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < mn; i += (size_t)gridDim.x * blockDim.x) {
    // slabs are summed in index order (deterministic); eight loads are put in flight per batch
    float s = 0.f;
    int p = 0;
    for (; p + 8 <= splits; p += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ws[(size_t)(p + u) * mn + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; p < splits; ++p) s += ws[(size_t)p * mn + i];
    const size_t m = i / N, n = i - m * N;
    if (bias != nullptr) s += bias[n];
    s = apply_act(s, act);
    if (mask != nullptr && !(mask[m * ldc + n] > 0.f)) s = 0.f;
    C[m * ldc + n] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Small products (the fc2 .. fc7 chain and its dX chain: <= 0.13 GFLOP each, launch- and latency-bound).  One
// workgroup per 16x16 output tile, no LDS staging and no split-K slabs: the four waves take every fourth 16-wide
// K chunk, read their operand fragments straight from global memory (the matrices are L2 resident; lane (i, kg)
// loads the 4 consecutive k of chunk position 4 kg, one value per MFMA -- the same K permutation on both operands),
// and are summed through LDS in wave order (deterministic).  A is k-major; B is k-major ([N,K], forward) or
// n-major ([K,N], dX = dY W).  v_mfma_f32_16x16x4_f32, exact fp32.
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x4g __attribute__((ext_vector_type(4)));

template <bool A_KMAJ, bool B_KMAJ>
__device__ __forceinline__ void gemm_skinny_body(const GemmArgs& g, const int bx, const int by) {
  __shared__ float red[7][64][4];             // up to 8 waves (long K): waves 1.. hand their tile to wave 0
  __shared__ float cred[8][16];
  const int nw = blockDim.x >> 6;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int i = lane & 15, kg = lane >> 4;
  const int m0 = by * 16, n0 = bx * 16;
  const int am = min(m0 + i, g.M - 1);            // clamped row of A / column of B owned by this lane
  const int bn = min(n0 + i, g.N - 1);
  const float* __restrict__ arow = A_KMAJ ? g.A + (size_t)am * g.lda : g.A + am;
  const float* __restrict__ bcol = B_KMAJ ? g.B + (size_t)bn * g.ldb : g.B + bn;
  const int nchunks = (g.K + 15) >> 4;
  const bool do_colsum = g.colsum != nullptr && bx == 0;     // bias gradient: column sums of A (m-major A only)
  f32x4g acc = {0.f, 0.f, 0.f, 0.f};
  float csum = 0.f;
  // wave 0's epilogue operands (bias, ReLU mask of the producing layer) are requested here, in front of the operand loads:
  // fetched after the reduction barrier they were one more exposed memory latency in a kernel that is three latencies long
  const int gn = n0 + i;
  float bv = 0.f, mk[4] = {1.f, 1.f, 1.f, 1.f};
  if (wave == 0 && gn < g.N) {
    if (g.bias != nullptr) bv = g.bias[gn];
    if (g.mask != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r) mk[r] = g.mask[(size_t)min(m0 + 4 * kg + r, g.M - 1) * g.ldc + gn];
    }
  }
  constexpr int U = 4;                            // chunks in flight per wave (8 measured slower)
  for (int c0 = wave; c0 < nchunks; c0 += nw * U) {
    f32x4g a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = 16 * (c0 + nw * u) + 4 * kg;
      if (A_KMAJ) {                               // K % 4 == 0: a quad is inside or outside as a whole
        const bool ok = k < g.K;
        a[u] = *reinterpret_cast<const f32x4g*>(arow + (ok ? k : 0));
        if (!ok) a[u] = (f32x4g){0.f, 0.f, 0.f, 0.f};
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = arow[(size_t)min(k + j, g.K - 1) * g.lda];
          a[u][j] = k + j < g.K ? v : 0.f;
        }
      }
      if (B_KMAJ) {
        b[u] = *reinterpret_cast<const f32x4g*>(bcol + (k < g.K ? k : 0));      // multiplied by a zero A quad when outside
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[u][j] = bcol[(size_t)min(k + j, g.K - 1) * g.ldb];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], b[u][j], acc, 0, 0, 0);
        csum += a[u][j];
      }
  }
  // D[row = 4 kg + r][col = i] ; waves 1..3 hand their tile to wave 0, summed in wave order
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave - 1][lane][r] = acc[r];
  }
  if (do_colsum) {
    csum += __shfl_xor(csum, 16, 64);
    csum += __shfl_xor(csum, 32, 64);
    if (kg == 0) cred[wave][i] = csum;
  }
  __syncthreads();
  if (wave != 0) return;
  if (do_colsum && kg == 0 && m0 + i < g.M) {
    float cs = cred[0][i];
    for (int w = 1; w < nw; ++w) cs += cred[w][i];
    g.colsum[m0 + i] = cs;
  }
  if (gn >= g.N) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gm = m0 + 4 * kg + r;
    if (gm < g.M) {
      float v = acc[r];
      for (int w = 0; w + 1 < nw; ++w) v += red[w][lane][r];          // wave order: deterministic
      v = apply_act(v + bv, g.act);
      if (!(mk[r] > 0.f)) v = 0.f;
      g.C[(size_t)gm * g.ldc + gn] = v;
    }
  }
}

template <bool A_KMAJ, bool B_KMAJ>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(const GemmArgs g) {
  gemm_skinny_body<A_KMAJ, B_KMAJ>(g, blockIdx.x, blockIdx.y);
}

// products the skinny kernel takes: small output, A k-major, 16-byte alignable operands, no bias-gradient column sums
static bool skinny_ok(const GemmArgs& g, int a_kmajor, int b_kmajor) {
  static const bool on = [] { const char* e = ava_env("AVA_GEMM_SKINNY"); return e == nullptr || atoi(e) != 0; }();
  static const int kmax = [] { const char* e = ava_env("AVA_GEMM_SKINNY_KMAX"); return e ? atoi(e) : 2048; }();
  if (!on) return false;
  if (g.colsum != nullptr && a_kmajor) return false;         // column sums are taken from an m-major A
  if ((size_t)g.M * g.N > 262144 || g.K > kmax) return false;
  if (a_kmajor && (g.K % 4 != 0 || g.lda % 4 != 0 || (reinterpret_cast<uintptr_t>(g.A) & 15) != 0)) return false;
  if (b_kmajor && (g.K % 4 != 0 || g.ldb % 4 != 0 || (reinterpret_cast<uintptr_t>(g.B) & 15) != 0)) return false;
  return true;
}

template <bool A_KMAJ, bool B_KMAJ>
__global__ __launch_bounds__(512) void gemm_skinny_grouped_kernel(const GemmGroup grp) {
  int p = 0;
  while (p + 1 < grp.n && (int)blockIdx.x >= grp.tile0[p + 1]) ++p;        // flat tile index -> (problem, tile)
  const GemmArgs& g = grp.g[p];
  const int tile = blockIdx.x - grp.tile0[p], tx = (g.N + 15) >> 4;
  gemm_skinny_body<A_KMAJ, B_KMAJ>(g, tile % tx, tile / tx);
}

static void plan(int M, int N, int K, int* bm, int* splits, int* klen) {
  *bm = (M >= 128 && N >= 128) ? 128 : 64;
  // small outputs with a short K (fc2, fc3x, fc6, fc7 and their dX): 64x64 tiles give 4x the workgroups per split,
  // so fewer, longer splits and a quarter of the slab traffic (measured -3.5 us per product at batch 256)
  if ((size_t)M * N <= 262144 && K <= 2048) *bm = 64;
  // batch-sized M against a wide N (fc8 forward, fc1 dX: 256 x 8192, K = 1024): 64x64 tiles fill the chip (512
  // workgroups) WITHOUT split-K, so no slabs and no reduce launch, bias / activation / mask in the epilogue
  // (same-box: 59.2 -> 52.4 us and 65.4 -> 57.0 us including the reduce kernel they no longer need)
  const bool wide = M <= 256 && K >= 512 && ceil_div(M, 64) * ceil_div(N, 64) >= 384;
  // ... and against a long K (fc1 forward, fc8 dX: 256 x 1024, K = 8192): 64 tiles x 16 splits (59.1 -> 54.1 us, 16 MB
  // of slabs instead of 32)
  const bool deep = M <= 256 && K >= 4096 && (size_t)M * N <= 262144;
  if (wide || deep) *bm = 64;
  { const char* e = ava_env("AVA_GEMM_BM"); if (e && M <= 256) *bm = atoi(e); }      // lab: tile size for the M = batch shapes
  const int tiles = ceil_div(M, *bm) * ceil_div(N, *bm);
  int s = 1;
  if (tiles < 384) {
    // aim at >= 2 workgroups per CU so that one workgroup's loads overlap another's MFMAs
    s = ceil_div(deep ? 1024 : 512, tiles);
    const int max_s = K / 32 > 0 ? K / 32 : 1;      // at least 2 K-steps per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
  }
  { const char* e = ava_env("AVA_GEMM_SPLITS"); if (e && tiles < 384) { s = atoi(e); if (s > K / 16) s = K / 16; if (s < 1) s = 1; } }
  int kl = ceil_div(ceil_div(K, s), 16) * 16;
  s = ceil_div(K, kl);
  *splits = s;
  *klen = kl;
}

extern "C" size_t ava_gemm_workspace_bytes(int M, int N, int K) {
  int bm, splits, klen;
  plan(M, N, K, &bm, &splits, &klen);
  int lbn, lsplits, lklen;
  for (int ak = 0; ak < 2; ++ak) {                          // whichever kernel ava_gemm picks for the operands it is given
    ava_gemm_limb_plan(M, N, K, ak, &lbn, &lsplits, &lklen);
    if (lsplits > splits) splits = lsplits;
  }
  return splits > 1 ? ((size_t)splits * M * N + (size_t)splits * M) * sizeof(float) : 0;
}

template <int BM, int BK, bool VEC>
static void launch_gemm(const GemmArgs& g, int a_k, int b_k, dim3 grid, hipStream_t st) {
  if (a_k && b_k) hipLaunchKernelGGL((gemm_kernel<BM, BM, true, true, BK, VEC>), grid, dim3(256), 0, st, g);
  else if (a_k && !b_k) hipLaunchKernelGGL((gemm_kernel<BM, BM, true, false, BK, VEC>), grid, dim3(256), 0, st, g);
  else if (!a_k && b_k) hipLaunchKernelGGL((gemm_kernel<BM, BM, false, true, BK, VEC>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((gemm_kernel<BM, BM, false, false, BK, VEC>), grid, dim3(256), 0, st, g);
}

static int gemm_impl(const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc,
                     const float* mask, float* colsum, int M, int N, int K, int a_kmajor, int b_kmajor, int act,
                     void* ws, size_t ws_bytes, ava_stream_t s, int* deferred_slabs);

extern "C" int ava_gemm(const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc,
                        const float* mask, float* colsum, int M, int N, int K, int a_kmajor, int b_kmajor, int act,
                        void* ws, size_t ws_bytes, ava_stream_t s) {
  return gemm_impl(A, lda, B, ldb, bias, C, ldc, mask, colsum, M, N, K, a_kmajor, b_kmajor, act, ws, ws_bytes, s, nullptr);
}

// internal (model.hip): the same product, but when it runs as exactly TWO split-K slabs of the limb kernel the reduce
// launch is left to the consumer -- a layout kernel that reads the product once anyway sums the slabs in the reduce
// kernel's order, adds the bias and applies the activation on the way in (bn.hip: load_nchw_quarter_slabs).  *slabs = 2:
// the slabs are ws[0 .. M*N) and ws[M*N .. 2*M*N), C is untouched; *slabs = 1: C is complete.
int ava_gemm_defer2(const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc,
                    const float* mask, float* colsum, int M, int N, int K, int a_kmajor, int b_kmajor, int act,
                    void* ws, size_t ws_bytes, ava_stream_t s, int* slabs) {
  *slabs = 1;
  return gemm_impl(A, lda, B, ldb, bias, C, ldc, mask, colsum, M, N, K, a_kmajor, b_kmajor, act, ws, ws_bytes, s, slabs);
}

static int gemm_impl(const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc,
                     const float* mask, float* colsum, int M, int N, int K, int a_kmajor, int b_kmajor, int act,
                     void* ws, size_t ws_bytes, ava_stream_t s, int* deferred_slabs) {
  if (A == nullptr || B == nullptr || C == nullptr || M <= 0 || N <= 0 || K <= 0) return AVA_EINVAL;
  int bm, splits, klen;
  plan(M, N, K, &bm, &splits, &klen);
  if (splits > 1 && (ws == nullptr || ws_bytes < ava_gemm_workspace_bytes(M, N, K))) return AVA_EWORKSPACE;
  GemmArgs g;
  g.dbg = 0;
  g.A = A; g.B = B; g.bias = bias; g.colsum = colsum; g.mask = mask;
  g.C = splits > 1 ? reinterpret_cast<float*>(ws) : C;
  g.M = M; g.N = N; g.K = K;
  g.lda = lda > 0 ? lda : (a_kmajor ? K : M);
  g.ldb = ldb > 0 ? ldb : (b_kmajor ? K : N);
  g.ldc = ldc > 0 ? ldc : N;
  g.klen = klen; g.splits = splits; g.act = act;
  // 16-byte loads: leading dimensions / base pointers 16-byte aligned, K % 4 == 0 for k-major operands,
  // M (N) % 4 == 0 for m-major (n-major) ones; anything else takes the scalar-load instantiation
  g.vec_a = (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (a_kmajor ? K % 4 == 0 && K >= 4 : M % 4 == 0 && M >= 4);
  g.vec_b = (g.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (b_kmajor ? K % 4 == 0 && K >= 4 : N % 4 == 0 && N >= 4);
  const bool vec = g.vec_a && g.vec_b;
  hipStream_t st = to_stream(s);
  if (ava_gemm_limb_ok(g, a_kmajor, b_kmajor)) {
    // the fc1 / fc8 products: three-limb bf16 matrix-core kernel (gemm_limb.hip), fp32-faithful
    int lbn;
    ava_gemm_limb_plan(M, N, K, a_kmajor, &lbn, &splits, &klen);
    if (splits > 1 && (ws == nullptr || ws_bytes < ((size_t)splits * M * N + (size_t)splits * M) * sizeof(float)))
      return AVA_EWORKSPACE;
    g.klen = klen; g.splits = splits;
    g.C = splits > 1 ? reinterpret_cast<float*>(ws) : C;
    const int rc = ava_gemm_limb_launch(g, a_kmajor, b_kmajor, lbn, st);
    if (rc != AVA_OK) return rc;
    if (splits == 2 && deferred_slabs != nullptr && mask == nullptr && colsum == nullptr && g.ldc == N) {
      *deferred_slabs = 2;                 // the consumer reduces
      return AVA_OK;
    }
    if (splits > 1) {
      const size_t mn = (size_t)M * N;
      int blocks = (int)((mn + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float*>(ws), bias,
                         mask, C, colsum, M, N, g.ldc, splits, act);
      AVA_CHECK_LAUNCH();
    }
    return AVA_OK;
  }
  if (skinny_ok(g, a_kmajor, b_kmajor)) {
    g.C = C;
    const dim3 sgrid(ceil_div(N, 16), ceil_div(M, 16));
    const dim3 sblock(K >= 512 ? 512 : 256);          // long K: eight waves share the K range of a tile
    if (a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_skinny_kernel<true, true>), sgrid, sblock, 0, st, g);
    else if (a_kmajor) hipLaunchKernelGGL((gemm_skinny_kernel<true, false>), sgrid, sblock, 0, st, g);
    else if (b_kmajor) hipLaunchKernelGGL((gemm_skinny_kernel<false, true>), sgrid, sblock, 0, st, g);
    else hipLaunchKernelGGL((gemm_skinny_kernel<false, false>), sgrid, sblock, 0, st, g);
    AVA_CHECK_LAUNCH();
    return AVA_OK;
  }
  dim3 grid(ceil_div(N, bm), ceil_div(M, bm), splits);
  static int bk32 = -1;
  if (bk32 < 0) { const char* e = ava_env("AVA_GEMM_BK"); bk32 = (e && atoi(e) == 16) ? 0 : 1; }
  if (!vec) {
    if (bm == 128) launch_gemm<128, 16, false>(g, a_kmajor, b_kmajor, grid, st);
    else launch_gemm<64, 16, false>(g, a_kmajor, b_kmajor, grid, st);
  } else if (bm == 128) {
    if (bk32 && klen % 32 == 0) launch_gemm<128, 32, true>(g, a_kmajor, b_kmajor, grid, st);
    else launch_gemm<128, 16, true>(g, a_kmajor, b_kmajor, grid, st);
  } else {
    if (bk32 && klen % 32 == 0 && klen >= 512) launch_gemm<64, 32, true>(g, a_kmajor, b_kmajor, grid, st);   // the long-K 64-tile shapes
    else launch_gemm<64, 16, true>(g, a_kmajor, b_kmajor, grid, st);
  }
  AVA_CHECK_LAUNCH();
  if (splits > 1) {
    const size_t mn = (size_t)M * N;
    int blocks = (int)((mn + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float*>(ws), bias,
                       mask, C, colsum, M, N, g.ldc, splits, act);
    AVA_CHECK_LAUNCH();
  }
  return AVA_OK;
}

// internal (model.hip): n <= 8 products with identical operand layouts in one launch (64x64 tiles, no split-K)
struct AvaGemmProblem {
  const float* A; int lda; const float* B; int ldb; const float* bias; float* C; int ldc; const float* mask;
  float* colsum; int M, N, K; int act;
};
int ava_gemm_grouped(const AvaGemmProblem* p, int n, int a_kmajor, int b_kmajor, hipStream_t st) {
  if (n < 1 || n > AVA_GEMM_GROUP_MAX) return AVA_EINVAL;
  GemmGroup grp;
  bool vec = true;
  int tx = 1, ty = 1;
  for (int i = 0; i < n; ++i) {
    GemmArgs& g = grp.g[i];
    if (p[i].A == nullptr || p[i].B == nullptr || p[i].C == nullptr || p[i].M <= 0 || p[i].N <= 0 || p[i].K <= 0)
      return AVA_EINVAL;
    g.A = p[i].A; g.B = p[i].B; g.bias = p[i].bias; g.C = p[i].C; g.colsum = p[i].colsum; g.mask = p[i].mask;
    g.M = p[i].M; g.N = p[i].N; g.K = p[i].K;
    g.lda = p[i].lda > 0 ? p[i].lda : (a_kmajor ? g.K : g.M);
    g.ldb = p[i].ldb > 0 ? p[i].ldb : (b_kmajor ? g.K : g.N);
    g.ldc = p[i].ldc > 0 ? p[i].ldc : g.N;
    g.klen = ceil_div(g.K, 16) * 16; g.splits = 1; g.act = p[i].act; g.dbg = 0;
    g.vec_a = (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0) &&
              (a_kmajor ? g.K % 4 == 0 && g.K >= 4 : g.M % 4 == 0 && g.M >= 4);
    g.vec_b = (g.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0) &&
              (b_kmajor ? g.K % 4 == 0 && g.K >= 4 : g.N % 4 == 0 && g.N >= 4);
    vec = vec && g.vec_a && g.vec_b;
    tx = tx > ceil_div(g.N, 64) ? tx : ceil_div(g.N, 64);
    ty = ty > ceil_div(g.M, 64) ? ty : ceil_div(g.M, 64);
  }
  bool skinny = true;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    skinny = skinny && skinny_ok(grp.g[i], a_kmajor, b_kmajor);
    grp.tile0[i] = tiles;
    tiles += ceil_div(grp.g[i].N, 16) * ceil_div(grp.g[i].M, 16);
  }
  grp.tile0[n] = tiles;
  grp.n = n;
  if (skinny) {
    const dim3 sgrid(tiles);
    if (a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_skinny_grouped_kernel<true, true>), sgrid, dim3(256), 0, st, grp);
    else if (a_kmajor) hipLaunchKernelGGL((gemm_skinny_grouped_kernel<true, false>), sgrid, dim3(256), 0, st, grp);
    else if (b_kmajor) hipLaunchKernelGGL((gemm_skinny_grouped_kernel<false, true>), sgrid, dim3(256), 0, st, grp);
    else hipLaunchKernelGGL((gemm_skinny_grouped_kernel<false, false>), sgrid, dim3(256), 0, st, grp);
    AVA_CHECK_LAUNCH();
    return AVA_OK;
  }
  const dim3 grid(tx, ty, n);
#define AVA_GG(AK, BK_)                                                                                          \
  if (vec) hipLaunchKernelGGL((gemm_grouped_kernel<64, AK, BK_, true>), grid, dim3(256), 0, st, grp);            \
  else hipLaunchKernelGGL((gemm_grouped_kernel<64, AK, BK_, false>), grid, dim3(256), 0, st, grp);
  if (a_kmajor && b_kmajor) { AVA_GG(true, true) }
  else if (a_kmajor && !b_kmajor) { AVA_GG(true, false) }
  else if (!a_kmajor && b_kmajor) { AVA_GG(false, true) }
  else { AVA_GG(false, false) }
#undef AVA_GG
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}
