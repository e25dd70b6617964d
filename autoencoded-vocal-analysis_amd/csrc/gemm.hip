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
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_EXP = 2 };

struct GemmArgs {
  const float* A;
  const float* B;
  const float* bias;
  float* C;          // output, or partial slabs [splits][M][N]
  float* colsum;
  const float* mask; // optional: C = mask > 0 ? v : 0 (ReLU backward of the producing layer), leading dim ldc
  int M, N, K;
  int lda, ldb, ldc; // leading dimension (elements) of the stored matrices
  int klen;          // K elements per split (multiple of 16)
  int splits;
  int act;
  int vec_a, vec_b;  // 16-byte loads legal
};

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_EXP) return expf(v);   // full-precision expf (vae.py:232)
  return v;
}

template <int BM, int BN, bool A_KMAJ, bool B_KMAJ>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs g) {
  constexpr int BK = 16;
  constexpr int LDA_S = BM + (A_KMAJ ? 2 : 4);
  constexpr int LDB_S = BN + (B_KMAJ ? 2 : 4);
  constexpr int WM = BM / 2, WN = BN / 2;        // wave tile
  constexpr int TM = WM / 32, TN = WN / 32;      // 32x32 MFMA tiles per wave
  constexpr int NA = BM * BK / 4 / 256;          // float4 loads per thread for A (2 or 1)
  constexpr int NB = BN * BK / 4 / 256;
  __shared__ __align__(16) float As2[2][BK * LDA_S];
  __shared__ __align__(16) float Bs2[2][BK * LDB_S];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int split = blockIdx.z;
  const int kbeg = split * g.klen;
  const int kend = min(g.K, kbeg + g.klen);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[NA], rb[NB];

  // ---- global -> register loads of one K tile ------------------------------------------------
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int v = t + i * 256;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (A_KMAJ) {
        const int m = v >> 2, kq = v & 3;
        const int gm = m0 + m, gk = k0 + 4 * kq;
        if (gm < g.M) {
          const float* p = g.A + (size_t)gm * g.lda + gk;
          if (g.vec_a && gk + 3 < kend) x = *reinterpret_cast<const float4*>(p);
          else {
            if (gk + 0 < kend) x.x = p[0];
            if (gk + 1 < kend) x.y = p[1];
            if (gk + 2 < kend) x.z = p[2];
            if (gk + 3 < kend) x.w = p[3];
          }
        }
      } else {
        const int k = v / (BM / 4), m4 = v % (BM / 4);
        const int gk = k0 + k, gm = m0 + 4 * m4;
        if (gk < kend) {
          const float* p = g.A + (size_t)gk * g.lda + gm;
          if (g.vec_a && gm + 3 < g.M) x = *reinterpret_cast<const float4*>(p);
          else {
            if (gm + 0 < g.M) x.x = p[0];
            if (gm + 1 < g.M) x.y = p[1];
            if (gm + 2 < g.M) x.z = p[2];
            if (gm + 3 < g.M) x.w = p[3];
          }
        }
      }
      ra[i] = x;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int v = t + i * 256;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (B_KMAJ) {
        const int n = v >> 2, kq = v & 3;
        const int gn = n0 + n, gk = k0 + 4 * kq;
        if (gn < g.N) {
          const float* p = g.B + (size_t)gn * g.ldb + gk;
          if (g.vec_b && gk + 3 < kend) x = *reinterpret_cast<const float4*>(p);
          else {
            if (gk + 0 < kend) x.x = p[0];
            if (gk + 1 < kend) x.y = p[1];
            if (gk + 2 < kend) x.z = p[2];
            if (gk + 3 < kend) x.w = p[3];
          }
        }
      } else {
        const int k = v / (BN / 4), n4 = v % (BN / 4);
        const int gk = k0 + k, gn = n0 + 4 * n4;
        if (gk < kend) {
          const float* p = g.B + (size_t)gk * g.ldb + gn;
          if (g.vec_b && gn + 3 < g.N) x = *reinterpret_cast<const float4*>(p);
          else {
            if (gn + 0 < g.N) x.x = p[0];
            if (gn + 1 < g.N) x.y = p[1];
            if (gn + 2 < g.N) x.z = p[2];
            if (gn + 3 < g.N) x.w = p[3];
          }
        }
      }
      rb[i] = x;
    }
  };
  auto store_tile = [&](int buf) {
    float* As = As2[buf];
    float* Bs = Bs2[buf];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int v = t + i * 256;
      if (A_KMAJ) {
        const int m = v >> 2, kq = v & 3;
        As[(4 * kq + 0) * LDA_S + m] = ra[i].x;
        As[(4 * kq + 1) * LDA_S + m] = ra[i].y;
        As[(4 * kq + 2) * LDA_S + m] = ra[i].z;
        As[(4 * kq + 3) * LDA_S + m] = ra[i].w;
      } else {
        const int k = v / (BM / 4), m4 = v % (BM / 4);
        *reinterpret_cast<float4*>(&As[k * LDA_S + 4 * m4]) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int v = t + i * 256;
      if (B_KMAJ) {
        const int n = v >> 2, kq = v & 3;
        Bs[(4 * kq + 0) * LDB_S + n] = rb[i].x;
        Bs[(4 * kq + 1) * LDB_S + n] = rb[i].y;
        Bs[(4 * kq + 2) * LDB_S + n] = rb[i].z;
        Bs[(4 * kq + 3) * LDB_S + n] = rb[i].w;
      } else {
        const int k = v / (BN / 4), n4 = v % (BN / 4);
        *reinterpret_cast<float4*>(&Bs[k * LDB_S + 4 * n4]) = rb[i];
      }
    }
  };

  float csum = 0.f;   // column sum of A (bias gradient), thread t < BM owns column t
  const bool do_colsum = g.colsum != nullptr && blockIdx.x == 0;

  // software pipeline: tile k+1 travels global -> registers while tile k is multiplied out of LDS buffer
  // (k & 1); it is written to the other buffer after the MFMAs, one barrier per K step.
  if (kbeg < kend) { load_tile(kbeg); store_tile(0); }
  __syncthreads();
  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) load_tile(k0 + BK);
    const float* As = As2[cur];
    const float* Bs = Bs2[cur];
    if (do_colsum && t < BM) {
#pragma unroll
      for (int k = 0; k < BK; ++k) csum += As[k * LDA_S + t];
    }
    const int kq = lane >> 5, li = lane & 31;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = As[(kk + kq) * LDA_S + wm * WM + i * 32 + li];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Bs[(kk + kq) * LDB_S + wn * WN + j * 32 + li];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: D[row][col], col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -------------
  float* Cout = g.C + (size_t)split * g.M * g.N;   // partial slab (splits > 1 only)
  const bool fin = g.splits == 1;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int gn = n0 + wn * WN + j * 32 + (lane & 31);
      if (gn < g.N) {
        const float bv = (fin && g.bias != nullptr) ? g.bias[gn] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int gm = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (gm < g.M) {
            float v = acc[i][j][r];
            if (fin) {
              v = apply_act(v + bv, g.act);
              if (g.mask != nullptr && !(g.mask[(size_t)gm * g.ldc + gn] > 0.f)) v = 0.f;
              g.C[(size_t)gm * g.ldc + gn] = v;
            } else {
              Cout[(size_t)gm * g.N + gn] = v;
            }
          }
        }
      }
    }
  if (do_colsum && t < BM && m0 + t < g.M) {
    if (fin) g.colsum[m0 + t] = csum;
    else g.C[(size_t)g.splits * g.M * g.N + (size_t)split * g.M + m0 + t] = csum;   // partial, behind the slabs
  }
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
    float s = 0.f;
    for (int p = 0; p < splits; ++p) s += ws[p * mn + i];
    const size_t m = i / N, n = i - m * N;
    if (bias != nullptr) s += bias[n];
    s = apply_act(s, act);
    if (mask != nullptr && !(mask[m * ldc + n] > 0.f)) s = 0.f;
    C[m * ldc + n] = s;
  }
}

static void plan(int M, int N, int K, int* bm, int* splits, int* klen) {
  *bm = (M >= 128 && N >= 128) ? 128 : 64;
  const int tiles = ceil_div(M, *bm) * ceil_div(N, *bm);
  int s = 1;
  if (tiles < 384) {
    // aim at >= 2 workgroups per CU so that one workgroup's loads overlap another's MFMAs
    s = ceil_div(512, tiles);
    const int max_s = K / 32 > 0 ? K / 32 : 1;      // at least 2 K-steps per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
  }
  int kl = ceil_div(ceil_div(K, s), 16) * 16;
  s = ceil_div(K, kl);
  *splits = s;
  *klen = kl;
}

extern "C" size_t ava_gemm_workspace_bytes(int M, int N, int K) {
  int bm, splits, klen;
  plan(M, N, K, &bm, &splits, &klen);
  return splits > 1 ? ((size_t)splits * M * N + (size_t)splits * M) * sizeof(float) : 0;
}

template <int BM>
static void launch_gemm(const GemmArgs& g, int a_k, int b_k, dim3 grid, hipStream_t st) {
  if (a_k && b_k) hipLaunchKernelGGL((gemm_kernel<BM, BM, true, true>), grid, dim3(256), 0, st, g);
  else if (a_k && !b_k) hipLaunchKernelGGL((gemm_kernel<BM, BM, true, false>), grid, dim3(256), 0, st, g);
  else if (!a_k && b_k) hipLaunchKernelGGL((gemm_kernel<BM, BM, false, true>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((gemm_kernel<BM, BM, false, false>), grid, dim3(256), 0, st, g);
}

extern "C" int ava_gemm(const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc,
                        const float* mask, float* colsum, int M, int N, int K, int a_kmajor, int b_kmajor, int act,
                        void* ws, size_t ws_bytes, ava_stream_t s) {
  if (A == nullptr || B == nullptr || C == nullptr || M <= 0 || N <= 0 || K <= 0) return AVA_EINVAL;
  int bm, splits, klen;
  plan(M, N, K, &bm, &splits, &klen);
  if (splits > 1 && (ws == nullptr || ws_bytes < ava_gemm_workspace_bytes(M, N, K))) return AVA_EWORKSPACE;
  GemmArgs g;
  g.A = A; g.B = B; g.bias = bias; g.colsum = colsum; g.mask = mask;
  g.C = splits > 1 ? reinterpret_cast<float*>(ws) : C;
  g.M = M; g.N = N; g.K = K;
  g.lda = lda > 0 ? lda : (a_kmajor ? K : M);
  g.ldb = ldb > 0 ? ldb : (b_kmajor ? K : N);
  g.ldc = ldc > 0 ? ldc : N;
  g.klen = klen; g.splits = splits; g.act = act;
  g.vec_a = (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  g.vec_b = (g.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
  hipStream_t st = to_stream(s);
  dim3 grid(ceil_div(N, bm), ceil_div(M, bm), splits);
  if (bm == 128) launch_gemm<128>(g, a_kmajor, b_kmajor, grid, st);
  else launch_gemm<64>(g, a_kmajor, b_kmajor, grid, st);
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
