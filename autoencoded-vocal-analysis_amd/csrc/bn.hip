// BatchNorm2d statistics / finalisation (forward and backward) and the NHWC <-> NCHW-flatten
// hand-offs between the convolutional and fully connected parts of the VAE.
//
// Replaces ATen's batch_norm / batch_norm_backward (nn.BatchNorm2d, ava/models/vae.py:135-141,
// 162-168) -- the *apply* part of BatchNorm lives in the prologue of the consuming convolution
// (conv.hip), the per-channel sums in the epilogue of the producing kernel; what remains here are
// the tiny per-channel reductions between them.  Per-workgroup partial sums are fp32, everything
// across workgroups is accumulated in fp64 in a fixed order (deterministic, and as accurate as the
// reference's CPU kernels, which accumulate float tensors in double).
#include <string.h>
#include "common.h"
#include "bn_fuse.h"
#include "bn_acc.h"

#define BN_EPS AVA_BN_EPS
#define BN_MOMENTUM AVA_BN_MOMENTUM

int ava_bn_finalize_bwd_ex(const float* partials, int nparts, int64_t n, int C, const float* gamma, const float* mean,
                           const float* invstd, float* dgamma, float* dbeta, float* A, float* Bc, float* Cc, int eval,
                           hipStream_t st);

// ---- statistics of a raw tensor x[n][C] (used for bn1, whose input has no producer kernel) --------
template <int C>
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int64_t n,
                                                       float* __restrict__ partials) {
  __shared__ float red[4][2 * C];
  float s1[C], s2[C];
#pragma unroll
  for (int c = 0; c < C; ++c) s1[c] = s2[c] = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (C == 1) {
    const int64_t n4 = n / 4;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      const float4 v = x4[i];
      s1[0] += (v.x + v.y) + (v.z + v.w);
      s2[0] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
      for (int64_t i = n4 * 4; i < n; ++i) { s1[0] += x[i]; s2[0] += x[i] * x[i]; }
  } else {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float v = x[p * C + c];
        s1[c] += v;
        s2[c] = fmaf(v, v, s2[c]);
      }
    }
  }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float r1 = wave_sum(s1[c]), r2 = wave_sum(s2[c]);
    if (l == 0) { red[w][c] = r1; red[w][C + c] = r2; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * C)
    partials[(size_t)blockIdx.x * 2 * C + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// column sums of partials[nparts][ncols] in fp64; result in sums[ncols] (shared memory, double).
// 8 independent loads are kept in flight per thread: the loop is latency- not bandwidth-bound.
__device__ void column_sums(const float* __restrict__ partials, int nparts, int ncols, double* sums /*[ncols]*/,
                            double* scratch /*[1024]*/) {
  const int t = threadIdx.x;
  const int groups = 1024 / ncols;
  const int col = t % ncols, grp = t / ncols;
  double s = 0.0;
  if (grp < groups) {
    const float* p = partials + col;
    int r = grp;
    for (; r + 7 * groups < nparts; r += 8 * groups) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(r + k * groups) * ncols];
      s += (((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3])) +
           (((double)v[4] + (double)v[5]) + ((double)v[6] + (double)v[7]));
    }
    for (; r < nparts; r += groups) s += (double)p[(size_t)r * ncols];
  }
  scratch[t] = grp < groups ? s : 0.0;
  __syncthreads();
  // Sum the per-group results of every column with a binary tree over the groups (fixed shape -> deterministic):
  // ceil(log2(groups)) rounds of one fp64 add per thread instead of one thread per column walking all its groups
  // (64 dependent LDS reads + adds for the 8-channel layers: 1.6 us of a 5 us kernel).
  int cnt = groups;
  while (cnt > 1) {
    const int half = (cnt + 1) >> 1;
    if (grp < cnt - half) scratch[t] += scratch[t + half * ncols];     // t = grp * ncols + col
    __syncthreads();
    cnt = half;
  }
  if (t < ncols) sums[t] = scratch[t];
  __syncthreads();
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partials, int nparts, double n,
                                                           int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* running_mean,
                                                           float* running_var, int64_t* num_batches, int train,
                                                           float* mean_o, float* invstd_o, float* scale_o,
                                                           float* shift_o) {
  __shared__ double sums[64];
  __shared__ double scratch[1024];
  if (train) column_sums(partials, nparts, 2 * C, sums, scratch);
  const int c = threadIdx.x;
  if (c < C) {
    double mean, var;
    if (train) {
      mean = sums[c] / n;
      var = sums[C + c] / n - mean * mean;      // biased variance
      if (var < 0.0) var = 0.0;
      if (running_mean != nullptr) {
        const double unb = n > 1.0 ? var * (n / (n - 1.0)) : var;
        running_mean[c] = (float)((1.0 - BN_MOMENTUM) * (double)running_mean[c] + BN_MOMENTUM * mean);
        running_var[c] = (float)((1.0 - BN_MOMENTUM) * (double)running_var[c] + BN_MOMENTUM * unb);
      }
    } else {
      mean = (double)running_mean[c];
      var = (double)running_var[c];
    }
    const float meanf = (float)mean;
    const float invstd = (float)(1.0 / sqrt(var + BN_EPS));
    const float sc = gamma[c] * invstd;
    mean_o[c] = meanf;
    invstd_o[c] = invstd;
    scale_o[c] = sc;
    shift_o[c] = beta[c] - meanf * sc;
  }
  if (train && threadIdx.x == 0 && num_batches != nullptr) *num_batches += 1;
}

__global__ __launch_bounds__(1024) void bn_finalize_bwd_kernel(const float* __restrict__ partials, int nparts,
                                                               double n, int C, const float* __restrict__ gamma,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, float* dgamma,
                                                               float* dbeta, float* A, float* Bc, float* Cc, int eval) {
  __shared__ double sums[64];
  __shared__ double scratch[1024];
  column_sums(partials, nparts, 2 * C, sums, scratch);
  const int c = threadIdx.x;
  if (c < C) {
    const double dB = sums[c];            // sum g
    const double dG = sums[C + c];        // sum g * xhat
    const double is = (double)invstd[c], gm = (double)gamma[c], mu = (double)mean[c];
    const double a = gm * is;
    // eval: the forward normalised with the (constant) running statistics -> dx = gamma*invstd*g, nothing else
    const double b = eval ? 0.0 : -gm * is * is * dG / n;
    dgamma[c] = (float)dG;
    dbeta[c] = (float)dB;
    A[c] = (float)a;
    Bc[c] = (float)b;
    Cc[c] = eval ? 0.f : (float)(-a * dB / n - b * mu);
  }
}

// ---- NCHW-flatten <-> NHWC hand-offs (per sample: 32 channels x P pixels, P = (H/8)*(W/8) = 256 at 128x128) ------
// One workgroup handles a slab of 32 channels x 64 pixels (2048 floats) through a padded LDS tile, so (P/64)*B
// workgroups keep every CU busy (one workgroup per sample left 3/4 of the waves idle: 20 us -> ~5 us).
// NCHW element (c, p) of sample b: b*32*P + c*P + p ; NHWC element: b*32*P + p*32 + c.  `nq` = P / 64 slabs per sample.
#define QPIX 64
// scalar store / round-trip of an activation element (float or bfloat16 bits in an unsigned short; conv_common.h has
// the vector forms used by the convolution kernels)
template <typename T> __device__ __forceinline__ float ava_stored_bn(float v);
template <> __device__ __forceinline__ float ava_stored_bn<float>(float v) { return v; }
template <> __device__ __forceinline__ float ava_stored_bn<unsigned short>(float v) { return (float)(__bf16)v; }
template <typename T> __device__ __forceinline__ void ava_store_bn(T* p, float v);
template <> __device__ __forceinline__ void ava_store_bn<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void ava_store_bn<unsigned short>(unsigned short* p, float v) {
  const __bf16 b = (__bf16)v;
  *p = *reinterpret_cast<const unsigned short*>(&b);
}
__device__ __forceinline__ void load_nchw_quarter(float (*tile)[QPIX + 1], const float* __restrict__ src, int b, int q, int P) {
  for (int i = threadIdx.x; i < 32 * QPIX; i += 256) {
    const int c = i / QPIX, p = i % QPIX;                    // 64 consecutive pixels of one channel: coalesced
    tile[c][p] = src[(size_t)b * 32 * P + (size_t)c * P + q * QPIX + p];
  }
}
// the same tile from a product left as two split-K slabs (gemm.hip: ava_gemm_defer2): summed as splitk_reduce_kernel sums
// them (0 + slab0 + slab1, + bias, activation), and written back to `full` (the NCHW tensor the reduce would have produced)
// when a later kernel reads it
__device__ __forceinline__ void load_nchw_quarter_slabs(float (*tile)[QPIX + 1], const float* __restrict__ s0,
                                                        const float* __restrict__ s1, const float* __restrict__ bias, int relu,
                                                        float* __restrict__ full, int b, int q, int P) {
  for (int i = threadIdx.x; i < 32 * QPIX; i += 256) {
    const int c = i / QPIX, p = i % QPIX;
    const size_t n = (size_t)c * P + q * QPIX + p, o = (size_t)b * 32 * P + n;
    float s = 0.f;
    s += s0[o];
    s += s1[o];
    if (bias != nullptr) s += bias[n];
    if (relu) s = fmaxf(s, 0.f);
    if (full != nullptr) full[o] = s;
    tile[c][p] = s;
  }
}
__device__ __forceinline__ void load_nhwc_quarter(float (*tile)[QPIX + 1], const float* __restrict__ src, int b, int q, int P) {
  for (int i = threadIdx.x; i < 32 * QPIX; i += 256) {
    const int p = i >> 5, c = i & 31;                        // the slab is one contiguous 8 KB range
    tile[c][p] = src[(size_t)b * 32 * P + (size_t)(q * QPIX) * 32 + i];
  }
}

// f8 [B][32*P] (c*P+p) -> out [B][P][32] (an ACTIVATION: stored as ACT), plus per-channel {sum, sum^2} partials for bn8
// SLABS: `in` is slab 0 of a deferred two-slab product, slab1 / bias / relu complete it and `full` receives the reduced tensor
template <typename ACT, bool SLABS = false>
__global__ __launch_bounds__(256) void nchw_to_nhwc_stats_kernel(const float* __restrict__ in, float* __restrict__ out_,
                                                                 float* __restrict__ partials, int B, int P,
                                                                 long long* acc_out, const float* __restrict__ slab1 = nullptr,
                                                                 const float* __restrict__ bias = nullptr, int relu = 0,
                                                                 float* __restrict__ full = nullptr) {
  ACT* __restrict__ out = reinterpret_cast<ACT*>(out_);
  __shared__ float tile[32][QPIX + 1];
  __shared__ float red[8][64];
  const int t = threadIdx.x;
  float s1 = 0.f, s2 = 0.f;                 // thread -> channel t&31, 8 threads per channel
  const int nq = P / QPIX;
  for (int w = blockIdx.x; w < nq * B; w += gridDim.x) {
    const int b = w / nq, q = w - b * nq;
    __syncthreads();
    if constexpr (SLABS) load_nchw_quarter_slabs(tile, in, slab1, bias, relu, full, b, q, P);
    else load_nchw_quarter(tile, in, b, q, P);
    __syncthreads();
    for (int i = t; i < 32 * QPIX; i += 256) {
      const int p = i >> 5, c = i & 31;     // c == t & 31 for every i of this thread
      const float v = ava_stored_bn<ACT>(tile[c][p]);     // statistics of what is stored
      ava_store_bn<ACT>(out + (size_t)b * 32 * P + (size_t)(q * QPIX) * 32 + i, v);
      s1 += v;
      s2 = fmaf(v, v, s2);
    }
  }
  red[t >> 5][t & 31] = s1;
  red[t >> 5][32 + (t & 31)] = s2;
  __syncthreads();
  if (t < 64) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][t];
    if (acc_out != nullptr) bn_acc_add(acc_out, t, s);       // t = which * 32 + channel: accumulated for the consumer (bn_acc.h)
    else partials[(size_t)blockIdx.x * 64 + t] = s;
  }
}

// y7 [B][256][32] -> out [B][32*256]
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int P) {
  __shared__ float tile[32][QPIX + 1];
  const int nq = P / QPIX;
  for (int w = blockIdx.x; w < nq * B; w += gridDim.x) {
    const int b = w / nq, q = w - b * nq;
    __syncthreads();
    load_nhwc_quarter(tile, in, b, q, P);
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * QPIX; i += 256) {
      const int c = i / QPIX, p = i % QPIX;
      out[(size_t)b * 32 * P + (size_t)c * P + q * QPIX + p] = tile[c][p];
    }
  }
}

// dU7 (NHWC) = (y7 > 0) ? dy7 (NCHW-flatten, from fc1's backward) : 0       (ReLU of vae.py:223)
// slab1 != null: dy is slab 0 of a deferred two-slab product (fc1's dX), summed on the way in
__global__ __launch_bounds__(256) void relu_mask_to_nhwc_kernel(const float* __restrict__ dy_nchw,
                                                                const float* __restrict__ y_nhwc,
                                                                float* __restrict__ du_nhwc, int B, int P,
                                                                const float* __restrict__ slab1) {
  __shared__ float tile[32][QPIX + 1];
  const int nq = P / QPIX;
  for (int w = blockIdx.x; w < nq * B; w += gridDim.x) {
    const int b = w / nq, q = w - b * nq;
    __syncthreads();
    // the mask operand is requested together with the tile (behind the barrier it was one more dependent round trip)
    constexpr int NE = 32 * QPIX / 256;
    float yv[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) yv[k] = y_nhwc[(size_t)b * 32 * P + (size_t)(q * QPIX) * 32 + threadIdx.x + 256 * k];
    if (slab1 != nullptr) load_nchw_quarter_slabs(tile, dy_nchw, slab1, nullptr, 0, nullptr, b, q, P);
    else load_nchw_quarter(tile, dy_nchw, b, q, P);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int i = threadIdx.x + 256 * k;
      const size_t o = (size_t)b * 32 * P + (size_t)(q * QPIX) * 32 + i;
      du_nhwc[o] = yv[k] > 0.f ? tile[i & 31][i >> 5] : 0.f;
    }
  }
}

// dF8 (NCHW-flatten) = (f8 > 0) ? A[c]*g + Bc[c]*f8 + Cc[c] : 0 ; g = dXhat8 (NHWC)   (bn8 backward + ReLU of fc8)
// ACT: bn8's input is the STORED copy of f8 (X[7]); with bf16 storage the Bc*x term must use the rounded value
template <typename ACT>
__global__ __launch_bounds__(256) void bn_bwd_apply_to_nchw_kernel(const float* __restrict__ g_nhwc,
                                                                   const float* __restrict__ f8_nchw,
                                                                   const float* __restrict__ A,
                                                                   const float* __restrict__ Bc,
                                                                   const float* __restrict__ Cc,
                                                                   float* __restrict__ out_nchw, int B, int P,
                                                                   const BnFin fin) {
  __shared__ float tile[32][QPIX + 1];
  __shared__ float coef[96];
  __shared__ double accvals[64];
  constexpr int NE = 32 * QPIX / 256;       // elements per thread and tile
  const int nq = P / QPIX;
  // The first tile's operands (the launch has one tile per workgroup at batch 256) are requested in front of the coefficient
  // finalisation: behind it the kernel was five dependent round trips long (counters, parameters, g, f8, store).
  float gv[NE], fv[NE];
  {
    const int w = blockIdx.x < nq * B ? blockIdx.x : 0, b = w / nq, q = w - b * nq;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int i = threadIdx.x + 256 * k, c = i / QPIX, p = i % QPIX;
      gv[k] = g_nhwc[(size_t)b * 32 * P + (size_t)(q * QPIX) * 32 + i];                 // load_nhwc_quarter's element i
      fv[k] = f8_nchw[(size_t)b * 32 * P + (size_t)c * P + q * QPIX + p];
    }
  }
  if (fin.acc != nullptr) {                 // bn8's A, Bc, Cc from the sums convt1's data-gradient kernel accumulated (bn_acc.h)
    bn_coef_from_acc(coef, accvals, fin, 0);
  } else {
    if (threadIdx.x < 32) { coef[threadIdx.x] = A[threadIdx.x]; coef[32 + threadIdx.x] = Bc[threadIdx.x]; coef[64 + threadIdx.x] = Cc[threadIdx.x]; }
    __syncthreads();
  }
  bool first = true;
  for (int w = blockIdx.x; w < nq * B; w += gridDim.x, first = false) {
    const int b = w / nq, q = w - b * nq;
    __syncthreads();
    if (first) {
#pragma unroll
      for (int k = 0; k < NE; ++k) { const int i = threadIdx.x + 256 * k; tile[i & 31][i >> 5] = gv[k]; }
    } else {
      load_nhwc_quarter(tile, g_nhwc, b, q, P);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int i = threadIdx.x + 256 * k, c = i / QPIX, p = i % QPIX;
      const size_t o = (size_t)b * 32 * P + (size_t)c * P + q * QPIX + p;
      const float f = ava_stored_bn<ACT>(first ? fv[k] : f8_nchw[o]);
      const float kA = coef[c], kB = coef[32 + c], kC = coef[64 + c];           // read unconditionally (not inside the select)
      const float du = fmaf(kA, tile[c][p], fmaf(kB, f, kC));
      out_nchw[o] = f > 0.f ? du : 0.f;
    }
  }
}

// eval mode (module.eval()): scale/shift of all 14 BatchNorm layers from their running statistics, one launch
struct BnEvalTable { const float* gamma[14]; const float* beta[14]; int C[14]; };
__global__ void bn_eval_all_kernel(const BnEvalTable tab, const float* __restrict__ running, float* __restrict__ save) {
  const int l = blockIdx.x, c = threadIdx.x;
  if (c < tab.C[l]) {
    const float mean = running[l * 32 + c];
    const float var = running[(14 + l) * 32 + c];
    const float invstd = (float)(1.0 / sqrt((double)var + BN_EPS));
    const float sc = tab.gamma[l][c] * invstd;
    float* o = save + l * 4 * 32;
    o[c] = mean; o[32 + c] = invstd; o[64 + c] = sc; o[96 + c] = tab.beta[l][c] - mean * sc;
  }
}
int ava_bn_eval_all(const float* const* gamma, const float* const* beta, const int* C, const float* running, float* save,
                    hipStream_t st) {
  BnEvalTable tab;
  for (int l = 0; l < 14; ++l) { tab.gamma[l] = gamma[l]; tab.beta[l] = beta[l]; tab.C[l] = C[l]; }
  hipLaunchKernelGGL(bn_eval_all_kernel, dim3(14), dim3(32), 0, st, tab, running, save);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

extern "C" int ava_bn_stats(const float* x, int64_t n, int C, float* partials, int* nparts, ava_stream_t s) {
  if (x == nullptr || partials == nullptr || n <= 0) return AVA_EINVAL;
  int64_t work = C == 1 ? n / 4 : n;
  int grid = (int)((work + 256 * 8 - 1) / (256 * 8));
  if (grid < 1) grid = 1;
  if (grid > 1024) grid = 1024;
  hipStream_t st = to_stream(s);
  switch (C) {
    case 1: hipLaunchKernelGGL(bn_stats_kernel<1>, dim3(grid), dim3(256), 0, st, x, n, partials); break;
    case 8: hipLaunchKernelGGL(bn_stats_kernel<8>, dim3(grid), dim3(256), 0, st, x, n, partials); break;
    case 16: hipLaunchKernelGGL(bn_stats_kernel<16>, dim3(grid), dim3(256), 0, st, x, n, partials); break;
    case 24: hipLaunchKernelGGL(bn_stats_kernel<24>, dim3(grid), dim3(256), 0, st, x, n, partials); break;
    case 32: hipLaunchKernelGGL(bn_stats_kernel<32>, dim3(grid), dim3(256), 0, st, x, n, partials); break;
    default: return AVA_EINVAL;
  }
  AVA_CHECK_LAUNCH();
  if (nparts != nullptr) *nparts = grid;
  return AVA_OK;
}

extern "C" int ava_bn_finalize(const float* partials, int nparts, int64_t n, int C, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, int64_t* num_batches,
                               int train, float* mean, float* invstd, float* scale, float* shift, ava_stream_t s) {
  if (C < 1 || C > 32 || gamma == nullptr || beta == nullptr || mean == nullptr) return AVA_EINVAL;
  if (train && partials == nullptr) return AVA_EINVAL;
  if (!train && (running_mean == nullptr || running_var == nullptr)) return AVA_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(1024), 0, to_stream(s), partials, nparts, (double)n, C, gamma,
                     beta, running_mean, running_var, num_batches, train, mean, invstd, scale, shift);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

extern "C" int ava_bn_finalize_bwd(const float* partials, int nparts, int64_t n, int C, const float* gamma,
                                   const float* mean, const float* invstd, float* dgamma, float* dbeta, float* A,
                                   float* Bc, float* Cc, ava_stream_t s) {
  return ava_bn_finalize_bwd_ex(partials, nparts, n, C, gamma, mean, invstd, dgamma, dbeta, A, Bc, Cc, 0, to_stream(s));
}
// eval = 1: backward of a BatchNorm that ran on its running statistics (module.eval())
int ava_bn_finalize_bwd_ex(const float* partials, int nparts, int64_t n, int C, const float* gamma, const float* mean,
                           const float* invstd, float* dgamma, float* dbeta, float* A, float* Bc, float* Cc, int eval,
                           hipStream_t st) {
  if (C < 1 || C > 32 || partials == nullptr || gamma == nullptr) return AVA_EINVAL;
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(1), dim3(1024), 0, st, partials, nparts, (double)n, C,
                     gamma, mean, invstd, dgamma, dbeta, A, Bc, Cc, eval);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

// internal (model.hip)
// the same from a product left as two split-K slabs: slab0 + slab1 + bias, ReLU; `full` receives the reduced NCHW tensor
int ava_nchw_to_nhwc_stats_slabs(const float* slab0, const float* slab1, const float* bias, int relu, float* full, float* out,
                                 float* partials, int B, int P, int act_bf16, long long* acc_out, int* nparts, hipStream_t st) {
  if (P < QPIX || P % QPIX != 0) return AVA_EINVAL;
  const int nw = (P / QPIX) * B;
  const int grid = nw < 1024 ? nw : 1024;
  if (act_bf16) hipLaunchKernelGGL((nchw_to_nhwc_stats_kernel<unsigned short, true>), dim3(grid), dim3(256), 0, st, slab0, out, partials, B, P, acc_out, slab1, bias, relu, full);
  else hipLaunchKernelGGL((nchw_to_nhwc_stats_kernel<float, true>), dim3(grid), dim3(256), 0, st, slab0, out, partials, B, P, acc_out, slab1, bias, relu, full);
  AVA_CHECK_LAUNCH();
  *nparts = grid;
  return AVA_OK;
}
int ava_nchw_to_nhwc_stats(const float* in, float* out, float* partials, int B, int P, int act_bf16, long long* acc_out,
                           int* nparts, hipStream_t st) {
  if (P < QPIX || P % QPIX != 0) return AVA_EINVAL;
  const int nw = (P / QPIX) * B;
  const int grid = nw < 1024 ? nw : 1024;
  if (act_bf16) hipLaunchKernelGGL(nchw_to_nhwc_stats_kernel<unsigned short>, dim3(grid), dim3(256), 0, st, in, out, partials, B, P, acc_out);
  else hipLaunchKernelGGL(nchw_to_nhwc_stats_kernel<float>, dim3(grid), dim3(256), 0, st, in, out, partials, B, P, acc_out);
  AVA_CHECK_LAUNCH();
  *nparts = grid;
  return AVA_OK;
}
static inline int layout_grid(int B, int P) { const int nw = (P / QPIX) * B; return nw < 2048 ? nw : 2048; }
int ava_nhwc_to_nchw(const float* in, float* out, int B, int P, hipStream_t st) {
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(layout_grid(B, P)), dim3(256), 0, st, in, out, B, P);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}
int ava_relu_mask_to_nhwc(const float* dy_nchw, const float* slab1, const float* y_nhwc, float* du, int B, int P, hipStream_t st) {
  hipLaunchKernelGGL(relu_mask_to_nhwc_kernel, dim3(layout_grid(B, P)), dim3(256), 0, st, dy_nchw, y_nhwc, du, B, P, slab1);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}
int ava_bn_bwd_apply_to_nchw(const float* g, const float* f8, const float* A, const float* Bc, const float* Cc,
                             float* out, int B, int P, int act_bf16, const BnFin* fin, hipStream_t st) {
  BnFin f = {};
  if (fin != nullptr) f = *fin;
  if (act_bf16) hipLaunchKernelGGL(bn_bwd_apply_to_nchw_kernel<unsigned short>, dim3(layout_grid(B, P)), dim3(256), 0, st, g, f8, A, Bc, Cc, out, B, P, f);
  else hipLaunchKernelGGL(bn_bwd_apply_to_nchw_kernel<float>, dim3(layout_grid(B, P)), dim3(256), 0, st, g, f8, A, Bc, Cc, out, B, P, f);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}
