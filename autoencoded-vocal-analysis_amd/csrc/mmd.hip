// MMD^2 between two sets of latent means.
//
// Replaces the pure-Python O(n^2) double loops of the reference's `_estimate_mmd2`
// (ava/plotting/mmd_plots.py:255-296), `_estimate_mmd2_linear_time` (:299-312) and the pair distances of
// `estimate_median_sigma` (:450-474).  The reference works on float64 numpy arrays; so do these kernels (fp64 VALU;
// the differences are formed directly, sum_k (x_k - y_k)^2, exactly as the reference does -- no |x|^2 + |y|^2 - 2xy
// expansion, which cancels for near pairs).  Every sum is a fixed-order two-stage reduction (per-workgroup partial,
// then one workgroup over the partials): deterministic, no atomics.
//
// Pairwise kernel: a workgroup owns a 64 x 64 tile of (i, j) pairs; the 64 + 64 latent rows are gathered through the
// index lists into LDS once (row stride z|1 doubles: conflict-free column walks) and each thread accumulates a 4 x 4
// block of squared distances in registers while sweeping the z latent dimensions, then adds exp(A * dist) of the
// pairs that exist (and, for the within-set terms, lie above the diagonal: i < j).
#include "common.h"

#define MMD_T 64

__global__ __launch_bounds__(256) void mmd_pair_kernel(const double* __restrict__ L, int z,
                                                       const int64_t* __restrict__ ia, int na,
                                                       const int64_t* __restrict__ ib, int nb, double A, int sym,
                                                       int tiles_j, double* __restrict__ partials) {
  extern __shared__ __align__(16) double sm[];
  const int zp = z | 1;
  double* xs = sm;                       // [64][zp]
  double* ys = sm + MMD_T * zp;          // [64][zp]
  __shared__ double red[4];
  const int t = threadIdx.x;
  const int ti = blockIdx.x / tiles_j, tj = blockIdx.x - ti * tiles_j;
  if (sym && tj < ti) {                  // below the diagonal: nothing to add
    if (t == 0) partials[blockIdx.x] = 0.0;
    return;
  }
  const int i0 = ti * MMD_T, j0 = tj * MMD_T;
  for (int e = t; e < MMD_T * z; e += 256) {
    const int r = e / z, k = e - r * z;
    const int gi = i0 + r, gj = j0 + r;
    xs[r * zp + k] = gi < na ? L[(size_t)ia[gi] * z + k] : 0.0;
    ys[r * zp + k] = gj < nb ? L[(size_t)ib[gj] * z + k] : 0.0;
  }
  __syncthreads();
  const int ty = t >> 4, tx = t & 15;
  double acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
  for (int k = 0; k < z; ++k) {
    double x[4], y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = xs[(ty + 16 * r) * zp + k];
#pragma unroll
    for (int c = 0; c < 4; ++c) y[c] = ys[(tx + 16 * c) * zp + k];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double d = x[r] - y[c];
        acc[r][c] = fma(d, d, acc[r][c]);
      }
  }
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int gi = i0 + ty + 16 * r, gj = j0 + tx + 16 * c;
      const bool ok = gi < na && gj < nb && (!sym || gi < gj);
      if (ok) s += exp(A * acc[r][c]);
    }
  s = wave_sum_d(s);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  if (t == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0..2] = the three normalised terms of mmd_plots.py:276-295, out[3] = term_1 + term_2 - term_3
__global__ __launch_bounds__(256) void mmd_finalize_kernel(const double* __restrict__ p1, int n_p1,
                                                           const double* __restrict__ p2, int n_p2,
                                                           const double* __restrict__ p3, int n_p3, double c1,
                                                           double c2, double c3, double* __restrict__ out) {
  __shared__ double red[3][4];
  const int t = threadIdx.x;
  double s[3] = {0.0, 0.0, 0.0};
  for (int i = t; i < n_p1; i += 256) s[0] += p1[i];
  for (int i = t; i < n_p2; i += 256) s[1] += p2[i];
  for (int i = t; i < n_p3; i += 256) s[2] += p3[i];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const double r = wave_sum_d(s[q]);
    if ((t & 63) == 0) red[q][t >> 6] = r;
  }
  __syncthreads();
  if (t == 0) {
    const double t1 = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * c1;
    const double t2 = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * c2;
    const double t3 = ((red[2][0] + red[2][1]) + (red[2][2] + red[2][3])) * c3;
    out[0] = t1; out[1] = t2; out[2] = t3;
    out[3] = t1 + t2 - t3;
  }
}

// out[p] = sum_k (L[a[p]][k] - L[b[p]][k])^2   (estimate_median_sigma's sampled pairs, mmd_plots.py:468-471)
__global__ __launch_bounds__(256) void pair_sqdist_kernel(const double* __restrict__ L, int z,
                                                          const int64_t* __restrict__ a,
                                                          const int64_t* __restrict__ b, int n,
                                                          double* __restrict__ out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const double* x = L + (size_t)a[p] * z;
  const double* y = L + (size_t)b[p] * z;
  double s = 0.0;
  for (int k = 0; k < z; ++k) {
    const double d = x[k] - y[k];
    s = fma(d, d, s);
  }
  out[p] = s;
}

// linear-time estimator (mmd_plots.py:299-312): one thread per i < m, h(x1,y1,x2,y2) = k(x1,x2)+k(y1,y2)-k(x1,y2)-k(x2,y1)
__global__ __launch_bounds__(256) void mmd_linear_kernel(const double* __restrict__ L, int z,
                                                         const int64_t* __restrict__ i1,
                                                         const int64_t* __restrict__ i2, int m, double A,
                                                         double* __restrict__ partials) {
  __shared__ double red[4];
  const int t = threadIdx.x;
  double s = 0.0;
  for (int i = blockIdx.x * 256 + t; i < m; i += gridDim.x * 256) {
    const double* x1 = L + (size_t)i1[2 * i] * z;
    const double* y1 = L + (size_t)i2[2 * i] * z;
    const double* x2 = L + (size_t)i1[2 * i + 1] * z;
    const double* y2 = L + (size_t)i2[2 * i + 1] * z;
    double dxx = 0.0, dyy = 0.0, dxy = 0.0, dyx = 0.0;
    for (int k = 0; k < z; ++k) {
      const double a1 = x1[k], b1 = y1[k], a2 = x2[k], b2 = y2[k];
      dxx = fma(a1 - a2, a1 - a2, dxx);
      dyy = fma(b1 - b2, b1 - b2, dyy);
      dxy = fma(a1 - b2, a1 - b2, dxy);
      dyx = fma(a2 - b1, a2 - b1, dyx);
    }
    s += exp(A * dxx) + exp(A * dyy) - exp(A * dxy) - exp(A * dyx);
  }
  s = wave_sum_d(s);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  if (t == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

static inline int mmd_tiles(int n) { return (n + MMD_T - 1) / MMD_T; }

extern "C" size_t ava_mmd2_workspace_bytes(int n1, int n2) {
  if (n1 < 0 || n2 < 0) return 0;
  const size_t t1 = mmd_tiles(n1), t2 = mmd_tiles(n2);
  return (t1 * t1 + t2 * t2 + t1 * t2 + 8) * sizeof(double);
}

static int launch_pairs(const double* L, int z, const int64_t* ia, int na, const int64_t* ib, int nb, double A, int sym,
                        double* partials, hipStream_t st) {
  const int ti = mmd_tiles(na), tj = mmd_tiles(nb);
  if (ti * tj == 0) return AVA_OK;
  const size_t lds = (size_t)2 * MMD_T * (z | 1) * sizeof(double);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mmd_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)((size_t)2 * MMD_T * 129 * sizeof(double))) != hipSuccess)
      return AVA_ELAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(mmd_pair_kernel, dim3(ti * tj), dim3(256), lds, st, L, z, ia, na, ib, nb, A, sym, tj, partials);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

extern "C" int ava_mmd2(const double* latent, int z, const int64_t* i1, int n1, const int64_t* i2, int n2, double sigma,
                        double* out4, void* ws, size_t ws_bytes, ava_stream_t s) {
  if (latent == nullptr || i1 == nullptr || i2 == nullptr || out4 == nullptr || ws == nullptr || z < 1 || z > 128 ||
      n1 < 2 || n2 < 2 || !(sigma > 0.0))
    return AVA_EINVAL;
  if (ws_bytes < ava_mmd2_workspace_bytes(n1, n2)) return AVA_EWORKSPACE;
  hipStream_t st = to_stream(s);
  const double A = -0.5 / (sigma * sigma);
  const int t1 = mmd_tiles(n1), t2 = mmd_tiles(n2);
  double* p1 = reinterpret_cast<double*>(ws);
  double* p2 = p1 + (size_t)t1 * t1;
  double* p3 = p2 + (size_t)t2 * t2;
  int rc = launch_pairs(latent, z, i1, n1, i1, n1, A, 1, p1, st);
  if (rc == AVA_OK) rc = launch_pairs(latent, z, i2, n2, i2, n2, A, 1, p2, st);
  if (rc == AVA_OK) rc = launch_pairs(latent, z, i1, n1, i2, n2, A, 0, p3, st);
  if (rc != AVA_OK) return rc;
  hipLaunchKernelGGL(mmd_finalize_kernel, dim3(1), dim3(256), 0, st, p1, t1 * t1, p2, t2 * t2, p3, t1 * t2,
                     2.0 / ((double)n1 * (n1 - 1)), 2.0 / ((double)n2 * (n2 - 1)), 2.0 / ((double)n1 * n2), out4);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

extern "C" int ava_mmd2_linear(const double* latent, int z, const int64_t* i1, const int64_t* i2, int m, double sigma,
                               double* out, void* ws, size_t ws_bytes, ava_stream_t s) {
  if (latent == nullptr || i1 == nullptr || i2 == nullptr || out == nullptr || ws == nullptr || z < 1 || m < 1 ||
      !(sigma > 0.0))
    return AVA_EINVAL;
  int grid = (m + 255) / 256;
  if (grid > 1024) grid = 1024;
  if (ws_bytes < (size_t)(grid + 8) * sizeof(double)) return AVA_EWORKSPACE;
  hipStream_t st = to_stream(s);
  double* p = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL(mmd_linear_kernel, dim3(grid), dim3(256), 0, st, latent, z, i1, i2, m, -0.5 / (sigma * sigma), p);
  AVA_CHECK_LAUNCH();
  // term / m through the same finalise kernel: out[0] = sum / m (out[1..3] scratch)
  hipLaunchKernelGGL(mmd_finalize_kernel, dim3(1), dim3(256), 0, st, p, grid, p, 0, p, 0, 1.0 / (double)m, 0.0, 0.0, p + grid);
  AVA_CHECK_LAUNCH();
  if (hipMemcpyAsync(out, p + grid, sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) return AVA_ELAUNCH;
  return AVA_OK;
}

extern "C" int ava_pair_sqdist(const double* latent, int z, const int64_t* a, const int64_t* b, int n, double* out,
                               ava_stream_t s) {
  if (latent == nullptr || a == nullptr || b == nullptr || out == nullptr || z < 1 || n < 1) return AVA_EINVAL;
  hipLaunchKernelGGL(pair_sqdist_kernel, dim3((n + 255) / 256), dim3(256), 0, to_stream(s), latent, z, a, b, n, out);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}
