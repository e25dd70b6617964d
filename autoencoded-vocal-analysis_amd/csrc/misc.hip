// Latent block (reparameterisation + entropy + prior), ELBO assembly, Adam and the counter RNG.
//
//  * latent_fwd/bwd replace torch.exp (vae.py:232), LowRankMultivariateNormal.__init__/rsample/entropy
//    (vae.py:312-313,323; torch/distributions/lowrank_multivariate_normal.py:17-38,98-139,214-252)
//    and their autograd; rank 1, so the "capacitance" matrix is the scalar K = 1 + sum u^2/d.
//  * elbo_finalize assembles vae.py:316-323 from the per-sample / per-workgroup partial sums.
//  * adam_flat is torch.optim.Adam's single-tensor update (torch/optim/adam.py:414-547) over the
//    whole flat parameter arena in one launch.
#include "common.h"
#include "bn_acc.h"

#define LOG_2PI 1.8378770664093453

int ava_latent_bwd_scaled(const float* z, const float* dz_dec, const float* u, const float* d, const float* eps_w,
                          const float* eps_d, float* dmu, float* du, float* dlogd, int B, int zdim, const float* scale,
                          hipStream_t st);
int ava_adam_flat_guarded(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                          double eps, int step, const int* skip_if_set, hipStream_t st);

// one wave per sample; lane j handles latent dims j, j+64
__global__ __launch_bounds__(256) void latent_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ u,
                                                         const float* __restrict__ logd,
                                                         const float* __restrict__ eps_w,
                                                         const float* __restrict__ eps_d, float* __restrict__ d_out,
                                                         float* __restrict__ z_out, float* __restrict__ sums,
                                                         int* __restrict__ status, int B, int zdim) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float ew = eps_w[b];
  float sz2 = 0.f, su2d = 0.f, slogd = 0.f;
  bool bad = false;
  for (int j = lane; j < zdim; j += 64) {
    const size_t i = (size_t)b * zdim + j;
    const float a = logd[i];
    const float d = expf(a);
    const float uu = u[i];
    const float z = mu[i] + uu * ew + sqrtf(d) * eps_d[i];
    d_out[i] = d;
    z_out[i] = z;
    sz2 = fmaf(z, z, sz2);
    su2d += uu * uu / d;
    slogd += logf(d);
    bad |= !(d > 0.f);
  }
  sz2 = wave_sum(sz2);
  su2d = wave_sum(su2d);
  slogd = wave_sum(slogd);
  if (bad && status != nullptr) atomicOr(status, 1);
  if (lane == 0) {
    const float K = 1.f + su2d;
    sums[2 * b] = sz2;
    sums[2 * b + 1] = 0.5f * ((float)(zdim * (1.0 + LOG_2PI)) + logf(K) + slogd);
  }
}

// g = z + dz_dec ; dmu = g ; du = g*eps_w - (u/d)/K ; dlogd = 0.5*g*eps_d*sqrt(d) - 0.5*(1 - u^2/(d K))
__global__ __launch_bounds__(256) void latent_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dz,
                                                         const float* __restrict__ u, const float* __restrict__ d,
                                                         const float* __restrict__ eps_w,
                                                         const float* __restrict__ eps_d, float* __restrict__ dmu,
                                                         float* __restrict__ du, float* __restrict__ dlogd, int B,
                                                         int zdim, const float* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  // scale = d(result)/d(loss) of a caller that backpropagates something other than the loss itself (null: 1).  dz
  // already carries it (the decoder's backward started from the scaled seed); the prior and entropy terms get it here.
  const float sc = scale != nullptr ? scale[0] : 1.f;
  float su2d = 0.f;
  for (int j = lane; j < zdim; j += 64) {
    const size_t i = (size_t)b * zdim + j;
    su2d += u[i] * u[i] / d[i];
  }
  const float K = 1.f + wave_sum(su2d);
  const float ew = eps_w[b];
  for (int j = lane; j < zdim; j += 64) {
    const size_t i = (size_t)b * zdim + j;
    const float g = fmaf(sc, z[i], dz[i]);
    const float uu = u[i], dd = d[i];
    dmu[i] = g;
    du[i] = g * ew - sc * ((uu / dd) / K);
    dlogd[i] = 0.5f * g * eps_d[i] * sqrtf(dd) - sc * (0.5f * (1.f - uu * uu / (dd * K)));
  }
}

__global__ __launch_bounds__(256) void elbo_finalize_kernel(const float* __restrict__ latent_sums, int B,
                                                            const float* __restrict__ sse_partials, int nparts,
                                                            int stride, int zdim, float prec, double xdim,
                                                            float* __restrict__ loss_out,
                                                            double* __restrict__ loss_accum) {
  __shared__ double red[3][4];
  double sz2 = 0.0, sh = 0.0, sse = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) { sz2 += (double)latent_sums[2 * b]; sh += (double)latent_sums[2 * b + 1]; }
  for (int p = threadIdx.x; p < nparts; p += 256) sse += (double)sse_partials[(size_t)p * stride];
  sz2 = wave_sum_d(sz2); sh = wave_sum_d(sh); sse = wave_sum_d(sse);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][w] = sz2; red[1][w] = sh; red[2][w] = sse; }
  __syncthreads();
  if (threadIdx.x == 0) {
    sz2 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    sh = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    sse = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    // -elbo = 0.5*(sum z^2 + zdim ln 2pi) + 0.5*X_DIM*ln(2pi/prec) + 0.5*prec*SSE - sum H   (vae.py:316-323)
    const double c1 = 0.5 * zdim * LOG_2PI;
    const double c2 = 0.5 * xdim * (LOG_2PI - log((double)prec));      // X_DIM = H*W (16384 for the reference's 128 x 128)
    const float loss = (float)(0.5 * sz2 + c1 + c2 + 0.5 * (double)prec * sse - sh);
    loss_out[0] = loss;
    if (loss_accum != nullptr) *loss_accum += (double)loss;      // running epoch sum (train_epoch's `train_loss += loss.item()`)
    loss_out[1] = (float)sz2;
    loss_out[2] = (float)sse;
    loss_out[3] = (float)sh;
  }
}

__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n4,
                                                        float one_minus_b1, float b2, float one_minus_b2,
                                                        float step_size, float sqrt_bc2, float eps,
                                                        const int* __restrict__ skip_if_set) {
  if (skip_if_set != nullptr && *skip_if_set != 0) {           // the forward flagged d <= 0 / NaN: no update (vae.py:312)
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(const_cast<int*>(skip_if_set) + 1, 1);    // word 1 counts the skipped steps
    return;
  }
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pp = p4[i], mm = m4[i], vv = v4[i];
    const float4 gg = g4[i];
#define AVA_ADAM1(c)                                                        \
    mm.c = mm.c + (gg.c - mm.c) * one_minus_b1;       /* exp_avg.lerp_ */   \
    vv.c = vv.c * b2 + (one_minus_b2 * gg.c) * gg.c;  /* mul_/addcmul_ */   \
    pp.c = pp.c - step_size * (mm.c / (sqrtf(vv.c) / sqrt_bc2 + eps));
    AVA_ADAM1(x) AVA_ADAM1(y) AVA_ADAM1(z) AVA_ADAM1(w)
#undef AVA_ADAM1
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
  }
}

__global__ void fill_normal_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = ava_normal_hash((uint64_t)i + offset, seed);
}

extern "C" int ava_latent_fwd(const float* mu, const float* u, const float* logd, const float* eps_w,
                              const float* eps_d, float* d, float* z, float* sums, int* status, int B, int zdim,
                              ava_stream_t s) {
  if (B <= 0 || zdim <= 0 || mu == nullptr || eps_w == nullptr || eps_d == nullptr) return AVA_EINVAL;
  hipLaunchKernelGGL(latent_fwd_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, to_stream(s), mu, u, logd, eps_w, eps_d, d,
                     z, sums, status, B, zdim);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}
extern "C" int ava_latent_bwd(const float* z, const float* dz_dec, const float* u, const float* d, const float* eps_w,
                              const float* eps_d, float* dmu, float* du, float* dlogd, int B, int zdim,
                              ava_stream_t s) {
  if (B <= 0 || zdim <= 0 || z == nullptr) return AVA_EINVAL;
  return ava_latent_bwd_scaled(z, dz_dec, u, d, eps_w, eps_d, dmu, du, dlogd, B, zdim, nullptr, to_stream(s));
}
int ava_latent_bwd_scaled(const float* z, const float* dz_dec, const float* u, const float* d, const float* eps_w,
                          const float* eps_d, float* dmu, float* du, float* dlogd, int B, int zdim, const float* scale,
                          hipStream_t st) {
  hipLaunchKernelGGL(latent_bwd_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, st, z, dz_dec, u, d, eps_w, eps_d, dmu, du,
                     dlogd, B, zdim, scale);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}
// A backward whose root is not the loss itself (d(result)/d(loss) = scale[0] != 1, ava_set_backward_scale): everything the
// forward left behind for the backward is LINEAR in the seed prec * (xhat - x), so it is scaled in place, in one launch --
//   * the seed itself (convt6's fused backward gathers convt7's data gradient from it),
//   * convt7's weight / bias gradient partial rows formed by its training forward (conv_thin_kernels.h: FOLD), and
//   * the two BatchNorm-backward sums of bn14 the same forward added to its accumulator slot (bn_acc.h): read in the
//     consumer's fixed order, multiplied in fp64, the slot cleared and the products written back as exact limbs of shard 0.
// scale[0] == 1 (loss.backward() with the loss as the root: autograd hands over a tensor of ones) returns at once without
// touching memory: that path is bit-identical to ava_backward without a scale.
__global__ __launch_bounds__(256) void scale_backward_roots_kernel(float* __restrict__ seed, int64_t n4, int seed_blocks,
                                                                   float* __restrict__ wg, int64_t nwg,
                                                                   long long* __restrict__ slot, const float* __restrict__ scale) {
  const float sc = scale[0];
  if (sc == 1.f) return;
  if ((int)blockIdx.x < seed_blocks) {
    float4* v4 = reinterpret_cast<float4*>(seed);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)seed_blocks * 256) {
      float4 t = v4[i];
      t.x *= sc; t.y *= sc; t.z *= sc; t.w *= sc;
      v4[i] = t;
    }
    return;
  }
  const int wb = (int)blockIdx.x - seed_blocks, nwb = (int)gridDim.x - seed_blocks;
  if (wg != nullptr)
    for (int64_t i = (int64_t)wb * 256 + threadIdx.x; i < nwg; i += (int64_t)nwb * 256) wg[i] *= sc;
  if (wb == 0 && slot != nullptr) {
    const int t = threadIdx.x;
    double v = 0.0;
    if (t < 64) v = bn_acc_read(slot, t) * (double)sc;          // NaN when the slot is poisoned: re-poisoned below
    __syncthreads();
    for (int i = t; i < AVA_ACC_SLOT_LL; i += 256) slot[i] = 0;
    __syncthreads();
    if (t < 64) {
      const double ad = fabs(v) * 0x1p48;
      if (!(ad < 0x1p95)) slot[192] = 1;
      else {
        const double l2 = floor(ad * 0x1p-64), r1 = ad - l2 * 0x1p64;
        const double l1 = floor(r1 * 0x1p-32), l0 = floor(r1 - l1 * 0x1p32);
        const long long sg = v < 0.0 ? -1 : 1;
        slot[t] = sg * (long long)l0; slot[64 + t] = sg * (long long)l1; slot[128 + t] = sg * (long long)l2;
      }
    }
  }
}
int ava_scale_backward_roots(float* seed, int64_t n, float* wg, int64_t nwg, long long* slot, const float* scale, hipStream_t st) {
  if (seed == nullptr || scale == nullptr || n % 4 != 0) return AVA_EINVAL;
  const int64_t n4 = n / 4;
  int seed_blocks = (int)((n4 + 255) / 256);
  if (seed_blocks > 2048) seed_blocks = 2048;
  int wg_blocks = (wg != nullptr || slot != nullptr) ? (int)((nwg + 255) / 256) : 0;
  if (wg_blocks > 256) wg_blocks = 256;
  if ((wg != nullptr || slot != nullptr) && wg_blocks < 1) wg_blocks = 1;
  hipLaunchKernelGGL(scale_backward_roots_kernel, dim3(seed_blocks + wg_blocks), dim3(256), 0, st, seed, n4, seed_blocks, wg, nwg,
                     slot, scale);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}
int ava_elbo_finalize_strided(const float* latent_sums, int B, const float* sse_partials, int nparts, int stride,
                              int zdim, float prec, int xdim, float* loss_out, double* loss_accum, hipStream_t st) {
  hipLaunchKernelGGL(elbo_finalize_kernel, dim3(1), dim3(256), 0, st, latent_sums, B, sse_partials, nparts, stride, zdim,
                     prec, (double)xdim, loss_out, loss_accum);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}
extern "C" int ava_elbo_finalize(const float* latent_sums, int B, const float* sse_partials, int nparts, int zdim,
                                 float prec, float* loss_out, ava_stream_t s) {
  if (latent_sums == nullptr || sse_partials == nullptr || loss_out == nullptr) return AVA_EINVAL;
  return ava_elbo_finalize_strided(latent_sums, B, sse_partials, nparts, 2, zdim, prec, 16384, loss_out, nullptr, to_stream(s));
}
extern "C" int ava_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                             double beta2, double eps, int step, ava_stream_t s) {
  return ava_adam_flat_guarded(p, g, m, v, n, lr, beta1, beta2, eps, step, nullptr, to_stream(s));
}
int ava_adam_flat_guarded(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                          double eps, int step, const int* skip_if_set, hipStream_t st) {
  if (p == nullptr || g == nullptr || m == nullptr || v == nullptr || n <= 0 || n % 4 != 0 || step < 1)
    return AVA_EINVAL;
  const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
  const float step_size = (float)(lr / bc1);
  const float sqrt_bc2 = (float)sqrt(bc2);
  int64_t n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_flat_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, n4,
                     (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_size, sqrt_bc2, (float)eps, skip_if_set);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}
extern "C" int ava_fill_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, ava_stream_t s) {
  if (out == nullptr || n <= 0) return AVA_EINVAL;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(fill_normal_kernel, dim3(blocks), dim3(256), 0, to_stream(s), out, n, seed, offset);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}
