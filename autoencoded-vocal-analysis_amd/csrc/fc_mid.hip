// The small fully connected middle of the network as ONE launch per direction (gfx950, exact fp32 v_mfma_f32_16x16x4_f32).
//
// Between fc2 and fc7 the reference (ava/models/vae.py:225-232 encode, :312-313 rsample, :259-261 decode) runs five tiny
// layers and the reparameterisation on [batch, <= 256] tensors: fc31|fc32|fc33 (256 -> 3 x 64), fc41 / fc42 / fc43 (64 -> z),
// the rank-1 Gaussian sample, fc5 (z -> 64), fc6 (64 -> 256).  As separate launches each of them is a 4-6 us dependent kernel
// that does < 1 us of work (launch-latency bound: DESIGN.md section 3).  No BatchNorm sits between them, so the batch rows are
// independent: a workgroup takes 16 rows through the whole chain -- every intermediate lives in LDS, is written to global memory
// once (the backward pass reads it) and feeds the next product from LDS.  The mirror kernel runs the data-gradient chain
// dh6 -> dh5 -> dz -> (latent backward) -> dh3 -> dh2.  Weights (0.3 MB in all) come from L2.
//
// One product step: C[16 x 16] += A[16 x K] (LDS, row-major, padded rows) x B[K x 16]; lane (m | n = lane & 15, kg = lane >> 4) holds
// A[m][k] and B[k][n] for k = 16 s + 4 kg + j of step (s, j) -- any permutation of k sums the same terms.  The 16-column tiles of a
// layer are dealt to the workgroup's eight waves.
#include "common.h"

typedef float fm_f32x4 __attribute__((ext_vector_type(4)));

// One 16-column tile of a layer's weights in registers, requested EARLY: a stage's weights do not depend on the stage before
// it, so a wave asks for them before it multiplies the current stage -- the L2 / HBM latency of every layer but the first
// hides behind the previous product and its barrier (the chain is a sequence of latencies, not of work).
// KMAX: compile-time bound of K (a multiple of 16); k >= K and columns >= N read as zero.
template <int KMAX>
struct FmWTile {
  float w[KMAX / 4];                     // element 4 s + j: k = 16 s + 4 kg + j
  // k-major weights W[N][ldw] (torch Linear.weight used as x W^T)
  __device__ __forceinline__ void load_kmajor(const float* __restrict__ W, int ldw, int n0, int N, int K, int lane) {
    const int mn = lane & 15, kg = lane >> 4;
    const bool nok = n0 + mn < N;
    const float* __restrict__ wrow = W + (size_t)(nok ? n0 + mn : 0) * ldw;
    const bool vec = (ldw & 3) == 0 && (K & 3) == 0 && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
#pragma unroll
    for (int s = 0; s < KMAX / 16; ++s) {
      const int k = 16 * s + 4 * kg;
      if (vec) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (nok && k < K) t = *reinterpret_cast<const float4*>(wrow + k);
        w[4 * s] = t.x; w[4 * s + 1] = t.y; w[4 * s + 2] = t.z; w[4 * s + 3] = t.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) w[4 * s + j] = (nok && k + j < K) ? wrow[k + j] : 0.f;
      }
    }
  }
  // n-major weights W[K][ldw] (dX = dY W with torch's [out][in] weight)
  __device__ __forceinline__ void load_nmajor(const float* __restrict__ W, int ldw, int n0, int N, int K, int lane) {
    const int mn = lane & 15, kg = lane >> 4;
    const bool nok = n0 + mn < N;
    const float* __restrict__ wcol = W + (nok ? n0 + mn : 0);
#pragma unroll
    for (int s = 0; s < KMAX / 16; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 16 * s + 4 * kg + j;
        w[4 * s + j] = (nok && k < K) ? wcol[(size_t)k * ldw] : 0.f;
      }
  }
  // C[16 x 16] = A_lds[16 x K] x tile; A rows are zero-padded to a multiple of 16 beyond K
  __device__ __forceinline__ fm_f32x4 mma(const float* __restrict__ A, int lda, int K, int lane) const {
    const int mn = lane & 15, kg = lane >> 4;
    const float* __restrict__ arow = A + mn * lda + 4 * kg;
    fm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KMAX / 16; ++s) {
      if (16 * s < K) {                                                   // wave-uniform
        const float4 av = *reinterpret_cast<const float4*>(arow + 16 * s);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, w[4 * s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, w[4 * s + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, w[4 * s + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, w[4 * s + 3], acc, 0, 0, 0);
      }
    }
    return acc;
  }
};

// rows [r0, r0 + 16) of a [B x N] global matrix into a zero-padded LDS tile [16][ld] (ld >= round16(N))
__device__ __forceinline__ void fm_load_rows(float* __restrict__ dst, int ld, const float* __restrict__ src, int lds, int r0,
                                             int B, int N, int t, int nt) {
  const int n16 = (N + 15) & ~15;
  for (int e = t; e < 16 * n16; e += nt) {
    const int r = e / n16, c = e - r * n16;
    dst[r * ld + c] = (r0 + r < B && c < N) ? src[(size_t)(r0 + r) * lds + c] : 0.f;
  }
}

struct FcMidFwdArgs {
  const float* h3_in;                    // unused (kept for layout compatibility)
  const float *W3, *b3;                  // unused: fc31|fc32|fc33 stays a launch of its own (its 196 KB of weights per workgroup)
  const float *W41, *b41, *W42, *b42, *W43, *b43;      // [z,64] heads
  const float *W5, *b5, *W6, *b6;        // [64,z], [256,64]
  const float *eps_w, *eps_d;            // [B], [B,z]
  float *h3;                             // INPUT here: relu(fc31|32|33(h2)) [B,192]
  float *mu, *u, *logd, *d, *z, *lat_sums, *h5, *h6;
  int* status;
  int B, zdim;
  unsigned long long* stamps;            // lab only
};

// lab: stage time stamps (s_memrealtime, 100 MHz) of workgroup 0 / wave 0, written to a caller-given buffer (tools/lab/fc_mid_probe.py)
#ifdef AVA_LAB
static unsigned long long* g_fm_stamps = nullptr;
extern "C" int ava_fc_mid_debug_stamps(unsigned long long* p) { g_fm_stamps = p; return 0; }
#define FM_STAMP(i) do { if (a.stamps != nullptr && blockIdx.x == 0 && t == 0) a.stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FM_STAMP(i) do { } while (0)
#endif

#define FM_LD(n) ((((n) + 15) & ~15) + 4)          // padded row length: rows 4 banks apart, 16-byte aligned
#define FM_THREADS 1024                            // 16 waves: every stage is ONE pass of <= 16 column tiles, one row per wave

__global__ __launch_bounds__(FM_THREADS) void fc_mid_fwd_kernel(const FcMidFwdArgs a) {
  __shared__ __align__(16) float s_h3[16 * FM_LD(192)];
  __shared__ __align__(16) float s_hd[3][16 * FM_LD(128)];       // mu, u, logd
  __shared__ __align__(16) float s_z[16 * FM_LD(128)];
  __shared__ __align__(16) float s_h5[16 * FM_LD(64)];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r0 = blockIdx.x * 16, B = a.B, z = a.zdim;
  const int mn = lane & 15, kg = lane >> 4;
  constexpr int L3 = FM_LD(192), LZ = FM_LD(128), L5 = FM_LD(64);
  const int ztiles = (z + 15) >> 4;

  FM_STAMP(0);
  // the heads' weights (24 KB) and fc6's (64 KB: the largest, needed last) are requested before anything else
  FmWTile<64> w4;
  const int hh = wave / ztiles, hnt = wave - hh * ztiles;                 // head and column tile of this wave (wave < 3 ztiles)
  const bool has4 = wave < 3 * ztiles;
  if (has4) w4.load_kmajor(hh == 0 ? a.W41 : (hh == 1 ? a.W42 : a.W43), 64, 16 * hnt, z, 64, lane);
  FmWTile<64> w6;
  w6.load_kmajor(a.W6, 64, 16 * wave, 256, 64, lane);
  fm_load_rows(s_h3, L3, a.h3, 192, r0, B, 192, t, FM_THREADS);
  for (int e = t; e < 16 * LZ; e += FM_THREADS) s_z[e] = 0.f;            // k padding of fc5's operand
  __syncthreads();
  FM_STAMP(1);
  FM_STAMP(2);

  // ---- heads: mu | u | log d = h3[:, 64 h : 64 h + 64] W4h^T + b4h; fc5's weights and the noise are requested first ----
  FmWTile<128> w5;
  if (wave < 4) w5.load_kmajor(a.W5, z, 16 * wave, 64, z, lane);
  const int brow = r0 + wave;                                             // the latent stage: one row per wave
  float ew = 0.f, ed[2] = {0.f, 0.f};
  if (brow < B) {
    ew = a.eps_w[brow];
#pragma unroll
    for (int i = 0; i < 2; ++i) ed[i] = (lane + 64 * i < z) ? a.eps_d[(size_t)brow * z + lane + 64 * i] : 0.f;
  }
  if (has4) {
    const float* bb = hh == 0 ? a.b41 : (hh == 1 ? a.b42 : a.b43);
    float* out = hh == 0 ? a.mu : (hh == 1 ? a.u : a.logd);
    const fm_f32x4 acc = w4.mma(s_h3 + 64 * hh, L3, 64, lane);
    const int n = 16 * hnt + mn;
    const float bias = n < z ? bb[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * kg + r;
      const float v = acc[r] + bias;
      if (n < z) {
        s_hd[hh][row * LZ + n] = v;
        if (r0 + row < B) out[(size_t)(r0 + row) * z + n] = v;
      }
    }
  }
  // 3 * ztiles can exceed 16 only for z > 80: the remaining (head, tile) pairs in a second pass
  for (int q = wave + 16; q < 3 * ztiles; q += 16) {
    const int h = q / ztiles, nt = q - h * ztiles;
    FmWTile<64> wq;
    wq.load_kmajor(h == 0 ? a.W41 : (h == 1 ? a.W42 : a.W43), 64, 16 * nt, z, 64, lane);
    const float* bb = h == 0 ? a.b41 : (h == 1 ? a.b42 : a.b43);
    float* out = h == 0 ? a.mu : (h == 1 ? a.u : a.logd);
    const fm_f32x4 acc = wq.mma(s_h3 + 64 * h, L3, 64, lane);
    const int n = 16 * nt + mn;
    const float bias = n < z ? bb[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * kg + r;
      const float v = acc[r] + bias;
      if (n < z) {
        s_hd[h][row * LZ + n] = v;
        if (r0 + row < B) out[(size_t)(r0 + row) * z + n] = v;
      }
    }
  }
  __syncthreads();
  FM_STAMP(3);

  // ---- rsample + entropy (misc.hip: latent_fwd_kernel, same operations in the same order): one row per wave ----
  if (brow < B) {                                                          // wave-uniform
    float sz2 = 0.f, su2d = 0.f, slogd = 0.f;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = lane + 64 * i;
      if (j < z) {
        const size_t gi = (size_t)brow * z + j;
        const float lg = s_hd[2][wave * LZ + j];
        const float d = expf(lg);
        const float uu = s_hd[1][wave * LZ + j];
        const float zz = s_hd[0][wave * LZ + j] + uu * ew + sqrtf(d) * ed[i];
        a.d[gi] = d;
        a.z[gi] = zz;
        s_z[wave * LZ + j] = zz;
        sz2 = fmaf(zz, zz, sz2);
        su2d += uu * uu / d;
        slogd += logf(d);
        bad |= !(d > 0.f);
      }
    }
    sz2 = wave_sum(sz2);
    su2d = wave_sum(su2d);
    slogd = wave_sum(slogd);
    if (bad && a.status != nullptr) atomicOr(a.status, 1);
    if (lane == 0) {
      const float K = 1.f + su2d;
      a.lat_sums[2 * brow] = sz2;
      a.lat_sums[2 * brow + 1] = 0.5f * ((float)(z * (1.0 + 1.8378770664093453)) + logf(K) + slogd);
    }
  }
  __syncthreads();
  FM_STAMP(4);

  // ---- h5 = relu(z W5^T + b5): 4 column tiles ----
  if (wave < 4) {
    const fm_f32x4 acc = w5.mma(s_z, LZ, z, lane);
    const int n = 16 * wave + mn;
    const float bias = a.b5[n];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * kg + r;
      const float v = fmaxf(acc[r] + bias, 0.f);
      s_h5[row * L5 + n] = v;
      if (r0 + row < B) a.h5[(size_t)(r0 + row) * 64 + n] = v;
    }
  }
  __syncthreads();
  FM_STAMP(5);

  // ---- h6 = relu(h5 W6^T + b6): 16 column tiles ----
  {
    const fm_f32x4 acc = w6.mma(s_h5, L5, 64, lane);
    const int n = 16 * wave + mn;
    const float bias = a.b6[n];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * kg + r;
      if (r0 + row < B) a.h6[(size_t)(r0 + row) * 256 + n] = fmaxf(acc[r] + bias, 0.f);
    }
  }
  FM_STAMP(6);
}

struct FcMidBwdArgs {
  const float* dh6;                      // [B,256] = (dh7 W7) masked by h6 > 0
  const float *W6, *W5, *W41, *W42, *W43, *W3;
  const float *h5, *h3, *h2;             // ReLU masks of the producers
  const float *z, *u, *d, *eps_w, *eps_d;
  const float* scale;                    // d(result)/d(loss) (null: 1), see latent_bwd_kernel
  float *dh5, *dz, *dmu, *du, *dlogd, *dh3, *dh2;
  int B, zdim;
  unsigned long long* stamps;            // lab only
};

__global__ __launch_bounds__(FM_THREADS) void fc_mid_bwd_kernel(const FcMidBwdArgs a) {
  __shared__ __align__(16) float s_g6[16 * FM_LD(256)];
  __shared__ __align__(16) float s_g5[16 * FM_LD(64)];
  __shared__ __align__(16) float s_gz[16 * FM_LD(128)];
  __shared__ __align__(16) float s_gh[3][16 * FM_LD(128)];       // dmu, du, dlogd
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r0 = blockIdx.x * 16, B = a.B, z = a.zdim;
  const int mn = lane & 15, kg = lane >> 4;
  constexpr int L6 = FM_LD(256), L5 = FM_LD(64), LZ = FM_LD(128);
  const int ztiles = (z + 15) >> 4;

  // W6's tiles are requested before anything else
  FmWTile<256> w6;
  if (wave < 4) w6.load_nmajor(a.W6, 64, 16 * wave, 64, 256, lane);
  fm_load_rows(s_g6, L6, a.dh6, 256, r0, B, 256, t, FM_THREADS);
  for (int e = t; e < 3 * 16 * LZ; e += FM_THREADS) (&s_gh[0][0])[e] = 0.f;     // k padding of the heads' operands
  for (int e = t; e < 16 * LZ; e += FM_THREADS) s_gz[e] = 0.f;
  // the latent stage's operands (one row per wave)
  const int brow = r0 + wave;
  float ew = 0.f, ed[2] = {0.f, 0.f}, uu[2] = {0.f, 0.f}, dd[2] = {1.f, 1.f}, zz[2] = {0.f, 0.f};
  if (brow < B) {
    ew = a.eps_w[brow];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = lane + 64 * i;
      if (j < z) {
        const size_t gi = (size_t)brow * z + j;
        ed[i] = a.eps_d[gi]; uu[i] = a.u[gi]; dd[i] = a.d[gi]; zz[i] = a.z[gi];
      }
    }
  }
  __syncthreads();

  // ---- dh5 = (dh6 W6) masked by h5 > 0: W6 is [256][64], 4 column tiles; W5's tiles are requested first ----
  FmWTile<64> w5;
  if (wave < ztiles) w5.load_nmajor(a.W5, z, 16 * wave, z, 64, lane);
  if (wave < 4) {
    const fm_f32x4 acc = w6.mma(s_g6, L6, 256, lane);
    const int n = 16 * wave + mn;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * kg + r;
      const bool live = r0 + row < B;
      const float v = (live && a.h5[(size_t)(r0 + row) * 64 + n] > 0.f) ? acc[r] : 0.f;
      s_g5[row * L5 + n] = v;
      if (live) a.dh5[(size_t)(r0 + row) * 64 + n] = v;
    }
  }
  __syncthreads();

  // ---- dz = dh5 W5: W5 is [64][z]; the heads' weights are requested first ----
  FmWTile<128> w4;
  const int hh = wave >> 2, hnt = wave & 3;                                // (head, column tile) of this wave, wave < 12
  if (wave < 12) w4.load_nmajor(hh == 0 ? a.W41 : (hh == 1 ? a.W42 : a.W43), 64, 16 * hnt, 64, z, lane);
  if (wave < ztiles) {
    const fm_f32x4 acc = w5.mma(s_g5, L5, 64, lane);
    const int n = 16 * wave + mn;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * kg + r;
      if (n < z) {
        s_gz[row * LZ + n] = acc[r];
        if (r0 + row < B) a.dz[(size_t)(r0 + row) * z + n] = acc[r];
      }
    }
  }
  __syncthreads();

  // ---- latent backward (misc.hip: latent_bwd_kernel): one row per wave ----
  if (brow < B) {                                                          // wave-uniform
    const float sc = a.scale != nullptr ? a.scale[0] : 1.f;
    float su2d = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (lane + 64 * i < z) su2d += uu[i] * uu[i] / dd[i];
    const float K = 1.f + wave_sum(su2d);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = lane + 64 * i;
      if (j < z) {
        const size_t gi = (size_t)brow * z + j;
        const float g = fmaf(sc, zz[i], s_gz[wave * LZ + j]);
        const float gmu = g;
        const float gu = g * ew - sc * ((uu[i] / dd[i]) / K);
        const float gl = 0.5f * g * ed[i] * sqrtf(dd[i]) - sc * (0.5f * (1.f - uu[i] * uu[i] / (dd[i] * K)));
        a.dmu[gi] = gmu; a.du[gi] = gu; a.dlogd[gi] = gl;
        s_gh[0][wave * LZ + j] = gmu; s_gh[1][wave * LZ + j] = gu; s_gh[2][wave * LZ + j] = gl;
      }
    }
  }
  __syncthreads();

  // ---- dh3[:, 64 h : 64 h + 64] = (d{mu,u,logd} W4h) masked by h3 > 0: W4h is [z][64], 3 x 4 column tiles ----
  if (wave < 12) {
    const fm_f32x4 acc = w4.mma(s_gh[hh], LZ, z, lane);
    const int n = 64 * hh + 16 * hnt + mn;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * kg + r;
      const bool live = r0 + row < B;
      const float v = (live && a.h3[(size_t)(r0 + row) * 192 + n] > 0.f) ? acc[r] : 0.f;
      if (live) a.dh3[(size_t)(r0 + row) * 192 + n] = v;
    }
  }
}

int ava_fc_mid_fwd(const FcMidFwdArgs& a_, hipStream_t st) {
  FcMidFwdArgs a = a_;
  if (a.B < 1 || a.zdim < 1 || a.zdim > 128) return AVA_EINVAL;
#ifdef AVA_LAB
  a.stamps = g_fm_stamps;
#else
  a.stamps = nullptr;
#endif
  hipLaunchKernelGGL(fc_mid_fwd_kernel, dim3((a.B + 15) / 16), dim3(FM_THREADS), 0, st, a);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}
int ava_fc_mid_bwd(const FcMidBwdArgs& a_, hipStream_t st) {
  FcMidBwdArgs a = a_;
  a.stamps = nullptr;
  if (a.B < 1 || a.zdim < 1 || a.zdim > 128) return AVA_EINVAL;
  hipLaunchKernelGGL(fc_mid_bwd_kernel, dim3((a.B + 15) / 16), dim3(FM_THREADS), 0, st, a);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}
