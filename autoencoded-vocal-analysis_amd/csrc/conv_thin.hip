// The four "thin" convolutions at full resolution (128x128, stride 1): conv1 (1 -> 8 channels) and convt7
// (8 -> 1) forward, their backward-data counterparts (convt7's is again 1 -> 8, conv1's 8 -> 1) and both
// weight gradients.  One side has a single channel, so the matrix cores would idle 15/16 of the time; these
// are HBM-bound VALU kernels: a thread owns a vertical strip of 4 pixels (lanes run along x, so LDS reads of
// neighbouring lanes are 4/32 bytes apart: no bank conflicts; 72 wave-uniform weights in scalar registers,
// 288 FMAs per thread and tile), the input window is staged in LDS with the prologue
// applied (zero padding after BatchNorm), outputs leave as 16-byte stores.  Same ConvArgs / WgradArgs /
// partial-row conventions as conv.hip.
#include "conv_common.h"

#define THIN_W 128          // image width handled (Wo == Wi == 128)
#define THIN_TH 8           // tile: 8 rows x 128 columns
#define THIN_IR (THIN_TH + 2)
#define THIN_IC (THIN_W + 2)

// sum N per-thread values over the workgroup: wave shuffles, then one LDS exchange (lds: [4][N] floats);
// result i is written to out[i] by thread i.  Fixed order -> deterministic.
template <int N>
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
    out[threadIdx.x] = (lds[threadIdx.x] + lds[N + threadIdx.x]) + (lds[2 * N + threadIdx.x] + lds[3 * N + threadIdx.x]);
}

// stage a 1-channel [10 x 130] window (origin row gy0, column -1) with the prologue applied
template <int PRO>
__device__ __forceinline__ void thin_stage1(float* __restrict__ lds, const float* __restrict__ in,
                                            const float* __restrict__ in2, float ca, float cb, float cc, int b,
                                            int H, int gy0) {
  for (int v = threadIdx.x; v < THIN_IR * THIN_IC; v += 256) {
    const int r = v / THIN_IC, c = v - r * THIN_IC;
    const int gy = gy0 + r, gx = c - 1;
    float o = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < THIN_W) {
      const size_t off = ((size_t)b * H + gy) * THIN_W + gx;
      o = prologue<PRO>(in[off], PRO == PRO_BWD ? in2[off] : 0.f, ca, cb, cc);
    }
    lds[v] = o;
  }
}

// ---------------------------------------------------------------------------------------------------------
// 1 -> 8 channels: conv1 forward (PRO_BN, EPI_FWD), convt7 backward-data (PRO_ID, EPI_BWD)
// ---------------------------------------------------------------------------------------------------------
template <int PRO, int EPI>
__global__ __launch_bounds__(256) void thin_1to8_kernel(const ConvArgs a) {
  __shared__ float tile[THIN_IR * THIN_IC];
  __shared__ float red[4 * 16];
  const int t = threadIdx.x, ty0 = (t >> 7) * 4, x = t & 127;      // pixels (ty0 + p, x), p = 0..3
  const float ca = a.pa ? a.pa[0] : 0.f, cb = a.pb ? a.pb[0] : 0.f, cc = a.pc ? a.pc[0] : 0.f;
  const float* __restrict__ G = a.G;                 // [9][1][8]
  float s1[8], s2[8];
#pragma unroll
  for (int co = 0; co < 8; ++co) s1[co] = s2[co] = 0.f;
  const int tiles_y = a.Ho / THIN_TH;
  for (TileWalk walk(a.ntiles); walk.valid(); walk.advance()) {
    const int tl = walk.cur;
    const int b = tl / tiles_y, oy0 = (tl - b * tiles_y) * THIN_TH;
    __syncthreads();
    thin_stage1<PRO>(tile, a.in, a.in2, ca, cb, cc, b, a.Hi, oy0 - 1);
    __syncthreads();
    float acc[4][8];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int co = 0; co < 8; ++co) acc[p][co] = 0.f;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float in[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) in[j] = tile[(ty0 + j) * THIN_IC + x + kx];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int co = 0; co < 8; ++co) {
          const float w = G[(ky * 3 + kx) * 8 + co];
#pragma unroll
          for (int p = 0; p < 4; ++p) acc[p][co] = fmaf(in[p + ky], w, acc[p][co]);
        }
    }
    const size_t opix0 = ((size_t)b * a.Ho + oy0 + ty0) * THIN_W + x;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const size_t opix = opix0 + (size_t)p * THIN_W - p;      // "+ p" below walks rows, not columns
      if (EPI == EPI_FWD) {
#pragma unroll
        for (int co = 0; co < 8; ++co) {
          float v = acc[p][co] + a.bias[co];
          if (a.relu) v = fmaxf(v, 0.f);
          acc[p][co] = v;
          s1[co] += v;
          s2[co] = fmaf(v, v, s2[co]);
        }
      } else {
        const float4 xa = *reinterpret_cast<const float4*>(a.epi_x + (opix + p) * 8);
        const float4 xb = *reinterpret_cast<const float4*>(a.epi_x + (opix + p) * 8 + 4);
        const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
        for (int co = 0; co < 8; ++co) {
          const float xh = (xv[co] - a.epi_mean[co]) * a.epi_invstd[co];
          s1[co] += acc[p][co];
          s2[co] = fmaf(acc[p][co], xh, s2[co]);
        }
      }
      if (a.out != nullptr) {
        float4* o = reinterpret_cast<float4*>(a.out + (opix + p) * 8);
        o[0] = make_float4(acc[p][0], acc[p][1], acc[p][2], acc[p][3]);
        o[1] = make_float4(acc[p][4], acc[p][5], acc[p][6], acc[p][7]);
      }
    }
  }
  float sv[16];
#pragma unroll
  for (int co = 0; co < 8; ++co) { sv[co] = s1[co]; sv[8 + co] = s2[co]; }
  thin_block_reduce<16>(sv, red, a.partials != nullptr ? a.partials + (size_t)blockIdx.x * 16 : nullptr);
}

// ---------------------------------------------------------------------------------------------------------
// 8 -> 1 channels: convt7 forward (PRO_BN, EPI_SSE), conv1 backward-data (PRO_BWD / PRO_ID, EPI_BWD)
// ---------------------------------------------------------------------------------------------------------
template <int PRO, int EPI>
__global__ __launch_bounds__(256) void thin_8to1_kernel(const ConvArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* tile = smem;                                   // [10][130][8]
  float* coef = smem + THIN_IR * THIN_IC * 8;           // [3][32]
  float* red = coef + 96;                               // [4][2]
  const int t = threadIdx.x, ty0 = (t >> 7) * 4, x = t & 127;
  if (t < 96) {
    const int which = t >> 5, c = t & 31;
    const float* src = which == 0 ? a.pa : (which == 1 ? a.pb : a.pc);
    coef[t] = (src != nullptr && c < 8) ? src[c] : 0.f;
  }
  const float* __restrict__ G = a.G;                    // [9][8][1]
  float s1 = 0.f, s2 = 0.f;
  TileStager<8, PRO, THIN_IR, THIN_IC> stg;
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
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float* px = tile + ((ty0 + j) * THIN_IC + x + kx) * 8;
        const float4 u = *reinterpret_cast<const float4*>(px);
        const float4 w4 = *reinterpret_cast<const float4*>(px + 4);
        const float in[8] = {u.x, u.y, u.z, u.w, w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int p = j - ky;                         // output row fed by input row j through tap ky
          if (p >= 0 && p < 4) {
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) acc[p] = fmaf(in[ci], G[(ky * 3 + kx) * 8 + ci], acc[p]);
          }
        }
      }
    const size_t opix0 = ((size_t)b * a.Ho + oy0 + ty0) * THIN_W + x;
    if (EPI == EPI_SSE) {
      const float bias = a.bias[0];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const size_t opix = opix0 + (size_t)p * THIN_W;
        const float v = acc[p] + bias;
        if (a.epi_x != nullptr) {
          const float r = v - a.epi_x[opix];
          a.out2[opix] = a.prec * r;
          s1 = fmaf(r, r, s1);
        }
        if (a.out != nullptr) a.out[opix] = v;
      }
    } else {
      const float m = a.epi_mean[0], is = a.epi_invstd[0];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const size_t opix = opix0 + (size_t)p * THIN_W;
        s1 += acc[p];
        s2 = fmaf(acc[p], (a.epi_x[opix] - m) * is, s2);
        if (a.out != nullptr) a.out[opix] = acc[p];
      }
    }
  }
  const float sv[2] = {s1, s2};
  thin_block_reduce<2>(sv, red, a.partials != nullptr ? a.partials + (size_t)blockIdx.x * 2 : nullptr);
}

// ---------------------------------------------------------------------------------------------------------
// weight gradients.  dy-side prologue is applied on the fly to each thread's own 4 pixels (no LDS needed for dy)
// ---------------------------------------------------------------------------------------------------------
// conv1: CIN = 1, COUT = 8.  dG[9][8], db[8]
template <int DYPRO>
__global__ __launch_bounds__(256) void thin_wgrad_1to8_kernel(const WgradArgs a) {
  __shared__ float tile[THIN_IR * THIN_IC];
  __shared__ float red[4 * 80];
  const int t = threadIdx.x, ty0 = (t >> 7) * 4, x = t & 127;
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
    thin_stage1<PRO_BN>(tile, a.x, nullptr, xa, xb, 0.f, b, a.Hi, oy0 - 1);
    __syncthreads();
    const size_t opix0 = ((size_t)b * a.Ho + oy0 + ty0) * THIN_W + x;
    float du[4][8];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const size_t opix = opix0 + (size_t)p * THIN_W - p;
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
      for (int j = 0; j < 6; ++j) in[j] = tile[(ty0 + j) * THIN_IC + x + kx];
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
  thin_block_reduce<80>(sv, red, a.partials + (size_t)blockIdx.x * 80);
}

// convt7: CIN = 8, COUT = 1.  dG[9][8][1], db[1]
template <int DYPRO>
__global__ __launch_bounds__(256) void thin_wgrad_8to1_kernel(const WgradArgs a) {
  extern __shared__ __align__(16) float smem[];
  float* tile = smem;
  float* coef = smem + THIN_IR * THIN_IC * 8;
  const int t = threadIdx.x, ty0 = (t >> 7) * 4, x = t & 127;
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
  TileStager<8, PRO_BN, THIN_IR, THIN_IC> stg;
  stg.init();
  const int tiles_y = a.Ho / THIN_TH;
  for (TileWalk walk(a.ntiles); walk.valid(); walk.advance()) {
    const int tl = walk.cur;
    const int b = tl / tiles_y, oy0 = (tl - b * tiles_y) * THIN_TH;
    __syncthreads();
    stg.load(a.x, nullptr, b, a.Hi, a.Wi, oy0 - 1, -1);
    stg.store(tile, coef);
    __syncthreads();
    const size_t opix0 = ((size_t)b * a.Ho + oy0 + ty0) * THIN_W + x;
    float du[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const size_t opix = opix0 + (size_t)p * THIN_W;
      du[p] = prologue<DYPRO>(a.dy[opix], DYPRO == PRO_BWD ? a.dy2[opix] : 0.f, da, db, dc);
    }
    bacc += (du[0] + du[1]) + (du[2] + du[3]);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float* px = tile + ((ty0 + j) * THIN_IC + x + kx) * 8;
        const float4 u = *reinterpret_cast<const float4*>(px);
        const float4 w4 = *reinterpret_cast<const float4*>(px + 4);
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
  thin_block_reduce<73>(sv, smem, a.partials + (size_t)blockIdx.x * 73);     // tiles are dead: reuse their LDS
}

// ---------------------------------------------------------------------------------------------------------
// dispatch (called from conv.hip before the generic kernels); AVA_EINVAL = shape not handled here
// ---------------------------------------------------------------------------------------------------------
static const size_t kThin8Lds = (size_t)(THIN_IR * THIN_IC * 8 + 96 + 8) * sizeof(float);

template <typename K>
static int thin_set_lds(K kernel) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)kThin8Lds) == hipSuccess ? AVA_OK : AVA_ELAUNCH;
}

int ava_conv3x3_thin(const ConvArgs& a0, int grid, int Cin, int Cout, int mode, int pro, int epi, hipStream_t st) {
  if (mode != MODE_S1 || a0.Wo != THIN_W || a0.Ho % THIN_TH != 0) return AVA_EINVAL;
  if (a0.bn.counter != nullptr) return AVA_EINVAL;       // fused BatchNorm finalisation lives in the generic kernels
  ConvArgs a = a0;
  a.ntiles = a.B * (a.Ho / THIN_TH);                     // workgroups beyond ntiles still write their (zero) partial row
  if (Cin == 1 && Cout == 8) {
    if (pro == PRO_BN && epi == EPI_FWD) hipLaunchKernelGGL((thin_1to8_kernel<PRO_BN, EPI_FWD>), dim3(grid), dim3(256), 0, st, a);
    else if (pro == PRO_ID && epi == EPI_BWD) hipLaunchKernelGGL((thin_1to8_kernel<PRO_ID, EPI_BWD>), dim3(grid), dim3(256), 0, st, a);
    else if (pro == PRO_BWD && epi == EPI_BWD) hipLaunchKernelGGL((thin_1to8_kernel<PRO_BWD, EPI_BWD>), dim3(grid), dim3(256), 0, st, a);
    else return AVA_EINVAL;
  } else if (Cin == 8 && Cout == 1) {
    static bool attr = false;
    if (!attr) {
      if (thin_set_lds(&thin_8to1_kernel<PRO_BN, EPI_SSE>) != AVA_OK || thin_set_lds(&thin_8to1_kernel<PRO_BWD, EPI_BWD>) != AVA_OK ||
          thin_set_lds(&thin_8to1_kernel<PRO_ID, EPI_BWD>) != AVA_OK)
        return AVA_ELAUNCH;
      attr = true;
    }
    if (pro == PRO_BN && epi == EPI_SSE) hipLaunchKernelGGL((thin_8to1_kernel<PRO_BN, EPI_SSE>), dim3(grid), dim3(256), kThin8Lds, st, a);
    else if (pro == PRO_BWD && epi == EPI_BWD) hipLaunchKernelGGL((thin_8to1_kernel<PRO_BWD, EPI_BWD>), dim3(grid), dim3(256), kThin8Lds, st, a);
    else if (pro == PRO_ID && epi == EPI_BWD) hipLaunchKernelGGL((thin_8to1_kernel<PRO_ID, EPI_BWD>), dim3(grid), dim3(256), kThin8Lds, st, a);
    else return AVA_EINVAL;
  } else {
    return AVA_EINVAL;
  }
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

int ava_conv3x3_wgrad_thin(const WgradArgs& a0, int grid, int Cin, int Cout, int mode, int dy_pro, hipStream_t st) {
  if (mode != MODE_S1 || a0.Wo != THIN_W || a0.Ho % THIN_TH != 0) return AVA_EINVAL;
  WgradArgs a = a0;
  a.ntiles = a.B * (a.Ho / THIN_TH);
  if (Cin == 1 && Cout == 8) {
    if (dy_pro == PRO_BWD) hipLaunchKernelGGL((thin_wgrad_1to8_kernel<PRO_BWD>), dim3(grid), dim3(256), 0, st, a);
    else if (dy_pro == PRO_ID) hipLaunchKernelGGL((thin_wgrad_1to8_kernel<PRO_ID>), dim3(grid), dim3(256), 0, st, a);
    else return AVA_EINVAL;
  } else if (Cin == 8 && Cout == 1) {
    static bool attr = false;
    if (!attr) {
      if (thin_set_lds(&thin_wgrad_8to1_kernel<PRO_BWD>) != AVA_OK || thin_set_lds(&thin_wgrad_8to1_kernel<PRO_ID>) != AVA_OK)
        return AVA_ELAUNCH;
      attr = true;
    }
    if (dy_pro == PRO_BWD) hipLaunchKernelGGL((thin_wgrad_8to1_kernel<PRO_BWD>), dim3(grid), dim3(256), kThin8Lds, st, a);
    else if (dy_pro == PRO_ID) hipLaunchKernelGGL((thin_wgrad_8to1_kernel<PRO_ID>), dim3(grid), dim3(256), kThin8Lds, st, a);
    else return AVA_EINVAL;
  } else {
    return AVA_EINVAL;
  }
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}
