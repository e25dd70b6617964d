// 3x3 gather convolutions for the VAE encoder/decoder, NHWC fp32, gfx950.
//
// Replaces the ATen conv / conv_transpose / their backward kernels the reference reaches through
// nn.Conv2d / nn.ConvTranspose2d (ava/models/vae.py:128-134,155-161,217-223,263-269) together with
// the BatchNorm apply in front of every convolution (the zero padding is inserted AFTER BatchNorm),
// the bias + ReLU behind it and the reductions the next BatchNorm needs.
//
// All seven ops of a layer pair (conv / convT, forward / backward-data) are one of three gather
// patterns over a weight table G[tap][cin][cout] (see ava_pack_conv_weight):
//   S1   out(y,x)  = sum_t in(y+ky-1, x+kx-1) G[t]         conv s1, convT s1, and their bwd-data
//   DOWN out(y,x)  = sum_t in(2y+ky-1, 2x+kx-1) G[t]       conv s2 fwd, convT s2 bwd-data
//   UP   out(oy,ox)= sum_{t: parity ok} in((oy+1-ky)/2, (ox+1-kx)/2) G[t]   convT s2 fwd, conv s2 bwd-data
//
//
// This file holds the entry points (ava_conv3x3, ava_conv3x3_wgrad, geometry queries), the weight packing and the
// weight-gradient reduction; the kernels live in conv_mfma.hip / conv_ws.hip (matrix cores, >= 8 channels on both
// sides), conv_fused.hip (fused backward) and conv_thin.hip (packed-FMA kernels of the 1- and 8-channel layers).
#include <string.h>
#include "conv_common.h"

// ------------------------------------------------------------------------------------------------
// small helpers: weight packing, wgrad reduction
// ------------------------------------------------------------------------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ g, int c0, int c1, int swap,
                                   int flip) {
  const int n = c0 * c1 * 9;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int tt = i % 9, r = i / 9, i1 = r % c1, i0 = r / c1;      // w[i0][i1][tt]
    const int tg = flip ? 8 - tt : tt;
    const int gi = swap ? (tg * c1 + i1) * c0 + i0 : (tg * c0 + i0) * c1 + i1;
    g[gi] = w[i];
  }
}

// dw (reference layout [c0][c1][9]) from partial dG rows; kind 0: conv (G[t][ci=c1][co=c0]),
// kind 1: convT s1 (G[8-t][c0][c1]), kind 2: convT s2 (G[t][c0][c1]).
// 256 threads = 32 row entries x 8 row groups: reads are coalesced along a partial row, the rows are
// summed in fp64 in a fixed order (deterministic), the (tiny) result is scattered to the weight layout.
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ partials, int nparts,
                                                  float* __restrict__ dw, float* __restrict__ dbias, int cin,
                                                  int cout, int kind, int block) {
  __shared__ double red[8][33];
  const int nw = 9 * cin * cout, row = nw + cout;
  const int e = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int gi = block * 32 + e;                      // index inside a partial row
  double s = 0.0;
  if (gi < row) {
    const float* p = partials + gi;
    int r = rg;
    // sixteen loads in flight per thread, consumed in the grouping of the four-at-a-time loop below (same sums, same
    // order): with four in flight a thread went through 16 dependent round trips for 512 partial rows
    for (; r + 120 < nparts; r += 128) {
      float v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = p[(size_t)(r + 8 * k) * row];
#pragma unroll
      for (int k = 0; k < 16; k += 4) s += ((double)v[k] + (double)v[k + 1]) + ((double)v[k + 2] + (double)v[k + 3]);
    }
    for (; r + 24 < nparts; r += 32) {                // 4 independent loads in flight per thread
      const float a0 = p[(size_t)r * row], a1 = p[(size_t)(r + 8) * row];
      const float a2 = p[(size_t)(r + 16) * row], a3 = p[(size_t)(r + 24) * row];
      s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
    }
    for (; r < nparts; r += 8) s += (double)p[(size_t)r * row];
  }
  red[rg][e] = s;
  __syncthreads();
  if (rg == 0 && gi < row) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) tot += red[k][e];
    if (gi >= nw) { dbias[gi - nw] = (float)tot; return; }
    // gi = (tg*cin + a)*cout + b  in gather form G[tg][cin index a][cout index b]
    const int b = gi % cout, r = gi / cout, a = r % cin, tg = r / cin;
    int i;
    if (kind == 0) i = (b * cin + a) * 9 + tg;                 // W[co=b][ci=a][t]
    else if (kind == 1) i = (a * cout + b) * 9 + (8 - tg);     // Wt[ci=a][co=b][8-t]
    else i = (a * cout + b) * 9 + tg;                          // Wt[ci=a][co=b][t]
    dw[i] = (float)tot;
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partials, int nparts,
                                                           float* __restrict__ dw, float* __restrict__ dbias,
                                                           int cin, int cout, int kind) {
  wgrad_reduce_body(partials, nparts, dw, dbias, cin, cout, kind, blockIdx.x);
}

__global__ __launch_bounds__(256) void wgrad_reduce_all_kernel(const WgradReduceTable tab) {
  if (blockIdx.x == 0 && tab.fin0.acc != nullptr) {
    __shared__ float coef[96];
    __shared__ double accvals[64];
    bn_coef_from_acc(coef, accvals, tab.fin0, 0);           // publishes d gamma, d beta of bn1 (and its unused A, Bc, Cc)
  }
  int i = 0;
#pragma unroll 1
  for (int k = 1; k < tab.n; ++k)
    if ((int)blockIdx.x >= tab.e[k].block0) i = k;
  const WgradReduceEntry e = tab.e[i];
  wgrad_reduce_body(e.partials, e.nparts, e.dw, e.dbias, e.cin, e.cout, e.kind, blockIdx.x - e.block0);
}

int ava_conv_wgrad_reduce_all(const WgradReduceTable& tab, int total_blocks, hipStream_t st) {
  hipLaunchKernelGGL(wgrad_reduce_all_kernel, dim3(total_blocks), dim3(256), 0, st, tab);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int ava_conv3x3_mfma(const ConvArgs& a, int grid, int Cin, int Cout, int mode, int pro, int epi, hipStream_t st);
int ava_conv3x3_wgrad_mfma(const WgradArgs& a, int grid, int Cin, int Cout, int mode, int dy_pro, hipStream_t st);
int ava_conv3x3_thin(const ConvArgs& a, int grid, int Cin, int Cout, int mode, int pro, int epi, hipStream_t st);
int ava_conv3x3_wgrad_thin(const WgradArgs& a, int grid, int Cin, int Cout, int mode, int dy_pro, hipStream_t st);
int ava_conv3x3_wgrad_thin_rows(const WgradArgs& a, int grid, int Cin, int Cout, int mode);

#ifdef AVA_LAB
// lab build: AVA_CONV_IMPL=valu forces the version-0 VALU kernels (lab/conv_valu.hip) everywhere
int ava_conv3x3_valu(const ConvArgs& a, int grid, int Cin, int Cout, int mode, int pro, int epi, int tw, hipStream_t st);
int ava_conv3x3_wgrad_valu(const WgradArgs& a, int grid, int Cin, int Cout, int mode, int dy_pro, int tw, hipStream_t st);
static bool use_mfma() {
  static int cached = -1;
  if (cached < 0) {
    const char* e = ava_env("AVA_CONV_IMPL");
    cached = (e != nullptr && strcmp(e, "valu") == 0) ? 0 : 1;
  }
  return cached == 1;
}
#else
static constexpr bool use_mfma() { return true; }
#endif

static int tile_w(int Wo) { return Wo >= 32 ? 32 : 16; }

static int conv_geometry(int B, int Ho, int Wo, int* tiles_y, int* tiles_x) {
  const int tw = tile_w(Wo), th = 256 / tw;
  if (Wo % tw != 0 || Ho % th != 0) return -1;
  *tiles_y = Ho / th;
  *tiles_x = Wo / tw;
  return B * (*tiles_y) * (*tiles_x);
}

// rows of the per-workgroup partial statistics buffer: one per 128 output pixels, at most 1024 (the matrix-core
// kernels launch one resident wave of workgroups, which can be up to twice the number of 256-pixel tiles for the
// 16 x 16 layers; rows of workgroups that are not launched are zero-filled)
extern "C" int ava_conv_grid(int B, int Ho, int Wo, int mode) {
  (void)mode;
  int ty, tx;
  const int nt = conv_geometry(B, Ho, Wo, &ty, &tx);
  if (nt < 0) return AVA_EINVAL;
  return 2 * nt < 1024 ? 2 * nt : 1024;
}
// rows of the weight-gradient partial buffer: one per 128 output pixels, at most 512 (see ava_conv_grid)
extern "C" int ava_conv_wgrad_grid(int B, int Ho, int Wo, int mode) {
  (void)mode;
  int ty, tx;
  const int nt = conv_geometry(B, Ho, Wo, &ty, &tx);
  if (nt < 0) return AVA_EINVAL;
  return 2 * nt < 512 ? 2 * nt : 512;
}

int ava_conv3x3_ex(const float* in, const float* in2, const float* pa, const float* pb, const float* pc,
                   const float* G, const float* bias, float* out, float* out2, const float* epi_x,
                   const float* epi_mean, const float* epi_invstd, float* partials, int B, int Hi, int Wi, int Cin,
                   int Cout, int mode, int pro, int epi, int relu, float prec, int act_bf16, const ConvAcc* acc, ava_stream_t s);

extern "C" int ava_conv3x3(const float* in, const float* in2, const float* pa, const float* pb, const float* pc,
                           const float* G, const float* bias, float* out, float* out2, const float* epi_x,
                           const float* epi_mean, const float* epi_invstd, float* partials, int B, int Hi, int Wi,
                           int Cin, int Cout, int mode, int pro, int epi, int relu, float prec, ava_stream_t s) {
  return ava_conv3x3_ex(in, in2, pa, pb, pc, G, bias, out, out2, epi_x, epi_mean, epi_invstd, partials, B, Hi, Wi, Cin,
                        Cout, mode, pro, epi, relu, prec, 0, nullptr, s);
}

// the same entry for the model driver (model.hip); act_bf16: the activations among the operands (a forward layer's
// input and output, the saved activation in2, the raw x of the BatchNorm-backward sums) are stored as bfloat16
int ava_conv3x3_ex(const float* in, const float* in2, const float* pa, const float* pb, const float* pc,
                   const float* G, const float* bias, float* out, float* out2, const float* epi_x,
                   const float* epi_mean, const float* epi_invstd, float* partials, int B, int Hi, int Wi, int Cin,
                   int Cout, int mode, int pro, int epi, int relu, float prec, int act_bf16, const ConvAcc* acc,
                   ava_stream_t s) {
  ConvArgs a;
#ifdef AVA_LAB
  a.stamps = nullptr;
#endif
  a.rc = acc != nullptr ? acc->rc : RecompArgs{};
  a.fold = acc != nullptr ? acc->fold : ThinFold{};
  a.act_bf16 = act_bf16;
  a.acc_out = acc != nullptr ? acc->acc_out : nullptr;
  if (acc != nullptr) a.fin = acc->fin;
  else { a.fin = BnFin{}; a.fin.acc = nullptr; }
  a.in = in; a.in2 = in2; a.pa = pa; a.pb = pb; a.pc = pc; a.G = G; a.bias = bias; a.out = out; a.out2 = out2;
  a.epi_x = epi_x; a.epi_mean = epi_mean; a.epi_invstd = epi_invstd; a.partials = partials;
  a.B = B; a.Hi = Hi; a.Wi = Wi; a.relu = relu; a.prec = prec;
  { static int dbg = -1; if (dbg < 0) { const char* e = ava_env("AVA_DBG"); dbg = e ? atoi(e) : 0; } a.dbg = dbg; }
  a.Ho = mode == MODE_S1 ? Hi : (mode == MODE_DOWN ? Hi / 2 : Hi * 2);
  a.Wo = mode == MODE_S1 ? Wi : (mode == MODE_DOWN ? Wi / 2 : Wi * 2);
  a.ntiles = conv_geometry(B, a.Ho, a.Wo, &a.tiles_y, &a.tiles_x);
  if (a.ntiles <= 0 || in == nullptr || G == nullptr) return AVA_EINVAL;
  if (epi != EPI_BWD && bias == nullptr) return AVA_EINVAL;
  if (pro == PRO_BWD && in2 == nullptr) return AVA_EINVAL;
  const int grid = 2 * a.ntiles < 1024 ? 2 * a.ntiles : 1024;     // == ava_conv_grid
  a.part_rows = grid;
  const int tw = tile_w(a.Wo);
  hipStream_t st = to_stream(s);
  if (use_mfma() && epi != EPI_SSE) {
    const int rc = ava_conv3x3_mfma(a, grid, Cin, Cout, mode, pro, epi, st);
    if (rc != AVA_EINVAL) return rc;       // AVA_EINVAL: no matrix-core instantiation for this shape
  }
  if (use_mfma()) {                        // single-channel layers at 128x128: dedicated VALU kernels
    const int rc = ava_conv3x3_thin(a, grid, Cin, Cout, mode, pro, epi, st);
    if (rc != AVA_EINVAL) return rc;
  }
#ifdef AVA_LAB
  return ava_conv3x3_valu(a, grid, Cin, Cout, mode, pro, epi, tw, st);
#else
  (void)tw;
  return AVA_EINVAL;                       // no kernel for this (Cin, Cout, mode, size)
#endif
}

int ava_conv3x3_wgrad_ex(const float* x, const float* xa, const float* xb, const float* dy, const float* dy2,
                         const float* da, const float* db_, const float* dc, float* partials, int B, int Hi, int Wi,
                         int Cin, int Cout, int mode, int dy_pro, int act_bf16, ava_stream_t s);
extern "C" int ava_conv3x3_wgrad(const float* x, const float* xa, const float* xb, const float* dy, const float* dy2,
                                 const float* da, const float* db_, const float* dc, float* partials, int B, int Hi,
                                 int Wi, int Cin, int Cout, int mode, int dy_pro, ava_stream_t s) {
  return ava_conv3x3_wgrad_ex(x, xa, xb, dy, dy2, da, db_, dc, partials, B, Hi, Wi, Cin, Cout, mode, dy_pro, 0, s);
}
int ava_conv3x3_wgrad_ex(const float* x, const float* xa, const float* xb, const float* dy, const float* dy2,
                         const float* da, const float* db_, const float* dc, float* partials, int B, int Hi, int Wi,
                         int Cin, int Cout, int mode, int dy_pro, int act_bf16, ava_stream_t s) {
  WgradArgs a;
  a.act_bf16 = act_bf16;
  a.x = x; a.xa = xa; a.xb = xb; a.dy = dy; a.dy2 = dy2; a.da = da; a.db = db_; a.dc = dc; a.partials = partials;
  a.B = B; a.Hi = Hi; a.Wi = Wi;
  a.Ho = mode == MODE_S1 ? Hi : (mode == MODE_DOWN ? Hi / 2 : Hi * 2);
  a.Wo = mode == MODE_S1 ? Wi : (mode == MODE_DOWN ? Wi / 2 : Wi * 2);
  a.ntiles = conv_geometry(B, a.Ho, a.Wo, &a.tiles_y, &a.tiles_x);
  if (a.ntiles <= 0 || x == nullptr || dy == nullptr || partials == nullptr) return AVA_EINVAL;
  if (dy_pro == PRO_BWD && dy2 == nullptr) return AVA_EINVAL;
  const int grid = 2 * a.ntiles < 512 ? 2 * a.ntiles : 512;   // == ava_conv_wgrad_grid; the matrix-core kernels may launch fewer
  const int tw = tile_w(a.Wo);
  hipStream_t st = to_stream(s);
  if (use_mfma()) {
    int rc = ava_conv3x3_wgrad_mfma(a, grid, Cin, Cout, mode, dy_pro, st);
    if (rc != AVA_EINVAL) return rc;
    rc = ava_conv3x3_wgrad_thin(a, grid, Cin, Cout, mode, dy_pro, st);
    if (rc != AVA_EINVAL) return rc;
  }
#ifdef AVA_LAB
  return ava_conv3x3_wgrad_valu(a, grid, Cin, Cout, mode, dy_pro, tw, st);
#else
  (void)tw;
  return AVA_EINVAL;
#endif
}

// Two layers' weight gradients in one launch where a pair kernel exists (conv_mfma.hip: the 16 x 16 layers), otherwise
// one after the other.  Same partial rows either way.
int ava_conv3x3_wgrad_mfma_pair(const WgradArgs& a, int grid_a, const WgradCall& ca, const WgradArgs& b, int grid_b,
                                const WgradCall& cb, hipStream_t st);
int ava_conv3x3_wgrad_pair(const WgradCall& p, const WgradCall& q, int B, int act_bf16, ava_stream_t s) {
  static const bool on = [] { const char* e = ava_env("AVA_WGRAD_PAIR"); return e == nullptr || atoi(e) != 0; }();
  if (on && use_mfma()) {
    WgradArgs w[2];
    int grid[2];
    const WgradCall* c[2] = {&p, &q};
    bool ok = true;
    for (int i = 0; i < 2 && ok; ++i) {
      WgradArgs& a = w[i];
      a.act_bf16 = act_bf16;
      a.x = c[i]->x; a.xa = c[i]->xa; a.xb = c[i]->xb; a.dy = c[i]->dy; a.dy2 = c[i]->dy2;
      a.da = c[i]->da; a.db = c[i]->db; a.dc = c[i]->dc; a.partials = c[i]->partials;
      a.B = B; a.Hi = c[i]->Hi; a.Wi = c[i]->Wi;
      const int mode = c[i]->mode;
      a.Ho = mode == MODE_S1 ? a.Hi : (mode == MODE_DOWN ? a.Hi / 2 : a.Hi * 2);
      a.Wo = mode == MODE_S1 ? a.Wi : (mode == MODE_DOWN ? a.Wi / 2 : a.Wi * 2);
      a.ntiles = conv_geometry(B, a.Ho, a.Wo, &a.tiles_y, &a.tiles_x);
      ok = a.ntiles > 0 && a.x != nullptr && a.dy != nullptr && a.partials != nullptr &&
           !(c[i]->dy_pro == PRO_BWD && a.dy2 == nullptr);
      grid[i] = 2 * a.ntiles < 512 ? 2 * a.ntiles : 512;
    }
    if (ok) {
      const int rc = ava_conv3x3_wgrad_mfma_pair(w[0], grid[0], p, w[1], grid[1], q, to_stream(s));
      if (rc != AVA_EINVAL) return rc;
    }
  }
  int rc = ava_conv3x3_wgrad_ex(p.x, p.xa, p.xb, p.dy, p.dy2, p.da, p.db, p.dc, p.partials, B, p.Hi, p.Wi, p.Cin, p.Cout,
                                p.mode, p.dy_pro, act_bf16, s);
  if (rc != AVA_OK) return rc;
  return ava_conv3x3_wgrad_ex(q.x, q.xa, q.xb, q.dy, q.dy2, q.da, q.db, q.dc, q.partials, B, q.Hi, q.Wi, q.Cin, q.Cout,
                              q.mode, q.dy_pro, act_bf16, s);
}


// partial rows ava_conv3x3_wgrad writes for this shape (<= ava_conv_wgrad_grid): the matrix-core kernels launch one
// resident wave of workgroups, which depends on the kernel's occupancy
int ava_conv_wgrad_rows_ex(int B, int Hi, int Wi, int Cin, int Cout, int mode, int dy_pro, int act_bf16);
extern "C" int ava_conv_wgrad_rows(int B, int Hi, int Wi, int Cin, int Cout, int mode, int dy_pro) {
  return ava_conv_wgrad_rows_ex(B, Hi, Wi, Cin, Cout, mode, dy_pro, 0);
}
int ava_conv_wgrad_rows_ex(int B, int Hi, int Wi, int Cin, int Cout, int mode, int dy_pro, int act_bf16) {
  WgradArgs a = {};
  a.act_bf16 = act_bf16;
  a.B = B; a.Hi = Hi; a.Wi = Wi;
  a.Ho = mode == MODE_S1 ? Hi : (mode == MODE_DOWN ? Hi / 2 : Hi * 2);
  a.Wo = mode == MODE_S1 ? Wi : (mode == MODE_DOWN ? Wi / 2 : Wi * 2);
  a.ntiles = conv_geometry(B, a.Ho, a.Wo, &a.tiles_y, &a.tiles_x);
  if (a.ntiles <= 0) return AVA_EINVAL;
  const int grid = 2 * a.ntiles < 512 ? 2 * a.ntiles : 512;
  if (use_mfma()) {
    const int rows = ava_conv3x3_wgrad_mfma(a, grid, Cin, Cout, mode, dy_pro, nullptr);   // partials == NULL: query
    if (rows > 0) return rows;
    const int trows = ava_conv3x3_wgrad_thin_rows(a, grid, Cin, Cout, mode);
    if (trows > 0) return trows;
  }
  return grid;
}

extern "C" int ava_pack_conv_weight(const float* w, float* g, int c_first, int c_second, int kind, ava_stream_t s) {
  if (w == nullptr || g == nullptr || kind < 0 || kind > 6) return AVA_EINVAL;
  const int swap = (kind == 0 || kind == 5 || kind == 6) ? 1 : 0;
  const int flip = (kind == 1 || kind == 3) ? 1 : 0;
  const int n = c_first * c_second * 9;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, to_stream(s), w, g, c_first, c_second,
                     swap, flip);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

extern "C" int ava_conv_wgrad_reduce(const float* partials, int nparts, float* dw, float* dbias, int Cin, int Cout,
                                     int kind, ava_stream_t s) {
  if (partials == nullptr || dw == nullptr || dbias == nullptr || kind < 0 || kind > 2) return AVA_EINVAL;
  const int n = 9 * Cin * Cout + Cout;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(n, 32)), dim3(256), 0, to_stream(s), partials, nparts, dw,
                     dbias, Cin, Cout, kind);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}
