// Whole-path driver: VAE.forward / loss.backward() / Adam.step of ava/models/vae.py:273-353 as a
// fixed sequence of kernel launches on one HIP stream, over a caller-provided workspace.
// Host-only bookkeeping lives in `ava_model`; nothing here allocates device memory or synchronises.
#include <stdlib.h>
#include <string.h>
#include <string>
#include <map>
#include "conv_common.h"
#include "conv_fused.h"

// internal entry points of the other translation units
int ava_nchw_to_nhwc_stats(const float* in, float* out, float* partials, int B, int P, int act_bf16, long long* acc_out,
                           int* nparts, hipStream_t st);
int ava_conv3x3_wgrad_ex(const float* x, const float* xa, const float* xb, const float* dy, const float* dy2,
                         const float* da, const float* db_, const float* dc, float* partials, int B, int Hi, int Wi,
                         int Cin, int Cout, int mode, int dy_pro, int act_bf16, ava_stream_t s);
int ava_conv_wgrad_rows_ex(int B, int Hi, int Wi, int Cin, int Cout, int mode, int dy_pro, int act_bf16);
int ava_bn_eval_all(const float* const* gamma, const float* const* beta, const int* C, const float* running, float* save,
                    hipStream_t st);
int ava_conv3x3_ex(const float* in, const float* in2, const float* pa, const float* pb, const float* pc,
                   const float* G, const float* bias, float* out, float* out2, const float* epi_x,
                   const float* epi_mean, const float* epi_invstd, float* partials, int B, int Hi, int Wi, int Cin,
                   int Cout, int mode, int pro, int epi, int relu, float prec, int act_bf16, const ConvAcc* acc, ava_stream_t s);
int ava_nhwc_to_nchw(const float* in, float* out, int B, int P, hipStream_t st);
int ava_relu_mask_to_nhwc(const float* dy_nchw, const float* slab1, const float* y_nhwc, float* du, int B, int P, hipStream_t st);
int ava_nchw_to_nhwc_stats_slabs(const float* slab0, const float* slab1, const float* bias, int relu, float* full, float* out,
                                 float* partials, int B, int P, int act_bf16, long long* acc_out, int* nparts, hipStream_t st);
int ava_gemm_defer2(const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc,
                    const float* mask, float* colsum, int M, int N, int K, int a_kmajor, int b_kmajor, int act,
                    void* ws, size_t ws_bytes, ava_stream_t s, int* slabs);
int ava_bn_bwd_apply_to_nchw(const float* g, const float* f8, const float* A, const float* Bc, const float* Cc,
                             float* out, int B, int P, int act_bf16, const BnFin* fin, hipStream_t st);
int ava_bn_finalize_bwd_ex(const float* partials, int nparts, int64_t n, int C, const float* gamma, const float* mean,
                           const float* invstd, float* dgamma, float* dbeta, float* A, float* Bc, float* Cc, int eval,
                           hipStream_t st);
int ava_latent_bwd_scaled(const float* z, const float* dz_dec, const float* u, const float* d, const float* eps_w,
                          const float* eps_d, float* dmu, float* du, float* dlogd, int B, int zdim, const float* scale,
                          hipStream_t st);
int ava_scale_backward_roots(float* seed, int64_t n, float* wg, int64_t nwg, long long* slot, const float* scale, hipStream_t st);
int ava_adam_flat_guarded(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                          double eps, int step, const int* skip_if_set, hipStream_t st);
int ava_elbo_finalize_strided(const float* latent_sums, int B, const float* sse_partials, int nparts, int stride,
                              int zdim, float prec, int xdim, float* loss_out, double* loss_accum, hipStream_t st);

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_EXP = 2 };

// fc_mid.hip: fc31|32|33 -> heads -> rsample -> fc5 -> fc6 (and the mirror data-gradient chain) as one launch each
struct FcMidFwdArgs {
  const float* h3_in;
  const float *W3, *b3;
  const float *W41, *b41, *W42, *b42, *W43, *b43;
  const float *W5, *b5, *W6, *b6;
  const float *eps_w, *eps_d;
  float *h3, *mu, *u, *logd, *d, *z, *lat_sums, *h5, *h6;
  int* status;
  int B, zdim;
  unsigned long long* stamps;
};
struct FcMidBwdArgs {
  const float* dh6;
  const float *W6, *W5, *W41, *W42, *W43, *W3;
  const float *h5, *h3, *h2;
  const float *z, *u, *d, *eps_w, *eps_d;
  const float* scale;
  float *dh5, *dz, *dmu, *du, *dlogd, *dh3, *dh2;
  int B, zdim;
  unsigned long long* stamps;
};
int ava_fc_mid_fwd(const FcMidFwdArgs& a, hipStream_t st);
int ava_fc_mid_bwd(const FcMidBwdArgs& a, hipStream_t st);

#define NCONV 14
#define NPARAM 80
#define ALIGN_F 64

// ---- optional per-category timing with HIP events on the launch stream (bench.py roofline leg) ----
enum { CAT_CONV_FWD = 0, CAT_CONV_BWD_DATA, CAT_CONV_WGRAD, CAT_BN, CAT_GEMM, CAT_LAYOUT, CAT_LATENT_LOSS, CAT_ADAM,
       CAT_PACK, NCAT };
#define PROF_MAX_EVENTS 1024
struct Prof {
  bool on = false;
  int mode = 0;                   // 1: an event after every launch group; 2: only where the kernel FAMILY changes
  int n = 0;
  hipEvent_t ev[PROF_MAX_EVENTS];
  int cat[PROF_MAX_EVENTS];
  int created = 0;
  // family (conv+BatchNorm+pack = 0, everything else = 1, call begin = -1) of every mark of a step, learnt in mode 1
  int seq_fam[PROF_MAX_EVENTS];
  int seq_n = 0;
  int idx = 0;                    // position in seq_fam of the next mark (mode 2)
};
static inline int prof_family(int cat) {
  if (cat < 0) return -1;
  return (cat == CAT_CONV_FWD || cat == CAT_CONV_BWD_DATA || cat == CAT_CONV_WGRAD || cat == CAT_BN || cat == CAT_PACK) ? 0 : 1;
}

struct ConvLayer {
  int cin, cout, mode;            // forward gather mode
  int hi, ho;                     // input / output height at the reference's 128 x 128 (kLayers); the model's own
                                  // per-layer sizes are these scaled by H/128 and W/128 (ava_model::lay)
  int transposed;                 // 0: Conv2d, 1: ConvTranspose2d
  int pw, pb, pg, pbeta;          // parameter indices: weight, bias, bn gamma, bn beta
};
struct LayerDims { int hi, wi, ho, wo; };

// named_parameters() order (vae.py:125-168): conv1..7 (w,b) 0..13 ; bn1..7 (w,b) 14..27 ; fc1,fc2,fc31,fc32,fc33,
// fc41,fc42,fc43,fc5..fc8 (w,b) 28..51 ; convt1..7 (w,b) 52..65 ; bn8..14 (w,b) 66..79
static const ConvLayer kLayers[NCONV] = {
    {1, 8, MODE_S1, 128, 128, 0, 0, 1, 14, 15},    {8, 8, MODE_DOWN, 128, 64, 0, 2, 3, 16, 17},
    {8, 16, MODE_S1, 64, 64, 0, 4, 5, 18, 19},     {16, 16, MODE_DOWN, 64, 32, 0, 6, 7, 20, 21},
    {16, 24, MODE_S1, 32, 32, 0, 8, 9, 22, 23},    {24, 24, MODE_DOWN, 32, 16, 0, 10, 11, 24, 25},
    {24, 32, MODE_S1, 16, 16, 0, 12, 13, 26, 27},  {32, 24, MODE_S1, 16, 16, 1, 52, 53, 66, 67},
    {24, 24, MODE_UP, 16, 32, 1, 54, 55, 68, 69},  {24, 16, MODE_S1, 32, 32, 1, 56, 57, 70, 71},
    {16, 16, MODE_UP, 32, 64, 1, 58, 59, 72, 73},  {16, 8, MODE_S1, 64, 64, 1, 60, 61, 74, 75},
    {8, 8, MODE_UP, 64, 128, 1, 62, 63, 76, 77},   {8, 1, MODE_S1, 128, 128, 1, 64, 65, 78, 79},
};

struct ParamInfo { int64_t off, numel; };

// Arena order: named_parameters() order, except that the three 256->64 head layers are grouped as
// fc31.w fc32.w fc33.w | fc31.b fc32.b fc33.b so that they form one [192,256] matrix and one [192] bias
// (a single GEMM forward, a single pair of GEMMs backward).  Checkpoints are unaffected: the Python
// parameters are views into the arena keyed by name.
static void build_param_table(int z, int F, ParamInfo* tab, int64_t* total) {
  int64_t numel[NPARAM];
  int p = 0;
  const int enc[7][2] = {{1, 8}, {8, 8}, {8, 16}, {16, 16}, {16, 24}, {24, 24}, {24, 32}};
  for (int i = 0; i < 7; ++i) { numel[p++] = (int64_t)enc[i][0] * enc[i][1] * 9; numel[p++] = enc[i][1]; }
  const int bne[7] = {1, 8, 8, 16, 16, 24, 24};
  for (int i = 0; i < 7; ++i) { numel[p++] = bne[i]; numel[p++] = bne[i]; }
  // fc1.in = fc8.out = F = 32 * (H/8) * (W/8): 8192 at the reference's 128 x 128 (vae.py:142,153,224,262)
  const int fc[12][2] = {{F, 1024}, {1024, 256}, {256, 64}, {256, 64}, {256, 64}, {64, z}, {64, z}, {64, z},
                         {z, 64}, {64, 256}, {256, 1024}, {1024, F}};
  for (int i = 0; i < 12; ++i) { numel[p++] = (int64_t)fc[i][0] * fc[i][1]; numel[p++] = fc[i][1]; }
  const int dec[7][2] = {{32, 24}, {24, 24}, {24, 16}, {16, 16}, {16, 8}, {8, 8}, {8, 1}};
  for (int i = 0; i < 7; ++i) { numel[p++] = (int64_t)dec[i][0] * dec[i][1] * 9; numel[p++] = dec[i][1]; }
  const int bnd[7] = {32, 24, 24, 16, 16, 8, 8};
  for (int i = 0; i < 7; ++i) { numel[p++] = bnd[i]; numel[p++] = bnd[i]; }
  int order[NPARAM], n = 0;
  for (int i = 0; i < 32; ++i) order[n++] = i;
  order[n++] = 32; order[n++] = 34; order[n++] = 36;      // fc31.w fc32.w fc33.w
  order[n++] = 33; order[n++] = 35; order[n++] = 37;      // fc31.b fc32.b fc33.b
  for (int i = 38; i < NPARAM; ++i) order[n++] = i;
  int64_t cur = 0;
  for (int k = 0; k < NPARAM; ++k) {
    const int i = order[k];
    tab[i].off = cur;
    tab[i].numel = numel[i];
    cur += (numel[i] + ALIGN_F - 1) / ALIGN_F * ALIGN_F;
  }
  *total = cur;
}

struct ava_model {
  int z, maxB;
  int H, W;                 // spectrogram size (128 x 128 in the reference, vae.py:33); W in {128, 256}, H % 64 == 0
  int P8, F;                // pixels at the bottleneck (H/8 * W/8) and fc1.in = fc8.out = 32 * P8
  int act_bf16;             // 1: the activations between the convolutions (X[1..13]) are stored as bfloat16
  LayerDims lay[NCONV];
  float prec;
  float *P, *G, *M, *V;
  float* bn_running;
  int64_t* bn_batches;
  ParamInfo tab[NPARAM];
  int64_t arena;
  // workspace carving
  float* X[NCONV];          // raw input of conv layer l (X[0] is the caller's x; X[7] = f8 NHWC)
  float *y7, *y7t;          // conv7 output NHWC and NCHW-flatten
  float *h1, *h2, *h3, *mu, *u, *logd, *d, *zs, *lat_sums;
  float *h5, *h6, *h7, *f8;
  float *xrec, *seed;
  float* bn_part;           // [1024][64]
  float* bn_save;           // [14][4][32]: mean, invstd, scale, shift
  float* bn_bwd;            // [14][3][32]: A, Bc, Cc
  long long* bn_acc;        // [29 slots][8 shards][200]: in-kernel BatchNorm sums (bn_acc.h), zeroed by the pack launch
  int acc0_slot;            // slot (0 or 28) the NEXT statistics launch adds bn1's input sums to; -1: neither is known to be zero yet
  int acc0_used;            // slot the last training forward used (conv1's kernel finalises from it)
  float* Gf[NCONV];
  float* Gb[NCONV];
  float *gA, *gB;           // gradient ping-pong, B*131072 floats each
  int dy7_slabs;            // 2: fc1's dX of the last backward sits in dy7_ws as two split-K slabs (summed by part 3's layout kernel)
  float* dy7_ws;            // workspace of that ONE product only: no other launch may reuse it between the backward parts
  size_t dy7_ws_bytes;
  int cu_reserve;           // CUs this model's persistent grids leave free (ava_model_set_cu_reserve); -1: the process-wide setting
  float* wg_part[NCONV];    // wgrad partial rows, one region per layer (reduced in one launch at the end)
  float *dF8, *dh7, *dh6, *dh5, *dz, *dmu, *du, *dlogd, *dh3, *dh2, *dh1, *dy7;
  float* gemm_ws;
  size_t gemm_ws_bytes;
  float* loss_dev;          // 4 floats scratch when the caller passes none
  int* status_dev;
  const float* eps_w_last;
  const float* eps_d_last;  // noise of the last forward (caller-owned; needed again by backward)
  int sse_parts;
  int lastB;                // batch of the forward whose intermediates are in the workspace; 0: none (ava_backward refuses)
  int last_train;           // BatchNorm mode of that forward: backward uses the matching BatchNorm derivative
  int* status_last;         // status word of that forward (ava_adam_step skips the update when it is set)
  const float* bwd_scale;   // device scalar d(result)/d(loss) for the next backward (ava_set_backward_scale); null = 1
  int fold13;               // the last forward left convt7's weight-gradient partials (wg13_rows rows of wg_part[13]) and the
                            // BatchNorm-backward sums of its input behind (conv_thin_kernels.h: FOLD), for loss scale 1
  int wg13_rows;            // rows of wg_part[13] the weight-gradient reduction has to sum
  Prof prof;
  std::map<std::string, std::pair<const float*, int64_t>> dbg;
};

static inline float* bn_mean(ava_model* m, int l) { return m->bn_save + (l * 4 + 0) * 32; }
static inline float* bn_invstd(ava_model* m, int l) { return m->bn_save + (l * 4 + 1) * 32; }
static inline float* bn_scale(ava_model* m, int l) { return m->bn_save + (l * 4 + 2) * 32; }
static inline float* bn_shift(ava_model* m, int l) { return m->bn_save + (l * 4 + 3) * 32; }
static inline float* bn_A(ava_model* m, int l) { return m->bn_bwd + (l * 3 + 0) * 32; }
static inline float* bn_B(ava_model* m, int l) { return m->bn_bwd + (l * 3 + 1) * 32; }
static inline float* bn_C(ava_model* m, int l) { return m->bn_bwd + (l * 3 + 2) * 32; }
static inline float* PP(ava_model* m, int idx) { return m->P + m->tab[idx].off; }
static inline float* GG(ava_model* m, int idx) { return m->G + m->tab[idx].off; }

struct Carver {
  char* base;
  size_t off;
  float* take(size_t floats) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += (floats * sizeof(float) + 255) / 256 * 256;
    return p;
  }
};

// largest split-K workspace any product of a step needs, for EVERY batch size up to maxB: the split plans are not monotonic
// in the batch (a 128-row product may run as four slabs where the 256-row one runs as two), and a model created for a large
// batch also runs smaller ones
static size_t max_gemm_ws_at(int z, int B, int F) {
  size_t mx = 0;
  const int shapes[][3] = {{B, 1024, F}, {B, 256, 1024}, {B, 192, 256}, {B, z, 64}, {B, 64, z}, {B, 256, 64},
                           {B, 1024, 256}, {B, F, 1024},
                           // dX products
                           {B, F, 1024}, {B, 1024, 256}, {B, 256, 192}, {B, 64, z}, {B, z, 64}, {B, 64, 256},
                           {B, 256, 1024}, {B, 1024, F},
                           // dW products (K = batch)
                           {F, 1024, B}, {1024, 256, B}, {256, 64, B}, {64, z, B}, {z, 64, B}, {192, 256, B},
                           {256, 1024, B}, {1024, F, B}};
  for (auto& s : shapes) {
    size_t b = ava_gemm_workspace_bytes(s[0], s[1], s[2]);
    if (b > mx) mx = b;
  }
  return mx;
}
static size_t max_gemm_ws(int z, int maxB, int F) {
  size_t mx = 0;
  for (int b = 1; b <= maxB; ++b) { const size_t v = max_gemm_ws_at(z, b, F); if (v > mx) mx = v; }
  return mx;
}
static size_t max_dy7_ws(int maxB, int F) {
  size_t mx = 0;
  for (int b = 1; b <= maxB; ++b) { const size_t v = ava_gemm_workspace_bytes(b, F, 1024); if (v > mx) mx = v; }
  return mx;
}

static size_t wgrad_part_floats(const ava_model* m, int B, int l) {
  const ConvLayer& L = kLayers[l];
  const LayerDims& D = m->lay[l];
  int grid = ava_conv_wgrad_grid(B, D.ho, D.wo, L.mode);
  const int fg = ava_conv_fused_grid_for(B, D.hi, D.wi, L.cin, L.cout, L.mode);     // fused backward kernel's rows
  if (fg > grid) grid = fg;
  if (l == NCONV - 1) {            // convt7: its training forward writes the partial rows (one per row of its SSE partials)
    const int fw = ava_conv_grid(B, D.ho, D.wo, L.mode);
    if (fw > grid) grid = fw;
  }
  return (size_t)grid * (9 * L.cin * L.cout + L.cout);
}

static void carve(ava_model* m, void* ws, size_t* total) {
  Carver c{reinterpret_cast<char*>(ws), 0};
  const size_t B = (size_t)m->maxB;
  const int z = m->z;
  const size_t F = (size_t)m->F, XD = (size_t)m->H * m->W;
  m->X[0] = nullptr;
  for (int l = 1; l < NCONV; ++l) {
    const ConvLayer& L = kLayers[l];
    if (l == 7) m->X[l] = c.take(B * F);
    else m->X[l] = c.take(B * m->lay[l].hi * m->lay[l].wi * L.cin);
  }
  m->y7 = c.take(B * F); m->y7t = c.take(B * F);
  m->h1 = c.take(B * 1024); m->h2 = c.take(B * 256); m->h3 = c.take(B * 192);
  m->mu = c.take(B * z); m->u = c.take(B * z); m->logd = c.take(B * z); m->d = c.take(B * z); m->zs = c.take(B * z);
  m->lat_sums = c.take(B * 2);
  m->h5 = c.take(B * 64); m->h6 = c.take(B * 256); m->h7 = c.take(B * 1024); m->f8 = c.take(B * F);
  m->xrec = c.take(B * XD); m->seed = c.take(B * XD);
  m->bn_part = c.take(1024 * 64);
  m->bn_save = c.take(NCONV * 4 * 32);
  m->bn_bwd = c.take(NCONV * 3 * 32);
  m->bn_acc = reinterpret_cast<long long*>(c.take((size_t)AVA_ACC_SLOTS * AVA_ACC_SLOT_LL * 2));
  for (int l = 0; l < NCONV; ++l) {
    m->Gf[l] = c.take(9 * kLayers[l].cin * kLayers[l].cout);
    m->Gb[l] = c.take(9 * kLayers[l].cin * kLayers[l].cout);
  }
  m->gA = c.take(B * XD * 8); m->gB = c.take(B * XD * 8);          // largest activation: 8 channels at full resolution
  for (int l = 0; l < NCONV; ++l) m->wg_part[l] = c.take(wgrad_part_floats(m, (int)B, l));
  m->dF8 = c.take(B * F); m->dh7 = c.take(B * 1024); m->dh6 = c.take(B * 256); m->dh5 = c.take(B * 64);
  m->dz = c.take(B * z); m->dmu = c.take(B * z); m->du = c.take(B * z); m->dlogd = c.take(B * z);
  m->dh3 = c.take(B * 192); m->dh2 = c.take(B * 256); m->dh1 = c.take(B * 1024); m->dy7 = c.take(B * F);
  m->gemm_ws_bytes = max_gemm_ws(z, (int)B, m->F);
  m->gemm_ws = c.take(m->gemm_ws_bytes / sizeof(float) + 64);
  m->dy7_ws_bytes = max_dy7_ws((int)B, m->F);
  m->dy7_ws = c.take(m->dy7_ws_bytes / sizeof(float) + 64);
  m->loss_dev = c.take(64);
  m->status_dev = reinterpret_cast<int*>(c.take(64));
  *total = c.off;
}

extern "C" int ava_version(void) { return 100; }

// CUs left free by every persistent launch (common.h: ava_scale_grid).  The grids are sized deep inside the launchers, which
// read the CURRENT value: a thread-local that every model entry point (forward, backward part, Adam, encode, decode) sets for
// its own duration from the model's setting (ava_model_set_cu_reserve; -1 = the process-wide default below) and restores on
// return (ReserveScope).  So a producer / consumer pair of launches inside one entry point (partial rows written by one
// kernel, counted by the next) always sees one value, two models or two host threads cannot disturb each other, and the
// per-kernel entry points (ava_conv3x3 ..., tests and tools) run under the process-wide default.
static int g_cu_reserve_default = 0;
static thread_local int t_cu_reserve = -1;             // -1: no model entry point is active on this thread
int ava_cu_reserve(void) { return t_cu_reserve >= 0 ? t_cu_reserve : g_cu_reserve_default; }
extern "C" int ava_set_cu_reserve(int cus) {
  if (cus < 0 || cus > 128) return AVA_EINVAL;
  g_cu_reserve_default = cus;
  return AVA_OK;
}
extern "C" int ava_get_cu_reserve(void) { return g_cu_reserve_default; }
struct ReserveScope {
  int saved;
  explicit ReserveScope(const ava_model* m) : saved(t_cu_reserve) {
    if (m != nullptr) t_cu_reserve = m->cu_reserve >= 0 ? m->cu_reserve : g_cu_reserve_default;
  }
  ~ReserveScope() { t_cu_reserve = saved; }
};
extern "C" int ava_model_set_cu_reserve(ava_model* m, int cus) {
  if (m == nullptr || cus < -1 || cus > 128) return AVA_EINVAL;
  m->cu_reserve = cus;
  return AVA_OK;
}
extern "C" int ava_model_get_cu_reserve(const ava_model* m) { return m == nullptr ? -1 : m->cu_reserve; }

// Test / measurement helper: `workgroups` persistent 256-thread workgroups that hold their wave slots for `usec`
// microseconds (s_memrealtime, 100 MHz) and do nothing else -- stands in for a collective's persistent kernel on a side
// stream (tests/test_gpu_reserve.py, tools/reserve_bench.py).
__global__ __launch_bounds__(256) void occupy_kernel(long long ticks, unsigned long long* sink) {
  extern __shared__ unsigned char occupy_lds[];          // dynamic LDS only limits how many of these share a CU
  if (ticks < 0) occupy_lds[threadIdx.x] = 0;
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  long long n = 0;
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); ++n; }
  if (sink != nullptr && threadIdx.x == 0 && blockIdx.x == 0) *sink = (unsigned long long)n;
}
extern "C" int ava_occupy_cus(int workgroups, int lds_bytes, float usec, ava_stream_t s) {
  if (workgroups < 1 || workgroups > 1024 || !(usec > 0.f) || usec > 20000.f || lds_bytes < 0 || lds_bytes > 160 * 1024)
    return AVA_EINVAL;                                   // bounded: never a hang
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess) return AVA_ELAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(256), (size_t)lds_bytes, to_stream(s), (long long)(usec * 100.f), nullptr);
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

// spectrogram sizes the kernels cover: W = 128 or 256 (the full-resolution 1- and 8-channel layers run 2*W-thread
// workgroups that own whole rows), H a multiple of 128 (16-row tiles of the layers at H/8 x W/8)
// Validated sizes only (gradient-vs-oracle tests at 128x128, 256x256, 128x256, 256x128): taller images would reach grid /
// partial-row clamps and 32-bit offset ranges no test exercises.
static bool size_ok(int H, int W) { return (W == 128 || W == 256) && (H == 128 || H == 256); }

static void set_geometry(ava_model* m, int H, int W) {
  m->H = H; m->W = W;
  m->P8 = (H / 8) * (W / 8);
  m->F = 32 * m->P8;
  for (int l = 0; l < NCONV; ++l) {
    const ConvLayer& L = kLayers[l];
    m->lay[l].hi = L.hi * H / 128; m->lay[l].wi = L.hi * W / 128;
    m->lay[l].ho = L.ho * H / 128; m->lay[l].wo = L.ho * W / 128;
  }
}

extern "C" int64_t ava_arena_floats_hw(int z_dim, int H, int W) {
  if (!size_ok(H, W)) return -1;
  ParamInfo tab[NPARAM];
  int64_t total;
  build_param_table(z_dim, 32 * (H / 8) * (W / 8), tab, &total);
  return total;
}
extern "C" int64_t ava_param_offset_hw(int z_dim, int H, int W, int index, int64_t* numel) {
  if (index < 0 || index >= NPARAM || !size_ok(H, W)) return -1;
  ParamInfo tab[NPARAM];
  int64_t total;
  build_param_table(z_dim, 32 * (H / 8) * (W / 8), tab, &total);
  if (numel) *numel = tab[index].numel;
  return tab[index].off;
}
extern "C" size_t ava_workspace_bytes_hw(int z_dim, int H, int W, int max_batch) {
  if (!size_ok(H, W)) return 0;
  ava_model tmp;
  tmp.z = z_dim;
  tmp.maxB = max_batch;
  tmp.act_bf16 = 0;
  set_geometry(&tmp, H, W);
  size_t total = 0;
  carve(&tmp, nullptr, &total);
  return total;
}
extern "C" int64_t ava_arena_floats(int z_dim) { return ava_arena_floats_hw(z_dim, 128, 128); }
extern "C" int64_t ava_param_offset(int z_dim, int index, int64_t* numel) { return ava_param_offset_hw(z_dim, 128, 128, index, numel); }
extern "C" size_t ava_workspace_bytes(int z_dim, int max_batch) { return ava_workspace_bytes_hw(z_dim, 128, 128, max_batch); }

extern "C" int ava_model_create_ex(ava_model** out, int z_dim, int H, int W, int act_dtype, int max_batch,
                                   float model_precision, float* params, float* grads, float* exp_avg,
                                   float* exp_avg_sq, float* bn_running, int64_t* bn_batches, void* workspace,
                                   size_t workspace_bytes) {
  if (out == nullptr || z_dim < 1 || z_dim > 128 || max_batch < 1 || params == nullptr || workspace == nullptr ||
      !size_ok(H, W) || (act_dtype != 0 && act_dtype != 1))
    return AVA_EINVAL;
  ava_model* m = new ava_model();
  m->z = z_dim; m->maxB = max_batch; m->prec = model_precision;
  m->act_bf16 = act_dtype;
  set_geometry(m, H, W);
  m->P = params; m->G = grads; m->M = exp_avg; m->V = exp_avg_sq;
  m->bn_running = bn_running; m->bn_batches = bn_batches;
  build_param_table(z_dim, m->F, m->tab, &m->arena);
  size_t need = 0;
  carve(m, workspace, &need);
  if (need > workspace_bytes) { delete m; return AVA_EWORKSPACE; }
  m->lastB = 0;
  m->last_train = 1;
  m->cu_reserve = -1;
  m->dy7_slabs = 1;
  m->acc0_slot = -1; m->acc0_used = -1;
  m->status_last = nullptr;
  m->bwd_scale = nullptr;
  m->sse_parts = 0;
  m->fold13 = 0;
  m->wg13_rows = 0;
  const size_t B = max_batch;
  const int64_t F = m->F, XD = (int64_t)H * W;
  for (int l = 1; l < NCONV; ++l) {
    char nm[16];
    if (l < 7) { snprintf(nm, sizeof nm, "y%d", l); }
    else if (l == 7) { snprintf(nm, sizeof nm, "f8t"); }
    else { snprintf(nm, sizeof nm, "d%d", l - 7); }
    m->dbg[nm] = {m->X[l], (int64_t)(l == 7 ? B * F : B * m->lay[l].hi * m->lay[l].wi * kLayers[l].cin)};
  }
  m->dbg["y7"] = {m->y7, (int64_t)B * F}; m->dbg["y7t"] = {m->y7t, (int64_t)B * F};
  m->dbg["h1"] = {m->h1, (int64_t)B * 1024}; m->dbg["h2"] = {m->h2, (int64_t)B * 256};
  m->dbg["h3"] = {m->h3, (int64_t)B * 192};
  m->dbg["mu"] = {m->mu, (int64_t)B * z_dim}; m->dbg["u"] = {m->u, (int64_t)B * z_dim};
  m->dbg["logd"] = {m->logd, (int64_t)B * z_dim}; m->dbg["d"] = {m->d, (int64_t)B * z_dim};
  m->dbg["z"] = {m->zs, (int64_t)B * z_dim};
  m->dbg["h5"] = {m->h5, (int64_t)B * 64}; m->dbg["h6"] = {m->h6, (int64_t)B * 256};
  m->dbg["h7"] = {m->h7, (int64_t)B * 1024}; m->dbg["f8"] = {m->f8, (int64_t)B * F};
  m->dbg["xrec"] = {m->xrec, (int64_t)B * XD}; m->dbg["seed"] = {m->seed, (int64_t)B * XD};
  m->dbg["bn_save"] = {m->bn_save, NCONV * 4 * 32}; m->dbg["bn_bwd"] = {m->bn_bwd, NCONV * 3 * 32};
  m->dbg["dz"] = {m->dz, (int64_t)B * z_dim}; m->dbg["dF8"] = {m->dF8, (int64_t)B * F};
  m->dbg["wg13"] = {m->wg_part[NCONV - 1], (int64_t)wgrad_part_floats(m, (int)B, NCONV - 1)};   // convt7's weight-gradient partial rows
  // (no "dy7" entry: fc1's data gradient usually stays as two split-K slabs in dy7_ws, summed by the encoder's first kernel)
  *out = m;
  return AVA_OK;
}
extern "C" int ava_model_create_hw(ava_model** out, int z_dim, int H, int W, int max_batch, float model_precision,
                                   float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* bn_running,
                                   int64_t* bn_batches, void* workspace, size_t workspace_bytes) {
  return ava_model_create_ex(out, z_dim, H, W, 0, max_batch, model_precision, params, grads, exp_avg, exp_avg_sq,
                             bn_running, bn_batches, workspace, workspace_bytes);
}
extern "C" int ava_model_create(ava_model** out, int z_dim, int max_batch, float model_precision, float* params,
                                float* grads, float* exp_avg, float* exp_avg_sq, float* bn_running,
                                int64_t* bn_batches, void* workspace, size_t workspace_bytes) {
  return ava_model_create_hw(out, z_dim, 128, 128, max_batch, model_precision, params, grads, exp_avg, exp_avg_sq,
                             bn_running, bn_batches, workspace, workspace_bytes);
}
extern "C" void ava_model_destroy(ava_model* m) { delete m; }
extern "C" const float* ava_last_z(ava_model* m) { return m->zs; }
extern "C" const float* ava_last_xrec(ava_model* m) { return m->xrec; }
extern "C" const float* ava_debug_buffer(ava_model* m, const char* name, int64_t* floats) {
  auto it = m->dbg.find(name);
  if (it == m->dbg.end()) return nullptr;
  if (floats) *floats = it->second.second;
  return it->second.first;
}

// record an event after a launch (group) of category `cat`; the time since the previous event is
// attributed to `cat` by ava_profile_read
static inline void mark(ava_model* m, int cat, hipStream_t st) {
  Prof& p = m->prof;
  if (!p.on || p.n >= PROF_MAX_EVENTS) return;
  if (p.mode == 2) {
    // coarse pass: ~100 event records per step stretch the step by 10-15 %; here an event is recorded only after the
    // last launch group of a run of same-family kernels (sequence learnt by a mode-1 step), ~20 per step
    const int i = p.idx++;
    const bool known = i < p.seq_n && p.seq_fam[i] == prof_family(cat);
    const bool boundary = !known || cat < 0 || i + 1 >= p.seq_n || p.seq_fam[i + 1] != p.seq_fam[i];
    if (!boundary) return;
  } else if (p.n < PROF_MAX_EVENTS) {
    p.seq_fam[p.n] = prof_family(cat);
  }
  if (p.n >= p.created) { if (hipEventCreate(&p.ev[p.created]) != hipSuccess) return; p.created++; }
  hipEventRecord(p.ev[p.n], st);
  p.cat[p.n] = cat;
  p.n++;
}

extern "C" int ava_profile_enable(ava_model* m, int on) {
  if (m == nullptr) return AVA_EINVAL;
  m->prof.on = on != 0;
  m->prof.mode = on == 2 ? 2 : 1;
  m->prof.n = 0;
  m->prof.idx = 0;
  return AVA_OK;
}
// synchronises on the last event, sums elapsed ms per category into ms[NCAT] (adds to it), returns the
// number of categories; resets the event list
extern "C" int ava_profile_read(ava_model* m, float* ms, int* launches) {
  if (m == nullptr || ms == nullptr) return AVA_EINVAL;
  Prof& p = m->prof;
  if (p.n > 0) hipEventSynchronize(p.ev[p.n - 1]);
  for (int i = 1; i < p.n; ++i) {
    if (p.cat[i] < 0) continue;           // step-begin marker
    float e = 0.f;
    if (hipEventElapsedTime(&e, p.ev[i - 1], p.ev[i]) == hipSuccess) {
      ms[p.cat[i]] += e;
      if (launches) launches[p.cat[i]] += 1;
    }
  }
  if (p.mode == 1 && p.n > 0) p.seq_n = p.n;      // the family sequence of one whole step, for the coarse mode
  p.n = 0;
  p.idx = 0;
  return NCAT;
}

#define TRY(expr)            \
  do {                       \
    int _rc = (expr);        \
    if (_rc != AVA_OK) return _rc; \
  } while (0)

// ---- all 28 weight tables in one launch ---------------------------------------------------------------
struct PackEntry { const float* w; float* g; int c0, c1, swap, flip; };
struct PackTable { PackEntry e[2 * NCONV]; long long* acc; int nacc; int keep0, keep1; long long* in_acc; };
// acc: BatchNorm accumulators to zero (bn_acc.h), except elements [keep0, keep1): the slot `in_acc` the statistics blocks of
// this very launch add bn1's input sums to (zeroed by the previous launch: two slots alternate)
__global__ void pack_all_kernel(const PackTable tab) {
  {
    const int gt = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x, nt = gridDim.x * gridDim.y * blockDim.x;
    for (int i = gt; i < tab.nacc; i += nt) tab.acc[i] = 0;      // (no statistics in this launch: nothing to keep)
  }
  const PackEntry e = tab.e[blockIdx.y];
  const int n = e.c0 * e.c1 * 9;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int tt = i % 9, r = i / 9, i1 = r % e.c1, i0 = r / e.c1;
    const int tg = e.flip ? 8 - tt : tt;
    const int gi = e.swap ? (tg * e.c1 + i1) * e.c0 + i0 : (tg * e.c0 + i0) * e.c1 + i1;
    e.g[gi] = e.w[i];
  }
}
// pack_all_kernel plus the statistics of the raw 1-channel input (bn1: its input has no producer kernel) in ONE launch:
// blocks [0, nstats) sum x and x^2 exactly like bn_stats_kernel<1> (same grid-stride order, same partial rows), the
// rest pack the weight tables.  Both are independent first kernels of a training forward; one launch less per step.
__global__ __launch_bounds__(256) void pack_stats_kernel(const PackTable tab, const float* __restrict__ x, int64_t n,
                                                         float* __restrict__ partials, int nstats, float* __restrict__ eps,
                                                         int64_t neps, uint64_t seed, uint64_t offset) {
  constexpr int NPACK = 7 * 2 * NCONV;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < tab.nacc; i += gridDim.x * 256)                 // every block first
    if (i < tab.keep0 || i >= tab.keep1) tab.acc[i] = 0;
  if ((int)blockIdx.x >= nstats + NPACK) {              // third role: the step's rsample noise (ava_forward_noise)
    const int nb = gridDim.x - nstats - NPACK, bi = blockIdx.x - nstats - NPACK;
    for (int64_t i = (int64_t)bi * 256 + threadIdx.x; i < neps; i += (int64_t)nb * 256)
      eps[i] = ava_normal_hash((uint64_t)i + offset, seed);
    return;
  }
  if ((int)blockIdx.x >= nstats) {
    const int idx = blockIdx.x - nstats, by = idx / 7, bx = idx - 7 * by;
    const PackEntry e = tab.e[by];
    const int cnt = e.c0 * e.c1 * 9;
    for (int i = bx * 256 + threadIdx.x; i < cnt; i += 7 * 256) {
      const int tt = i % 9, r = i / 9, i1 = r % e.c1, i0 = r / e.c1;
      const int tg = e.flip ? 8 - tt : tt;
      const int gi = e.swap ? (tg * e.c1 + i1) * e.c0 + i0 : (tg * e.c0 + i0) * e.c1 + i1;
      e.g[gi] = e.w[i];
    }
    return;
  }
  __shared__ float red[4][2];
  float s1 = 0.f, s2 = 0.f;
  const int64_t stride = (int64_t)nstats * 256, n4 = n / 4;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 v = x4[i];
    s1 += (v.x + v.y) + (v.z + v.w);
    s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t i = n4 * 4; i < n; ++i) { s1 += x[i]; s2 += x[i] * x[i]; }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const float r1 = wave_sum(s1), r2 = wave_sum(s2);
  if (l == 0) { red[w][0] = r1; red[w][1] = r2; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const float tot = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (tab.in_acc != nullptr) bn_acc_add(tab.in_acc, 32 * threadIdx.x, tot);     // one channel: values 0 (sum x) and 32 (sum x^2)
    else partials[(size_t)blockIdx.x * 2 + threadIdx.x] = tot;
  }
}

static bool acc_enabled(bool bwd);
static long long* acc_slot(ava_model* m, int slot);
// x_stats != nullptr: also the bn1 input statistics of x_stats[n] (training forward); *nstats_out = partial rows written
struct NoiseGen { float* eps; int64_t n; uint64_t seed, offset; };   // eps == nullptr: no noise to generate

static int pack_weights(ava_model* m, bool with_bwd, hipStream_t st, const float* x_stats = nullptr, int64_t n = 0,
                        int* nstats_out = nullptr, NoiseGen ng = NoiseGen{nullptr, 0, 0, 0}) {
  PackTable tab;
  for (int l = 0; l < NCONV; ++l) {
    const ConvLayer& L = kLayers[l];
    const float* w = PP(m, L.pw);
    // stored dims: Conv2d [cout][cin][9] ; ConvTranspose2d [cin][cout][9]
    const int c0 = L.transposed ? L.cin : L.cout, c1 = L.transposed ? L.cout : L.cin;
    int kf, kb;
    if (!L.transposed) { kf = 0; kb = L.mode == MODE_S1 ? 3 : 4; }
    else { kf = L.mode == MODE_S1 ? 1 : 2; kb = L.mode == MODE_S1 ? 5 : 6; }
    tab.e[2 * l] = {w, m->Gf[l], c0, c1, (kf == 0) ? 1 : 0, (kf == 1) ? 1 : 0};
    tab.e[2 * l + 1] = {w, m->Gb[l], c0, c1, (kb == 5 || kb == 6) ? 1 : 0, (kb == 3) ? 1 : 0};
  }
  tab.acc = m->bn_acc;
  tab.nacc = AVA_ACC_SLOTS * AVA_ACC_SLOT_LL;
  tab.keep0 = tab.keep1 = 0; tab.in_acc = nullptr;
  m->acc0_used = -1;
  mark(m, -1, st);
  static const bool fuse = [] { const char* e = ava_env("AVA_PACK_STATS"); return e == nullptr || atoi(e) != 0; }();
  if (x_stats != nullptr && nstats_out != nullptr && fuse && (reinterpret_cast<uintptr_t>(x_stats) & 15) == 0) {
    int64_t work = n / 4;                                  // same grid rule as ava_bn_stats (C = 1)
    int nstats = (int)((work + 256 * 8 - 1) / (256 * 8));
    if (nstats < 1) nstats = 1;
    if (nstats > 1024) nstats = 1024;
    int nnoise = ng.eps != nullptr ? (int)((ng.n + 255) / 256) : 0;
    if (nnoise > 64) nnoise = 64;
    // under stream capture the alternation would be frozen into the graph (every replay adding to the same slot): a
    // captured step keeps bn1's finalisation launch, and the slot state is unknown afterwards
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    if (capturing) m->acc0_slot = -1;
    if (acc_enabled(false) && m->acc0_slot >= 0) {      // bn1's sums into the slot the previous pack launch zeroed
      tab.in_acc = acc_slot(m, m->acc0_slot);
      tab.keep0 = m->acc0_slot * AVA_ACC_SLOT_LL; tab.keep1 = tab.keep0 + AVA_ACC_SLOT_LL;
      m->acc0_used = m->acc0_slot;
      m->acc0_slot = m->acc0_slot == 0 ? 28 : 0;        // this launch zeroes the other one for the next step
    } else {
      m->acc0_slot = capturing ? -1 : 0;                // this launch zeroes every slot: usable from the next (eager) step on
    }
    hipLaunchKernelGGL(pack_stats_kernel, dim3(nstats + 7 * 2 * NCONV + nnoise), dim3(256), 0, st, tab, x_stats, n, m->bn_part,
                       nstats, ng.eps, ng.n, ng.seed, ng.offset);
    AVA_CHECK_LAUNCH();
    *nstats_out = nstats;
    mark(m, CAT_PACK, st);
    return AVA_OK;
  }
  if (nstats_out != nullptr) *nstats_out = 0;
  if (ng.eps != nullptr) {                                // not fused: the noise gets its own launch
    TRY(ava_fill_normal(ng.eps, ng.n, ng.seed, ng.offset, reinterpret_cast<ava_stream_t>(st)));
    mark(m, CAT_LATENT_LOSS, st);
  }
  hipLaunchKernelGGL(pack_all_kernel, dim3(7, with_bwd ? 2 * NCONV : 2 * NCONV), dim3(256), 0, st, tab);
  AVA_CHECK_LAUNCH();
  {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    m->acc0_slot = capturing ? -1 : 0;                    // every slot zeroed (when the launch really runs now)
  }
  mark(m, CAT_PACK, st);
  return AVA_OK;
}

#ifdef AVA_LAB
// what-if probe (results are WRONG): AVA_SKIP_BN_FIN=1 drops the BatchNorm finalisation launches (the coefficient
// buffers keep the values of the last full step), bounding what removing them from the critical path can gain
static bool lab_skip_bn_fin() {
  static const int v = [] { const char* e = ava_env("AVA_SKIP_BN_FIN"); return e ? atoi(e) : 0; }();
  static int calls = 0;
  return v != 0 && ++calls > 28 * 30;            // the first 30 steps run normally (finite coefficients)
}
#else
static constexpr bool lab_skip_bn_fin() { return false; }
#endif

// ---- BatchNorm sums accumulated in the producing kernel and finalised in the consumer's prologue (bn_acc.h) ----------
// Forward: BatchNorm j (input of layer j) when layer j runs the wave-specialised / plain matrix-core forward kernel (the
// consumer side) and its input comes from such a kernel, from conv1's packed-FMA kernel (j = 1) or from the
// fc8 -> NHWC layout kernel (j = 7), and the direct convt6 / convt7 kernels (j = 12, 13): j = 1..13.  Backward: BatchNorm j when the backward of layer j (fused kernel; j = 5:
// the wave-specialised data-gradient kernel; j = 13: convt7's weight-gradient + sums kernel) hands over to the fused
// backward, or to the data-gradient kernel, of layer j-1 (j = 7: to bn8's layout kernel; j = 1: to conv1's packed-FMA backward;
// j = 0: to the weight-gradient reduction that ends the pass): j = 13 .. 0.  Forward bn1's sums come from the pack launch,
// which also zeroes the accumulators: two slots (0 and 28) alternate, the launch adds to the one its predecessor zeroed.
static bool acc_enabled(bool bwd) {
#ifdef AVA_LAB
  static const int on = [] {
    const char* e = ava_env("AVA_BN_ACC");
    if (e != nullptr) return atoi(e);
    // any kernel-selection switch may route a layer to a kernel without the accumulator hooks: finalisation launches then
    static const char* const sel[] = {"AVA_CONV_FUSED", "AVA_FUSED_WS", "AVA_CONV_IMPL", "AVA_CONV_WS", "AVA_CONV_WS_BWD",
                                      "AVA_THIN_WS", "AVA_THIN_FWD_DIRECT", "AVA_THIN_STATS_DIRECT", "AVA_UP88_DIRECT",
                                      "AVA_PACK_STATS"};
    for (const char* n : sel)
      if (ava_env(n) != nullptr) return 0;
    return 1;
  }();
  (void)bwd;
  return on != 0;
#else
  (void)bwd;
  return true;
#endif
}
static bool acc_pair_fwd(const ava_model* m, int j) {
  (void)m;
  return acc_enabled(false) && ((j >= 1 && j <= 6) || (j >= 7 && j <= 13));
}
static bool acc_pair_bwd(const ava_model* m, int j) {
  (void)m;
  return acc_enabled(true) && (j >= 0 && j <= 13);
}
static long long* acc_slot(ava_model* m, int slot) { return m->bn_acc + (size_t)slot * AVA_ACC_SLOT_LL; }
static BnFin fin_none() { BnFin f = {}; f.acc = nullptr; return f; }
static BnFin fin_fwd(ava_model* m, int j, int B) {
  const ConvLayer& L = kLayers[j];
  BnFin f = {};
  f.acc = acc_slot(m, j);
  f.gamma = PP(m, L.pg); f.beta = PP(m, L.pbeta);
  f.save = m->bn_save + (size_t)j * 4 * 32;
  f.running_mean = m->bn_running + j * 32;
  f.running_var = m->bn_running + (NCONV + j) * 32;
  f.num_batches = reinterpret_cast<long long*>(m->bn_batches + j);
  f.n = (double)B * m->lay[j].hi * m->lay[j].wi;
  f.C = L.cin; f.backward = 0; f.eval = 0;
  return f;
}
static BnFin fin_bwd(ava_model* m, int j, int B) {
  const ConvLayer& L = kLayers[j];
  BnFin f = {};
  f.acc = acc_slot(m, 14 + j);
  f.gamma = PP(m, L.pg);
  f.mean = bn_mean(m, j); f.invstd = bn_invstd(m, j);
  f.dgamma = GG(m, L.pg); f.dbeta = GG(m, L.pbeta);
  f.abc = m->bn_bwd + (size_t)j * 3 * 32;
  f.n = (double)B * m->lay[j].hi * m->lay[j].wi;
  f.C = L.cin; f.backward = 1; f.eval = m->last_train ? 0 : 1;
  return f;
}

static int finalize_fwd(ava_model* m, int l, int nparts, int64_t n, hipStream_t st) {
  if (lab_skip_bn_fin()) return AVA_OK;
  const ConvLayer& L = kLayers[l];
  const int rc = ava_bn_finalize(m->bn_part, nparts, n, L.cin, PP(m, L.pg), PP(m, L.pbeta), m->bn_running + l * 32,
                                 m->bn_running + (NCONV + l) * 32, m->bn_batches + l, 1, bn_mean(m, l), bn_invstd(m, l),
                                 bn_scale(m, l), bn_shift(m, l), st);
  mark(m, CAT_BN, st);
  return rc;
}
static int finalize_bwd(ava_model* m, int l, int nparts, int64_t n, hipStream_t st) {
  if (lab_skip_bn_fin()) return AVA_OK;
  const ConvLayer& L = kLayers[l];
  // a forward in eval mode normalised with the running statistics: they are constants, so dx = gamma*invstd*g
  // (no batch-statistic terms); dgamma / dbeta keep their forms with xhat built from the running statistics
  const int rc = ava_bn_finalize_bwd_ex(m->bn_part, nparts, n, L.cin, PP(m, L.pg), bn_mean(m, l), bn_invstd(m, l),
                                        GG(m, L.pg), GG(m, L.pbeta), bn_A(m, l), bn_B(m, l), bn_C(m, l),
                                        m->last_train ? 0 : 1, st);
  mark(m, CAT_BN, st);
  return rc;
}
static int bn_eval_all(ava_model* m, hipStream_t st) {
  const float* gamma[NCONV]; const float* beta[NCONV]; int C[NCONV];
  for (int l = 0; l < NCONV; ++l) { gamma[l] = PP(m, kLayers[l].pg); beta[l] = PP(m, kLayers[l].pbeta); C[l] = kLayers[l].cin; }
  const int rc = ava_bn_eval_all(gamma, beta, C, m->bn_running, m->bn_save, st);
  mark(m, CAT_BN, st);
  return rc;
}

static int gemm(ava_model* m, const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc,
                const float* mask, float* colsum, int M, int N, int K, int ak, int bk, int act, hipStream_t st) {
  struct Mk { ava_model* m; hipStream_t st; ~Mk() { mark(m, CAT_GEMM, st); } } _mk{m, st};
  return ava_gemm(A, lda, B, ldb, bias, C, ldc, mask, colsum, M, N, K, ak, bk, act, m->gemm_ws, m->gemm_ws_bytes, st);
}

// the product with its split-K reduce left to the consumer when it runs as two slabs (gemm.hip: ava_gemm_defer2)
static int gemm_defer2(ava_model* m, const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc,
                       int M, int N, int K, int ak, int bk, int act, hipStream_t st, int* slabs) {
  struct Mk { ava_model* m; hipStream_t st; ~Mk() { mark(m, CAT_GEMM, st); } } _mk{m, st};
  return ava_gemm_defer2(A, lda, B, ldb, bias, C, ldc, nullptr, nullptr, M, N, K, ak, bk, act, m->gemm_ws, m->gemm_ws_bytes, st, slabs);
}

struct AvaGemmProblem {
  const float* A; int lda; const float* B; int ldb; const float* bias; float* C; int ldc; const float* mask;
  float* colsum; int M, N, K; int act;
};
int ava_gemm_grouped(const AvaGemmProblem* p, int n, int a_kmajor, int b_kmajor, hipStream_t st);
static int gemm_group(ava_model* m, const AvaGemmProblem* p, int n, int ak, int bk, hipStream_t st) {
  const int rc = ava_gemm_grouped(p, n, ak, bk, st);
  mark(m, CAT_GEMM, st);
  return rc;
}

// parameter indices of the fully connected layers
enum { FC1 = 28, FC2 = 30, FC31 = 32, FC32 = 34, FC33 = 36, FC41 = 38, FC42 = 40, FC43 = 42, FC5 = 44, FC6 = 46,
       FC7 = 48, FC8 = 50 };

// the version-0 (VALU) conv kernels have no NCHW second output: keep the transpose launch for them
static bool conv7_writes_nchw() {
  static const bool on = [] { const char* e = ava_env("AVA_CONV_IMPL"); return !(e != nullptr && strcmp(e, "valu") == 0); }();
  return on;
}

// y1 recomputed instead of stored (conv_recomp.h).  Measured (profiles/r03): correct (bit-identical y1, all step tests
// green) but SLOWER in this VALU form -- conv2's forward 44 -> 67 us (the staging waves' 9-tap recomputation costs ~40 us of
// vector issue chip-wide) and the store-free conv1 pass takes the 36 us of the storing one (that kernel is bound by its
// LDS-staged compute chain, not by its 128 MiB of stores).  Lab switch only (AVA_RECOMP_Y1=1); the product stores y1.
static bool recomp_y1() {
  static const bool on = [] { const char* e = ava_env("AVA_RECOMP_Y1"); return e != nullptr && atoi(e) != 0; }();
  return on;
}
static RecompArgs recomp_args(ava_model* m) {
  RecompArgs rc;
  rc.G1 = m->Gf[0]; rc.bias1 = PP(m, kLayers[0].pb); rc.pa1 = bn_scale(m, 0); rc.pb1 = bn_shift(m, 0);
  return rc;
}
// Writes y1 into its workspace slot with the kernel (and therefore the arithmetic) whose store-free form took its
// statistics: for debugging / the tests' mask read-back, and for backward kernels that still read the tensor.
static int materialize_y1(ava_model* m, const float* x, int B, hipStream_t st) {
  const ConvLayer& L = kLayers[0];
  const LayerDims& D = m->lay[0];
  ConvAcc acc;
  acc.fin = fin_none(); acc.acc_out = nullptr;
  TRY(ava_conv3x3_ex(x, nullptr, bn_scale(m, 0), bn_shift(m, 0), nullptr, m->Gf[0], PP(m, L.pb), m->X[1], nullptr, nullptr,
                     nullptr, nullptr, m->bn_part, B, D.hi, D.wi, L.cin, L.cout, L.mode, PRO_BN, EPI_FWD, 1, 0.f, m->act_bf16,
                     &acc, reinterpret_cast<ava_stream_t>(st)));
  mark(m, CAT_CONV_FWD, st);
  return AVA_OK;
}
extern "C" int ava_debug_materialize(ava_model* m, const float* x, int B, ava_stream_t s) {
  if (m == nullptr || x == nullptr || B < 1 || B > m->maxB) return AVA_EINVAL;
  if (!recomp_y1()) return AVA_OK;
  return materialize_y1(m, x, B, to_stream(s));
}

// the small fully connected middle as one launch per direction (fc_mid.hip); lab: AVA_FC_MID=0 keeps the separate launches
static bool fc_mid_on() {
  static const bool on = [] { const char* e = ava_env("AVA_FC_MID"); return e == nullptr || atoi(e) != 0; }();
  return on;
}

// stop_at_fc2 (historic name): stop behind fc31|32|33, the caller continues with the fused middle (forward_impl)
static int encoder_forward(ava_model* m, const float* x, int B, int train, float* mu, float* u, float* logd_or_d,
                           int last_act, hipStream_t st, int pre_nparts = 0, bool stop_at_fc2 = false) {
  const int z = m->z;
  int nparts = pre_nparts;                 // > 0: pack_stats_kernel already wrote the input statistics' partial rows
  if (train) {
    if (nparts <= 0) {
      TRY(ava_bn_stats(x, (int64_t)B * m->H * m->W, 1, m->bn_part, &nparts, reinterpret_cast<ava_stream_t>(st)));
      mark(m, CAT_BN, st);
    }
    if (m->acc0_used < 0) TRY(finalize_fwd(m, 0, nparts, (int64_t)B * m->H * m->W, st));      // else: inside conv1's kernel
  } else {
    TRY(bn_eval_all(m, st));       // all 14 layers from the running statistics (also serves the decoder)
  }
  const bool rc1 = recomp_y1();
  for (int l = 0; l < 7; ++l) {
    const ConvLayer& L = kLayers[l];
    const LayerDims& D = m->lay[l];
    // y1 = relu(conv1(bn1 x)) is never stored (conv_recomp.h): conv1's launch only takes bn2's batch statistics
    // (training; nothing at all on running statistics) and conv2's kernel recomputes its y1 window from x
    if (rc1 && l == 0 && !train) continue;
    const float* in = (l == 0 || (rc1 && l == 1)) ? x : m->X[l];
    float* out = l == 6 ? m->y7 : ((rc1 && l == 0) ? nullptr : m->X[l + 1]);
    // conv7's matrix-core kernel also writes the NCHW-flatten copy fc1 reads (saves the transpose launch)
    float* nchw = (l == 6 && conv7_writes_nchw()) ? m->y7t : nullptr;
    ConvAcc acc;
    acc.fin = (train && acc_pair_fwd(m, l)) ? fin_fwd(m, l, B) : fin_none();                 // BatchNorm l: finalised in this kernel
    if (train && l == 0 && m->acc0_used >= 0) { acc.fin = fin_fwd(m, 0, B); acc.fin.acc = acc_slot(m, m->acc0_used); }
    acc.acc_out = (train && l < 6 && acc_pair_fwd(m, l + 1)) ? acc_slot(m, l + 1) : nullptr;    // BatchNorm l+1: summed by this kernel
    if (rc1 && l == 1) acc.rc = recomp_args(m);
    TRY(ava_conv3x3_ex(in, nullptr, bn_scale(m, l), bn_shift(m, l), nullptr, m->Gf[l], PP(m, L.pb), out, nchw,
                       nullptr, nullptr, nullptr, m->bn_part, B, D.hi, D.wi, L.cin, L.cout, L.mode, PRO_BN, EPI_FWD, 1,
                       0.f, m->act_bf16, &acc, reinterpret_cast<ava_stream_t>(st)));
    mark(m, CAT_CONV_FWD, st);
    if (train && l < 6 && acc.acc_out == nullptr)
      TRY(finalize_fwd(m, l + 1, ava_conv_grid(B, D.ho, D.wo, L.mode), (int64_t)B * D.ho * D.wo, st));
  }
  if (!conv7_writes_nchw()) TRY(ava_nhwc_to_nchw(m->y7, m->y7t, B, m->P8, st));
  mark(m, CAT_LAYOUT, st);
  TRY(gemm(m, m->y7t, 0, PP(m, FC1), 0, PP(m, FC1 + 1), m->h1, 0, nullptr, nullptr, B, 1024, m->F, 1, 1, ACT_RELU, st));
  TRY(gemm(m, m->h1, 0, PP(m, FC2), 0, PP(m, FC2 + 1), m->h2, 0, nullptr, nullptr, B, 256, 1024, 1, 1, ACT_RELU, st));
  // fc31|fc32|fc33 as one [192,256] layer (arena keeps the three weights, then the three biases, contiguous)
  TRY(gemm(m, m->h2, 0, PP(m, FC31), 0, PP(m, FC31 + 1), m->h3, 0, nullptr, nullptr, B, 192, 256, 1, 1, ACT_RELU, st));
  if (stop_at_fc2) return AVA_OK;               // the caller continues with the fused middle (heads .. fc6)
  // the three 64 -> z heads (mu, u, log d) on the 64-wide slices of h3: one grouped launch
  const AvaGemmProblem heads[3] = {
      {m->h3 + 0, 192, PP(m, FC41), 0, PP(m, FC41 + 1), mu, 0, nullptr, nullptr, B, z, 64, ACT_NONE},
      {m->h3 + 64, 192, PP(m, FC42), 0, PP(m, FC42 + 1), u, 0, nullptr, nullptr, B, z, 64, ACT_NONE},
      {m->h3 + 128, 192, PP(m, FC43), 0, PP(m, FC43 + 1), logd_or_d, 0, nullptr, nullptr, B, z, 64, last_act}};
  TRY(gemm_group(m, heads, 3, 1, 1, st));
  return AVA_OK;
}

// from_fc7: h6 already exists (the fused middle wrote it)
static bool dd6_fused(const ava_model* m);
static bool acc_pair_bwd(const ava_model* m, int j);
// convt7's TRAINING forward also forms its weight-gradient partials and the BatchNorm-backward sums of its input
// (conv_thin_kernels.h: FOLD).  Lab build: AVA_FOLD13=0 keeps the separate kernel in the backward.
static bool fold13_on(const ava_model* m) {
  static const bool on = [] { const char* e = ava_env("AVA_FOLD13"); return e == nullptr || atoi(e) != 0; }();
  return on && m->G != nullptr && acc_pair_bwd(m, 13) && m->lay[13].hi % 8 == 0;
}

// `fold`: the caller is a training forward a backward may follow (forward_impl)
static int decoder_forward(ava_model* m, const float* zin, const float* x_target, int B, int train, float* xrec,
                           hipStream_t st, bool from_fc7 = false, bool fold = false) {
  const int z = m->z;
  if (!from_fc7) {
    TRY(gemm(m, zin, 0, PP(m, FC5), 0, PP(m, FC5 + 1), m->h5, 0, nullptr, nullptr, B, 64, z, 1, 1, ACT_RELU, st));
    TRY(gemm(m, m->h5, 0, PP(m, FC6), 0, PP(m, FC6 + 1), m->h6, 0, nullptr, nullptr, B, 256, 64, 1, 1, ACT_RELU, st));
  }
  TRY(gemm(m, m->h6, 0, PP(m, FC7), 0, PP(m, FC7 + 1), m->h7, 0, nullptr, nullptr, B, 1024, 256, 1, 1, ACT_RELU, st));
  // fc8: when the product runs as two split-K slabs, the layout kernel behind it sums them (+ bias, ReLU) on the way in and
  // writes f8 as well -- one launch and one pass over the tensor fewer
  int slabs8 = 1;
  TRY(gemm_defer2(m, m->h7, 0, PP(m, FC8), 0, PP(m, FC8 + 1), m->f8, 0, B, m->F, 1024, 1, 1, ACT_RELU, st, &slabs8));
  int nparts = 0;
  long long* acc7 = (train && acc_pair_fwd(m, 7)) ? acc_slot(m, 7) : nullptr;     // bn8's sums: finalised by convt1's kernel
  if (slabs8 == 2) {
    const float* s0 = reinterpret_cast<const float*>(m->gemm_ws);
    TRY(ava_nchw_to_nhwc_stats_slabs(s0, s0 + (size_t)B * m->F, PP(m, FC8 + 1), 1, m->f8, m->X[7], m->bn_part, B, m->P8,
                                     m->act_bf16, acc7, &nparts, st));
  } else {
    TRY(ava_nchw_to_nhwc_stats(m->f8, m->X[7], m->bn_part, B, m->P8, m->act_bf16, acc7, &nparts, st));
  }
  mark(m, CAT_LAYOUT, st);
  if (train && acc7 == nullptr) TRY(finalize_fwd(m, 7, nparts, (int64_t)B * m->P8, st));
  for (int l = 7; l < NCONV; ++l) {
    const ConvLayer& L = kLayers[l];
    const LayerDims& D = m->lay[l];
    const bool last = l == NCONV - 1;
    float* out = last ? xrec : m->X[l + 1];
    ConvAcc acc;
    acc.fin = (train && acc_pair_fwd(m, l)) ? fin_fwd(m, l, B) : fin_none();
    acc.acc_out = (train && !last && acc_pair_fwd(m, l + 1)) ? acc_slot(m, l + 1) : nullptr;
    if (last && fold && train && x_target != nullptr && fold13_on(m)) {
      acc.fold.wg_partials = m->wg_part[l];
      acc.fold.acc_out = acc_slot(m, 14 + l);
      acc.fold.mean = bn_mean(m, l); acc.fold.invstd = bn_invstd(m, l);
      m->fold13 = 1;
      m->wg13_rows = ava_conv_grid(B, D.ho, D.wo, L.mode);
    }
    TRY(ava_conv3x3_ex(m->X[l], nullptr, bn_scale(m, l), bn_shift(m, l), nullptr, m->Gf[l], PP(m, L.pb), out,
                       last ? m->seed : nullptr, last ? x_target : nullptr, nullptr, nullptr, m->bn_part, B, D.hi, D.wi,
                       L.cin, L.cout, L.mode, PRO_BN, last ? EPI_SSE : EPI_FWD, 1, m->prec, m->act_bf16, &acc,
                       reinterpret_cast<ava_stream_t>(st)));
    mark(m, CAT_CONV_FWD, st);
    if (train && !last && acc.acc_out == nullptr)
      TRY(finalize_fwd(m, l + 1, ava_conv_grid(B, D.ho, D.wo, L.mode), (int64_t)B * D.ho * D.wo, st));
    if (last) m->sse_parts = ava_conv_grid(B, D.ho, D.wo, L.mode);
  }
  return AVA_OK;
}

static int forward_impl(ava_model* m, const float* x, int B, const float* eps_w, const float* eps_d, int bn_train,
                        float* loss_out, double* loss_accum, int* status_out, ava_stream_t s, NoiseGen ng) {
  if (m == nullptr || x == nullptr || eps_w == nullptr || eps_d == nullptr || B < 1 || B > m->maxB) return AVA_EINVAL;
  const ReserveScope rs(m);
  hipStream_t st = to_stream(s);
  const int z = m->z;
  int pre = 0;
  TRY(pack_weights(m, true, st, bn_train ? x : nullptr, (int64_t)B * m->H * m->W, &pre, ng));
  const bool mid = fc_mid_on();
  TRY(encoder_forward(m, x, B, bn_train, m->mu, m->u, m->logd, ACT_NONE, st, pre, mid));
  mark(m, CAT_LAYOUT, st);
  if (mid) {
    // the three heads, rsample + entropy, fc5 and fc6: one launch (16 batch rows per workgroup, fc_mid.hip)
    FcMidFwdArgs fa;
    fa.h3_in = nullptr; fa.W3 = nullptr; fa.b3 = nullptr;
    fa.W41 = PP(m, FC41); fa.b41 = PP(m, FC41 + 1); fa.W42 = PP(m, FC42); fa.b42 = PP(m, FC42 + 1);
    fa.W43 = PP(m, FC43); fa.b43 = PP(m, FC43 + 1);
    fa.W5 = PP(m, FC5); fa.b5 = PP(m, FC5 + 1); fa.W6 = PP(m, FC6); fa.b6 = PP(m, FC6 + 1);
    fa.eps_w = eps_w; fa.eps_d = eps_d;
    fa.h3 = m->h3; fa.mu = m->mu; fa.u = m->u; fa.logd = m->logd; fa.d = m->d; fa.z = m->zs; fa.lat_sums = m->lat_sums;
    fa.h5 = m->h5; fa.h6 = m->h6; fa.status = status_out; fa.B = B; fa.zdim = z;
    TRY(ava_fc_mid_fwd(fa, st));
  } else {
    TRY(ava_latent_fwd(m->mu, m->u, m->logd, eps_w, eps_d, m->d, m->zs, m->lat_sums, status_out, B, z, st));
  }
  m->eps_w_last = eps_w;          // backward reads the same noise: the caller keeps it alive until then
  m->eps_d_last = eps_d;
  mark(m, CAT_LATENT_LOSS, st);
  m->fold13 = 0;
  TRY(decoder_forward(m, m->zs, x, B, bn_train, m->xrec, st, mid, true));
  TRY(ava_elbo_finalize_strided(m->lat_sums, B, m->bn_part, m->sse_parts, 2, z, m->prec, m->H * m->W,
                                loss_out != nullptr ? loss_out : m->loss_dev, loss_accum, st));
  mark(m, CAT_LATENT_LOSS, st);
  m->lastB = B;
  m->last_train = bn_train ? 1 : 0;
  m->status_last = status_out;
  m->bwd_scale = nullptr;
  return AVA_OK;
}

extern "C" int ava_forward(ava_model* m, const float* x, int B, const float* eps_w, const float* eps_d, int bn_train,
                           float* loss_out, double* loss_accum, int* status_out, ava_stream_t s) {
  return forward_impl(m, x, B, eps_w, eps_d, bn_train, loss_out, loss_accum, status_out, s, NoiseGen{nullptr, 0, 0, 0});
}

// ava_forward with the step's rsample noise drawn inside the first launch: eps[0..B) = eps_W, eps[B..B + B*z) = eps_D
// are element i + offset of the counter stream ava_fill_normal(seed) produces (bit-identical to calling it first)
extern "C" int ava_forward_noise(ava_model* m, const float* x, int B, float* eps, uint64_t seed, uint64_t offset,
                                 int bn_train, float* loss_out, double* loss_accum, int* status_out, ava_stream_t s) {
  if (m == nullptr || eps == nullptr || B < 1) return AVA_EINVAL;
  const int64_t n = (int64_t)B * (m->z + 1);
  return forward_impl(m, x, B, eps, eps + B, bn_train, loss_out, loss_accum, status_out, s, NoiseGen{eps, n, seed, offset});
}

extern "C" int ava_encode(ava_model* m, const float* x, int B, int bn_train, float* mu, float* u, float* d,
                          ava_stream_t s) {
  if (m == nullptr || x == nullptr || mu == nullptr || u == nullptr || d == nullptr || B < 1 || B > m->maxB)
    return AVA_EINVAL;
  const ReserveScope rs(m);
  hipStream_t st = to_stream(s);
  m->lastB = 0;                       // the saved activations of the last ava_forward are overwritten
  m->fold13 = 0;
  TRY(pack_weights(m, false, st));
  return encoder_forward(m, x, B, bn_train, mu, u, d, ACT_EXP, st);
}

extern "C" int ava_decode(ava_model* m, const float* z, int B, int bn_train, float* x_rec, ava_stream_t s) {
  if (m == nullptr || z == nullptr || x_rec == nullptr || B < 1 || B > m->maxB) return AVA_EINVAL;
  const ReserveScope rs(m);
  hipStream_t st = to_stream(s);
  m->lastB = 0;
  m->fold13 = 0;
  TRY(pack_weights(m, false, st));
  if (!bn_train) TRY(bn_eval_all(m, st));
  return decoder_forward(m, z, nullptr, B, bn_train, x_rec, st);
}

// ---- all 14 weight-gradient reductions are issued per layer (partials buffer is shared) --------------
// workgroups (= partial rows) of layer l's fused backward kernel; 0: the layer runs the separate kernels
static int fused_grid(const ava_model* m, int l, int B) {
  static const bool on = [] { const char* e = ava_env("AVA_CONV_FUSED"); return e == nullptr || atoi(e) != 0; }();
  if (!on) return 0;
  const ConvLayer& L = kLayers[l];
  return ava_conv_fused_grid_for(B, m->lay[l].hi, m->lay[l].wi, L.cin, L.cout, L.mode);
}

// convt7's data gradient formed inside convt6's fused backward (the shape that kernel is instantiated for: W = 128 tiles of
// 32 x 4 low-resolution pixels).  Lab build: AVA_DD6_FUSED=0 restores the separate data-gradient launch.
static bool dd6_fused(const ava_model* m) {
  static const bool on = [] { const char* e = ava_env("AVA_DD6_FUSED"); return e == nullptr || atoi(e) != 0; }();
  return on && fused_grid(m, 12, m->lastB > 0 ? m->lastB : 1) > 0 && m->lay[12].wi % 32 == 0 && m->lay[12].hi % 4 == 0;
}

int ava_conv3x3_wgrad_pair(const WgradCall& p, const WgradCall& q, int B, int act_bf16, ava_stream_t s);

// `defer`: a layer without a fused kernel records its weight-gradient call there instead of launching it (the caller
// issues two layers' calls as one pair launch; gin / gin2 must stay valid until then)
static int conv_layer_backward(ava_model* m, int l, const float* x0, const float* gin, const float* gin2,
                               const float* ca, const float* cb, const float* cc, int pro, float* gout, int B,
                               hipStream_t st, WgradCall* defer = nullptr) {
  const ConvLayer& L = kLayers[l];
  const LayerDims& D = m->lay[l];
  const float* X = l == 0 ? x0 : m->X[l];
  // layers with a fused kernel: data gradient, BatchNorm-backward sums and weight/bias partials from one pass
  const int fgrid = fused_grid(m, l, B);
  if (l == 13) {
    // convt7: the training forward has already left this layer's weight-gradient partials and BatchNorm-backward sums behind
    // (FOLD) and convt6's kernel forms its data gradient itself -- nothing to launch.  They are those of loss scale 1; both are
    // linear in the seed, so a backward with another scale (ava_set_backward_scale) has had them multiplied in place together
    // with the seed (backward_part0: ava_scale_backward_roots).  After a forward that did not fold the separate kernel runs.
    if (m->fold13 && dd6_fused(m)) return AVA_OK;
    if (m->fold13) {
      if (hipMemsetAsync(acc_slot(m, 14 + l), 0, (size_t)AVA_ACC_SLOT_LL * sizeof(long long), st) != hipSuccess) return AVA_ELAUNCH;
      m->fold13 = 0;
    }
    m->wg13_rows = fgrid;
  }
  if (fgrid > 0 && (gout != nullptr || l == 0)) {
    FusedArgs a = {};
    a.x = X; a.xa = bn_scale(m, l); a.xb = bn_shift(m, l);
    a.dy = gin; a.dy2 = gin2; a.da = ca; a.db = cb; a.dc = cc;
    a.Gb = m->Gb[l]; a.dx = gout; a.mean = bn_mean(m, l); a.invstd = bn_invstd(m, l);
    a.bn_partials = m->bn_part; a.wg_partials = m->wg_part[l];
    a.B = B; a.Hi = D.hi; a.Wi = D.wi; a.Ho = D.ho; a.Wo = D.wo;
    a.act_bf16 = m->act_bf16;
    // BatchNorm l+1's A, Bc, Cc finalised in this kernel / BatchNorm l's sums accumulated by it (bn_acc.h); the thin
    // kernels of conv1 / convt7 (Cin or Cout = 1) keep arrays and partial rows
    const bool thin = L.cin == 1 || L.cout == 1;
    a.fin = ((!thin || l == 0) && pro == PRO_BWD && acc_pair_bwd(m, l + 1)) ? fin_bwd(m, l + 1, B) : fin_none();   // l = 0: conv1's packed-FMA backward
    a.acc_out = ((!thin || l == 13 || l == 0) && acc_pair_bwd(m, l)) ? acc_slot(m, 14 + l) : nullptr;   // l = 13: convt7's sums kernel; l = 0: conv1's
    a.tiles_y = a.tiles_x = a.ntiles = 0;
    if (dd6_fused(m)) {
      // convt7's data gradient dd6 (8 channels at full resolution: 128 MiB written and read back at batch 256) is never
      // stored: convt7's launch only forms its weight gradient and BatchNorm sums, and convt6's fused backward gathers its
      // dy window from the 1-channel seed in the staging waves (conv_recomp.h)
      if (l == 13) a.skip_dx = 1;
      if (l == 12) { a.dy = m->seed; a.rcd.G1 = m->Gb[13]; }
    }
    // conv1's backward recomputes y1 from the x window it stages instead of reading it (conv_thin_kernels.h: RECY)
    if (l == 0 && pro == PRO_BWD) a.rc = recomp_args(m);
    TRY(ava_conv3x3_bwd_fused_launch(a, L.cin, L.cout, L.mode, pro, st));
    mark(m, CAT_CONV_BWD_DATA, st);
    if (a.acc_out != nullptr) return AVA_OK;                  // finalised by the next backward kernel
    return finalize_bwd(m, l, fgrid, (int64_t)B * D.hi * D.wi, st);
  }
  // Layers without a fused kernel (the four at 16x16).  The data gradient w.r.t. the BatchNorm output (plus the
  // BatchNorm-backward sums against X) goes FIRST: its prologue can finalise BatchNorm l+1's A, Bc, Cc from the
  // accumulated sums (bn_acc.h) and workgroup 0 publishes them, so the weight-gradient kernel behind it reads the arrays.
  const int bmode = L.mode == MODE_S1 ? MODE_S1 : (L.mode == MODE_DOWN ? MODE_UP : MODE_DOWN);
  ConvAcc acc;
  acc.fin = (pro == PRO_BWD && acc_pair_bwd(m, l + 1)) ? fin_bwd(m, l + 1, B) : fin_none();
  acc.acc_out = acc_pair_bwd(m, l) ? acc_slot(m, 14 + l) : nullptr;
  TRY(ava_conv3x3_ex(gin, gin2, ca, cb, cc, m->Gb[l], nullptr, gout, nullptr, X, bn_mean(m, l), bn_invstd(m, l),
                     m->bn_part, B, D.ho, D.wo, L.cout, L.cin, bmode, pro, EPI_BWD, 0, 0.f, m->act_bf16, &acc,
                     reinterpret_cast<ava_stream_t>(st)));
  mark(m, CAT_CONV_BWD_DATA, st);
  if (acc.acc_out == nullptr) TRY(finalize_bwd(m, l, ava_conv_grid(B, D.hi, D.wi, bmode), (int64_t)B * D.hi * D.wi, st));
  // weight + bias gradient (forward gather form), reduced into the reference layout inside the grad arena
  if (defer != nullptr) {
    *defer = WgradCall{X, bn_scale(m, l), bn_shift(m, l), gin, gin2, ca, cb, cc, m->wg_part[l], D.hi, D.wi, L.cin, L.cout, L.mode, pro};
    return AVA_OK;
  }
  TRY(ava_conv3x3_wgrad_ex(X, bn_scale(m, l), bn_shift(m, l), gin, gin2, ca, cb, cc, m->wg_part[l], B, D.hi, D.wi, L.cin,
                           L.cout, L.mode, pro, m->act_bf16, st));
  mark(m, CAT_CONV_WGRAD, st);
  return AVA_OK;
}

// weight/bias gradient reductions of conv layers [l0, l1) in one launch
static int reduce_wgrads(ava_model* m, int l0, int l1, int B, hipStream_t st) {
  WgradReduceTable tab;
  int blocks = 0, n = 0;
  for (int l = l0; l < l1; ++l, ++n) {
    const ConvLayer& L = kLayers[l];
    tab.e[n].partials = m->wg_part[l];
    tab.e[n].dw = GG(m, L.pw);
    tab.e[n].dbias = GG(m, L.pb);
    const int fg = l == 13 && m->wg13_rows > 0 ? m->wg13_rows : fused_grid(m, l, B);
    tab.e[n].nparts = fg > 0 ? fg : ava_conv_wgrad_rows_ex(B, m->lay[l].hi, m->lay[l].wi, L.cin, L.cout, L.mode, l == 13 || l == 6 ? PRO_ID : PRO_BWD, m->act_bf16);
    tab.e[n].cin = L.cin; tab.e[n].cout = L.cout;
    tab.e[n].kind = !L.transposed ? 0 : (L.mode == MODE_S1 ? 1 : 2);
    tab.e[n].block0 = blocks;
    blocks += ceil_div(9 * L.cin * L.cout + L.cout, 32);
  }
  tab.n = n;
  tab.fin0 = (l0 == 0 && acc_pair_bwd(m, 0)) ? fin_bwd(m, 0, B) : fin_none();     // bn1's own gradient: no consumer kernel, finalised here
  TRY(ava_conv_wgrad_reduce_all(tab, blocks, st));
  mark(m, CAT_CONV_WGRAD, st);
  return AVA_OK;
}

// Gradient buckets for the data-parallel all-reduce, in the order backward completes them:
//   bucket 0 = [fc8.weight .. end of arena)   fc8, convt1..7, bn8..14 (33.6 MB)  -- complete after part 0
//   bucket 1 = [fc1.weight .. fc1.bias)       fc1's weight alone      (33.6 MB)  -- complete after part 1
//   bucket 2 = [fc1.bias .. fc8.weight)       fc1.bias, fc2 .. fc7    (2.5 MB)   -- complete after part 2
//   bucket 3 = [0 .. fc1.weight)              conv1..7, bn1..7        (84 KB)    -- complete after part 3
// so the two large all-reduces run under the fully connected and the encoder halves of backward; fc1's weight gradient goes
// out the moment its product has been enqueued (it is the last big product of the fully connected chain), with the rest
// of part 1's old tail (fc1's dX, the eight small weight gradients) still to run beside it.
#define AVA_BACKWARD_PARTS 4
extern "C" int ava_backward_num_parts(void) { return AVA_BACKWARD_PARTS; }
extern "C" int ava_grad_bucket(ava_model* m, int bucket, int64_t* offset, int64_t* count) {
  if (m == nullptr || offset == nullptr || count == nullptr || bucket < 0 || bucket >= AVA_BACKWARD_PARTS) return AVA_EINVAL;
  const int64_t fc1 = m->tab[28].off, fc1b = m->tab[29].off, fc8 = m->tab[50].off;      // fc1.weight, fc1.bias, fc8.weight
  if (bucket == 0) { *offset = fc8; *count = m->arena - fc8; }
  else if (bucket == 1) { *offset = fc1; *count = fc1b - fc1; }
  else if (bucket == 2) { *offset = fc1b; *count = fc8 - fc1b; }
  else { *offset = 0; *count = fc1; }
  return AVA_OK;
}

// `whole`: called from ava_backward (no bucket has to be complete before the end): the decoder's weight-gradient
// reduction is deferred and runs together with the encoder's in ONE launch at the end of part 2
static int backward_part0(ava_model* m, const float* x, int B, hipStream_t st, bool whole);
static int backward_part1(ava_model* m, const float* x, int B, hipStream_t st);
static int backward_part1b(ava_model* m, const float* x, int B, hipStream_t st);
static int backward_part2(ava_model* m, const float* x, int B, hipStream_t st, bool whole);

extern "C" int ava_set_backward_scale(ava_model* m, const float* loss_scale) {
  if (m == nullptr) return AVA_EINVAL;
  m->bwd_scale = loss_scale;
  return AVA_OK;
}

extern "C" int ava_backward(ava_model* m, const float* x, int B, ava_stream_t s) {
  if (m == nullptr || x == nullptr || B != m->lastB || m->G == nullptr) return AVA_EINVAL;
  const ReserveScope rs(m);
  TRY(backward_part0(m, x, B, to_stream(s), true));
  TRY(backward_part1(m, x, B, to_stream(s)));
  TRY(backward_part1b(m, x, B, to_stream(s)));
  return backward_part2(m, x, B, to_stream(s), true);
}
// part 0: decoder convolutions, bn8, fc8's weight gradient;  part 1: the data-gradient chain of the fully connected layers,
// the latent block and fc1's weight gradient;  part 2: fc1's data gradient and the small weight gradients;  part 3: the
// encoder convolutions.  Each completes the gradient bucket of the same number.
extern "C" int ava_backward_part(ava_model* m, const float* x, int B, int part, ava_stream_t s) {
  if (m == nullptr || x == nullptr || B != m->lastB || m->G == nullptr || part < 0 || part >= AVA_BACKWARD_PARTS)
    return AVA_EINVAL;
  const ReserveScope rs(m);
  if (part == 0) return backward_part0(m, x, B, to_stream(s), false);
  if (part == 1) return backward_part1(m, x, B, to_stream(s));
  if (part == 2) return backward_part1b(m, x, B, to_stream(s));
  return backward_part2(m, x, B, to_stream(s), false);
}

// offset (floats) of the upper half of a gradient ping-pong buffer (B * H * W * 8 floats each, model workspace)
static size_t grad_half(const ava_model* m, int B) { return (size_t)B * m->H * m->W * 4; }

static int backward_part0(ava_model* m, const float* x, int B, hipStream_t st, bool whole) {
  // ---- decoder convolutions, last to first ----
  float* gcur = m->gA;
  float* gnext = m->gB;
  mark(m, -1, st);
  if (m->bwd_scale != nullptr) {       // d(result)/d(loss) may differ from 1: scale the roots of the backward (here and latent_bwd)
    const bool folded = m->fold13 && dd6_fused(m);          // ... and what convt7's forward left behind for this backward
    // (a lab build without the fused gather reruns the separate kernel on the scaled seed instead: conv_layer_backward)
    TRY(ava_scale_backward_roots(m->seed, (int64_t)B * m->H * m->W, folded ? m->wg_part[13] : nullptr,
                                 folded ? (int64_t)m->wg13_rows * 73 : 0, folded ? acc_slot(m, 14 + 13) : nullptr, m->bwd_scale, st));
    mark(m, CAT_LAYOUT, st);
  }
  TRY(conv_layer_backward(m, 13, x, m->seed, nullptr, nullptr, nullptr, nullptr, PRO_ID, gcur, B, st));
  const bool pair = fused_grid(m, 8, B) == 0 && fused_grid(m, 7, B) == 0;
  WgradCall wc[2];
  for (int l = 12; l >= 7; --l) {
    // dU_l = (X_{l+1} > 0) ? A*g + Bc*X_{l+1} + Cc : 0 with the coefficients of BatchNorm l+1
    // convt2 + convt1 (16 x 16, no fused kernel): both data gradients first, then the two weight gradients as ONE launch.
    // convt1's data gradient goes to the upper half of the buffer that still holds convt2's dU (both are small)
    float* gout = (pair && l == 7) ? gnext + grad_half(m, B) : gnext;
    TRY(conv_layer_backward(m, l, x, gcur, m->X[l + 1], bn_A(m, l + 1), bn_B(m, l + 1), bn_C(m, l + 1), PRO_BWD, gout,
                            B, st, pair && l <= 8 ? &wc[8 - l] : nullptr));
    gnext = gcur; gcur = gout;
  }
  if (pair) {
    TRY(ava_conv3x3_wgrad_pair(wc[0], wc[1], B, m->act_bf16, reinterpret_cast<ava_stream_t>(st)));
    mark(m, CAT_CONV_WGRAD, st);
  }
  // gcur = dXhat_8 (NHWC [B,256,32]); through bn8 and fc8's ReLU back to NCHW-flatten
  const BnFin fin8 = acc_pair_bwd(m, 7) ? fin_bwd(m, 7, B) : fin_none();      // bn8: finalised inside the layout kernel
  TRY(ava_bn_bwd_apply_to_nchw(gcur, m->f8, bn_A(m, 7), bn_B(m, 7), bn_C(m, 7), m->dF8, B, m->P8, m->act_bf16, &fin8, st));
  mark(m, CAT_LAYOUT, st);
  if (!whole) TRY(reduce_wgrads(m, 7, NCONV, B, st));
  TRY(gemm(m, m->dF8, 0, m->h7, 0, nullptr, GG(m, FC8), 0, nullptr, GG(m, FC8 + 1), m->F, 1024, B, 0, 0, ACT_NONE, st));
  return AVA_OK;
}

static int backward_part1(ava_model* m, const float* x, int B, hipStream_t st) {
  const int z = m->z;
  mark(m, -1, st);
  // ---- fully connected layers.  dX = (dY W) masked by the producer's ReLU runs as a chain; the weight
  // gradients dW = dY^T X (+ db = column sums) only need buffers that stay valid, so the two big ones are
  // issued in place and the eight small ones are collected into ONE grouped launch at the end. ----
  TRY(gemm(m, m->dF8, 0, PP(m, FC8), 0, nullptr, m->dh7, 0, m->h7, nullptr, B, 1024, m->F, 1, 0, ACT_NONE, st));
  TRY(gemm(m, m->dh7, 0, PP(m, FC7), 0, nullptr, m->dh6, 0, m->h6, nullptr, B, 256, 1024, 1, 0, ACT_NONE, st));
  if (fc_mid_on()) {
    // dh6 -> dh5 -> dz -> latent backward -> dh3 in one launch (fc_mid.hip); fc31|32|33's data gradient stays a launch
    FcMidBwdArgs ba;
    ba.dh6 = m->dh6; ba.W6 = PP(m, FC6); ba.W5 = PP(m, FC5); ba.W41 = PP(m, FC41); ba.W42 = PP(m, FC42); ba.W43 = PP(m, FC43);
    ba.W3 = PP(m, FC31); ba.h5 = m->h5; ba.h3 = m->h3; ba.h2 = m->h2;
    ba.z = m->zs; ba.u = m->u; ba.d = m->d; ba.eps_w = m->eps_w_last; ba.eps_d = m->eps_d_last; ba.scale = m->bwd_scale;
    ba.dh5 = m->dh5; ba.dz = m->dz; ba.dmu = m->dmu; ba.du = m->du; ba.dlogd = m->dlogd; ba.dh3 = m->dh3; ba.dh2 = m->dh2;
    ba.B = B; ba.zdim = z;
    TRY(ava_fc_mid_bwd(ba, st));
    mark(m, CAT_LATENT_LOSS, st);
    TRY(gemm(m, m->dh3, 0, PP(m, FC31), 0, nullptr, m->dh2, 0, m->h2, nullptr, B, 256, 192, 1, 0, ACT_NONE, st));
  } else {
  TRY(gemm(m, m->dh6, 0, PP(m, FC6), 0, nullptr, m->dh5, 0, m->h5, nullptr, B, 64, 256, 1, 0, ACT_NONE, st));
  TRY(gemm(m, m->dh5, 0, PP(m, FC5), 0, nullptr, m->dz, 0, nullptr, nullptr, B, z, 64, 1, 0, ACT_NONE, st));
  // ---- latent block ----
  TRY(ava_latent_bwd_scaled(m->zs, m->dz, m->u, m->d, m->eps_w_last, m->eps_d_last, m->dmu, m->du, m->dlogd, B, z,
                            m->bwd_scale, st));
  mark(m, CAT_LATENT_LOSS, st);
  // ---- heads: dX of fc41/42/43 into the three 64-wide slices of dh3 (masked by h3's ReLU), one launch ----
  const AvaGemmProblem hdx[3] = {
      {m->dmu, 0, PP(m, FC41), 0, nullptr, m->dh3 + 0, 192, m->h3 + 0, nullptr, B, 64, z, ACT_NONE},
      {m->du, 0, PP(m, FC42), 0, nullptr, m->dh3 + 64, 192, m->h3 + 64, nullptr, B, 64, z, ACT_NONE},
      {m->dlogd, 0, PP(m, FC43), 0, nullptr, m->dh3 + 128, 192, m->h3 + 128, nullptr, B, 64, z, ACT_NONE}};
  TRY(gemm_group(m, hdx, 3, 1, 0, st));
  TRY(gemm(m, m->dh3, 0, PP(m, FC31), 0, nullptr, m->dh2, 0, m->h2, nullptr, B, 256, 192, 1, 0, ACT_NONE, st));
  }
  TRY(gemm(m, m->dh2, 0, PP(m, FC2), 0, nullptr, m->dh1, 0, m->h1, nullptr, B, 1024, 256, 1, 0, ACT_NONE, st));
  // fc1's weight gradient (its bias gradient -- the column sums -- is written by the same launch but belongs to bucket 2:
  // nothing reads it before part 2 is complete)
  return gemm(m, m->dh1, 0, m->y7t, 0, nullptr, GG(m, FC1), 0, nullptr, GG(m, FC1 + 1), 1024, m->F, B, 0, 0, ACT_NONE, st);
}

static int backward_part1b(ava_model* m, const float* x, int B, hipStream_t st) {
  const int z = m->z;
  (void)x;
  mark(m, -1, st);
  // fc1's dX: left as two slabs where it runs so (the last part's ReLU-mask / layout kernel sums them).  The product runs in
  // a workspace region of its own (dy7_ws), so no launch between the parts -- whatever split-K product is added there
  // later -- can overwrite the slabs
  m->dy7_slabs = 1;
  {
    struct Mk { ava_model* m; hipStream_t st; ~Mk() { mark(m, CAT_GEMM, st); } } _mk{m, st};
    TRY(ava_gemm_defer2(m->dh1, 0, PP(m, FC1), 0, nullptr, m->dy7, 0, nullptr, nullptr, B, m->F, 1024, 1, 0, ACT_NONE,
                        m->dy7_ws, m->dy7_ws_bytes, st, &m->dy7_slabs));
  }
  // ---- the eight small weight gradients (K = batch): fc7, fc6, fc5, fc41/42/43, fc31|32|33, fc2 ----
  const AvaGemmProblem dws[8] = {
      {m->dh7, 0, m->h6, 0, nullptr, GG(m, FC7), 0, nullptr, GG(m, FC7 + 1), 1024, 256, B, ACT_NONE},
      {m->dh6, 0, m->h5, 0, nullptr, GG(m, FC6), 0, nullptr, GG(m, FC6 + 1), 256, 64, B, ACT_NONE},
      {m->dh5, 0, m->zs, 0, nullptr, GG(m, FC5), 0, nullptr, GG(m, FC5 + 1), 64, z, B, ACT_NONE},
      {m->dmu, 0, m->h3 + 0, 192, nullptr, GG(m, FC41), 0, nullptr, GG(m, FC41 + 1), z, 64, B, ACT_NONE},
      {m->du, 0, m->h3 + 64, 192, nullptr, GG(m, FC42), 0, nullptr, GG(m, FC42 + 1), z, 64, B, ACT_NONE},
      {m->dlogd, 0, m->h3 + 128, 192, nullptr, GG(m, FC43), 0, nullptr, GG(m, FC43 + 1), z, 64, B, ACT_NONE},
      {m->dh3, 0, m->h2, 0, nullptr, GG(m, FC31), 0, nullptr, GG(m, FC31 + 1), 192, 256, B, ACT_NONE},
      {m->dh2, 0, m->h1, 0, nullptr, GG(m, FC2), 0, nullptr, GG(m, FC2 + 1), 256, 1024, B, ACT_NONE}};
  return gemm_group(m, dws, 8, 0, 0, st);
}

static int backward_part2(ava_model* m, const float* x, int B, hipStream_t st, bool whole) {
  // ---- encoder convolutions ----
  float* gcur = m->gA;
  float* gnext = m->gB;
  mark(m, -1, st);
  if (recomp_y1()) TRY(materialize_y1(m, x, B, st));       // TEMPORARY: until conv2's / conv1's backward recompute y1 themselves
  if (m->dy7_slabs == 2) {
    const float* s0 = reinterpret_cast<const float*>(m->dy7_ws);
    TRY(ava_relu_mask_to_nhwc(s0, s0 + (size_t)B * m->F, m->y7, gcur, B, m->P8, st));
  } else {
    TRY(ava_relu_mask_to_nhwc(m->dy7, nullptr, m->y7, gcur, B, m->P8, st));  // dU_7 (ReLU of conv7)
  }
  mark(m, CAT_LAYOUT, st);
  // conv7 + conv6 (16 x 16, no fused kernel): both data gradients first, then the two weight gradients as ONE launch
  const bool pair = fused_grid(m, 6, B) == 0 && fused_grid(m, 5, B) == 0;
  WgradCall wc[2];
  TRY(conv_layer_backward(m, 6, x, gcur, nullptr, nullptr, nullptr, nullptr, PRO_ID, gnext, B, st, pair ? &wc[0] : nullptr));
  { float* t = gcur; gcur = gnext; gnext = t; }
  bool upper = false;
  for (int l = 5; l >= 0; --l) {
    // conv6's data gradient goes to the upper half of the buffer that still holds conv7's dU (both are small)
    float* gout = l == 0 ? nullptr : ((pair && l == 5) ? gnext + grad_half(m, B) : gnext);
    TRY(conv_layer_backward(m, l, x, gcur, m->X[l + 1], bn_A(m, l + 1), bn_B(m, l + 1), bn_C(m, l + 1), PRO_BWD, gout, B, st,
                            pair && l == 5 ? &wc[1] : nullptr));
    if (pair && l == 5) {
      TRY(ava_conv3x3_wgrad_pair(wc[0], wc[1], B, m->act_bf16, reinterpret_cast<ava_stream_t>(st)));
      mark(m, CAT_CONV_WGRAD, st);
    }
    float* freed = upper ? gcur - grad_half(m, B) : gcur;      // an upper half read by this layer: the whole buffer is free again
    gcur = gout; gnext = freed;
    upper = pair && l == 5;
  }
  m->bwd_scale = nullptr;              // consumed
  return reduce_wgrads(m, whole ? 0 : 0, whole ? NCONV : 7, B, st);      // encoder (whole: all 14) weight/bias gradients
}

// the same update on the slice [offset, offset + count) of the arenas (data parallelism with a sharded optimizer:
// every rank updates 1/N of each gradient bucket and the parameters are all-gathered; dist.py)
extern "C" int ava_adam_step_range(ava_model* m, int64_t offset, int64_t count, double lr, double beta1, double beta2,
                                   double eps, int step, ava_stream_t s) {
  if (m == nullptr || m->G == nullptr || m->M == nullptr || m->V == nullptr || offset < 0 || count <= 0 ||
      offset + count > m->arena || offset % 4 != 0 || count % 4 != 0)
    return AVA_EINVAL;
  const ReserveScope rs(m);
  mark(m, -1, to_stream(s));
  const int rc = ava_adam_flat_guarded(m->P + offset, m->G + offset, m->M + offset, m->V + offset, count, lr, beta1, beta2,
                                       eps, step, m->status_last, to_stream(s));
  mark(m, CAT_ADAM, to_stream(s));
  return rc;
}

extern "C" int ava_adam_step(ava_model* m, double lr, double beta1, double beta2, double eps, int step,
                             ava_stream_t s) {
  if (m == nullptr || m->G == nullptr || m->M == nullptr || m->V == nullptr) return AVA_EINVAL;
  const ReserveScope rs(m);
  mark(m, -1, to_stream(s));
  // the reference raises inside forward() when d is not positive (vae.py:312) and never reaches optimizer.step():
  // the update is skipped on the device when the last forward set its status word
  const int rc = ava_adam_flat_guarded(m->P, m->G, m->M, m->V, m->arena, lr, beta1, beta2, eps, step, m->status_last,
                                       to_stream(s));
  mark(m, CAT_ADAM, to_stream(s));
  return rc;
}
