/*
 * ava_hip.h -- C ABI of libava_hip.so: the MI355X (gfx950) implementation of the VAE
 * training hot path of pearsonlab/autoencoded-vocal-analysis (reference file
 * ava/models/vae.py).
 *
 * The reference has no FFI of its own (it is 100 % Python on top of torch); the entry
 * points below are what a binding for this path replaces, one per implicit library op
 * the reference dispatches, plus a fused whole-step driver.
 * Every function takes raw device pointers, sizes and a hipStream_t (passed as void*),
 * returns 0 on success or a negative AVA_E* code, allocates nothing and never
 * synchronises: all scratch memory comes from the caller-provided workspace.
 *
 * Layouts: activations are NHWC fp32 (`[B,H,W,C]`) between the convolutions, row-major
 * `[B,features]` on the fully connected side with the reference's NCHW flatten order
 * (c*256 + h*16 + w, vae.py:224,262) at the two boundaries.  Parameters live in ONE
 * flat fp32 arena in named_parameters() order (offsets: ava_param_offset), shadowed by
 * three arenas of the same shape for the gradient, Adam exp_avg and exp_avg_sq.
 */
#ifndef AVA_HIP_H
#define AVA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVA_OK 0
#define AVA_EINVAL (-1)   /* bad argument (shape, null pointer, unsupported channel count) */
#define AVA_ELAUNCH (-2)  /* hipGetLastError() reported a launch failure                    */
#define AVA_EWORKSPACE (-3) /* workspace too small                                          */

typedef void* ava_stream_t;            /* hipStream_t */
typedef struct ava_model ava_model;    /* opaque: pointer table + workspace carving, host memory only */

/* ---- library / layout queries ------------------------------------------------------------- */
int ava_version(void);
/* number of floats of the flat parameter arena for this z_dim (tensors padded to 64 floats) */
int64_t ava_arena_floats(int z_dim);
/* offset (in floats) of parameter `index` (0..79, reference named_parameters() order,
 * vae.py:125-168) inside the arena; numel returned through *numel (may be NULL) */
int64_t ava_param_offset(int z_dim, int index, int64_t* numel);
/* bytes of scratch the model needs for batches up to max_batch */
size_t ava_workspace_bytes(int z_dim, int max_batch);

/* ---- model object: replaces VAE.__init__/_build_network bookkeeping (vae.py:80-168) -------- */
/* bn_running: [2][14][32] floats = running_mean then running_var of bn1..bn14, 32 slots per layer
 * (channel counts 1,8,8,16,16,24,24,32,24,24,16,16,8,8); bn_batches: 14 int64 counters. */
int ava_model_create(ava_model** out, int z_dim, int max_batch, float model_precision,
                     float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                     float* bn_running, int64_t* bn_batches, void* workspace, size_t workspace_bytes);
void ava_model_destroy(ava_model* m);
/* The same for spectrograms of H x W instead of the reference's module constant X_SHAPE = (128, 128) (vae.py:33-36;
 * BASELINE config 5 is 256 x 256): every layer keeps its channels, strides and 3x3 kernels, the spatial sizes scale, and
 * fc1.in = fc8.out = 32 * (H/8) * (W/8) replaces the literal 8192 of vae.py:142,153,224,262.  W must be 128 or 256 and H
 * a multiple of 128 (128..1024); other sizes return -1 / 0 / AVA_EINVAL.  The functions without the _hw suffix are
 * these with H = W = 128. */
int64_t ava_arena_floats_hw(int z_dim, int H, int W);
int64_t ava_param_offset_hw(int z_dim, int H, int W, int index, int64_t* numel);
size_t ava_workspace_bytes_hw(int z_dim, int H, int W, int max_batch);
int ava_model_create_hw(ava_model** out, int z_dim, int H, int W, int max_batch, float model_precision,
                        float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                        float* bn_running, int64_t* bn_batches, void* workspace, size_t workspace_bytes);
/* ... and with the storage type of the activations between the convolutions (BASELINE configs[4]: "bf16 conv + fp32
 * ELBO"): act_dtype 0 = float32 (the reference), 1 = bfloat16 -- the thirteen tensors that connect the 14 conv layers
 * (and are kept for the backward pass) are stored as bf16, rounded to nearest even by the producing kernel; every
 * product still accumulates in fp32 on the matrix cores / packed FMAs, and BatchNorm statistics (taken of the
 * rounded values), gradients, the fully connected layers, the ELBO and Adam (fp32 master weights) stay fp32.  Same
 * workspace size as ava_workspace_bytes_hw. */
int ava_model_create_ex(ava_model** out, int z_dim, int H, int W, int act_dtype, int max_batch, float model_precision,
                        float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                        float* bn_running, int64_t* bn_batches, void* workspace, size_t workspace_bytes);

/* ---- whole-path entry points ------------------------------------------------------------------ */
/* VAE.forward (vae.py:273-327): encode -> rsample -> decode -> -ELBO.
 *   x [B,128,128]; eps_w [B], eps_d [B,z]: the two normal draws of rsample in reference order.
 *   bn_train: 1 = batch statistics + running-stat update (module.train()), 0 = running stats.
 *   loss_out (device, 4 floats): {-ELBO, sum z^2, SSE, sum entropy}.
 *   loss_accum (device double, may be NULL): -ELBO is also added to it (the epoch loops' running sum,
 *     vae.py:351,382, kept on the device so that no step needs a host sync).
 *   status_out (device int[2], may be NULL): word 0 is OR-ed with 1 when some d is not > 0 (reference raises
 *     ValueError); sticky -- the caller clears it after reading.  Word 1 is incremented by every ava_adam_step that
 *     skipped its update because word 0 was set (so the caller can take those steps back out of its step count).
 * x, eps_w, eps_d must stay valid until ava_backward has run.
 * Leaves every intermediate needed by ava_backward in the workspace, together with the BatchNorm mode: the backward of
 * a bn_train = 0 forward differentiates the running-statistics form (dx = gamma*invstd*g), as autograd does for
 * model.eval(); loss = model(x); loss.backward().  ava_encode / ava_decode overwrite those intermediates: an
 * ava_backward after them returns AVA_EINVAL until the next ava_forward. */
int ava_forward(ava_model* m, const float* x, int B, const float* eps_w, const float* eps_d,
                int bn_train, float* loss_out, double* loss_accum, int* status_out, ava_stream_t s);
/* The same with the noise of rsample() (vae.py:313) drawn by the device inside the forward's first launch instead of
 * being passed in: eps (device, B*(z_dim+1) floats) receives eps_W [B] then eps_D [B,z_dim], elements offset ..
 * offset + B*(z_dim+1) - 1 of the counter stream of ava_fill_normal(seed) -- bit-identical to ava_fill_normal(eps, ...)
 * followed by ava_forward(m, x, B, eps, eps + B, ...), one launch less.  eps must stay valid until ava_backward. */
int ava_forward_noise(ava_model* m, const float* x, int B, float* eps, uint64_t seed, uint64_t offset,
                      int bn_train, float* loss_out, double* loss_accum, int* status_out, ava_stream_t s);
/* loss.backward() (vae.py:352) for the forward that just ran: fills the gradient arena
 * (overwrites: the reference zero_grad()s before every step, vae.py:348). */
int ava_backward(ava_model* m, const float* x, int B, ava_stream_t s);
/* autograd hands loss.backward() a grad_output; when the caller backpropagates c*loss (or a sum the loss is part of)
 * it is not 1.  loss_scale: device pointer to that scalar, read by the NEXT ava_backward / ava_backward_part sequence
 * (the seed gradient and the prior/entropy terms are multiplied by it: the gradient is linear in it), then forgotten.
 * NULL (the default after every ava_forward) means 1.  Replaces a 70 MB pass over the gradient arena. */
int ava_set_backward_scale(ava_model* m, const float* loss_scale);
/* The same backward in ava_backward_num_parts() (= 4) consecutive parts, so that a data-parallel caller can
 * all-reduce each part's gradients while the next part runs.  Part p completes gradient bucket p, a contiguous
 * range of the arena returned by ava_grad_bucket (floats): bucket 0 = fc8, convt1..7, bn8..14 (tail of the arena),
 * bucket 1 = fc1.weight alone (the largest tensor: it goes out as soon as its product is enqueued), bucket 2 = fc1.bias,
 * fc2..fc7, bucket 3 = conv1..7, bn1..7 (head).  The buckets tile the arena. */
int ava_backward_num_parts(void);
int ava_backward_part(ava_model* m, const float* x, int B, int part, ava_stream_t s);
int ava_grad_bucket(ava_model* m, int bucket, int64_t* offset, int64_t* count);
/* Data parallelism (the reference is single-device: vae.py:112-115).  Every convolution / large-GEMM launch is ONE
 * resident wave of persistent workgroups over a static tile partition; a collective's own persistent workgroups (RCCL)
 * need wave slots beside them.  ava_set_cu_reserve(r) sizes all those grids for (256 - r) CUs from now on (process-wide,
 * 0 <= r <= 128; default 0).  Results are bit-reproducible for a given r.  ava_occupy_cus(w, lds, usec, stream) launches w
 * 256-thread workgroups with `lds` bytes of LDS each (which bounds how many share a CU) that only hold their slots for
 * usec microseconds (<= 20 ms): a stand-in for such a collective in tests. */
int ava_set_cu_reserve(int cus);
int ava_get_cu_reserve(void);
/* The same per model: the persistent grids of THIS model's entry points (forward, backward parts, encode, decode) are sized
 * for (256 - cus) CUs from the next call on; -1 (the default) follows the process-wide setting.  Adam is not affected: its
 * kernel is a grid-stride launch of small blocks with no static tile partition, which fills whatever wave slots a collective
 * leaves free and needs no co-residency.  Each entry point reads the
 * value once, at its start, so launches that hand partial rows to each other inside one call always agree on the grid. */
int ava_model_set_cu_reserve(ava_model* m, int cus);
int ava_model_get_cu_reserve(const ava_model* m);
int ava_occupy_cus(int workgroups, int lds_bytes, float usec, ava_stream_t s);
/* torch.optim.Adam.step (torch/optim/adam.py:414-547), one fused pass over the four arenas.
 * `step` is the 1-based count after increment.  When the last ava_forward raised its status word (some d not > 0: the
 * reference raises ValueError inside forward and never reaches optimizer.step(), vae.py:312,353) the kernel leaves
 * parameters and moments untouched. */
int ava_adam_step(ava_model* m, double lr, double beta1, double beta2, double eps, int step, ava_stream_t s);
/* The same update restricted to the floats [offset, offset + count) of the four arenas (offset, count multiples of 4):
 * a data-parallel caller that reduce-scatters the gradient buckets lets every rank update its 1/N slice of each bucket
 * and all-gathers the parameters (same bytes on the wire as an all-reduce, Adam's HBM traffic divided by N). */
int ava_adam_step_range(ava_model* m, int64_t offset, int64_t count, double lr, double beta1, double beta2, double eps,
                        int step, ava_stream_t s);
/* VAE.encode (vae.py:216-233): mu,u,d [B,z] (d = exp(.)); bn_train as above. */
int ava_encode(ava_model* m, const float* x, int B, int bn_train, float* mu, float* u, float* d, ava_stream_t s);
/* VAE.decode (vae.py:258-270): z [B,z] -> x_rec [B,16384]. */
int ava_decode(ava_model* m, const float* z, int B, int bn_train, float* x_rec, ava_stream_t s);
/* pointers into the workspace after ava_forward: z [B,z] and x_rec [B,16384] */
const float* ava_last_z(ava_model* m);
const float* ava_last_xrec(ava_model* m);
/* name -> workspace buffer of an intermediate (tests): "y1".."y7","d1".."d6","f8","mu","u","logd",... */
const float* ava_debug_buffer(ava_model* m, const char* name, int64_t* floats);
/* Intermediates the step does not keep in memory because their consumers recompute them (y1 = relu(conv1(bn1 x)),
 * vae.py:217) are written into their ava_debug_buffer slots ("y1") by the kernel whose store-free form took their
 * statistics: same arithmetic, bit for bit.  Call after the ava_forward whose intermediates are wanted, with its x. */
int ava_debug_materialize(ava_model* m, const float* x, int B, ava_stream_t s);

/* Optional timing of the driver's launches with HIP events recorded on the launch stream
 * (bench.py's roofline leg).  Categories, in order: conv forward, conv backward-data, conv weight-grad
 * (+ its reduction), BatchNorm statistics/finalise, GEMM, layout hand-offs, latent+ELBO, Adam, weight pack.
 * ava_profile_read adds elapsed milliseconds (and launch-group counts) per category and clears the list.
 * on = 1: an event after every launch group (~100 per step; the records stretch the step by 10-15 %).
 * on = 2: coarse pass, after at least one step was read in mode 1: events only where the kernel FAMILY changes
 * (conv + BatchNorm + pack vs everything else, ~20 per step); a run's time is added to the category of its last
 * launch group, so only the family sums are meaningful -- they are the un-stretched figures the roofline uses. */
#define AVA_PROFILE_CATEGORIES 9
int ava_profile_enable(ava_model* m, int on);
int ava_profile_read(ava_model* m, float* ms, int* launches);

/* ---- per-op entry points (what the reference reaches through ATen) ---------------------------- */
/* counter-based standard normals (replaces torch's normal_() in rsample when no noise is injected) */
int ava_fill_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, ava_stream_t s);

/* pack a Conv2d weight [Cout,Cin,3,3] / ConvTranspose2d weight [Cin,Cout,3,3] into the
 * gather form G[9][Cin_g][Cout_g] used by the kernels.
 *   kind: 0 conv fwd, 1 convT stride-1 fwd, 2 convT stride-2 fwd,
 *         3 conv stride-1 bwd-data, 4 conv stride-2 bwd-data, 5 convT stride-1 bwd-data, 6 convT stride-2 bwd-data */
int ava_pack_conv_weight(const float* w, float* g, int c_first, int c_second, int kind, ava_stream_t s);

/* 3x3 gather convolution, NHWC.  mode: 0 same-resolution (conv s1 / convT s1 / their bwd-data),
 * 1 down x2 (conv s2 fwd, convT s2 bwd-data), 2 up x2 (convT s2 fwd, conv s2 bwd-data).
 * Prologue applied to the input while it is staged into LDS (zero padding is applied AFTER it):
 *   pro 0: v*pa[c] + pb[c]                      (BatchNorm apply, vae.py:217-223)
 *   pro 1: (in2 > 0) ? pa[c]*v + pb[c]*in2 + pc[c] : 0   (ReLU mask + BatchNorm backward, in2 = saved activation)
 *   pro 2: v
 * Epilogue:
 *   epi 0: + bias, optional ReLU, store, per-channel {sum, sum^2} partials (next BatchNorm's statistics);
 *          out2 (optional, stride-1 layers with >= 8 channels on both sides): a second copy in NCHW order
 *          [B][Cout][Ho][Wo], the flatten order nn.Linear reads after conv7 (vae.py:224)
 *   epi 1: store, per-channel {sum g, sum g*xhat} partials with xhat = (epi_x - mean)*invstd (BatchNorm backward sums)
 *   epi 2: + bias, store x_rec, r = x_rec - epi_x, store prec*r to out2, partial {sum r^2}  (vae.py:319-320)
 * partials: [ava_conv_grid(...)][2*Cout] floats. */
int ava_conv_grid(int B, int Ho, int Wo, int mode);
int ava_conv3x3(const float* in, const float* in2, const float* pa, const float* pb, const float* pc,
                const float* G, const float* bias, float* out, float* out2,
                const float* epi_x, const float* epi_mean, const float* epi_invstd, float* partials,
                int B, int Hi, int Wi, int Cin, int Cout, int mode, int pro, int epi, int relu, float prec,
                ava_stream_t s);
/* weight/bias gradient of the same gather convolution: dG[9][Cin][Cout] and db[Cout] partials per
 * workgroup ([grid][9*Cin*Cout + Cout]); x side uses prologue 0 (BatchNorm apply), dy side pro 1 or 2. */
int ava_conv3x3_wgrad(const float* x, const float* xa, const float* xb,
                      const float* dy, const float* dy2, const float* da, const float* db_, const float* dc,
                      float* partials, int B, int Hi, int Wi, int Cin, int Cout, int mode, int dy_pro,
                      ava_stream_t s);
int ava_conv_wgrad_grid(int B, int Ho, int Wo, int mode);
/* rows of `partials` that ava_conv3x3_wgrad actually writes for this shape (<= ava_conv_wgrad_grid, which sizes the
 * buffer): the matrix-core kernels launch one resident wave of workgroups.  Reduce exactly this many rows, or zero
 * the buffer first. */
int ava_conv_wgrad_rows(int B, int Hi, int Wi, int Cin, int Cout, int mode, int dy_pro);
/* Fused backward of one layer (autograd's conv backward behind loss.backward(), ava/models/vae.py:349): ONE pass over
 * x, dy (and dy2) produces what ava_conv3x3(pro 1|2, epi 1) in the backward-data pattern and ava_conv3x3_wgrad produce
 * separately -- dx = gradient w.r.t. the BatchNorm output [B,Hi,Wi,Cin], bn_partials [grid][2*Cin] = {sum dx,
 * sum dx*xhat} with xhat = (x - mean)*invstd, wg_partials [grid][9*Cin*Cout + Cout].  Gb = backward-data weights
 * (pack kinds 3..6).  grid = ava_conv_fused_grid(...); 0 means the shape has no fused instantiation (the layers
 * with 8 or 16 channels on both sides and the 1 -> 8 layer have one) and ava_conv3x3_bwd_fused returns AVA_EINVAL
 * for it.  The 1 -> 8 layer (first layer: nothing upstream) never forms dx: pass dx = NULL, the sums are exact. */
int ava_conv_fused_grid(int B, int Hi, int Wi, int Cin, int Cout, int mode);
int ava_conv3x3_bwd_fused(const float* x, const float* xa, const float* xb,
                          const float* dy, const float* dy2, const float* da, const float* db_, const float* dc,
                          const float* Gb, float* dx, const float* mean, const float* invstd,
                          float* bn_partials, float* wg_partials,
                          int B, int Hi, int Wi, int Cin, int Cout, int mode, int dy_pro, ava_stream_t s);
/* reduce the per-workgroup partials and scatter into the reference weight layout
 * (kind as in ava_pack_conv_weight, 0..2 only) */
int ava_conv_wgrad_reduce(const float* partials, int nparts, float* dw, float* dbias,
                          int Cin, int Cout, int kind, ava_stream_t s);

/* BatchNorm2d statistics of a raw tensor [n, C] (channel innermost) -> partials [grid][2C] */
int ava_bn_stats(const float* x, int64_t n, int C, float* partials, int* nparts, ava_stream_t s);
/* partials -> mean, invstd, scale = gamma*invstd, shift = beta - mean*scale; train: running-stat update
 * (momentum 0.1, unbiased variance, eps 1e-5); eval: uses running stats. */
int ava_bn_finalize(const float* partials, int nparts, int64_t n, int C, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int64_t* num_batches, int train,
                    float* mean, float* invstd, float* scale, float* shift, ava_stream_t s);
/* backward: partials {sum g, sum g*xhat} -> dgamma, dbeta and the per-channel coefficients
 * dx = A*g + Bc*x + Cc */
int ava_bn_finalize_bwd(const float* partials, int nparts, int64_t n, int C, const float* gamma,
                        const float* mean, const float* invstd, float* dgamma, float* dbeta,
                        float* A, float* Bc, float* Cc, ava_stream_t s);

/* C[M,N] = relu_mask(act(A*B + bias)) on the fp32 matrix cores (nn.Linear and its two backward products).
 *   a_kmajor: 1 = A stored [M,K] (K contiguous), 0 = A stored [K,M] ; lda = stored leading dimension (0: dense)
 *   b_kmajor: 1 = B stored [N,K] (K contiguous), 0 = B stored [K,N] ; ldb likewise ; ldc: row stride of C (0: N)
 *   act: 0 none, 1 ReLU, 2 exp ; bias may be NULL ; split-K partial slabs go to `ws`
 *   mask (may be NULL, row stride ldc): C = mask > 0 ? value : 0  (ReLU backward of the layer that produced mask)
 *   colsum (may be NULL): receives sum over K of A (bias gradient when A = dY^T) */
size_t ava_gemm_workspace_bytes(int M, int N, int K);
int ava_gemm(const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc,
             const float* mask, float* colsum, int M, int N, int K, int a_kmajor, int b_kmajor, int act,
             void* ws, size_t ws_bytes, ava_stream_t s);

/* latent block: d = exp(a), z = mu + u*eps_w + sqrt(d)*eps_d, per-sample sum z^2 and entropy
 * (torch/distributions/lowrank_multivariate_normal.py:17-38,214-252).  sums: [B][2]. */
int ava_latent_fwd(const float* mu, const float* u, const float* logd, const float* eps_w, const float* eps_d,
                   float* d, float* z, float* sums, int* status, int B, int zdim, ava_stream_t s);
/* closed-form backward: g = z + dz_dec */
int ava_latent_bwd(const float* z, const float* dz_dec, const float* u, const float* d, const float* eps_w,
                   const float* eps_d, float* dmu, float* du, float* dlogd, int B, int zdim, ava_stream_t s);
/* assemble -ELBO from the partial sums (vae.py:316-323); loss_out = {loss, sum z^2, SSE, sum H} */
int ava_elbo_finalize(const float* latent_sums, int B, const float* sse_partials, int nparts,
                      int zdim, float prec, float* loss_out, ava_stream_t s);
/* Adam over flat arenas */
/* hyper-parameters are doubles like the Python floats torch receives; each is rounded to fp32 exactly where
 * torch's kernels round it (1-beta1, beta2, 1-beta2, lr/bias_correction1, sqrt(bias_correction2), eps) */
int ava_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                  double eps, int step, ava_stream_t s);

/* ---- input feeding ------------------------------------------------------------------- */
/* Loader bytes -> fp32 spectrograms on the device: replaces the per-item CPU conversion numpy_to_tensor
 * (ava/models/utils.py:444-446, applied by SyllableDataset.__getitem__, ava/models/vae_dataset.py:125-145) with a
 * device-side cast of the raw batch, same rounding as torch's .type(torch.FloatTensor).
 * src_dtype: 0 float32 (copy), 1 float64, 2 uint8, 3 float16, 4 bfloat16.  src and dst 16-byte aligned device
 * pointers, n elements. */
int ava_cast_to_f32(const void* src, int src_dtype, int64_t n, float* dst, ava_stream_t s);
/* HOST function (no GPU involved): collate a batch into a page-locked ring slot -- dst[i] = src[idx[i]] for n rows of
 * row_bytes each on `threads` host threads (idx == NULL: the contiguous rows first .. first+n-1).  Replaces the
 * per-item collation of the reference's DataLoader (ava/models/vae_dataset.py:89-96) for array-backed datasets. */
int ava_host_gather_rows(void* dst, const void* src, const int64_t* idx, int64_t first, int64_t n, size_t row_bytes,
                         int threads);

/* ---- MMD^2 between sets of latent means (downstream consumer of get_latent) ------------ */
/* _estimate_mmd2 (ava/plotting/mmd_plots.py:255-296, Gretton et al. 2012, unbiased quadratic-time estimator with a
 * Gaussian kernel of bandwidth sigma).  latent: [N][z] float64 row-major on the device (what VAE.get_latent returns,
 * vae.py:538); i1 / i2: device int64 index lists of the two conditions (already subsampled if the caller wants max_n).
 * out4 (device, 4 doubles) = {term_1, term_2, term_3, term_1 + term_2 - term_3}.  n1, n2 >= 2 (the reference divides
 * by n*(n-1)); z <= 128.  ws: ava_mmd2_workspace_bytes(n1, n2) bytes of device scratch. */
size_t ava_mmd2_workspace_bytes(int n1, int n2);
int ava_mmd2(const double* latent, int z, const int64_t* i1, int n1, const int64_t* i2, int n2, double sigma,
             double* out4, void* ws, size_t ws_bytes, ava_stream_t s);
/* _estimate_mmd2_linear_time (mmd_plots.py:299-312): m = min(len(i1), len(i2)) / 2 quadruples (i1[2i], i2[2i],
 * i1[2i+1], i2[2i+1]); out (device double) = sum h / m.  ws: (min(ceil(m/256), 1024) + 8) doubles. */
int ava_mmd2_linear(const double* latent, int z, const int64_t* i1, const int64_t* i2, int m, double sigma,
                    double* out, void* ws, size_t ws_bytes, ava_stream_t s);
/* squared distances of n index pairs, out[p] = |latent[a[p]] - latent[b[p]]|^2: the sampled pairs of
 * estimate_median_sigma (mmd_plots.py:450-474; the median itself is taken by the caller). */
int ava_pair_sqdist(const double* latent, int z, const int64_t* a, const int64_t* b, int n, double* out,
                    ava_stream_t s);

/* ---- shotgun spectrograms on the device -------------------------------------------
 * get_spec (ava/preprocessing/utils.py:18-110) for a batch of n windows as FixedWindowDataset.__getitem__ issues it
 * (ava/models/window_vae_dataset.py:213-224: get_spec(max(0, onset - shoulder), offset + shoulder, audio[file], p,
 * fs=fs, target_times=linspace(onset, offset, T))), with the audio of all files resident in ONE device buffer.
 *   audio / audio_dtype   concatenated samples of all files; 0 = int16, 1 = int32, 2 = float32, 3 = float64
 *                         (what scipy.io.wavfile.read returns, window_vae_dataset.py:167)
 *   file_off, file_len    [files] first sample / number of samples of each file in `audio` (device)
 *   file_idx, t1, t2      [n] file of each window, get_spec's t1 / t2 in seconds (device)
 *   target_times          [n][T] get_spec's target_times (device); target_freqs [F] (utils.py:80-88, device)
 *   max_samples           >= max over windows of round(t2 fs) - round(t1 fs): sizes the STFT scratch
 *   window, scale         scipy.signal.get_window('hann', nperseg) (device) and sqrt(1 / sum(window)^2): what
 *                         scipy.signal.stft(..., scaling='spectrum') applies (utils.py:74)
 *   spec_min, spec_max    p['spec_min_val'], p['spec_max_val'] (utils.py:100-103); fill_value utils.py:19
 *   out [n][F][T] fp32    the clipped spectrograms; out_max [n] or NULL: their maxima (min_spec_val test,
 *                         window_vae_dataset.py:229-231)
 *   normalize, q_lo,      p['within_syll_normalize'] (utils.py:104-108): subtract np.quantile(spec, p['normalize_quantile']),
 *   q_gamma               floor at 0, divide by max + 1e-12; the quantile is a[q_lo] + (a[q_lo+1] - a[q_lo]) * q_gamma over the
 *                         sorted F*T values (numpy's 'linear' method: q_lo and q_gamma as numpy derives them from q and F*T)
 * nperseg: 64..2048 (a power of two: radix-2 transform; any other length: direct fp64 transform of the needed bins, as
 * scipy.signal.stft takes any length, ava/preprocessing/utils.py:66-68), 0 <= noverlap < nperseg; T <= 512.
 * All arithmetic is fp64.  Windows the reference answers with zeros (utils.py:68-69) come out as zeros. */
size_t ava_spec_workspace_bytes(int n, int max_samples, int nperseg, int noverlap, int F, int T, int normalize);
int ava_get_spec_batch(const void* audio, int audio_dtype, const int64_t* file_off, const int64_t* file_len,
                       const int32_t* file_idx, const double* t1, const double* t2, const double* target_times,
                       int n, int max_samples, double fs, int nperseg, int noverlap, const double* window,
                       double scale, const double* target_freqs, int F, int T, double spec_min, double spec_max,
                       double fill_value, int remove_dc, int normalize, int q_lo, double q_gamma, float* out,
                       float* out_max, void* ws, size_t ws_bytes, ava_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* AVA_HIP_H */
