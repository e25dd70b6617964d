// Second stage of the two-process investigation (profiles/NOTES.md items 43, 44): the PRODUCT's forward, without torch.
//
// tools/lab/two_proc_repro.hip (a stand-alone kernel with the fold kernel's shape) does not show the effect; the product under
// torch (tools/lab/share_probe.py) does, about once in 240 forwards per process.  This harness removes torch and Python: two
// forked children (forked before anything touches the GPU) link libava_hip.so, create a model through the C ABI and run
// ava_forward over and over -- lock-stepped at a process-shared barrier, taking turns under a mutex, or alone -- and compare
// what convt7's training forward leaves behind (its 1024 x 73 weight-gradient partial rows), the loss words and the
// reconstruction with their own first forward, bit for bit.
//
// build: hipcc -O2 -o tools/lab/two_proc_fold tools/lab/two_proc_fold.cpp -Iinclude -Lautoencoded-vocal-analysis_amd/csrc \
//        -lava_hip -Wl,-rpath,'$ORIGIN/../../autoencoded-vocal-analysis_amd/csrc' -lpthread
// usage: two_proc_fold <lockstep|turns|solo|mixed|pair|streams> [forwards] [B] [backward 0|1] [gemm|conv|thin|adam]
//   mixed: child 1 does not wait at the barrier (free-running: its kernels overlap child 0's at arbitrary phases)
//   pair: child 1 is a second PROCESS that launches ONE kernel family over and over (which family disturbs the forward?)
//   streams: ONE process, the same kernel family on a second stream beside the forwards (is a second process needed at all?)
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ava_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); _exit(3); } } while (0)
#define AK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "ava error %d at %s:%d\n", r_, __FILE__, __LINE__); _exit(4); } } while (0)

struct Shared { pthread_barrier_t bar; pthread_mutex_t turn; int bad[2]; volatile int done[2]; };

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
static float u11(uint32_t& s) { return ((int)(lcg(s) >> 9) - (1 << 22)) * (1.f / (1 << 22)); }

// A "neighbour": one kernel family launched over and over (the other process of mode pair, or the second stream of mode
// streams): gemm = the limb GEMM (bf16 MFMA 16x16x32, 512-thread wave-specialised workgroups), conv = a matrix-core conv
// forward (16 -> 16 channels, 64 x 64), adam = the flat streaming update (VALU + HBM only), thin = conv1's packed-FMA forward.
struct Neighbour {
  const char* kind; hipStream_t st;
  float *A = nullptr, *Bm = nullptr, *C = nullptr, *G = nullptr, *pa = nullptr, *pb = nullptr, *part = nullptr; void* ws = nullptr; size_t wsb = 0;
  int64_t n = 0;
  void init(const char* k, hipStream_t s) {
    kind = k; st = s;
    if (!strcmp(k, "gemm")) {
      CK(hipMalloc(&A, 256 * 8192 * 4)); CK(hipMalloc(&Bm, (size_t)1024 * 8192 * 4)); CK(hipMalloc(&C, 256 * 1024 * 4));
      CK(hipMemset(A, 0, 256 * 8192 * 4)); CK(hipMemset(Bm, 0, (size_t)1024 * 8192 * 4));
      wsb = ava_gemm_workspace_bytes(256, 1024, 8192); CK(hipMalloc(&ws, wsb + 16));
    } else if (!strcmp(k, "conv") || !strcmp(k, "thin")) {
      const int B = 64, ci = !strcmp(k, "thin") ? 1 : 16, co = !strcmp(k, "thin") ? 8 : 16, hw = !strcmp(k, "thin") ? 128 : 64;
      CK(hipMalloc(&A, (size_t)B * hw * hw * ci * 4)); CK(hipMalloc(&C, (size_t)B * hw * hw * co * 4)); CK(hipMalloc(&G, 9 * ci * co * 4));
      CK(hipMalloc(&pa, 128)); CK(hipMalloc(&pb, 128)); CK(hipMalloc(&Bm, 128));
      CK(hipMemset(A, 0, (size_t)B * hw * hw * ci * 4)); CK(hipMemset(G, 0, 9 * ci * co * 4)); CK(hipMemset(pa, 0, 128)); CK(hipMemset(pb, 0, 128)); CK(hipMemset(Bm, 0, 128));
      const int grid = ava_conv_grid(B, hw, hw, 0);
      CK(hipMalloc(&part, (size_t)grid * 2 * co * 4));
    } else if (!strcmp(k, "adam")) {
      n = 16 << 20;
      CK(hipMalloc(&A, n * 4)); CK(hipMalloc(&Bm, n * 4)); CK(hipMalloc(&C, n * 4)); CK(hipMalloc(&G, n * 4));
      CK(hipMemset(A, 0, n * 4)); CK(hipMemset(Bm, 0, n * 4)); CK(hipMemset(C, 0, n * 4)); CK(hipMemset(G, 0, n * 4));
    }
  }
  void launch() {
    if (!strcmp(kind, "gemm")) AK(ava_gemm(A, 0, Bm, 0, nullptr, C, 0, nullptr, nullptr, 256, 1024, 8192, 1, 1, 0, ws, wsb, st));
    else if (!strcmp(kind, "conv")) AK(ava_conv3x3(A, nullptr, pa, pb, nullptr, G, Bm, C, nullptr, nullptr, nullptr, nullptr, part, 64, 64, 64, 16, 16, 0, 0, 0, 1, 0.f, st));
    else if (!strcmp(kind, "thin")) AK(ava_conv3x3(A, nullptr, pa, pb, nullptr, G, Bm, C, nullptr, nullptr, nullptr, nullptr, part, 64, 128, 128, 1, 8, 0, 0, 0, 1, 0.f, st));
    else if (!strcmp(kind, "adam")) AK(ava_adam_flat(A, Bm, C, G, n, 1e-3, 0.9, 0.999, 1e-8, 1, st));
  }
};

static int neighbour_child(Shared* sh, const char* kind) {
  CK(hipSetDevice(0));
  hipStream_t st; CK(hipStreamCreate(&st));
  pthread_mutex_lock(&sh->turn);                           // (its set-up memsets are kernels too: not beside child 0's reference)
  Neighbour nb; nb.init(kind, st);
  CK(hipDeviceSynchronize());
  pthread_mutex_unlock(&sh->turn);
  long launches = 0;
  pthread_barrier_wait(&sh->bar);                          // child 0 has taken its reference
  while (!sh->done[0]) { for (int i = 0; i < 8; ++i) nb.launch(); CK(hipStreamSynchronize(st)); launches += 8; }
  printf("neighbour (%s): %ld launches\n", kind, launches); fflush(stdout);
  return 0;
}

static int child(int rank, Shared* sh, const char* mode, int forwards, int B, int with_bwd, const char* nkind = nullptr) {
  const bool turns = strcmp(mode, "turns") == 0, solo = strcmp(mode, "solo") == 0, mixed = strcmp(mode, "mixed") == 0;
  const bool pair = strcmp(mode, "pair") == 0, streams = strcmp(mode, "streams") == 0;
  const bool lock = !solo && !streams && !pair && !(mixed && rank == 1);
  CK(hipSetDevice(0));
  const int z = 32, H = 128, W = 128;
  const int64_t total = ava_arena_floats_hw(z, H, W);
  const size_t wsb = ava_workspace_bytes_hw(z, H, W, B);
  float *params, *grads, *m1, *m2, *bnr, *x, *ew, *ed, *loss;
  int64_t* bnb; void* ws; int* status;
  CK(hipMalloc(&params, total * 4)); CK(hipMalloc(&grads, total * 4)); CK(hipMalloc(&m1, total * 4)); CK(hipMalloc(&m2, total * 4));
  CK(hipMalloc(&bnr, 2 * 14 * 32 * 4)); CK(hipMalloc(&bnb, 14 * 8)); CK(hipMalloc(&ws, wsb));
  CK(hipMalloc(&x, (size_t)B * H * W * 4)); CK(hipMalloc(&ew, B * 4)); CK(hipMalloc(&ed, B * z * 4)); CK(hipMalloc(&loss, 16)); CK(hipMalloc(&status, 8));
  std::vector<float> hp(total), hx((size_t)B * H * W), hn(B * (z + 1)), hb(2 * 14 * 32);
  uint32_t s = 777u;
  for (auto& v : hp) v = 0.05f * u11(s);
  // BatchNorm weights 1, biases 0 would hide nothing; give gammas a spread around 1 instead
  for (int idx = 0; idx < 80; ++idx) {
    int64_t n = 0; const int64_t o = ava_param_offset_hw(z, H, W, idx, &n);
    (void)o; (void)n;
  }
  for (auto& v : hx) v = 0.5f + 0.5f * u11(s);
  for (auto& v : hn) v = u11(s);
  for (int i = 0; i < 14 * 32; ++i) { hb[i] = 0.f; hb[14 * 32 + i] = 1.f; }
  CK(hipMemcpy(params, hp.data(), total * 4, hipMemcpyHostToDevice));
  CK(hipMemset(grads, 0, total * 4)); CK(hipMemset(m1, 0, total * 4)); CK(hipMemset(m2, 0, total * 4));
  CK(hipMemcpy(bnr, hb.data(), hb.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(bnb, 0, 14 * 8));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(ew, hn.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ed, hn.data() + B, B * z * 4, hipMemcpyHostToDevice));
  CK(hipMemset(status, 0, 8));
  ava_model* m = nullptr;
  AK(ava_model_create_ex(&m, z, H, W, 0, B, 10.f, params, grads, m1, m2, bnr, bnb, ws, wsb));
  hipStream_t st; CK(hipStreamCreate(&st));
  int64_t nf8 = 0;
  // run one forward first: the debug buffers exist after the workspace was carved
  const bool alone = solo || streams;
  if (!alone) pthread_mutex_lock(&sh->turn);
  AK(ava_forward(m, x, B, ew, ed, 1, loss, nullptr, status, st));
  CK(hipStreamSynchronize(st));
  const float* wg13 = ava_debug_buffer(m, "wg13", &nf8);  // convt7's weight-gradient partial rows, 73 floats each
  if (wg13 == nullptr) { fprintf(stderr, "no wg13 buffer\n"); _exit(5); }
  const size_t nwg = (size_t)nf8, nrec = (size_t)B * H * W;
  const int nrows = (int)(nwg / 73);
  std::vector<float> ref(nwg), got(nwg), refx(nrec), gotx(nrec), refg(total), gotg(total);
  float refl[4], gotl[4];
  // the reference forward: alone on the chip.  BatchNorm running statistics move with every forward but do not enter a
  // training forward's arithmetic, so every later forward must reproduce this one bit for bit.
  AK(ava_forward(m, x, B, ew, ed, 1, loss, nullptr, status, st));
  if (with_bwd) AK(ava_backward(m, x, B, st));
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(ref.data(), wg13, nwg * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(refx.data(), ava_last_xrec(m), nrec * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(refl, loss, 16, hipMemcpyDeviceToHost));
  if (with_bwd) CK(hipMemcpy(refg.data(), grads, total * 4, hipMemcpyDeviceToHost));
  if (!alone) pthread_mutex_unlock(&sh->turn);
  printf("child %d: B=%d loss %.6g, reference taken\n", rank, B, refl[0]); fflush(stdout);
  Neighbour nb; hipStream_t st2 = nullptr;
  if (streams) { CK(hipStreamCreate(&st2)); nb.init(nkind, st2); }
  if (pair) pthread_barrier_wait(&sh->bar);
  int bad = 0, shown = 0, badx = 0, badg = 0;
  for (int f = 0; f < forwards; ++f) {
    if (mixed && rank == 1 && sh->done[0]) break;
    if (lock) pthread_barrier_wait(&sh->bar);
    if (turns) pthread_mutex_lock(&sh->turn);
    if (streams) for (int i = 0; i < 6; ++i) nb.launch();            // the second stream's kernels run beside the forward
    AK(ava_forward(m, x, B, ew, ed, 1, loss, nullptr, status, st));
    if (with_bwd) AK(ava_backward(m, x, B, st));
    if (streams) for (int i = 0; i < 6; ++i) nb.launch();
    CK(hipStreamSynchronize(st));
    if (streams) CK(hipStreamSynchronize(st2));
    if (turns) pthread_mutex_unlock(&sh->turn);
    CK(hipMemcpy(got.data(), wg13, nwg * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gotl, loss, 16, hipMemcpyDeviceToHost));
    if (memcmp(gotl, refl, 16) != 0) ++badx;
    if (with_bwd) { CK(hipMemcpy(gotg.data(), grads, total * 4, hipMemcpyDeviceToHost)); if (memcmp(gotg.data(), refg.data(), total * 4) != 0) ++badg; }
    if (memcmp(got.data(), ref.data(), nwg * 4) != 0) {
      ++bad;
      if (shown < 6) {
        ++shown;
        int rows = 0; int colhist[73] = {0};
        for (int r = 0; r < nrows; ++r) { bool rb = false; for (int c = 0; c < 73; ++c) if (memcmp(&got[r * 73 + c], &ref[r * 73 + c], 4) != 0) { rb = true; ++colhist[c]; } rows += rb; }
        printf("child %d forward %d: %d partial rows differ; columns:", rank, f, rows);
        for (int c = 0; c < 73; ++c) if (colhist[c]) printf(" %d(x%d)", c, colhist[c]);
        printf("\n"); fflush(stdout);
      }
    }
  }
  if ((mixed || pair) && rank == 0) sh->done[0] = 1;
  printf("child %d: fold partials differ in %d forwards, loss words in %d, gradients in %d (of %d)\n", rank, bad, badx, badg, forwards); fflush(stdout);
  sh->bad[rank] = bad + badg;
  ava_model_destroy(m);
  return 0;
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "lockstep";
  const int forwards = argc > 2 ? atoi(argv[2]) : 2000, B = argc > 3 ? atoi(argv[3]) : 8, with_bwd = argc > 4 ? atoi(argv[4]) : 0;
  const char* nkind = argc > 5 ? argv[5] : "gemm";       // pair / streams: what runs beside the forwards
  const bool solo = strcmp(mode, "solo") == 0, mixed = strcmp(mode, "mixed") == 0, pair = strcmp(mode, "pair") == 0, streams = strcmp(mode, "streams") == 0;
  const int nproc = (solo || streams) ? 1 : 2;
  Shared* sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  pthread_barrierattr_t ba; pthread_barrierattr_init(&ba); pthread_barrierattr_setpshared(&ba, PTHREAD_PROCESS_SHARED);
  pthread_barrier_init(&sh->bar, &ba, mixed ? 1 : nproc);
  pthread_mutexattr_t ma; pthread_mutexattr_init(&ma); pthread_mutexattr_setpshared(&ma, PTHREAD_PROCESS_SHARED);
  pthread_mutex_init(&sh->turn, &ma);
  sh->bad[0] = sh->bad[1] = 0; sh->done[0] = sh->done[1] = 0;
  pid_t pids[2];
  for (int r = 0; r < nproc; ++r) {
    pids[r] = fork();                                     // before any HIP call in this process
    if (pids[r] == 0) _exit((pair && r == 1) ? neighbour_child(sh, nkind) : child(r, sh, mode, mixed && r == 1 ? 1000000 : forwards, B, with_bwd, nkind));
  }
  int rc = 0;
  for (int r = 0; r < nproc; ++r) { int st = 0; waitpid(pids[r], &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1; }
  if (pair || streams) printf("neighbour kind: %s\n", nkind);
  printf("mode %s B=%d bwd=%d: forwards that differ: child0 %d, child1 %d (children ok: %s)\n", mode, B, with_bwd, sh->bad[0], sh->bad[1], rc == 0 ? "yes" : "NO");
  return rc;
}
