// Stand-alone reproducer attempt for profiles/NOTES.md item 43: NO product code.
//
// Question: do two PROCESSES with kernels on one MI355X at the same time perturb long-lived register accumulators of a
// kernel shaped like convt7's training forward (thin_8to1_direct_fold_kernel: __launch_bounds__(256, 2), 36 packed fp32
// accumulators = 72 VGPRs carried across a persistent tile loop, ten 16-byte global loads per thread and tile kept in
// registers in two affine forms, an LDS exchange of 24 partial sums per thread, a second LDS tile, three workgroup barriers per
// tile, v_pk_fma_f32 throughout)?  If this kernel shows it, the cause is the platform; if it does not while the product
// kernel does, the cause is in the product kernel.
//
// The parent forks two children BEFORE anything touches the GPU (no exec of a GPU process, children only).  Each child
// builds identical inputs, runs the kernel `launches` times per repeat and compares EVERY accumulator of EVERY thread (not
// sums) with its own first launch, bit for bit.  Modes:
//   lockstep  the two children meet at a process-shared barrier before every launch (kernels of both on the chip together)
//   turns     the same, but a process-shared mutex is held from launch to completion (never both on the chip)
//   solo      one child only
//   streams   one child, and a neighbour kernel on a SECOND STREAM of the same process beside every launch
//             (argv[6]: 0 bf16 MFMA, 1 fp32 MFMA, 2 LDS traffic, 3 packed FMAs, 4 the product's limb GEMM,
//             5 v_cvt_pk_bf16_f32, 6 v_pk_add_f32 with op_sel + v_mov_b64, 7 and/sub/shift/max, 8 MFMA + v_cvt_pk_bf16_f32)
// Output: per child, launches that differ, and for the first few of them which workgroups / waves / accumulator indices.
//
// build: hipcc -O3 --offload-arch=gfx950 -o tools/lab/two_proc_repro tools/lab/two_proc_repro.hip -lpthread \
//        -Lautoencoded-vocal-analysis_amd/csrc -lava_hip -Wl,-rpath,'$ORIGIN/../../autoencoded-vocal-analysis_amd/csrc'   (the library only for neighbour kind 4)
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int W = 128, TH = 8, IR = TH + 2, IC = W + 2, NT = 2 * W, NACC = 36;

struct Args {
  const float* in;      // [B][H][W][8]
  const float* ex;      // [B][H][W]
  const float* G;       // [9][8]
  float* out2;          // [B][H][W]
  float* accs;          // [grid][NT][72]: every thread's accumulators, unreduced
  int B, H, ntiles;
};

__global__ __launch_bounds__(NT, 2) void fold_like_kernel(const Args a) {
  __shared__ float U[2][3][TH][IC];
  __shared__ float dUt[TH * IC];
  const int t = threadIdx.x, h = t & 1, x = t >> 1;
  if (t < 2 * 3 * TH) { float* row = &U[0][0][0][0] + t * IC; row[0] = 0.f; row[IC - 1] = 0.f; }
  if (t < TH) { dUt[t * IC] = 0.f; dUt[t * IC + IC - 1] = 0.f; }
  float ca[4], cb[4], ha[4], hb[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    ca[c] = 0.75f + 0.03125f * (4 * h + c); cb[c] = -0.125f * (c + 1);
    ha[c] = 1.25f - 0.0625f * (4 * h + c);  hb[c] = 0.0625f * (c - 2);
  }
  f2 w2[9][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f2 v = {a.G[tap * 8 + 4 * h + 2 * q], a.G[tap * 8 + 4 * h + 2 * q + 1]};
      asm volatile("" : "+v"(v));
      w2[tap][q] = v;
    }
  const int xo = t % W, r0 = (t / W) * 4;
  f2 facc[9][2];
#pragma unroll
  for (int k = 0; k < 9; ++k) facc[k][0] = facc[k][1] = f2{0.f, 0.f};
  float s1 = 0.f;
  const int tiles_y = a.H / TH;
  // static partition: workgroup g owns tiles g, g + grid, ...
  for (int tl = blockIdx.x; tl < a.ntiles; tl += gridDim.x) {
    const int b = tl / tiles_y, oy0 = (tl - b * tiles_y) * TH;
    const float* __restrict__ xin = a.in + ((size_t)b * a.H * W + x) * 8 + 4 * h;
    f2 xn[IR][2], xh[IR][2];
#pragma unroll
    for (int j = 0; j < IR; ++j) {
      const int gy = oy0 - 1 + j;
      const bool ok = gy >= 0 && gy < a.H;
      const f4 v = *reinterpret_cast<const f4*>(xin + (size_t)min(max(gy, 0), a.H - 1) * W * 8);
      xn[j][0] = ok ? f2{fmaf(ca[0], v[0], cb[0]), fmaf(ca[1], v[1], cb[1])} : f2{0.f, 0.f};
      xn[j][1] = ok ? f2{fmaf(ca[2], v[2], cb[2]), fmaf(ca[3], v[3], cb[3])} : f2{0.f, 0.f};
      xh[j][0] = ok ? f2{fmaf(ha[0], v[0], hb[0]), fmaf(ha[1], v[1], hb[1])} : f2{0.f, 0.f};
      xh[j][1] = ok ? f2{fmaf(ha[2], v[2], hb[2]), fmaf(ha[3], v[3], hb[3])} : f2{0.f, 0.f};
    }
    const size_t opix0 = ((size_t)b * a.H + oy0 + r0) * W + xo;
    float ex[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) ex[p] = a.ex[opix0 + (size_t)p * W];
    float u[TH][3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        f2 sacc = {0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int q = 0; q < 2; ++q) sacc = __builtin_elementwise_fma(xn[r + ky][q], w2[ky * 3 + kx][q], sacc);
        u[r][kx] = sacc[0] + sacc[1];
      }
    __syncthreads();
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int r = 0; r < TH; ++r) U[h][kx][r][x + 1] = u[r][kx];
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float v = 0.01f;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) v += U[0][kx][r0 + p][xo + kx] + U[1][kx][r0 + p][xo + kx];
      const float r = v - ex[p];
      const float sd = 0.1f * r;
      a.out2[opix0 + (size_t)p * W] = sd;
      dUt[(r0 + p) * IC + xo + 1] = sd;
      s1 = fmaf(r, r, s1);
    }
    __syncthreads();
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float d[TH];
#pragma unroll
      for (int r = 0; r < TH; ++r) d[r] = dUt[r * IC + x + 2 - kx];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int r = 0; r < TH; ++r) {
          f2 dv = {d[r], d[r]};
#ifdef SAFE_SPLAT
          asm volatile("" : "+v"(dv));                     // the splat becomes a real aligned register pair: no op_sel on the FMA's operand
#endif
#pragma unroll
          for (int q = 0; q < 2; ++q) facc[ky * 3 + kx][q] = __builtin_elementwise_fma(xh[r + ky][q], dv, facc[ky * 3 + kx][q]);
        }
    }
  }
  float* dst = a.accs + ((size_t)blockIdx.x * NT + t) * 73;
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int q = 0; q < 2; ++q) { dst[k * 4 + 2 * q] = facc[k][q][0]; dst[k * 4 + 2 * q + 1] = facc[k][q][1]; }
  dst[72] = s1;
}

// ---- neighbours for mode "streams" (ONE process, second stream): which instruction mix beside the kernel disturbs it? ----
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// kind 0: back-to-back v_mfma_f32_16x16x32_bf16 on registers; 1: v_mfma_f32_16x16x4_f32; 2: ds_read_b128 / ds_write_b128 traffic over
// 64 KB of LDS, no matrix instruction; 3: plain packed FMAs (no matrix, no LDS)
__global__ __launch_bounds__(512, 2) void neighbour_kernel(float* out, int iters, int kind) {
  extern __shared__ __align__(16) unsigned char nsm[];
  const int t = threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {1.f, 1.f, 1.f, 1.f};
  if (kind == 0) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (t + i)); b[i] = (__bf16)(0.002f * (t - i)); }
    for (int i = 0; i < iters; ++i) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc2, 0, 0, 0);
    }
  } else if (kind == 1) {
    const float a = 0.001f * t, b = 0.002f * t;
    for (int i = 0; i < iters; ++i) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc2, 0, 0, 0);
    }
  } else if (kind == 2) {
    f4* l = reinterpret_cast<f4*>(nsm);
    for (int i = t; i < 4096; i += 512) l[i] = f4{(float)i, 1.f, 2.f, 3.f};
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
      const f4 v = l[(t * 7 + i * 13) & 4095];
      acc += v;
      l[(t + 512 * (i & 7)) & 4095] = acc;
    }
  } else if (kind == 5) {                                 // v_cvt_pk_bf16_f32 (the limb split's conversion), back to back
    float x = 0.001f * t, y = 1.5f + 0.002f * t; unsigned r = 0, r2 = 0;
    for (int i = 0; i < iters * 4; ++i) {
      asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %3, %2" : "=v"(r), "=v"(r2) : "v"(x), "v"(y));
      x += 1.f; y += __uint_as_float((r & 0xffff0000u)) * 1e-9f + __uint_as_float(r2 << 16) * 1e-9f;
    }
    acc[0] = x + y;
  } else if (kind == 6) {                                 // v_pk_add_f32 / v_mov_b64
    f2 x = {0.5f, 0.25f}, y = {1.0001f, 0.9999f};
    for (int i = 0; i < iters * 4; ++i) {
      asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_mov_b64 %1, %0\n\tv_pk_add_f32 %1, %1, %0 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(x), "+v"(y));
    }
    acc[0] = x[0] + y[1];
  } else if (kind == 7) {                                 // the limb split's plain-VALU mix: v_and / v_sub_f32 / v_lshlrev / v_max
    float x = 0.001f * t, y = 0.f; unsigned m = 0xffff0000u;
    for (int i = 0; i < iters * 4; ++i) {
      const float hi = __uint_as_float(__float_as_uint(x) & m);
      const float lo = x - hi;
      y = fmaxf(y, __uint_as_float(__float_as_uint(lo) << 1));
      x += 0.37f;
    }
    acc[0] = x + y;
  } else if (kind == 9 || kind == 10) {                   // MFMA interleaved with a plain v_add_f32 (9: bf16 MFMA, 10: fp32 MFMA)
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (t + i)); b[i] = (__bf16)(0.002f * (t - i)); }
    float x = 0.001f * t;
    for (int i = 0; i < iters; ++i) {
      if (kind == 9) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.5f, acc, 0, 0, 0);
      asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(x));
    }
    acc2[0] = x;
  } else if (kind == 8) {                                 // MFMA interleaved with v_cvt_pk_bf16_f32 (what a staging + matrix wave pair issues)
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (t + i)); b[i] = (__bf16)(0.002f * (t - i)); }
    float x = 0.001f * t, y = 1.5f; unsigned r = 0;
    for (int i = 0; i < iters; ++i) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
      x += __uint_as_float(r << 16) * 1e-9f + 1.f;
    }
    acc2[0] = x;
  } else {
    f2 x = {0.5f, 0.25f}, y = {1.0001f, 0.9999f}, z = {0.f, 0.f};
    for (int i = 0; i < iters * 4; ++i) { z = __builtin_elementwise_fma(x, y, z); x = __builtin_elementwise_fma(z, y, x); }
    acc[0] = z[0] + x[1];
  }
  out[(size_t)blockIdx.x * 512 + t] = acc[0] + acc[1] + acc[2] + acc[3] + acc2[0] + acc2[3];
}

// neighbour kind 4: the PRODUCT's limb GEMM (libava_hip.so: ava_gemm, 256 x 1024 x 8192) -- the one kernel family that disturbs the
// product's convt7 forward from a second stream (tools/lab/two_proc_fold.cpp); does it disturb this stand-alone kernel too?
extern "C" size_t ava_gemm_workspace_bytes(int M, int N, int K);
extern "C" int ava_gemm(const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc, const float* mask,
                        float* colsum, int M, int N, int K, int a_kmajor, int b_kmajor, int act, void* ws, size_t ws_bytes, void* s);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); _exit(3); } } while (0)

struct Shared {
  pthread_barrier_t bar;
  pthread_mutex_t turn;
  int bad[2];
};

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }

static int child(int rank, Shared* sh, int nproc, bool turns, int reps, int launches, int B, int grid, int nkind = -1) {
  CK(hipSetDevice(0));
  const int H = 128;
  const size_t nin = (size_t)B * H * W * 8, npix = (size_t)B * H * W, nacc = (size_t)grid * NT * 73;
  std::vector<float> hin(nin), hex(npix), hG(72);
  uint32_t s = 12345u;
  for (auto& v : hin) v = ((int)(lcg(s) >> 9) - (1 << 22)) * (1.f / (1 << 22));
  for (auto& v : hex) v = ((int)(lcg(s) >> 9) - (1 << 22)) * (1.f / (1 << 22));
  for (auto& v : hG) v = ((int)(lcg(s) >> 9) - (1 << 22)) * (0.3f / (1 << 22));
  float *din, *dex, *dG, *dout2, *dacc;
  CK(hipMalloc(&din, nin * 4)); CK(hipMalloc(&dex, npix * 4)); CK(hipMalloc(&dG, 72 * 4));
  CK(hipMalloc(&dout2, npix * 4)); CK(hipMalloc(&dacc, nacc * 4));
  CK(hipMemcpy(din, hin.data(), nin * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dex, hex.data(), npix * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dG, hG.data(), 72 * 4, hipMemcpyHostToDevice));
  Args a{din, dex, dG, dout2, dacc, B, H, B * (H / TH)};
  std::vector<float> ref(nacc), got(nacc);
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // reference: alone on the chip (the other child waits)
  if (nproc == 2) pthread_mutex_lock(&sh->turn);
  CK(hipMemsetAsync(dacc, 0xFF, nacc * 4, st));
  fold_like_kernel<<<grid, NT, 0, st>>>(a);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(ref.data(), dacc, nacc * 4, hipMemcpyDeviceToHost));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < 10; ++i) fold_like_kernel<<<grid, NT, 0, st>>>(a);
  CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  if (nproc == 2) pthread_mutex_unlock(&sh->turn);
  printf("child %d: B=%d grid=%d kernel %.1f us alone\n", rank, B, grid, ms * 100.f); fflush(stdout);
  int bad = 0, shown = 0;
  hipStream_t st2 = nullptr; float* nout = nullptr;
  if (nkind >= 0) {
    CK(hipStreamCreate(&st2)); CK(hipMalloc(&nout, 512 * 512 * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(neighbour_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  }
  float *gA = nullptr, *gB = nullptr, *gC = nullptr; void* gws = nullptr; size_t gwsb = 0;
  if (nkind == 4) {
    CK(hipMalloc(&gA, 256 * 8192 * 4)); CK(hipMalloc(&gB, (size_t)1024 * 8192 * 4)); CK(hipMalloc(&gC, 256 * 1024 * 4));
    CK(hipMemset(gA, 0, 256 * 8192 * 4)); CK(hipMemset(gB, 0, (size_t)1024 * 8192 * 4));
    gwsb = ava_gemm_workspace_bytes(256, 1024, 8192); CK(hipMalloc(&gws, gwsb + 16));
    CK(hipDeviceSynchronize());
  }
  for (int rep = 0; rep < reps; ++rep) {
    for (int l = 0; l < launches; ++l) {
      if (nproc == 2) pthread_barrier_wait(&sh->bar);
      if (turns) pthread_mutex_lock(&sh->turn);
      CK(hipMemsetAsync(dacc, 0xFF, nacc * 4, st));
      if (nkind == 4) { for (int i = 0; i < 3; ++i) if (ava_gemm(gA, 0, gB, 0, nullptr, gC, 0, nullptr, nullptr, 256, 1024, 8192, 1, 1, 0, gws, gwsb, st2) != 0) _exit(6); }
      else if (nkind >= 0) neighbour_kernel<<<256, 512, nkind == 2 ? 65536 : 0, st2>>>(nout, 4000, nkind);     // one workgroup per CU, ~50-100 us
      fold_like_kernel<<<grid, NT, 0, st>>>(a);
      CK(hipStreamSynchronize(st));
      if (nkind >= 0) CK(hipStreamSynchronize(st2));
      if (turns) pthread_mutex_unlock(&sh->turn);
      CK(hipMemcpy(got.data(), dacc, nacc * 4, hipMemcpyDeviceToHost));
      if (memcmp(got.data(), ref.data(), nacc * 4) != 0) {
        ++bad;
        if (shown < 5) {
          ++shown;
          int nwg = 0; long nvals = 0; int idxhist[73] = {0};
          for (int g = 0; g < grid; ++g) {
            bool wgbad = false;
            for (int t = 0; t < NT; ++t)
              for (int k = 0; k < 73; ++k) {
                const size_t o = ((size_t)g * NT + t) * 73 + k;
                if (memcmp(&got[o], &ref[o], 4) != 0) { wgbad = true; ++nvals; ++idxhist[k];
                  if (nvals <= 6) printf("  child %d rep %d launch %d: wg %d thread %d (wave %d lane %d) acc %d: %.9g vs ref %.9g\n", rank, rep, l, g, t, t >> 6, t & 63, k, got[o], ref[o]); }
              }
            nwg += wgbad;
          }
          printf("child %d rep %d launch %d: %d workgroups, %ld values differ; accumulator indices:", rank, rep, l, nwg, nvals);
          for (int k = 0; k < 73; ++k) if (idxhist[k]) printf(" %d(x%d)", k, idxhist[k]);
          printf("\n"); fflush(stdout);
        }
      }
    }
  }
  printf("child %d: %d of %d launches differ from the reference launch\n", rank, bad, reps * launches); fflush(stdout);
  sh->bad[rank] = bad;
  return 0;
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "lockstep";
  const int reps = argc > 2 ? atoi(argv[2]) : 20, launches = argc > 3 ? atoi(argv[3]) : 50;
  const int B = argc > 4 ? atoi(argv[4]) : 8, grid = argc > 5 ? atoi(argv[5]) : 128;
  const bool streams = strcmp(mode, "streams") == 0;
  const int nkind = streams ? (argc > 6 ? atoi(argv[6]) : 0) : -1;
  const int nproc = (strcmp(mode, "solo") == 0 || streams) ? 1 : 2;
  const bool turns = strcmp(mode, "turns") == 0;
  Shared* sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  pthread_barrierattr_t ba; pthread_barrierattr_init(&ba); pthread_barrierattr_setpshared(&ba, PTHREAD_PROCESS_SHARED);
  pthread_barrier_init(&sh->bar, &ba, nproc);
  pthread_mutexattr_t ma; pthread_mutexattr_init(&ma); pthread_mutexattr_setpshared(&ma, PTHREAD_PROCESS_SHARED);
  pthread_mutex_init(&sh->turn, &ma);
  sh->bad[0] = sh->bad[1] = 0;
  pid_t pids[2];
  for (int r = 0; r < nproc; ++r) {
    pids[r] = fork();                                   // before any HIP call in this process
    if (pids[r] == 0) _exit(child(r, sh, nproc, turns, reps, launches, B, grid, nkind));
  }
  int rc = 0;
  for (int r = 0; r < nproc; ++r) { int st = 0; waitpid(pids[r], &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1; }
  if (streams) printf("neighbour kind %d (0 bf16 MFMA, 1 fp32 MFMA, 2 LDS traffic, 3 packed FMA)\n", nkind);
  printf("mode %s B=%d grid=%d: launches that differ: child0 %d, child1 %d of %d each (children ok: %s)\n", mode, B, grid, sh->bad[0], sh->bad[1],
         reps * launches, rc == 0 ? "yes" : "NO");
  return rc;
}
