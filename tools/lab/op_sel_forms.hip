// Which operand-select forms of the packed fp32 instructions are executed wrongly on gfx950 while a wave of ANOTHER kernel (eight
// neighbour kinds: one matrix instruction type each) that
// alternates v_mfma_f32_16x16x32_bf16 with vector-ALU instructions shares the SIMD?  (profiles/NOTES.md item 44; the effect was
// found as op_sel:[0,1,0] on v_pk_fma_f32 in tools/lab/two_proc_repro.hip.)  One process, two streams: the victim kernel runs
// every form in a long loop on known operands and compares with the scalar result computed by plain v_fma_f32 in the same lane;
// the neighbour runs beside it.  Output: per form, wrong results (count, lanes), with and without the neighbour.
//
// build: hipcc -O2 --offload-arch=gfx950 -o tools/lab/op_sel_forms tools/lab/op_sel_forms.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(3); } } while (0)

constexpr int NFORM = 10;
static const char* kForm[NFORM] = {
    "v_pk_fma_f32 plain", "v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_fma_f32 op_sel:[0,0,1]",
    "v_pk_fma_f32 op_sel_hi:[1,0,1]", "v_pk_fma_f32 op_sel_hi:[0,1,1]", "v_pk_fma_f32 op_sel_hi:[1,1,0]",
    "v_pk_mul_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel:[1,0]", "v_pk_fma_f32 op_sel:[1,1,0] op_sel_hi:[0,0,1]"};

// scalar reference instructions as inline asm: the compiler must not turn the expectation itself into packed instructions
__device__ __forceinline__ float sfma(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float smul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sadd(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#define fmaf sfma
// per form: expected (lo, hi) from the operand halves it selects
__device__ __forceinline__ f2 expect(int form, f2 a, f2 b, f2 c) {
  switch (form) {
    case 0: return f2{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])};
    case 1: return f2{fmaf(a[0], b[1], c[0]), fmaf(a[1], b[1], c[1])};
    case 2: return f2{fmaf(a[1], b[0], c[0]), fmaf(a[1], b[1], c[1])};
    case 3: return f2{fmaf(a[0], b[0], c[1]), fmaf(a[1], b[1], c[1])};
    case 4: return f2{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[0], c[1])};
    case 5: return f2{fmaf(a[0], b[0], c[0]), fmaf(a[0], b[1], c[1])};
    case 6: return f2{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[0])};
    case 7: return f2{smul(a[0], b[1]), smul(a[1], b[1])};
    case 8: return f2{sadd(a[1], b[0]), sadd(a[1], b[1])};
    default: return f2{fmaf(a[1], b[1], c[0]), fmaf(a[0], b[0], c[1])};
  }
}
#undef fmaf

__global__ __launch_bounds__(256, 2) void victim_kernel(unsigned* bad, int iters) {
  const int t = threadIdx.x, g = blockIdx.x * 256 + t;
  unsigned nbad[NFORM];
#pragma unroll
  for (int f = 0; f < NFORM; ++f) nbad[f] = 0;
  f2 a = {1.0f + 0.001f * (g & 1023), 2.0f - 0.003f * (g & 511)}, b = {0.5f + 0.002f * (g & 255), -1.25f + 0.004f * (g & 127)};
  f2 c = {0.125f * (g & 7), -0.375f * (g & 15)};
  for (int i = 0; i < iters; ++i) {
    f2 r[NFORM];
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r[0]) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r[1]) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r[2]) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(r[3]) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r[4]) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r[5]) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(r[6]) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r[7]) : "v"(a), "v"(b));
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r[8]) : "v"(a), "v"(b));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,0,1]" : "=v"(r[9]) : "v"(a), "v"(b), "v"(c));
#pragma unroll
    for (int f = 0; f < NFORM; ++f) {
      const f2 e = expect(f, a, b, c);
      if (__float_as_uint(e[0]) != __float_as_uint(r[f][0])) nbad[f] += 1;
      if (__float_as_uint(e[1]) != __float_as_uint(r[f][1])) nbad[f] += 0x10000;
    }
    a[0] += 0.0009765625f; b[1] -= 0.001953125f; c[0] += 0.25f;
  }
#pragma unroll
  for (int f = 0; f < NFORM; ++f) bad[(size_t)g * NFORM + f] = nbad[f];
}

// neighbours: one matrix instruction type each, back to back (the loop counter is scalar); kind 1 adds one plain vector instruction per
// matrix instruction
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f16acc __attribute__((ext_vector_type(16)));
static const char* kNeigh[] = {"v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x32_bf16 + v_add_f32", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_16x16x32_f16",
                               "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x2_f32", "v_mfma_f32_16x16x16_bf16 (the 8-pass bf16 form)", "no matrix instruction: v_fma_f32 only"};
constexpr int NNEIGH = 8;
__global__ __launch_bounds__(512, 2) void neighbour_kernel(float* out, int iters, int kind) {
  const int t = threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  f16acc acc16 = {};
  bf16x8 a, b;
  f16x8 ha, hb;
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4 a4, b4;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (t + i)); b[i] = (__bf16)(0.002f * (t - i)); ha[i] = (_Float16)(0.001f * (t + i)); hb[i] = (_Float16)(0.002f * (t - i)); }
  for (int i = 0; i < 4; ++i) { a4[i] = a[i]; b4[i] = b[i]; }
  float x = 0.001f * t;
  for (int i = 0; i < iters; ++i) {
    if (kind <= 1) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    else if (kind == 2) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, 0.5f, acc, 0, 0, 0);
    else if (kind == 3) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc, 0, 0, 0);
    else if (kind == 4) acc16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc16, 0, 0, 0);
    else if (kind == 5) acc16 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 0.5f, acc16, 0, 0, 0);
    else if (kind == 6) acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc, 0, 0, 0);
    else asm volatile("v_fma_f32 %0, %0, 1.0, 1.0" : "+v"(x));
    if (kind == 1) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(x));
  }
  out[(size_t)blockIdx.x * 512 + t] = acc[0] + acc[3] + x + acc16[0] + acc16[15];
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 20;
  CK(hipSetDevice(0));
  const int grid = 512, iters = 2000;
  unsigned* dbad; float* nout;
  CK(hipMalloc(&dbad, (size_t)grid * 256 * NFORM * 4)); CK(hipMalloc(&nout, 512 * 512 * 4));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  std::vector<unsigned> h((size_t)grid * 256 * NFORM);
  for (int nk = -1; nk < NNEIGH; ++nk) {
    unsigned long lo[NFORM] = {0}, hi[NFORM] = {0}; unsigned lanes[NFORM][4] = {{0}};
    for (int l = 0; l < launches; ++l) {
      if (nk >= 0) neighbour_kernel<<<256, 512, 0, s2>>>(nout, 6000, nk);
      victim_kernel<<<grid, 256, 0, s1>>>(dbad, iters);
      CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
      CK(hipMemcpy(h.data(), dbad, h.size() * 4, hipMemcpyDeviceToHost));
      for (size_t g = 0; g < (size_t)grid * 256; ++g)
        for (int f = 0; f < NFORM; ++f) {
          const unsigned v = h[g * NFORM + f];
          if (v) { lo[f] += v & 0xffff; hi[f] += v >> 16; lanes[f][(g & 63) >> 4] += 1; }
        }
    }
    printf("neighbour: %s\n", nk < 0 ? "none" : kNeigh[nk]);
    for (int f = 0; f < NFORM; ++f)
      printf("  %-52s wrong low halves %8lu, wrong high halves %8lu   (threads hit, by lane quarter 0-15 / 16-31 / 32-47 / 48-63: %u %u %u %u)\n",
             kForm[f], lo[f], hi[f], lanes[f][0], lanes[f][1], lanes[f][2], lanes[f][3]);
  }
  return 0;
}
