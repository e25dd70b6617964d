// Shared between gemm.hip (fp32 MFMA kernels, planning, C entry points) and gemm_limb.hip (the fp32-faithful
// three-limb bf16 MFMA kernel for the large products).
#pragma once
#include "common.h"

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_EXP = 2 };

struct GemmArgs {
  const float* A;
  const float* B;
  const float* bias;
  float* C;          // output, or partial slabs [splits][M][N]
  float* colsum;
  const float* mask; // optional: C = mask > 0 ? v : 0 (ReLU backward of the producing layer), leading dim ldc
  int M, N, K;
  int lda, ldb, ldc; // leading dimension (elements) of the stored matrices
  int klen;          // K elements per split (multiple of 16; of 32 for the limb kernel)
  int splits;
  int act;
  int vec_a, vec_b;  // 16-byte loads legal
  int dbg;           // lab build only: ablation bits of the limb kernel (1 no MFMA, 2 no split / LDS write, 4 no global loads, 8 no output)
};

#ifdef __HIPCC__
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_EXP) return expf(v);   // full-precision expf (vae.py:232)
  return v;
}
#endif

// gemm_limb.hip
bool ava_gemm_limb_ok(const GemmArgs& g, int a_kmajor, int b_kmajor);
void ava_gemm_limb_plan(int M, int N, int K, int a_kmajor, int* bn, int* splits, int* klen);
int ava_gemm_limb_launch(const GemmArgs& g, int a_kmajor, int b_kmajor, int bn, hipStream_t st);
