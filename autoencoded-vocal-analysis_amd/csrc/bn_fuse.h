// BatchNorm finalisation fused into the kernel that produced the per-workgroup partial sums: the LAST
// workgroup to publish its partial row reduces all rows (fp64, fixed order -> deterministic) and writes
// the per-channel coefficients the next kernel's prologue consumes.  Removes 28 tiny launches per step.
//
// Hand-off protocol (cdna_hip_programming.md, Guideline 16 / "In-launch split-K reduction", write-through
// form): the partial row is stored with agent-scope (sc1, write-through) stores -- see bn_partial_store --,
// every storing wave drains them (s_waitcnt vmcnt(0)), workgroup barrier, lane 0 does a relaxed agent-scope
// fetch_add on the ticket counter; the workgroup that draws the last ticket does an agent-scope ACQUIRE
// fence, barrier, and only then reads the other workgroups' rows with plain loads.  No release fence: the
// first version had one per workgroup and each of them wrote back the whole XCD L2, which is full of the
// kernel's own freshly written activations (+0.6 ms/step).  Placement independent (no assumption on
// dispatch order or XCD).  The counter is zero on entry (zeroed once at model creation) and the last
// arriver resets it for the next launch on the stream.
#pragma once
#include "common.h"

#define AVA_BN_EPS 1e-5
#define AVA_BN_MOMENTUM 0.1

struct BnFuse {
  int* counter;        // nullptr: no fused finalisation
  int mode;            // 1: forward statistics (train), 2: backward sums
  int C;
  double n;            // elements per channel
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  int64_t* num_batches;
  float* mean;         // fwd: out, bwd: in
  float* invstd;       // fwd: out, bwd: in
  float* scale;        // fwd out
  float* shift;        // fwd out
  float* dgamma;       // bwd out
  float* dbeta;        // bwd out
  float* A;            // bwd out: dx = A*g + Bc*x + Cc
  float* Bc;
  float* Cc;
};

// store one element of a partial row: write-through when a fused finalisation will read it in this launch
__device__ __forceinline__ void bn_partial_store(const BnFuse& f, float* p, float v) {
  if (f.counter != nullptr) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// forward statistics -> mean / invstd / scale / shift (+ running statistics update); c < C
__device__ __forceinline__ void bn_fwd_channel(const BnFuse& f, int c, double s1, double s2) {
  const double mean = s1 / f.n;
  double var = s2 / f.n - mean * mean;          // biased
  if (var < 0.0) var = 0.0;
  if (f.running_mean != nullptr) {
    const double unb = f.n > 1.0 ? var * (f.n / (f.n - 1.0)) : var;
    f.running_mean[c] = (float)((1.0 - AVA_BN_MOMENTUM) * (double)f.running_mean[c] + AVA_BN_MOMENTUM * mean);
    f.running_var[c] = (float)((1.0 - AVA_BN_MOMENTUM) * (double)f.running_var[c] + AVA_BN_MOMENTUM * unb);
  }
  const float meanf = (float)mean;
  const float invstd = (float)(1.0 / sqrt(var + AVA_BN_EPS));
  const float sc = f.gamma[c] * invstd;
  f.mean[c] = meanf;
  f.invstd[c] = invstd;
  f.scale[c] = sc;
  f.shift[c] = f.beta[c] - meanf * sc;
}

// backward sums {sum g, sum g*xhat} -> dgamma, dbeta, and dx = A*g + Bc*x + Cc
__device__ __forceinline__ void bn_bwd_channel(const BnFuse& f, int c, double dB, double dG) {
  const double is = (double)f.invstd[c], gm = (double)f.gamma[c], mu = (double)f.mean[c];
  const double a = gm * is;
  const double b = -gm * is * is * dG / f.n;
  f.dgamma[c] = (float)dG;
  f.dbeta[c] = (float)dB;
  f.A[c] = (float)a;
  f.Bc[c] = (float)b;
  f.Cc[c] = (float)(-a * dB / f.n - b * mu);
}

// Call from EVERY thread of EVERY workgroup (256 threads) after the workgroup's partial row has been stored.
// `scratch`: >= 256 doubles + 4 ints of LDS not otherwise in use.
__device__ __forceinline__ void bn_fused_finalize(const BnFuse& f, const float* __restrict__ partials, int nparts,
                                                  double* scratch) {
  if (f.counter == nullptr) return;
  int* flag = reinterpret_cast<int*>(scratch + 256);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's partial-row stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    const int ticket = __hip_atomic_fetch_add(f.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == (int)gridDim.x - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *flag = last;
  }
  __syncthreads();
  if (*flag == 0) return;
  // ---- last arriver: column sums of partials[nparts][2C] in fp64, fixed order ----
  const int ncols = 2 * f.C;
  const int groups = 256 / ncols;
  const int t = threadIdx.x;
  const int col = t % ncols, grp = t / ncols;
  double s = 0.0;
  if (grp < groups)
    for (int r = grp; r < nparts; r += groups) s += (double)partials[(size_t)r * ncols + col];
  scratch[t] = grp < groups ? s : 0.0;
  __syncthreads();
  if (t < f.C) {
    double a = 0.0, b = 0.0;
    for (int gI = 0; gI < groups; ++gI) { a += scratch[gI * ncols + t]; b += scratch[gI * ncols + f.C + t]; }
    if (f.mode == 1) bn_fwd_channel(f, t, a, b);
    else bn_bwd_channel(f, t, a, b);
  }
  if (t == 0) {
    if (f.mode == 1 && f.num_batches != nullptr) *f.num_batches += 1;
    __hip_atomic_store(f.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  }
}
