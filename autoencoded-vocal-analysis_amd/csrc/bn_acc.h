// BatchNorm sums accumulated INSIDE the producing kernel, finalised in the prologue of the consuming kernel:
// removes the one-workgroup bn_finalize launches (5 us kernel + a dependent-launch boundary each, 28 per step) from
// the step's critical path for the layer pairs whose kernels use it (model.hip: acc_pair_fwd / acc_pair_bwd).
//
// Determinism.  Every workgroup adds its fp32 partial sums with INTEGER atomics, so the total does not depend on the
// order of arrival: a partial p is split exactly into three 32-bit limbs of the fixed-point number |p| * 2^48
// (|p| < 2^48; bits below 2^-48 are truncated, a deterministic function of p) and the signed limbs are added to three
// int64 counters -- 1024 workgroups cannot overflow them.  The reader recombines sum(limb_k) * 2^(32k - 48) in fp64.
// Non-finite partials raise a flag word instead (the reader then yields NaN, as a sum containing them would).
// Contention.  The counters of a slot are replicated over 8 shards (workgroup index mod 8): one shard takes at most
// 1/8 of the arrivals; measured cost of 64 values x 3 limbs at the end of a 512-workgroup kernel: +1.1 us
// (tools/lab/atomic_probe.hip; a single shard costs +9 us).
#pragma once
#include "common.h"

#define AVA_ACC_SHARDS 8
#define AVA_ACC_SHARD_LL 200            // per shard: [3 limbs][64 values] + flag word at 192 (+ padding)
#define AVA_ACC_SLOT_LL (AVA_ACC_SHARDS * AVA_ACC_SHARD_LL)
#define AVA_ACC_SLOTS 29                // BatchNorm l forward statistics: slot l; backward sums: slot 14 + l; slot 28: the
                                        // second buffer of bn1's forward sums (their producer is the launch that zeroes the rest)
#define AVA_BN_EPS_D 1e-5
#define AVA_BN_MOM_D 0.1

// What the consumer needs to turn the accumulated sums of one BatchNorm layer into its prologue coefficients.
struct BnFin {
  const long long* acc;        // slot base; null: the launch reads ready-made coefficient arrays instead
  const float* gamma;
  const float* beta;           // forward only
  const float* mean;           // backward: forward statistics of the layer (bn_save)
  const float* invstd;
  float* save;                 // forward: bn_save row [4][32] = mean, invstd, scale, shift   (published by workgroup 0)
  float* running_mean;         // forward, training: momentum update by workgroup 0 (may be null)
  float* running_var;
  long long* num_batches;
  float* dgamma;               // backward: gradient arena entries (published by workgroup 0)
  float* dbeta;
  float* abc;                  // backward: bn_bwd row [3][32] = A, Bc, Cc (published by workgroup 0)
  double n;                    // elements per channel
  int C;
  int backward;                // 0: sums are {sum x, sum x^2}; 1: {sum g, sum g*xhat}
  int eval;                    // backward of a forward that ran on running statistics: Bc = Cc = 0
};

#ifdef __HIPCC__
// producer: add the workgroup's partial sum of value `idx` (0..63: [which][32 channels]) to the slot
__device__ __forceinline__ void bn_acc_add(long long* slot, int idx, float p) {
  long long* a = slot + (size_t)(blockIdx.x & (AVA_ACC_SHARDS - 1)) * AVA_ACC_SHARD_LL;
  const double ad = fabs((double)p) * 0x1p48;                 // exact
  if (!(ad < 0x1p95)) {                                       // NaN, Inf or |p| >= 2^47: poison the slot
    atomicOr(reinterpret_cast<unsigned long long*>(a + 192), 1ull);
    return;
  }
  const double l2 = floor(ad * 0x1p-64), r1 = ad - l2 * 0x1p64;
  const double l1 = floor(r1 * 0x1p-32), l0 = floor(r1 - l1 * 0x1p32);
  const long long sg = p < 0.f ? -1 : 1;
  if (l0 != 0.0) atomicAdd(reinterpret_cast<unsigned long long*>(a + idx), (unsigned long long)(sg * (long long)l0));
  if (l1 != 0.0) atomicAdd(reinterpret_cast<unsigned long long*>(a + 64 + idx), (unsigned long long)(sg * (long long)l1));
  if (l2 != 0.0) atomicAdd(reinterpret_cast<unsigned long long*>(a + 128 + idx), (unsigned long long)(sg * (long long)l2));
}

// consumer: value `idx` of the slot (fixed shard order: deterministic); NaN when the slot is poisoned.
// Plain loads: the counters were written by atomics of the PREVIOUS kernel, and a kernel boundary makes every earlier write
// visible; the 32 lines of a slot are then served by the XCD's L2 to all but the first workgroup that asks (agent-scope
// atomic loads took every workgroup's request to the memory side).
__device__ __forceinline__ double bn_acc_read(const long long* slot, int idx) {
  long long v0[AVA_ACC_SHARDS], v1[AVA_ACC_SHARDS], v2[AVA_ACC_SHARDS];
  unsigned long long bad = 0;
#pragma unroll
  for (int sh = 0; sh < AVA_ACC_SHARDS; ++sh) {
    const long long* a = slot + (size_t)sh * AVA_ACC_SHARD_LL;
    v0[sh] = a[idx]; v1[sh] = a[64 + idx]; v2[sh] = a[128 + idx];
    bad |= (unsigned long long)a[192];
  }
  double s = 0.0;
#pragma unroll
  for (int sh = 0; sh < AVA_ACC_SHARDS; ++sh) s += ((double)v0[sh] + (double)v1[sh] * 0x1p32 + (double)v2[sh] * 0x1p64) * 0x1p-48;
  return bad ? __builtin_nan("") : s;
}

// Consumer prologue, called by ALL threads of the workgroup (>= 64 threads; contains a barrier): fills coef[3][32]
// (LDS) with {scale, shift, 0} (forward) or {A, Bc, Cc} (backward), zero beyond C, exactly as bn_finalize_kernel /
// bn_finalize_bwd_kernel compute them; workgroup 0 also publishes what later kernels read from global memory.
// The wave of threads t0 .. t0+63 does the work (pick a wave with nothing slow in flight: a wave's loads return in order,
// so reads queued behind a tile prefetch would wait for it): its 64 lanes read the 64 values, lanes 0..31 finalise with
// the upper half's value handed over by a shuffle.  The per-channel parameters are requested BEFORE the counters, so the
// chain is one memory latency long.  `vals` is unused (kept for the callers' scratch declarations).
// `ext` (LDS [64], forward only, may be null): also hands the caller {mean [32], invstd [32]} of the layer (a forward kernel
// that needs xhat itself: convt7's forward forms its own weight gradient, conv_thin_kernels.h FOLD).
__device__ __forceinline__ void bn_coef_from_acc(float* coef, double* vals, const BnFin& f, int t0 = 0, float* ext = nullptr) {
  (void)vals;
  const int t = (int)threadIdx.x - t0;
  if (t >= 0 && t < 64) {
    const int c = t & 31;
    const bool live = c < f.C, lo = t < 32;
    const bool pub = blockIdx.x == 0;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, r0 = 0.f, r1 = 0.f;
    if (live && lo) {
      p0 = f.gamma[c];
      if (!f.backward) {
        p1 = f.beta[c];
        if (pub && f.running_mean != nullptr) { r0 = f.running_mean[c]; r1 = f.running_var[c]; }
      } else { p1 = f.invstd[c]; p2 = f.mean[c]; }
    }
    const double mine = live ? bn_acc_read(f.acc, t) : 0.0;
    const double upper = __shfl(mine, c + 32, 64);          // value 32 + c
    if (lo) {
      float k0 = 0.f, k1 = 0.f, k2 = 0.f;
      if (live) {
        if (!f.backward) {
          const double mean = mine / f.n;
          double var = upper / f.n - mean * mean;          // biased variance
          if (var < 0.0) var = 0.0;
          const float meanf = (float)mean;
          const float invstd = (float)(1.0 / sqrt(var + AVA_BN_EPS_D));
          const float sc = p0 * invstd;
          k0 = sc;
          k1 = p1 - meanf * sc;
          if (ext != nullptr) { ext[c] = meanf; ext[32 + c] = invstd; }
          if (pub) {
            f.save[c] = meanf; f.save[32 + c] = invstd; f.save[64 + c] = k0; f.save[96 + c] = k1;
            if (f.running_mean != nullptr) {
              const double unb = f.n > 1.0 ? var * (f.n / (f.n - 1.0)) : var;
              f.running_mean[c] = (float)((1.0 - AVA_BN_MOM_D) * (double)r0 + AVA_BN_MOM_D * mean);
              f.running_var[c] = (float)((1.0 - AVA_BN_MOM_D) * (double)r1 + AVA_BN_MOM_D * unb);
            }
          }
        } else {
          const double dB = mine, dG = upper;
          const double is = (double)p1, gm = (double)p0, mu = (double)p2;
          const double a = gm * is;
          const double b = f.eval ? 0.0 : -gm * is * is * dG / f.n;
          k0 = (float)a;
          k1 = (float)b;
          k2 = f.eval ? 0.f : (float)(-a * dB / f.n - b * mu);
          if (pub) {
            f.dgamma[c] = (float)dG; f.dbeta[c] = (float)dB;
            f.abc[c] = k0; f.abc[32 + c] = k1; f.abc[64 + c] = k2;
          }
        }
      }
      coef[c] = k0; coef[32 + c] = k1; coef[64 + c] = k2;
      if (ext != nullptr && !live) { ext[c] = 0.f; ext[32 + c] = 0.f; }
      if (c == 0 && pub && !f.backward && f.num_batches != nullptr) *f.num_batches += 1;
    }
  }
  __syncthreads();
}
#endif
