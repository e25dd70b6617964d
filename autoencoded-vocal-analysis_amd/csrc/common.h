// Shared device helpers for the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/ava_hip.h"

#define AVA_WAVE 64

#define AVA_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t _e = hipGetLastError();                       \
    if (_e != hipSuccess) return AVA_ELAUNCH;                \
  } while (0)

static inline hipStream_t to_stream(ava_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Tuning switches (AVA_* environment variables, ablation bits, alternative tile shapes and the superseded kernels they
// select) exist only in the lab build (`make lab` -> libava_hip_lab.so, -DAVA_LAB; tools/README.md).  The product
// library has ONE code path per shape: ava_env() is a constant there and everything behind it folds away.
#ifdef AVA_LAB
static inline const char* ava_env(const char* name) { return getenv(name); }
#define AVA_DBG_BIT(args, bit) (((args).dbg & (bit)) != 0)
#else
static inline const char* ava_env(const char*) { return nullptr; }
#define AVA_DBG_BIT(args, bit) false
#endif

// sum over the 64 lanes of a wave; result valid in every lane
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// CUs every persistent ("one resident wave of workgroups") launch leaves free: ava_set_cu_reserve (model.hip).  Under data
// parallelism RCCL's persistent workgroups hold wave slots for the length of an all-reduce; a conv grid sized for the
// whole chip would then find fewer free slots than workgroups, and the surplus ones -- each owning a STATIC share of the
// tiles -- would run as a second wave (up to 2x the kernel's time).  A grid sized for (CUs - reserve) fits beside them.
// The tile partition is a function of the grid size only, so results are bit-reproducible for a given setting.
int ava_cu_reserve(void);
static inline int ava_scale_grid(int grid_for_all_cus) {
  const int r = ava_cu_reserve();
  if (r <= 0) return grid_for_all_cus;
  const long g = (long)grid_for_all_cus * (256 - r) / 256;
  return g < 1 ? 1 : (int)g;
}
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Counter-based standard normal: splitmix64 finaliser + Box-Muller; matches ava_amd.synthetic.u01 / gauss (SURVEY
// Appendix E) so that injected and device-generated noise agree to float rounding when seeded alike.  Element i of a
// stream depends on (i + offset, seed) only, never on the launch geometry.
#ifdef __HIPCC__
__device__ __forceinline__ double ava_u01_hash(uint64_t i, uint64_t salt) {
  uint64_t x = i + salt * 0x9E3779B97F4A7C15ull;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ float ava_normal_hash(uint64_t i, uint64_t seed) {
  const double u1 = ava_u01_hash(i, seed), u2 = ava_u01_hash(i, seed + 7777);
  return (float)(sqrt(-2.0 * log(1.0 - u1)) * cos(6.283185307179586 * u2));
}
#endif
