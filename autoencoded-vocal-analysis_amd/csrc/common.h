// Shared device helpers for the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ava_hip.h"

#define AVA_WAVE 64

#define AVA_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t _e = hipGetLastError();                       \
    if (_e != hipSuccess) return AVA_ELAUNCH;                \
  } while (0)

static inline hipStream_t to_stream(ava_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

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
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
