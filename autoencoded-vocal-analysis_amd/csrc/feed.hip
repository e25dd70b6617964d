// Input feeding: the device-side cast of raw loader bytes and the native collation of
// a batch into the page-locked ring (host threads, no Python in the copy loop).
#include <hip/hip_fp16.h>
#include <string.h>
#include <thread>
#include <vector>
#include "common.h"

// ---- input feeding: raw loader bytes -> fp32 spectrograms ON THE DEVICE ----------------
// Replaces the per-item CPU conversion `torch.from_numpy(x).type(torch.FloatTensor)` of the reference's loader
// (ava/models/utils.py:444-446 applied in ava/models/vae_dataset.py:138-139): the batch crosses PCIe in the dtype it
// is stored in (float64 spectrograms as written by the preprocessing step: 2x the bytes but no 4M-element CPU pass;
// uint8: a quarter of the bytes) and is converted here with the same rounding as torch's cast (f64 -> f32
// round-to-nearest-even; u8 / f16 / bf16 -> f32 exact).  16 bytes per lane on the read side.
enum { AVA_DT_F32 = 0, AVA_DT_F64 = 1, AVA_DT_U8 = 2, AVA_DT_F16 = 3, AVA_DT_BF16 = 4 };

template <int DT>
__global__ __launch_bounds__(256) void cast_to_f32_kernel(const void* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  if (DT == AVA_DT_F64) {                       // 2 doubles (16 B) per lane and iteration
    const double2* s2 = reinterpret_cast<const double2*>(src);
    float2* d2 = reinterpret_cast<float2*>(dst);
    const int64_t n2 = n / 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) {
      const double2 v = s2[i];
      d2[i] = make_float2((float)v.x, (float)v.y);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) dst[n - 1] = (float)reinterpret_cast<const double*>(src)[n - 1];
  } else if (DT == AVA_DT_U8) {                 // 16 bytes per lane -> four float4 stores
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    const int64_t n16 = n / 16;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
      const uint4 v = s4[i];
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        d4[4 * i + q] = make_float4((float)(w[q] & 0xffu), (float)((w[q] >> 8) & 0xffu), (float)((w[q] >> 16) & 0xffu),
                                    (float)(w[q] >> 24));
    }
    if (blockIdx.x == 0)
      for (int64_t i = n16 * 16 + threadIdx.x; i < n; i += 256) dst[i] = (float)reinterpret_cast<const unsigned char*>(src)[i];
  } else if (DT == AVA_DT_F16 || DT == AVA_DT_BF16) {     // 8 halves (16 B) per lane
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    const int64_t n8 = n / 8;
    auto cvt = [](unsigned short h) -> float {
      if (DT == AVA_DT_BF16) return __uint_as_float((unsigned)h << 16);
      return __half2float(__ushort_as_half(h));
    };
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
      const uint4 v = s4[i];
      d4[2 * i] = make_float4(cvt(v.x & 0xffff), cvt(v.x >> 16), cvt(v.y & 0xffff), cvt(v.y >> 16));
      d4[2 * i + 1] = make_float4(cvt(v.z & 0xffff), cvt(v.z >> 16), cvt(v.w & 0xffff), cvt(v.w >> 16));
    }
    if (blockIdx.x == 0)
      for (int64_t i = n8 * 8 + threadIdx.x; i < n; i += 256) dst[i] = cvt(reinterpret_cast<const unsigned short*>(src)[i]);
  } else {                                      // f32: plain copy (kept so that one call covers every loader dtype)
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) d4[i] = s4[i];
    if (blockIdx.x == 0)
      for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) dst[i] = reinterpret_cast<const float*>(src)[i];
  }
}

extern "C" int ava_cast_to_f32(const void* src, int src_dtype, int64_t n, float* dst, ava_stream_t s) {
  if (src == nullptr || dst == nullptr || n <= 0) return AVA_EINVAL;
  if ((reinterpret_cast<uintptr_t>(src) & 15) != 0 || (reinterpret_cast<uintptr_t>(dst) & 15) != 0) return AVA_EINVAL;
  int64_t work = src_dtype == AVA_DT_F64 ? n / 2 : (src_dtype == AVA_DT_U8 ? n / 16 : (src_dtype == AVA_DT_F32 ? n / 4 : n / 8));
  int blocks = (int)((work + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = to_stream(s);
  switch (src_dtype) {
    case AVA_DT_F32: hipLaunchKernelGGL(cast_to_f32_kernel<AVA_DT_F32>, dim3(blocks), dim3(256), 0, st, src, dst, n); break;
    case AVA_DT_F64: hipLaunchKernelGGL(cast_to_f32_kernel<AVA_DT_F64>, dim3(blocks), dim3(256), 0, st, src, dst, n); break;
    case AVA_DT_U8: hipLaunchKernelGGL(cast_to_f32_kernel<AVA_DT_U8>, dim3(blocks), dim3(256), 0, st, src, dst, n); break;
    case AVA_DT_F16: hipLaunchKernelGGL(cast_to_f32_kernel<AVA_DT_F16>, dim3(blocks), dim3(256), 0, st, src, dst, n); break;
    case AVA_DT_BF16: hipLaunchKernelGGL(cast_to_f32_kernel<AVA_DT_BF16>, dim3(blocks), dim3(256), 0, st, src, dst, n); break;
    default: return AVA_EINVAL;
  }
  AVA_CHECK_LAUNCH();
  return AVA_OK;
}

// ---- host side: gather n rows of row_bytes each, src[idx[i]] -> dst[i] (dst: a slot of the page-locked ring) --------
// Replaces default_collate's per-item torch.stack of the reference's DataLoader (ava/models/vae_dataset.py:89-96) for
// array-backed datasets.  Runs on `threads` std::threads (the copy is memory-bound: one core moves ~5 GB/s, a 16 MiB
// batch needs ~4 to keep up with a 2 ms step); ctypes releases the GIL for the duration of the call, so the Python
// thread that launches the step's kernels is not disturbed.  idx == NULL: rows first .. first + n - 1.
extern "C" int ava_host_gather_rows(void* dst, const void* src, const int64_t* idx, int64_t first, int64_t n,
                                    size_t row_bytes, int threads) {
  if (dst == nullptr || src == nullptr || n < 0 || row_bytes == 0) return AVA_EINVAL;
  if (n == 0) return AVA_OK;
  if (threads < 1) threads = 1;
  if (threads > 16) threads = 16;
  {  // at least 2 MiB per thread: starting a thread costs more than copying less (uint8 batches: 8 threads 2.1 ms, 4 threads 0.14 ms)
    const size_t per = (size_t)2 << 20, total = (size_t)n * row_bytes;
    const int cap = (int)(total / per);
    if (threads > cap) threads = cap < 1 ? 1 : cap;
  }
  if ((int64_t)threads > n) threads = (int)n;
  auto work = [=](int64_t lo, int64_t hi) {
    char* d = static_cast<char*>(dst);
    const char* s = static_cast<const char*>(src);
    if (idx == nullptr) {
      memcpy(d + (size_t)lo * row_bytes, s + (size_t)(first + lo) * row_bytes, (size_t)(hi - lo) * row_bytes);
    } else {
      for (int64_t i = lo; i < hi; ++i) memcpy(d + (size_t)i * row_bytes, s + (size_t)idx[i] * row_bytes, row_bytes);
    }
  };
  if (threads == 1) { work(0, n); return AVA_OK; }
  std::vector<std::thread> pool;
  const int64_t step = (n + threads - 1) / threads;
  for (int t = 1; t < threads; ++t) {
    const int64_t lo = t * step, hi = lo + step < n ? lo + step : n;
    if (lo < hi) pool.emplace_back(work, lo, hi);
  }
  work(0, step < n ? step : n);
  for (auto& th : pool) th.join();
  return AVA_OK;
}
