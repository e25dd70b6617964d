// What does it cost a persistent kernel to end with per-workgroup integer atomics into a small set of accumulators?
// (design probe for replacing the BatchNorm finalisation launches by in-kernel accumulation)
// Build: hipcc -O3 --offload-arch=gfx950 tools/lab/atomic_probe.hip -o tools/lab/atomic_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// body: stream `n4` float4 per workgroup (some real work), then `nval` atomics per workgroup (one per lane of wave 0..)
__global__ __launch_bounds__(256) void probe(const float4* __restrict__ in, float4* __restrict__ out, size_t n4_per_wg,
                                             unsigned long long* acc, int nval, int shards, int stride_ll, int limbs, int mode) {
  const size_t base = (size_t)blockIdx.x * n4_per_wg;
  float4 s = make_float4(0, 0, 0, 0);
  for (size_t i = threadIdx.x; i < n4_per_wg; i += 256) { float4 v = in[base + i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; out[base + i] = v; }
  if (mode == 0) return;
  const int t = threadIdx.x;
  if (t < nval) {
    unsigned long long* a = acc + (size_t)(blockIdx.x % shards) * stride_ll;
    const long long q = (long long)(s.x * 1048576.f) + t;
    for (int l = 0; l < limbs; ++l) {
      if (mode == 1) atomicAdd(a + (size_t)l * nval + t, (unsigned long long)(q + l));                 // dense: 16 values per 128-B line
      else atomicAdd(a + ((size_t)l * nval + t) * 16, (unsigned long long)(q + l));                     // one value per 128-B line
    }
  }
}

int main() {
  const int grid = 512;
  const size_t n4 = 8192;                       // 128 KB in + 128 KB out per workgroup: ~25 us kernel
  float4 *in, *out; unsigned long long* acc;
  CHECK(hipMalloc(&in, grid * n4 * 16)); CHECK(hipMalloc(&out, grid * n4 * 16));
  CHECK(hipMemset(in, 0, grid * n4 * 16));
  CHECK(hipMalloc(&acc, 64 << 20)); CHECK(hipMemset(acc, 0, 64 << 20));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto run = [&](int nval, int shards, int limbs, int mode) {
    const int stride = mode == 2 ? nval * limbs * 16 + 16 : ((nval * limbs + 15) / 16) * 16 + 16;
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, in, out, n4, acc, nval, shards, stride, limbs, mode);
    CHECK(hipEventRecord(e0));
    const int reps = 50;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, in, out, n4, acc, nval, shards, stride, limbs, mode);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3f * ms / reps;
  };
  printf("baseline (no atomics)                  : %.2f us\n", run(64, 1, 1, 0));
  for (int mode = 1; mode <= 2; ++mode)
    for (int shards : {1, 8, 64})
      for (int nval : {16, 64})
        for (int limbs : {1, 3})
          printf("mode %d (%s) shards %2d values %2d limbs %d : %.2f us\n", mode, mode == 1 ? "dense " : "padded", shards, nval, limbs, run(nval, shards, limbs, mode));
  printf("baseline again                         : %.2f us\n", run(64, 1, 1, 0));
  return 0;
}
