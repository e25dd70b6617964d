// Does the 256 MB Infinity Cache serve a tensor that a kernel wrote a moment ago, and does the ORDER in which the consumer
// walks it matter (LRU: a consumer that walks in the producer's order meets the OLDEST lines first)?
//   hipcc --offload-arch=gfx950 -O3 tools/lab/mall_probe.hip -o tools/lab/mall_probe && tools/lab/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// chunk c (16 KB) is handled by workgroup (rev ? nchunks - 1 - c : c) mod grid, in increasing c per workgroup
__global__ __launch_bounds__(256) void wr(float4* p, long nchunks, int rev, float v) {
  for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const long cc = rev ? nchunks - 1 - c : c;
    float4* q = p + cc * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[threadIdx.x + 256 * i] = make_float4(v, v, v, v);
  }
}
__global__ __launch_bounds__(256) void rd(const float4* p, long nchunks, int rev, float* out) {
  float s = 0.f;
  for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const long cc = rev ? nchunks - 1 - c : c;
    const float4* q = p + cc * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float4 v = q[threadIdx.x + 256 * i]; s += v.x + v.y + v.z + v.w; }
  }
  if (s == 123.456f) out[0] = s;
}
int main() {
  const int grid = 2048;
  float* out; CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  float4* junk; const long junk_mb = 600; CK(hipMalloc(&junk, junk_mb << 20));
  for (long mb : {32L, 64L, 134L, 200L, 268L, 400L}) {
    const long nch = (mb << 20) / 16384;
    float4* buf; CK(hipMalloc(&buf, nch * 16384));
    hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, buf, nch, 0, 1.f);
    float res[3][3] = {};
    for (int mode = 0; mode < 3; ++mode) {          // 0: cold read (600 MB of other traffic in between), 1: write then read ascending, 2: write asc then read descending
      float tw = 0, tr = 0;
      const int reps = 8;
      for (int r = 0; r < reps + 2; ++r) {
        if (mode == 0) hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, junk, (junk_mb << 20) / 16384, 0, 2.f);
        CK(hipEventRecord(e0));
        if (mode != 0) hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, buf, nch, 0, 1.f);
        CK(hipEventRecord(e1));
        hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, buf, nch, mode == 2 ? 1 : 0, out);
        CK(hipEventRecord(e2));
        CK(hipEventSynchronize(e2));
        float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
        if (r >= 2) { tw += a; tr += b; }
      }
      res[mode][0] = tw / reps * 1e3f; res[mode][1] = tr / reps * 1e3f;
    }
    const double gb = (double)nch * 16384 / 1e9;
    printf("%4ld MB: write %6.1f us %5.0f GB/s | read cold %6.1f us %5.0f GB/s | read after write, same order %6.1f us %5.0f GB/s | reversed order %6.1f us %5.0f GB/s\n",
           mb, res[1][0], gb / res[1][0] * 1e6, res[0][1], gb / res[0][1] * 1e6, res[1][1], gb / res[1][1] * 1e6, res[2][1], gb / res[2][1] * 1e6);
    CK(hipFree(buf));
  }
  return 0;
}
