// Does a noinline device function see (a) the kernel's arguments through the kernarg segment pointer and (b) the kernel's dynamic
// LDS?  (the mechanism conv_fused_limb.hip's role functions would rely on)   hipcc --offload-arch=gfx950 -O3 noinline_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Args { const float* x; float* y; int n; int pad[20]; float scale; };
template <int ROLE>
__device__ __attribute__((noinline)) void role() {
  Args a;
  __builtin_memcpy(&a, (const void*)__builtin_amdgcn_kernarg_segment_ptr(), sizeof(Args));
  extern __shared__ __align__(16) unsigned char smem[];
  float* s = reinterpret_cast<float*>(smem);
  const int t = threadIdx.x;
  if (ROLE == 0) { if (t < 64) s[t] = a.x[t] * a.scale; }
  __syncthreads();
  if (ROLE == 1) { a.y[blockIdx.x * 64 + (t - 64)] = s[t - 64] + (float)a.n; }
  __syncthreads();
}
__global__ __launch_bounds__(128) void kern(const Args a) {
  const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  if (w == 0) role<0>(); else role<1>();
}
int main() {
  float *x, *y; hipMalloc(&x, 64 * 4); hipMalloc(&y, 4 * 64 * 4);
  float hx[64]; for (int i = 0; i < 64; ++i) hx[i] = i; hipMemcpy(x, hx, sizeof hx, hipMemcpyHostToDevice);
  Args a = {}; a.x = x; a.y = y; a.n = 7; a.scale = 2.f;
  hipLaunchKernelGGL(kern, dim3(4), dim3(128), 1024, 0, a);
  hipError_t e = hipDeviceSynchronize();
  float hy[256]; hipMemcpy(hy, y, sizeof hy, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; ++i) if (hy[i] != 2.f * (i % 64) + 7.f) ++bad;
  printf("sync %d bad %d  y[5]=%g (want 17)\n", (int)e, bad, hy[5]);
  return bad != 0;
}
