// Floor of a dependent kernel launch on this GPU as a function of grid, block and dynamic LDS: empty kernels back to back on
// one stream, and the same chain replayed as a HIP graph.   hipcc --offload-arch=gfx950 -O2 launch_probe.hip -o launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern "C" __global__ void k_empty(float* p) { if (p != nullptr && threadIdx.x == 9999) p[0] = 1.f; }
extern "C" __global__ void k_touch(float* p, int n) {
  extern __shared__ float s[];
  s[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x < n) p[blockIdx.x] = s[1];
}
static float run(void (*launch)(hipStream_t), hipStream_t st, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch(st);
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) launch(st);
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / iters;
}
static float* g_p; static int g_grid, g_block, g_lds, g_kind;
static void launch_one(hipStream_t st) {
  if (g_kind == 0) hipLaunchKernelGGL(k_empty, dim3(g_grid), dim3(g_block), g_lds, st, g_p);
  else hipLaunchKernelGGL(k_touch, dim3(g_grid), dim3(g_block), g_lds, st, g_p, g_grid);
}
int main() {
  hipStream_t st; (void)hipStreamCreate(&st);
  (void)hipMalloc(&g_p, 1 << 20);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_empty), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_touch), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int grids[] = {64, 256, 512, 1024, 4096}, blocks[] = {256, 512}, ldss[] = {0, 64 * 1024, 112 * 1024, 150 * 1024};
  for (int kind = 0; kind < 2; ++kind)
    for (int b : blocks) for (int l : ldss) for (int g : grids) {
      g_kind = kind; g_grid = g; g_block = b; g_lds = l;
      const float us = run(launch_one, st, 400);
      // the same launch as a graph of 20 dependent nodes
      hipGraph_t graph; hipGraphExec_t exec;
      (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
      for (int i = 0; i < 20; ++i) launch_one(st);
      (void)hipStreamEndCapture(st, &graph);
      (void)hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(exec, st);
      (void)hipEventRecord(e0, st);
      for (int i = 0; i < 20; ++i) (void)hipGraphLaunch(exec, st);
      (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
      float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("%s block %3d lds %6d grid %4d : stream %6.2f us/launch   graph %6.2f us/node\n", kind ? "touch" : "empty", b, l, g, us, ms * 1000.f / 400);
      (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph);
    }
  return 0;
}
