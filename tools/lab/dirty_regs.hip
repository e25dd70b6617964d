// A kernel that leaves NaN patterns in every vector register and in all the LDS it can get, launched over and over on a side
// stream while a training step runs on the main one: a wave of the step that starts where such a wave ended finds NaNs wherever
// it reads a register or an LDS word it has not written itself.  (The counterpart of tools/lab/poison_ws.py for on-chip state;
// profiles/NOTES.md item 43.)   hipcc --offload-arch=gfx950 -O2 -shared -fPIC dirty_regs.hip -o libdirty.so
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(256) void dirty_kernel(float* sink, int lds_words) {
  extern __shared__ float s[];
  for (int i = threadIdx.x; i < lds_words; i += 256) s[i] = __int_as_float(0x7fc0dead);
  __syncthreads();
#define D8(b) "v_mov_b32 v" #b ", 0x7fc0beef\n"
  asm volatile(
      "s_nop 0\n"
      "v_mov_b32 v255, 0x7fc0beef\n v_mov_b32 v254, 0x7fc0beef\n v_mov_b32 v253, 0x7fc0beef\n v_mov_b32 v252, 0x7fc0beef\n"
      "v_mov_b32 v251, 0x7fc0beef\n v_mov_b32 v250, 0x7fc0beef\n v_mov_b32 v249, 0x7fc0beef\n v_mov_b32 v248, 0x7fc0beef\n"
      ::: "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "memory");
  // every other register: a loop over v_movreld would need m0 games; 31 more explicit blocks do
#define BLK(a, b, c, d, e, f, g, h)                                                                                              \
  asm volatile("v_mov_b32 v" #a ", 0x7fc0beef\n v_mov_b32 v" #b ", 0x7fc0beef\n v_mov_b32 v" #c ", 0x7fc0beef\n v_mov_b32 v" #d  \
               ", 0x7fc0beef\n v_mov_b32 v" #e ", 0x7fc0beef\n v_mov_b32 v" #f ", 0x7fc0beef\n v_mov_b32 v" #g                   \
               ", 0x7fc0beef\n v_mov_b32 v" #h ", 0x7fc0beef\n" ::: "v" #a, "v" #b, "v" #c, "v" #d, "v" #e, "v" #f, "v" #g, "v" #h)
  BLK(240, 241, 242, 243, 244, 245, 246, 247); BLK(232, 233, 234, 235, 236, 237, 238, 239);
  BLK(224, 225, 226, 227, 228, 229, 230, 231); BLK(216, 217, 218, 219, 220, 221, 222, 223);
  BLK(208, 209, 210, 211, 212, 213, 214, 215); BLK(200, 201, 202, 203, 204, 205, 206, 207);
  BLK(192, 193, 194, 195, 196, 197, 198, 199); BLK(184, 185, 186, 187, 188, 189, 190, 191);
  BLK(176, 177, 178, 179, 180, 181, 182, 183); BLK(168, 169, 170, 171, 172, 173, 174, 175);
  BLK(160, 161, 162, 163, 164, 165, 166, 167); BLK(152, 153, 154, 155, 156, 157, 158, 159);
  BLK(144, 145, 146, 147, 148, 149, 150, 151); BLK(136, 137, 138, 139, 140, 141, 142, 143);
  BLK(128, 129, 130, 131, 132, 133, 134, 135); BLK(120, 121, 122, 123, 124, 125, 126, 127);
  BLK(112, 113, 114, 115, 116, 117, 118, 119); BLK(104, 105, 106, 107, 108, 109, 110, 111);
  BLK(96, 97, 98, 99, 100, 101, 102, 103); BLK(88, 89, 90, 91, 92, 93, 94, 95);
  BLK(80, 81, 82, 83, 84, 85, 86, 87); BLK(72, 73, 74, 75, 76, 77, 78, 79);
  BLK(64, 65, 66, 67, 68, 69, 70, 71); BLK(56, 57, 58, 59, 60, 61, 62, 63);
  BLK(48, 49, 50, 51, 52, 53, 54, 55); BLK(40, 41, 42, 43, 44, 45, 46, 47);
  BLK(32, 33, 34, 35, 36, 37, 38, 39); BLK(24, 25, 26, 27, 28, 29, 30, 31);
  if (sink != nullptr && threadIdx.x == 9999) sink[0] = s[0];
}
extern "C" int dirty_launch(int grid, int lds_bytes, int times, void* stream) {
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dirty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  for (int i = 0; i < times; ++i)
    hipLaunchKernelGGL(dirty_kernel, dim3(grid), dim3(256), (size_t)lds_bytes, reinterpret_cast<hipStream_t>(stream), nullptr, lds_bytes / 4);
  return (int)hipGetLastError();
}
