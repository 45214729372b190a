// Experiment: workgroup / wave dispatch rate of one MI355X.  Kernels whose waves do (almost) nothing, or live for a fixed number
// of cycles: duration / workgroups as a function of the block size, the dynamic LDS size and the wave lifetime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
extern __shared__ unsigned dyn_lds[];
__global__ void k_live(int cycles, unsigned* sink) {
  if (cycles > 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < cycles) __builtin_amdgcn_s_sleep(2);
  }
  if (sink && threadIdx.x == 12345) *sink = dyn_lds[0];
}
template <int NV>
__global__ void k_regs(int cycles, unsigned* sink, const unsigned* src) {  // NV live VGPRs
  unsigned v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = src ? src[threadIdx.x + i * 64] : i;
  if (cycles > 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)(__builtin_amdgcn_s_memtime() - t0) < cycles) __builtin_amdgcn_s_sleep(2);
  }
  unsigned a = 0;
#pragma unroll
  for (int i = 0; i < NV; ++i) a ^= v[i];
  if (sink && a == 0x12345) *sink = a;
}
static float time_launch(void (*k)(int, unsigned*), int grid, int threads, size_t lds, int cycles, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, cycles, (unsigned*)nullptr);
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, cycles, (unsigned*)nullptr);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best;
  }
  return best * 1e3f;
}
int main() {
  CK(hipFuncSetAttribute((const void*)k_live, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int grids[] = {8192, 65536};
  printf("threads lds_KB cycles grid   us     ns/WG  ns/wave\n");
  for (int cycles : {0, 1500, 6000})
    for (int threads : {64, 256, 512, 1024})
      for (size_t lds : {(size_t)0, (size_t)4096, (size_t)16384})
        for (int g : grids) {
          const float us = time_launch(k_live, g, threads, lds, cycles, 8);
          printf("%6d %5zu %6d %6d %8.1f %6.2f %6.2f\n", threads, lds / 1024, cycles, g, us, us * 1e3 / g, us * 1e3 / g / (threads / 64));
        }
  return 0;
}
