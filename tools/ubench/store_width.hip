// store_width.hip -- what does a wave's store of 64 consecutive elements cost by element width?  (owner tiles' flush: 2-byte stores)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/store_width.hip -o /tmp/store_width && /tmp/store_width
// Every wave stores ITER times 64 consecutive elements of T (lane = element) at column-strided addresses (like a frame column per
// band column); 1024 blocks x 256 threads.  Reported: ns per wave-level store instruction per CU, and GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

template <typename T>
__global__ __launch_bounds__(256) void k_store(T* base, int iters, size_t col_stride_elems, int rows_per_wave) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  T* p = base + (size_t)wave * rows_per_wave + lane;
  T v = (T)lane;
  for (int i = 0; i < iters; ++i) {
    p[(size_t)i * col_stride_elems] = v;
    v = (T)(v + 1);
  }
}
struct u128 { unsigned x, y, z, w; __device__ u128(int a = 0) : x(a), y(a), z(a), w(a) {} __device__ u128 operator+(int b) const { u128 r; r.x = x + b; r.y = y; r.z = z; r.w = w; return r; } };

template <typename T>
int run(const char* name) {
  const int blocks = 4096, iters = 64;
  const size_t waves = (size_t)blocks * 4, rows = 64;
  const size_t col_stride = waves * rows;  // elements: one "column" = every wave's 64 rows
  T* d;
  CK(hipMalloc(&d, col_stride * iters * sizeof(T)));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_store<T>, dim3(blocks), dim3(256), 0, 0, d, iters, col_stride, (int)rows);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep == 2) {
      const double instr = (double)waves * iters;
      printf("%-6s %8.1f us   %6.2f ns per wave store per CU (256 CUs)   %7.1f GB/s\n", name, ms * 1e3, ms * 1e6 / instr * 256.0,
             instr * 64 * sizeof(T) / ms / 1e6);
    }
  }
  CK(hipFree(d));
  return 0;
}
int main() {
  if (run<uint16_t>("u16")) return 1;
  if (run<uint32_t>("u32")) return 1;
  if (run<uint64_t>("u64")) return 1;
  if (run<u128>("u128")) return 1;
  return 0;
}
