// Experiment: how many kernel launches per second does one MI355X accept, as a function of the number of streams and of
// the kernel's duration?  (The pipelined frame loop costs ~5.5 us per kernel launch whatever the kernel does.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void k_spin(int ticks, unsigned* sink) {  // ticks of the 100 MHz counter
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((int)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) {}
  if (sink && threadIdx.x == 12345) *sink = 1;
}
// a K0-like kernel: every thread reads 4 x 16 bytes, block-reduces, one atomic per block
__global__ __launch_bounds__(256) void k_read(const ulonglong2* __restrict__ src, size_t n16, unsigned long long* out) {
  const size_t stride = (size_t)gridDim.x * 256;
  unsigned long long acc = 0;
  ulonglong2 v[4];
  const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = src[(i0 + j * stride) % n16];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = acc > v[j].x ? acc : v[j].x, acc = acc > v[j].y ? acc : v[j].y;
  for (int o = 32; o > 0; o >>= 1) { unsigned long long w = __shfl_xor(acc, o, 64); acc = acc > w ? acc : w; }
  __shared__ unsigned long long s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) acc = acc > s[w] ? acc : s[w];
    __hip_atomic_fetch_max(out + (blockIdx.x & 31), acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double run(int n_streams, int blocks, int threads, int ticks, int n_launch, int n_threads) {
  std::vector<hipStream_t> st(n_streams * n_threads);
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto work = [&](int tix, int n) {
    for (int i = 0; i < n; ++i)
      hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(threads), 0, st[tix * n_streams + i % n_streams], ticks, (unsigned*)nullptr);
    for (int i = 0; i < n_streams; ++i) CK(hipStreamSynchronize(st[tix * n_streams + i]));
  };
  work(0, 200);
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t, n_launch);
  for (auto& t : th) t.join();
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  for (auto& s : st) CK(hipStreamDestroy(s));
  return us / (n_launch * n_threads);
}
static double run_read(int n_streams, int n_bufs, int n_launch, int n_threads) {
  const size_t n16 = 500000;  // 8 MB per buffer
  std::vector<ulonglong2*> bufs(n_bufs);
  for (auto& b : bufs) { CK(hipMalloc(&b, n16 * 16)); CK(hipMemset(b, 1, n16 * 16)); }
  unsigned long long* out; CK(hipMalloc(&out, 64 * 8 * 64)); CK(hipMemset(out, 0, 64 * 8 * 64));
  std::vector<hipStream_t> st(n_streams * n_threads);
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto work = [&](int tix, int n) {
    for (int i = 0; i < n; ++i)
      hipLaunchKernelGGL(k_read, dim3(489), dim3(256), 0, st[tix * n_streams + i % n_streams], bufs[(i + tix) % n_bufs], n16,
                         out + 64 * ((tix * n_streams + i % n_streams) % 64));
    for (int i = 0; i < n_streams; ++i) CK(hipStreamSynchronize(st[tix * n_streams + i]));
  };
  work(0, 200);
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t, n_launch);
  for (auto& t : th) t.join();
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  for (auto& s : st) CK(hipStreamDestroy(s));
  for (auto& b : bufs) CK(hipFree(b));
  CK(hipFree(out));
  return us / (n_launch * n_threads);
}
int main(int argc, char** argv) {
  if (argc > 1) {
    printf("K0-like 8 MB read kernel (489 x 256 threads), us per launch:\n");
    for (int nb : {1, 8, 32})
      for (int ns : {1, 2, 4, 8})
        printf("buffers %2d streams %d : 1 thread %6.2f   2 threads %6.2f   3 threads %6.2f\n", nb, ns, run_read(ns, nb, 20000, 1),
               run_read(ns, nb, 20000, 2), run_read(ns, nb, 20000, 3));
    return 0;
  }
  printf("us per launch (host thread x streams, kernel = blocks x threads spinning `dur` us)\n");
  for (int dur : {0, 3, 10})
    for (int blocks : {1, 256, 1200})
      for (int ns : {1, 4, 8}) {
        printf("dur %2d us blocks %4d streams %d : 1 thread %6.2f   2 threads %6.2f\n", dur, blocks, ns,
               run(ns, blocks, 256, dur * 100, 20000, 1), run(ns, blocks, 256, dur * 100, 20000, 2));
      }
  return 0;
}
