// dma_mix: the ingest's DMA pattern in isolation -- small H2D copies (a packet: 720 KB) on one stream while 6.2 MB D2H copies (a
// result frame) run on another, whole or in pieces: how long does an H2D copy take alone, and behind / beside a D2H copy?
//   hipcc -O2 tools/ubench/dma_mix.cpp -o tools/ubench/dma_mix && tools/ubench/dma_mix
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const size_t pkt = 720u << 10, frame = 6220800, N = 64;
  void *dpk, *dfr, *hpk, *hfr;
  CK(hipSetDevice(0));
  CK(hipMalloc(&dpk, pkt * 4));
  CK(hipMalloc(&dfr, frame));
  CK(hipHostMalloc(&hpk, pkt * 4 * N, hipHostMallocDefault));
  CK(hipHostMalloc(&hfr, frame * 4, hipHostMallocDefault));
  memset(hpk, 1, pkt * 4 * N);
  memset(hfr, 1, frame * 4);
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t a, a2, b;
  const int two = getenv("TWO_H2D_STREAMS") ? atoi(getenv("TWO_H2D_STREAMS")) : 0;  // packets alternate between two H2D streams
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&a2, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi));
  std::vector<hipEvent_t> e0(4 * N), e1(4 * N);
  for (auto& e : e0) CK(hipEventCreate(&e));
  for (auto& e : e1) CK(hipEventCreate(&e));
  const size_t pieces[] = {0, 16u << 20, 4u << 20};
  for (size_t piece : pieces) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      double host_block = 0;
      for (size_t i = 0; i < N; ++i) {
        for (int k = 0; k < 4; ++k) {
          hipStream_t sa = two && (k & 1) ? a2 : a;
          CK(hipEventRecord(e0[4 * i + k], sa));
          CK(hipMemcpyAsync((char*)dpk + k * pkt, (char*)hpk + (4 * i + k) * pkt, pkt, hipMemcpyHostToDevice, sa));
          CK(hipEventRecord(e1[4 * i + k], sa));
        }
        if (piece) {
          auto c0 = std::chrono::steady_clock::now();
          for (size_t off = 0; off < frame; off += piece)
            CK(hipMemcpyAsync((char*)hfr + (i % 4) * frame + off, (char*)dfr + off, std::min(piece, frame - off), hipMemcpyDeviceToHost, b));
          host_block += std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
        }
      }
      CK(hipDeviceSynchronize());
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      std::vector<float> us;
      for (size_t i = 0; i < 4 * N; ++i) {
        float ms;
        CK(hipEventElapsedTime(&ms, e0[i], e1[i]));
        us.push_back(ms * 1e3f);
      }
      std::sort(us.begin(), us.end());
      if (rep)
        printf("D2H piece %8zu: %5.2f ms for %zu x (4 packets in + 1 frame out) = %6.1f us per frame | an H2D copy: p50 %5.1f p90 %5.1f max %6.1f us | "
               "host time in the D2H enqueues %5.2f ms\n", piece, dt * 1e3, N, dt / N * 1e6, us[us.size() / 2], us[us.size() * 9 / 10], us.back(),
               host_block * 1e3);
    }
  }
  return 0;
}
