// pinned_bw: H2D / D2H DMA rate of several pinned host buffers allocated one after the other (hipHostMalloc, and malloc + madvise
// (MADV_HUGEPAGE) + hipHostRegister), with where their pages live (/proc/self/numa_maps).  Does the rate depend on the buffer?
//   hipcc -O2 tools/ubench/pinned_bw.cpp -o /tmp/pinned_bw && /tmp/pinned_bw
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static std::string where(void* p) {
  FILE* f = fopen("/proc/self/numa_maps", "r");
  if (!f) return "?";
  char line[4096];
  char key[32];
  snprintf(key, sizeof key, "%lx ", (unsigned long)p);
  std::string out = "(no numa_maps entry at this address)";
  while (fgets(line, sizeof line, f))
    if (!strncmp(line, key, strlen(key))) { out = line; if (!out.empty() && out.back() == '\n') out.pop_back(); break; }
  fclose(f);
  return out.substr(0, 160);
}

int main(int argc, char** argv) {
  const size_t bytes = 128u << 20;
  const int nbuf = 4;
  void* dev;
  CK(hipSetDevice(0));
  CK(hipMalloc(&dev, bytes));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<std::pair<std::string, void*>> bufs;
  for (int i = 0; i < nbuf; ++i) {
    void* p;
    CK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    memset(p, 1, bytes);
    bufs.push_back({"hipHostMalloc #" + std::to_string(i), p});
    void* junk;  // (other allocations in between, as an application has them)
    CK(hipHostMalloc(&junk, 48u << 20, hipHostMallocDefault));
    memset(junk, 2, 48u << 20);
  }
  for (int i = 0; i < 2; ++i) {
    void* p = aligned_alloc(2u << 20, bytes);
    madvise(p, bytes, MADV_HUGEPAGE);
    memset(p, 3, bytes);
    CK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    bufs.push_back({"aligned 2 MB + MADV_HUGEPAGE + hipHostRegister #" + std::to_string(i), p});
  }
  for (auto& b : bufs) {
    double best[2] = {0, 0};
    for (int dir = 0; dir < 2; ++dir)
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (size_t off = 0; off < bytes; off += 8u << 20)
          CK(dir == 0 ? hipMemcpyAsync((char*)dev + off, (char*)b.second + off, 8u << 20, hipMemcpyHostToDevice, s)
                      : hipMemcpyAsync((char*)b.second + off, (char*)dev + off, 8u << 20, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rep) best[dir] = std::max(best[dir], bytes / dt / 1e9);
      }
    printf("%-52s H2D %5.1f GB/s  D2H %5.1f GB/s   %s\n", b.first.c_str(), best[0], best[1], where(b.second).c_str());
  }
  FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
  char line[256] = "?";
  if (f) { if (!fgets(line, sizeof line, f)) line[0] = 0; fclose(f); }
  printf("transparent_hugepage/enabled: %s", line);
  return 0;
}
