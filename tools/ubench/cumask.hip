// Experiment: which physical CUs does a hipExtStreamCreateWithCUMask bit select on MI355X (8 XCDs x 32 CUs)?
// Each block records XCC_ID and HW_ID; the host prints, per mask, how many blocks ran on each XCD and on how many distinct CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <vector>
__global__ void k_where(unsigned* out, int spin) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main() {
  const int nb = 4096;
  unsigned* d; CK(hipMalloc(&d, nb * 8));
  std::vector<unsigned> h(2 * nb);
  auto run = [&](const char* name, std::vector<uint32_t> mask) {
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    CK(hipMemsetAsync(d, 0xff, nb * 8, s));
    hipLaunchKernelGGL(k_where, dim3(nb), dim3(64), 0, s, d, 200);  // 2 us per block
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> cus;
    for (int i = 0; i < nb; ++i) {
      unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
      unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
      cus[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    printf("%-28s", name);
    int tot = 0;
    for (auto& kv : cus) { printf(" xcc%u:%zu", kv.first, kv.second.size()); tot += kv.second.size(); }
    printf("  total CUs %d\n", tot);
    CK(hipStreamDestroy(s));
  };
  auto bits = [&](int lo, int hi) { std::vector<uint32_t> m(8, 0); for (int i = lo; i < hi; ++i) m[i / 32] |= 1u << (i % 32); return m; };
  run("all 256", bits(0, 256));
  run("bits 0..31", bits(0, 32));
  run("bits 0..63", bits(0, 64));
  run("bits 0..127", bits(0, 128));
  run("bits 128..255", bits(128, 256));
  { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; i += 2) m[i / 32] |= 1u << (i % 32); run("even bits", m); }
  { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; ++i) if ((i % 8) < 5) m[i / 32] |= 1u << (i % 32); run("i%8<5 (160)", m); }
  { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; ++i) if ((i / 8) % 8 < 5) m[i / 32] |= 1u << (i % 32); run("(i/8)%8<5 (160)", m); }
  { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; ++i) if ((i % 32) < 20) m[i / 32] |= 1u << (i % 32); run("i%32<20 (160)", m); }
  return 0;
}
