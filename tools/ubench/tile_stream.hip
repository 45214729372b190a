// tile_stream.hip -- the floor of the headline K1's ACCESS PATTERN: same grid (320 tiles x 32 frames of
// C-1M), same blocks (512 threads), same 16-byte event loads (8 consecutive events per lane: one uint4 of x, one of y, four of t),
// same plain 2-byte cell stores (2 columns x 1320 rows per tile, lanes = consecutive rows) -- and NOTHING else: no LUT / X-map
// bands, no LDS slots, no ds_max, no per-event arithmetic beyond what keeps the loads alive.  What it tells:
//   * the time this access pattern cannot beat on this chip, as a function of how many blocks a CU holds (K1 is LDS-bound at
//     three blocks of 48.6 KB per CU: `lds` bytes of dynamic LDS are allocated and not used to set the same occupancy);
//   * what ONE dependent round trip in front of the event loads costs (`bnd`: the tile's event range comes from a 16-byte
//     boundary record, as k_cols_bounds leaves it for K1) -- the link the speculative variant of (ii) would cut;
//   * the cost of the stores alone / the loads alone.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/tile_stream.hip -o tools/ubench/tile_stream && tools/ubench/tile_stream
// 24 B/event x 32 M events = 768 MB "algorithmic" per launch; the bytes that really move: 12 B/event read + the
// frame's 2 x 1320 x 2 B per tile written = 384 + 54 MB.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

constexpr int THREADS = 512, EPT = 8, TILES = 320, FRAMES = 32, ROWS = 1320, W = 2;
constexpr int N = 1000000;  // events per frame
constexpr int GROUPS = 4;   // distinct groups of 32 frames (1.5 GB of events)

// K1's bands on top of the pattern: per tile 30.7 KB of rectify LUT (16 camera columns around the tile's scan position) and 5.3 KB
// of X-map (its two time columns) from tables that live in L2, straight into LDS (global_load_lds_dwordx4), the LUT band behind
// the event loads as in K1; one s_waitcnt vmcnt(0) + barrier before the stores
constexpr int CAM_H = 480, CAM_W = 640, WX = 16;
template <bool BANDS>
__global__ __launch_bounds__(THREADS) void k_tile_bands(const uint16_t* __restrict__ xs, const uint16_t* __restrict__ ys, const long long* __restrict__ ts,
                                                        const unsigned* __restrict__ lut, const uint16_t* __restrict__ xmap, uint16_t* __restrict__ frame16) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int tile = blockIdx.x, f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, q0w = tid & ~63;
  const size_t fo = (size_t)f * N;
  const int lb_s = (int)((long long)tile * N / TILES), lb_e = (int)((long long)(tile + 1) * N / TILES);
  uint4* l_xm = reinterpret_cast<uint4*>(smem);                      // 5.3 KB (+ slack)
  uint4* l_lut = reinterpret_cast<uint4*>(smem + 8 * 1024);          // 30.7 KB
  const int nq_xm = W * ROWS * 2 / 16, nq_lut = WX * CAM_H * 4 / 16;
  if (BANDS) {
    const uint4* g_xm = reinterpret_cast<const uint4*>(xmap + (size_t)tile * W * ROWS);
    for (int q0 = q0w; q0 < nq_xm; q0 += THREADS)
      __builtin_amdgcn_global_load_lds((glb_void*)(g_xm + min(q0 + lane, nq_xm - 1)), (lds_void*)(l_xm + q0), 16, 0, 0);
  }
  const int a0 = lb_s & ~(EPT - 1);
  const int base = min(a0 + tid * EPT, (N - 1) & ~(EPT - 1));
  uint4 xv = make_uint4(0, 0, 0, 0), yv = xv;
  longlong2 tv[4] = {};
  if (a0 + tid * EPT < lb_e) {
    xv = *(const uint4*)(xs + fo + base);
    yv = *(const uint4*)(ys + fo + base);
#pragma unroll
    for (int q = 0; q < 4; ++q) tv[q] = *(const longlong2*)(ts + fo + base + 2 * q);
    asm volatile("" : "+v"(xv.x), "+v"(xv.y), "+v"(xv.z), "+v"(xv.w), "+v"(yv.x), "+v"(yv.y), "+v"(yv.z), "+v"(yv.w));
#pragma unroll
    for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(tv[q].x), "+v"(tv[q].y));
  }
  if (BANDS) {
    const int x_lo = min(max(tile * W - WX / 2, 0), CAM_W - WX);
    const uint4* g_lut = reinterpret_cast<const uint4*>(lut + (size_t)x_lo * CAM_H);
    for (int q0 = q0w; q0 < nq_lut; q0 += THREADS)
      __builtin_amdgcn_global_load_lds((glb_void*)(g_lut + min(q0 + lane, nq_lut - 1)), (lds_void*)(l_lut + q0), 16, 0, 0);
  }
  unsigned acc = xv.x ^ xv.y ^ xv.z ^ xv.w ^ yv.x ^ yv.y ^ yv.z ^ yv.w;
#pragma unroll
  for (int q = 0; q < 4; ++q) acc ^= (unsigned)tv[q].x ^ (unsigned)tv[q].y;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (BANDS) acc ^= reinterpret_cast<const unsigned*>(smem)[(tid * 37) & 8191];
  uint16_t* col = frame16 + ((size_t)f * TILES * W + (size_t)tile * W) * ROWS;
  for (int i = tid; i < W * ROWS; i += THREADS) col[i] = (uint16_t)(acc + i);
}

template <bool BANDS>
int run_bands(const char* name, const uint16_t* xs, const uint16_t* ys, const long long* ts, const unsigned* lut, const uint16_t* xmap, uint16_t* frame16) {
  auto kern = k_tile_bands<BANDS>;
  const size_t lds = 49 * 1024;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float sum = 0, best = 1e9f;
  const int reps = 12;
  for (int rep = 0; rep < reps + 2; ++rep) {
    const size_t g = (size_t)(rep % GROUPS) * N * FRAMES;
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(TILES, FRAMES), dim3(THREADS), lds, 0, xs + g, ys + g, ts + g, lut, xmap, frame16);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep >= 2) {
      sum += ms;
      best = ms < best ? ms : best;
    }
  }
  printf("%-34s lds %5zu B  avg %7.1f us  min %7.1f us   (K1's skeleton of memory operations: events + stores%s, one wait + one barrier)\n", name, lds, sum / reps * 1e3,
         best * 1e3, BANDS ? " + 36 KB of bands per tile out of L2 into LDS" : "");
  return 0;
}

template <bool BND, bool LOADS, bool STORES, bool SPEC>
__global__ __launch_bounds__(THREADS) void k_tile(const uint16_t* __restrict__ xs, const uint16_t* __restrict__ ys, const long long* __restrict__ ts,
                                                  const int4* __restrict__ bounds, uint16_t* __restrict__ frame16, unsigned* sink) {
  extern __shared__ unsigned char smem[];
  const int tile = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
  const size_t fo = (size_t)f * N;
  // the tile's event range: computed (an evenly filled scan) or read from its boundary record (one dependent round trip)
  int lb_s = (int)((long long)tile * N / TILES), lb_e = (int)((long long)(tile + 1) * N / TILES);
  unsigned acc = 0;
  uint4 xv = make_uint4(0, 0, 0, 0), yv = xv;
  longlong2 tv[4] = {};
  const auto load = [&](int s, int e) {
    const int a0 = s & ~(EPT - 1);
    const int base = min(a0 + tid * EPT, (N - 1) & ~(EPT - 1));
    if (a0 + tid * EPT < e) {
      xv = *(const uint4*)(xs + fo + base);
      yv = *(const uint4*)(ys + fo + base);
#pragma unroll
      for (int q = 0; q < 4; ++q) tv[q] = *(const longlong2*)(ts + fo + base + 2 * q);
    }
  };
  if (SPEC && LOADS) load(lb_s, lb_e);  // issued BEFORE the boundary record has arrived (the estimate is exact here)
  if (BND) {
    const int4 b_lo = bounds[(size_t)f * (TILES + 1) + tile], b_hi = bounds[(size_t)f * (TILES + 1) + tile + 1];
    lb_s = b_lo.x;
    lb_e = b_hi.x;
  }
  if (LOADS && !SPEC) load(lb_s, lb_e);
  if (LOADS) {
    acc = xv.x ^ xv.y ^ xv.z ^ xv.w ^ yv.x ^ yv.y ^ yv.z ^ yv.w;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc ^= (unsigned)tv[q].x ^ (unsigned)tv[q].y;
    if (SPEC) acc ^= (unsigned)(lb_s + lb_e);
  }
  if (STORES) {
    uint16_t* col = frame16 + ((size_t)f * TILES * W + (size_t)tile * W) * ROWS;
    for (int i = tid; i < W * ROWS; i += THREADS) col[i] = (uint16_t)(acc + i);
  } else if (acc == 0x12345678u) {
    sink[0] = acc;
  }
  if (smem[0] == 77 && acc == 0x87654321u) sink[1] = 1;  // (keeps the dynamic LDS allocation)
}

template <bool BND, bool LOADS, bool STORES, bool SPEC>
int run(const char* name, size_t lds, const uint16_t* xs, const uint16_t* ys, const long long* ts, const int4* bounds, uint16_t* frame16, unsigned* sink) {
  auto kern = k_tile<BND, LOADS, STORES, SPEC>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float best = 1e9f, sum = 0;
  const int reps = 12;
  for (int rep = 0; rep < reps + 2; ++rep) {
    CK(hipEventRecord(a));
    const size_t g = (size_t)(rep % GROUPS) * N * FRAMES;  // the groups take turns: nothing is re-read out of the 256 MiB MALL (as bench.py)
    hipLaunchKernelGGL(kern, dim3(TILES, FRAMES), dim3(THREADS), lds, 0, xs + g, ys + g, ts + g, bounds, frame16, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep >= 2) {
      best = ms < best ? ms : best;
      sum += ms;
    }
  }
  const double moved = (LOADS ? 12.0 * N * FRAMES : 0.0) + (STORES ? 2.0 * W * ROWS * TILES * FRAMES : 0.0);
  printf("%-34s lds %5zu B  avg %7.1f us  min %7.1f us   moved %6.1f MB at %5.2f TB/s   algorithmic 768 MB at %5.2f TB/s = %.2f of 8\n", name, lds,
         sum / reps * 1e3, best * 1e3, moved / 1e6, moved / (sum / reps) / 1e9, 768e6 / (sum / reps) / 1e9, 768e6 / (sum / reps) / 1e9 / 8.0);
  return 0;
}

int main(int argc, char** argv) {
  uint16_t *xs, *ys, *frame16;
  long long* ts;
  int4* bounds;
  unsigned* sink;
  const size_t n = (size_t)N * FRAMES;
  CK(hipMalloc(&xs, n * 2 * GROUPS));
  CK(hipMalloc(&ys, n * 2 * GROUPS));
  CK(hipMalloc(&ts, n * 8 * GROUPS));
  CK(hipMalloc(&frame16, (size_t)FRAMES * TILES * W * ROWS * 2));
  CK(hipMalloc(&bounds, sizeof(int4) * FRAMES * (TILES + 1)));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(xs, 1, n * 2 * GROUPS));
  CK(hipMemset(ys, 2, n * 2 * GROUPS));
  CK(hipMemset(ts, 3, n * 8 * GROUPS));
  std::vector<int4> hb((size_t)FRAMES * (TILES + 1));
  for (int f = 0; f < FRAMES; ++f)
    for (int t = 0; t <= TILES; ++t) hb[(size_t)f * (TILES + 1) + t] = make_int4((int)((long long)t * N / TILES), 0, 0, 0);
  CK(hipMemcpy(bounds, hb.data(), hb.size() * sizeof(int4), hipMemcpyHostToDevice));
  CK(hipDeviceSynchronize());
  const size_t ldss[] = {0, 49 * 1024, 63 * 1024};  // 4 blocks per CU (thread-bound: 2048 threads) / 3 (K1's occupancy) / 2
  for (size_t lds : ldss) {
    if (run<false, true, true, false>("loads + stores", lds, xs, ys, ts, bounds, frame16, sink)) return 1;
    if (run<true, true, true, false>("boundary record -> loads + stores", lds, xs, ys, ts, bounds, frame16, sink)) return 1;
    if (run<true, true, true, true>("speculative loads | boundary record", lds, xs, ys, ts, bounds, frame16, sink)) return 1;
  }
  {
    unsigned* lut;
    uint16_t* xmap;
    CK(hipMalloc(&lut, (size_t)CAM_W * CAM_H * 4));
    CK(hipMalloc(&xmap, (size_t)TILES * W * ROWS * 2 + 4096));
    CK(hipMemset(lut, 5, (size_t)CAM_W * CAM_H * 4));
    CK(hipMemset(xmap, 6, (size_t)TILES * W * ROWS * 2 + 4096));
    CK(hipDeviceSynchronize());
    if (run_bands<false>("events + stores, wait + barrier", xs, ys, ts, lut, xmap, frame16)) return 1;
    if (run_bands<true>("... + LUT / X-map bands (LDS-direct)", xs, ys, ts, lut, xmap, frame16)) return 1;
  }
  if (run<false, true, false, false>("loads only", 49 * 1024, xs, ys, ts, bounds, frame16, sink)) return 1;
  if (run<false, false, true, false>("stores only", 49 * 1024, xs, ys, ts, bounds, frame16, sink)) return 1;
  if (run<false, true, false, false>("loads only", 0, xs, ys, ts, bounds, frame16, sink)) return 1;
  if (run<false, false, true, false>("stores only", 0, xs, ys, ts, bounds, frame16, sink)) return 1;
  return 0;
}
