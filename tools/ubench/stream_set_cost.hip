// stream_set_cost.hip -- does a HIP API call cost more on streams created LATER in a process?  (profiles/r05_ingest.md, section 2)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/stream_set_cost.hip -o /tmp/stream_set_cost && /tmp/stream_set_cost
// For each of several SETS of four streams created one set after the other (priorities high / high / normal / high like the
// ingest's; the earlier sets stay alive, or are destroyed first with argv[1] = "destroy"), the host time of: an empty kernel
// launch, hipEventRecord + hipStreamWaitEvent between two streams of the set, a 720 KB pinned H2D copy call -- each averaged
// over 4000 calls with the GPU kept just busy enough that nothing queues up (a stream query every 64 calls).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k_empty(int* p) { if (p && threadIdx.x == 12345) *p = 1; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const bool destroy = argc > 1 && !strcmp(argv[1], "destroy");
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  void *h = nullptr, *d = nullptr;
  const size_t bytes = 720 << 10;
  CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
  CK(hipMalloc(&d, bytes));
  // (an engine's four slot streams exist before the ingest's in the real process)
  std::vector<hipStream_t> pre(4);
  for (auto& s : pre) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<std::vector<hipStream_t>> sets;
  const int N = 4000;
  for (int set = 0; set < 4; ++set) {
    if (destroy && !sets.empty()) {
      for (auto s : sets.back()) CK(hipStreamDestroy(s));
      sets.pop_back();
    }
    std::vector<hipStream_t> st(4);
    const int prio[4] = {hi, hi, (lo + hi) / 2, hi};
    for (int i = 0; i < 4; ++i) CK(hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, prio[i]));
    sets.push_back(st);
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[0], (int*)nullptr);
    CK(hipDeviceSynchronize());
    double t0 = now();
    for (int i = 0; i < N; ++i) {
      hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[0], (int*)nullptr);
      if ((i & 63) == 63) (void)hipStreamQuery(st[0]);
    }
    const double t_launch = (now() - t0) / N;
    CK(hipDeviceSynchronize());
    t0 = now();
    for (int i = 0; i < N; ++i) {
      CK(hipEventRecord(ev, st[2]));
      CK(hipStreamWaitEvent(st[0], ev, 0));
      if ((i & 63) == 63) (void)hipStreamQuery(st[0]);
    }
    const double t_ev = (now() - t0) / N;
    CK(hipDeviceSynchronize());
    t0 = now();
    for (int i = 0; i < N / 4; ++i) {
      CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[2]));
      if ((i & 7) == 7) CK(hipStreamSynchronize(st[2]));
    }
    const double t_cp = (now() - t0) / (N / 4);
    CK(hipDeviceSynchronize());
    // the chain the ingest issues per packet, GPU idle in between: copy -> event -> wait -> 3 kernels, then wait for the last one
    t0 = now();
    for (int i = 0; i < 500; ++i) {
      CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st[2]));
      CK(hipEventRecord(ev, st[2]));
      CK(hipStreamWaitEvent(st[0], ev, 0));
      for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[0], (int*)nullptr);
      CK(hipStreamSynchronize(st[0]));
    }
    const double t_chain = (now() - t0) / 500;
    printf("set %d (%s): kernel launch %.2f us, event record + stream wait %.2f us, 720 KB H2D call %.2f us (incl. a sync every 8), copy -> event -> 3 kernels -> sync %.1f us\n",
           set, destroy ? "earlier sets destroyed" : "earlier sets alive", t_launch * 1e6, t_ev * 1e6, t_cp * 1e6, t_chain * 1e6);
    CK(hipEventDestroy(ev));
  }
  return 0;
}
