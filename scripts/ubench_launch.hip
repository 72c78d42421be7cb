// ubench_launch.hip — what one kernel launch costs the HOST on this box (the rebalancer's decision loop and the rank batches are bound by it):
// back-to-back launches of an empty kernel on one stream (small / 1 KB of arguments), and the same sequence replayed as a hipGraph.
// build: hipcc --offload-arch=gfx950 -O2 -o scripts/ubench_launch scripts/ubench_launch.hip ; run on the GPU box (env HIP_FORCE_DEV_KERNARG=0/1)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { unsigned long long w[120]; };
__global__ void k_small(unsigned* p, unsigned i) { if (p && threadIdx.x == 12345u) p[0] = i; }
__global__ void k_slow(unsigned* p, unsigned i) {  // ~7 us of nothing
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 700ull) __builtin_amdgcn_s_sleep(8);
  if (p && threadIdx.x == 12345u) p[0] = i;
}
__global__ void k_big(Big b, unsigned* p) { if (p && threadIdx.x == 12345u) p[0] = (unsigned)b.w[7]; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned* d;
  hipMalloc(&d, 64);
  Big b{};
  const int N = 640, REP = 20;
  for (int mode = 0; mode < 4; ++mode) {
    double enq = 0, tot = 0;
    for (int r = 0; r < REP + 2; ++r) {
      hipStreamSynchronize(s);
      const double t0 = now();
      for (int i = 0; i < N; ++i) {
        if (mode == 0) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, d, (unsigned)i);
        else if (mode == 1) hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, b, d);
        else if (mode == 2) hipLaunchKernelGGL(k_small, dim3(391), dim3(512), 0, s, d, (unsigned)i);
        else hipLaunchKernelGGL(k_slow, dim3(1), dim3(64), 0, s, d, (unsigned)i);
      }
      const double t1 = now();
      hipStreamSynchronize(s);
      const double t2 = now();
      if (r >= 2) enq += t1 - t0, tot += t2 - t0;
    }
    std::printf("%s: %.2f us per launch to enqueue, %.2f us per launch until done\n", mode == 0 ? "1 block, 16 B of arguments" : mode == 1 ? "1 block, 960 B of arguments" : mode == 2 ? "391 blocks of 512, 16 B" : "1 block that takes 7 us",
                enq / REP / N, tot / REP / N);
  }
  // the same 640 launches as a graph
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, d, (unsigned)i);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  double enq = 0, tot = 0;
  for (int r = 0; r < REP + 2; ++r) {
    hipStreamSynchronize(s);
    const double t0 = now();
    hipGraphLaunch(ge, s);
    const double t1 = now();
    hipStreamSynchronize(s);
    const double t2 = now();
    if (r >= 2) enq += t1 - t0, tot += t2 - t0;
  }
  std::printf("hipGraph of %d one-block kernels: %.2f us per node to launch, %.2f us per node until done\n", N, enq / REP / N, tot / REP / N);
  return 0;
}
