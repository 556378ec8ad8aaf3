// Launch-ordering micro-benchmark for the stage-overlap design: how long after kernel X ends does a dependent
// kernel Y start, when the dependency is (a) plain stream order, (b) stream order with an event record in between, (c) an event
// across two streams, and (d) does hipExtAnyOrderLaunch let two kernels of ONE stream run side by side on gfx950?
// Device timestamps (wall_clock64, 100 MHz) written by single-wave kernels that spin for a given time.
//   hipcc --offload-arch=gfx950 -O2 tools/queue_latency.hip -o tools/queue_latency && tools/queue_latency
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin_flag(unsigned long long* stamps, int slot, unsigned long long ticks, unsigned int* flag, unsigned int value) {
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) stamps[2 * slot] = t0;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    stamps[2 * slot + 1] = wall_clock64();
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // the kernel itself publishes "done"
  }
}
__global__ void spin(unsigned long long* stamps, int slot, unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) stamps[2 * slot] = t0;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) stamps[2 * slot + 1] = wall_clock64();
}
int main() {
  unsigned long long* d; CK(hipMalloc(&d, 64 * sizeof(unsigned long long)));
  unsigned long long h[64];
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  hipEvent_t e1, e2; CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  const unsigned long long T = 5000;   // 50 us at 100 MHz
  auto gap = [&](int from, int to) { return ((double)h[2 * to] - (double)h[2 * from + 1]) / 100.0; };   // us from the end of `from` to the start of `to`
  unsigned int* flag = nullptr;
  CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
  CK(hipMemset(flag, 0, 8));
  unsigned int seq = 0;
  const char* names[] = {"(a) same stream, back to back", "(b) same stream, event record between", "(c) other stream through an event", "(c2) there and back: A -> B -> A",
                         "(d) any-order launch in one stream: start of 2nd minus start of 1st",
                         "(e) other stream through hipStreamWriteValue32 / hipStreamWaitValue32", "(f) other stream: the kernel stores a flag, hipStreamWaitValue32 polls it"};
  for (int mode = 0; mode < 7; ++mode) {
    std::vector<double> g;
    for (int rep = 0; rep < 30; ++rep) {
      ++seq;
      if (mode == 6) hipLaunchKernelGGL(spin_flag, dim3(1), dim3(64), 0, a, d, 0, T, flag, seq);
      else hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, d, 0, T);
      if (mode == 5) { CK(hipStreamWriteValue32(a, flag, seq, 0)); CK(hipStreamWaitValue32(b, flag, seq, hipStreamWaitValueGte, 0xffffffffu)); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, d, 1, T); }
      if (mode == 6) { CK(hipStreamWaitValue32(b, flag, seq, hipStreamWaitValueGte, 0xffffffffu)); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, d, 1, T); }
      if (mode == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, d, 1, T);
      if (mode == 1) { CK(hipEventRecord(e1, a)); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, d, 1, T); }
      if (mode == 2) { CK(hipEventRecord(e1, a)); CK(hipStreamWaitEvent(b, e1, 0)); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, d, 1, T); }
      if (mode == 3) { CK(hipEventRecord(e1, a)); CK(hipStreamWaitEvent(b, e1, 0)); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, d, 1, T);
                       CK(hipEventRecord(e2, b)); CK(hipStreamWaitEvent(a, e2, 0)); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, d, 2, T); }
      if (mode == 4) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, nullptr, nullptr, hipExtAnyOrderLaunch, d, 1, T);
      CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
      CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
      if (mode == 3) g.push_back(gap(0, 1) + gap(1, 2));
      else if (mode == 4) g.push_back(((double)h[2] - (double)h[0]) / 100.0);
      else g.push_back(gap(0, 1));
    }
    std::sort(g.begin(), g.end());
    printf("%-75s median %7.2f us  min %7.2f  max %7.2f\n", names[mode], g[g.size() / 2], g.front(), g.back());
  }
  return 0;
}
