// tools/probes/handoff_probe.cpp : how long after kernel A ends on one stream does a dependent kernel B start on another?  Kernels stamp
// their own begin / end with the device's real-time counter (100 MHz): no profiler, no timing events.
//   same stream            A ; B
//   event                  A ; hipEventRecord(e, s0) ; hipStreamWaitEvent(s1, e) ; B
//   flag                   A ; signal kernel on s0 (one thread: stores the frame number) ; wait kernel on s1 (one wave spins on it) ; B
// hipcc --offload-arch=gfx950 -O2 -o handoff_probe handoff_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
__global__ void k_work(float* p, int n, int iters, unsigned long long* stamp) {
  if (blockIdx.x == 0 && threadIdx.x == 0) stamp[0] = wall_clock64();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = p[i % n];
  for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f;
  p[i % n] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(&stamp[1], wall_clock64());
}
__global__ void k_signal(volatile unsigned int* flag, unsigned int value) { *flag = value; __threadfence_system(); }
__global__ void k_wait(volatile unsigned int* flag, unsigned int value, unsigned int* timeouts) {
  const unsigned long long t0 = wall_clock64();
  while ((int)(*flag - value) < 0) {
    __builtin_amdgcn_s_sleep(8);
    if (wall_clock64() - t0 > 20000000ull) { atomicAdd(timeouts, 1u); break; }   // 0.2 s: never hang
  }
}
int main() {
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t e, back; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&back, hipEventDisableTiming));
  const int n = 1 << 20, R = 200;
  float* p; CK(hipMalloc(&p, n * sizeof(float))); CK(hipMemset(p, 0, n * sizeof(float)));
  unsigned long long* st; CK(hipMalloc(&st, R * 4 * sizeof(unsigned long long)));
  unsigned int* flag; CK(hipMalloc(&flag, 8)); CK(hipMemset(flag, 0, 8));
  std::vector<unsigned long long> h(R * 4);
  // what a barrier packet costs a stream even when nothing has to be waited for: A ; [record | wait on a long-complete event | both] ; B on ONE stream
  {
    hipEvent_t done; CK(hipEventCreateWithFlags(&done, hipEventDisableTiming)); CK(hipEventRecord(done, s1)); CK(hipDeviceSynchronize());
    const char* what[4] = {"nothing", "a record", "a satisfied wait", "a record and two satisfied waits"};
    for (int v = 0; v < 4; ++v) {
      CK(hipMemset(st, 0, R * 4 * sizeof(unsigned long long))); CK(hipDeviceSynchronize());
      for (int r = 0; r < R; ++r) {
        hipLaunchKernelGGL(k_work, dim3(2048), dim3(256), 0, s0, p, n, 600, st + r * 4);
        if (v == 1 || v == 3) hipEventRecord(e, s0);
        if (v == 2 || v == 3) hipStreamWaitEvent(s0, done, 0);
        if (v == 3) hipStreamWaitEvent(s0, done, 0);
        hipLaunchKernelGGL(k_work, dim3(2048), dim3(256), 0, s0, p, n, 600, st + r * 4 + 2);
      }
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), st, R * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      std::vector<double> gap;
      for (int r = 20; r < R - 1; ++r) gap.push_back(((double)h[r * 4 + 2] - (double)h[r * 4 + 1]) / 100.0);
      std::sort(gap.begin(), gap.end());
      printf("one stream, %s between A and B: A end -> B begin median %.1f us (p10 %.1f, p90 %.1f)\n", what[v], gap[gap.size() / 2], gap[gap.size() / 10], gap[gap.size() * 9 / 10]);
    }
  }
  const char* names[4] = {"same stream", "event", "flag", "flag + event"};   // flag + event: the event orders (always correct), the flag kernel only keeps the consumer's queue spinning until the event is complete
  for (int mode = 0; mode < 4; ++mode) {
    for (int busy = 0; busy < 2; ++busy) {                       // busy: A' keeps s0 busy behind A (like a lane that runs on)
      CK(hipMemset(st, 0, R * 4 * sizeof(unsigned long long)));
      CK(hipDeviceSynchronize());
      for (int r = 0; r < R; ++r) {
        hipLaunchKernelGGL(k_work, dim3(2048), dim3(256), 0, s0, p, n, 600, st + r * 4);
        hipStream_t sb = mode == 0 ? s0 : s1;
        if (mode == 1) { hipEventRecord(e, s0); hipStreamWaitEvent(s1, e, 0); }
        if (mode == 2) { hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, s0, flag, (unsigned)(mode * 100000 + busy * 1000 + r + 1)); hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, s1, flag, (unsigned)(mode * 100000 + busy * 1000 + r + 1), flag + 1); }
        if (mode == 3) {
          hipEventRecord(e, s0);
          hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, s0, flag, (unsigned)(mode * 100000 + busy * 1000 + r + 1));
          hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, s1, flag, (unsigned)(mode * 100000 + busy * 1000 + r + 1), flag + 1);
          hipStreamWaitEvent(s1, e, 0);
        }
        if (busy) hipLaunchKernelGGL(k_work, dim3(2048), dim3(256), 0, s0, p, n, 600, st + R * 4 - 2);   // (stamps thrown away)
        hipLaunchKernelGGL(k_work, dim3(2048), dim3(256), 0, sb, p, n, 600, st + r * 4 + 2);
        if (mode != 0) { hipEventRecord(back, s1); hipStreamWaitEvent(s0, back, 0); }                    // next round's A behind this B
      }
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), st, R * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      std::vector<double> gap, dur;
      for (int r = 20; r < R - 1; ++r) { gap.push_back(((double)h[r * 4 + 2] - (double)h[r * 4 + 1]) / 100.0); dur.push_back(((double)h[r * 4 + 1] - (double)h[r * 4]) / 100.0); }
      std::sort(gap.begin(), gap.end()); std::sort(dur.begin(), dur.end());
      printf("%-12s %s: A end -> B begin median %.1f us (p10 %.1f, p90 %.1f); A itself %.1f us\n", names[mode], busy ? "busy" : "idle", gap[gap.size() / 2], gap[gap.size() / 10], gap[gap.size() * 9 / 10], dur[dur.size() / 2]);
    }
  }
  unsigned int to = 0; CK(hipMemcpy(&to, flag + 1, 4, hipMemcpyDeviceToHost));
  printf("wait-kernel timeouts: %u\n", to);
  return 0;
}
