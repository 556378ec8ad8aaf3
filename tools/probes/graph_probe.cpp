// tools/probes/graph_probe.cpp : what does one frame's worth of launches cost the host -- issued call by call on four streams tied by events
// (16 kernels, 12 event operations, as the c2 frame) against one hipGraphLaunch of the same DAG captured from those streams?
// hipcc --offload-arch=gfx950 -O2 -o graph_probe graph_probe.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_work(float* p, int n, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = p[i];
  for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f;
  p[i] = v;
}
struct Lanes { hipStream_t s[4]; hipEvent_t e[8]; float* buf[4]; };
static void issue_frame(Lanes& L, int n, int iters) {
  // lane 0: 3 kernels (pre) -> lane 1: 3 kernels (integrate) -> lane 2: 4 kernels (draw) -> lane 3: 6 kernels (fill); fork/join through events
  auto launch = [&](int lane) { hipLaunchKernelGGL(k_work, dim3((n + 255) / 256), dim3(256), 0, L.s[lane], L.buf[lane], n, iters); };
  for (int k = 0; k < 3; ++k) launch(0);
  hipEventRecord(L.e[0], L.s[0]); hipStreamWaitEvent(L.s[1], L.e[0], 0); hipStreamWaitEvent(L.s[2], L.e[0], 0);
  for (int k = 0; k < 3; ++k) launch(1);
  launch(2);
  hipEventRecord(L.e[1], L.s[1]); hipStreamWaitEvent(L.s[2], L.e[1], 0);
  for (int k = 0; k < 3; ++k) launch(2);
  hipEventRecord(L.e[2], L.s[2]); hipStreamWaitEvent(L.s[3], L.e[2], 0);
  for (int k = 0; k < 6; ++k) launch(3);
  hipEventRecord(L.e[3], L.s[3]);
}
int main() {
  Lanes L;
  const int n = 1 << 16;
  for (int i = 0; i < 4; ++i) { CK(hipStreamCreateWithFlags(&L.s[i], hipStreamNonBlocking)); CK(hipMalloc(&L.buf[i], n * sizeof(float))); CK(hipMemset(L.buf[i], 0, n * sizeof(float))); }
  for (int i = 0; i < 8; ++i) CK(hipEventCreateWithFlags(&L.e[i], hipEventDisableTiming));
  using clk = std::chrono::steady_clock;
  auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  for (int iters : {1, 2000}) {
    for (int f = 0; f < 200; ++f) issue_frame(L, n, iters);
    CK(hipDeviceSynchronize());
    const int F = 2000;
    auto t0 = clk::now();
    for (int f = 0; f < F; ++f) {
      issue_frame(L, n, iters);
      hipStreamWaitEvent(L.s[0], L.e[2], 0);                 // (the next frame's lane 0 behind this frame's draw, like pre_gate)
    }
    auto t1 = clk::now();
    CK(hipDeviceSynchronize());
    auto t2 = clk::now();
    printf("direct, iters %4d: issue %.1f us/frame, finished %.1f us/frame\n", iters, us(t0, t1) / F, us(t0, t2) / F);
    // the same DAG as a graph: capture from lane 0 (fork into the others through the events), join back before EndCapture
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(L.s[0], hipStreamCaptureModeGlobal));
    issue_frame(L, n, iters);
    CK(hipStreamWaitEvent(L.s[0], L.e[3], 0));
    CK(hipEventRecord(L.e[4], L.s[1])); CK(hipStreamWaitEvent(L.s[0], L.e[4], 0));
    CK(hipEventRecord(L.e[5], L.s[2])); CK(hipStreamWaitEvent(L.s[0], L.e[5], 0));
    CK(hipStreamEndCapture(L.s[0], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    size_t nn = 0; hipGraphGetNodes(g, nullptr, &nn);
    for (int f = 0; f < 200; ++f) CK(hipGraphLaunch(ge, L.s[0]));
    CK(hipDeviceSynchronize());
    t0 = clk::now();
    for (int f = 0; f < F; ++f) CK(hipGraphLaunch(ge, L.s[0]));
    t1 = clk::now();
    CK(hipDeviceSynchronize());
    t2 = clk::now();
    printf("graph (%zu nodes), iters %4d: issue %.1f us/frame, finished %.1f us/frame\n", nn, iters, us(t0, t1) / F, us(t0, t2) / F);
    // two graphs alternating on two streams (consecutive frames overlap like the lanes do)
    t0 = clk::now();
    for (int f = 0; f < F; ++f) CK(hipGraphLaunch(ge, L.s[f & 1 ? 3 : 0]));
    t1 = clk::now();
    CK(hipDeviceSynchronize());
    t2 = clk::now();
    printf("graph on alternating streams, iters %4d: issue %.1f us/frame, finished %.1f us/frame\n", iters, us(t0, t1) / F, us(t0, t2) / F);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
