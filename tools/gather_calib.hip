// FETCH_SIZE calibration on KNOWN byte counts (VERDICT r01 "Next round" 2): MI355X_MICROARCH.md says the counter reports half the
// bytes of a wide coalesced read on gfx950 and that other access shapes are uncalibrated.  The integrate / march kernels gather
// 16-byte LUT texels and 4-byte voxels, so before their FETCH_SIZE is doubled the same counter is read on four access shapes
// whose bytes are known:
//   k_calib_stream     coalesced float4 stream over 512 MiB                                   bytes = 512 MiB
//   k_calib_gather23   every 16-byte texel of a 128 MiB table exactly once, scattered order     bytes >= 128 MiB (64-B sectors: 512 MiB)
//   k_calib_gather26   the same over a 1 GiB table (larger than the 256 MiB Infinity Cache)     bytes >= 1 GiB
//   k_calib_tap8       the 2x2x2 texel neighbourhood of scattered positions in a 128^3 x 16 B LUT (the trilinear footprint)
// Run:  rocprofv3 --pmc FETCH_SIZE -d <dir> -o calib -- tools/gather_calib     (and once more with --kernel-trace --stats for the times)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_calib_stream(const float4* __restrict__ t, size_t n, float* __restrict__ out) {
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = t[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 123.456f) out[0] = acc;
}
template <int kLog2N>
__device__ __forceinline__ void gather_body(const float4* __restrict__ t, float* __restrict__ out) {
  constexpr unsigned long long N = 1ull << kLog2N;
  float acc = 0.0f;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned long long j = (i * 2654435761ull + 12345ull) & (N - 1);      // odd multiplier: a bijection on [0, N)
    const float4 v = t[j];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_calib_gather23(const float4* __restrict__ t, float* __restrict__ out) { gather_body<23>(t, out); }
__global__ __launch_bounds__(256) void k_calib_gather26(const float4* __restrict__ t, float* __restrict__ out) { gather_body<26>(t, out); }
__global__ __launch_bounds__(256) void k_calib_tap8(const float4* __restrict__ t, unsigned n_pos, float* __restrict__ out) {
  float acc = 0.0f;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pos; i += gridDim.x * blockDim.x) {
    const unsigned j = (i * 2654435761u + 12345u) & ((1u << 21) - 1u);            // a texel of the 128^3 grid
    const unsigned x = j & 127u, y = (j >> 7) & 127u, z = j >> 14;
    const unsigned x1 = x < 127u ? x + 1 : x, y1 = y < 127u ? y + 1 : y, z1 = z < 127u ? z + 1 : z;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 v = t[(((k & 4) ? z1 : z) * 128u + ((k & 2) ? y1 : y)) * 128u + ((k & 1) ? x1 : x)];
      acc += v.x + v.y;
    }
  }
  if (acc == 123.456f) out[0] = acc;
}
int main() {
  const size_t big = 1ull << 30;                       // 1 GiB
  float4* t; float* out;
  CK(hipMalloc(&t, big)); CK(hipMalloc(&out, 64));
  CK(hipMemset(t, 0, big));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timed = [&](const char* name, double bytes, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(a)); for (int i = 0; i < 10; ++i) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
    printf("%-18s %8.3f ms   useful bytes %.1f MiB   %.1f GB/s of useful bytes\n", name, ms, bytes / 1048576.0, bytes / ms / 1e6);
  };
  timed("k_calib_stream", 512.0 * 1048576, [&] { hipLaunchKernelGGL(k_calib_stream, dim3(4096), dim3(256), 0, 0, t, (size_t)(512ull << 20) / 16, out); });
  timed("k_calib_gather23", 128.0 * 1048576, [&] { hipLaunchKernelGGL(k_calib_gather23, dim3(4096), dim3(256), 0, 0, t, out); });
  timed("k_calib_gather26", 1024.0 * 1048576, [&] { hipLaunchKernelGGL(k_calib_gather26, dim3(8192), dim3(256), 0, 0, t, out); });
  timed("k_calib_tap8", (double)(1u << 21) * 8 * 16, [&] { hipLaunchKernelGGL(k_calib_tap8, dim3(2048), dim3(256), 0, 0, t, 1u << 21, out); });
  CK(hipDeviceSynchronize());
  return 0;
}
