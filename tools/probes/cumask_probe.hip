// Which CUs does bit i of a hipExtStreamCreateWithCUMask mask select on a MI355X (8 XCDs x 32 CUs)?  Launches a grid of short
// workgroups on streams with a few masks and histograms where they ran: XCC_ID (hwreg 20) and HW_ID (hwreg 4: CU_ID bits 8-11, SH_ID bit 12,
// SE_ID bits 13-15).  Run: tools/probes/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_where(uint32_t* __restrict__ out, int spin) {
  const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20), hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  float a = (float)threadIdx.x;
  for (int k = 0; k < spin; ++k) a = a * 1.0001f + 0.5f;               // long enough that the whole grid is resident at once
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 15u) | (((hw >> 8) & 15u) << 4) | (((hw >> 12) & 1u) << 8) | (((hw >> 13) & 7u) << 9) | (a == 1.5f ? 1u << 31 : 0u);
}

static void run(const char* name, const uint32_t mask[8]) {
  hipStream_t st;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, mask);
  if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); return; }
  const int blocks = 2048;
  uint32_t* d; CK(hipMalloc((void**)&d, blocks * 4));
  hipLaunchKernelGGL(k_where, dim3(blocks), dim3(256), 0, st, d, 20000);
  CK(hipStreamSynchronize(st));
  std::vector<uint32_t> h(blocks); CK(hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost));
  int per_xcc[16] = {}; std::vector<int> cus(16 * 128, 0);
  for (uint32_t v : h) { per_xcc[v & 15]++; cus[(v & 15) * 128 + ((v >> 4) & 127)]++; }
  printf("%-28s blocks per XCC:", name);
  for (int x = 0; x < 8; ++x) printf(" %4d", per_xcc[x]);
  printf("   distinct (SE,SH,CU) per XCC:");
  for (int x = 0; x < 8; ++x) { int n = 0; for (int k = 0; k < 128; ++k) n += cus[x * 128 + k] != 0; printf(" %2d", n); }
  printf("\n");
  CK(hipFree(d)); CK(hipStreamDestroy(st));
}

int main() {
  uint32_t m[8];
  for (int k = 0; k < 8; ++k) m[k] = 0xffffffffu;
  run("all 256 bits", m);
  memset(m, 0, sizeof(m)); m[0] = 0xffffffffu;
  run("bits 0-31", m);
  memset(m, 0, sizeof(m)); for (int i = 0; i < 256; i += 8) m[i >> 5] |= 1u << (i & 31);
  run("bits i % 8 == 0", m);
  memset(m, 0, sizeof(m)); for (int i = 0; i < 256; ++i) if ((i & 7) < 3) m[i >> 5] |= 1u << (i & 31);
  run("bits i % 8 < 3", m);
  memset(m, 0, sizeof(m)); for (int i = 0; i < 64; ++i) m[i >> 5] |= 1u << (i & 31);
  run("bits 0-63", m);
  memset(m, 0, sizeof(m)); for (int i = 0; i < 256; ++i) if ((i >> 3) < 8) m[i >> 5] |= 1u << (i & 31);
  run("bits i / 8 < 8", m);
  return 0;
}
