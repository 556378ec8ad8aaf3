// Frame packing + brick occupancy kernels (gfx950).
//   pack_frame / pack_color : upload-side re-layout into the HBM-resident image formats
//   mark_bricks             : pre_normal.fs:22-33 call site of mark_brick(), inc_bricks.glsl:40-58
//   update_occupied         : updateOccupiedBricks(), recon_integration.cpp:430-445, without the host round trip
#include "sampling.hpp"
#include "bricks_dev.hpp"

namespace rr {

__global__ __launch_bounds__(256) void k_pack_frame(const float2* __restrict__ depth_rg, const float* __restrict__ quality,
                                                    const float* __restrict__ silhouette, float4* __restrict__ dqs, float* __restrict__ depth, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = depth_rg[i].x;
    dqs[i] = make_float4(d, quality[i], silhouette[i], 0.0f);
    depth[i] = d;
  }
}
__global__ __launch_bounds__(256) void k_pack_color(const uint8_t* __restrict__ rgb, uchar4* __restrict__ rgba, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    rgba[i] = make_uchar4(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], 255);
}
__global__ __launch_bounds__(256) void k_fill_u32(uint32_t* __restrict__ p, uint32_t v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

static inline int grid_for(size_t n, int block = 256, int cap = 2048) {
  size_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > (size_t)cap ? cap : g));
}
void launch_pack_frame(hipStream_t st, const float* depth_rg, const float* quality, const float* silhouette, float4* dqs, float* depth, size_t n) {
  hipLaunchKernelGGL(k_pack_frame, dim3(grid_for(n)), dim3(256), 0, st, (const float2*)depth_rg, quality, silhouette, dqs, depth, n);
}
void launch_pack_color(hipStream_t st, const uint8_t* rgb, uchar4* rgba, size_t n) {
  hipLaunchKernelGGL(k_pack_color, dim3(grid_for(n)), dim3(256), 0, st, rgb, rgba, n);
}
void launch_fill_u32(hipStream_t st, uint32_t* p, uint32_t v, size_t n) {
  hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(n)), dim3(256), 0, st, p, v, n);
}

// One thread per depth pixel and stream (pre_normal.fs:22-33 runs per fragment of every layer).
__global__ __launch_bounds__(256) void k_mark_bricks(StreamTable T, FrameImages F, Bricks B) {
  const int px = blockIdx.x * 64 + (threadIdx.x & 63);
  const int py = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int layer = blockIdx.z;
  bool own = false, nbr = false;
  uint32_t id_own = 0, id_nbr = 0;
  if (px < F.w && py < F.h) {
    const float u = ((float)px + 0.5f) / (float)F.w, v = ((float)py + 0.5f) / (float)F.h;
    const int nx = axis_nearest(u, F.w), ny = axis_nearest(v, F.h);     // NEAREST fetch at the pixel's own centre
    const float d = F.depth[((size_t)layer * F.h + ny) * F.w + nx];
    if (!(d <= 0.0f || d >= 1.0f)) {                                    // is_outside(), pre_normal.fs:22-24
      const StreamLut& L = T.s[layer];
      const float3 pos = tex3d_rgba_xyz(L.xyz, L.xyz_res, u, v, d);     // world position, :32
      mark_brick_ids(B, pos, own, id_own, nbr, id_nbr);
    }
  }
  wave_count(B.counters, id_nbr, nbr);
  wave_count(B.counters, id_own, own);
}
void launch_mark_bricks(hipStream_t st, const StreamTable& T, const FrameImages& F, const Bricks& B) {
  dim3 grid((F.w + 63) / 64, (F.h + 3) / 4, T.n);
  hipLaunchKernelGGL(k_mark_bricks, grid, dim3(256), 0, st, T, F, B);
}

// flags[b] = counter[b] >= min_voxels, the compacted occupied list and its length (one atomic per wave)
__global__ __launch_bounds__(256) void k_update_occupied(Bricks B, uint32_t min_voxels) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool occ = (b < B.n) && (B.counters[b] >= min_voxels);
  if (b < B.n) B.flags[b] = occ ? 1 : 0;
  const unsigned long long m = __ballot(occ);
  if (m == 0ull) return;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(B.num_occupied, (uint32_t)__popcll(m));
  base = __shfl(base, 0);
  if (occ) B.occupied[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)b;
}
void launch_update_occupied(hipStream_t st, const Bricks& B, uint32_t min_voxels, int zero_count) {
  if (zero_count) hipMemsetAsync(B.num_occupied, 0, sizeof(uint32_t), st);
  hipLaunchKernelGGL(k_update_occupied, dim3((B.n + 255) / 256), dim3(256), 0, st, B, min_voxels);
}

}  // namespace rr
