// K0 + K1: TSDF integration (tsdf_integration.vs:23-59 driven by ReconIntegration::integrate(),
// recon_integration.cpp:242-269), fused with the volume clear.
//
// Launch shape: one 256-thread workgroup per 8x8x8 storage tile (2 KiB of HBM, written as two fully
// coalesced 1 KiB wave-stores per wave).  Voxels that no occupied brick lists keep the clear value
// -limit (the reference clears the volume and then only draws the occupied bricks' voxel lists);
// a wave whose ballot of "drawn" voxels is empty skips the stream loop entirely.
#include "sampling.hpp"

namespace rr {

// tsdf_integration.vs:28-55 for one voxel centre p (unit-cube coordinates)
__device__ __forceinline__ float integrate_voxel(const StreamTable& T, const FrameImages& F, float limit, float px, float py, float pz) {
  float weighted_tsd = limit;
  float total_weight = 0.0f;
  for (int i = 0; i < T.n; ++i) {
    const StreamLut& L = T.s[i];
    const float3 pc = tex3d_rgba_xyz(L.inv, L.inv_res, px, py, pz);
    const Dqs q = dqs_fetch(F, i, pc.x, pc.y);
    if (dqs_silhouette(q) < 1.0f) {
      if (weighted_tsd >= limit) {
        weighted_tsd = -limit;
        continue;
      }
    }
    const float sdist = pc.z - dqs_depth(q);
    if (sdist <= -limit) {
      weighted_tsd = -limit;
    } else if (sdist >= limit) {
    } else {
      const float weight = dqs_quality(q);
      weighted_tsd = (weighted_tsd * total_weight + weight * sdist) / (total_weight + weight);
      total_weight += weight;
    }
  }
  return weighted_tsd;
}

// Is voxel (x,y,z) in the index list of at least one occupied brick?  (volume_sampler.cpp:50-62 lists,
// restated as per-axis voxel -> brick tables; 1 candidate per axis in every aligned configuration.)
__device__ __forceinline__ bool voxel_drawn(const Bricks& B, int x, int y, int z) {
  const int fx = B.vox_first[0][x], nx = B.vox_count[0][x];
  const int fy = B.vox_first[1][y], ny = B.vox_count[1][y];
  const int fz = B.vox_first[2][z], nz = B.vox_count[2][z];
  bool any = false;
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i)
        any |= B.flags[((size_t)(fz + k) * B.res[1] + (fy + j)) * B.res[0] + (fx + i)] != 0;
  return any;
}

__global__ __launch_bounds__(256) void k_integrate(StreamTable T, FrameImages F, Volume V, Bricks B, int use_bricks, int n_tiles) {
  // XCD-aware mapping: blocks b, b+8, b+16.. share an XCD (round-robin dispatch); give each XCD one
  // contiguous run of tiles so neighbouring tiles (shared LUT texels / image pixels) hit the same L2.
  int tile = blockIdx.x;
  if ((n_tiles & 7) == 0) tile = (blockIdx.x & 7) * (n_tiles >> 3) + (blockIdx.x >> 3);
  const int tx = tile % V.ntx, ty = (tile / V.ntx) % V.nty, tz = V.own_tz0 + tile / (V.ntx * V.nty);
  float* __restrict__ out = V.data + ((((size_t)(tz - V.tz0) * V.nty + ty) * V.ntx + tx) << 9);
  const float sx = 1.0f / (float)V.res[0], sy = 1.0f / (float)V.res[1], sz = 1.0f / (float)V.res[2];   // volume_sampler.cpp:36-38
  const float limit = V.limit;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int l = threadIdx.x + half * 256;
    const int x = tx * 8 + (l & 7), y = ty * 8 + ((l >> 3) & 7), z = tz * 8 + (l >> 6);
    bool drawn = (x < V.res[0]) && (y < V.res[1]) && (z < V.res[2]);
    if (drawn && use_bricks) drawn = voxel_drawn(B, x, y, z);
    float v = -limit;                                                   // clearImage(-limit), :249-250
    if (__ballot(drawn) != 0ull) {
      if (drawn) v = integrate_voxel(T, F, limit, ((float)x + 0.5f) * sx, ((float)y + 0.5f) * sy, ((float)z + 0.5f) * sz);
    }
    out[l] = v;
  }
}

void launch_integrate(hipStream_t st, const StreamTable& T, const FrameImages& F, const Volume& V, const Bricks& B, int use_bricks) {
  const int n_tiles = V.ntx * V.nty * (V.own_tz1 - V.own_tz0);
  hipLaunchKernelGGL(k_integrate, dim3(n_tiles), dim3(256), 0, st, T, F, V, B, use_bricks, n_tiles);
}

// ---- linear <-> tile-major conversion for the download/upload entry points
__global__ __launch_bounds__(256) void k_volume_to_linear(Volume V, float* __restrict__ lin) {
  const size_t n = (size_t)V.res[0] * V.res[1] * V.res[2];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % V.res[0]), y = (int)((i / V.res[0]) % V.res[1]), z = (int)(i / ((size_t)V.res[0] * V.res[1]));
    const int tz = z >> 3;
    lin[i] = (tz >= V.tz0 && tz < V.tz1) ? V.data[vol_index(V, x, y, z)] : 0.0f;
  }
}
__global__ __launch_bounds__(256) void k_volume_from_linear(Volume V, const float* __restrict__ lin) {
  const size_t n = (size_t)V.res[0] * V.res[1] * V.res[2];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % V.res[0]), y = (int)((i / V.res[0]) % V.res[1]), z = (int)(i / ((size_t)V.res[0] * V.res[1]));
    const int tz = z >> 3;
    if (tz >= V.tz0 && tz < V.tz1) V.data[vol_index(V, x, y, z)] = lin[i];
  }
}
void launch_volume_to_linear(hipStream_t st, const Volume& V, float* linear) {
  hipLaunchKernelGGL(k_volume_to_linear, dim3(2048), dim3(256), 0, st, V, linear);
}
void launch_volume_from_linear(hipStream_t st, const Volume& V, const float* linear) {
  hipLaunchKernelGGL(k_volume_from_linear, dim3(2048), dim3(256), 0, st, V, linear);
}

}  // namespace rr
