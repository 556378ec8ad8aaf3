// K0 + K1: TSDF integration (tsdf_integration.vs:23-59 driven by ReconIntegration::integrate(),
// recon_integration.cpp:242-269).
//
// The reference clears the whole volume and then draws the voxel lists of the occupied bricks, one GL
// draw per brick.  Here one frame is two launches over 8x8x8 storage tiles (2 KiB each, tile-major):
//   k_classify_clear_tiles  which tiles contain a voxel of an occupied brick's list -> compact "active" list; tiles that
//                      are inactive but still hold old surface data are reset to -limit in the same pass
//                      (tiles already clear are not touched: no dense 4*V store per frame)
//   k_integrate_tiles* active tiles only (persistent workgroups pulling from the list); a tile is written
//                      as two fully coalesced 1 KiB stores per wave; the _lds variant stages the LUT box in LDS
// Without bricks (setUseBricks(false)) every tile is active and the dense kernel runs on a plain grid.
#include <cstdlib>

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

// index of integrated tile `tile` (int_tz0-relative) in the per-stored-tile tables (tz0-relative)
__device__ __forceinline__ uint32_t stored_tile_index(const Volume& V, int tile) { return (uint32_t)tile + (uint32_t)((V.int_tz0 - V.tz0) * V.nty * V.ntx); }
__device__ __forceinline__ void tile_coords(const Volume& V, int tile, int& tx, int& ty, int& tz) {
  tx = tile % V.ntx; ty = (tile / V.ntx) % V.nty; tz = V.int_tz0 + tile / (V.ntx * V.nty);
}

// One lane per owned tile: is any occupied brick among those whose voxel lists can reach into the tile?  Active tiles go
// to the compacted work list (one atomic per wave); inactive tiles that hold anything but the clear value are reset on
// the spot by the whole wave (2 x 16 B per lane per tile).  Tiles already clear are not touched.
__global__ __launch_bounds__(256) void k_classify_clear_tiles(Volume V, Bricks B, TileState S) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * blockDim.x + threadIdx.x;
  bool active = false, need_clear = false;
  if (tile < S.n) {
    int t[3];
    tile_coords(V, tile, t[0], t[1], t[2]);
    int b0[3], b1[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { b0[a] = B.tile_b0[a][t[a]]; b1[a] = B.tile_b1[a][t[a]]; }   // host-built brick span per tile index
    for (int k = b0[2]; k <= b1[2]; ++k)
      for (int j = b0[1]; j <= b1[1]; ++j)
        for (int i = b0[0]; i <= b1[0]; ++i)
          active |= B.flags[((size_t)k * B.res[1] + j) * B.res[0] + i] != 0;
    need_clear = !active && S.cls[tile] != kTileMinus && !V.slot;     // sparse pool: a tile without a slot IS the clear value
    if (need_clear || (V.slot && !active)) S.cls[tile] = kTileMinus;
  }
  const unsigned long long am = __ballot(active);
  if (am) {
    const int leader = __ffsll((long long)am) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(S.count, (uint32_t)__popcll(am));
    base = __shfl(base, leader);
    const uint32_t pos = base + (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
    if (active) S.list[pos] = (uint32_t)tile;
    if (active && V.slot) V.slot[stored_tile_index(V, tile)] = pos < V.pool_tiles ? pos : kNoSlot;   // slot = position in this frame's list
  }
  if (V.slot && tile < S.n && !active) V.slot[stored_tile_index(V, tile)] = kNoSlot;
  unsigned long long m = __ballot(need_clear);
  const float4 cv = make_float4(-V.limit, -V.limit, -V.limit, -V.limit);
  const int wave_base = tile - lane;
  while (m) {
    const int bpos = __ffsll((long long)m) - 1;
    m &= m - 1;
    int tx, ty, tz;
    tile_coords(V, wave_base + bpos, tx, ty, tz);
    float4* __restrict__ out = (float4*)(V.data + ((((size_t)(tz - V.tz0) * V.nty + ty) * V.ntx + tx) << 9));
    out[lane] = cv;
    out[lane + 64] = cv;
  }
}

// The same classification with work that follows the SCENE instead of the volume (k_classify_clear_tiles walks every tile:
// 12 us at 512^3, 77 us at 1024^3, milliseconds at 4096^3).  Two independent parts in one launch:
//   A  blocks [0, kScatterBlocks): one lane per OCCUPIED BRICK marks the storage tiles its voxel list reaches into; the first
//      marker of a tile this frame (atomicExch of a per-tile frame stamp) appends it to the active list (and deals its slot in
//      a sparse pool);
//   B  the other blocks: one lane per tile of the PREVIOUS frame's active list -- the only tiles that can hold anything but the
//      clear value -- repeats the per-tile test; tiles that dropped out are reset by the whole wave (or lose their slot).
constexpr int kScatterBlocks = 64;
//   C  (optional) blocks past kScatterBlocks + kStaleBlocks: the peel tiles of the coming draw that the previous draw touched are
//      reset here instead of in a launch of their own (k_clear_peel_tiles, k_raymarch.hip) -- independent work, same stream
//   D  (optional) kZeroBlocks blocks zero the spare brick-counter buffer: the next clearOccupiedBricks() is then a pointer swap
constexpr int kStaleBlocks = 192, kZeroBlocks = 128;
__global__ __launch_bounds__(256) void k_classify_lists(Volume V, Bricks B, TileState S, uint32_t frame, PeelClear PC) {
  const int lane = threadIdx.x & 63;
  if (blockIdx.x >= kScatterBlocks + kStaleBlocks + kZeroBlocks) {                  // ---- part C
    if (!PC.peels) return;
    const int t = (blockIdx.x - (kScatterBlocks + kStaleBlocks + kZeroBlocks)) * 4 + (threadIdx.x >> 6);
    if (t >= PC.n_tiles || !PC.touched_prev[t]) return;
    const int px = (t % PC.ntx) * 8 + (lane & 7), py = (t / PC.ntx) * 8 + (lane >> 3);
    if (px < PC.w && py < PC.h) PC.peels[(size_t)py * PC.w + px] = make_uint4(__float_as_uint(1.0f), 0u, __float_as_uint(1.0f), 0u);   // clear (1,0,1,0)
    return;
  }
  if (blockIdx.x < kScatterBlocks) {                                               // ---- part A
    const uint32_t n_occ = *B.num_occupied;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n_occ; w += kScatterBlocks * blockDim.x) {
      const uint32_t id = B.occupied[w];
      const uint32_t plane = (uint32_t)(B.res[0] * B.res[1]);
      const int bz = (int)(id / plane), by = (int)((id % plane) / (uint32_t)B.res[0]), bx = (int)(id % (uint32_t)B.res[0]);
      const int x0 = B.brick_t0[0][bx], x1 = B.brick_t1[0][bx], y0 = B.brick_t0[1][by], y1 = B.brick_t1[1][by];
      const int z0 = max((int)B.brick_t0[2][bz], V.int_tz0), z1 = min((int)B.brick_t1[2][bz], V.int_tz1 - 1);
      for (int tz = z0; tz <= z1; ++tz)
        for (int ty = y0; ty <= y1; ++ty)
          for (int tx = x0; tx <= x1; ++tx) {
            const int tile = ((tz - V.int_tz0) * V.nty + ty) * V.ntx + tx;
            if (atomicExch(&S.stamp[tile], frame) == frame) continue;              // another brick of this frame was first
            const uint32_t pos = atomicAdd(S.count, 1u);
            S.list[pos] = (uint32_t)tile;
            if (V.slot) V.slot[stored_tile_index(V, tile)] = pos < V.pool_tiles ? pos : kNoSlot;
          }
    }
    return;
  }
  if (blockIdx.x >= kScatterBlocks + kStaleBlocks) {                                // ---- part D: zero the spare brick counters
    if (!PC.zero) return;
    uint4* __restrict__ z = (uint4*)PC.zero;
    const uint32_t quads = PC.zero_words >> 2;                                      // the buffer is padded to 64 words
    for (uint32_t i = (blockIdx.x - (kScatterBlocks + kStaleBlocks)) * blockDim.x + threadIdx.x; i < quads; i += kZeroBlocks * blockDim.x) z[i] = make_uint4(0, 0, 0, 0);
    return;
  }
  const uint32_t n_prev = *S.prev_count;                                           // ---- part B
  const uint32_t nthreads = kStaleBlocks * blockDim.x;
  for (uint32_t base = (blockIdx.x - kScatterBlocks) * blockDim.x; base < n_prev; base += nthreads) {   // block-uniform: the ballots below need whole waves
    const uint32_t i = base + threadIdx.x;
    bool stale = false;
    int tile = 0;
    if (i < n_prev) {
      tile = (int)S.prev_list[i];
      int t[3];
      tile_coords(V, tile, t[0], t[1], t[2]);
      bool active = false;
      for (int k = B.tile_b0[2][t[2]]; k <= B.tile_b1[2][t[2]]; ++k)
        for (int j = B.tile_b0[1][t[1]]; j <= B.tile_b1[1][t[1]]; ++j)
          for (int ii = B.tile_b0[0][t[0]]; ii <= B.tile_b1[0][t[0]]; ++ii)
            active |= B.flags[((size_t)k * B.res[1] + j) * B.res[0] + ii] != 0;
      stale = !active;
      if (stale) {
        S.cls[tile] = kTileMinus;
        if (V.slot) V.slot[stored_tile_index(V, tile)] = kNoSlot;                  // sparse pool: no slot IS the clear value
      }
    }
    unsigned long long m = V.slot ? 0ull : __ballot(stale);
    const float4 cv = make_float4(-V.limit, -V.limit, -V.limit, -V.limit);
    while (m) {                                                                     // dense storage: the whole wave resets the tile
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      const int tl = __shfl(tile, src);
      int tx, ty, tz;
      tile_coords(V, tl, tx, ty, tz);
      float4* __restrict__ out = (float4*)(V.data + ((((size_t)(tz - V.tz0) * V.nty + ty) * V.ntx + tx) << 9));
      out[lane] = cv;
      out[lane + 64] = cv;
    }
  }
}

// Shared tile loop: which tile does work item w map to
template <bool kList>
__device__ __forceinline__ int work_tile(const TileState& S, int w) {
  if (kList) return (int)S.list[w];
  // XCD-aware mapping: blocks b, b+8, b+16.. share an XCD (round-robin dispatch); give each XCD one
  // contiguous run of tiles so neighbouring tiles (shared LUT texels / image pixels) hit the same L2.
  return ((S.n & 7) == 0) ? (w & 7) * (S.n >> 3) + (w >> 3) : w;
}

// Generic path: every tap straight from global memory.  Used when a tile's LUT neighbourhood does not fit
// the LDS budget (inverse LUT much finer than the TSDF).
template <bool kList>
__global__ __launch_bounds__(256) void k_integrate_tiles(StreamTable T, FrameImages F, Volume V, Bricks B, TileState S, int per_voxel_check) {
  const float sx = 1.0f / (float)V.res[0], sy = 1.0f / (float)V.res[1], sz = 1.0f / (float)V.res[2];   // volume_sampler.cpp:36-38
  const float limit = V.limit;
  const int n_work = kList ? (int)*S.count : S.n;
  if (kList && blockIdx.x == 0 && threadIdx.x == 0) *S.next_count = 0u;              // the previous list was consumed by the classify launch
  for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
    const int tile = work_tile<kList>(S, w);
    int tx, ty, tz;
    tile_coords(V, tile, tx, ty, tz);
    if (V.slot && (uint32_t)w >= V.pool_tiles) continue;              // sparse pool exhausted: the tile stays unallocated (reads -limit)
    float* __restrict__ out = V.slot ? V.data + ((size_t)w << 9) : V.data + ((((size_t)(tz - V.tz0) * V.nty + ty) * V.ntx + tx) << 9);
    float v[2];
    bool in[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int l = threadIdx.x + half * 256;
      const int x = tx * 8 + (l & 7), y = ty * 8 + ((l >> 3) & 7), z = tz * 8 + (l >> 6);
      in[half] = (x < V.res[0]) && (y < V.res[1]) && (z < V.res[2]);
      bool drawn = in[half];
      if (drawn && per_voxel_check) drawn = voxel_drawn(B, x, y, z);
      v[half] = -limit;                                                 // clearImage(-limit), :249-250
      if (drawn) v[half] = integrate_voxel(T, F, limit, ((float)x + 0.5f) * sx, ((float)y + 0.5f) * sy, ((float)z + 0.5f) * sz);
      out[l] = v[half];
    }
    if (threadIdx.x == 0) S.cls[tile] = kTileMixed;
  }
}

// LDS path.  The voxel grid is axis aligned with the inverse LUT, so the 512 voxels of a tile only ever touch a
// small box of LUT texels per stream (4^3 at 512^3 over a 128^3 LUT) and the trilinear filter is separable over the
// tile.  Per tile and chunk of kChunk streams:
//   A  24 lanes per stream evaluate the per-axis GL filter set-up (i0, i1, weight) of the tile's 8 voxel coordinates
//   B  the workgroup copies each stream's texel box HBM/L2 -> LDS with one batch of independent 16-B loads
//   X  lerp along x for every (row of the box, voxel x)            dz*dy*8 lerps  (128)
//   Y  lerp along y for every (box z-plane, voxel y, voxel x)      dz*8*8  lerps  (256)
//   Z  each voxel lerps its two z-planes (512), issues the image gathers of ALL streams of the chunk together, and
//      only then runs the order-dependent fusion rule on registers
// The x -> y -> z order and every operand are those of tex3d_rgba_xyz, so the result is bit-identical, at 1.75 instead
// of 7 three-component lerps per voxel and stream, and a voxel pays two dependent memory round trips per chunk.
// Tuning knobs, values from A/B runs on MI355X (c2 / c1 integrate, ms): chunk 2, 4 waves/SIMD, 512-texel cap 0.139 / 0.509;
// + both voxels of a thread in flight 0.108 / 0.463; + 384-texel cap, 5 waves 0.096 / 0.384; chunk 1, 6 waves/SIMD
// 0.085 / 0.347; 7 waves/SIMD (72 VGPRs, no spill) 0.070 / 0.324 (chosen; 8 waves spills: 0.081 / 0.388).  The kernel is bound
// by dependent-load latency per workgroup, so occupancy wins.  A software-pipelined variant (phase A once per tile, next
// stream's box prefetched into registers, double-buffered box, fusion deferred behind the next stream's X/Y passes: three
// barriers per stream instead of five, both round trips off the critical path) was bit-identical and SLOWER (0.084 / 0.414 at
// 80 VGPRs + 23 KB LDS): the extra live state costs more occupancy than the overlap returns.
#ifndef RR_K1_BOXCAP
#define RR_K1_BOXCAP 384
#endif
#ifndef RR_K1_BOUNDS
#define RR_K1_BOUNDS 7
#endif
#ifndef RR_K1_UNROLL_H
#define RR_K1_UNROLL_H 2
#endif
constexpr int kBoxCap = RR_K1_BOXCAP;   // LUT texels per stream held in LDS; also caps the y-pass planes (dz * 64)
constexpr int kRowCap = RR_K1_BOXCAP;   // x-pass results: dz*dy rows of 8
#ifndef RR_K1_CHUNK
#define RR_K1_CHUNK 1
#endif
constexpr int kChunk = RR_K1_CHUNK;

// kThreads = 256 (shipped): two voxels per thread (z and z + 4); 128 (RR_K1_THREADS=128, experiment): four voxels per thread --
// half the workgroup, twice as many tiles in flight per CU, meant to fill the last, mostly empty round of resident workgroups.
// Measured slower (see launch_integrate): the per-tile chain gets longer by more than the extra tiles in flight return.
template <bool kList, int kThreads>
__global__ __launch_bounds__(kThreads, RR_K1_BOUNDS) void k_integrate_tiles_lds(StreamTable T, FrameImages F, Volume V, Bricks B, TileState S, int per_voxel_check) {
  __shared__ float4 s_box[kChunk][kBoxCap];     // the texel box; after the x-pass it is reused for the y-pass results
  __shared__ float4 s_row[kChunk][kRowCap];     // x-pass results
  __shared__ int s_i0[kChunk][3][8], s_i1[kChunk][3][8];
  __shared__ float s_w[kChunk][3][8];
  const float step[3] = {1.0f / (float)V.res[0], 1.0f / (float)V.res[1], 1.0f / (float)V.res[2]};       // volume_sampler.cpp:36-38
  const float limit = V.limit;
  const int n_work = kList ? (int)*S.count : S.n;
  const int tid = threadIdx.x;
  if (kList && blockIdx.x == 0 && tid == 0) *S.next_count = 0u;                    // the previous list was consumed by the classify launch
  for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
    const int tile = work_tile<kList>(S, w);
    int t3[3];
    tile_coords(V, tile, t3[0], t3[1], t3[2]);
    if (V.slot && (uint32_t)w >= V.pool_tiles) continue;              // sparse pool exhausted: the tile stays unallocated (reads -limit)
    float* __restrict__ out = V.slot ? V.data + ((size_t)w << 9) : V.data + ((((size_t)(t3[2] - V.tz0) * V.nty + t3[1]) * V.ntx + t3[0]) << 9);
    constexpr int kVox = 512 / kThreads, kZStep = kThreads / 64;        // voxels per thread; their z spacing
    // the voxels of this thread share x and y
    const int lx = tid & 7, ly = (tid >> 3) & 7, lz = tid >> 6;
    const int x = t3[0] * 8 + lx, y = t3[1] * 8 + ly;
    bool drawn[kVox];
    float tsd[kVox], wsum[kVox];
#pragma unroll
    for (int h = 0; h < kVox; ++h) {
      const int z = t3[2] * 8 + lz + kZStep * h;
      drawn[h] = (x < V.res[0]) && (y < V.res[1]) && (z < V.res[2]);
      if (drawn[h] && per_voxel_check) drawn[h] = voxel_drawn(B, x, y, z);
      tsd[h] = limit;                                                   // tsdf_integration.vs:28-29
      wsum[h] = 0.0f;
    }
    for (int cb = 0; cb < T.n; cb += kChunk) {
      const int nc = min(kChunk, T.n - cb);
      __syncthreads();                                                  // previous readers of s_* are done
      if (tid < nc * 24) {                                              // phase A
        const int c = tid / 24, a = (tid % 24) >> 3, k = tid & 7;
        const int coord = min(t3[a] * 8 + k, V.res[a] - 1);             // padding voxels reuse the last real coordinate
        const Axis ax = axis_linear(((float)coord + 0.5f) * step[a], T.s[cb + c].inv_res[a]);
        s_i0[c][a][k] = ax.i0; s_i1[c][a][k] = ax.i1; s_w[c][a][k] = ax.a;
      }
      __syncthreads();
      for (int c = 0; c < nc; ++c) {                                    // phase B
        const StreamLut& L = T.s[cb + c];
        const int mx = s_i0[c][0][0], my = s_i0[c][1][0], mz = s_i0[c][2][0];
        const int dx = s_i1[c][0][7] - mx + 1, dy = s_i1[c][1][7] - my + 1, dz = s_i1[c][2][7] - mz + 1;
        const int n = min(dx * dy * dz, kBoxCap);
        for (int e = tid; e < n; e += kThreads) {
          const int bx = e % dx, by = (e / dx) % dy, bz = e / (dx * dy);
          // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate and this launch is VALU-issue bound): operands are LUT
          // coordinates / resolutions <= 2048, the texel index fits 32 bits (tsdf_set_calibration rejects larger LUTs)
          s_box[c][e] = L.inv[(uint32_t)__mul24(__mul24(mz + bz, L.inv_res[1]) + (my + by), L.inv_res[0]) + (uint32_t)(mx + bx)];
        }
      }
      __syncthreads();
      for (int c = 0; c < nc; ++c) {                                    // pass X
        const int mx = s_i0[c][0][0];
        const int dx = s_i1[c][0][7] - mx + 1, dy = s_i1[c][1][7] - s_i0[c][1][0] + 1, dz = s_i1[c][2][7] - s_i0[c][2][0] + 1;
        const int n1 = min(dy * dz * 8, kRowCap);
        for (int e = tid; e < n1; e += kThreads) {
          const int k = e & 7, row = e >> 3;
          const int rb = __mul24(row, dx);
          const float4 a = s_box[c][rb + (s_i0[c][0][k] - mx)], b = s_box[c][rb + (s_i1[c][0][k] - mx)];
          const float3 r = lerp3(a, b, s_w[c][0][k]);
          s_row[c][e] = make_float4(r.x, r.y, r.z, 0.0f);
        }
      }
      __syncthreads();
      for (int c = 0; c < nc; ++c) {                                    // pass Y (overwrites the box)
        const int my = s_i0[c][1][0];
        const int dy = s_i1[c][1][7] - my + 1, dz = s_i1[c][2][7] - s_i0[c][2][0] + 1;
        const int n2 = min(dz * 64, kBoxCap);
        for (int e = tid; e < n2; e += kThreads) {
          const int k = e & 7, j = (e >> 3) & 7, bz = e >> 6;
          const int zb = __mul24(bz, dy);
          const float4 a = s_row[c][((zb + (s_i0[c][1][j] - my)) << 3) + k], b = s_row[c][((zb + (s_i1[c][1][j] - my)) << 3) + k];
          const float3 r = lerp3(a, b, s_w[c][1][j]);
          s_box[c][e] = make_float4(r.x, r.y, r.z, 0.0f);
        }
      }
      __syncthreads();
      bool any_drawn = false;
#pragma unroll
      for (int h = 0; h < kVox; ++h) any_drawn |= drawn[h];
      if (__ballot(any_drawn) != 0ull) {                                // pass Z + fusion
#pragma unroll RR_K1_UNROLL_H
        for (int h = 0; h < kVox; ++h) {
          const int kz = lz + kZStep * h;
          float3 pc[kChunk];
          Dqs q[kChunk];
#pragma unroll
          for (int c = 0; c < kChunk; ++c) {
            if (c < nc) {
              const int mz = s_i0[c][2][0];
              const float4 a = s_box[c][(((s_i0[c][2][kz] - mz) << 3) + ly) * 8 + lx], b = s_box[c][(((s_i1[c][2][kz] - mz) << 3) + ly) * 8 + lx];
              pc[c] = lerp3(a, b, s_w[c][2][kz]);
            }
          }
#pragma unroll
          for (int c = 0; c < kChunk; ++c)
            if (c < nc && drawn[h]) q[c] = dqs_fetch(F, cb + c, pc[c].x, pc[c].y);
          if (drawn[h]) {
            float weighted_tsd = tsd[h], total_weight = wsum[h];
#pragma unroll
            for (int c = 0; c < kChunk; ++c) {                          // tsdf_integration.vs:30-55, in stream order
              if (c < nc) {
                bool skip = false;
                if (dqs_silhouette(q[c]) < 1.0f) {
                  if (weighted_tsd >= limit) { weighted_tsd = -limit; skip = true; }
                }
                if (!skip) {
                  const float sdist = pc[c].z - dqs_depth(q[c]);
                  if (sdist <= -limit) {
                    weighted_tsd = -limit;
                  } else if (sdist >= limit) {
                  } else {
                    const float weight = dqs_quality(q[c]);
                    weighted_tsd = (weighted_tsd * total_weight + weight * sdist) / (total_weight + weight);
                    total_weight += weight;
                  }
                }
              }
            }
            tsd[h] = weighted_tsd; wsum[h] = total_weight;
          }
        }
      }
    }
#pragma unroll
    for (int h = 0; h < kVox; ++h) out[tid + kThreads * h] = drawn[h] ? tsd[h] : -limit;   // clearImage(-limit), :249-250
    if (tid == 0) S.cls[tile] = kTileMixed;
  }
}

void launch_integrate(hipStream_t st, const StreamTable& T, const FrameImages& F, const Volume& V, const Bricks& B, const TileState& S, int use_bricks, int lds_ok,
                      int full_classify, uint32_t frame_stamp, int phase, const PeelClear* pc) {
  // phase 1: tile classification + stale-tile clear; phase 2: the integrate kernel; 0: both (the split lets the caller time the kernel alone)
  if (use_bricks) {
    if (phase != 2) {
      if (full_classify) hipLaunchKernelGGL(k_classify_clear_tiles, dim3((S.n + 255) / 256), dim3(256), 0, st, V, B, S);
      else {
        PeelClear none{};
        const PeelClear& P = pc ? *pc : none;
        const int extra = P.peels ? (P.n_tiles + 3) / 4 : 0;
        hipLaunchKernelGGL(k_classify_lists, dim3(kScatterBlocks + kStaleBlocks + kZeroBlocks + extra), dim3(256), 0, st, V, B, S, frame_stamp, P);
      }
    }
    if (phase == 1) return;
    const dim3 grid(S.n < 4096 ? S.n : 4096);
    // A/B (c2 / c3 / c4 integrate, us): 256 threads 69 / 119 / 383, 128 threads 74 / 128 / 415 -- the smaller workgroup loses
    static const int wg = (getenv("RR_K1_THREADS") && atoi(getenv("RR_K1_THREADS")) == 128) ? 128 : 256;
    if (lds_ok && wg == 128) hipLaunchKernelGGL((k_integrate_tiles_lds<true, 128>), grid, dim3(128), 0, st, T, F, V, B, S, S.uniform ? 0 : 1);
    else if (lds_ok) hipLaunchKernelGGL((k_integrate_tiles_lds<true, 256>), grid, dim3(256), 0, st, T, F, V, B, S, S.uniform ? 0 : 1);
    else hipLaunchKernelGGL(k_integrate_tiles<true>, grid, dim3(256), 0, st, T, F, V, B, S, S.uniform ? 0 : 1);
  } else {
    if (phase == 1) return;
    if (lds_ok) hipLaunchKernelGGL((k_integrate_tiles_lds<false, 256>), dim3(S.n), dim3(256), 0, st, T, F, V, B, S, 0);
    else hipLaunchKernelGGL(k_integrate_tiles<false>, dim3(S.n), dim3(256), 0, st, T, F, V, B, S, 0);
  }
}
int integrate_box_cap() { return kBoxCap; }

__global__ __launch_bounds__(256) void k_fill_u8(uint8_t* __restrict__ p, uint8_t v, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}
void launch_mark_all_mixed(hipStream_t st, const TileState& S) {
  hipLaunchKernelGGL(k_fill_u8, dim3((S.n + 255) / 256 > 1024 ? 1024 : (S.n + 255) / 256), dim3(256), 0, st, S.cls, kTileMixed, S.n);
}

// ---- linear <-> tile-major conversion for the download/upload entry points
__global__ __launch_bounds__(256) void k_volume_to_linear(Volume V, float* __restrict__ lin) {
  const size_t n = (size_t)V.res[0] * V.res[1] * V.res[2];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % V.res[0]), y = (int)((i / V.res[0]) % V.res[1]), z = (int)(i / ((size_t)V.res[0] * V.res[1]));
    const int tz = z >> 3;
    lin[i] = (tz >= V.tz0 && tz < V.tz1) ? (V.slot ? tsdf_tap_sparse(V, x, y, z) : V.data[vol_index(V, x, y, z)]) : 0.0f;
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
