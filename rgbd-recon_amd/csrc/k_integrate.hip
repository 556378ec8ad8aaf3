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
// Round 4: a voxel lies in one brick per axis, two where the reference's float arithmetic lets neighbouring bricks overlap by a voxel: the (up to) eight flags are
// requested TOGETHER.  As three nested loops with a load and a wait in the innermost one the rim tiles of the occupied set paid up to eight dependent round trips
// at their head (5 us of the c2 launch: 41.7 -> 36 us without the test, a timing experiment).  More than two bricks per axis: the loops.
__device__ __forceinline__ bool voxel_drawn(const Bricks& B, int x, int y, int z) {
  const int fx = B.vox_first[0][x], nx = B.vox_count[0][x];
  const int fy = B.vox_first[1][y], ny = B.vox_count[1][y];
  const int fz = B.vox_first[2][z], nz = B.vox_count[2][z];
  bool any = false;
  if (__ballot(nx > 2 || ny > 2 || nz > 2) == 0ull) {                    // wave-uniform
    uint8_t f[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int i = c & 1, j = (c >> 1) & 1, k = c >> 2;
      const bool valid = i < nx && j < ny && k < nz;
      f[c] = B.flags[valid ? ((size_t)(fz + k) * B.res[1] + (fy + j)) * B.res[0] + (fx + i) : (size_t)0];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) any |= ((c & 1) < nx && ((c >> 1) & 1) < ny && (c >> 2) < nz) && f[c] != 0;
    return any;
  }
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i)
        any |= B.flags[((size_t)(fz + k) * B.res[1] + (fy + j)) * B.res[0] + (fx + i)] != 0;
  return any;
}

// index of integrated tile `tile` (int_tz0-relative) in the per-stored-tile tables (tz0-relative)
__device__ __forceinline__ uint32_t stored_tile_index(const Volume& V, int tile) { return (uint32_t)tile + (uint32_t)((V.int_tz0 - V.tz0) * V.nty * V.ntx); }
__device__ __forceinline__ uint32_t fast_div(uint32_t n, FastDiv f) { return (uint32_t)(((uint64_t)n * f.m) >> f.k); }
__device__ __forceinline__ void tile_coords(const Volume& V, int tile, int& tx, int& ty, int& tz) {
  const uint32_t layer = fast_div((uint32_t)tile, V.div_layer), in_layer = (uint32_t)tile - layer * (uint32_t)(V.ntx * V.nty);
  const uint32_t row = fast_div(in_layer, V.div_row);
  tx = (int)(in_layer - row * (uint32_t)V.ntx); ty = (int)row; tz = V.int_tz0 + (int)layer;
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
  // kGroup list entries per wave and round: lanes < kGroup test one tile each, then the whole wave resets the stale ones (2 x 1 KiB
  // stores per tile).  A moving scene makes EVERY previous tile stale; with 64 entries per wave only n_prev / 64 waves had work and
  // each reset 64 tiles one after another (25 us at c2); 8 per wave spread the same resets over 8 x as many waves.
  constexpr uint32_t kGroup = 8;
  const uint32_t n_waves = kStaleBlocks * (blockDim.x >> 6), wave = (blockIdx.x - kScatterBlocks) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  for (uint32_t base = wave * kGroup; base < n_prev; base += n_waves * kGroup) {   // wave-uniform: the ballot below needs the whole wave
    const uint32_t i = base + (uint32_t)lane;
    bool stale = false;
    int tile = 0;
    if (lane < (int)kGroup && i < n_prev) {
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

// What the tile's 2 KiB hold after this integrate: kTileMinus iff EVERY stored voxel is the clear value (exactly), else kTileMixed.
// The dense march leaps over whole runs of kTileMinus tiles (k_march_box, k_raymarch.hip); for the culled path an exact class only
// means that such a tile needs no reset when it stops being active.  One __syncthreads_and per tile.
__device__ __forceinline__ void store_tile_class(const TileState& S, int tile, bool mine_all_clear) {
  // one ballot per wave, one LDS flag per wave, ONE barrier (round 4; __syncthreads_and is a DPP reduction, an LDS atomic and three barriers).  The flags are
  // read by thread 0 right behind the barrier and written again one whole tile later, behind that tile's own barriers.
  __shared__ int s_wave_clear[4];
  const bool wave_clear = __ballot(!mine_all_clear) == 0ull;
  if ((threadIdx.x & 63) == 0) s_wave_clear[threadIdx.x >> 6] = wave_clear ? 1 : 0;
  __syncthreads();
  if (threadIdx.x == 0) S.cls[tile] = (s_wave_clear[0] & s_wave_clear[1] & s_wave_clear[2] & s_wave_clear[3]) ? kTileMinus : kTileMixed;
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
    store_tile_class(S, tile, v[0] == -limit && v[1] == -limit);
  }
}

// LDS path.  The voxel grid is axis aligned with the inverse LUT, so the 512 voxels of a tile only ever touch a small box of
// LUT texels per stream (3^3 .. 5^3 at the BASELINE sizes).  Per tile:
//   A  (once, all streams) 24 lanes per stream evaluate the per-axis GL filter set-up (i0, i1, weight) of the tile's 8 voxel
//      coordinates per axis
// and per stream:
//   B  the workgroup copies the stream's texel box HBM/L2 -> LDS with one batch of independent 16-B loads
//   kSep: X  lerp along x for every (box row, voxel x), Y  lerp along y for every (box plane, voxel y, voxel x) -- the filter is
//         separable over a tile -- and Z reads 2 taps and does the z lerp: 1.75 instead of 7 lerps per voxel and stream
//   else: Z reads the voxel's 8 box texels and lerps x -> y -> z itself (for LUT boxes whose rows / planes exceed s_row / s_box)
//   Z  then fetches the 2x2 image footprint and runs the order-dependent fusion rule on registers; (tsd, weight) are carried
//      across the streams in registers
// Operands and the x -> y -> z order are those of tex3d_rgba_xyz in every form: bit-identical.
// History, values from A/B runs on MI355X (c2 / c1 integrate, ms).  Separable form tuned for occupancy, because every workgroup
// is a chain of dependent round trips: 2 streams per chunk, 4 waves/SIMD 0.139 / 0.509; both voxels of a thread in flight
// 0.108 / 0.463; smaller LDS cap, 5 waves 0.096 / 0.384; 1 stream per chunk, 6 waves 0.085 / 0.347; 7 waves/SIMD (72 VGPRs)
// 0.070 / 0.324; a software-pipelined variant 0.084 / 0.414; 128-thread workgroups 0.074 / -; device code without the
// vectorizers 0.062 / 0.305.  Then the direct form (0.063 / 0.291), on it: division-free box indexing 0.0616 / 0.281, phase A
// hoisted out of the stream loop 0.0604 / 0.277, 8 waves/SIMD at 63 VGPRs 0.0580 / 0.269, hardware reciprocal 0.0577 / 0.260 --
// and with all of that in place the separable passes once more: 0.0542 / 0.2518 (the x pass alone: 0.0556 / 0.253).  Neither the
// vector pipe (60 %) nor the LDS pipe (65 %) is saturated; the passes cut the traffic of both.  (Sharing the x/y-lerped LUT
// planes between z-neighbour voxels of a thread on top of the direct form: 0.064 / 0.321.)
// ---- uniform (tile, stream) pairs (round 2, dense launches).  Most of a dense volume is free space: for most tiles and streams EVERY
// voxel takes the same branch of the fusion rule -- the tile projects onto background pixels (silhouette 0 and no depth: "carve if
// untouched"), or lies wholly in front of the measured surface (-limit), or wholly behind it (nothing) -- and the per-voxel image
// gather only confirms it.  That can be PROVEN per tile and stream from bounds, without touching a voxel:
//   * (u, v, z) of a voxel depends on the calibration volume and the voxel grid only, so its min / max over the tile's 512 voxels is a static
//     table (k_tile_bounds evaluates the very expression the integrate kernels evaluate; the rounding slack added on top of it dates from
//     rounds 2 / 3, when the table held the hull of the tile's LUT texel box, and is kept: it costs nothing);
//   * the image taps of all voxels then lie in one pixel rectangle, and k_frame_ranges (k_bricks.hip) holds min / max of depth and
//     silhouette per 8x8-pixel cell: a silhouette range of exactly {0} (or {1}) makes every bilinear silhouette exactly 0 (or 1), and a
//     depth range bounds sdist = z - depth for every voxel, nearest texel included.
// Outcome per pair: kPairFull (evaluate as before), kPairCarve (tsd >= limit ? -limit : tsd), kPairNeg (tsd = -limit), kPairNop.  Any
// NaN / non-finite value, an oversized rectangle or a mixed range fails the tests (comparisons with NaN are false) -> kPairFull.
// The classes of a launch are worked out by a pass of its own (k_pair_masks, below) and read by the integrate kernels with one scalar
// load per work item.
constexpr int kPairFull = 0, kPairCarve = 1, kPairNeg = 2, kPairNop = 3;
// Per (tile, stream) the (u, v, z) range over the tile's voxels -- static: it depends on the calibration volume and the voxel
// grid only -- is computed once (k_tile_bounds, below) as two float4 {u0, u1, v0, v1}, {z0, z1, -, -}; NaN marks a box with a
// non-finite or far-away texel.  The class of (tile, stream) for THIS frame is worked out by a half wave (two streams per call, up to
// 32 range cells each: a lane loads one cell of the rectangle, DPP row shifts and one row broadcast reduce them).
__device__ __forceinline__ float half_min_f32(float v) {
#define RR_DPP_MIN(ctrl, rm, bm) v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rm, bm, false)))
  RR_DPP_MIN(0x111, 0xf, 0xf); RR_DPP_MIN(0x112, 0xf, 0xf); RR_DPP_MIN(0x114, 0xf, 0xe); RR_DPP_MIN(0x118, 0xf, 0xc);
  RR_DPP_MIN(0x142, 0xa, 0xf);                                          // row_bcast:15 -> lane 31 holds rows 0-1, lane 63 rows 2-3
#undef RR_DPP_MIN
  const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31)), hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  return (threadIdx.x & 32) ? hi : lo;
}
__device__ __forceinline__ float half_max_f32(float v) { return -half_min_f32(-v); }
// in two steps, so that a caller can have the cell loads of several pairs in flight before the first reduction: (1) the texel rectangle of
// the pair's bilinear footprints and this lane's range cell -- loaded unconditionally (index 0 for lanes outside the rectangle) --,
// (2) the reductions and the class
struct PairCell { float4 q; float z0, z1; bool ok, mine; };
__device__ __forceinline__ PairCell pair_cell(const float4* __restrict__ ranges, int rcw, int rch, int img_w, int img_h, int i, bool valid, float4 b0, float4 b1) {
  const int r = threadIdx.x & 31;
  const float u0 = b0.x, u1 = b0.y, v0 = b0.z, v1 = b0.w;
  PairCell c;
  c.z0 = b1.x; c.z1 = b1.y;
  c.ok = valid && (u0 <= u1);                                          // NaN: the box holds something non-finite
  const float su = 1.0e-5f * (1.0f + fmaxf(fabsf(u0), fabsf(u1))), sv = 1.0e-5f * (1.0f + fmaxf(fabsf(v0), fabsf(v1)));
  const float wf = (float)img_w, hf = (float)img_h;
  const int x0 = (int)__builtin_amdgcn_fmed3f(floorf((u0 - su) * wf - 0.5f), 0.0f, wf - 1.0f), x1 = (int)__builtin_amdgcn_fmed3f(floorf((u1 + su) * wf - 0.5f) + 1.0f, 0.0f, wf - 1.0f);
  const int y0 = (int)__builtin_amdgcn_fmed3f(floorf((v0 - sv) * hf - 0.5f), 0.0f, hf - 1.0f), y1 = (int)__builtin_amdgcn_fmed3f(floorf((v1 + sv) * hf - 0.5f) + 1.0f, 0.0f, hf - 1.0f);
  const int cx0 = x0 >> 3, cy0 = y0 >> 3, cw = (x1 >> 3) - cx0 + 1, chh = (y1 >> 3) - cy0 + 1;
  c.ok = c.ok && cw >= 1 && chh >= 1 && __mul24(cw, chh) <= 32;
  c.mine = c.ok && r < __mul24(cw, chh);
  const int ry = (int)(((float)r + 0.5f) * __builtin_amdgcn_rcpf((float)cw));           // r / cw
  const size_t idx = c.mine ? (size_t)__mul24(__mul24(i, rch) + cy0 + ry, rcw) + (cx0 + r - __mul24(ry, cw)) : 0;
  c.q = ranges[idx];
  return c;
}
__device__ __forceinline__ int pair_class(const PairCell& c, float limit) {
  const float inf = __builtin_inff();
  float d0 = c.mine ? c.q.x : inf, d1 = c.mine ? c.q.y : -inf, s0 = c.mine ? c.q.z : inf, s1 = c.mine ? c.q.w : -inf;
  d0 = half_min_f32(d0); d1 = half_max_f32(d1); s0 = half_min_f32(s0); s1 = half_max_f32(s1);
  const float sz = 1.0e-5f * (1.0f + fmaxf(fabsf(c.z0), fabsf(c.z1)) + fmaxf(fabsf(d0), fabsf(d1)));   // lerp slack of z + the rounding of z - depth
  const bool all_le = (c.z1 + sz) - d0 <= -limit - sz, all_ge = (c.z0 - sz) - d1 >= limit + sz;      // (false for NaN / infinite ranges)
  const bool sil0 = s0 == 0.0f && s1 == 0.0f, sil1 = s0 == 1.0f && s1 == 1.0f;
  if (!c.ok) return kPairFull;
  if (sil0 && all_ge) return kPairCarve;                               // silhouette < 1: tsd >= limit -> -limit; otherwise sdist >= limit: nothing
  if ((sil0 || sil1) && all_le) return kPairNeg;                       // carved by the silhouette rule or by sdist <= -limit: -limit either way
  if (sil1 && all_ge) return kPairNop;                                 // behind the surface: nothing
  return kPairFull;
}
// the static half: one wave per (stored tile, stream) reduces min / max of (u, v, z) = texture(cv_xyz_inv[i], voxel centre).xyz over the tile's 512
// voxels -- the very values the integrate kernels compute (same coordinates, same taps, same x -> y -> z lerps: tex3d_rgba_xyz), so the bounds are
// tight (round 4).  Rounds 2 / 3 took the hull of the tile's LUT texel box instead: up to 1.5 x wider per axis at a 4:1 LUT, which left fewer uniform
// pairs (the dense c1 launch 118.5 -> 104.2 us with the tight bounds, c2 43.8 -> 42.5, c4 194.5 -> 188.0: profiles/r04_k1_rect_negative.txt, A/B 3).
// Built once per calibration (8 trilinear fetches per lane and stream; c2: 1 M wave-items).
__global__ __launch_bounds__(256) void k_tile_bounds(StreamTable T, Volume V, float4* __restrict__ bounds, int n_tiles) {
  const int ln = threadIdx.x & 63;
  const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (long long)n_tiles * T.n) return;
  const int tile = (int)(item / T.n), i = (int)(item % T.n);
  const int t3[3] = {tile % V.ntx, (tile / V.ntx) % V.nty, V.tz0 + tile / (V.ntx * V.nty)};
  const StreamLut& L = T.s[i];
  const float sx = 1.0f / (float)V.res[0], sy = 1.0f / (float)V.res[1], sz = 1.0f / (float)V.res[2];   // volume_sampler.cpp:36-38
  // padding voxels reuse the last real coordinate, as in phase A of the integrate kernels
  const float px = ((float)min(t3[0] * 8 + (ln & 7), V.res[0] - 1) + 0.5f) * sx, py = ((float)min(t3[1] * 8 + (ln >> 3), V.res[1] - 1) + 0.5f) * sy;
  const float inf = __builtin_inff();
  float u0 = inf, u1 = -inf, v0 = inf, v1 = -inf, z0 = inf, z1 = -inf;
  bool bad = false;
  for (int kz = 0; kz < 8; ++kz) {
    const float pz = ((float)min(t3[2] * 8 + kz, V.res[2] - 1) + 0.5f) * sz;
    const float3 t = tex3d_rgba_xyz(L.inv, L.inv_res, px, py, pz);
    bad |= !(fabsf(t.x) < 1.0e4f && fabsf(t.y) < 1.0e4f && fabsf(t.z) < 1.0e4f);        // NaN, infinities, far-away garbage
    u0 = fminf(u0, t.x); u1 = fmaxf(u1, t.x); v0 = fminf(v0, t.y); v1 = fmaxf(v1, t.y); z0 = fminf(z0, t.z); z1 = fmaxf(z1, t.z);
  }
  const bool any_bad = __ballot(bad) != 0ull;
  u0 = wave_min_f32(u0); u1 = wave_max_f32(u1); v0 = wave_min_f32(v0); v1 = wave_max_f32(v1); z0 = wave_min_f32(z0); z1 = wave_max_f32(z1);
  if (ln == 0) {
    const float q = __builtin_nanf("");
    bounds[2 * item] = any_bad ? make_float4(q, q, q, q) : make_float4(u0, u1, v0, v1);
    bounds[2 * item + 1] = any_bad ? make_float4(q, q, q, q) : make_float4(z0, z1, 0.0f, 0.0f);
  }
}
void launch_tile_bounds(hipStream_t st, const StreamTable& T, const Volume& V, float4* bounds) {
  const long long items = (long long)V.n_stored_tiles * T.n;
  hipLaunchKernelGGL(k_tile_bounds, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, T, V, bounds, V.n_stored_tiles);
}

// The per-frame half of the uniform-pair shortcut as a pass of its own: one wave per work item (list entry or tile) classifies the item's
// (tile, stream) pairs two streams at a time and packs the classes, 2 bits per stream, into masks[work item]; the integrate kernels
// read their item's word with one scalar load.  (Classifying in the integrate kernel's prologue put two dependent global round trips
// in front of every tile's chain: c2 53.2 -> 51.8 us; as a pass of its own the shortcut is worth what it skips.)
#ifdef RR_PAIR_STATS       // instrumented build (tools/pair_stats.py): [work items, pairs, uniform pairs, items with every pair uniform]
__device__ unsigned long long g_pair_stats[4];
extern "C" int32_t tsdf_debug_pair_stats(unsigned long long out[4], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pair_stats), sizeof(g_pair_stats)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[4] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_pair_stats), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
template <bool kList>
__global__ __launch_bounds__(256) void k_pair_masks(StreamTable T, FrameImages F, Volume V, Bricks B, TileState S, int per_voxel_check,
                                                    const float4* __restrict__ tile_bounds, uint32_t* __restrict__ masks, uint4* __restrict__ recs, ProjCache PC,
                                                    unsigned long long* __restrict__ vmasks) {
  const int ln = threadIdx.x & 63;
  const int n_work = kList ? (int)*S.count : S.n;
  const int n_waves = gridDim.x * 4;
  if (PC.items && blockIdx.x == 0 && threadIdx.x == 0) *PC.n_slow_next = 0u;   // (last read by the LDS kernel of the previous frame)
  for (int w = blockIdx.x * 4 + (threadIdx.x >> 6); w < n_work; w += n_waves) {
    const int tile = work_tile<kList>(S, w);
    const uint32_t st = stored_tile_index(V, tile);
    uint32_t pairs = 0;
    for (int cb = 0; cb < T.n; cb += 4) {                               // four streams per round: the two half-wave pairs' loads in flight together
      PairCell c[2];
      float4 b0[2], b1[2];
      int si[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        si[u] = cb + 2 * u + (ln >> 5);
        const size_t o = 2 * ((size_t)st * T.n + (si[u] < T.n ? si[u] : 0));
        b0[u] = tile_bounds[o]; b1[u] = tile_bounds[o + 1];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) c[u] = pair_cell(F.ranges, F.rcw, F.rch, F.w, F.h, si[u], si[u] < T.n, b0[u], b1[u]);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pair = pair_class(c[u], V.limit), s0 = cb + 2 * u;
        if (s0 < T.n) pairs |= (uint32_t)__builtin_amdgcn_readlane(pair, 0) << (2 * s0);
        if (s0 + 1 < T.n) pairs |= (uint32_t)__builtin_amdgcn_readlane(pair, 32) << (2 * s0 + 2);
      }
    }
    // bit 31: where bricks and tiles do not coincide the integrate kernel asks per voxel "is it in the list of an occupied brick" (two
    // dependent global round trips at the head of the tile's chain, 7 of c2's 51 us).  When every voxel coordinate of the tile is listed
    // and EVERY brick that reaches into the tile is occupied -- the usual case away from the rim of the occupied set -- the answer is yes
    // for the whole tile, and this pass can say so: one flag per lane of the tile's (at most 64) bricks
    if (per_voxel_check) {
      int t3[3];
      tile_coords(V, tile, t3[0], t3[1], t3[2]);
      const int b0x = B.tile_b0[0][t3[0]], nbx = (int)B.tile_b1[0][t3[0]] - b0x + 1, b0y = B.tile_b0[1][t3[1]], nby = (int)B.tile_b1[1][t3[1]] - b0y + 1;
      const int b0z = B.tile_b0[2][t3[2]], nbz = (int)B.tile_b1[2][t3[2]] - b0z + 1;
      const bool listed = B.tile_full[0][t3[0]] && B.tile_full[1][t3[1]] && B.tile_full[2][t3[2]];
      // (vmasks) this lane's voxel column and -- lanes 0 .. 7 -- the tile's eight layers in the per-axis voxel -> brick tables: requested with the loads above,
      // used below when the tile's voxels have to be answered one by one
      const int xr = t3[0] * 8 + (ln & 7), yr = t3[1] * 8 + (ln >> 3);
      const int x = min(xr, V.res[0] - 1), y = min(yr, V.res[1] - 1), zl = min(t3[2] * 8 + (ln & 7), V.res[2] - 1);
      int fx = 0, nx = 0, fy = 0, ny = 0, fz_l = 0, nz_l = 0;
      if (vmasks) { fx = B.vox_first[0][x]; nx = B.vox_count[0][x]; fy = B.vox_first[1][y]; ny = B.vox_count[1][y]; fz_l = B.vox_first[2][zl]; nz_l = B.vox_count[2][zl]; }
      const int nb = (nbx > 0 && nby > 0 && nbz > 0) ? __mul24(__mul24(nbx, nby), nbz) : 0;
      const bool few = nb > 0 && nb <= 64;                              // (wave-uniform) one lane per brick that reaches into the tile
      int occ = 1;
      if (few && ln < nb) {
        const int bz = ln / __mul24(nbx, nby), rem = ln - __mul24(bz, __mul24(nbx, nby)), by = rem / nbx, bx = rem - __mul24(by, nbx);
        occ = B.flags[((size_t)(b0z + bz) * B.res[1] + (b0y + by)) * B.res[0] + (b0x + bx)] != 0 ? 1 : 0;
      }
      const bool all = listed && few && __ballot(occ == 0) == 0ull;
      if (all && T.n <= 15) pairs |= 0x80000000u;                       // (16 streams use all 32 bits for their classes)
      else if (vmasks) {
        // Round 4: the per-voxel answers themselves, here -- 8 x 64 bits per tile (bit y * 8 + x of word z: "inside the volume and in an occupied brick's list") --
        // for the tiles the integrate kernel would otherwise test voxel by voxel at the head of their chain (a timing experiment without the test: 41.7 -> 36.3 us
        // at c2).  The flags of the (at most 64) bricks that reach into the tile are in the lanes already: as ONE 64-bit word (brick (bx, by, bz) of the tile's span =
        // bit (bz * nby + by) * nbx + bx), a voxel's candidates in x and y as a 64-bit pattern per lane, the layer's candidates in z as a shift -- no further flag
        // loads, the per-axis voxel -> brick tables are this pass's only additional ones.
        const bool xy_in = xr < V.res[0] && yr < V.res[1];
        if (few) {
          const unsigned long long occ_bits = __ballot(occ != 0 && ln < nb);
          unsigned long long mxy = 0ull;                                  // this voxel column's candidate bricks within one brick layer
          for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) mxy |= 1ull << (__mul24(fy + j - b0y, nbx) + (fx + i - b0x));
          const int plane = __mul24(nbx, nby);
          for (int z8 = 0; z8 < 8; ++z8) {
            const int z = t3[2] * 8 + z8;
            const int fz = __builtin_amdgcn_readlane(fz_l, z8), nz = __builtin_amdgcn_readlane(nz_l, z8);
            unsigned long long layer = 0ull;                              // (wave-uniform) the occupied bricks of the layers this z is listed in, moved to layer 0
            for (int k = 0; k < nz; ++k) layer |= occ_bits >> __mul24(fz + k - b0z, plane);
            const bool dr = xy_in && z < V.res[2] && (layer & mxy) != 0ull;
            const unsigned long long m = __ballot(dr);
            if (ln == 0) vmasks[(size_t)w * 8 + z8] = m;
          }
        } else {
          for (int z8 = 0; z8 < 8; ++z8) {                                // more than 64 bricks reach into the tile: flag by flag
            const int z = t3[2] * 8 + z8;
            bool dr = false;
            if (xy_in && z < V.res[2]) dr = voxel_drawn(B, x, y, z);
            const unsigned long long m = __ballot(dr);
            if (ln == 0) vmasks[(size_t)w * 8 + z8] = m;
          }
        }
      }
    }
    if (ln == 0) masks[w] = pairs;
    if (recs && ln == 0) {                                              // the work item's record (k_integrate_tiles_rec): everything its tile head needs in one 16-byte load
      int c3[3];
      tile_coords(V, tile, c3[0], c3[1], c3[2]);
      const uint32_t interior = (c3[0] * 8 + 7 < V.res[0] && c3[1] * 8 + 7 < V.res[1] && c3[2] * 8 + 7 < V.res[2]) ? 1u : 0u;
      recs[w] = make_uint4(pairs, (uint32_t)tile, (uint32_t)c3[0] | ((uint32_t)c3[1] << 10) | ((uint32_t)c3[2] << 20) | (interior << 30), st);
    }
    // projection cache: is the tile's (u, v, z) table in the pool?  No: deal it a slot (the LDS kernel fills it while it integrates the
    // tile, this frame) -- or, pool exhausted, leave the tile to the LDS kernel for good
    if (PC.items && ln == 0) {
      uint32_t it = PC.slot[st];
      if (it == kNoSlot) {
        it = kItemNone;
        if (*(volatile uint32_t*)PC.alloc < PC.cap) {                    // (saturating: an exhausted pool costs no atomic per tile and frame)
          const uint32_t a = atomicAdd(PC.alloc, 1u);
          if (a < PC.cap) { PC.slot[st] = a; it = kItemFresh | a; }
        }
        atomicAdd(PC.n_slow, 1u);
      }
      PC.items[w] = it;
    }
#ifdef RR_PAIR_STATS
    if (ln == 0) {
      int u = 0;
      for (int i = 0; i < T.n; ++i) u += ((pairs >> (2 * i)) & 3u) != 0u;
      if (pairs >> 31) atomicAdd(&g_pair_stats[3], 1ull << 32);       // (high half of [3]: tiles whose bricks are all occupied)
      atomicAdd(&g_pair_stats[0], 1ull); atomicAdd(&g_pair_stats[1], (unsigned long long)T.n); atomicAdd(&g_pair_stats[2], (unsigned long long)u);
      if (u == T.n) atomicAdd(&g_pair_stats[3], 1ull);
    }
#endif
  }
}

#ifdef RR_K1_TRACE          // instrumented build (tools/k1_phase_trace.py): s_memtime stamps of the first tile of every workgroup of the culled LDS launch
__device__ unsigned long long g_k1_trace[4096 * 16];
extern "C" int32_t tsdf_debug_k1_trace(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k1_trace), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#define RR_STAMP(slot) do { if (kList && kRanges && !kCache) { __builtin_amdgcn_sched_barrier(0); unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); \
  __builtin_amdgcn_sched_barrier(0); if (tid == 0 && blockIdx.x < 4096 && (slot) < 16) g_k1_trace[blockIdx.x * 16 + (slot)] = _t; } } while (0)
#else
#define RR_STAMP(slot) do { } while (0)
#endif
#ifndef RR_K1_BOXCAP
#define RR_K1_BOXCAP 384
#endif
#ifndef RR_K1_BOUNDS
#define RR_K1_BOUNDS 8      // waves per SIMD: the direct form needs 63 VGPRs (no spill); 7 waves: c2 60.4 / c1 276.8 us, 8 waves: 58.0 / 268.8 us
#endif
constexpr int kBoxCap = RR_K1_BOXCAP;   // LUT texels per stream held in LDS
constexpr int kRowCap = 512;           // (separable form) x-pass results: dz * dy rows of 8
static_assert(kBoxCap <= 1024, "phase B's division-free index decomposition is exact below 1024 only");

// kCache: the launch shares its work items with k_integrate_cached (below): items whose projection is cached are skipped here, and an
// item that was dealt a fresh slot evaluates EVERY stream per voxel (no uniform-pair shortcut) and writes each voxel's (u, v, z) through
// to the slot -- the bits the cached kernel reads from the next frame on are the bits this kernel used
template <bool kList, bool kSep, bool kRanges = false, bool kCache = false>
__global__ __launch_bounds__(256, RR_K1_BOUNDS) void k_integrate_tiles_lds(StreamTable T, FrameImages F, Volume V, Bricks B, TileState S, int per_voxel_check,
                                                                             const uint32_t* __restrict__ pair_masks, ProjCache PC) {
  __shared__ float4 s_box[kBoxCap];             // the stream's texel box, x fastest: ((z - mz) * dy + (y - my)) * dx + (x - mx)
  __shared__ float4 s_row[kSep ? kRowCap : 1];  // (separable form) x-lerped rows: ((z - mz) * dy + (y - my)) * 8 + voxel x
  __shared__ int s_i0a[TSDF_MAX_STREAMS][3][8], s_i1a[TSDF_MAX_STREAMS][3][8];   // per stream, axis and voxel coordinate of the tile: the two texel indices ...
  __shared__ float s_wa[TSDF_MAX_STREAMS][3][8];                                 // ... and the weight of the GL LINEAR filter
  const float step[3] = {1.0f / (float)V.res[0], 1.0f / (float)V.res[1], 1.0f / (float)V.res[2]};       // volume_sampler.cpp:36-38
  const float limit = V.limit;
  const int n_work = kList ? (int)*S.count : S.n;
  const int tid = threadIdx.x;
  if (kList && blockIdx.x == 0 && tid == 0) *S.next_count = 0u;                    // the previous list was consumed by the classify launch
  if (kCache && *PC.n_slow == 0u) return;                                          // steady state: every work item is cached
  // INVARIANT of the tile loop: every exit from an iteration after phase A has touched s_* passes through store_tile_class(), whose
  // workgroup barrier is what lets the next iteration overwrite s_* (there is no barrier at the head of the loop).  The `continue`s
  // below are workgroup-uniform and sit in front of phase A.
  // Round 4 (in-kernel stamps, profiles/r04_k1_phase_trace_*.txt): 29 % of a tile's 24 k cycles went by BEFORE its first stream -- the list entry (a
  // dependent scalar load), then phase A's per-lane loads of the LUT resolutions out of the kernel arguments (a vector-memory round trip per tile for
  // values that never change).  Now: a thread's phase-A assignment (stream, axis, coordinate) is fixed by its index, so its LUT resolution is loaded
  // once per workgroup (into LDS); the next tile's list entry and pair classes are requested at the head of the current tile.
  __shared__ int s_invres[TSDF_MAX_STREAMS][3], s_vres[3];                         // LUT / volume resolutions: phase A reads them per (stream, axis) -- from LDS, once per workgroup
  __shared__ float s_step[3];
  if (tid < T.n * 3) s_invres[tid / 3][tid % 3] = T.s[tid / 3].inv_res[tid % 3];
  if (tid < 3) { s_vres[tid] = tid == 0 ? V.res[0] : (tid == 1 ? V.res[1] : V.res[2]); s_step[tid] = tid == 0 ? step[0] : (tid == 1 ? step[1] : step[2]); }
  __syncthreads();
  int tile_next = blockIdx.x < n_work ? work_tile<kList>(S, blockIdx.x) : 0;
  uint32_t pairs_next = kRanges && blockIdx.x < n_work ? pair_masks[blockIdx.x] : 0u;
  for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
    RR_STAMP(0);                                                                     // tile head
    [[maybe_unused]] int _stream_slot = 3;
    uint32_t item = kItemNone;
    if (kCache) { item = PC.items[w]; if (item < kItemFresh) continue; }           // cached: k_integrate_cached's
    const bool fill = kCache && item != kItemNone;                                   // workgroup-uniform
    float* __restrict__ const fill_base = fill ? PC.data + (size_t)(item & ~kItemFresh) * PC.slot_floats : nullptr;
    const int tile = kCache ? work_tile<kList>(S, w) : tile_next;
    uint32_t pairs = kRanges ? (kCache ? pair_masks[w] : pairs_next) : 0u;         // (kRanges) from k_pair_masks: 2 bits per stream -- which streams treat every voxel of this tile alike --, bit 31: every brick reaching into the tile is occupied
    {                                                                                // the next tile's, in flight across this one (scalar loads: the indices are workgroup-uniform)
      const int wn = w + (int)gridDim.x;
      if (!kCache && wn < n_work) { tile_next = work_tile<kList>(S, wn); if (kRanges) pairs_next = pair_masks[wn]; }
    }
    int t3[3];
    tile_coords(V, tile, t3[0], t3[1], t3[2]);
    if (V.slot && (uint32_t)w >= V.pool_tiles) continue;              // sparse pool exhausted: the tile stays unallocated (reads -limit)
    float* __restrict__ out = V.slot ? V.data + ((size_t)w << 9) : V.data + ((((size_t)(t3[2] - V.tz0) * V.nty + t3[1]) * V.ntx + t3[0]) << 9);
    constexpr int kVox = 2;                                             // voxels per thread: local z = lz and lz + 4, same x and y
    const int lx = tid & 7, ly = (tid >> 3) & 7, lz = tid >> 6;
    const int x = t3[0] * 8 + lx, y = t3[1] * 8 + ly;
    bool drawn[kVox];
    float tsd[kVox], wsum[kVox];
    if (fill) pairs &= T.n <= 15 ? 0x80000000u : 0u;                    // a slot is filled for all streams: no shortcut this once
    const bool check_voxels = per_voxel_check && !(kRanges && T.n <= 15 && (pairs >> 31));
#pragma unroll
    for (int h = 0; h < kVox; ++h) {
      const int z = t3[2] * 8 + lz + 4 * h;
      drawn[h] = (x < V.res[0]) && (y < V.res[1]) && (z < V.res[2]);
      if (drawn[h] && check_voxels) drawn[h] = voxel_drawn(B, x, y, z);
      tsd[h] = limit;                                                   // tsdf_integration.vs:28-29
      wsum[h] = 0.0f;
    }
    // (the previous tile's readers of s_* are done: store_tile_class() at its end is a workgroup barrier)                // (kRanges) 2 bits per stream, from k_pair_masks: which streams treat every voxel of this tile alike
    for (int t = tid; t < T.n * 24; t += 256) {                         // phase A, all streams at once
      const int i = t / 24, a = (t % 24) >> 3, k = t & 7;
      const int ta = a == 0 ? t3[0] : (a == 1 ? t3[1] : t3[2]);
      const int coord = min(ta * 8 + k, s_vres[a] - 1);                 // padding voxels reuse the last real coordinate
      const Axis ax = axis_linear(((float)coord + 0.5f) * s_step[a], s_invres[i][a]);
      s_i0a[i][a][k] = ax.i0; s_i1a[i][a][k] = ax.i1; s_wa[i][a][k] = ax.a;
    }
    RR_STAMP(1);                                                                     // own part of phase A done (incl. the voxel checks)
    __syncthreads();
    RR_STAMP(2);                                                                     // phase A barrier passed
    for (int i = 0; i < T.n; ++i) {
      if (kRanges) {
        const int pair = (int)((pairs >> (2 * i)) & 3u);                // workgroup-uniform
        if (pair != kPairFull) {                                        // no box, no passes, no gathers: the branch is the same for every voxel
#pragma unroll
          for (int h = 0; h < kVox; ++h)
            if (drawn[h]) {
              if (pair == kPairNeg) tsd[h] = -limit;
              else if (pair == kPairCarve && tsd[h] >= limit) tsd[h] = -limit;
            }
          continue;
        }
      }
      const StreamLut& L = T.s[i];
      const int (*s_i0)[8] = s_i0a[i];
      const int (*s_i1)[8] = s_i1a[i];
      const float (*s_w)[8] = s_wa[i];
      if (i) __syncthreads();                                           // the previous stream's readers of s_box are done
      const int mx = s_i0[0][0], my = s_i0[1][0], mz = s_i0[2][0];
      const int dx = s_i1[0][7] - mx + 1, dy = s_i1[1][7] - my + 1, dz = s_i1[2][7] - mz + 1;
      {                                                                 // phase B
        const int n = min(__mul24(__mul24(dx, dy), dz), kBoxCap);
        // e -> (bx, by, bz) without integer division (three of them are ~45 VALU instructions, and VALU issue is this launch's largest cost):
        // floor((e + .5) * (1 / d)) in fp32 equals e / d for all 0 <= e < 1024, 1 <= d <= 1024 (checked exhaustively; the quotient is at
        // least 0.5 / d away from the next integer, four orders of magnitude more than the 1 ulp of the hardware reciprocal)
        const float rdx = __builtin_amdgcn_rcpf((float)dx), rdy = __builtin_amdgcn_rcpf((float)dy);
        for (int e = tid; e < n; e += 256) {
          const int row = (int)(((float)e + 0.5f) * rdx);               // e / dx
          const int bz = (int)(((float)row + 0.5f) * rdy);              // row / dy
          const int bx = e - __mul24(row, dx), by = row - __mul24(bz, dy);
          // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): operands are LUT
          // coordinates / resolutions <= 2048, the texel index fits 32 bits (tsdf_set_calibration rejects larger LUTs)
          s_box[e] = L.inv[(uint32_t)__mul24(__mul24(mz + bz, L.inv_res[1]) + (my + by), L.inv_res[0]) + (uint32_t)(mx + bx)];
        }
      }
      RR_STAMP(_stream_slot);                                                        // own box texels stored
      __syncthreads();
      RR_STAMP(_stream_slot + 1);                                                    // phase B barrier passed
      if (kSep) {                                                       // passes X and Y
        const int n1 = min(__mul24(__mul24(dy, dz), 8), kRowCap);
        for (int e = tid; e < n1; e += 256) {
          const int k = e & 7, rb = __mul24(e >> 3, dx);
          const float3 r = lerp3(s_box[rb + (s_i0[0][k] - mx)], s_box[rb + (s_i1[0][k] - mx)], s_w[0][k]);
          s_row[e] = make_float4(r.x, r.y, r.z, 0.0f);
        }
        __syncthreads();
        const int n2 = min(dz << 6, kBoxCap);
        for (int e = tid; e < n2; e += 256) {                           // the y-lerped planes overwrite the box
          const int k = e & 7, j = (e >> 3) & 7, zb = __mul24(e >> 6, dy);
          const float3 r = lerp3(s_row[((zb + (s_i0[1][j] - my)) << 3) + k], s_row[((zb + (s_i1[1][j] - my)) << 3) + k], s_w[1][j]);
          s_box[e] = make_float4(r.x, r.y, r.z, 0.0f);
          if (fill) {                                                   // write-through: plane e >> 6 of stream i, voxel column e & 63
            float* __restrict__ const o = fill_base + (uint32_t)(__mul24(__mul24(i, (int)PC.dz_max), 64) + e) * 3u;
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
          }
        }
        __syncthreads();
      }
      RR_STAMP(_stream_slot + 2);                                                    // passes X and Y done
      bool any_drawn = false;
#pragma unroll
      for (int h = 0; h < kVox; ++h) any_drawn |= drawn[h];
      if (__ballot(any_drawn) != 0ull) {                                // phase Z
        const int x0 = s_i0[0][lx] - mx, x1 = s_i1[0][lx] - mx;
        const int y0 = __mul24(s_i0[1][ly] - my, dx), y1 = __mul24(s_i1[1][ly] - my, dx);
        const int pl = __mul24(dx, dy);
        const float wx = s_w[0][lx], wy = s_w[1][ly];
#pragma unroll
        for (int h = 0; h < kVox; ++h) {
          const int kz = lz + 4 * h;
          const int z0 = __mul24(s_i0[2][kz] - mz, pl), z1 = __mul24(s_i1[2][kz] - mz, pl);
          float3 pc;                                                    // texture(cv_xyz_inv[i], position).xyz, :31
          if (kSep) {
            pc = lerp3(s_box[(((s_i0[2][kz] - mz) << 3) + ly) * 8 + lx], s_box[(((s_i1[2][kz] - mz) << 3) + ly) * 8 + lx], s_w[2][kz]);
          } else {
            const float3 c00 = lerp3(s_box[z0 + y0 + x0], s_box[z0 + y0 + x1], wx), c10 = lerp3(s_box[z0 + y1 + x0], s_box[z0 + y1 + x1], wx);
            const float3 c01 = lerp3(s_box[z1 + y0 + x0], s_box[z1 + y0 + x1], wx), c11 = lerp3(s_box[z1 + y1 + x0], s_box[z1 + y1 + x1], wx);
            pc = lerp3(lerp3(c00, c10, wy), lerp3(c01, c11, wy), s_w[2][kz]);
          }
          // the gather and the fusion rule are two branches on purpose: in one branch the compiler keeps voxel 1's loads behind
          // voxel 0's arithmetic; apart, the unrolled loop has both voxels' gathers in flight together
          Dqs q;
          if (drawn[h]) q = dqs_fetch(F, i, pc.x, pc.y);
          if (drawn[h]) {
            float weighted_tsd = tsd[h], total_weight = wsum[h];        // tsdf_integration.vs:30-55, in stream order
            bool skip = false;
            if (dqs_silhouette(q) < 1.0f) {
              if (weighted_tsd >= limit) { weighted_tsd = -limit; skip = true; }
            }
            if (!skip) {
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
            tsd[h] = weighted_tsd; wsum[h] = total_weight;
          }
        }
      }
      RR_STAMP(_stream_slot + 3);                                                    // phase Z done (gathers + fusion)
      _stream_slot += 4;
    }
    RR_STAMP(13);
#pragma unroll
    for (int h = 0; h < kVox; ++h) { tsd[h] = drawn[h] ? tsd[h] : -limit; out[tid + 256 * h] = tsd[h]; }   // clearImage(-limit), :249-250
    RR_STAMP(14);
    store_tile_class(S, tile, tsd[0] == -limit && tsd[1] == -limit);
    RR_STAMP(15);
  }
}


// Record form (round 4): the LDS form for launches with pair classes, dense storage and no projection cache -- every BASELINE configuration --, with the
// tile head the in-kernel stamps asked for (profiles/r04_k1_phase_trace_*.txt: a quarter of a tile's cycles went by before its first stream, in a chain of
// scalar loads -- list entry, pair classes, a dozen kernel-argument reloads forced by SGPR pressure -- each waited for at once).  k_pair_masks leaves a
// 16-byte record per work item and the head is one load, requested a tile ahead; the per-voxel "is it drawn" test is a template parameter (off where
// bricks and tiles coincide: no brick tables in the hot variant).
// Wave priority by phase (A/B hook, tools/build_k1_variant.sh prioN "-DRR_K1_PRIO=N"): does a tile's head starve behind the other waves' long VALU runs?
#ifdef RR_K1_PRIO
#define RR_PRIO(site) do { constexpr int _v = RR_K1_PRIO, _s = (site); \
  if (_v == 1) { if (_s == 0 || _s == 7) __builtin_amdgcn_s_setprio(3); else if (_s == 3) __builtin_amdgcn_s_setprio(0); } \
  else if (_v == 2) { if (_s == 0) __builtin_amdgcn_s_setprio(3); else if (_s == 1) __builtin_amdgcn_s_setprio(0); } \
  else if (_v == 3) { if (_s == 4) __builtin_amdgcn_s_setprio(0); else if (_s == 6 || _s == 0) __builtin_amdgcn_s_setprio(3); } \
  else if (_v == 4) { if (_s == 4) __builtin_amdgcn_s_setprio(3); else if (_s == 6 || _s == 0) __builtin_amdgcn_s_setprio(0); } } while (0)
#else
#define RR_PRIO(site) do { } while (0)
#endif
template <bool kList, bool kCheck>
__global__ __launch_bounds__(256, RR_K1_BOUNDS) void k_integrate_tiles_rec(StreamTable T, FrameImages F, Volume V, Bricks B, TileState S, const uint4* __restrict__ recs,
                                                                     const unsigned long long* __restrict__ vmasks) {
  [[maybe_unused]] constexpr bool kSep = true, kRanges = true, kCache = false;
  __shared__ float4 s_box[kBoxCap];             // the stream's texel box, x fastest: ((z - mz) * dy + (y - my)) * dx + (x - mx)
  __shared__ float4 s_row[kSep ? kRowCap : 1];  // (separable form) x-lerped rows: ((z - mz) * dy + (y - my)) * 8 + voxel x
  __shared__ int s_i0a[TSDF_MAX_STREAMS][3][8], s_i1a[TSDF_MAX_STREAMS][3][8];   // per stream, axis and voxel coordinate of the tile: the two texel indices ...
  __shared__ float s_wa[TSDF_MAX_STREAMS][3][8];                                 // ... and the weight of the GL LINEAR filter
  const float step[3] = {1.0f / (float)V.res[0], 1.0f / (float)V.res[1], 1.0f / (float)V.res[2]};       // volume_sampler.cpp:36-38
  const float limit = V.limit;
  const int n_work = kList ? (int)*S.count : S.n;
  const int tid = threadIdx.x;
  if (kList && blockIdx.x == 0 && tid == 0) *S.next_count = 0u;                    // the previous list was consumed by the classify launch
  // INVARIANT of the tile loop: every exit from an iteration after phase A has touched s_* passes through store_tile_class(), whose
  // workgroup barrier is what lets the next iteration overwrite s_* (there is no barrier at the head of the loop).  The `continue`s
  // below are workgroup-uniform and sit in front of phase A.
  // Round 4 (in-kernel stamps, profiles/r04_k1_phase_trace_*.txt): 29 % of a tile's 24 k cycles went by BEFORE its first stream -- the list entry (a
  // dependent scalar load), then phase A's per-lane loads of the LUT resolutions out of the kernel arguments (a vector-memory round trip per tile for
  // values that never change).  Now: a thread's phase-A assignment (stream, axis, coordinate) is fixed by its index, so its LUT resolution is loaded
  // once per workgroup (into LDS); the next tile's list entry and pair classes are requested at the head of the current tile.
  __shared__ int s_invres[TSDF_MAX_STREAMS][3], s_vres[3];                         // LUT / volume resolutions: phase A reads them per (stream, axis) -- from LDS, once per workgroup
  __shared__ float s_step[3];
  if (tid < T.n * 3) s_invres[tid / 3][tid % 3] = T.s[tid / 3].inv_res[tid % 3];
  if (tid < 3) { s_vres[tid] = tid == 0 ? V.res[0] : (tid == 1 ? V.res[1] : V.res[2]); s_step[tid] = tid == 0 ? step[0] : (tid == 1 ? step[1] : step[2]); }
  __syncthreads();
  // The work item's record (k_pair_masks): {pair classes, tile id, tile coordinates + "every voxel lies inside the volume", stored tile index}.  The next
  // item's record is requested at the head of the current one -- as a VECTOR load (mbcnt keeps the compiler from proving the address uniform): scalar loads
  // share their counter with the LDS operations, whose waits would expose it at once, while a vector load's counter is in order and waited for by number.
#ifdef RR_K1_SREC       // A/B (tools/build_k1_variant.sh srec "-DRR_K1_SREC"): the record through the scalar cache instead of the texture-address queue
  const int lane0 = 0;
#else
  const int lane0 = (int)__builtin_amdgcn_mbcnt_lo(0u, 0u);                          // 0 in every lane
#endif
  uint4 rec_next = recs[(blockIdx.x < n_work ? blockIdx.x : 0) + lane0];
  // (kCheck) the tile's per-voxel "is it drawn" words for this thread's two voxel layers (k_pair_masks; meaningful when the record's bit 31 is clear): with the record, a tile ahead
  [[maybe_unused]] unsigned long long vm_next[2] = {0ull, 0ull};
  if (kCheck) {
    const size_t w0 = (size_t)(blockIdx.x < n_work ? blockIdx.x : 0) * 8 + (size_t)(threadIdx.x >> 6);
    vm_next[0] = vmasks[w0 + lane0]; vm_next[1] = vmasks[w0 + 4 + lane0];
  }
  for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
    RR_STAMP(0);                                                                     // tile head
    RR_PRIO(0);
    [[maybe_unused]] int _stream_slot = 3;
    const uint32_t pairs = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec_next.x), packed = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec_next.z);
    const int tile = __builtin_amdgcn_readfirstlane((int)rec_next.y);
    const uint32_t stored = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec_next.w);
    [[maybe_unused]] const unsigned long long vm[2] = {vm_next[0], vm_next[1]};
    {
      const int wn = w + (int)gridDim.x, wl = wn < n_work ? wn : w;
      rec_next = recs[wl + lane0];
      if (kCheck) { const size_t o = (size_t)wl * 8 + (size_t)(threadIdx.x >> 6); vm_next[0] = vmasks[o + lane0]; vm_next[1] = vmasks[o + 4 + lane0]; }
    }
    const int t3[3] = {(int)(packed & 1023u), (int)((packed >> 10) & 1023u), (int)((packed >> 20) & 1023u)};
    const bool interior = (packed >> 30) & 1u;                          // all 512 voxels of the tile lie inside the volume
    float* __restrict__ out = V.data + ((size_t)stored << 9);
    constexpr int kVox = 2;                                             // voxels per thread: local z = lz and lz + 4, same x and y
    const int lx = tid & 7, ly = (tid >> 3) & 7, lz = tid >> 6;
    const int x = t3[0] * 8 + lx, y = t3[1] * 8 + ly;
    bool drawn[kVox];
    float tsd[kVox], wsum[kVox];
    const bool check_voxels = kCheck && !(T.n <= 15 && (pairs >> 31));
#pragma unroll
    for (int h = 0; h < kVox; ++h) {
      const int z = t3[2] * 8 + lz + 4 * h;
      drawn[h] = interior || ((x < V.res[0]) && (y < V.res[1]) && (z < V.res[2]));
      tsd[h] = limit;                                                   // tsdf_integration.vs:28-29
      wsum[h] = 0.0f;
    }
    if (kCheck && check_voxels) {                                         // (workgroup-uniform) the answers of k_pair_masks: bit y * 8 + x of the layer's word
#pragma unroll
      for (int h = 0; h < kVox; ++h) drawn[h] = ((vm[h] >> (tid & 63)) & 1ull) != 0ull;
    }
    // (the previous tile's readers of s_* are done: store_tile_class() at its end is a workgroup barrier)                // (kRanges) 2 bits per stream, from k_pair_masks: which streams treat every voxel of this tile alike
    for (int t = tid; t < T.n * 24; t += 256) {                         // phase A, all streams at once
      const int i = t / 24, a = (t % 24) >> 3, k = t & 7;
      const int ta = a == 0 ? t3[0] : (a == 1 ? t3[1] : t3[2]);
      const int coord = min(ta * 8 + k, s_vres[a] - 1);                 // padding voxels reuse the last real coordinate
      const Axis ax = axis_linear(((float)coord + 0.5f) * s_step[a], s_invres[i][a]);
      s_i0a[i][a][k] = ax.i0; s_i1a[i][a][k] = ax.i1; s_wa[i][a][k] = ax.a;
    }
    RR_STAMP(1);                                                                     // own part of phase A done (incl. the voxel checks)
    __syncthreads();
    RR_STAMP(2);                                                                     // phase A barrier passed
    RR_PRIO(1);
    for (int i = 0; i < T.n; ++i) {
      if (kRanges) {
        const int pair = (int)((pairs >> (2 * i)) & 3u);                // workgroup-uniform
        if (pair != kPairFull) {                                        // no box, no passes, no gathers: the branch is the same for every voxel
#pragma unroll
          for (int h = 0; h < kVox; ++h)
            if (drawn[h]) {
              if (pair == kPairNeg) tsd[h] = -limit;
              else if (pair == kPairCarve && tsd[h] >= limit) tsd[h] = -limit;
            }
          continue;
        }
      }
      const StreamLut& L = T.s[i];
      const int (*s_i0)[8] = s_i0a[i];
      const int (*s_i1)[8] = s_i1a[i];
      const float (*s_w)[8] = s_wa[i];
      if (i) __syncthreads();                                           // the previous stream's readers of s_box are done
      const int mx = s_i0[0][0], my = s_i0[1][0], mz = s_i0[2][0];
      const int dx = s_i1[0][7] - mx + 1, dy = s_i1[1][7] - my + 1, dz = s_i1[2][7] - mz + 1;
      {                                                                 // phase B
        const int n = min(__mul24(__mul24(dx, dy), dz), kBoxCap);
        // e -> (bx, by, bz) without integer division (three of them are ~45 VALU instructions, and VALU issue is this launch's largest cost):
        // floor((e + .5) * (1 / d)) in fp32 equals e / d for all 0 <= e < 1024, 1 <= d <= 1024 (checked exhaustively; the quotient is at
        // least 0.5 / d away from the next integer, four orders of magnitude more than the 1 ulp of the hardware reciprocal)
        const float rdx = __builtin_amdgcn_rcpf((float)dx), rdy = __builtin_amdgcn_rcpf((float)dy);
        for (int e = tid; e < n; e += 256) {
          const int row = (int)(((float)e + 0.5f) * rdx);               // e / dx
          const int bz = (int)(((float)row + 0.5f) * rdy);              // row / dy
          const int bx = e - __mul24(row, dx), by = row - __mul24(bz, dy);
          // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): operands are LUT
          // coordinates / resolutions <= 2048, the texel index fits 32 bits (tsdf_set_calibration rejects larger LUTs)
          s_box[e] = L.inv[(uint32_t)__mul24(__mul24(mz + bz, L.inv_res[1]) + (my + by), L.inv_res[0]) + (uint32_t)(mx + bx)];
        }
      }
      RR_STAMP(_stream_slot);                                                        // own box texels stored
      RR_PRIO(3);
      __syncthreads();
      RR_STAMP(_stream_slot + 1);                                                    // phase B barrier passed
      if (kSep) {                                                       // passes X and Y
        const int n1 = min(__mul24(__mul24(dy, dz), 8), kRowCap);
        for (int e = tid; e < n1; e += 256) {
          const int k = e & 7, rb = __mul24(e >> 3, dx);
          const float3 r = lerp3(s_box[rb + (s_i0[0][k] - mx)], s_box[rb + (s_i1[0][k] - mx)], s_w[0][k]);
          s_row[e] = make_float4(r.x, r.y, r.z, 0.0f);
        }
        __syncthreads();
        const int n2 = min(dz << 6, kBoxCap);
        for (int e = tid; e < n2; e += 256) {                           // the y-lerped planes overwrite the box
          const int k = e & 7, j = (e >> 3) & 7, zb = __mul24(e >> 6, dy);
          const float3 r = lerp3(s_row[((zb + (s_i0[1][j] - my)) << 3) + k], s_row[((zb + (s_i1[1][j] - my)) << 3) + k], s_w[1][j]);
          s_box[e] = make_float4(r.x, r.y, r.z, 0.0f);
        }
        __syncthreads();
      }
      RR_STAMP(_stream_slot + 2);                                                    // passes X and Y done
      RR_PRIO(4);
      bool any_drawn = false;
#pragma unroll
      for (int h = 0; h < kVox; ++h) any_drawn |= drawn[h];
      if (__ballot(any_drawn) != 0ull) {                                // phase Z
        const int x0 = s_i0[0][lx] - mx, x1 = s_i1[0][lx] - mx;
        const int y0 = __mul24(s_i0[1][ly] - my, dx), y1 = __mul24(s_i1[1][ly] - my, dx);
        const int pl = __mul24(dx, dy);
        const float wx = s_w[0][lx], wy = s_w[1][ly];
#pragma unroll
        for (int h = 0; h < kVox; ++h) {
          const int kz = lz + 4 * h;
          const int z0 = __mul24(s_i0[2][kz] - mz, pl), z1 = __mul24(s_i1[2][kz] - mz, pl);
          float3 pc;                                                    // texture(cv_xyz_inv[i], position).xyz, :31
          if (kSep) {
            pc = lerp3(s_box[(((s_i0[2][kz] - mz) << 3) + ly) * 8 + lx], s_box[(((s_i1[2][kz] - mz) << 3) + ly) * 8 + lx], s_w[2][kz]);
          } else {
            const float3 c00 = lerp3(s_box[z0 + y0 + x0], s_box[z0 + y0 + x1], wx), c10 = lerp3(s_box[z0 + y1 + x0], s_box[z0 + y1 + x1], wx);
            const float3 c01 = lerp3(s_box[z1 + y0 + x0], s_box[z1 + y0 + x1], wx), c11 = lerp3(s_box[z1 + y1 + x0], s_box[z1 + y1 + x1], wx);
            pc = lerp3(lerp3(c00, c10, wy), lerp3(c01, c11, wy), s_w[2][kz]);
          }
          // the gather and the fusion rule are two branches on purpose: in one branch the compiler keeps voxel 1's loads behind
          // voxel 0's arithmetic; apart, the unrolled loop has both voxels' gathers in flight together
          Dqs q;
          if (drawn[h]) q = dqs_fetch(F, i, pc.x, pc.y);
          if (drawn[h]) {
            float weighted_tsd = tsd[h], total_weight = wsum[h];        // tsdf_integration.vs:30-55, in stream order
            bool skip = false;
            if (dqs_silhouette(q) < 1.0f) {
              if (weighted_tsd >= limit) { weighted_tsd = -limit; skip = true; }
            }
            if (!skip) {
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
            tsd[h] = weighted_tsd; wsum[h] = total_weight;
          }
        }
      }
      RR_STAMP(_stream_slot + 3);                                                    // phase Z done (gathers + fusion)
      RR_PRIO(6);
      _stream_slot += 4;
    }
    RR_STAMP(13);
    RR_PRIO(7);
#pragma unroll
    for (int h = 0; h < kVox; ++h) { tsd[h] = drawn[h] ? tsd[h] : -limit; out[tid + 256 * h] = tsd[h]; }   // clearImage(-limit), :249-250
    RR_STAMP(14);
    store_tile_class(S, tile, tsd[0] == -limit && tsd[1] == -limit);
    RR_STAMP(15);
  }
}


// Cached form (round 3): the work items whose x/y-filtered LUT planes are in the pool (ProjCache, tsdf_common.hpp).  One WAVE per
// z-plane of a tile (64 voxels, lane = x + 8 y), nothing shared between waves: no LDS, no barrier.  Per plane: for each stream the
// pair-mask pass left to per-voxel evaluation, the two cached planes the voxels' z filter taps (two coalesced 768-byte reads; the filter
// set-up of the plane's one z coordinate is wave-uniform) and one lerp give texture(cv_xyz_inv[i], position).xyz -- the same operands and
// operation as the last step of k_integrate_tiles_lds, hence the same bits --, then the 2x2 image footprints of kChunk streams in
// flight together, the fusion rule in stream order on registers, one 256-byte store.
// Unit u -> (work item, plane): the eight planes of a tile go to blocks of equal blockIdx % 8, i.e. (round-robin dispatch) to ONE XCD,
// whose L2 then fetches each cached plane from HBM once although up to three waves read it.  Speed only, never correctness.
// HBM per tile: 2 KiB stored + dz x 768 B x (streams evaluated per voxel) read + the image footprints (L2 / MALL resident).
// The loop is software-pipelined by hand, because a wave's vector-memory operations complete IN ORDER (one vmcnt counter for loads
// and stores): a persistent wave that stores its plane and then loads the next unit's meta data, planes and footprints one after the
// other pays four dependent round trips per unit plus the store's acknowledgement (measured: 44 us per launch at c2, 23 us of it
// with the plane loads and the gathers compiled out).  So
//   * the meta data of a unit (cache slot, pair classes, tile id) are SCALAR loads -- the kernel arguments they come through are
//     `const __restrict__`, the index is wave-uniform --, counted by lgkmcnt, and they are issued one unit ahead;
//   * the cached planes of the NEXT unit (its first kPre streams) are requested after the fusion rule of the current unit and BEFORE
//     its stores: when the next iteration waits for them, the stores behind them in the queue do not hold it up, and they have had the
//     current unit's whole gather phase to arrive.
// What is left on a unit's critical path is its footprint gathers.
#ifndef RR_K1C_CHUNK
#define RR_K1C_CHUNK 1      // streams whose footprints are gathered together (2 needs more than the 64 VGPRs of 8 waves per SIMD: spills)
#endif
#ifndef RR_K1C_PRE
#define RR_K1C_PRE 1        // streams of the NEXT unit whose planes are prefetched
#endif
#ifndef RR_K1C_GRID
#define RR_K1C_GRID 8192
#endif
#ifndef RR_K1C_BOUNDS
#define RR_K1C_BOUNDS 8
#endif
__device__ __forceinline__ float3 load_f3(const float* __restrict__ p) { return make_float3(p[0], p[1], p[2]); }
struct CachedUnit {                                                     // everything wave-uniform
  uint32_t item, pairs;
  int tile, kz;
  bool live;
};
struct CachedPlanes { float3 a[RR_K1C_PRE], b[RR_K1C_PRE]; float w[RR_K1C_PRE]; };
template <bool kList>
__device__ __forceinline__ CachedUnit cached_unit(int u, int n_work, const uint32_t* __restrict__ items, const uint32_t* __restrict__ pair_masks,
                                                  const uint32_t* __restrict__ list, const TileState& S) {
  CachedUnit c;
  const int w = ((u >> 6) << 3) + (u & 7);
  c.kz = (u >> 3) & 7;
  c.live = w < n_work;
  const int wc = c.live ? w : 0;
  c.item = items[wc];
  c.pairs = pair_masks[wc];                                             // 2 bits per stream (k_pair_masks), bit 31: every brick reaching into the tile is occupied
  c.tile = kList ? (int)list[wc] : work_tile<false>(S, wc);
  c.live = c.live && c.item < kItemFresh;                               // else not cached (yet): the LDS kernel's
  return c;
}
// the two cached planes the z filter of plane kz taps in stream i, and its weight (wave-uniform indices; phase A of the LDS kernel: padding
// voxels reuse the last real coordinate)
__device__ __forceinline__ void cached_planes(const ProjCache& PC, const Volume& V, const CachedUnit& c, int tz, int i, int ln, float3& pa, float3& pb, float& wz) {
  const float step_z = 1.0f / (float)V.res[2];                          // volume_sampler.cpp:36-38
  const float cz = ((float)min(tz * 8 + c.kz, V.res[2] - 1) + 0.5f) * step_z, cz0 = ((float)min(tz * 8, V.res[2] - 1) + 0.5f) * step_z;
  const Axis az = axis_linear(cz, PC.inv_rz[i]);
  const int mz = axis_linear(cz0, PC.inv_rz[i]).i0;
  const float* __restrict__ const bi = PC.data + (size_t)c.item * PC.slot_floats + (uint32_t)__mul24(i, (int)PC.dz_max) * 192u + (uint32_t)ln * 3u;
  pa = load_f3(bi + (uint32_t)(az.i0 - mz) * 192u);
  pb = load_f3(bi + (uint32_t)(az.i1 - mz) * 192u);
  wz = az.a;
}
template <bool kList, int kChunk>
__global__ __launch_bounds__(64, RR_K1C_BOUNDS) void k_integrate_cached(int n_streams, FrameImages F, Volume V, Bricks B, TileState S, int per_voxel_check,
                                                                         const uint32_t* __restrict__ count, const uint32_t* __restrict__ list,
                                                                         const uint32_t* __restrict__ pair_masks, const uint32_t* __restrict__ items, ProjCache PC) {
  constexpr int kPre = RR_K1C_PRE;
  static_assert(kPre % kChunk == 0, "whole chunks are prefetched");
  const float limit = V.limit;
  const int n_work = kList ? (int)*count : S.n;
  const int n_units = ((n_work + 7) >> 3) << 6;                         // 8 planes x work items, padded to whole groups of 8 items
  const int ln = threadIdx.x, lx = ln & 7, ly = ln >> 3;
  int u = blockIdx.x;
  if (u >= n_units) return;
  CachedUnit cur = cached_unit<kList>(u, n_work, items, pair_masks, list, S);
  int t3[3];
  tile_coords(V, cur.tile, t3[0], t3[1], t3[2]);
  CachedPlanes P;
#pragma unroll
  for (int i = 0; i < kPre; ++i)
    if (cur.live && i < n_streams && ((cur.pairs >> (2 * i)) & 3u) == (uint32_t)kPairFull) cached_planes(PC, V, cur, t3[2], i, ln, P.a[i], P.b[i], P.w[i]);
  for (;;) {
    const int un = u + (int)gridDim.x;
    const bool more = un < n_units;
    const CachedUnit nxt = cached_unit<kList>(more ? un : u, n_work, items, pair_masks, list, S);   // scalar loads, in flight while this unit is worked
    float tsd = limit, wsum = 0.0f;                                     // tsdf_integration.vs:28-29
    if (cur.live) {
      const int x = t3[0] * 8 + lx, y = t3[1] * 8 + ly, z = t3[2] * 8 + cur.kz;
      bool drawn = (x < V.res[0]) && (y < V.res[1]) && (z < V.res[2]);
      const bool check_voxels = per_voxel_check && !(n_streams <= 15 && (cur.pairs >> 31));
      if (drawn && check_voxels) drawn = voxel_drawn(B, x, y, z);
      for (int c0 = 0; c0 < n_streams; c0 += kChunk) {
        float3 pa[kChunk], pb[kChunk], pc[kChunk];
        float wz[kChunk];
        Dqs q[kChunk];
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
          const int i = c0 + k;
          if (i < n_streams && ((cur.pairs >> (2 * i)) & 3u) == (uint32_t)kPairFull) {
            if (c0 < kPre) { pa[k] = P.a[(c0 + k) % kPre]; pb[k] = P.b[(c0 + k) % kPre]; wz[k] = P.w[(c0 + k) % kPre]; }   // (c0 is a multiple of kChunk: the test is per chunk)
            else cached_planes(PC, V, cur, t3[2], i, ln, pa[k], pb[k], wz[k]);
          }
        }
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {                              // every lane gathers (the coordinates of padding / undrawn voxels are valid ones)
          const int i = c0 + k;
          if (i < n_streams && ((cur.pairs >> (2 * i)) & 3u) == (uint32_t)kPairFull) {
            pc[k] = lerp3(pa[k], pb[k], wz[k]);                         // texture(cv_xyz_inv[i], position).xyz, :31
            q[k] = dqs_fetch(F, i, pc[k].x, pc[k].y);
          }
        }
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
          const int i = c0 + k;
          if (i >= n_streams) break;
          const int pair = (int)((cur.pairs >> (2 * i)) & 3u);          // wave-uniform
          if (pair != kPairFull) {                                      // the branch is the same for every voxel of the tile
            if (pair == kPairNeg) tsd = -limit;
            else if (pair == kPairCarve && tsd >= limit) tsd = -limit;
            continue;
          }
          float weighted_tsd = tsd, total_weight = wsum;                // tsdf_integration.vs:30-55, in stream order
          bool skip = false;
          if (dqs_silhouette(q[k]) < 1.0f) {
            if (weighted_tsd >= limit) { weighted_tsd = -limit; skip = true; }
          }
          if (!skip) {
            const float sdist = pc[k].z - dqs_depth(q[k]);
            if (sdist <= -limit) {
              weighted_tsd = -limit;
            } else if (sdist >= limit) {
            } else {
              const float weight = dqs_quality(q[k]);
              weighted_tsd = (weighted_tsd * total_weight + weight * sdist) / (total_weight + weight);
              total_weight += weight;
            }
          }
          tsd = weighted_tsd; wsum = total_weight;
        }
      }
      tsd = drawn ? tsd : -limit;                                       // clearImage(-limit), :249-250
    }
    // the next unit's planes, requested before this unit's stores (see above)
    int n3[3];
    tile_coords(V, nxt.tile, n3[0], n3[1], n3[2]);
#pragma unroll
    for (int i = 0; i < kPre; ++i)
      if (more && nxt.live && i < n_streams && ((nxt.pairs >> (2 * i)) & 3u) == (uint32_t)kPairFull) cached_planes(PC, V, nxt, n3[2], i, ln, P.a[i], P.b[i], P.w[i]);
    __builtin_amdgcn_sched_barrier(0);
    if (cur.live) {
      float* __restrict__ out = V.data + ((((size_t)(t3[2] - V.tz0) * V.nty + t3[1]) * V.ntx + t3[0]) << 9) + (cur.kz << 6);
      out[ln] = tsd;
      // tile class: the waves of a tile share nothing, so none of them can say "all 512 voxels are the clear value"; kTileMixed is always
      // safe (an exact class only saves the reset of the tile when it leaves the active set, and lets the DENSE march leap over it)
      if (ln == 0 && cur.kz == 0) S.cls[cur.tile] = kTileMixed;
    }
    if (!more) break;
    cur = nxt; u = un;
    t3[0] = n3[0]; t3[1] = n3[1]; t3[2] = n3[2];
  }
}

// counts of the last launch's work items, for bench.py's algorithmic byte count (never in the frame path)
template <bool kList>
__global__ __launch_bounds__(256) void k_item_stats(StreamTable T, TileState S, const uint32_t* __restrict__ masks, ProjCache PC, uint32_t* __restrict__ out) {
  const int n_work = kList ? (int)*S.count : S.n;
  uint32_t cached = 0, full = 0, slow = 0;
  for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < n_work; w += gridDim.x * blockDim.x) {
    const uint32_t it = PC.items[w];
    if (it >= kItemFresh) { ++slow; continue; }
    ++cached;
    const uint32_t m = masks[w];
    for (int i = 0; i < T.n; ++i) full += ((m >> (2 * i)) & 3u) == (uint32_t)kPairFull;
  }
  atomicAdd(&out[1], cached); atomicAdd(&out[2], full); atomicAdd(&out[3], slow);
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (uint32_t)n_work;
}
void launch_item_stats(hipStream_t st, const StreamTable& T, const TileState& S, int use_bricks, const uint32_t* pair_masks, const ProjCache& PC, uint32_t* out) {
  hipMemsetAsync(out, 0, 4 * sizeof(uint32_t), st);
  if (use_bricks) hipLaunchKernelGGL(k_item_stats<true>, dim3(64), dim3(256), 0, st, T, S, pair_masks, PC, out);
  else hipLaunchKernelGGL(k_item_stats<false>, dim3(64), dim3(256), 0, st, T, S, pair_masks, PC, out);
}

void launch_integrate(hipStream_t st, const StreamTable& T, const FrameImages& F, const Volume& V, const Bricks& B, const TileState& S, int use_bricks, int lds_ok,
                      int full_classify, uint32_t frame_stamp, int phase, const PeelClear* pc, const float4* tile_bounds, uint32_t* pair_masks, const ProjCache* proj,
                      uint4* work_recs) {
  // phase 1: tile classification + stale-tile clear; 2: pair-mask pass + integrate kernel(s); 3: the pair-mask pass alone; 4: the integrate
  // kernel(s) alone; 0: everything (the split lets the caller time the kernels separately)
  const ProjCache none_pc{};
  const bool ranges = F.ranges && tile_bounds && pair_masks && lds_ok >= 2;
  const bool cached = ranges && proj && proj->data;
  const bool rec = ranges && !cached && !V.slot && work_recs;           // k_integrate_tiles_rec
  const ProjCache& PC = cached ? *proj : none_pc;
  const int pvc = use_bricks ? (S.uniform ? 0 : 1) : 0;
  if (use_bricks && phase < 2) {
    if (full_classify) hipLaunchKernelGGL(k_classify_clear_tiles, dim3((S.n + 255) / 256), dim3(256), 0, st, V, B, S);
    else {
      PeelClear none{};
      const PeelClear& P = pc ? *pc : none;
      const int extra = P.peels ? (P.n_tiles + 3) / 4 : 0;
      hipLaunchKernelGGL(k_classify_lists, dim3(kScatterBlocks + kStaleBlocks + kZeroBlocks + extra), dim3(256), 0, st, V, B, S, frame_stamp, P);
    }
  }
  if (phase == 1) return;
  if (ranges && phase != 4) {
    unsigned long long* const vmasks = (rec && pvc) ? (unsigned long long*)(work_recs + S.n) : nullptr;   // (the record buffer holds 16 + 64 bytes per tile: abi.cpp)
    if (use_bricks) hipLaunchKernelGGL(k_pair_masks<true>, dim3(2048), dim3(256), 0, st, T, F, V, B, S, pvc, tile_bounds, pair_masks, rec ? work_recs : nullptr, PC, vmasks);
    else hipLaunchKernelGGL(k_pair_masks<false>, dim3((S.n + 3) / 4 < 4096 ? (S.n + 3) / 4 : 4096), dim3(256), 0, st, T, F, V, B, S, 0, tile_bounds, pair_masks, rec ? work_recs : nullptr, PC, (unsigned long long*)nullptr);
  }
  if (phase == 3) return;
  if (use_bricks) {
    // Workgroups of the culled launch (they stride over the work list).  2048 = the 8 x 256 a MI355X holds at once: beside the other lanes' kernels (stage overlap)
    // a c2 frame takes 112 - 113 us with it against 119 with 4096 (the launch alone 43.9 against 42.9 us: queued workgroups of this kernel no longer take the slots
    // a co-runner's workgroups wait for), c3 the same either way; the 25 000-tile launch of a 1024^3 volume wants the larger grid (c4: 2 834 against 2 551 frames/s).
    static const int forced = [] { const char* e = getenv("RR_K1_GRID"); return e ? atoi(e) : 0; }();                  // A/B hook
    const int cap = forced > 0 ? forced : (S.n <= 262144 ? 2048 : 4096);
    const dim3 grid(S.n < cap ? S.n : cap);
    if (cached) {
      hipLaunchKernelGGL((k_integrate_cached<true, RR_K1C_CHUNK>), dim3(RR_K1C_GRID), dim3(64), 0, st, T.n, F, V, B, S, pvc, S.count, S.list, pair_masks, PC.items, PC);
      hipLaunchKernelGGL((k_integrate_tiles_lds<true, true, true, true>), grid, dim3(256), 0, st, T, F, V, B, S, pvc, pair_masks, PC);
    }
    else if (rec && pvc) hipLaunchKernelGGL((k_integrate_tiles_rec<true, true>), grid, dim3(256), 0, st, T, F, V, B, S, work_recs, (const unsigned long long*)(work_recs + S.n));
    else if (rec) hipLaunchKernelGGL((k_integrate_tiles_rec<true, false>), grid, dim3(256), 0, st, T, F, V, B, S, work_recs, (const unsigned long long*)nullptr);
    else if (ranges) hipLaunchKernelGGL((k_integrate_tiles_lds<true, true, true>), grid, dim3(256), 0, st, T, F, V, B, S, pvc, pair_masks, PC);
    else if (lds_ok >= 2) hipLaunchKernelGGL((k_integrate_tiles_lds<true, true>), grid, dim3(256), 0, st, T, F, V, B, S, pvc, nullptr, PC);
    else if (lds_ok) hipLaunchKernelGGL((k_integrate_tiles_lds<true, false>), grid, dim3(256), 0, st, T, F, V, B, S, pvc, nullptr, PC);
    else hipLaunchKernelGGL(k_integrate_tiles<true>, grid, dim3(256), 0, st, T, F, V, B, S, pvc);
  } else {
    if (cached) {
      hipLaunchKernelGGL((k_integrate_cached<false, RR_K1C_CHUNK>), dim3((unsigned)(((S.n + 7) >> 3) << 6)), dim3(64), 0, st, T.n, F, V, B, S, 0, S.count, S.list, pair_masks, PC.items, PC);
      hipLaunchKernelGGL((k_integrate_tiles_lds<false, true, true, true>), dim3(S.n), dim3(256), 0, st, T, F, V, B, S, 0, pair_masks, PC);
    }
    else if (ranges) {
      // at most 16 384 workgroups striding over the tiles instead of one per tile: the launch alone is as fast (119 us at c1), the frame beside the other
      // lanes 2.5 % faster (4 349 against 4 242 frames/s; 8 192: 4 380 but the launch alone 125 us, 2 048: 3 796); RR_K1_DENSE_GRID: A/B hook
      static const int dcap = [] { const char* e = getenv("RR_K1_DENSE_GRID"); return e ? atoi(e) : 16384; }();
      const dim3 dgrid(dcap > 0 && dcap < S.n ? dcap : S.n);
      if (rec) hipLaunchKernelGGL((k_integrate_tiles_rec<false, false>), dgrid, dim3(256), 0, st, T, F, V, B, S, work_recs, (const unsigned long long*)nullptr);
      else hipLaunchKernelGGL((k_integrate_tiles_lds<false, true, true>), dgrid, dim3(256), 0, st, T, F, V, B, S, 0, pair_masks, PC);
    }
    else if (lds_ok >= 2) hipLaunchKernelGGL((k_integrate_tiles_lds<false, true>), dim3(S.n), dim3(256), 0, st, T, F, V, B, S, 0, nullptr, PC);
    else if (lds_ok) hipLaunchKernelGGL((k_integrate_tiles_lds<false, false>), dim3(S.n), dim3(256), 0, st, T, F, V, B, S, 0, nullptr, PC);
    else hipLaunchKernelGGL(k_integrate_tiles<false>, dim3(S.n), dim3(256), 0, st, T, F, V, B, S, 0);
  }
}
int integrate_box_cap() { return kBoxCap; }
int integrate_row_cap() { return kRowCap; }

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
