// Frame packing + brick occupancy kernels (gfx950).
//   pack_frame_fused / pack_color / frame_ranges : upload-side re-layout into the HBM-resident image formats
//   mark_bricks             : pre_normal.fs:22-33 call site of mark_brick(), inc_bricks.glsl:40-58
//   update_occupied         : updateOccupiedBricks(), recon_integration.cpp:430-445, without the host round trip
#include "sampling.hpp"
#include "bricks_dev.hpp"

namespace rr {

// Per 8x8-pixel cell of the packed {depth, quality, silhouette} image: {min depth, max depth, min silhouette, max silhouette}.  One wave per
// cell (a lane per pixel), four cells per workgroup.  A NaN anywhere in a cell poisons its range to (-inf, +inf): the consumer's tests
// then fail and it falls back to the per-voxel evaluation.
__global__ __launch_bounds__(256) void k_frame_ranges(const float4* __restrict__ dqs, int n_streams, int w, int h, int rcw, int rch, float4* __restrict__ ranges) {
  const int cell = blockIdx.x * 4 + (threadIdx.x >> 6), ln = threadIdx.x & 63;
  if (cell >= n_streams * rcw * rch) return;
  const int i = cell / (rcw * rch), r = cell % (rcw * rch), cy = r / rcw, cx = r % rcw;
  const int px = cx * 8 + (ln & 7), py = cy * 8 + (ln >> 3);
  float d0 = __builtin_inff(), d1 = -__builtin_inff(), s0 = __builtin_inff(), s1 = -__builtin_inff();
  bool nan = false;
  if (px < w && py < h) {
    const float4 t = dqs[((size_t)i * h + py) * w + px];
    d0 = d1 = t.x; s0 = s1 = t.z;
    nan = (t.x != t.x) || (t.z != t.z);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    d0 = fminf(d0, __shfl_xor(d0, m)); d1 = fmaxf(d1, __shfl_xor(d1, m));
    s0 = fminf(s0, __shfl_xor(s0, m)); s1 = fmaxf(s1, __shfl_xor(s1, m));
  }
  if (__ballot(nan) != 0ull) { d0 = s0 = -__builtin_inff(); d1 = s1 = __builtin_inff(); }
  if (ln == 0) ranges[cell] = make_float4(d0, d1, s0, s1);
}
void launch_frame_ranges(hipStream_t st, const float4* dqs, int n_streams, int w, int h, float4* ranges) {
  const int rcw = (w + 7) / 8, rch = (h + 7) / 8, cells = n_streams * rcw * rch;
  hipLaunchKernelGGL(k_frame_ranges, dim3((cells + 3) / 4), dim3(256), 0, st, dqs, n_streams, w, h, rcw, rch, ranges);
}
__global__ __launch_bounds__(256) void k_pack_color(const uint8_t* __restrict__ rgb, uchar4* __restrict__ rgba, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    rgba[i] = make_uchar4(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], 255);
}
// The three of them in ONE launch (round 3): what has to happen to a NEW frame before the path can read it -- NetKinectArray hands the
// path a new frame per integrate(), kinect_client.cpp:586-599 -- is its re-layout, and as three dependent launches (11.5 + 8.4 + 5.0 us at
// 4 x 640 x 480) it cost an eighth of the frame.  Blocks [0, cell_blocks): one wave per 8x8-pixel cell reads its 64 pixels of the three
// source arrays, writes the packed texel and the depth plane and reduces the cell's range (DPP); the remaining blocks turn RGB8 into RGBA8,
// four pixels per thread (12 bytes in, 16 out).  HBM bound: 23.4 MB in, 29.8 MB out at c2.
constexpr uint32_t kPackZeroBlocks = 64;
__global__ __launch_bounds__(256) void k_pack_frame_fused(const float2* __restrict__ depth_rg, const float* __restrict__ quality, const float* __restrict__ silhouette,
                                                          float4* __restrict__ dqs, float* __restrict__ depth, float4* __restrict__ ranges, int n_streams, int w, int h,
                                                          int rcw, int rch, int cell_blocks, const uint8_t* __restrict__ rgb, uchar4* __restrict__ rgba, uint32_t n_quads, uint32_t n_px,
                                                          int colour_blocks, uint4* __restrict__ zero, uint32_t zero_quads) {
  if ((int)blockIdx.x >= cell_blocks + colour_blocks) {                  // ---- (optional) the brick counters of the frame this one starts: clearOccupiedBricks()'s fill rides along
    for (uint32_t i = (blockIdx.x - (uint32_t)(cell_blocks + colour_blocks)) * blockDim.x + threadIdx.x; i < zero_quads; i += kPackZeroBlocks * blockDim.x) zero[i] = make_uint4(0, 0, 0, 0);
    return;
  }
  if ((int)blockIdx.x >= cell_blocks) {                                  // ---- colour (rgb == nullptr: the grid has no such blocks)
    const uint32_t q = (blockIdx.x - (uint32_t)cell_blocks) * blockDim.x + threadIdx.x;   // pixels 4q .. 4q + 3
    if (q >= n_quads) return;
    if (4u * q + 4u > n_px) {                                            // the last, partial quad: byte by byte
      for (uint32_t k = 4u * q; k < n_px; ++k) rgba[k] = make_uchar4(rgb[3 * k], rgb[3 * k + 1], rgb[3 * k + 2], 255);
      return;
    }
    const uint3 s = ((const uint3*)rgb)[q];                              // bytes r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
    uint4 o;
    o.x = (s.x & 0x00ffffffu) | 0xff000000u;
    o.y = (s.x >> 24) | ((s.y & 0x0000ffffu) << 8) | 0xff000000u;
    o.z = (s.y >> 16) | ((s.z & 0x000000ffu) << 16) | 0xff000000u;
    o.w = (s.z >> 8) | 0xff000000u;
    ((uint4*)rgba)[q] = o;
    return;
  }
  const int cell = blockIdx.x * 4 + (threadIdx.x >> 6), ln = threadIdx.x & 63;
  if (cell >= n_streams * rcw * rch) return;
  // four cells of a workgroup lie side by side in x: a row of the block is 32 consecutive pixels
  const int i = cell / (rcw * rch), r = cell % (rcw * rch), cy = r / rcw, cx = r % rcw;
  const int px = cx * 8 + (ln & 7), py = cy * 8 + (ln >> 3);
  const float inf = __builtin_inff();
  float d0 = inf, d1 = -inf, s0 = inf, s1 = -inf;
  bool nan = false;
  if (px < w && py < h) {
    const size_t p = ((size_t)i * h + py) * w + px;
    const float d = depth_rg[p].x, q = quality[p], s = silhouette[p];
    dqs[p] = make_float4(d, q, s, 0.0f);
    depth[p] = d;
    d0 = d1 = d; s0 = s1 = s;
    nan = (d != d) || (s != s);
  }
  d0 = wave_min_f32(d0); d1 = wave_max_f32(d1); s0 = wave_min_f32(s0); s1 = wave_max_f32(s1);
  if (__ballot(nan) != 0ull) { d0 = s0 = -inf; d1 = s1 = inf; }          // a NaN anywhere poisons the cell's range (k_frame_ranges)
  if (ln == 0) ranges[cell] = make_float4(d0, d1, s0, s1);
}
void launch_pack_frame_fused(hipStream_t st, const float* depth_rg, const float* quality, const float* silhouette, float4* dqs, float* depth, float4* ranges,
                             int n_streams, int w, int h, const uint8_t* rgb, uchar4* rgba, size_t n_color_px, uint32_t* zero, uint32_t zero_words) {
  const int rcw = (w + 7) / 8, rch = (h + 7) / 8, cells = n_streams * rcw * rch, cell_blocks = (cells + 3) / 4;
  const uint32_t n_quads = rgb ? (uint32_t)((n_color_px + 3) / 4) : 0u;
  const int colour_blocks = (int)((n_quads + 255) / 256);
  hipLaunchKernelGGL(k_pack_frame_fused, dim3((unsigned)(cell_blocks + colour_blocks) + (zero ? kPackZeroBlocks : 0u)), dim3(256), 0, st, (const float2*)depth_rg, quality, silhouette,
                     dqs, depth, ranges, n_streams, w, h, rcw, rch, cell_blocks, rgb, rgba, n_quads, (uint32_t)n_color_px, colour_blocks, (uint4*)zero, zero_words >> 2);
}
__global__ __launch_bounds__(256) void k_fill_u32(uint32_t* __restrict__ p, uint32_t v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

static inline int grid_for(size_t n, int block = 256, int cap = 2048) {
  size_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > (size_t)cap ? cap : g));
}
void launch_pack_color(hipStream_t st, const uint8_t* rgb, uchar4* rgba, size_t n) {
  hipLaunchKernelGGL(k_pack_color, dim3(grid_for(n)), dim3(256), 0, st, rgb, rgba, n);
}
void launch_fill_u32(hipStream_t st, uint32_t* p, uint32_t v, size_t n) {
  hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(n)), dim3(256), 0, st, p, v, n);
}

// One thread per depth pixel and stream (pre_normal.fs:22-33 runs per fragment of every layer).
__global__ __launch_bounds__(256) void k_mark_bricks(StreamTable T, FrameImages F, Bricks B, uint32_t* __restrict__ zero_word, PeelClear PC) {
  if (zero_word && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *zero_word = 0u;   // the count the coming updateOccupiedBricks() adds to
  if ((int)blockIdx.z == T.n) {                                           // ---- (optional) one more layer of blocks: the peel tiles the coming draw would reset (k_clear_peel_tiles, k_raymarch.hip)
    const int lane = threadIdx.x & 63, nb = (int)(gridDim.x * gridDim.y);
    for (int t = ((int)(blockIdx.y * gridDim.x + blockIdx.x)) * 4 + (int)(threadIdx.x >> 6); t < PC.n_tiles; t += nb * 4) {
      if (!PC.touched_prev[t]) continue;
      const int px = (t % PC.ntx) * 8 + (lane & 7), py = (t / PC.ntx) * 8 + (lane >> 3);
      if (px < PC.w && py < PC.h) PC.peels[(size_t)py * PC.w + px] = make_uint4(__float_as_uint(1.0f), 0u, __float_as_uint(1.0f), 0u);   // clear (1,0,1,0)
    }
    return;
  }
  // (round 4) a wave is an 8 x 8-pixel cell of a 16 x 16 block, not a 64 x 4 strip: the pixels of a cell fall into a third as many bricks as those of a strip, and
  // wave_count below is a loop over the wave's distinct bricks (14 % of the cells of a c2 frame hold a depth against 19 % of the strips)
  const int cell = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int px = blockIdx.x * 16 + ((cell & 1) << 3) + (ln & 7);
  const int py = blockIdx.y * 16 + ((cell >> 1) << 3) + (ln >> 3);
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
void launch_mark_bricks(hipStream_t st, const StreamTable& T, const FrameImages& F, const Bricks& B, uint32_t* zero_word, const PeelClear* pc) {
  dim3 grid((F.w + 15) / 16, (F.h + 15) / 16, T.n + (pc ? 1 : 0));
  hipLaunchKernelGGL(k_mark_bricks, grid, dim3(256), 0, st, T, F, B, zero_word, pc ? *pc : PeelClear{});
}

// flags[b] = counter[b] >= min_voxels, the compacted occupied list and its length (one atomic per wave)
// Four bricks per lane (one 16-byte load, one packed 4-byte flag store): a brick per lane made this a 1 040-workgroup launch of
// one-element threads, 10 us for 1.3 MB.  The occupied list is unordered (as the atomics of the reference's readback leave no
// order either): within a wave it is written bit plane by bit plane.
__global__ __launch_bounds__(256) void k_update_occupied(Bricks B, uint32_t min_voxels, uint32_t* __restrict__ next_count) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;                    // bricks 4q .. 4q + 3
  if (q == 0) *next_count = 0u;                                           // the count word the NEXT update accumulates into
  const int lane = threadIdx.x & 63;
  const int b0 = 4 * q;
  uint32_t bits = 0;
  if (b0 < B.n) {
    const uint4 cnt = ((const uint4*)B.counters)[q];                      // the allocation is padded to 64 words
    bits = (cnt.x >= min_voxels ? 1u : 0u) | ((b0 + 1 < B.n && cnt.y >= min_voxels) ? 2u : 0u) | ((b0 + 2 < B.n && cnt.z >= min_voxels) ? 4u : 0u) |
           ((b0 + 3 < B.n && cnt.w >= min_voxels) ? 8u : 0u);
    const uint32_t packed = (bits & 1u) | ((bits & 2u) << 7) | ((bits & 4u) << 14) | ((bits & 8u) << 21);   // one flag byte per brick
    if (b0 + 3 < B.n) ((uint32_t*)B.flags)[q] = packed;
    else for (int j = 0; j < 4 && b0 + j < B.n; ++j) B.flags[b0 + j] = (uint8_t)((bits >> j) & 1u);
  }
  const unsigned long long m0 = __ballot(bits & 1u), m1 = __ballot(bits & 2u), m2 = __ballot(bits & 4u), m3 = __ballot(bits & 8u);
  const uint32_t n0 = (uint32_t)__popcll(m0), n1 = (uint32_t)__popcll(m1), n2 = (uint32_t)__popcll(m2), n3 = (uint32_t)__popcll(m3);
  if ((m0 | m1 | m2 | m3) == 0ull) return;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(B.num_occupied, n0 + n1 + n2 + n3);
  base = __shfl(base, 0);
  const unsigned long long below = (1ull << lane) - 1ull;
  if (bits & 1u) B.occupied[base + (uint32_t)__popcll(m0 & below)] = (uint32_t)b0;
  if (bits & 2u) B.occupied[base + n0 + (uint32_t)__popcll(m1 & below)] = (uint32_t)(b0 + 1);
  if (bits & 4u) B.occupied[base + n0 + n1 + (uint32_t)__popcll(m2 & below)] = (uint32_t)(b0 + 2);
  if (bits & 8u) B.occupied[base + n0 + n1 + n2 + (uint32_t)__popcll(m3 & below)] = (uint32_t)(b0 + 3);
}
void launch_update_occupied(hipStream_t st, const Bricks& B, uint32_t min_voxels, uint32_t* next_count) {
  const int quads = (B.n + 3) / 4;
  hipLaunchKernelGGL(k_update_occupied, dim3((quads + 255) / 256), dim3(256), 0, st, B, min_voxels, next_count);
}

}  // namespace rr
