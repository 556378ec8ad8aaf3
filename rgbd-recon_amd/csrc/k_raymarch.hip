// K5 + K2: brick depth limits (bricks.vs/gs/fs via drawDepthLimits(), recon_integration.cpp:408-428)
// and the TSDF raymarch (tsdf_raymarch.fs:62-134 via draw(), :176-240).
//
// The reference rasterises every exposed face of every occupied brick with MIN blending to get, per
// pixel, (min z, -max z, min back-face z).  Here one wave per occupied brick scatters the same values
// with integer atomics on the float bit patterns (order independent, hence deterministic); the work is
// proportional to the occupied bricks' screen footprint, not to rays x grid cells.  The raymarch then runs
// one ray per pixel (the reference shades the front and the back cube face with identical results,
// SURVEY.md Appendix C.3).
#include <cstdlib>

#include "sampling.hpp"
#include "shading_dev.hpp"

namespace rr {

// Unnormalised world/volume directions of the ray through pixel centre (fx, fy)
__device__ __forceinline__ float3 pixel_dir_world(const ViewParams& P, float fx, float fy) {
  const float4 p = mat_mul(P.img_to_eye, fx, fy, 1.0f, 1.0f);
  const float4 wd = mat_mul(P.mv_inv, p.x / p.w, p.y / p.w, p.z / p.w, 0.0f);
  return make_float3(wd.x, wd.y, wd.z);
}
__device__ __forceinline__ float3 pixel_dir_vol(const ViewParams& P, float fx, float fy) {
  const float4 p = mat_mul(P.img_to_eye, fx, fy, 1.0f, 1.0f);
  const float4 wd = mat_mul(P.mv_inv, p.x / p.w, p.y / p.w, p.z / p.w, 0.0f);
  const float4 vd = mat_mul(P.v2w_inv, wd.x, wd.y, wd.z, wd.w);
  return make_float3(vd.x, vd.y, vd.z);
}
// screenToVol(), tsdf_raymarch.fs:376-383
__device__ __forceinline__ float3 screen_to_vol(const ViewParams& P, float fx, float fy, float fz) {
  const float4 p = mat_mul(P.img_to_eye, fx, fy, fz, 1.0f);
  const float4 ws = mat_mul(P.mv_inv, p.x / p.w, p.y / p.w, p.z / p.w, 1.0f);
  const float4 vp = mat_mul(P.v2w_inv, ws.x, ws.y, ws.z, ws.w);
  return make_float3(vp.x, vp.y, vp.z);
}

// brick_occupied(get_id(index + offset)) of bricks.gs:26-43 with the shader's wrapping uint arithmetic
__device__ __forceinline__ bool neighbour_gt10(const Bricks& B, int ix, int iy, int iz, int axis, int dir) {
  uint32_t i[3] = {(uint32_t)ix, (uint32_t)iy, (uint32_t)iz};
  i[axis] += (uint32_t)dir;
  const uint32_t id = i[2] * (uint32_t)B.res[1] * (uint32_t)B.res[0] + i[1] * (uint32_t)B.res[0] + i[0];
  return id < (uint32_t)B.n ? (B.counters[id] > 10u) : false;
}

// Depth peels live as float bit patterns so that MIN/MAX blending becomes integer atomics (all values are in
// [0,1], where uint order == float order): x = min z, y = MAX z (the reference keeps min(-z)), z = min back-face z.
__global__ __launch_bounds__(256) void k_clear_peels(uint4* __restrict__ peels, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) peels[i] = make_uint4(__float_as_uint(1.0f), 0u, __float_as_uint(1.0f), 0u);   // clear (1,0,1,0), recon_integration.cpp:144
}

// Image-space dirty tiles: k_depth_limits flags every 8x8-pixel tile it writes a peel into.  The next frame resets only the
// peels of flagged tiles, and k_march neither reads peels nor rewrites clear values for tiles that are untouched in both
// frames (c2: 1085 of 14400 tiles see a brick) -- the dense 15 MB peel clear and ~35 MB of per-frame clear traffic go away.
__global__ __launch_bounds__(256) void k_clear_peel_tiles(uint4* __restrict__ peels, int w, int h, int ntx, int n_tiles, const uint8_t* __restrict__ touched_prev) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), ln = threadIdx.x & 63;
  if (t >= n_tiles || !touched_prev[t]) return;
  const int px = (t % ntx) * 8 + (ln & 7), py = (t / ntx) * 8 + (ln >> 3);
  if (px < w && py < h) peels[(size_t)py * w + px] = make_uint4(__float_as_uint(1.0f), 0u, __float_as_uint(1.0f), 0u);
}

// One wave per occupied brick (persistent, pulling from the compacted list).  For each exposed face the wave
// sweeps the pixel bounding box of the projected face and applies the reference's fragment rule to every pixel
// centre whose ray crosses the face rectangle: z from the ray/plane intersection, near/far clip, MIN blend.
__global__ __launch_bounds__(64) void k_depth_limits(ViewParams P, Bricks B, uint4* __restrict__ peels, uint8_t* __restrict__ touched, int ntx) {
  const int lane = threadIdx.x;
  const int n_occ = (int)*B.num_occupied;
  const float o[3] = {P.cam_world[0], P.cam_world[1], P.cam_world[2]};
  for (int w = blockIdx.x; w < n_occ; w += gridDim.x) {
    const uint32_t id = B.occupied[w];
    int idx[3];
    idx[2] = (int)(id / (uint32_t)(B.res[0] * B.res[1]));                // index_3d(), inc_bricks.glsl:30-38
    const uint32_t rem = id % (uint32_t)(B.res[0] * B.res[1]);
    idx[1] = (int)(rem / (uint32_t)B.res[0]);
    idx[0] = (int)(rem % (uint32_t)B.res[0]);
    float lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {                                         // to_world(), inc_bricks.glsl:22-24
      lo[a] = (float)idx[a] * B.size[a] + B.bbox_min[a] + 0.0f * B.size[a];
      hi[a] = (float)idx[a] * B.size[a] + B.bbox_min[a] + 1.0f * B.size[a];
    }
    // One sweep over the pixel bounding box of the whole brick instead of one per face: the ray direction of a pixel is
    // computed once and tested against the (up to six) drawn faces, and the per-face MIN/MAX blends collapse into at most three
    // atomics per pixel.  Every (pixel, face) pair passes exactly the tests it passed in a per-face sweep -- a pixel outside a
    // face's own bounding box fails that face's rectangle test -- so the peels are unchanged.
    bool drawn[6], front[6];
    float coord[6];
#pragma unroll
    for (int f = 0; f < 6; ++f) {
      const int a = f >> 1, dir = (f & 1) ? 1 : -1;
      drawn[f] = !neighbour_gt10(B, idx[0], idx[1], idx[2], a, dir);        // shared face culled in the GS, bricks.gs:26-43
      coord[f] = dir < 0 ? lo[a] : hi[a];
      front[f] = dir > 0 ? (o[a] > coord[f]) : (o[a] < coord[f]);           // gl_FrontFacing of an outward-wound cube
    }
    if (!(drawn[0] | drawn[1] | drawn[2] | drawn[3] | drawn[4] | drawn[5])) continue;
    // pixel bounding box of the brick (conservative: +-1 px; whole screen if a corner is behind the eye)
    float bx0 = 3.0e38f, bx1 = -3.0e38f, by0 = 3.0e38f, by1 = -3.0e38f;
    bool behind = false;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float4 e = mat_mul(P.mv, (c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2], 1.0f);
      const float4 cl = mat_mul(P.proj, e.x, e.y, e.z, e.w);
      behind |= !(cl.w > 1.0e-6f);
      const float wx = (cl.x / cl.w * 0.5f + 0.5f) * (float)P.w, wy = (cl.y / cl.w * 0.5f + 0.5f) * (float)P.h;
      bx0 = fminf(bx0, wx); bx1 = fmaxf(bx1, wx); by0 = fminf(by0, wy); by1 = fmaxf(by1, wy);
    }
    int x0 = 0, x1 = P.w - 1, y0 = 0, y1 = P.h - 1;
    if (!behind) {
      x0 = max((int)floorf(fmaxf(bx0, -1.0e6f)) - 1, 0); x1 = min((int)floorf(fminf(bx1, 1.0e6f)) + 1, P.w - 1);
      y0 = max((int)floorf(fmaxf(by0, -1.0e6f)) - 1, 0); y1 = min((int)floorf(fminf(by1, 1.0e6f)) + 1, P.h - 1);
    }
    const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
    if (bw <= 0 || bh <= 0) continue;
    for (int k = lane; k < bw * bh; k += 64) {
      const int px = x0 + k % bw, py = y0 + k / bw;
      const float3 dw = pixel_dir_world(P, (float)px + 0.5f, (float)py + 0.5f);
      const float d[3] = {dw.x, dw.y, dw.z};
      float zmin = 2.0f, zmax = -1.0f, zback = 2.0f;
#pragma unroll
      for (int f = 0; f < 6; ++f) {
        if (!drawn[f]) continue;
        const int a = f >> 1, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
        const float t = (coord[f] - o[a]) / d[a];
        if (!(t > 0.0f)) continue;
        const float q1 = o[a1] + t * d[a1], q2 = o[a2] + t * d[a2];
        if (q1 < lo[a1] || q1 > hi[a1] || q2 < lo[a2] || q2 > hi[a2]) continue;
        float hp[3];
        hp[a] = coord[f]; hp[a1] = q1; hp[a2] = q2;
        const float ez = P.mv.m[2] * hp[0] + P.mv.m[6] * hp[1] + P.mv.m[10] * hp[2] + P.mv.m[14] * 1.0f;
        const float zw = (P.proj.m[10] * ez + P.proj.m[14]) / (-ez) * 0.5f + 0.5f;
        if (!(zw >= 0.0f && zw <= 1.0f)) continue;                        // near / far clip
        zmin = fminf(zmin, zw); zmax = fmaxf(zmax, zw);                   // bricks.fs:6 with GL_MIN blending on (z, -z, back z)
        if (!front[f]) zback = fminf(zback, zw);
      }
      if (zmax < 0.0f) continue;                                          // no face of this brick under the pixel
      uint32_t* pp = (uint32_t*)&peels[(size_t)py * P.w + px];
      atomicMin(pp + 0, __float_as_uint(zmin));
      atomicMax(pp + 1, __float_as_uint(zmax));
      if (zback <= 1.0f) atomicMin(pp + 2, __float_as_uint(zback));
      if (touched) touched[(py >> 3) * ntx + (px >> 3)] = 1;
    }
  }
}
void launch_clear_peels(hipStream_t st, float4* peels, int n) { hipLaunchKernelGGL(k_clear_peels, dim3((n + 255) / 256), dim3(256), 0, st, (uint4*)peels, n); }
void launch_depth_limits(hipStream_t st, const ViewParams& P, const Bricks& B, float4* peels, uint8_t* touched_cur, const uint8_t* touched_prev, int already_cleared) {
  const int n = P.w * P.h, ntx = (P.w + 7) / 8, n_tiles = ntx * ((P.h + 7) / 8);
  // touched_prev == nullptr: no tile history (first frame, resized view, ...): reset every peel
  if (touched_prev && already_cleared) { /* integrate()'s classify launch reset the touched tiles (k_classify_lists part C) */ }
  else if (touched_prev) hipLaunchKernelGGL(k_clear_peel_tiles, dim3((n_tiles + 3) / 4), dim3(256), 0, st, (uint4*)peels, P.w, P.h, ntx, n_tiles, touched_prev);
  else hipLaunchKernelGGL(k_clear_peels, dim3((n + 255) / 256), dim3(256), 0, st, (uint4*)peels, n);
  const int grid = B.n < 8192 ? B.n : 8192;
  hipLaunchKernelGGL(k_depth_limits, dim3(grid), dim3(64), 0, st, P, B, (uint4*)peels, touched_cur, ntx);
}

// ------------------------------------------------------------------------------------------- K2
// blendColors(), tsdf_raymarch.fs:295-330.  Streams are taken kShadeChunk at a time: the inverse-LUT taps of the whole chunk are
// issued together, then the colour-LUT taps, then the image footprints; the accumulation itself stays in stream order.
#ifndef RR_SHADE_CHUNK
#define RR_SHADE_CHUNK 4
#endif
constexpr int kShadeChunk = RR_SHADE_CHUNK;   // streams whose taps are in flight together
__device__ float4 blend_colors(const StreamTable& T, const FrameImages& F, float limit, float3 sp) {
  float3 tc = make_float3(0, 0, 0), tc2 = make_float3(0, 0, 0);
  float tw = 0.0f, tw2 = 0.0f;
  for (int cb = 0; cb < T.n; cb += kShadeChunk) {
    const int nc = min(kShadeChunk, T.n - cb);
    float3 pc[kShadeChunk], col[kShadeChunk];
    float2 pcol[kShadeChunk];
    Dqs q[kShadeChunk];
#pragma unroll
    for (int c = 0; c < kShadeChunk; ++c)
      if (c < nc) pc[c] = tex3d_rgba_xyz(T.s[cb + c].inv, T.s[cb + c].inv_res, sp.x, sp.y, sp.z);
#pragma unroll
    for (int c = 0; c < kShadeChunk; ++c)
      if (c < nc) pcol[c] = tex3d_rg(T.s[cb + c].uv, T.s[cb + c].uv_res, pc[c].x, pc[c].y, pc[c].z);
#pragma unroll
    for (int c = 0; c < kShadeChunk; ++c)
      if (c < nc) { col[c] = color_bilinear(F, cb + c, pcol[c].x, pcol[c].y); q[c] = dqs_fetch(F, cb + c, pc[c].x, pc[c].y); }
#pragma unroll
    for (int c = 0; c < kShadeChunk; ++c)
      if (c < nc) {
        const float dist = fabsf(dqs_depth(q[c]) - pc[c].z);
        float quality = 0.0f;
        if (dist < limit) quality = dqs_quality(q[c]);
        const float de = dist + 0.01f;
        tc.x = tc.x + col[c].x * quality / de; tc.y = tc.y + col[c].y * quality / de; tc.z = tc.z + col[c].z * quality / de;
        tw += quality / de;
        tc2.x = tc2.x + col[c].x / dist; tc2.y = tc2.y + col[c].y / dist; tc2.z = tc2.z + col[c].z / dist;
        tw2 += 1.0f / dist;
      }
  }
  if (tw > 0.0f) return make_float4(tc.x / tw, tc.y / tw, tc.z / tw, 1.0f);
  return make_float4(tc2.x / tw2, tc2.y / tw2, tc2.z / tw2, -1.0f);
}
// blendCameras() with getWeights(), tsdf_raymarch.fs:346-361, :151-166
__device__ float3 blend_cameras(const StreamTable& T, const FrameImages& F, float limit, float3 sp) {
  float3 tc = make_float3(0, 0, 0);
  float tw = 0.0f;
  for (int i = 0; i < T.n; ++i) {
    const StreamLut& L = T.s[i];
    const float3 pc = tex3d_rgba_xyz(L.inv, L.inv_res, sp.x, sp.y, sp.z);
    const Dqs q = dqs_fetch(F, i, pc.x, pc.y);
    float w = 0.0f;
    if (fabsf(dqs_depth(q) - pc.z) < limit) w = dqs_quality(q);
    tc.x = tc.x + c_camera_colors[i & 7][0] * w; tc.y = tc.y + c_camera_colors[i & 7][1] * w; tc.z = tc.z + c_camera_colors[i & 7][2] * w;
    tw += w;
  }
  tc.x = tc.x / tw; tc.y = tc.y / tw; tc.z = tc.z / tw;
  if (tw <= 0.0f) tc = make_float3(1.0f, 1.0f, 1.0f);
  return tc;
}
// Which slab owns a sample (multi-GPU): the voxel plane floor(pos.z * rz), clamped.
__device__ __forceinline__ bool sample_owned(const Volume& V, float pz) {
  const float f = fminf(fmaxf(floorf(pz * (float)V.res[2]), 0.0f), (float)(V.res[2] - 1));
  const int tz = ((int)f) >> 3;
  return tz >= V.own_tz0 && tz < V.own_tz1;
}

// K2 runs as two launches.  k_march (one thread per pixel, 8x8 pixel tile per wave) finds the first zero crossing and
// appends it to a compact hit list; k_shade (one thread per hit, dense waves) does submitFragment().  Rays that hit
// nothing cost no shading registers or divergence, and the hit list length never goes through the host.
struct Hit { float x, y, z; uint32_t pix; };   // refined sample position (volume space) + pixel index

#ifndef RR_MARCH_BATCH
#define RR_MARCH_BATCH 8
#endif
#ifndef RR_LONG_LANES
#define RR_LONG_LANES 16
#endif
#ifndef RR_MARCH_BOUNDS
#define RR_MARCH_BOUNDS 1
#endif
// samples in flight per ray in the first march pass: with depth limits (rays of ~8 samples, capped at `cap`) a small batch
// wastes fewer fetches past the hit and needs fewer registers; the dense march (hundreds of samples per ray) wants the deep one
constexpr int kBatchDense = RR_MARCH_BATCH;
#ifndef RR_MARCH_BATCH_SKIP
#define RR_MARCH_BATCH_SKIP 3
#endif
constexpr int kBatchSkip = RR_MARCH_BATCH_SKIP;
#ifndef RR_LONG_BATCH
#define RR_LONG_BATCH 3
#endif
constexpr int kLongBatch = RR_LONG_BATCH;   // samples per lane and round in the long-ray pass

#ifdef RR_MARCH_STATS     // instrumented build (tools/march_stats.py): [max (wave cycles << 32 | pixel), max pre-run steps, max batches, max max_n, working waves, sum of wave cycles]
__device__ unsigned long long g_march_stats[8];
extern "C" int32_t tsdf_debug_march_stats(unsigned long long out[8], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_march_stats), sizeof(g_march_stats)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_march_stats), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif
template <bool kPartial, bool kSparse, int kBatch>
__global__ __launch_bounds__(256, RR_MARCH_BOUNDS) void k_march(ViewParams P, Volume V, RayTarget R, Hit* __restrict__ hits, uint32_t* __restrict__ hit_count,
                                                                 LongRay* __restrict__ longs, uint32_t* __restrict__ long_count, uint32_t cap) {
  constexpr bool partial = kPartial;
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const float limit = V.limit, sd = limit * 0.5f;                       // sampleDistance, :34
  const int px = blockIdx.x * 16 + (wv & 1) * 8 + (ln & 7);
  const int py = blockIdx.y * 16 + (wv >> 1) * 8 + (ln >> 3);
  const bool inside = px < P.w && py < P.h;
  if (R.touched_cur) {
    // this wave IS one 8x8 image tile: untouched now -> nothing to march; untouched before as well -> already holds clear values
    const int ntx = (P.w + 7) >> 3, tx = blockIdx.x * 2 + (wv & 1), ty = blockIdx.y * 2 + (wv >> 1);
    if (tx >= ntx || ty * 8 >= P.h) return;
    const int t = ty * ntx + tx;
    const uint8_t cur = R.touched_cur[t], before = R.touched_prev[t], before_target = R.touched_prev_target[t];
    if (ln == 0) {
      // what the hole filling has to look at: the tiles of this draw and of the two before (the oldest mask is read as touched_prev_target
      // when two pyramids alternate, else through touched_recycle before it is zeroed)
      if (R.fill_mask) R.fill_mask[t] = cur | before | before_target | R.touched_recycle[t];
      R.touched_recycle[t] = 0;                                          // the oldest mask becomes the next draw's (empty) current one
    }
    if (!cur) {
      if (inside) {
        if (before_target || R.rewrite_target) {                         // what the draw that last wrote THIS target left here
          const size_t oi = (size_t)py * R.stride + px;
          R.color[oi] = make_float4(R.clear[0], R.clear[1], R.clear[2], R.clear[3]);
          R.depth[oi] = 1.0f;
        }
        if (before || R.rewrite_all) R.nsamples[(size_t)py * P.w + px] = 0.0f;   // (one sample-count image: the previous draw's)
      }
      return;
    }
  }
#ifdef RR_MARCH_STATS
  const unsigned long long t_start = wall_clock64();
  uint32_t st_pre = 0, st_batches = 0;
#endif
  const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
  const float3 dn = normalize3(pixel_dir_vol(P, fx, fy));
  const float3 step = make_float3(dn.x * sd, dn.y * sd, dn.z * sd);     // :64
  uint32_t max_n = 0;
  float3 pos = make_float3(0, 0, 0);
  bool covered = inside;
  if (inside) {
    if (P.skip) {                                                       // getStartPos(ivec2(gl_FragCoord.xy - viewport_offset)), :70, :384-393
      // gl_FragCoord = viewport origin + pixel + 0.5 (window coordinates); the shader subtracts its viewport_offset uniform again.
      // origin == offset (what the client sets, kinect_client.cpp:650-662) gives back the pixel centre exactly; anything else
      // shifts the peel lookup and the unprojection the way the reference's arithmetic does (out-of-range texelFetch -> 0).
      const float qx = ((float)(px + P.vp_org[0]) + 0.5f) - P.vp_off[0], qy = ((float)(py + P.vp_org[1]) + 0.5f) - P.vp_off[1];
      const int cx = (int)qx, cy = (int)qy;
      uint4 dmb = make_uint4(0u, 0u, 0u, 0u);
      if (cx >= 0 && cy >= 0 && cx < P.w && cy < P.h) dmb = ((const uint4*)R.peels)[(size_t)cy * P.w + cx];
      const float dm_r = __uint_as_float(dmb.x), dm_g = -__uint_as_float(dmb.y), dm_b = __uint_as_float(dmb.z);   // g = min(-z) = -max z
      float r = dm_r;
      r = (r >= dm_b) ? 0.0f : r;
      const float3 pf = screen_to_vol(P, qx, qy, r);
      float3 pb = screen_to_vol(P, qx, qy, -dm_g);
      if (r >= 1.0f) pb = pf;
      pos = pf;
      const float3 dd = make_float3(pf.x - pb.x, pf.y - pb.y, pf.z - pb.z);
      max_n = (uint32_t)ceilf(sqrtf(dd.x * dd.x + dd.y * dd.y + dd.z * dd.z) / sd);   // :73
    } else {                                                            // intersectBox(), :363-374
      const float3 o = make_float3(P.cam_vol[0], P.cam_vol[1], P.cam_vol[2]);
      const float3 inv = make_float3(1.0f / step.x, 1.0f / step.y, 1.0f / step.z);
      const float3 tbot = make_float3(inv.x * (0.0f - o.x), inv.y * (0.0f - o.y), inv.z * (0.0f - o.z));
      const float3 ttop = make_float3(inv.x * (1.0f - o.x), inv.y * (1.0f - o.y), inv.z * (1.0f - o.z));
      const float3 tmn = make_float3(fminf(ttop.x, tbot.x), fminf(ttop.y, tbot.y), fminf(ttop.z, tbot.z));
      const float3 tmx = make_float3(fmaxf(ttop.x, tbot.x), fmaxf(ttop.y, tbot.y), fmaxf(ttop.z, tbot.z));
      const float t0 = fmaxf(fmaxf(tmn.x, tmn.y), fmaxf(tmn.x, tmn.z));
      const float t1 = fminf(fminf(tmx.x, tmx.y), fminf(tmx.x, tmx.z));
      if (!(t0 <= t1) || t1 < 0.0f) covered = false;                    // no fragment: pixel not under the cube
      else {
        const float t_near = t0 < 0.0f ? 0.0f : t0;
        pos = make_float3(o.x + step.x * t_near, o.y + step.y * t_near, o.z + step.z * t_near);
        max_n = (uint32_t)ceilf(fabsf(t1 - t_near));
      }
    }
  }
  // The march, :89-110.  Sample positions never depend on the densities, so (a) while the ray is in space known to hold
  // nothing but -limit, whole runs of samples are accounted for by just performing the reference's `pos += step`
  // additions, and (b) elsewhere four consecutive samples are fetched together and then examined in order.  Positions,
  // densities and counts are those of the one-at-a-time loop.
  float prev = -limit;
  bool prev_valid = true;
  float3 pos_prev = pos;
  uint32_t n = 0;
  bool hit = false;
  bool deferred = false;
  float3 hit_pos = pos;
  float hit_d = 0.0f;
  if (partial) {
    // Slab mode: ownership is by voxel plane and z is monotonic along the ray, so this rank's samples form ONE contiguous
    // run.  Samples before it cost only the reference's `pos += step`; the run ends the march (a miss reports max_n, which
    // every rank computes identically).
    while (n < max_n && !sample_owned(V, pos.z)) {
      pos_prev = pos;
      pos = make_float3(pos.x + step.x, pos.y + step.y, pos.z + step.z);
      n += 1;
      prev_valid = false;
#ifdef RR_MARCH_STATS
      ++st_pre;
#endif
    }
  }
  while (n < max_n && !hit) {
#ifdef RR_MARCH_STATS
    ++st_batches;
#endif
    float3 p[kBatch];
    float d[kBatch];
    bool own[kBatch];
    p[0] = pos;
#pragma unroll
    for (int k = 1; k < kBatch; ++k) p[k] = make_float3(p[k - 1].x + step.x, p[k - 1].y + step.y, p[k - 1].z + step.z);
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      own[k] = (n + k < max_n) && (!partial || sample_owned(V, p[k].z));
      d[k] = tex3d_tsdf<kSparse, !kPartial>(V, p[k].x, p[k].y, p[k].z);   // unconditional: taps are clamped into the allocation, and a predicated
                                                      // fetch would make the compiler wait for each sample's loads separately
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      if (!hit && n < max_n) {
        n += 1;
        if (own[k]) {
          if (d[k] > 0.0f) {
            hit = true;
            hit_pos = p[k];
            hit_d = d[k];
          } else {
            prev = d[k];
            prev_valid = true;
            pos_prev = p[k];
          }
        } else {
          prev_valid = false;
          pos_prev = p[k];
        }
      }
    }
    if (!hit) pos = make_float3(p[kBatch - 1].x + step.x, p[kBatch - 1].y + step.y, p[kBatch - 1].z + step.z);
    if (partial && !hit && !prev_valid) { n = max_n; break; }          // the batch ended outside the slab: nothing further is ours
    if (!partial && !hit && n >= cap && n < max_n) { deferred = true; break; }   // a long ray: k_march_long finishes it, one wave per ray
  }
  // Long rays are few (c2: the mean ray has 8 samples, the longest 92) but a wave lasts as long as its longest ray and the
  // launch as long as its slowest wave: hand them to a second kernel that puts 64 lanes on one ray.
  const unsigned long long dm = __ballot(deferred);
  if (dm) {
    const int leader = __ffsll((long long)dm) - 1;
    uint32_t base = 0;
    if (ln == leader) base = atomicAdd(long_count, (uint32_t)__popcll(dm));
    base = __shfl(base, leader);
    if (deferred) {
      LongRay r;
      r.pix = (uint32_t)(py * P.w + px); r.n = n; r.max_n = max_n; r.prev = prev;
      r.x = pos.x; r.y = pos.y; r.z = pos.z; r.pad = 0.0f;
      longs[base + (uint32_t)__popcll(dm & ((1ull << ln) - 1ull))] = r;
    }
  }
  if (hit) {                                                            // approximate ray-cell intersection, :99-101
    if (partial && !prev_valid) prev = tex3d_tsdf<kSparse>(V, pos_prev.x, pos_prev.y, pos_prev.z);   // sample n-1 is a neighbour's: read it from the halo
    const float kk = prev / (hit_d - prev);
    pos = make_float3((hit_pos.x - step.x) - step.x * kk, (hit_pos.y - step.y) - step.y * kk, (hit_pos.z - step.z) - step.z * kk);
  }
  // compact the hits of this wave into the list (one atomic per wave)
  const unsigned long long hm = __ballot(hit);
  if (hm) {
    const int leader = __ffsll((long long)hm) - 1;
    uint32_t base = 0;
    if (ln == leader) base = atomicAdd(hit_count, (uint32_t)__popcll(hm));
    base = __shfl(base, leader);
    if (hit) {
      Hit h;
      h.x = pos.x; h.y = pos.y; h.z = pos.z; h.pix = (uint32_t)(py * P.w + px);
      hits[base + (uint32_t)__popcll(hm & ((1ull << ln) - 1ull))] = h;
    }
  }
  if (inside && !deferred) {
    if (!hit) {                                                         // discard: the target keeps its clear value
      const size_t oi = (size_t)py * R.stride + px;
      R.color[oi] = make_float4(R.clear[0], R.clear[1], R.clear[2], R.clear[3]);
      R.depth[oi] = 1.0f;
    }
    const float ns = (float)n * 0.0027f;                                // writeNumSamples(), :395-398: imageStore at ivec2(gl_FragCoord.xy)
    const int sx = px + P.vp_org[0], sy = py + P.vp_org[1];
    if (sx >= 0 && sy >= 0 && sx < P.w && sy < P.h) R.nsamples[(size_t)sy * P.w + sx] = covered ? ((partial && !hit) ? -ns : ns) : 0.0f;
  }
#ifdef RR_MARCH_STATS
  {
    const unsigned long long dt = wall_clock64() - t_start;             // 100 MHz ticks
    uint32_t m1 = st_pre, m2 = st_batches, m3 = max_n;
    for (int o = 32; o; o >>= 1) { m1 = max(m1, (uint32_t)__shfl_xor((int)m1, o)); m2 = max(m2, (uint32_t)__shfl_xor((int)m2, o)); m3 = max(m3, (uint32_t)__shfl_xor((int)m3, o)); }
    if (ln == 0) {
      atomicMax(&g_march_stats[1], (unsigned long long)m1);
      atomicMax(&g_march_stats[2], (unsigned long long)m2);
      atomicMax(&g_march_stats[3], (unsigned long long)m3);
      atomicMax(&g_march_stats[0], (dt << 32) | (unsigned long long)(uint32_t)(py * P.w + px));
      atomicAdd(&g_march_stats[4], 1ull);
      atomicAdd(&g_march_stats[5], dt);
    }
  }
#endif
}

// Dense march through LDS-staged voxel boxes (round 2; whole-volume dense storage, no depth limits: configs[1]).
// k_march gathers every tap of every sample from global memory: 8 scattered dword loads per lane and sample, each touching a
// handful of cache lines -- the texture-address path, not the vector ALUs (43 % issue) and not HBM (5 %), sets its 205 us at c1.
// But the 64 rays of a wave (one 8x8-pixel tile) are nearly parallel and step in lockstep: over S steps their taps stay inside one
// small axis-aligned box of voxels, and at 256^3 a voxel is tapped by ~4 rays x several steps.  Per batch of S steps a wave
//   1. takes the integer range of tap indices each lane will touch (from its first and last sample position of the batch) and
//      reduces min / max over the wave: the box [B0, B1] per axis (x origin aligned down to 4 voxels);
//   2. copies the box into its own LDS region with aligned 16-byte loads along x -- indices clamped on the way in, so a box cell IS
//      the GL CLAMP_TO_EDGE texel of its coordinate and the lanes need no clamps of their own;
//   3. marches its S samples with all eight taps from LDS.
// Positions are the reference's chain of `pos += step` additions and the trilinear filter is tsdf_fetch's (same operands, x -> y
// -> z): bit-identical.  The first / last positions only SIZE the box; a tap that falls outside it after all (rounding of the
// addition chain against pos + k * step, a NaN) is detected per sample and fetched from global memory instead, so exactness never
// rests on the box estimate.  S adapts (16, 8, 4, 2, 1) until the box fits kBoxFloats.
#ifndef RR_BOX_FLOATS
#define RR_BOX_FLOATS 2560
#endif
#ifndef RR_BOX_STEPS
#define RR_BOX_STEPS 16
#endif
#ifndef RR_BOX_SUB
#define RR_BOX_SUB 4
#endif
#ifndef RR_BOX_BOUNDS
#define RR_BOX_BOUNDS 2
#endif
constexpr int kBoxFloats = RR_BOX_FLOATS;     // per wave: 10 KiB -> 40 KiB per workgroup
constexpr int kBoxSteps = RR_BOX_STEPS;
#ifndef RR_BOX_LEAP
#define RR_BOX_LEAP 32
#endif
#ifndef RR_BOX_LDS_DMA
#define RR_BOX_LDS_DMA 1
#endif
constexpr int kLeap = RR_BOX_LEAP;            // samples per leap over all-clear tiles
#ifdef RR_BOX_STATS      // instrumented build (tools/build_variant.sh): [batches, batches with a box, sum of S, samples from LDS, samples from global, box floats, retries, samples in all-clear boxes]
__device__ unsigned long long g_box_stats[8];
extern "C" int32_t tsdf_debug_box_stats(unsigned long long out[8], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_box_stats), sizeof(g_box_stats)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_box_stats), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#define RR_STAT(i, v) do { if (ln == 0) atomicAdd(&g_box_stats[i], (unsigned long long)(v)); } while (0)
#define RR_STAT_LANE(i, v) atomicAdd(&g_box_stats[i], (unsigned long long)(v))
#else
#define RR_STAT(i, v) do { } while (0)
#define RR_STAT_LANE(i, v) do { } while (0)
#endif
// min over the 64 lanes, returned wave-uniform.  DPP row shifts + row broadcasts (the classic GCN wave reduction): eight VALU
// instructions and one v_readlane, no LDS round trips (six ds_bpermute per value cost the first version of this kernel ~0.3 us of
// exposed latency per batch).  Lanes without a source in a row shift keep their own value (old operand = self).
__device__ __forceinline__ int wave_min_i32(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));   // row_shr:1
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));   // row_shr:2
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xe, false));   // row_shr:4 (banks 1-3)
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xc, false));   // row_shr:8 (banks 2-3): lane 15 of each row holds the row's min
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3: lane 63 holds the wave's min
  return __builtin_amdgcn_readlane(v, 63);
}
// kDB (round 4): two boxes per wave -- while the samples of a batch are taken from one, the LDS-direct loads of the NEXT batch's box (sized from the
// positions the lanes will have after this batch: one multiply-add instead of the chain of additions, with margin; every sample still checks that its
// taps lie in the box it reads, so exactness does not rest on the prediction) are in flight into the other.  A 128-thread workgroup (16 x 8 pixels): two
// waves x 2 x 10 KiB.  What this kernel's length is made of is the longest chain of batches of any pixel tile (tools/box_stats.py: 0.9 box batches per wave on
// average, 15 - 25 for the tiles whose rays graze the surface), so a round trip hidden per batch shortens the launch.
template <bool kDB>
__global__ __launch_bounds__(kDB ? 128 : 256, kDB ? 2 : RR_BOX_BOUNDS) void k_march_box(ViewParams P, Volume V, RayTarget R, Hit* __restrict__ hits, uint32_t* __restrict__ hit_count) {
  __shared__ float4 s_all[kDB ? 2 : 4][(kDB ? 2 : 1) * kBoxFloats / 4];
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  float* s_box = (float*)s_all[wv];                                     // (kDB: the box in use; the other one is s_other)
  [[maybe_unused]] float* s_other = (float*)s_all[wv] + (kDB ? kBoxFloats : 0);
  const float limit = V.limit, sd = limit * 0.5f;
  const int px = blockIdx.x * 16 + (wv & 1) * 8 + (ln & 7);
  const int py = kDB ? blockIdx.y * 8 + (ln >> 3) : blockIdx.y * 16 + (wv >> 1) * 8 + (ln >> 3);
  const bool inside = px < P.w && py < P.h;
  const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
  const float3 dn = normalize3(pixel_dir_vol(P, fx, fy));
  const float3 step = make_float3(dn.x * sd, dn.y * sd, dn.z * sd);     // :64
  uint32_t max_n = 0;
  float3 pos = make_float3(0, 0, 0);
  bool covered = inside;
  if (inside) {                                                         // intersectBox(), :363-374
    const float3 o = make_float3(P.cam_vol[0], P.cam_vol[1], P.cam_vol[2]);
    const float3 inv = make_float3(1.0f / step.x, 1.0f / step.y, 1.0f / step.z);
    const float3 tbot = make_float3(inv.x * (0.0f - o.x), inv.y * (0.0f - o.y), inv.z * (0.0f - o.z));
    const float3 ttop = make_float3(inv.x * (1.0f - o.x), inv.y * (1.0f - o.y), inv.z * (1.0f - o.z));
    const float3 tmn = make_float3(fminf(ttop.x, tbot.x), fminf(ttop.y, tbot.y), fminf(ttop.z, tbot.z));
    const float3 tmx = make_float3(fmaxf(ttop.x, tbot.x), fmaxf(ttop.y, tbot.y), fmaxf(ttop.z, tbot.z));
    const float t0 = fmaxf(fmaxf(tmn.x, tmn.y), fmaxf(tmn.x, tmn.z));
    const float t1 = fminf(fminf(tmx.x, tmx.y), fminf(tmx.x, tmx.z));
    if (!(t0 <= t1) || t1 < 0.0f) covered = false;                      // no fragment: pixel not under the cube
    else {
      const float t_near = t0 < 0.0f ? 0.0f : t0;
      pos = make_float3(o.x + step.x * t_near, o.y + step.y * t_near, o.z + step.z * t_near);
      max_n = (uint32_t)ceilf(fabsf(t1 - t_near));
    }
  }
  const float nx = (float)V.res[0], ny = (float)V.res[1], nz = (float)V.res[2];
  float prev = -limit;
  uint32_t n = 0;
  bool hit = false;
  float3 hit_pos = pos;
  float hit_d = 0.0f;
  int S = kBoxSteps;
  [[maybe_unused]] bool have_pref = false, podd = false;               // (kDB) the other box holds / is receiving the next batch's voxels: its geometry and batch length
  [[maybe_unused]] int pbx0 = 0, pby0 = 0, pbz0 = 0, pex = 0, pey = 0, pez = 0, pS = 0;
  bool try_leap = true;                                                 // wave-uniform: the last thing seen was empty space
  const uint8_t* __restrict__ cls = V.cls;
  const float ml = -limit;
  while (__ballot(!hit && n < max_n) != 0ull) {                         // wave-uniform: the reductions and the box copy need every lane
    const bool live = !hit && n < max_n;
    // ---- 0. leap: integrate() leaves an exact class per 8^3 storage tile (kTileMinus = every voxel is -limit).  If all tiles the
    // wave's next kLeap samples can tap are of that class, those samples are -limit without looking at a single voxel: count, add.
    // One byte per tile and lane instead of a box of voxels; tried while the wave is in empty space, dropped at the first failure and
    // taken up again when a sampled box turns out all clear.
    if (try_leap) {
      const uint32_t rem = max_n - n;
      const float last = (float)((live ? min((uint32_t)kLeap, rem) : 1u) - 1u);
      const float ax = pos.x * nx - 0.5f, ay = pos.y * ny - 0.5f, az = pos.z * nz - 0.5f;
      const float bx = (pos.x + step.x * last) * nx - 0.5f, by = (pos.y + step.y * last) * ny - 0.5f, bz = (pos.z + step.z * last) * nz - 0.5f;
      const bool ok = live && fabsf(ax) < 1.0e6f && fabsf(ay) < 1.0e6f && fabsf(az) < 1.0e6f && fabsf(bx) < 1.0e6f && fabsf(by) < 1.0e6f && fabsf(bz) < 1.0e6f;
      constexpr float kSlackL = 0.05f;                                   // as kSlack below, for up to kLeap additions
      const int big = 0x3fffffff;
      // tile coordinates of the clamped tap range (CLAMP_TO_EDGE: taps outside the volume read its border voxels)
      const int rx = V.res[0] - 1, ry = V.res[1] - 1, rz = V.res[2] - 1;
      const int lx = ok ? min(max((int)floorf(fminf(ax, bx) - kSlackL), 0), rx) >> 3 : big, ly = ok ? min(max((int)floorf(fminf(ay, by) - kSlackL), 0), ry) >> 3 : big;
      const int lz = ok ? min(max((int)floorf(fminf(az, bz) - kSlackL), 0), rz) >> 3 : big;
      const int hx = ok ? min(max((int)floorf(fmaxf(ax, bx) + kSlackL) + 1, 0), rx) >> 3 : -big, hy = ok ? min(max((int)floorf(fmaxf(ay, by) + kSlackL) + 1, 0), ry) >> 3 : -big;
      const int hz = ok ? min(max((int)floorf(fmaxf(az, bz) + kSlackL) + 1, 0), rz) >> 3 : -big;
      const bool odd_l = __ballot(live && !ok) != 0ull;
      const int tx0 = wave_min_i32(lx), ty0 = wave_min_i32(ly), tz0 = wave_min_i32(lz);
      const int tx1 = -wave_min_i32(-hx), ty1 = -wave_min_i32(-hy), tz1 = -wave_min_i32(-hz);
      const int cx = tx1 - tx0 + 1, cy = ty1 - ty0 + 1, cz = tz1 - tz0 + 1;
      bool leap = !odd_l && tx0 != big && cx > 0 && cy > 0 && cz > 0 && cx <= 64 && cy <= 64 && cz <= 64 && __mul24(__mul24(cx, cy), cz) <= 4 * 64;
      if (leap) {
        const int cnt = __mul24(__mul24(cx, cy), cz);
        const float rcx = __builtin_amdgcn_rcpf((float)cx), rcy = __builtin_amdgcn_rcpf((float)cy);
        bool clear = true;
        for (int it = ln; it < cnt; it += 64) {
          const int row = (int)(((float)it + 0.5f) * rcx);               // it / cx
          const int kx = it - __mul24(row, cx);
          const int kz = (int)(((float)row + 0.5f) * rcy);               // row / cy
          const int ky = row - __mul24(kz, cy);
          clear &= cls[(uint32_t)__mul24(__mul24(tz0 + kz - V.tz0, V.nty) + (ty0 + ky), V.ntx) + (uint32_t)(tx0 + kx)] == kTileMinus;
        }
        leap = __ballot(!clear) == 0ull;
      }
      if (leap) {
        float3 e = pos;
        for (int k = 0; k < kLeap; ++k) e = make_float3(e.x + step.x, e.y + step.y, e.z + step.z);   // the reference's chain of additions
        if (live) { n += min((uint32_t)kLeap, max_n - n); prev = ml; }
        pos = e;
        RR_STAT(7, kLeap);
        continue;
      }
      try_leap = false;
    }
    // ---- 1. the box of this batch
    int bx0 = 0, by0 = 0, bz0 = 0, ex = 0, ey = 0, ez = 0;
    bool fits = false, odd = false;                                      // odd: some live lane has a non-finite position
    const bool prefetched = kDB && have_pref;                            // (wave-uniform) this batch's box was requested during the previous batch
    int s_batch = S;                                                     // the batch length the box was sized for
    if (prefetched) { bx0 = pbx0; by0 = pby0; bz0 = pbz0; ex = pex; ey = pey; ez = pez; fits = true; odd = podd; s_batch = pS; have_pref = false; }
    else for (;;) {
      const uint32_t rem = max_n - n;
      const float last = (float)((live ? min((uint32_t)S, rem) : 1u) - 1u);
      const float ax = pos.x * nx - 0.5f, ay = pos.y * ny - 0.5f, az = pos.z * nz - 0.5f;
      const float bx = (pos.x + step.x * last) * nx - 0.5f, by = (pos.y + step.y * last) * ny - 0.5f, bz = (pos.z + step.z * last) * nz - 0.5f;
      // lowest / highest tap index per axis; dead lanes and non-finite positions contribute nothing (they take the global path)
      const bool ok = live && fabsf(ax) < 1.0e6f && fabsf(ay) < 1.0e6f && fabsf(az) < 1.0e6f && fabsf(bx) < 1.0e6f && fabsf(by) < 1.0e6f && fabsf(bz) < 1.0e6f;
      const int big = 0x3fffffff;
      // kSlack voxels of margin: the samples are reached by S - 1 rounded additions, the estimate by one multiply-add; the two differ
      // by at most ~S ulps of a coordinate near 1, i.e. < 16 * 6e-8 * 4096 = 4e-3 voxels at the largest volume -- with the margin every
      // sample of a finite ray provably taps inside the box (the per-sample test of the sampling path below stays as a second line)
      constexpr float kSlack = 0.01f;
      const int lx = ok ? (int)floorf(fminf(ax, bx) - kSlack) : big, ly = ok ? (int)floorf(fminf(ay, by) - kSlack) : big, lz = ok ? (int)floorf(fminf(az, bz) - kSlack) : big;
      const int hx = ok ? (int)floorf(fmaxf(ax, bx) + kSlack) + 1 : -big, hy = ok ? (int)floorf(fmaxf(ay, by) + kSlack) + 1 : -big, hz = ok ? (int)floorf(fmaxf(az, bz) + kSlack) + 1 : -big;
      odd = __ballot(live && !ok) != 0ull;
      const int mlx = wave_min_i32(lx), mly = wave_min_i32(ly), mlz = wave_min_i32(lz);
      const int mhx = -wave_min_i32(-hx), mhy = -wave_min_i32(-hy), mhz = -wave_min_i32(-hz);
      if (mlx == big) { fits = false; break; }                          // no lane has a finite position: global path
      bx0 = mlx & ~3; by0 = mly; bz0 = mlz;
      ex = ((mhx - bx0 + 1) + 3) & ~3; ey = mhy - by0 + 1; ez = mhz - bz0 + 1;
      fits = ex > 0 && ey > 0 && ez > 0 && ex <= 1024 && ey <= 1024 && ez <= 1024 && __mul24(__mul24(ex, ey), ez) <= kBoxFloats;
      if (fits || S == 1) break;
      S >>= 1;
      RR_STAT(6, 1);
    }
    if (!prefetched) s_batch = S;
    RR_STAT(0, 1); RR_STAT(1, fits ? 1 : 0); RR_STAT(2, fits ? S : 0); RR_STAT(5, fits ? __mul24(__mul24(ex, ey), ez) : 0);
    // ---- 2. global -> LDS, clamped on the way in.  While it passes through the registers every voxel is compared with the clear
    // value: a box that holds nothing but -limit (free space in front of the surfaces, most of the volume) makes every sample inside
    // it exactly -limit -- lerp(a, a, t) = a + (a - a) * t = a in fp32 -- so step 3 only has to count and to perform the position
    // additions.  (The per-sample tile-class lookups of round 1 lost to their own instruction cost; here the test costs four compares
    // per 16-byte load and is shared by all samples of the batch.)
    bool all_clear = true;
    const float* __restrict__ d = V.data;
    // LDS-direct copy of a box whose x range lies inside the volume: the loads only (no wait)
    [[maybe_unused]] const auto dma_issue = [&](float* dst, int qx0, int qy0, int qz0, int qex, int qey, int qez) {
      const int q = qex >> 2, items = __mul24(__mul24(qey, qez), q);
      const float rq = __builtin_amdgcn_rcpf((float)q), rey = __builtin_amdgcn_rcpf((float)qey);
      for (int r = 0; (r << 6) < items; ++r) {
        const int it = ln + (r << 6);
        if (it < items) {
          const int row = (int)(((float)it + 0.5f) * rq);                  // it / q
          const int qi = it - __mul24(row, q);
          const int rz = (int)(((float)row + 0.5f) * rey);                 // row / ey
          const int ry = row - __mul24(rz, qey);
          const int gy = min(max(qy0 + ry, 0), V.res[1] - 1), gz = min(max(qz0 + rz, 0), V.res[2] - 1);
          const float* src = d + vol_off_y(V, gy) + vol_off_z(V, gz) + vol_off_x(qx0 + (qi << 2));
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)(dst + (r << 8)), 16, 0, 0);
        }
      }
    };
    if (fits) {
      const int q = ex >> 2, rows = __mul24(ey, ez), items = __mul24(rows, q);
      const float rq = __builtin_amdgcn_rcpf((float)q), rey = __builtin_amdgcn_rcpf((float)ey);
#if RR_BOX_LDS_DMA
      // Round 4: a box whose x range lies inside the volume (all but the boxes at its two x faces) is copied by LDS-direct loads (global_load_lds_dwordx4,
      // gfx950): item `it` of the box is the 16 bytes at s_box + 4 it -- lane-linear, which is exactly the layout such a load writes (LDS base in M0 + 16 B
      // per lane) -- so all rounds of a batch are in flight together and no voxel passes through a register.  (The register copy below waited for each
      // round's four dword loads before it issued the next: up to ten L2 round trips per batch, 40 % of this kernel's cycles in s_waitcnt.)
      if (prefetched || (bx0 >= 0 && bx0 + ex <= V.res[0])) {            // wave-uniform (a prefetched box is one of these)
        if (!prefetched) dma_issue(s_box, bx0, by0, bz0, ex, ey, ez);
        __builtin_amdgcn_s_waitcnt(0x0f70);                                // vmcnt(0): the box has landed
        __builtin_amdgcn_wave_barrier();
        for (int it = ln; it < items; it += 64) {                          // the clear test, from LDS
          const float4 v = *(const float4*)(s_box + (it << 2));
          all_clear &= (v.x == ml) & (v.y == ml) & (v.z == ml) & (v.w == ml);
        }
      } else
#endif
      for (int it = ln; it < items; it += 64) {
        const int row = (int)(((float)it + 0.5f) * rq);                  // it / q   (exact for it < 2^20, see k_integrate_tiles_lds)
        const int qi = it - __mul24(row, q);
        const int rz = (int)(((float)row + 0.5f) * rey);                 // row / ey
        const int ry = row - __mul24(rz, ey);
        const int gy = min(max(by0 + ry, 0), V.res[1] - 1), gz = min(max(bz0 + rz, 0), V.res[2] - 1);
        const uint32_t oyz = vol_off_y(V, gy) + vol_off_z(V, gz);
        const int gx = bx0 + (qi << 2);
        float4 v;
        if (gx >= 0 && gx + 3 < V.res[0]) v = *(const float4*)(d + oyz + vol_off_x(gx));          // four voxels of one tile row: 16-byte aligned
        else {
          const int c0 = min(max(gx, 0), V.res[0] - 1), c1 = min(max(gx + 1, 0), V.res[0] - 1), c2 = min(max(gx + 2, 0), V.res[0] - 1), c3 = min(max(gx + 3, 0), V.res[0] - 1);
          v = make_float4(d[oyz + vol_off_x(c0)], d[oyz + vol_off_x(c1)], d[oyz + vol_off_x(c2)], d[oyz + vol_off_x(c3)]);
        }
        *(float4*)(s_box + (__mul24(row, ex) + (qi << 2))) = v;
        all_clear &= (v.x == ml) & (v.y == ml) & (v.z == ml) & (v.w == ml);
      }
    }
    const bool empty = fits && __ballot(!all_clear) == 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- 3. S samples, kSub at a time: positions never depend on the densities, so the taps of kSub consecutive samples are read
    // together (their LDS latencies overlap) and then examined in order -- the same trick as k_march's batched fetches.
    // (No box at any S -- a tile whose rays entered the cube through different faces: eight samples straight from global memory,
    // then the box is tried again.)
    const int pl = __mul24(ex, ey);
    const int s_run = fits ? s_batch : 8;
    constexpr int kSub = RR_BOX_SUB;
    float3 p = pos;
    if (empty && !odd) {
      // every sample of the batch is -limit: no hit, prev_density = -limit, the counter advances by the samples this ray still had;
      // the positions still go through the reference's chain of additions
      for (int k = 0; k < s_run; ++k) p = make_float3(p.x + step.x, p.y + step.y, p.z + step.z);
      if (live) { n += min((uint32_t)s_run, max_n - n); prev = ml; }
      pos = p;
      RR_STAT(7, s_run);
      __builtin_amdgcn_wave_barrier();
      try_leap = true;                                                    // back in empty space
      continue;
    }
    if constexpr (kDB) {
      // ---- the next batch's box, requested now: the lanes that will still be live (a hit in this batch only removes lanes), at the positions one
      // multiply-add predicts (the chain of s_run additions differs by < s_run ulps: 2e-3 voxels at the largest volume; margin 0.02)
      const uint32_t done_now = min((uint32_t)s_run, max_n - n);
      const uint32_t n1 = n + (live ? done_now : 0u);
      const bool live1 = live && n1 < max_n;
      const float adv = (float)s_run;
      const float3 p1 = make_float3(pos.x + step.x * adv, pos.y + step.y * adv, pos.z + step.z * adv);
      const float last1 = (float)((live1 ? min((uint32_t)S, max_n - n1) : 1u) - 1u);
      const float ax = p1.x * nx - 0.5f, ay = p1.y * ny - 0.5f, az = p1.z * nz - 0.5f;
      const float bx = (p1.x + step.x * last1) * nx - 0.5f, by = (p1.y + step.y * last1) * ny - 0.5f, bz = (p1.z + step.z * last1) * nz - 0.5f;
      const bool ok = live1 && fabsf(ax) < 1.0e6f && fabsf(ay) < 1.0e6f && fabsf(az) < 1.0e6f && fabsf(bx) < 1.0e6f && fabsf(by) < 1.0e6f && fabsf(bz) < 1.0e6f;
      const int big = 0x3fffffff;
      constexpr float kSlackP = 0.02f;
      const int lx = ok ? (int)floorf(fminf(ax, bx) - kSlackP) : big, ly = ok ? (int)floorf(fminf(ay, by) - kSlackP) : big, lz = ok ? (int)floorf(fminf(az, bz) - kSlackP) : big;
      const int hx = ok ? (int)floorf(fmaxf(ax, bx) + kSlackP) + 1 : -big, hy = ok ? (int)floorf(fmaxf(ay, by) + kSlackP) + 1 : -big, hz = ok ? (int)floorf(fmaxf(az, bz) + kSlackP) + 1 : -big;
      podd = __ballot(live1 && !ok) != 0ull;
      const int mlx = wave_min_i32(lx), mly = wave_min_i32(ly), mlz = wave_min_i32(lz);
      const int mhx = -wave_min_i32(-hx), mhy = -wave_min_i32(-hy), mhz = -wave_min_i32(-hz);
      if (fits && mlx != big && !podd) {                                 // (only behind a batch that had a box itself: the steady state of a long march)
        pbx0 = mlx & ~3; pby0 = mly; pbz0 = mlz;
        pex = ((mhx - pbx0 + 1) + 3) & ~3; pey = mhy - pby0 + 1; pez = mhz - pbz0 + 1;
        const bool pfits = pex > 0 && pey > 0 && pez > 0 && pex <= 1024 && pey <= 1024 && pez <= 1024 && __mul24(__mul24(pex, pey), pez) <= kBoxFloats;
        if (pfits && pbx0 >= 0 && pbx0 + pex <= V.res[0]) { dma_issue(s_other, pbx0, pby0, pbz0, pex, pey, pez); have_pref = true; pS = S; }
      }
    }
    for (int k0 = 0; k0 < s_run; k0 += kSub) {
      float3 q[kSub];
      float dv[kSub];
      q[0] = p;
#pragma unroll
      for (int j = 1; j < kSub; ++j) q[j] = make_float3(q[j - 1].x + step.x, q[j - 1].y + step.y, q[j - 1].z + step.z);
      float wx[kSub], wy[kSub], wz[kSub];
      int cell[kSub];
      bool want[kSub];
      bool stray = false;                                                 // a wanted sample whose taps leave the box estimate
#pragma unroll
      for (int j = 0; j < kSub; ++j) {
        want[j] = !hit && n + (uint32_t)j < max_n && k0 + j < s_run;
        const float fxv = q[j].x * nx - 0.5f, fyv = q[j].y * ny - 0.5f, fzv = q[j].z * nz - 0.5f;    // axis_linear: f = u * n - 0.5
        const float flx = floorf(fxv), fly = floorf(fyv), flz = floorf(fzv);
        wx[j] = fxv - flx; wy[j] = fyv - fly; wz[j] = fzv - flz;
        const int ix = (int)flx - bx0, iy = (int)fly - by0, iz = (int)flz - bz0;
        const bool in_box = (unsigned)ix < (unsigned)(ex - 1) && (unsigned)iy < (unsigned)(ey - 1) && (unsigned)iz < (unsigned)(ez - 1);
        stray |= want[j] && !in_box;
        cell[j] = (want[j] && in_box) ? __mul24(iz, pl) + __mul24(iy, ex) + ix : 0;
      }
      if (fits && __ballot(stray) == 0ull) {
        // the common case, free of divergence: every lane reads eight taps per sample (idle lanes read cell 0), all reads of the
        // sub-batch are issued before the first lerp
        float t[kSub][8];
#pragma unroll
        for (int j = 0; j < kSub; ++j) {
          const float* b = s_box + cell[j];
          t[j][0] = b[0]; t[j][1] = b[1]; t[j][2] = b[ex]; t[j][3] = b[ex + 1];
          t[j][4] = b[pl]; t[j][5] = b[pl + 1]; t[j][6] = b[pl + ex]; t[j][7] = b[pl + ex + 1];
        }
#pragma unroll
        for (int j = 0; j < kSub; ++j) {
          const float c00 = lerpf(t[j][0], t[j][1], wx[j]), c10 = lerpf(t[j][2], t[j][3], wx[j]);
          const float c01 = lerpf(t[j][4], t[j][5], wx[j]), c11 = lerpf(t[j][6], t[j][7], wx[j]);
          dv[j] = lerpf(lerpf(c00, c10, wy[j]), lerpf(c01, c11, wy[j]), wz[j]);
        }
        RR_STAT(3, kSub);
      } else {
        // no box, or some lane's taps fall outside the estimate (rounding of the addition chain, a NaN): the whole sub-batch from
        // global memory -- the same operands, the same result
#pragma unroll
        for (int j = 0; j < kSub; ++j) dv[j] = want[j] ? tex3d_tsdf<false, true>(V, q[j].x, q[j].y, q[j].z) : 0.0f;
        RR_STAT(4, kSub);
      }
#pragma unroll
      for (int j = 0; j < kSub; ++j) {
        if (!hit && n < max_n && k0 + j < s_run) {
          n += 1;
          if (dv[j] > 0.0f) { hit = true; hit_pos = q[j]; hit_d = dv[j]; }
          else prev = dv[j];
        }
      }
      p = make_float3(q[kSub - 1].x + step.x, q[kSub - 1].y + step.y, q[kSub - 1].z + step.z);
#pragma unroll
      for (int j = kSub - 1; j >= 1; --j) if (s_run - k0 == j) p = q[j];     // a last, shorter sub-batch: the chain stops after j additions
    }
    pos = p;
    __builtin_amdgcn_wave_barrier();                                      // the next batch overwrites the box
    if (kDB && have_pref) { float* t = s_box; s_box = s_other; s_other = t; }   // the box being filled becomes the box in use
    if (fits && S < kBoxSteps && (n & 63u) == 0u) S <<= 1;                // try a longer batch again now and then (wave-uniform only if n is: see below)
    S = __builtin_amdgcn_readfirstlane(S);
  }
  if (kDB && have_pref) __builtin_amdgcn_s_waitcnt(0x0f70);             // a box still on its way must have landed before the wave (and its LDS) goes
  float3 out_pos = pos;
  if (hit) {                                                            // approximate ray-cell intersection, :99-101
    const float kk = prev / (hit_d - prev);
    out_pos = make_float3((hit_pos.x - step.x) - step.x * kk, (hit_pos.y - step.y) - step.y * kk, (hit_pos.z - step.z) - step.z * kk);
  }
  const unsigned long long hm = __ballot(hit);
  if (hm) {
    const int leader = __ffsll((long long)hm) - 1;
    uint32_t base = 0;
    if (ln == leader) base = atomicAdd(hit_count, (uint32_t)__popcll(hm));
    base = __shfl(base, leader);
    if (hit) {
      Hit h;
      h.x = out_pos.x; h.y = out_pos.y; h.z = out_pos.z; h.pix = (uint32_t)(py * P.w + px);
      hits[base + (uint32_t)__popcll(hm & ((1ull << ln) - 1ull))] = h;
    }
  }
  if (inside) {
    if (!hit) {                                                         // discard: the target keeps its clear value
      const size_t oi = (size_t)py * R.stride + px;
      R.color[oi] = make_float4(R.clear[0], R.clear[1], R.clear[2], R.clear[3]);
      R.depth[oi] = 1.0f;
    }
    const float ns = (float)n * 0.0027f;                                // writeNumSamples(), :395-398
    const int sx = px + P.vp_org[0], sy = py + P.vp_org[1];
    if (sx >= 0 && sy >= 0 && sx < P.w && sy < P.h) R.nsamples[(size_t)sy * P.w + sx] = covered ? ns : 0.0f;
  }
}

// Second pass of the march: kLanes lanes per long ray, kLongBatch consecutive samples per lane (kLanes * kLongBatch samples of a
// ray per round trip, 64 / kLanes rays per wave; shapes measured in DESIGN.md section 4).  Lane j starts 8j samples further along the ray; it gets there by performing the reference's own
// chain of `pos += step` additions, so positions, densities, the first positive sample and the sample count are exactly
// those of the one-at-a-time loop (:89-110).  A single lane walking a 92-sample ray issues ~14 k dependent instructions;
// here the same ray is two rounds of ~1.4 k.
struct StreamTable; struct FrameImages;
template <bool kSparse>
__device__ __forceinline__ void shade_hit(const ViewParams& P, const StreamTable& T, const FrameImages& F, const Volume& V, const RayTarget& R, const Hit& h);
template <bool kSparse>
__device__ __forceinline__ void march_long(const ViewParams& P, const Volume& V, const RayTarget& R, const LongRay* __restrict__ longs,
                                           const uint32_t* __restrict__ long_count, uint32_t n_blocks, const StreamTable* T, const FrameImages* F) {
  constexpr int kLanes = RR_LONG_LANES, kRays = 64 / kLanes;             // lanes per ray, rays per wave
  const int ln = threadIdx.x & 63, g = ln / kLanes, j = ln % kLanes;
  const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = n_blocks * 4u;
  const uint32_t count = *long_count;
  const float sd = V.limit * 0.5f;
  for (uint32_t r0 = wave * (uint32_t)kRays; r0 < count; r0 += n_waves * (uint32_t)kRays) {      // wave-uniform loop: the shuffles below need every lane
    const uint32_t r = r0 + (uint32_t)g;
    const bool live = r < count;
    LongRay L = longs[live ? r : r0];
    const int px = (int)(L.pix % (uint32_t)P.w), py = (int)(L.pix / (uint32_t)P.w);
    const float3 dn = normalize3(pixel_dir_vol(P, (float)px + 0.5f, (float)py + 0.5f));
    const float3 step = make_float3(dn.x * sd, dn.y * sd, dn.z * sd);
    float3 pos = make_float3(L.x, L.y, L.z);
    float prev = L.prev;
    uint32_t n = L.n;
    bool done = !live, hit = false;
    float3 hit_pos = pos;
    float hit_d = 0.0f;
    while (__ballot(!done)) {
      float3 p[kLongBatch];
      float d[kLongBatch];
      p[0] = pos;
      for (int t = 0; t < j; ++t) {
#pragma unroll
        for (int k = 0; k < kLongBatch; ++k) p[0] = make_float3(p[0].x + step.x, p[0].y + step.y, p[0].z + step.z);
      }
#pragma unroll
      for (int k = 1; k < kLongBatch; ++k) p[k] = make_float3(p[k - 1].x + step.x, p[k - 1].y + step.y, p[k - 1].z + step.z);
#pragma unroll
      for (int k = 0; k < kLongBatch; ++k) d[k] = tex3d_tsdf<kSparse, true>(V, p[k].x, p[k].y, p[k].z);   // (the long pass only exists for whole-volume contexts)     // unconditional: taps are clamped into the allocation
      // examine this lane's eight samples in order
      bool lhit = false, lprev_set = false;
      float lprev = 0.0f, lhd = 0.0f;
      float3 lhp = pos;
      uint32_t lk = 0;
#pragma unroll
      for (int k = 0; k < kLongBatch; ++k) {
        const bool valid = n + (uint32_t)(kLongBatch * j + k) < L.max_n;
        if (!lhit && valid) {
          if (d[k] > 0.0f) { lhit = true; lk = (uint32_t)k; lhp = p[k]; lhd = d[k]; }
          else { lprev = d[k]; lprev_set = true; }
        }
      }
      // the ray's first hit is in the lowest lane of its group that found one
      const unsigned long long gm = (__ballot(lhit && !done) >> (kLanes * g)) & (kLanes >= 64 ? ~0ull : ((1ull << (kLanes & 63)) - 1ull));
      const float d7_before = __shfl(d[kLongBatch - 1], ln > 0 ? ln - 1 : 0);               // last sample of the lane one batch earlier
      const int last = kLanes * g + kLanes - 1;
      const float d7_last = __shfl(d[kLongBatch - 1], last);
      const float3 p7_last = make_float3(__shfl(p[kLongBatch - 1].x, last), __shfl(p[kLongBatch - 1].y, last), __shfl(p[kLongBatch - 1].z, last));
      if (!done) {
        if (gm) {
          const int jw = __ffsll((long long)gm) - 1;
          // broadcast the winner's result to the group (every lane of the group keeps a copy; lane 0 writes it out)
          const int src = kLanes * g + jw;
          const float wprev = lprev_set ? lprev : (j > 0 ? d7_before : prev);
          prev = __shfl(wprev, src);
          hit_pos = make_float3(__shfl(lhp.x, src), __shfl(lhp.y, src), __shfl(lhp.z, src));
          hit_d = __shfl(lhd, src);
          n += (uint32_t)(kLongBatch * jw) + __shfl(lk, src) + 1u;
          hit = true; done = true;
        } else if (L.max_n - n <= (uint32_t)(kLanes * kLongBatch)) {
          n = L.max_n; done = true;
        } else {
          prev = d7_last;
          pos = make_float3(p7_last.x + step.x, p7_last.y + step.y, p7_last.z + step.z);
          n += (uint32_t)(kLanes * kLongBatch);
        }
      }
    }
    if (live && j == 0) {
      if (hit) {                                                        // approximate ray-cell intersection, :99-101
        const float kk = prev / (hit_d - prev);
        Hit h;
        h.x = (hit_pos.x - step.x) - step.x * kk; h.y = (hit_pos.y - step.y) - step.y * kk; h.z = (hit_pos.z - step.z) - step.z * kk;
        h.pix = L.pix;
        shade_hit<kSparse>(P, *T, *F, V, R, h);                         // no list: the hit is shaded where it was found
      } else {
        const size_t oi = (size_t)py * R.stride + px;
        R.color[oi] = make_float4(R.clear[0], R.clear[1], R.clear[2], R.clear[3]);
        R.depth[oi] = 1.0f;
      }
      const int sx = px + P.vp_org[0], sy = py + P.vp_org[1];           // writeNumSamples(), :395-398
      if (sx >= 0 && sy >= 0 && sx < P.w && sy < P.h) R.nsamples[(size_t)sy * P.w + sx] = (float)n * 0.0027f;
    }
  }
}

// submitFragment(), :116-134, one thread per hit.  Thread 0 also re-arms the OTHER hit counter for the next frame.
template <bool kSparse>
__device__ __forceinline__ void shade_hit(const ViewParams& P, const StreamTable& T, const FrameImages& F, const Volume& V, const RayTarget& R, const Hit& h) {
  const float limit = V.limit, sd = limit * 0.5f;
  const float3 pos = make_float3(h.x, h.y, h.z);
  const int px = (int)(h.pix % (uint32_t)P.w), py = (int)(h.pix / (uint32_t)P.w);
  const float gx = tex3d_tsdf<kSparse>(V, pos.x + sd, pos.y, pos.z) - tex3d_tsdf<kSparse>(V, pos.x - sd, pos.y, pos.z);
  const float gy = tex3d_tsdf<kSparse>(V, pos.x, pos.y + sd, pos.z) - tex3d_tsdf<kSparse>(V, pos.x, pos.y - sd, pos.z);
  const float gz = tex3d_tsdf<kSparse>(V, pos.x, pos.y, pos.z + sd) - tex3d_tsdf<kSparse>(V, pos.x, pos.y, pos.z - sd);
  const float3 gn = normalize3(make_float3(gx, gy, gz));              // get_gradient(), :140-149
  const float4 vn4 = mat_mul(P.normal, -gn.x, -gn.y, -gn.z, 0.0f);
  const float3 vn = normalize3(make_float3(vn4.x, vn4.y, vn4.z));
  const float4 vp4 = mat_mul(P.mv_v2w, pos.x, pos.y, pos.z, 1.0f);
  const float3 vp = make_float3(vp4.x, vp4.y, vp4.z);
  float4 col;
  if (P.shade_mode == 3) {
    const float3 bc = blend_cameras(T, F, limit, pos);
    col = make_float4(bc.x, bc.y, bc.z, 1.0f);
  } else {
    const float4 dc = blend_colors(T, F, limit, pos);
    const float3 s = shade(P, vp, vn, make_float3(dc.x, dc.y, dc.z));
    col = make_float4(s.x, s.y, s.z, dc.w);
  }
  float fd = (P.proj.m[10] * vp.z + P.proj.m[14]) / -vp.z * 0.5f + 0.5f;   // gl_FragDepth, :133
  fd = fminf(fmaxf(fd, 0.0f), 1.0f);
  const size_t oi = (size_t)py * R.stride + px;
  const bool pass = fd < 1.0f;                                        // GL_LESS against the cleared 1.0
  R.color[oi] = pass ? col : make_float4(R.clear[0], R.clear[1], R.clear[2], R.clear[3]);
  R.depth[oi] = pass ? fd : 1.0f;
}

// the hit list of the first march pass, one thread per hit, blocks [first_block, gridDim.x)
template <bool kSparse>
__device__ __forceinline__ void shade_list(const ViewParams& P, const StreamTable& T, const FrameImages& F, const Volume& V, const RayTarget& R,
                                           const Hit* __restrict__ hits, const uint32_t* __restrict__ hit_count, uint32_t* __restrict__ next_count, uint32_t first_block) {
  const uint32_t b = blockIdx.x - first_block, nb = gridDim.x - first_block;
  if (b == 0 && threadIdx.x == 0) { next_count[0] = 0u; next_count[2] = 0u; }   // hit + long-ray counters of the next frame
  const uint32_t n_hits = *hit_count;
  for (uint32_t i = b * blockDim.x + threadIdx.x; i < n_hits; i += nb * blockDim.x) shade_hit<kSparse>(P, T, F, V, R, hits[i]);
}
#ifndef RR_SHADE_BOUNDS
#define RR_SHADE_BOUNDS 1
#endif
template <bool kSparse>
__global__ __launch_bounds__(256, RR_SHADE_BOUNDS) void k_shade(ViewParams P, StreamTable T, FrameImages F, Volume V, RayTarget R, const Hit* __restrict__ hits,
                                               const uint32_t* __restrict__ hit_count, uint32_t* __restrict__ next_count) {
  shade_list<kSparse>(P, T, F, V, R, hits, hit_count, next_count, 0u);
}
// ONE launch for the two independent jobs that follow the first march pass: blocks [0, kLongBlocks) finish the long rays
// (and shade their own hits on the spot), the rest shade the first pass's hit list.  Saves a dependent launch (~5 us of
// ramp) and overlaps two latency-bound kernels: 15 + 19 us -> ~21 us.
#ifndef RR_LONG_BLOCKS
#define RR_LONG_BLOCKS 512
#endif
#ifndef RR_SHADE_BLOCKS
#define RR_SHADE_BLOCKS 1024
#endif
constexpr uint32_t kLongBlocks = RR_LONG_BLOCKS, kShadeBlocks = RR_SHADE_BLOCKS;
template <bool kSparse>
__global__ __launch_bounds__(256, RR_SHADE_BOUNDS) void k_shade_and_long(ViewParams P, StreamTable T, FrameImages F, Volume V, RayTarget R, const Hit* __restrict__ hits,
                                                        const uint32_t* __restrict__ hit_count, uint32_t* __restrict__ next_count,
                                                        const LongRay* __restrict__ longs, const uint32_t* __restrict__ long_count) {
  if (blockIdx.x < kLongBlocks) march_long<kSparse>(P, V, R, longs, long_count, kLongBlocks, &T, &F);
  else shade_list<kSparse>(P, T, F, V, R, hits, hit_count, next_count, kLongBlocks);
}
void launch_raymarch(hipStream_t st, const ViewParams& P, const StreamTable& T, const FrameImages& F, const Volume& V, const RayTarget& R, int partial,
                     void* hit_list, uint32_t* hit_counters, int parity, int phase, void* long_list, uint32_t cap, int box_march) {
  // phase 2: k_march alone; phase 3: k_shade alone; 0: everything (the split lets the caller time the march kernel alone)
  dim3 grid((P.w + 15) / 16, (P.h + 15) / 16);
  // counters: [hit parity 0, hit parity 1, long parity 0, long parity 1]
  const bool two_pass = !partial && P.skip && long_list && cap != 0xffffffffu;
  const bool sparse = V.slot != nullptr;
  if (phase != 3) {
    const uint32_t cap1 = two_pass ? cap : 0xffffffffu;
    LongRay* const ll = partial ? nullptr : (LongRay*)long_list;
#define RR_LAUNCH_MARCH(PART, SP, B) hipLaunchKernelGGL((k_march<PART, SP, B>), grid, dim3(256), 0, st, P, V, R, (Hit*)hit_list, hit_counters + parity, ll, hit_counters + 2 + parity, cap1)
    if (partial) { if (sparse) RR_LAUNCH_MARCH(true, true, kBatchDense); else RR_LAUNCH_MARCH(true, false, kBatchDense); }
    else if (two_pass) { if (sparse) RR_LAUNCH_MARCH(false, true, kBatchSkip); else RR_LAUNCH_MARCH(false, false, kBatchSkip); }
    else if (!sparse && !P.skip && box_march == 2) hipLaunchKernelGGL(k_march_box<true>, dim3(grid.x, (P.h + 7) / 8), dim3(128), 0, st, P, V, R, (Hit*)hit_list, hit_counters + parity);
    else if (!sparse && !P.skip && box_march) hipLaunchKernelGGL(k_march_box<false>, grid, dim3(256), 0, st, P, V, R, (Hit*)hit_list, hit_counters + parity);
    else { if (sparse) RR_LAUNCH_MARCH(false, true, kBatchDense); else RR_LAUNCH_MARCH(false, false, kBatchDense); }
#undef RR_LAUNCH_MARCH
  }
  if (phase == 2) return;
  if (two_pass) {
    if (sparse) hipLaunchKernelGGL(k_shade_and_long<true>, dim3(kLongBlocks + kShadeBlocks), dim3(256), 0, st, P, T, F, V, R, (const Hit*)hit_list, hit_counters + parity, hit_counters + (parity ^ 1),
                                   (const LongRay*)long_list, hit_counters + 2 + parity);
    else hipLaunchKernelGGL(k_shade_and_long<false>, dim3(kLongBlocks + kShadeBlocks), dim3(256), 0, st, P, T, F, V, R, (const Hit*)hit_list, hit_counters + parity, hit_counters + (parity ^ 1),
                            (const LongRay*)long_list, hit_counters + 2 + parity);
    return;
  }
  if (sparse) hipLaunchKernelGGL(k_shade<true>, dim3(1024), dim3(256), 0, st, P, T, F, V, R, (const Hit*)hit_list, hit_counters + parity, hit_counters + (parity ^ 1));
  else hipLaunchKernelGGL(k_shade<false>, dim3(1024), dim3(256), 0, st, P, T, F, V, R, (const Hit*)hit_list, hit_counters + parity, hit_counters + (parity ^ 1));
}

// ------------------------------------------------------------------------------------------- multi-GPU image exchange
// partial image: [rgba 16 B][depth 4 B][nsamples 4 B] planes of w*h pixels
__global__ __launch_bounds__(256) void k_export_partial(RayTarget R, int w, int h, float4* __restrict__ c, float* __restrict__ d, float* __restrict__ ns) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  const int x = i % w, y = i / w;
  c[i] = R.color[(size_t)y * R.stride + x];
  d[i] = R.depth[(size_t)y * R.stride + x];
  ns[i] = R.nsamples[i];
}
void launch_export_partial(hipStream_t st, const RayTarget& R, int w, int h, void* dst) {
  const size_t n = (size_t)w * h;
  float4* c = (float4*)dst;
  float* d = (float*)((char*)dst + n * 16);
  float* ns = d + n;
  hipLaunchKernelGGL(k_export_partial, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, R, w, h, c, d, ns);
}
// Per pixel: the slab whose ray segment hit first (smallest positive sample count) wins; no hit anywhere:
// clear colour, and the common miss count.  Exact because every rank steps the same global ray.
__global__ __launch_bounds__(256) void k_composite(const char* __restrict__ g, int nr, RayTarget R, int w, int h) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)w * h;
  if (i >= (int)n) return;
  int best = -1;
  float best_ns = 0.0f, miss_ns = 0.0f;
  for (int r = 0; r < nr; ++r) {
    const char* base = g + (size_t)r * n * 24;
    const float ns = ((const float*)(base + n * 20))[i];
    if (ns > 0.0f) { if (best < 0 || ns < best_ns) { best = r; best_ns = ns; } }
    else miss_ns = fmaxf(miss_ns, -ns);
  }
  const int x = i % w, y = i / w;
  float4 c = make_float4(R.clear[0], R.clear[1], R.clear[2], R.clear[3]);
  float d = 1.0f;
  if (best >= 0) {
    const char* base = g + (size_t)best * n * 24;
    c = ((const float4*)base)[i];
    d = ((const float*)(base + n * 16))[i];
  }
  R.color[(size_t)y * R.stride + x] = c;
  R.depth[(size_t)y * R.stride + x] = d;
  R.nsamples[i] = best >= 0 ? best_ns : miss_ns;
}
void launch_composite(hipStream_t st, const void* gathered, int n, const RayTarget& R, int w, int h) {
  const size_t np = (size_t)w * h;
  hipLaunchKernelGGL(k_composite, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, (const char*)gathered, n, R, w, h);
}

}  // namespace rr

namespace rr {

// ---- compact exchange: instead of 24 B for every view pixel, a slab ships one 32-byte record per ray that hit in it
struct HitRecord { uint32_t pix; float ns; float depth; uint32_t pad; float4 color; };
static_assert(sizeof(HitRecord) == 32, "HitRecord is 32 bytes");
// buffer = [written, hit, overflow, 5 pad words][records...]
// A whole-volume context marches in two passes: the rays of the second (wave-per-ray) pass are not on the hit list -- they were shaded,
// or cleared, where the pass ended.  Their list is still there, so they are exported too, every one of them: a long ray that missed
// ships (clear colour, depth 1, its sample count), which is what the pixel holds in the unpartitioned frame as well.
__global__ __launch_bounds__(256) void k_export_hits(RayTarget R, int w, const Hit* __restrict__ hits, const uint32_t* __restrict__ hit_count,
                                                     const LongRay* __restrict__ longs, const uint32_t* __restrict__ long_count,
                                                     uint32_t* __restrict__ header, HitRecord* __restrict__ rec, uint32_t capacity) {
  const uint32_t n_first = *hit_count, n_hits = n_first + (longs ? *long_count : 0u), n = n_hits < capacity ? n_hits : capacity;
  // header: [records written, rays that hit (may exceed the capacity: the receiver then asks again with a larger one), overflow flag]
  if (blockIdx.x == 0 && threadIdx.x == 0) { header[0] = n; header[1] = n_hits; header[2] = n_hits > capacity ? 1u : 0u; }
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t pix = i < n_first ? hits[i].pix : longs[i - n_first].pix;
    const int px = (int)(pix % (uint32_t)w), py = (int)(pix / (uint32_t)w);
    HitRecord r;
    r.pix = pix; r.ns = R.nsamples[pix]; r.depth = R.depth[(size_t)py * R.stride + px]; r.pad = 0u;
    r.color = R.color[(size_t)py * R.stride + px];
    rec[i] = r;
  }
}
void launch_export_hits(hipStream_t st, const RayTarget& R, int w, const void* hit_list, const uint32_t* hit_count, const void* long_list, const uint32_t* long_count,
                        void* dst, uint32_t capacity) {
  hipLaunchKernelGGL(k_export_hits, dim3(512), dim3(256), 0, st, R, w, (const Hit*)hit_list, hit_count, (const LongRay*)long_list, long_count, (uint32_t*)dst,
                     (HitRecord*)((char*)dst + 32), capacity);
}

// composite on the gathering rank: (1) every pixel starts as "no rank hit" (clear colour, depth 1, the common miss count),
// (2) every record bids for its pixel with the key (sample count, rank, index) -- the smallest sample count is the first zero
// crossing along the ray --, (3) the winning record writes the pixel.
__global__ __launch_bounds__(256) void k_comp_init(RayTarget R, int w, int h, unsigned long long* __restrict__ key, int own_counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  const int x = i % w, y = i / w;
  R.color[(size_t)y * R.stride + x] = make_float4(R.clear[0], R.clear[1], R.clear[2], R.clear[3]);
  R.depth[(size_t)y * R.stride + x] = 1.0f;
  R.nsamples[i] = own_counts ? fabsf(R.nsamples[i]) : 0.0f;   // misses carry -count in slab mode (a context that did not march has none); a hit pixel is overwritten in (3)
  key[i] = ~0ull;
}
__device__ __forceinline__ unsigned long long hit_key(float ns, uint32_t rank, uint32_t idx) {
  return ((unsigned long long)__float_as_uint(ns) << 32) | ((unsigned long long)rank << 27) | idx;
}
__global__ __launch_bounds__(256) void k_comp_bid(const char* __restrict__ g, size_t stride_bytes, int n_ranks, unsigned long long* __restrict__ key) {
  const int r = blockIdx.y;
  if (r >= n_ranks) return;
  const uint32_t* header = (const uint32_t*)(g + (size_t)r * stride_bytes);
  const HitRecord* rec = (const HitRecord*)(g + (size_t)r * stride_bytes + 32);
  const uint32_t n = header[0];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if (rec[i].ns > 0.0f) atomicMin(&key[rec[i].pix], hit_key(rec[i].ns, (uint32_t)r, i));
}
__global__ __launch_bounds__(256) void k_comp_write(const char* __restrict__ g, size_t stride_bytes, int n_ranks, const unsigned long long* __restrict__ key,
                                                    RayTarget R, int w) {
  const int r = blockIdx.y;
  if (r >= n_ranks) return;
  const uint32_t* header = (const uint32_t*)(g + (size_t)r * stride_bytes);
  const HitRecord* rec = (const HitRecord*)(g + (size_t)r * stride_bytes + 32);
  const uint32_t n = header[0];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const HitRecord h = rec[i];
    if (!(h.ns > 0.0f) || key[h.pix] != hit_key(h.ns, (uint32_t)r, i)) continue;
    const int x = (int)(h.pix % (uint32_t)w), y = (int)(h.pix / (uint32_t)w);
    R.color[(size_t)y * R.stride + x] = h.color;
    R.depth[(size_t)y * R.stride + x] = h.depth;
    R.nsamples[h.pix] = h.ns;
  }
}
void launch_composite_hits(hipStream_t st, const void* gathered, size_t stride_bytes, int n, const RayTarget& R, int w, int h, unsigned long long* key, int own_counts) {
  const int np = w * h;
  hipLaunchKernelGGL(k_comp_init, dim3((np + 255) / 256), dim3(256), 0, st, R, w, h, key, own_counts);
  hipLaunchKernelGGL(k_comp_bid, dim3(128, n), dim3(256), 0, st, (const char*)gathered, stride_bytes, n, key);
  hipLaunchKernelGGL(k_comp_write, dim3(128, n), dim3(256), 0, st, (const char*)gathered, stride_bytes, n, key, R, w);
}

}  // namespace rr
