// K5 + K2: brick depth limits (bricks.vs/gs/fs via drawDepthLimits(), recon_integration.cpp:408-428)
// and the TSDF raymarch (tsdf_raymarch.fs:62-134 via draw(), :176-240).
//
// The reference rasterises every exposed face of every occupied brick with MIN blending to get, per
// pixel, (min z, -max z, min back-face z).  Here one wave per occupied brick scatters the same values
// with integer atomics on the float bit patterns (order independent, hence deterministic); the work is
// proportional to the occupied bricks' screen footprint, not to rays x grid cells.  The raymarch then runs
// one ray per pixel (the reference shades the front and the back cube face with identical results,
// SURVEY.md Appendix C.3).
#include "sampling.hpp"

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

// One wave per occupied brick (persistent, pulling from the compacted list).  For each exposed face the wave
// sweeps the pixel bounding box of the projected face and applies the reference's fragment rule to every pixel
// centre whose ray crosses the face rectangle: z from the ray/plane intersection, near/far clip, MIN blend.
__global__ __launch_bounds__(64) void k_depth_limits(ViewParams P, Bricks B, uint4* __restrict__ peels) {
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
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int a1 = (a + 1) % 3, a2 = (a + 2) % 3;
#pragma unroll
      for (int dir = -1; dir <= 1; dir += 2) {
        if (neighbour_gt10(B, idx[0], idx[1], idx[2], a, dir)) continue;  // shared face culled in the GS, bricks.gs:26-43
        const float coord = dir < 0 ? lo[a] : hi[a];
        // pixel bounding box of the face (conservative: +-1 px; whole screen if a corner is behind the eye)
        float bx0 = 3.0e38f, bx1 = -3.0e38f, by0 = 3.0e38f, by1 = -3.0e38f;
        bool behind = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float hp[3];
          hp[a] = coord; hp[a1] = (c & 1) ? hi[a1] : lo[a1]; hp[a2] = (c & 2) ? hi[a2] : lo[a2];
          const float4 e = mat_mul(P.mv, hp[0], hp[1], hp[2], 1.0f);
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
        const bool front = dir > 0 ? (o[a] > coord) : (o[a] < coord);     // gl_FrontFacing of an outward-wound cube
        for (int k = lane; k < bw * bh; k += 64) {
          const int px = x0 + k % bw, py = y0 + k / bw;
          const float3 dw = pixel_dir_world(P, (float)px + 0.5f, (float)py + 0.5f);
          const float d[3] = {dw.x, dw.y, dw.z};
          const float t = (coord - o[a]) / d[a];
          if (!(t > 0.0f)) continue;
          const float q1 = o[a1] + t * d[a1], q2 = o[a2] + t * d[a2];
          if (q1 < lo[a1] || q1 > hi[a1] || q2 < lo[a2] || q2 > hi[a2]) continue;
          float hp[3];
          hp[a] = coord; hp[a1] = q1; hp[a2] = q2;
          const float ez = P.mv.m[2] * hp[0] + P.mv.m[6] * hp[1] + P.mv.m[10] * hp[2] + P.mv.m[14] * 1.0f;
          const float zw = (P.proj.m[10] * ez + P.proj.m[14]) / (-ez) * 0.5f + 0.5f;
          if (!(zw >= 0.0f && zw <= 1.0f)) continue;                      // near / far clip
          uint32_t* pp = (uint32_t*)&peels[(size_t)py * P.w + px];
          const uint32_t zb = __float_as_uint(zw);
          atomicMin(pp + 0, zb);                                          // bricks.fs:6 with GL_MIN blending
          atomicMax(pp + 1, zb);
          if (!front) atomicMin(pp + 2, zb);
        }
      }
    }
  }
}
void launch_depth_limits(hipStream_t st, const ViewParams& P, const Bricks& B, float4* peels) {
  const int n = P.w * P.h;
  hipLaunchKernelGGL(k_clear_peels, dim3((n + 255) / 256), dim3(256), 0, st, (uint4*)peels, n);
  const int grid = B.n < 8192 ? B.n : 8192;
  hipLaunchKernelGGL(k_depth_limits, dim3(grid), dim3(64), 0, st, P, B, (uint4*)peels);
}

// ------------------------------------------------------------------------------------------- K2
__constant__ float c_camera_colors[8][3] = {   // shading.glsl:24-30, extended past 5 streams
    {228 / 255.0f, 26 / 255.0f, 28 / 255.0f}, {55 / 255.0f, 126 / 255.0f, 184 / 255.0f}, {77 / 255.0f, 175 / 255.0f, 74 / 255.0f},
    {152 / 255.0f, 78 / 255.0f, 163 / 255.0f}, {255 / 255.0f, 127 / 255.0f, 0 / 255.0f}, {255 / 255.0f, 255 / 255.0f, 51 / 255.0f},
    {166 / 255.0f, 86 / 255.0f, 40 / 255.0f}, {247 / 255.0f, 129 / 255.0f, 191 / 255.0f}};

__device__ __forceinline__ float3 color_bilinear(const FrameImages& F, int layer, float u, float v) {   // RGB8 LINEAR
  const Axis X = axis_linear(u, F.cw), Y = axis_linear(v, F.ch);
  const uchar4* __restrict__ b = F.color + (size_t)layer * F.cw * F.ch;
  const uchar4 t00 = b[(size_t)Y.i0 * F.cw + X.i0], t10 = b[(size_t)Y.i0 * F.cw + X.i1];
  const uchar4 t01 = b[(size_t)Y.i1 * F.cw + X.i0], t11 = b[(size_t)Y.i1 * F.cw + X.i1];
  float3 o;
  o.x = lerpf(lerpf(t00.x / 255.0f, t10.x / 255.0f, X.a), lerpf(t01.x / 255.0f, t11.x / 255.0f, X.a), Y.a);
  o.y = lerpf(lerpf(t00.y / 255.0f, t10.y / 255.0f, X.a), lerpf(t01.y / 255.0f, t11.y / 255.0f, X.a), Y.a);
  o.z = lerpf(lerpf(t00.z / 255.0f, t10.z / 255.0f, X.a), lerpf(t01.z / 255.0f, t11.z / 255.0f, X.a), Y.a);
  return o;
}

// blendColors(), tsdf_raymarch.fs:295-330
__device__ float4 blend_colors(const StreamTable& T, const FrameImages& F, float limit, float3 sp) {
  float3 tc = make_float3(0, 0, 0), tc2 = make_float3(0, 0, 0);
  float tw = 0.0f, tw2 = 0.0f;
  for (int i = 0; i < T.n; ++i) {
    const StreamLut& L = T.s[i];
    const float3 pc = tex3d_rgba_xyz(L.inv, L.inv_res, sp.x, sp.y, sp.z);
    const float2 pcol = tex3d_rg(L.uv, L.uv_res, pc.x, pc.y, pc.z);
    const float3 col = color_bilinear(F, i, pcol.x, pcol.y);
    const Dqs q = dqs_fetch(F, i, pc.x, pc.y);
    const float dist = fabsf(dqs_depth(q) - pc.z);
    float quality = 0.0f;
    if (dist < limit) quality = dqs_quality(q);
    const float de = dist + 0.01f;
    tc.x = tc.x + col.x * quality / de; tc.y = tc.y + col.y * quality / de; tc.z = tc.z + col.z * quality / de;
    tw += quality / de;
    tc2.x = tc2.x + col.x / dist; tc2.y = tc2.y + col.y / dist; tc2.z = tc2.z + col.z / dist;
    tw2 += 1.0f / dist;
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
// shade(), shading.glsl:32-69
__device__ float3 shade(const ViewParams& P, float3 vp, float3 vn, float3 diffuse) {
  if (P.shade_mode == 0) return diffuse;
  if (P.shade_mode == 1) {
    const float3 LP = make_float3(1.5f, 1.0f, 1.0f), LD = make_float3(1.0f, 0.9f, 0.7f);
    const float3 LA = make_float3(LD.x * 0.2f, LD.y * 0.2f, LD.z * 0.2f);
    float diff = 0.0f, spec = 0.0f;
    const float3 toLight = normalize3(make_float3(LP.x - vp.x, LP.y - vp.y, LP.z - vp.z));
    const float la = vn.x * toLight.x + vn.y * toLight.y + vn.z * toLight.z;
    if (!(la <= 0.0f)) {
      diff = fmaxf(la, 0.0f);
      const float3 tv = normalize3(make_float3(-vp.x, -vp.y, -vp.z));
      const float3 hv = normalize3(make_float3(toLight.x + tv.x, toLight.y + tv.y, toLight.z + tv.z));
      spec = powf(hv.x * vn.x + hv.y * vn.y + hv.z * vn.z, 20.0f);
      const float a = (1.0f - la) * (1.0f - la);
      spec *= 1.0f - a * a * a;
    }
    return make_float3(LA.x * 0.5f + LD.x * 0.5f * diff + 1.0f * 0.5f * spec,
                       LA.y * 0.5f + LD.y * 0.5f * diff + 1.0f * 0.5f * spec,
                       LA.z * 0.5f + LD.z * 0.5f * diff + 1.0f * 0.5f * spec);
  }
  if (P.shade_mode == 2) {
    const float4 r = mat_mul(P.glnormal_inv, vn.x, vn.y, vn.z, 0.0f);
    return make_float3(r.x, r.y, r.z);
  }
  return make_float3(1.0f, 1.0f, 1.0f);
}

// Which slab owns a sample (multi-GPU): the voxel plane floor(pos.z * rz), clamped.
__device__ __forceinline__ bool sample_owned(const Volume& V, float pz) {
  const float f = fminf(fmaxf(floorf(pz * (float)V.res[2]), 0.0f), (float)(V.res[2] - 1));
  const int tz = ((int)f) >> 3;
  return tz >= V.own_tz0 && tz < V.own_tz1;
}

__global__ __launch_bounds__(256) void k_raymarch(ViewParams P, StreamTable T, FrameImages F, Volume V, RayTarget R, int partial) {
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int px = blockIdx.x * 16 + (wv & 1) * 8 + (ln & 7);
  const int py = blockIdx.y * 16 + (wv >> 1) * 8 + (ln >> 3);
  if (px >= P.w || py >= P.h) return;
  const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
  const float limit = V.limit, sd = limit * 0.5f;                       // sampleDistance, :34
  const float3 dn = normalize3(pixel_dir_vol(P, fx, fy));
  const float3 step = make_float3(dn.x * sd, dn.y * sd, dn.z * sd);     // :64
  uint32_t max_n = 0;
  float3 pos = make_float3(0, 0, 0);
  bool covered = true;
  if (P.skip) {                                                         // getStartPos(), :384-393
    const uint4 dmb = ((const uint4*)R.peels)[(size_t)py * P.w + px];
    const float dm_r = __uint_as_float(dmb.x), dm_g = -__uint_as_float(dmb.y), dm_b = __uint_as_float(dmb.z);   // g = min(-z) = -max z
    float r = dm_r;
    r = (r >= dm_b) ? 0.0f : r;
    const float3 pf = screen_to_vol(P, fx, fy, r);
    float3 pb = screen_to_vol(P, fx, fy, -dm_g);
    if (r >= 1.0f) pb = pf;
    pos = pf;
    const float3 dd = make_float3(pf.x - pb.x, pf.y - pb.y, pf.z - pb.z);
    max_n = (uint32_t)ceilf(sqrtf(dd.x * dd.x + dd.y * dd.y + dd.z * dd.z) / sd);   // :73
  } else {                                                              // intersectBox(), :363-374
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
  float prev = -limit;                                                  // :89
  bool prev_valid = true;
  float3 pos_prev = pos;
  uint32_t n = 0;
  bool hit = false;
  while (n < max_n) {                                                   // :92-110
    n += 1;
    if (!partial || sample_owned(V, pos.z)) {
      const float density = tex3d_tsdf(V, pos.x, pos.y, pos.z);
      if (density > 0.0f) {
        if (!prev_valid) prev = tex3d_tsdf(V, pos_prev.x, pos_prev.y, pos_prev.z);   // sample n-1 lies in the halo
        const float k = prev / (density - prev);
        pos = make_float3((pos.x - step.x) - step.x * k, (pos.y - step.y) - step.y * k, (pos.z - step.z) - step.z * k);
        hit = true;
        break;
      }
      prev = density;
      prev_valid = true;
    } else {
      prev_valid = false;
    }
    pos_prev = pos;
    pos = make_float3(pos.x + step.x, pos.y + step.y, pos.z + step.z);
  }
  const size_t oi = (size_t)py * R.stride + px;
  float4 out = make_float4(R.clear[0], R.clear[1], R.clear[2], R.clear[3]);
  float out_d = 1.0f;
  if (hit) {                                                            // submitFragment(), :116-134
    const float gx = tex3d_tsdf(V, pos.x + sd, pos.y, pos.z) - tex3d_tsdf(V, pos.x - sd, pos.y, pos.z);
    const float gy = tex3d_tsdf(V, pos.x, pos.y + sd, pos.z) - tex3d_tsdf(V, pos.x, pos.y - sd, pos.z);
    const float gz = tex3d_tsdf(V, pos.x, pos.y, pos.z + sd) - tex3d_tsdf(V, pos.x, pos.y, pos.z - sd);
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
    if (fd < 1.0f) { out = col; out_d = fd; }                           // GL_LESS against the cleared 1.0
  }
  R.color[oi] = out;
  R.depth[oi] = out_d;
  const float ns = (float)n * 0.0027f;                                  // writeNumSamples(), :395-398
  R.nsamples[(size_t)py * P.w + px] = covered ? ((partial && !hit) ? -ns : ns) : 0.0f;
}
void launch_raymarch(hipStream_t st, const ViewParams& P, const StreamTable& T, const FrameImages& F, const Volume& V, const RayTarget& R, int partial) {
  dim3 grid((P.w + 15) / 16, (P.h + 15) / 16);
  hipLaunchKernelGGL(k_raymarch, grid, dim3(256), 0, st, P, T, F, V, R, partial);
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
