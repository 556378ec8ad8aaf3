// Point back-end (gfx950), SURVEY.md section 8 f4: kinect::ReconPoints::draw(), framework/reconstruction/recon_points.cpp:71-111
// with glsl/points.vs / points.gs / points.fs -- one GL point sprite per depth pixel and layer, depth-tested into the
// framebuffer.  The rasteriser's z-buffer becomes a 64-bit atomicMin per covered pixel on (window z bits, point id): z is
// in [0, 1) where uint order == float order, and the id (layer, then row-major pixel) is the draw order, so equal depths
// resolve exactly as GL_LESS does for primitives drawn in order -- deterministic whatever the scheduling.  A second kernel
// shades the winner of every pixel once (all varyings are `flat`).  The definitions GL leaves open are listed in the oracle.
#include "shading_dev.hpp"

namespace rr {

struct PointVertex { float3 pos_cs, pos_es; float2 tc; float xw, yw, zw, size; bool ok; };

__device__ __forceinline__ PointVertex point_vertex(const ViewParams& P, const PointParams& Q, const StreamTable& T, const FrameImages& F, int l, int x, int y) {
  PointVertex o;
  o.ok = false;
  const float stepX = 1.0f / (float)F.w, stepY = 1.0f / (float)F.h;                    // recon_points.cpp:44-45
  const float u = (float)(((double)x + 0.5) * (double)stepX), v = (float)(((double)y + 0.5) * (double)stepY);   // :48
  const float depth = F.depth[((size_t)l * F.h + y) * F.w + x];
  const StreamLut& L = T.s[l];
  o.pos_cs = tex3d_rgba_xyz(L.xyz, L.xyz_res, u, v, depth);                             // points.vs:27
  o.tc = tex3d_rg(L.uv, L.uv_res, u, v, depth);                                         // :29
  const bool in_box = o.pos_cs.x >= Q.bbox_min[0] && o.pos_cs.y >= Q.bbox_min[1] && o.pos_cs.z >= Q.bbox_min[2] &&
                      o.pos_cs.x <= Q.bbox_max[0] && o.pos_cs.y <= Q.bbox_max[1] && o.pos_cs.z <= Q.bbox_max[2];
  if (!in_box || depth <= 0.0f) return o;                                               // points.gs:36-38
  if (o.tc.x > 0.99f || o.tc.x < 0.01f || o.tc.y > 0.99f || o.tc.y < 0.01f) return o;   // points.fs:38-41
  const float4 pe = mat_mul(P.mv, o.pos_cs.x, o.pos_cs.y, o.pos_cs.z, 1.0f);
  o.pos_es = make_float3(pe.x, pe.y, pe.z);
  const float4 clip = mat_mul(Q.pmv, o.pos_cs.x, o.pos_cs.y, o.pos_cs.z, 1.0f);
  if (!(clip.w > 0.0f) || fabsf(clip.x) > clip.w || fabsf(clip.y) > clip.w || fabsf(clip.z) > clip.w) return o;
  o.xw = (clip.x / clip.w * 0.5f + 0.5f) * (float)P.w;
  o.yw = (clip.y / clip.w * 0.5f + 0.5f) * (float)P.h;
  o.zw = clip.z / clip.w * 0.5f + 0.5f;
  if (!(o.zw < 1.0f)) return o;
  const float max_size = P.shade_mode == 3 ? 4.0f : 10.0f;                              // points.gs:49-53
  o.size = fminf(fmaxf(max_size / sqrtf(o.pos_es.x * o.pos_es.x + o.pos_es.y * o.pos_es.y + o.pos_es.z * o.pos_es.z), 1.0f), 2047.0f);
  o.ok = true;
  return o;
}

__global__ __launch_bounds__(256) void k_points_clear(unsigned long long* __restrict__ key, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) key[i] = ((unsigned long long)__float_as_uint(1.0f) << 32) | 0xffffffffull;
}

__global__ __launch_bounds__(256) void k_points_scatter(ViewParams P, PointParams Q, StreamTable T, FrameImages F, unsigned long long* __restrict__ key) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), l = blockIdx.z;
  if (x >= F.w || y >= F.h) return;
  const PointVertex pv = point_vertex(P, Q, T, F, l, x, y);
  if (!pv.ok) return;
  const float half = pv.size * 0.5f;
  int x0 = (int)ceilf((pv.xw - half) - 0.5f), x1 = (int)ceilf((pv.xw + half) - 0.5f) - 1;
  int y0 = (int)ceilf((pv.yw - half) - 0.5f), y1 = (int)ceilf((pv.yw + half) - 0.5f) - 1;
  x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, P.w - 1); y1 = min(y1, P.h - 1);
  const unsigned long long k = ((unsigned long long)__float_as_uint(pv.zw) << 32) | (unsigned long long)(uint32_t)((l * F.h + y) * F.w + x);
  for (int py = y0; py <= y1; ++py)
    for (int px = x0; px <= x1; ++px) atomicMin(&key[(size_t)py * P.w + px], k);
}

__global__ __launch_bounds__(256) void k_points_resolve(ViewParams P, PointParams Q, StreamTable T, FrameImages F, const unsigned long long* __restrict__ key,
                                                        float4* __restrict__ fb_c, float* __restrict__ fb_d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.w * P.h) return;
  const unsigned long long k = key[i];
  const uint32_t id = (uint32_t)k;
  if (id == 0xffffffffu) { fb_c[i] = make_float4(0, 0, 0, 0); fb_d[i] = 1.0f; return; }
  const int x = (int)(id % (uint32_t)F.w), y = (int)((id / (uint32_t)F.w) % (uint32_t)F.h), l = (int)(id / (uint32_t)(F.w * F.h));
  const PointVertex pv = point_vertex(P, Q, T, F, l, x, y);
  float3 out;
  if (P.shade_mode == 3) out = make_float3(c_camera_colors[l & 7][0], c_camera_colors[l & 7][1], c_camera_colors[l & 7][2]);   // points.fs:69-71
  else {
    const float3 col = color_bilinear(F, l, pv.tc.x, pv.tc.y);
    float3 n = make_float3(0, 0, 0);
    if (Q.normals) { const float4 t = Q.normals[((size_t)l * F.h + y) * F.w + x]; n = make_float3(t.x, t.y, t.z); }
    const float* mi = P.mv_inv.m;                                                       // inverseTranspose(MV) * (n, 0)
    const float3 vn = make_float3(mi[0] * n.x + mi[1] * n.y + mi[2] * n.z, mi[4] * n.x + mi[5] * n.y + mi[6] * n.z, mi[8] * n.x + mi[9] * n.y + mi[10] * n.z);
    out = shade(P, pv.pos_es, vn, col);
  }
  fb_c[i] = make_float4(out.x, out.y, out.z, 1.0f);
  fb_d[i] = __uint_as_float((uint32_t)(k >> 32));
}

void launch_draw_points(hipStream_t st, const ViewParams& P, const PointParams& Q, const StreamTable& T, const FrameImages& F, unsigned long long* key,
                        float4* fb_c, float* fb_d) {
  const int n = P.w * P.h;
  hipLaunchKernelGGL(k_points_clear, dim3((n + 255) / 256), dim3(256), 0, st, key, n);
  hipLaunchKernelGGL(k_points_scatter, dim3((F.w + 63) / 64, (F.h + 3) / 4, T.n), dim3(256), 0, st, P, Q, T, F, key);
  hipLaunchKernelGGL(k_points_resolve, dim3((n + 255) / 256), dim3(256), 0, st, P, Q, T, F, key, fb_c, fb_d);
}

}  // namespace rr
