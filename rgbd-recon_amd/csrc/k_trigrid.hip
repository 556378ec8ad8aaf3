// Triangle-grid back-end (gfx950), SURVEY.md section 8 f4: kinect::ReconTrigrid::draw(), framework/reconstruction/recon_trigrid.cpp:85-148
// with glsl/trigrid_accum.{vs,gs,fs} and trigrid_normalize.fs.  Two triangles per depth-pixel cell and sensor:
//   stage 0  z-buffer of all surviving fragments           -> atomicMin on the window-z bit pattern (z in [0,1])
//   stage 1  quality-weighted shaded colour of every fragment within epsilon of the front surface, ONE/ONE blending
//                                                           -> four fp32 atomicAdd per fragment
//   stage 2  colour / weight where weight > 0, depth from stage 0
// One thread per (cell, sensor) sets up both triangles (the triangles are a pixel or two across: a scan of the bounding box
// is the whole rasteriser).  Coverage, interpolation and every test are the oracle's expressions in the oracle's order; the
// only difference is the ORDER of the additive blend (GL blends in draw order, atomics in arrival order), which moves the
// last bits of the colour, never the depth or the coverage.  Literal quirks kept: the vertex buffer's swapped loop bounds
// (cells x < H, y < W, recon_trigrid.cpp:53-54) and trigrid_accum.fs:69's extra half pixel.
#include "shading_dev.hpp"

namespace rr {

struct TriVert { float3 pos_cs, pos_es; float tcx, tcy, depth, quality, xw, yw, zw, iw; bool front; };
struct TriSetup { TriVert v[3]; float3 normal; float area; bool ok; };
struct TriFragment { float z, tcx, tcy, quality; float3 pos_es, pos_cs; };

__device__ __forceinline__ float len3(float3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ float3 sub3(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }

__device__ __forceinline__ TriVert tri_vertex(const ViewParams& P, const PointParams& Q, const StreamTable& T, const FrameImages& F, int l, int gx, int gy) {
  const float stepX = 1.0f / (float)F.w, stepY = 1.0f / (float)F.h;                    // recon_trigrid.cpp:51-52
  const float u = (float)(((double)gx + 0.5) * (double)stepX), v = (float)(((double)gy + 0.5) * (double)stepY);
  const int nx = axis_nearest(u, F.w), ny = axis_nearest(v, F.h);
  const float4 dq = F.dqs[((size_t)l * F.h + ny) * F.w + nx];
  TriVert t;
  t.depth = dq.x; t.quality = dq.y;
  const StreamLut& L = T.s[l];
  t.pos_cs = tex3d_rgba_xyz(L.xyz, L.xyz_res, u, v, t.depth);
  const float2 tc = tex3d_rg(L.uv, L.uv_res, u, v, t.depth);
  t.tcx = tc.x; t.tcy = tc.y;
  const float4 pe = mat_mul(P.mv, t.pos_cs.x, t.pos_cs.y, t.pos_cs.z, 1.0f);
  t.pos_es = make_float3(pe.x, pe.y, pe.z);
  const float4 clip = mat_mul(Q.pmv, t.pos_cs.x, t.pos_cs.y, t.pos_cs.z, 1.0f);
  t.front = clip.w > 0.0f;
  t.iw = 1.0f / clip.w;
  t.xw = (clip.x / clip.w * 0.5f + 0.5f) * (float)P.w;
  t.yw = (clip.y / clip.w * 0.5f + 0.5f) * (float)P.h;
  t.zw = clip.z / clip.w * 0.5f + 0.5f;
  return t;
}

__device__ __forceinline__ TriSetup tri_setup(float min_length, const TriVert& a, const TriVert& b, const TriVert& d) {   // trigrid_accum.gs
  TriSetup S; S.v[0] = a; S.v[1] = b; S.v[2] = d; S.ok = false;
  if (a.depth < 0.0f || b.depth < 0.0f || d.depth < 0.0f) return S;                    // validSurface, :31-42
  const float avg = (a.depth + b.depth + d.depth) / 3.0f;
  const float l = min_length * avg * 4.0f;
  if (!(len3(sub3(b.pos_cs, a.pos_cs)) < l && len3(sub3(d.pos_cs, a.pos_cs)) < l && len3(sub3(d.pos_cs, b.pos_cs)) < l)) return S;
  if (!(a.front && b.front && d.front)) return S;
  const float3 ea = sub3(b.pos_es, a.pos_es), eb = sub3(d.pos_es, a.pos_es);
  S.normal = normalize3(make_float3(ea.y * eb.z - eb.y * ea.z, ea.z * eb.x - eb.z * ea.x, ea.x * eb.y - eb.x * ea.y));   // :59
  S.area = (b.xw - a.xw) * (d.yw - a.yw) - (d.xw - a.xw) * (b.yw - a.yw);
  if (!(S.area != 0.0f)) return S;
  S.ok = true;
  return S;
}

__device__ __forceinline__ bool tri_fragment(const TriSetup& S, int px, int py, TriFragment& f) {
  const float x = (float)px + 0.5f, y = (float)py + 0.5f;
  const TriVert &a = S.v[0], &b = S.v[1], &d = S.v[2];
  const float e0 = (d.xw - b.xw) * (y - b.yw) - (d.yw - b.yw) * (x - b.xw);
  const float e1 = (a.xw - d.xw) * (y - d.yw) - (a.yw - d.yw) * (x - d.xw);
  const float e2 = (b.xw - a.xw) * (y - a.yw) - (b.yw - a.yw) * (x - a.xw);
  const float sgn = S.area > 0.0f ? 1.0f : -1.0f;
  const float ex[3] = {(d.xw - b.xw) * sgn, (a.xw - d.xw) * sgn, (b.xw - a.xw) * sgn}, ey[3] = {(d.yw - b.yw) * sgn, (a.yw - d.yw) * sgn, (b.yw - a.yw) * sgn};
  const float ee[3] = {e0 * sgn, e1 * sgn, e2 * sgn};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (ee[i] < 0.0f) return false;
    if (ee[i] == 0.0f && !(ey[i] > 0.0f || (ey[i] == 0.0f && ex[i] < 0.0f))) return false;
    if (!(ee[i] >= 0.0f)) return false;
  }
  const float l0 = e0 / S.area, l1 = e1 / S.area, l2 = e2 / S.area;
  f.z = l0 * a.zw + l1 * b.zw + l2 * d.zw;
  if (!(f.z >= 0.0f && f.z <= 1.0f)) return false;
  const float w0 = l0 * a.iw, w1 = l1 * b.iw, w2 = l2 * d.iw, iw = w0 + w1 + w2;
#define RR_IP(p, q, r) ((w0 * (p) + w1 * (q) + w2 * (r)) / iw)
  f.tcx = RR_IP(a.tcx, b.tcx, d.tcx); f.tcy = RR_IP(a.tcy, b.tcy, d.tcy); f.quality = RR_IP(a.quality, b.quality, d.quality);
  f.pos_es = make_float3(RR_IP(a.pos_es.x, b.pos_es.x, d.pos_es.x), RR_IP(a.pos_es.y, b.pos_es.y, d.pos_es.y), RR_IP(a.pos_es.z, b.pos_es.z, d.pos_es.z));
  f.pos_cs = make_float3(RR_IP(a.pos_cs.x, b.pos_cs.x, d.pos_cs.x), RR_IP(a.pos_cs.y, b.pos_cs.y, d.pos_cs.y), RR_IP(a.pos_cs.z, b.pos_cs.z, d.pos_cs.z));
#undef RR_IP
  return true;
}

__device__ __forceinline__ bool tri_fragment_kept(const PointParams& Q, const TriSetup& S, const TriFragment& f, float3& n) {   // trigrid_accum.fs:44-62
  const bool in_box = f.pos_cs.x >= Q.bbox_min[0] && f.pos_cs.y >= Q.bbox_min[1] && f.pos_cs.z >= Q.bbox_min[2] &&
                      f.pos_cs.x <= Q.bbox_max[0] && f.pos_cs.y <= Q.bbox_max[1] && f.pos_cs.z <= Q.bbox_max[2];
  if (!in_box) return false;
  if (f.tcx > 0.99f || f.tcx < 0.01f || f.tcy > 0.99f || f.tcy < 0.01f) return false;
  const float3 nn = normalize3(S.normal);
  n = make_float3(-nn.x, -nn.y, -nn.z);
  const float3 pe = normalize3(f.pos_es);
  if (n.x * pe.x + n.y * pe.y + n.z * pe.z > 0.0f) return false;
  return true;
}

template <int kStage>
__global__ __launch_bounds__(256) void k_trigrid(ViewParams P, PointParams Q, StreamTable T, FrameImages F, float min_length, uint32_t* __restrict__ zbuf,
                                                 float* __restrict__ acc) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), l = blockIdx.z;
  if (x >= F.h || y >= F.w) return;                                      // sic: cells x < height, y < width (recon_trigrid.cpp:53-54)
  const TriVert v00 = tri_vertex(P, Q, T, F, l, x, y), v10 = tri_vertex(P, Q, T, F, l, x + 1, y), v01 = tri_vertex(P, Q, T, F, l, x, y + 1),
                v11 = tri_vertex(P, Q, T, F, l, x + 1, y + 1);
#pragma unroll 1
  for (int t = 0; t < 2; ++t) {
    const TriSetup S = t == 0 ? tri_setup(min_length, v00, v10, v01) : tri_setup(min_length, v10, v11, v01);    // :55-61
    if (!S.ok) continue;
    const float minx = fminf(fminf(S.v[0].xw, S.v[1].xw), S.v[2].xw), maxx = fmaxf(fmaxf(S.v[0].xw, S.v[1].xw), S.v[2].xw);
    const float miny = fminf(fminf(S.v[0].yw, S.v[1].yw), S.v[2].yw), maxy = fmaxf(fmaxf(S.v[0].yw, S.v[1].yw), S.v[2].yw);
    if (!(maxx >= 0.0f && maxy >= 0.0f && minx <= (float)P.w && miny <= (float)P.h)) continue;
    const int x0 = (int)fmaxf(floorf(minx - 0.5f), 0.0f), x1 = (int)fminf(ceilf(maxx - 0.5f), (float)(P.w - 1));
    const int y0 = (int)fmaxf(floorf(miny - 0.5f), 0.0f), y1 = (int)fminf(ceilf(maxy - 0.5f), (float)(P.h - 1));
    for (int py = y0; py <= y1; ++py)
      for (int px = x0; px <= x1; ++px) {
        TriFragment f;
        if (!tri_fragment(S, px, py, f)) continue;
        float3 n;
        if (!tri_fragment_kept(Q, S, f, n)) continue;
        const size_t o = (size_t)py * P.w + px;
        if (kStage == 0) {
          atomicMin(&zbuf[o], __float_as_uint(f.z));                     // GL_LESS
        } else {
          const float depth_curr = __uint_as_float(zbuf[o]);
          const float4 pc = mat_mul(P.img_to_eye, ((float)px + 0.5f) + 0.5f, ((float)py + 0.5f) + 0.5f, depth_curr, 1.0f);   // sic, trigrid_accum.fs:69
          const float3 es = make_float3(pc.x / pc.w, pc.y / pc.w, pc.z / pc.w);
          if (0.075f < len3(sub3(es, f.pos_es))) continue;               // epsilon, recon_trigrid.cpp:35
          float3 col;
          if (P.shade_mode == 3) col = make_float3(c_camera_colors[l & 7][0], c_camera_colors[l & 7][1], c_camera_colors[l & 7][2]);
          else col = shade(P, f.pos_es, n, color_bilinear(F, l, f.tcx, f.tcy));
          atomicAdd(&acc[4 * o], col.x * f.quality); atomicAdd(&acc[4 * o + 1], col.y * f.quality);
          atomicAdd(&acc[4 * o + 2], col.z * f.quality); atomicAdd(&acc[4 * o + 3], f.quality);
        }
      }
  }
}

__global__ __launch_bounds__(256) void k_trigrid_clear(uint32_t* __restrict__ zbuf, float4* __restrict__ acc, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { zbuf[i] = __float_as_uint(1.0f); acc[i] = make_float4(0, 0, 0, 0); }
}
__global__ __launch_bounds__(256) void k_trigrid_normalize(const uint32_t* __restrict__ zbuf, const float4* __restrict__ acc, int n, float4* __restrict__ fb_c,
                                                           float* __restrict__ fb_d) {                        // trigrid_normalize.fs
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 a = acc[i];
  if (a.w > 0.0f) { fb_c[i] = make_float4(a.x / a.w, a.y / a.w, a.z / a.w, a.w / a.w); fb_d[i] = __uint_as_float(zbuf[i]); }
  else { fb_c[i] = make_float4(0, 0, 0, 0); fb_d[i] = 1.0f; }
}

void launch_draw_trigrid(hipStream_t st, const ViewParams& P, const PointParams& Q, const StreamTable& T, const FrameImages& F, float min_length, uint32_t* zbuf,
                         float4* acc, float4* fb_c, float* fb_d) {
  const int n = P.w * P.h;
  const dim3 cells((F.h + 63) / 64, (F.w + 3) / 4, T.n);
  hipLaunchKernelGGL(k_trigrid_clear, dim3((n + 255) / 256), dim3(256), 0, st, zbuf, acc, n);
  hipLaunchKernelGGL(k_trigrid<0>, cells, dim3(256), 0, st, P, Q, T, F, min_length, zbuf, (float*)acc);
  hipLaunchKernelGGL(k_trigrid<1>, cells, dim3(256), 0, st, P, Q, T, F, min_length, zbuf, (float*)acc);
  hipLaunchKernelGGL(k_trigrid_normalize, dim3((n + 255) / 256), dim3(256), 0, st, zbuf, acc, n, fb_c, fb_d);
}

}  // namespace rr
