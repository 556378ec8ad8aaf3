// Shading helpers shared by the raymarch (k_raymarch.hip) and the point back-end (k_points.hip): glsl/shading.glsl.
#ifndef RR_SHADING_DEV_HPP
#define RR_SHADING_DEV_HPP
#include "sampling.hpp"

namespace rr {

static __constant__ float c_camera_colors[8][3] = {   // shading.glsl:24-30, extended past 5 streams
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

// shade(), shading.glsl:32-69
__device__ __forceinline__ float3 shade(const ViewParams& P, float3 vp, float3 vn, float3 diffuse) {
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

}  // namespace rr
#endif
