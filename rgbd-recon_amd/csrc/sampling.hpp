// Device-side GL sampling semantics (OpenGL 4.4 §8.14 with the reference's sampler state,
// SURVEY.md Appendix A): manual fp32 filtering from plain HBM buffers, never hipTextureObject
// filtering (fixed-point weights would break parity).  lerp(a,b,t) = a + (b-a)*t, x then y then z.
// The translation unit is built with -ffp-contract=off: every rounding stays where it is written.
#pragma once
#include "tsdf_common.hpp"

namespace rr {

__device__ __forceinline__ float lerpf(float a, float b, float t) { return a + (b - a) * t; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

struct Axis { int i0, i1; float a; };
// LINEAR + CLAMP_TO_EDGE along one axis: f = u*n - 0.5, i0 = floor(f), weight = fract
__device__ __forceinline__ Axis axis_linear(float u, int n) {
  float f = u * (float)n - 0.5f;
  float fl = floorf(f);
  Axis r;
  r.a = f - fl;
  // clamp(floor, 0, n - 1) and clamp(floor + 1, 0, n - 1) as one v_med3_f32 each, on the integer-valued floats (exact below 2^24;
  // a NaN coordinate gives texel 0 for both, as the integer clamps did): 5 instead of 8 instructions per axis
  const float hi = (float)(n - 1);
  r.i0 = (int)__builtin_amdgcn_fmed3f(fl, 0.0f, hi);
  r.i1 = (int)__builtin_amdgcn_fmed3f(fl + 1.0f, 0.0f, hi);
  return r;
}
__device__ __forceinline__ int axis_nearest(float u, int n) {
  float f = floorf(u * (float)n);
  return (int)__builtin_amdgcn_fmed3f(f, 0.0f, (float)(n - 1));
}

__device__ __forceinline__ float3 lerp3(float4 a, float4 b, float t) {
  return make_float3(lerpf(a.x, b.x, t), lerpf(a.y, b.y, t), lerpf(a.z, b.z, t));
}
__device__ __forceinline__ float3 lerp3(float3 a, float3 b, float t) {
  return make_float3(lerpf(a.x, b.x, t), lerpf(a.y, b.y, t), lerpf(a.z, b.z, t));
}

// texture(sampler3D RGBA32F, p).xyz
__device__ __forceinline__ float3 tex3d_rgba_xyz(const float4* __restrict__ t, const int* res, float u, float v, float w) {
  const Axis X = axis_linear(u, res[0]), Y = axis_linear(v, res[1]), Z = axis_linear(w, res[2]);
  // 32-bit texel indices (a LUT has < 2^31 texels, checked at upload)
  const uint32_t zr0 = (uint32_t)Z.i0 * (uint32_t)res[1], zr1 = (uint32_t)Z.i1 * (uint32_t)res[1];
  const uint32_t r00 = (zr0 + Y.i0) * (uint32_t)res[0], r10 = (zr0 + Y.i1) * (uint32_t)res[0];
  const uint32_t r01 = (zr1 + Y.i0) * (uint32_t)res[0], r11 = (zr1 + Y.i1) * (uint32_t)res[0];
  const float3 c00 = lerp3(t[r00 + X.i0], t[r00 + X.i1], X.a);
  const float3 c10 = lerp3(t[r10 + X.i0], t[r10 + X.i1], X.a);
  const float3 c01 = lerp3(t[r01 + X.i0], t[r01 + X.i1], X.a);
  const float3 c11 = lerp3(t[r11 + X.i0], t[r11 + X.i1], X.a);
  return lerp3(lerp3(c00, c10, Y.a), lerp3(c01, c11, Y.a), Z.a);
}
// texture(sampler3D RG32F, p).xy
__device__ __forceinline__ float2 tex3d_rg(const float2* __restrict__ t, const int* res, float u, float v, float w) {
  const Axis X = axis_linear(u, res[0]), Y = axis_linear(v, res[1]), Z = axis_linear(w, res[2]);
  const uint32_t zr0 = (uint32_t)Z.i0 * (uint32_t)res[1], zr1 = (uint32_t)Z.i1 * (uint32_t)res[1];
  const uint32_t r00 = (zr0 + Y.i0) * (uint32_t)res[0], r10 = (zr0 + Y.i1) * (uint32_t)res[0];
  const uint32_t r01 = (zr1 + Y.i0) * (uint32_t)res[0], r11 = (zr1 + Y.i1) * (uint32_t)res[0];
  float2 a = t[r00 + X.i0], b = t[r00 + X.i1];
  const float2 c00 = make_float2(lerpf(a.x, b.x, X.a), lerpf(a.y, b.y, X.a));
  a = t[r10 + X.i0]; b = t[r10 + X.i1];
  const float2 c10 = make_float2(lerpf(a.x, b.x, X.a), lerpf(a.y, b.y, X.a));
  a = t[r01 + X.i0]; b = t[r01 + X.i1];
  const float2 c01 = make_float2(lerpf(a.x, b.x, X.a), lerpf(a.y, b.y, X.a));
  a = t[r11 + X.i0]; b = t[r11 + X.i1];
  const float2 c11 = make_float2(lerpf(a.x, b.x, X.a), lerpf(a.y, b.y, X.a));
  return make_float2(lerpf(lerpf(c00.x, c10.x, Y.a), lerpf(c01.x, c11.x, Y.a), Z.a),
                     lerpf(lerpf(c00.y, c10.y, Y.a), lerpf(c01.y, c11.y, Y.a), Z.a));
}

// The 2x2 footprint of the packed {depth, quality, silhouette} image at (u, v) of layer i.
struct Dqs { float4 t00, t10, t01, t11; float ax, ay; };
__device__ __forceinline__ Dqs dqs_fetch(const FrameImages& F, int layer, float u, float v) {
  const Axis X = axis_linear(u, F.w), Y = axis_linear(v, F.h);
  // uniform base + 32-bit BYTE offsets (the packed images of a context are far below 4 GiB: tsdf_create checks it): the loads become
  // `global_load_dwordx3 v, voffset, s[base]`, no 64-bit vector add per tap
  const char* __restrict__ b = (const char*)F.dqs;
  // 24-bit multiplies: full rate, and every operand (layer < 16, image rows/widths, w*h of a depth image) is far below 2^24
  const uint32_t base = (uint32_t)__mul24(layer, F.w * F.h), r0 = base + (uint32_t)__mul24(Y.i0, F.w), r1 = base + (uint32_t)__mul24(Y.i1, F.w);
  Dqs r;
  r.t00 = *(const float4*)(b + ((r0 + (uint32_t)X.i0) << 4)); r.t10 = *(const float4*)(b + ((r0 + (uint32_t)X.i1) << 4));
  r.t01 = *(const float4*)(b + ((r1 + (uint32_t)X.i0) << 4)); r.t11 = *(const float4*)(b + ((r1 + (uint32_t)X.i1) << 4));
  r.ax = X.a; r.ay = Y.a;
  return r;
}
__device__ __forceinline__ float dqs_silhouette(const Dqs& d) {   // LINEAR R32F
  return lerpf(lerpf(d.t00.z, d.t10.z, d.ax), lerpf(d.t01.z, d.t11.z, d.ax), d.ay);
}
__device__ __forceinline__ float dqs_quality(const Dqs& d) {      // LINEAR R32F
  return lerpf(lerpf(d.t00.y, d.t10.y, d.ax), lerpf(d.t01.y, d.t11.y, d.ax), d.ay);
}
// NEAREST RG32F .r out of the bilinear footprint.  With p = u*n the nearest texel is floor(p) and the bilinear
// pair is {floor(p - .5), +1}; f = p - .5 and its fraction a = f - floor(f) are exact in fp32 for p >= .25, so
// floor(p) = floor(f) + (a >= .5).  For p < .25 both pair members clamp to texel 0 and so does floor(p).
__device__ __forceinline__ float dqs_depth(const Dqs& d) {
  const bool xr = d.ax >= 0.5f, yr = d.ay >= 0.5f;
  const float lo = xr ? d.t10.x : d.t00.x, hi = xr ? d.t11.x : d.t01.x;
  return yr ? hi : lo;
}

// TSDF R32F, LINEAR + CLAMP_TO_EDGE, tile-major storage.  All offsets are 32-bit element indices (a context stores
// < 2^32 voxels); the address of tap (x,y,z) is the sum of three per-axis partial offsets, so the eight taps of a
// footprint cost six partials and eight adds instead of eight full index computations.
__device__ __forceinline__ uint32_t vol_off_x(int x) { return ((uint32_t)(x >> 3) << 9) + (uint32_t)(x & 7); }
// (24-bit multiplies are full rate, v_mul_lo_u32 quarter rate: tile coordinates and tiles-per-plane are far below 2^24 and the
// products -- tile indices of a context -- below 2^23)
__device__ __forceinline__ uint32_t vol_off_y(const Volume& V, int y) { return ((uint32_t)__mul24(y >> 3, V.ntx) << 9) + ((uint32_t)(y & 7) << 3); }
__device__ __forceinline__ uint32_t vol_off_z(const Volume& V, int z) { return ((uint32_t)__mul24((z >> 3) - V.tz0, V.nty * V.ntx) << 9) + ((uint32_t)(z & 7) << 6); }
__device__ __forceinline__ size_t vol_index(const Volume& V, int x, int y, int z) { return (size_t)(vol_off_x(x) + vol_off_y(V, y) + vol_off_z(V, z)); }

struct TsdfTaps { Axis X, Y, Z; };
// kWhole: the context is known to store every plane of the volume (the filter's own clamps already keep the taps inside).
template <bool kWhole = false>
__device__ __forceinline__ TsdfTaps tsdf_taps(const Volume& V, float u, float v, float w) {
  TsdfTaps t;
  t.X = axis_linear(u, V.res[0]); t.Y = axis_linear(v, V.res[1]); t.Z = axis_linear(w, V.res[2]);
  if (!kWhole) {
    // A slab context stores only planes [zlo, zhi].  Owned samples never leave them; a NaN position (NaN voxels exist,
    // tsdf_integration.vs:52) would clamp to plane 0, so keep the taps inside the allocation (the result is NaN anyway).
    t.Z.i0 = clampi(t.Z.i0, V.zlo, V.zhi);
    t.Z.i1 = clampi(t.Z.i1, V.zlo, V.zhi);
  }
  return t;
}
// sparse pool: the tap's tile goes through the slot table (one dependent load more per tap; unallocated tiles read -limit)
__device__ __forceinline__ float tsdf_tap_sparse(const Volume& V, int x, int y, int z) {
  const uint32_t tile = (uint32_t)__mul24((z >> 3) - V.tz0, V.nty * V.ntx) + (uint32_t)__mul24(y >> 3, V.ntx) + (uint32_t)(x >> 3);
  const uint32_t s = V.slot[tile];
  if (s == kNoSlot) return -V.limit;
  return V.data[((size_t)s << 9) + (uint32_t)(((z & 7) << 6) | ((y & 7) << 3) | (x & 7))];
}
// kSparse is a compile-time property of the kernel instantiation: the dense path must not carry the sparse one's code and
// registers (a run-time branch here cost the dense march 15 %)
template <bool kSparse>
__device__ __forceinline__ float tsdf_fetch(const Volume& V, const TsdfTaps& t) {
  if (kSparse) {
    const float c00 = lerpf(tsdf_tap_sparse(V, t.X.i0, t.Y.i0, t.Z.i0), tsdf_tap_sparse(V, t.X.i1, t.Y.i0, t.Z.i0), t.X.a);
    const float c10 = lerpf(tsdf_tap_sparse(V, t.X.i0, t.Y.i1, t.Z.i0), tsdf_tap_sparse(V, t.X.i1, t.Y.i1, t.Z.i0), t.X.a);
    const float c01 = lerpf(tsdf_tap_sparse(V, t.X.i0, t.Y.i0, t.Z.i1), tsdf_tap_sparse(V, t.X.i1, t.Y.i0, t.Z.i1), t.X.a);
    const float c11 = lerpf(tsdf_tap_sparse(V, t.X.i0, t.Y.i1, t.Z.i1), tsdf_tap_sparse(V, t.X.i1, t.Y.i1, t.Z.i1), t.X.a);
    return lerpf(lerpf(c00, c10, t.Y.a), lerpf(c01, c11, t.Y.a), t.Z.a);
  }
  const float* __restrict__ d = V.data;
  const uint32_t x0 = vol_off_x(t.X.i0), x1 = vol_off_x(t.X.i1);
  const uint32_t y0 = vol_off_y(V, t.Y.i0), y1 = vol_off_y(V, t.Y.i1);
  const uint32_t z0 = vol_off_z(V, t.Z.i0), z1 = vol_off_z(V, t.Z.i1);
  const uint32_t b00 = z0 + y0, b10 = z0 + y1, b01 = z1 + y0, b11 = z1 + y1;
  const float c00 = lerpf(d[b00 + x0], d[b00 + x1], t.X.a);
  const float c10 = lerpf(d[b10 + x0], d[b10 + x1], t.X.a);
  const float c01 = lerpf(d[b01 + x0], d[b01 + x1], t.X.a);
  const float c11 = lerpf(d[b11 + x0], d[b11 + x1], t.X.a);
  return lerpf(lerpf(c00, c10, t.Y.a), lerpf(c01, c11, t.Y.a), t.Z.a);
}
template <bool kSparse, bool kWhole = false>
__device__ __forceinline__ float tex3d_tsdf(const Volume& V, float u, float v, float w) { return tsdf_fetch<kSparse>(V, tsdf_taps<kWhole>(V, u, v, w)); }

// wave-wide min / max of a float: DPP row shifts + row broadcasts (8 VALU + 1 v_readlane), result wave-uniform.  fminf semantics: a NaN
// operand is ignored.
__device__ __forceinline__ float wave_min_f32(float v) {          // DPP row shifts + row broadcasts, result wave-uniform (see k_raymarch.hip)
#define RR_DPP_MIN(ctrl, rm, bm) v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rm, bm, false)))
  RR_DPP_MIN(0x111, 0xf, 0xf); RR_DPP_MIN(0x112, 0xf, 0xf); RR_DPP_MIN(0x114, 0xf, 0xe); RR_DPP_MIN(0x118, 0xf, 0xc);
  RR_DPP_MIN(0x142, 0xa, 0xf); RR_DPP_MIN(0x143, 0xc, 0xf);
#undef RR_DPP_MIN
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_f32(float v) { return -wave_min_f32(-v); }

__device__ __forceinline__ float4 mat_mul(const Mat4& a, float x, float y, float z, float w) {
  return make_float4(a.m[0] * x + a.m[4] * y + a.m[8] * z + a.m[12] * w,
                     a.m[1] * x + a.m[5] * y + a.m[9] * z + a.m[13] * w,
                     a.m[2] * x + a.m[6] * y + a.m[10] * z + a.m[14] * w,
                     a.m[3] * x + a.m[7] * y + a.m[11] * z + a.m[15] * w);
}
__device__ __forceinline__ float3 normalize3(float3 a) {
  const float s = 1.0f / sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
  return make_float3(a.x * s, a.y * s, a.z * s);
}

}  // namespace rr
