// Frame ingest decode kernels (gfx950), SURVEY.md section 8 f2.
// One wire message / one record per .stream file = per sensor [colour][depth] (NetKinectArray::readLoop,
// framework/NetKinectArray.cpp:513-523; readFromFiles :733-745).  The whole message is copied to HBM once; these kernels
// unpack it for all sensors in one launch each, into the layouts the pre-processing reads:
//   colour  RGB8 | DXT1 | DXT5 (NetKinectArray.cpp:118-130, :147-157)  -> uchar4 RGBA image
//   depth   float32 metres | 8-bit normalised (GL_LUMINANCE/GL_UNSIGNED_BYTE, :165-171) -> float raw depth
// The S3TC decode is the integer decode of the reference's vendored squish (external/squish colourblock.cpp:160-212,
// alpha.cpp:297-348), which the reference itself uses to read these blocks back (NetKinectArray.cpp:620).
// All of it is byte shuffling bound by HBM: <= 23 MB in, <= 10 MB out per frame.
#include "tsdf_common.hpp"

namespace rr {

// 4 RGB8 pixels (3 words) -> one 16-byte store
__global__ __launch_bounds__(256) void k_wire_rgb8(WireLayout L, uchar4* __restrict__ rgba) {
  const int l = blockIdx.y;
  const uint32_t quads = (uint32_t)(L.cw * L.ch) >> 2;
  const uint32_t* __restrict__ src = (const uint32_t*)(L.msg + (size_t)l * L.rec);
  uint4* __restrict__ dst = (uint4*)(rgba + (size_t)l * L.cw * L.ch);
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += gridDim.x * blockDim.x) {
    const uint32_t a = src[3 * q], b = src[3 * q + 1], c = src[3 * q + 2];      // R0G0B0R1 G1B1R2G2 B2R3G3B3
    uint4 o;
    o.x = (a & 0x00ffffffu) | 0xff000000u;
    o.y = (a >> 24) | ((b & 0xffffu) << 8) | 0xff000000u;
    o.z = (b >> 16) | ((c & 0xffu) << 16) | 0xff000000u;
    o.w = (c >> 8) | 0xff000000u;
    dst[q] = o;
  }
}

__device__ __forceinline__ void expand565(uint32_t v, int e[3]) {
  const int r = (v >> 11) & 31, g = (v >> 5) & 63, b = v & 31;
  e[0] = (r << 3) | (r >> 2); e[1] = (g << 2) | (g >> 4); e[2] = (b << 3) | (b >> 2);
}
__device__ __forceinline__ uint32_t pack_rgb(int r, int g, int b) { return (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16); }

// one thread per 4x4 block
template <bool kDxt5>
__global__ __launch_bounds__(256) void k_wire_dxt(WireLayout L, uchar4* __restrict__ rgba) {
  const int l = blockIdx.y;
  const int nbx = (L.cw + 3) >> 2, nby = (L.ch + 3) >> 2;
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= (uint32_t)(nbx * nby)) return;
  const uint32_t* __restrict__ blk = (const uint32_t*)(L.msg + (size_t)l * L.rec) + (size_t)k * (kDxt5 ? 4 : 2);
  uint32_t a_lo = 0, a_hi = 0, ends, idx;
  if (kDxt5) { const uint4 w = *(const uint4*)blk; a_lo = w.x; a_hi = w.y; ends = w.z; idx = w.w; }
  else { const uint2 w = *(const uint2*)blk; ends = w.x; idx = w.y; }
  const uint32_t c0 = ends & 0xffffu, c1 = ends >> 16;
  int e0[3], e1[3];
  expand565(c0, e0); expand565(c1, e1);
  const bool three = !kDxt5 && c0 <= c1;
  uint32_t pal[4];
  pal[0] = pack_rgb(e0[0], e0[1], e0[2]) | 0xff000000u;
  pal[1] = pack_rgb(e1[0], e1[1], e1[2]) | 0xff000000u;
  if (three) {
    pal[2] = pack_rgb((e0[0] + e1[0]) / 2, (e0[1] + e1[1]) / 2, (e0[2] + e1[2]) / 2) | 0xff000000u;
    pal[3] = 0u;
  } else {
    pal[2] = pack_rgb((2 * e0[0] + e1[0]) / 3, (2 * e0[1] + e1[1]) / 3, (2 * e0[2] + e1[2]) / 3) | 0xff000000u;
    pal[3] = pack_rgb((e0[0] + 2 * e1[0]) / 3, (e0[1] + 2 * e1[1]) / 3, (e0[2] + 2 * e1[2]) / 3) | 0xff000000u;
  }
  uint32_t at[8];
  uint64_t abits = 0;
  if (kDxt5) {
    const int a0 = a_lo & 0xff, a1 = (a_lo >> 8) & 0xff;
    at[0] = a0; at[1] = a1;
    if (a0 <= a1) {
#pragma unroll
      for (int i = 1; i < 5; ++i) at[1 + i] = (uint32_t)(((5 - i) * a0 + i * a1) / 5);
      at[6] = 0; at[7] = 255;
    } else {
#pragma unroll
      for (int i = 1; i < 7; ++i) at[1 + i] = (uint32_t)(((7 - i) * a0 + i * a1) / 7);
    }
    abits = ((uint64_t)a_hi << 16) | (a_lo >> 16);
  }
  const int bx = (int)(k % (uint32_t)nbx) * 4, by = (int)(k / (uint32_t)nbx) * 4;
  uchar4* __restrict__ img = rgba + (size_t)l * L.cw * L.ch;
  const bool whole = bx + 4 <= L.cw && (L.cw & 3) == 0;
#pragma unroll
  for (int py = 0; py < 4; ++py) {
    if (by + py >= L.ch) break;
    uint32_t px4[4];
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      uint32_t v = pal[(idx >> (8 * py + 2 * px)) & 3u];
      if (kDxt5) {
        uint32_t a = 0;
        const uint32_t code = (uint32_t)(abits >> (3 * (4 * py + px))) & 7u;
#pragma unroll
        for (int j = 0; j < 8; ++j) a = code == (uint32_t)j ? at[j] : a;      // register select, no scratch indexing
        v = (v & 0x00ffffffu) | (a << 24);
      }
      px4[px] = v;
    }
    uint32_t* row = (uint32_t*)(img + (size_t)(by + py) * L.cw + bx);
    if (whole) *(uint4*)row = make_uint4(px4[0], px4[1], px4[2], px4[3]);
    else
      for (int px = 0; px < 4 && bx + px < L.cw; ++px) row[px] = px4[px];
  }
}

__global__ __launch_bounds__(256) void k_wire_depth_f32(WireLayout L, float* __restrict__ raw) {
  const int l = blockIdx.y;
  const uint32_t n = (uint32_t)(L.w * L.h);
  const float* __restrict__ src = (const float*)(L.msg + (size_t)l * L.rec + L.cs);
  float* __restrict__ dst = raw + (size_t)l * n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}
// unsigned normalised: c / 255 (correctly rounded fp32 division, as the oracle)
__global__ __launch_bounds__(256) void k_wire_depth_u8(WireLayout L, float* __restrict__ raw) {
  const int l = blockIdx.y;
  const uint32_t quads = (uint32_t)(L.w * L.h) >> 2;
  const uint32_t* __restrict__ src = (const uint32_t*)(L.msg + (size_t)l * L.rec + L.cs);
  float4* __restrict__ dst = (float4*)(raw + (size_t)l * L.w * L.h);
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += gridDim.x * blockDim.x) {
    const uint32_t v = src[q];
    dst[q] = make_float4((float)(v & 0xffu) / 255.0f, (float)((v >> 8) & 0xffu) / 255.0f, (float)((v >> 16) & 0xffu) / 255.0f, (float)(v >> 24) / 255.0f);
  }
}

void launch_wire_unpack(hipStream_t st, const WireLayout& L, uchar4* rgba, float* raw) {
  const dim3 blk(256);
  auto grid = [&](uint32_t items) { uint32_t g = (items + 255) / 256; return dim3(g < 1 ? 1 : (g > 1024 ? 1024 : g), (unsigned)L.n); };
  if (L.cfmt == 0) hipLaunchKernelGGL(k_wire_rgb8, grid((uint32_t)(L.cw * L.ch) >> 2), blk, 0, st, L, rgba);
  else {
    const uint32_t nb = (uint32_t)(((L.cw + 3) >> 2) * ((L.ch + 3) >> 2));
    const dim3 g((nb + 255) / 256, (unsigned)L.n);
    if (L.cfmt == 1) hipLaunchKernelGGL(k_wire_dxt<false>, g, blk, 0, st, L, rgba);
    else hipLaunchKernelGGL(k_wire_dxt<true>, g, blk, 0, st, L, rgba);
  }
  if (L.dfmt == 0) hipLaunchKernelGGL(k_wire_depth_f32, grid((uint32_t)(L.w * L.h)), blk, 0, st, L, raw);
  else hipLaunchKernelGGL(k_wire_depth_u8, grid((uint32_t)(L.w * L.h) >> 2), blk, 0, st, L, raw);
}

}  // namespace rr
