// Image pre-processing P1..P5: glsl/pre_{morph,depth,boundary,normal,quality}.fs + inc_color.glsl as
// driven by NetKinectArray::processDepth / processTextures (framework/NetKinectArray.cpp:249-288, :309-426).
// One thread per depth pixel and layer; pass_TexCoord = (pixel + .5) / size, texSizeInv = 1 / size (:195).  A tap at
// pass_TexCoord + (dx, dy) * texSizeInv of a NEAREST array is the texel (x + dx, y + dy) clamped to the image (the coordinate
// is half a texel away from any texel border), LINEAR fetches go through the generic fp32 bilinear sampler.
// The products land directly in the layouts the path reads: the packed {depth, quality, silhouette} texel and the depth plane.
#include "sampling.hpp"
#include "bricks_dev.hpp"

namespace rr {

__device__ __forceinline__ int clamp_tap(int v, int n) { return min(max(v, 0), n - 1); }
// NEAREST fetch at pass_TexCoord + (dx,dy)*texSizeInv = texel clamp(x + dx), clamp(y + dy): the coordinate (x + .5) / W + dx * (1 / W), evaluated in
// fp32 as the shader does, is within W * 3e-7 texels of the centre of texel x + dx -- half a texel away from any border for every image this
// library accepts (tsdf_create: sides <= 65536) --, so floor(u * W) needs no float arithmetic (round 3 evaluated two IEEE divisions per tap)
__device__ __forceinline__ int tap_x(int x, int dx, int W) { return clamp_tap(x + dx, W); }
__device__ __forceinline__ int tap_y(int y, int dy, int H) { return clamp_tap(y + dy, H); }

// ---- pre_morph.fs: mode 0 = dilate(coords, 1) (:73-112, :123-127); mode 1 is a copy (:130-131)
// Two more layers of blocks ride along (round 4: what used to be launches of their own in front of the passes): layer N turns the frame's RGB8 colour
// into the RGBA8 the kernels read (four pixels per thread, as k_pack_frame_fused does), layer N + 1 zeroes the brick counters clearOccupiedBricks()
// flipped to.  Null pointers: nothing to do.
struct PreExtra { const uint8_t* rgb; uchar4* rgba; uint32_t n_px; uint4* zero; uint32_t zero_quads; };
__global__ __launch_bounds__(256) void k_pre_morph(PreParams P, PreBuffers B, PreExtra E) {
  if (B.cand_count && P.N > 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *B.cand_count = 0u;   // (the filter pass numbers this frame's candidate blocks)
  if ((int)blockIdx.z >= P.N) {
    const uint32_t nb = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
    if ((int)blockIdx.z == P.N) {                                         // ---- colour
      if (!E.rgb) return;
      const uint32_t n_quads = (E.n_px + 3u) >> 2;
      for (uint32_t q = b * blockDim.x + threadIdx.x; q < n_quads; q += nb * blockDim.x) {
        if (4u * q + 4u > E.n_px) {                                       // the last, partial quad: byte by byte
          for (uint32_t k = 4u * q; k < E.n_px; ++k) E.rgba[k] = make_uchar4(E.rgb[3 * k], E.rgb[3 * k + 1], E.rgb[3 * k + 2], 255);
          continue;
        }
        const uint3 s3 = ((const uint3*)E.rgb)[q];                        // bytes r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
        uint4 o;
        o.x = (s3.x & 0x00ffffffu) | 0xff000000u;
        o.y = (s3.x >> 24) | ((s3.y & 0x0000ffffu) << 8) | 0xff000000u;
        o.z = (s3.y >> 16) | ((s3.z & 0x000000ffu) << 16) | 0xff000000u;
        o.w = (s3.z >> 8) | 0xff000000u;
        ((uint4*)E.rgba)[q] = o;
      }
    } else if (E.zero) {                                                  // ---- brick counters
      for (uint32_t i = b * blockDim.x + threadIdx.x; i < E.zero_quads; i += nb * blockDim.x) E.zero[i] = make_uint4(0, 0, 0, 0);
    }
    return;
  }
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), l = blockIdx.z;
  if (x >= P.W || y >= P.H) return;
  const float* __restrict__ src = B.raw + (size_t)l * P.W * P.H;
  const float min_depth = 0.5f, max_depth = 4.5f, max_dist = 0.2f;
  const float depth = src[(size_t)tap_y(y, 0, P.H) * P.W + tap_x(x, 0, P.W)];
  float out;
  if (depth > min_depth && depth < max_depth) out = depth;              // is_valid && in_bbox (which returns true, :48)
  else {
    float s[9];
#pragma unroll
    for (int dy = -1; dy < 2; ++dy)
#pragma unroll
      for (int dx = -1; dx < 2; ++dx) s[(dy + 1) * 3 + dx + 1] = src[(size_t)tap_y(y, dy, P.H) * P.W + tap_x(x, dx, P.W)];
    float avg = 0.0f, num = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k) if (s[k] > min_depth && s[k] < max_depth) { avg += s[k]; num += 1.0f; }
    if (num == 0.0f) out = 0.0f;
    else {
      avg /= num;
      float nd = 0.0f;
      num = 0.0f;
#pragma unroll
      for (int k = 0; k < 9; ++k) if (s[k] > min_depth && s[k] < max_depth && fabsf(avg - s[k]) < max_dist) { nd += s[k]; num += 1.0f; }
      out = num > 0.0f ? nd / num : 0.0f;
    }
  }
  B.depth2[(size_t)l * P.W * P.H + (size_t)y * P.W + x] = out;
}

// inc_color.glsl:8-46 (the shader divides an already normalised colour by 255 again, :14-16: restated as written)
__device__ __forceinline__ float pivot_rgb(float n) { return (n > 0.04045f ? powf((n + 0.055f) / 1.055f, 2.4f) : n / 12.92f) * 100.0f; }
__device__ __forceinline__ float pivot_xyz(float n) { return n > 0.008856f ? powf(n, (float)(1.0 / 3.0)) : (903.3f * n + 16.0f) / 116.0f; }
__device__ __forceinline__ float3 rgb_to_lab(float3 rgb) {
  const float r = pivot_rgb(rgb.x / 255.0f), g = pivot_rgb(rgb.y / 255.0f), b = pivot_rgb(rgb.z / 255.0f);
  const float X = r * 0.4124f + g * 0.3576f + b * 0.1805f, Y = r * 0.2126f + g * 0.7152f + b * 0.0722f, Z = r * 0.0193f + g * 0.1192f + b * 0.9505f;
  const float x = pivot_xyz(X / 95.047f), y = pivot_xyz(Y / 100.000f), z = pivot_xyz(Z / 108.883f);
  return make_float3(fmaxf(0.0f, 116.0f * y - 16.0f), 500.0f * (x - y), 200.0f * (y - z));
}
__device__ __forceinline__ float3 color_bilinear_pre(const FrameImages& F, int layer, float u, float v) {   // RGB8 LINEAR
  const Axis X = axis_linear(u, F.cw), Y = axis_linear(v, F.ch);
  const uchar4* __restrict__ b = F.color + (size_t)layer * F.cw * F.ch;
  const uchar4 t00 = b[(size_t)Y.i0 * F.cw + X.i0], t10 = b[(size_t)Y.i0 * F.cw + X.i1];
  const uchar4 t01 = b[(size_t)Y.i1 * F.cw + X.i0], t11 = b[(size_t)Y.i1 * F.cw + X.i1];
  return make_float3(lerpf(lerpf(t00.x / 255.0f, t10.x / 255.0f, X.a), lerpf(t01.x / 255.0f, t11.x / 255.0f, X.a), Y.a),
                     lerpf(lerpf(t00.y / 255.0f, t10.y / 255.0f, X.a), lerpf(t01.y / 255.0f, t11.y / 255.0f, X.a), Y.a),
                     lerpf(lerpf(t00.z / 255.0f, t10.z / 255.0f, X.a), lerpf(t01.z / 255.0f, t11.z / 255.0f, X.a), Y.a));
}

// sample() / uncompress(), pre_depth.fs:51-72: the filter pass's input depth at a clamped tap
__device__ __forceinline__ float pre_sample(const PreParams& P, const float* __restrict__ src, int l, int yy, int xx) {
  float t = src[(size_t)clamp_tap(yy, P.H) * P.W + clamp_tap(xx, P.W)];
  if (P.compress[l] != 0) { const float sn = P.dc_scaled_near[l]; t = t < sn ? 0.0f : (t * t + 0.15f * sn) * P.dc_scale[l] + P.dc_near[l]; }
  return t;
}
// pre_depth.fs :131-143: the Lab colour the filter pass writes beside its depth -- the colour image looked up through cv_uv at the pixel's normalised
// input depth.  Round 4: only pre_boundary.fs reads that image, and only around its candidate pixels (1.3 % of the c2 frame, in 6 % of its 16 x 16
// blocks), while its six powf and eighteen IEEE divisions per pixel were two thirds of the filter pass's vector instructions: the boundary pass now
// evaluates it for the blocks that need it (k_pre_boundary), and the full image is produced when somebody asks for it (k_pre_lab, tsdf_download_preprocessed).
__device__ __forceinline__ float4 lab_of_pixel(const PreParams& P, const PreBuffers& B, const StreamTable& T, const FrameImages& F, int l, int x, int y) {
  const float depth = pre_sample(P, B.fdepth + (size_t)l * P.W * P.H, l, y, x);
  const float mn = P.cv_min[l], mx = P.cv_max[l];
  const float dn = (depth - mn) / (mx - mn);
  const float u = ((float)x + 0.5f) / (float)P.W, v = ((float)y + 0.5f) / (float)P.H;
  const StreamLut& L = T.s[l];
  const float2 cc = tex3d_rg(L.uv, L.uv_res, u, v, (dn <= 0.0f || dn >= 1.0f) ? 1.0f : dn);     // :136
  const float3 lab = rgb_to_lab(color_bilinear_pre(F, l, cc.x, cc.y));
  return make_float4(lab.x, lab.y, lab.z, 0.0f);
}
__global__ __launch_bounds__(256) void k_pre_lab(PreParams P, PreBuffers B, StreamTable T, FrameImages F) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), l = blockIdx.z;
  if (x < P.W && y < P.H) B.lab[(size_t)l * P.W * P.H + (size_t)y * P.W + x] = lab_of_pixel(P, B, T, F, l, x, y);
}

// ---- pre_depth.fs main() :129-154 with bilateral_filter :85-127.  The 13x13 window of a 16x16 pixel block is staged in
// LDS (28x28 depths, clamped taps), so the 169 taps per pixel are LDS reads.
__global__ __launch_bounds__(256) void k_pre_filter(PreParams P, PreBuffers B, StreamTable T) {
  __shared__ float s_d[28][29];
  const int l = blockIdx.z, bx = blockIdx.x * 16, by = blockIdx.y * 16;
  const float* __restrict__ src = B.fdepth + (size_t)l * P.W * P.H;
  const auto sample = [&](int yy, int xx) { return pre_sample(P, src, l, yy, xx); };
  const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4, x = bx + lx, y = by + ly;
  const bool inside = x < P.W && y < P.H;
  const float u = ((float)x + 0.5f) / (float)P.W, v = ((float)y + 0.5f) / (float)P.H;
  const float mn = P.cv_min[l], mx = P.cv_max[l];
  // the pixel's own part first (world position -> inside the bounding box?): the 13 x 13 window is staged in LDS only when some pixel of
  // the block lies in the box -- most blocks of a frame see background, and their 784 window loads and the barrier bought nothing (round 4)
  const float depth = inside ? sample(y, x) : 0.0f;
  const float dn = (depth - mn) / (mx - mn);
  const StreamLut& L = T.s[l];
  bool in_box = false;
  const size_t o = (size_t)l * P.W * P.H + (size_t)y * P.W + x;
  if (inside) {
    const float3 wp = tex3d_rgba_xyz(L.xyz, L.xyz_res, u, v, dn);
    in_box = wp.x >= P.bbox_min[0] && wp.y >= P.bbox_min[1] && wp.z >= P.bbox_min[2] && wp.x <= P.bbox_max[0] && wp.y <= P.bbox_max[1] && wp.z <= P.bbox_max[2];
  }
  const bool taps = __syncthreads_or(in_box && P.filter_textures) != 0;   // (workgroup-uniform)
  if (taps) {
    // the four rounds of window loads are requested together (as a loop the compiler waits for each round's load before the next: four L2 round trips)
    float wv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = min((int)threadIdx.x + 256 * r, 28 * 28 - 1), ty = i / 28, tx = i - ty * 28;
      wv[r] = sample(by + ty - 6, bx + tx - 6);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = (int)threadIdx.x + 256 * r, ty = i / 28, tx = i - ty * 28;
      if (i < 28 * 28) s_d[ty][tx] = wv[r];
    }
    __syncthreads();
  }
  float2 od = make_float2(0.0f, 0.0f);
  if (inside && in_box) {
    if (!P.filter_textures) od = make_float2(dn, 1.0f);
    else {
      const float dist_range_max = 0.35f * (depth / 4.5f), dist_range_max_inv = 1.0f / dist_range_max;   // :89-92
      // The 169 taps in the shader's order (the three sums are fp32: their order is part of the result), fully unrolled: computeGaussSpace of a
      // constant offset folds to a literal (round 3 evaluated an IEEE sqrt per tap at run time: 18 of its 42 vector instructions per tap),
      // a rejected tap adds +0.0f -- the identity for every value these sums can take (they start at +0 and never become -0) -- instead of
      // branching around the adds, and `num` counts every tap: 169, exactly.  16 vector instructions per tap.
      float depth_bf = 0.0f, w = 0.0f, w_range = 0.0f;
      const float num = 169.0f;
#pragma unroll
      for (int dy = -6; dy < 7; ++dy)
#pragma unroll
        for (int dx = -6; dx < 7; ++dx) {
          const float ds = s_d[ly + 6 + dy][lx + 6 + dx];
          const float dr = fabsf(ds - depth);
          const bool skip = (ds < mn) || (ds > mx) || (dr > dist_range_max);                           // is_outside, :74-76
          const float gs = 1.0f - __builtin_sqrtf((float)(dx * dx + dy * dy)) * (1.0f / 6.0f);         // computeGaussSpace (a compile-time constant)
          const float gr = 1.0f - fminf(dr, dist_range_max) * dist_range_max_inv;                       // computeGaussRange
          const float ws = gs * gr;
          const float wd = ws * ds;
          depth_bf += skip ? 0.0f : wd; w += skip ? 0.0f : ws; w_range += skip ? 0.0f : gr;
        }
      od = make_float2((depth_bf / w - mn) / (mx - mn), w_range / num);                                 // :124-126
    }
  }
  if (inside) B.depth_rg[o] = od;
  if (B.blk_flag) {                                                     // does the boundary pass compare colours in this block?  (its own test on its own input: k_pre_boundary)
    const bool cand = inside && !(od.x <= 0.0f) && !(od.y > 0.65f);
    const bool any = __syncthreads_or(cand) != 0;
    if (threadIdx.x == 0) {
      uint32_t f = 0u;
      if (any) { const uint32_t k = atomicAdd(B.cand_count, 1u); f = k + 1u; if (k < B.cand_cap) B.cand_list[k] = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; }
      B.blk_flag[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = f;
    }
  }
}

// ---- pre_boundary.fs main() :86-117, get_color_diff :37-55
// Round 4: 16 x 16-pixel blocks (the candidates -- pixels with a depth whose range quality is not above 0.65 -- run along silhouettes: 1.3 % of the pixels
// of a c2 frame, in 6 % of its blocks).  A block with a candidate first
// evaluates the Lab colour (lab_of_pixel) of its 22 x 22 neighbourhood into LDS -- the 5 x 5 taps of a bilinear fetch reach three pixels out --, the
// candidates' 26 bilinear fetches read that tile.  (A fetch at pass_TexCoord + (kx, ky) * texSizeInv, |k| <= 2, lies within W * 3e-7 texels of the centre of
// pixel (x + kx, y + ky) -- clamp_tap's argument above --, so its two texels per axis are among x + kx - 1 .. x + kx + 1: inside the tile.  The tile index is
// clamped all the same, for the memory's sake, not the value's.)
constexpr int kLabTile = 22, kLabHalo = 3;
// The launch is one-dimensional: its first cand_cap workgroups take the blocks the filter pass listed as holding a candidate (past the count: nothing to do),
// the others take the blocks in order and leave a listed block to its slot -- so the long blocks start at once instead of wherever the dispatcher reaches them
// (20.6 us of launch for 8 us of streaming + one 12 us block chain).
__global__ __launch_bounds__(256) void k_pre_boundary(PreParams P, PreBuffers B, StreamTable T, FrameImages F, int nbx, int nby) {
  __shared__ float4 s_lab[kLabTile][kLabTile];
  __shared__ float2 s_drg[20][20];
  uint32_t blk;
  if (B.blk_flag) {
    if (blockIdx.x < B.cand_cap) {
      if (blockIdx.x >= *B.cand_count) return;
      blk = B.cand_list[blockIdx.x];
    } else {
      blk = blockIdx.x - B.cand_cap;
      const uint32_t f = B.blk_flag[blk];
      if (f != 0u && f - 1u < B.cand_cap) return;                        // (a candidate block beyond the list's capacity is done here, in order)
    }
  } else blk = blockIdx.x;
  const int l = (int)(blk / (uint32_t)(nbx * nby)), rem = (int)(blk - (uint32_t)l * (uint32_t)(nbx * nby)), bx = (rem % nbx) * 16, by = (rem / nbx) * 16;
  const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4, x = bx + lx, y = by + ly;                     // a wave is 16 x 4 pixels: its stores are whole 128-byte / 64-byte row segments
  const bool inside = x < P.W && y < P.H;
  const size_t base = (size_t)l * P.W * P.H, o = base + (size_t)y * P.W + x;
  const float2* __restrict__ drg = B.depth_rg + base;
  float2 d = inside ? drg[(size_t)y * P.W + x] : make_float2(0.0f, 0.0f);
  const bool cand = inside && !(d.x <= 0.0f) && !(d.y > 0.65f);                         // valid_range, :27-30
  if (__syncthreads_or(cand) != 0) {
    // the candidates' 5 x 5 {depth, range quality} taps: staged too (with a load, a wait and a branch per tap the candidates' waves -- whose chain is
    // the launch's length -- paid 25 L2 round trips in a row)
    for (int i = threadIdx.x; i < 20 * 20; i += 256) {
      const int ty = i / 20, tx = i - ty * 20;
      s_drg[ty][tx] = drg[(size_t)clamp_tap(by - 2 + ty, P.H) * P.W + clamp_tap(bx - 2 + tx, P.W)];
    }
    for (int i = threadIdx.x; i < kLabTile * kLabTile; i += 256) {
      const int ty = i / kLabTile, tx = i - ty * kLabTile, px = bx - kLabHalo + tx, py = by - kLabHalo + ty;
      if (px >= 0 && py >= 0 && px < P.W && py < P.H) s_lab[ty][tx] = lab_of_pixel(P, B, T, F, l, px, py);
    }
    __syncthreads();
  }
  if (!inside) return;
  const auto lab_at = [&](int ix, int iy) {                                               // texel (ix, iy) of the Lab image (indices already clamped to it)
    const int tx = min(max(ix - (bx - kLabHalo), 0), kLabTile - 1), ty = min(max(iy - (by - kLabHalo), 0), kLabTile - 1);
    return s_lab[ty][tx];
  };
  const auto lab_bilinear = [&](float uu, float vv) {                                     // RGB32F LINEAR
    const Axis X = axis_linear(uu, P.W), Y = axis_linear(vv, P.H);
    const float4 t00 = lab_at(X.i0, Y.i0), t10 = lab_at(X.i1, Y.i0), t01 = lab_at(X.i0, Y.i1), t11 = lab_at(X.i1, Y.i1);
    return make_float3(lerpf(lerpf(t00.x, t10.x, X.a), lerpf(t01.x, t11.x, X.a), Y.a), lerpf(lerpf(t00.y, t10.y, X.a), lerpf(t01.y, t11.y, X.a), Y.a),
                       lerpf(lerpf(t00.z, t10.z, X.a), lerpf(t01.z, t11.z, X.a), Y.a));
  };
  const float u = ((float)x + 0.5f) / (float)P.W, v = ((float)y + 0.5f) / (float)P.H;
  const float tsx = 1.0f / (float)P.W, tsy = 1.0f / (float)P.H;
  if (d.x <= 0.0f) d.y = 0.0f;
  else if (!(d.y > 0.65f)) {                                                             // valid_range, :27-30
    const float3 color = lab_bilinear(u, v);
    float total = 0.0f, num = 0.0f;
    for (int ky = -2; ky < 3; ++ky)
      for (int kx = -2; kx < 3; ++kx) {
        const float2 s = s_drg[ly + 2 + ky][lx + 2 + kx];                // (staged with clamped coordinates: the texel tap_x / tap_y address)
        if (s.x > 0.0f && s.y > 0.65f) {
          num += 1.0f;
          const float3 cs = lab_bilinear(u + (float)kx * tsx, v + (float)ky * tsy);
          const float ex = color.x - cs.x, ey = color.y - cs.y, ez = color.z - cs.z;
          total += sqrtf(ex * ex + ey * ey + ez * ez);
        }
      }
    const float color_dist = (num < 16.0f * 0.5f) ? 1.0f : total / num;                  // total_samples = 16 (:23, :53)
    if (color_dist > 0.5f || !P.refine) { d.x = -1.0f; d.y = 0.1f; }
    else d.y = 1.0f;
  } else d.y = 0.0f;
  // (the packed texel {depth, quality, silhouette} is written once, by the quality pass, which takes the silhouette from depth_b: pre_boundary_silhouette)
  B.depth_b[o] = d;
  B.depth_plane[o] = d.x;
}
// the silhouette pre_boundary.fs wrote beside depth_b = {d.x, d.y'}: 1 exactly for the pixels that took its last branch (a depth, range quality above 0.65:
// d.y' = 0 and d.x untouched), 0 for the background (d.x <= 0) and for the candidates (d.y' = 0.1 or 1)
__device__ __forceinline__ float pre_boundary_silhouette(float2 db) { return (db.y == 0.0f && !(db.x <= 0.0f)) ? 1.0f : 0.0f; }

// ---- pre_normal.fs :26-56 including the mark_brick() call (:32-33)
// (round 4: a wave is an 8 x 8 cell of a 16 x 16 block, not a 64 x 4 strip: 14 % of the cells of a c2 frame hold a depth, 19 % of the strips)
__global__ __launch_bounds__(256) void k_pre_normal(PreParams P, PreBuffers B, StreamTable T, Bricks BR) {
  const int cell = threadIdx.x >> 6, ln = threadIdx.x & 63, l = blockIdx.z;
  const int x = blockIdx.x * 16 + ((cell & 1) << 3) + (ln & 7), y = blockIdx.y * 16 + ((cell >> 1) << 3) + (ln >> 3);
  bool own = false, nbr = false;
  uint32_t id_own = 0, id_nbr = 0;
  if (x < P.W && y < P.H) {
    const size_t base = (size_t)l * P.W * P.H;
    const float* __restrict__ dp = B.depth_plane + base;
    const float u = ((float)x + 0.5f) / (float)P.W, v = ((float)y + 0.5f) / (float)P.H;
    const float tsx = 1.0f / (float)P.W, tsy = 1.0f / (float)P.H;
    const float depth = dp[(size_t)tap_y(y, 0, P.H) * P.W + tap_x(x, 0, P.W)];
    float3 n = make_float3(0, 0, 0);
    if (!(depth <= 0.0f || depth >= 1.0f)) {
      const StreamLut& L = T.s[l];
      const float3 wp = tex3d_rgba_xyz(L.xyz, L.xyz_res, u, v, depth);
      mark_brick_ids(BR, wp, own, id_own, nbr, id_nbr);
      float dt = dp[(size_t)tap_y(y, 1, P.H) * P.W + x], db = dp[(size_t)tap_y(y, -1, P.H) * P.W + x];
      float dl = dp[(size_t)y * P.W + tap_x(x, -1, P.W)], dr = dp[(size_t)y * P.W + tap_x(x, 1, P.W)];
      dt = (dt <= 0.0f || dt >= 1.0f) ? depth : dt; db = (db <= 0.0f || db >= 1.0f) ? depth : db;
      dl = (dl <= 0.0f || dl >= 1.0f) ? depth : dl; dr = (dr <= 0.0f || dr >= 1.0f) ? depth : dr;
      const float3 wt = tex3d_rgba_xyz(L.xyz, L.xyz_res, u, v + tsy, dt), wb = tex3d_rgba_xyz(L.xyz, L.xyz_res, u, v - tsy, db);
      const float3 wl = tex3d_rgba_xyz(L.xyz, L.xyz_res, u - tsx, v, dl), wr = tex3d_rgba_xyz(L.xyz, L.xyz_res, u + tsx, v, dr);
      const float3 a = make_float3(wb.x - wt.x, wb.y - wt.y, wb.z - wt.z), b = make_float3(wl.x - wr.x, wl.y - wr.y, wl.z - wr.z);
      n = normalize3(make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x));
    }
    B.normal[base + (size_t)y * P.W + x] = make_float4(n.x, n.y, n.z, 0.0f);
  }
  wave_count(BR.counters, id_nbr, nbr);
  wave_count(BR.counters, id_own, own);
}

// ---- pre_quality.fs bilateral_filter :65-119 with normal_angle :43-48; same LDS staging as the filter pass
// Round 4: a wave is one 8 x 8-pixel cell of the block's 16 x 16 tile, and the pass -- the last writer of the packed texel -- also leaves the cell's
// {min depth, max depth, min silhouette, max silhouette} (what k_frame_ranges computed in a launch of its own; same NaN rule).
__global__ __launch_bounds__(256) void k_pre_quality(PreParams P, PreBuffers B, StreamTable T, float4* __restrict__ ranges, int rcw, int rch) {
  __shared__ float s_d[28][29];
  const int l = blockIdx.z, bx = blockIdx.x * 16, by = blockIdx.y * 16;
  const size_t base = (size_t)l * P.W * P.H;
  const float* __restrict__ src = B.depth_plane + base;
  const int cell = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int lx = ((cell & 1) << 3) + (ln & 7), ly = ((cell >> 1) << 3) + (ln >> 3), x = bx + lx, y = by + ly;
  const bool inside = x < P.W && y < P.H;
  const float u = ((float)x + 0.5f) / (float)P.W, v = ((float)y + 0.5f) / (float)P.H;
  const float depth = inside ? src[(size_t)y * P.W + x] : 0.0f;
  const bool valid = inside && !(depth <= 0.0f || depth >= 1.0f);
  // the window is staged only when some pixel of the block holds a depth (as in k_pre_filter)
  if (__syncthreads_or(valid) != 0) {
    float wv[4];                                                        // (four rounds of loads in flight together, as in k_pre_filter)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = min((int)threadIdx.x + 256 * r, 28 * 28 - 1), ty = i / 28, tx = i - ty * 28;
      wv[r] = src[(size_t)clamp_tap(by + ty - 6, P.H) * P.W + clamp_tap(bx + tx - 6, P.W)];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = (int)threadIdx.x + 256 * r, ty = i / 28, tx = i - ty * 28;
      if (i < 28 * 28) s_d[ty][tx] = wv[r];
    }
    __syncthreads();
  }
  float q = 0.0f;
  if (valid) {
    const float dist_range_max = 0.35f * (depth / 1.0f), dist_range_max_inv = 1.0f / dist_range_max;
    // as in k_pre_filter: unrolled, branch-free; `border` and `num` count taps (integers below 2^24: exact in any order)
    float w_range = 0.0f;
    int n_border = 0;
    const float num = 169.0f;
#pragma unroll
    for (int dy = -6; dy < 7; ++dy)
#pragma unroll
      for (int dx = -6; dx < 7; ++dx) {
        const float ds = s_d[ly + 6 + dy][lx + 6 + dx];
        const float dr = fabsf(ds - depth);
        const bool out = (ds <= 0.0f || ds >= 1.0f) || dr > dist_range_max;
        n_border += out ? 1 : 0;
        const float g = 1.0f - fminf(dr, dist_range_max) * dist_range_max_inv;
        w_range += out ? 0.0f : g;
      }
    const float border = (float)n_border;
    const float lateral = 1.0f - border / num;
    q = powf(lateral, 6.0f);
    q *= powf(w_range / num, 6.0f);
    q /= depth * 6.5f;
    const Axis X = axis_linear(u, P.W), Y = axis_linear(v, P.H);                          // kinect_normals LINEAR
    const float4* __restrict__ nm = B.normal + base;
    const float4 t00 = nm[(size_t)Y.i0 * P.W + X.i0], t10 = nm[(size_t)Y.i0 * P.W + X.i1], t01 = nm[(size_t)Y.i1 * P.W + X.i0], t11 = nm[(size_t)Y.i1 * P.W + X.i1];
    const float3 wn = make_float3(lerpf(lerpf(t00.x, t10.x, X.a), lerpf(t01.x, t11.x, X.a), Y.a), lerpf(lerpf(t00.y, t10.y, X.a), lerpf(t01.y, t11.y, X.a), Y.a),
                                  lerpf(lerpf(t00.z, t10.z, X.a), lerpf(t01.z, t11.z, X.a), Y.a));
    const float3 wp = tex3d_rgba_xyz(T.s[l].xyz, T.s[l].xyz_res, u, v, depth);
    const float3 tc = normalize3(make_float3(P.cam[l][0] - wp.x, P.cam[l][1] - wp.y, P.cam[l][2] - wp.z));
    const float angle = tc.x * wn.x + tc.y * wn.y + tc.z * wn.z;
    q *= powf(angle, 2.0f);
  }
  const float inf = __builtin_inff();
  float d0 = inf, d1 = -inf, s0 = inf, s1 = -inf;
  bool nan = false;
  if (inside) {
    const size_t o = base + (size_t)y * P.W + x;
    const float sil = pre_boundary_silhouette(B.depth_b[o]);
    B.dqs[o] = make_float4(depth, q, sil, 0.0f);
    d0 = d1 = depth; s0 = s1 = sil;
    nan = (depth != depth) || (sil != sil);
  }
  if (ranges) {
    d0 = wave_min_f32(d0); d1 = wave_max_f32(d1); s0 = wave_min_f32(s0); s1 = wave_max_f32(s1);
    if (__ballot(nan) != 0ull) { d0 = s0 = -inf; d1 = s1 = inf; }          // a NaN anywhere poisons the cell's range (k_frame_ranges)
    const int cx = (bx >> 3) + (cell & 1), cy = (by >> 3) + (cell >> 1);
    if (ln == 0 && cx < rcw && cy < rch) ranges[((size_t)l * rch + cy) * rcw + cx] = make_float4(d0, d1, s0, s1);
  }
}

void launch_preprocess(hipStream_t st, const PreParams& P, const PreBuffers& B, const StreamTable& T, const FrameImages& F, const Bricks& BR, float4* ranges,
                       const uint8_t* rgb, uchar4* rgba, size_t n_color_px, uint32_t* zero, uint32_t zero_words, int only) {
  const dim3 rows((P.W + 63) / 64, (P.H + 3) / 4, P.N), tiles((P.W + 15) / 16, (P.H + 15) / 16, P.N);
  const PreExtra E{rgb, rgba, (uint32_t)n_color_px, (uint4*)zero, zero_words >> 2};
  if (!only || only == 1) hipLaunchKernelGGL(k_pre_morph, dim3(rows.x, rows.y, P.N + ((rgb || zero) ? 2 : 0)), dim3(256), 0, st, P, B, E);
  if (only == 6 && (rgb || zero)) {                                      // the two extra layers alone (tsdf_frame_raw_dev: behind the lane's gate, the morph pass in front of it)
    PreParams P0 = P; P0.N = 0;
    hipLaunchKernelGGL(k_pre_morph, dim3(rows.x, rows.y, 2), dim3(256), 0, st, P0, B, E);
  }
  if (!only || only == 2) hipLaunchKernelGGL(k_pre_filter, tiles, dim3(256), 0, st, P, B, T);
  if (!only || only == 3) hipLaunchKernelGGL(k_pre_boundary, dim3(tiles.x * tiles.y * tiles.z + (B.blk_flag ? B.cand_cap : 0u)), dim3(256), 0, st, P, B, T, F, (int)tiles.x, (int)tiles.y);
  if (!only || only == 4) hipLaunchKernelGGL(k_pre_normal, tiles, dim3(256), 0, st, P, B, T, BR);
  if (!only || only == 5) hipLaunchKernelGGL(k_pre_quality, tiles, dim3(256), 0, st, P, B, T, ranges, (P.W + 7) / 8, (P.H + 7) / 8);
}

// the Lab image of the filter pass (PreBuffers::lab), for callers that read it back: the passes themselves no longer write it (lab_of_pixel)
void launch_pre_lab(hipStream_t st, const PreParams& P, const PreBuffers& B, const StreamTable& T, const FrameImages& F) {
  hipLaunchKernelGGL(k_pre_lab, dim3((P.W + 63) / 64, (P.H + 3) / 4, P.N), dim3(256), 0, st, P, B, T, F);
}

}  // namespace rr
