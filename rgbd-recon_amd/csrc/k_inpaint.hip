// K3 + K4 (+ K5' fused away): the hole-filling pyramid of fillColors(), recon_integration.cpp:279-338.
//
// The reference ping-pongs two 1.5w x h atlases: after the raymarch and after every pyramid level it
// re-renders the whole atlas into a w-wide "squeezed" copy (framebuffer_transfer.fs:13-17, 10 full-screen
// passes at 1280x720) and tsdf_inpaint.fs reads that copy with a 2/3 x-scale.  Here there is ONE atlas
// and no copy: inpaint composes the squeeze mapping analytically (squeezed column s holds atlas column
// floor(1.5 (s + .5))), and reproduces what the freshly cleared copy would contain for texels that were
// not written yet.  Results are identical to the two-atlas sequence.
#include "sampling.hpp"

namespace rr {

struct Texel { float4 c; float d; };

// texelFetch from the squeezed copy S as it exists while level `lod + 1` is being built, split into address + class so
// that all 16 taps of a pixel can be LOADED UNCONDITIONALLY and back to back (with early returns per tap the compiler waits
// for each tap's loads before it evaluates the next tap's branches: 16 dependent L2 round trips per pixel, ~6 us per
// pyramid level whatever its size).
//   cls 0: out-of-range texelFetch -> 0 (Appendix A), also atlas column/row past the allocation
//   cls 1: the freshly cleared copy's value (ViewLod::enable, view_lod.cpp:75-81): columns >= w of S are never written,
//          and pyramid levels > lod are still cleared in the reference when this level is built
//   cls 2: atlas texel at `off`: squeezed column s holds atlas column int(pass_TexCoord.x * resolution_tex.x)
// The 4 x 4 window is a product of 4 columns and 4 rows: the two IEEE divisions of the mapping are done once per column / row
// (8 per pixel instead of 32) and a tap's class and offset are combined from its column's and its row's part.
struct TapAddr { uint32_t off; int cls; };
struct TapCol { int c; bool oob, past_w, in_pyramid, past_aw; };      // squeezed column s -> atlas column c
struct TapRow { int sy; bool oob, below_level, past_h; };            // row y -> atlas row sy
__device__ __forceinline__ TapCol squeezed_col(const Atlas& A, int w, int s) {
  TapCol t;
  const float tu = ((float)s + 0.5f) / (float)w;                       // pass_TexCoord.x
  t.c = (int)(tu * (float)A.aw);                                       // ivec2(pass_TexCoord * resolution_tex).x
  t.oob = s < 0 || s >= A.aw; t.past_w = s >= w; t.in_pyramid = t.c >= w; t.past_aw = t.c >= A.aw;
  return t;
}
__device__ __forceinline__ TapRow squeezed_row(const Atlas& A, int lod, int y) {
  TapRow t;
  const float tv = ((float)y + 0.5f) / (float)A.h;
  t.sy = (int)(tv * (float)A.h);
  t.oob = y < 0 || y >= A.h; t.below_level = lod >= 1 && t.sy >= A.off[lod][1]; t.past_h = t.sy >= A.h;
  return t;
}
__device__ __forceinline__ TapAddr squeezed_addr(const Atlas& A, const TapCol& C, const TapRow& R) {
  const bool oob = C.oob || R.oob;
  const bool cleared = C.past_w || (C.in_pyramid && !R.below_level);
  const bool zero2 = C.past_aw || R.past_h;
  TapAddr t;
  t.cls = oob ? 0 : (cleared ? 1 : (zero2 ? 0 : 2));
  t.off = t.cls == 2 ? (uint32_t)R.sy * (uint32_t)A.aw + (uint32_t)C.c : 0u;
  return t;
}

// tsdf_inpaint.fs:34-89 for pixel (lx0, ly0) of level lod + 1
__device__ __forceinline__ void inpaint_pixel(const Atlas& A, int w, int lod, int lx0, int ly0) {
  const int rx = A.res[lod + 1][0], ry = A.res[lod + 1][1], ox = A.off[lod + 1][0], oy = A.off[lod + 1][1];
  const int fx = ox + lx0, fy = oy + ly0;                 // gl_FragCoord (pixel_center_integer)
  const float tcx = ((float)fx - (float)ox) / (float)rx, tcy = ((float)fy - (float)oy) / (float)ry;   // :37
  const int lx = (int)((float)A.off[lod][0] + (float)A.res[lod][0] * tcx);                           // to_lod_pos, :30-32
  const int ly = (int)((float)A.off[lod][1] + (float)A.res[lod][1] * tcy);
  const int pxi = (int)((float)lx * (float)(2.0 / 3.0)), pyi = (int)((float)ly * 1.0f);               // :38
  TapCol tcol[4];
  TapRow trow[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { tcol[k] = squeezed_col(A, w, pxi + k - 1); trow[k] = squeezed_row(A, lod, pyi + k - 1); }
  TapAddr ta[16];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) ta[x + y * 4] = squeezed_addr(A, tcol[x], trow[y]);                    // :45-47
  float4 tc[16];
  float td_[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { tc[i] = A.color[ta[i].off]; td_[i] = A.depth[ta[i].off]; }
  float depth_av = 0.0f;
  int num = 0;
  float4 smp[16];
  // the shader visits the window x-outer / y-inner (:43-44) and stores tap (x, y) at x + 4 y (:55): depth_av is summed in
  // THAT order, the totals below in index order -- fp32 sums, so the two orders are part of the result
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const int i = x + y * 4;
      float4 c = tc[i];
      float d = td_[i];
      if (ta[i].cls == 1) { c = make_float4(0.0f, 1.0f, 0.0f, 0.0f); d = 1.0f; }   // ViewLod::enable clear, view_lod.cpp:75-81
      if (ta[i].cls == 0) { c = make_float4(0.0f, 0.0f, 0.0f, 0.0f); d = 0.0f; }
      if (c.w <= 0.0f) c.x = -1.0f;
      else { depth_av += d; ++num; }
      smp[i] = make_float4(c.x, c.y, c.z, d);
    }
  const size_t o = (size_t)fy * A.aw + fx;
  if (num == 0) {                                                                                      // :59-68
    const float d = smp[1 + 1 * 4].w;                     // the centre tap (pxi, pyi)
    A.depth[o] = d;
    A.color[o] = (d < 1.0f) ? make_float4(0.0f, 0.0f, 0.0f, -1.0f) : make_float4(0.0f, 1.0f, 0.0f, 0.0f);
    return;
  }
  depth_av /= (float)num;
  float tr = 0.0f, tg = 0.0f, tb = 0.0f, td = 0.0f, tw = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (smp[i].x >= 0.0f && smp[i].w >= depth_av) {
      tr += smp[i].x * 1.0f; tg += smp[i].y * 1.0f; tb += smp[i].z * 1.0f; td += smp[i].w * 1.0f; tw += 1.0f;
    }
  A.color[o] = make_float4(tr / tw, tg / tw, tb / tw, 1.0f);
  A.depth[o] = td / tw;
}

// Dirty tiles (round 3).  92 % of the c2 picture is background: a level-0 tile (8x8 pixels) without a brick under it holds the clear value,
// and a pixel of level l + 1 whose whole window lies in clear tiles of level l comes out as the clear value again (no valid tap: depth of
// the centre tap = 1 -> colour (0,1,0,0), tsdf_inpaint.fs:59-68).  The march leaves one byte per level-0 tile -- the union of the tiles
// the last three draws touched (k_march: whatever THIS pyramid and the framebuffer may still hold from their previous use lies inside it) --
// and every level derives its own tile mask from its parent's: the window of level-(l + 1) pixel (x, y) is level-l columns 2x-3 .. 2x+6
// (through the 2/3 squeeze, res[l] = 2 res[l+1] or 2 res[l+1] + 1) and rows 2y-1 .. 2y+3, i.e. tile T reads parent tiles 2T-1 .. 2T+2 on both
// axes.  One exception: past its last row a level's window runs into the rows of level l - 1 (off[l].y + res[l].y = off[l-1].y), so
// from level 3 on the last tile row is always computed.  Clean tiles are not touched: they hold the clear value since the last full pass.
struct LevelMask { const uint8_t* parent; uint8_t* child; int pnx, pny, cnx, cny; int force_last_row; };   // parent == nullptr: every tile
__global__ __launch_bounds__(256) void k_inpaint_level(Atlas A, int w, int lod, LevelMask M) {
  if (M.parent) {
    __shared__ int s_dirty[4];
    if (threadIdx.x < 4) s_dirty[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x < 64) {                                              // tile q of the block's 2 x 2, parent k of its 4 x 4
      const int q = threadIdx.x >> 4, k = threadIdx.x & 15;
      const int tx = blockIdx.x * 2 + (q & 1), ty = blockIdx.y * 2 + (q >> 1);
      const int px = 2 * tx - 1 + (k & 3), py = 2 * ty - 1 + (k >> 2);
      if (tx < M.cnx && ty < M.cny) {
        if (px >= 0 && py >= 0 && px < M.pnx && py < M.pny && M.parent[py * M.pnx + px]) atomicOr(&s_dirty[q], 1);
        if (M.force_last_row && ty == M.cny - 1) atomicOr(&s_dirty[q], 1);
      }
    }
    __syncthreads();
    if (threadIdx.x < 4 && M.child) {
      const int tx = blockIdx.x * 2 + (threadIdx.x & 1), ty = blockIdx.y * 2 + (threadIdx.x >> 1);
      if (tx < M.cnx && ty < M.cny) M.child[ty * M.cnx + tx] = (uint8_t)s_dirty[threadIdx.x];
    }
    const int q = ((threadIdx.x >> 4) >> 3) * 2 + ((threadIdx.x & 15) >> 3);
    if (!s_dirty[q]) return;
  }
  const int lx0 = blockIdx.x * 16 + (threadIdx.x & 15), ly0 = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (lx0 >= A.res[lod + 1][0] || ly0 >= A.res[lod + 1][1]) return;
  inpaint_pixel(A, w, lod, lx0, ly0);
}
// The small tail of the pyramid (levels first_lod + 1 .. num_lods - 1, a few thousand pixels in all) in ONE
// workgroup: each level is a loop over its pixels, a workgroup barrier orders a level's global stores before
// the next level's loads (same CU, workgroup-scope fence), and the launch boundaries between them disappear.
#ifdef RR_TAIL_TRACE     // tools/build_k1_variant.sh tailtrace "-DRR_TAIL_TRACE" k_inpaint: s_memtime stamps of thread 0 at the level boundaries of the last launch (tools/tail_trace.py)
__device__ unsigned long long g_tail_trace[32];
#define RR_TAIL_STAMP(k) do { if (threadIdx.x == 0) g_tail_trace[k] = __builtin_readcyclecounter(); } while (0)
#else
#define RR_TAIL_STAMP(k) do { } while (0)
#endif
__global__ __launch_bounds__(1024) void k_inpaint_tail(Atlas A, int w, int first_lod) {
  RR_TAIL_STAMP(0);
  RR_TAIL_STAMP(1);
  for (int lod = first_lod; lod + 1 < A.num_lods; ++lod) {
    const int rx = A.res[lod + 1][0], n = rx * A.res[lod + 1][1];
    for (int i = threadIdx.x; i < n; i += blockDim.x) inpaint_pixel(A, w, lod, i % rx, i / rx);
    RR_TAIL_STAMP(2 + 2 * (lod - first_lod));
    __threadfence_block();
    __syncthreads();
    RR_TAIL_STAMP(3 + 2 * (lod - first_lod));
  }
}
constexpr int kTailPixels = 1024;   // levels with at most this many pixels go to the fused tail (one pass of 1024 threads each)
static void launch_inpaint_level_masked(hipStream_t st, const Atlas& A, int lod, const LevelMask& M) {
  const int w = A.res[0][0];
  dim3 grid((A.res[lod + 1][0] + 15) / 16, (A.res[lod + 1][1] + 15) / 16);
  hipLaunchKernelGGL(k_inpaint_level, grid, dim3(256), 0, st, A, w, lod, M);
}
void launch_inpaint_level(hipStream_t st, const Atlas& A, int lod) { launch_inpaint_level_masked(st, A, lod, LevelMask{}); }
// tile_mask: one byte per level-0 tile ((w + 7) / 8 per row), or nullptr = every tile; lvl_mask[0 / 1]: scratch for the masks of levels 1 / 2
// (each at least ((w / 2 + 7) / 8) * ((h / 2 + 7) / 8) bytes)
void launch_inpaint_pyramid(hipStream_t st, const Atlas& A, const uint8_t* tile_mask, uint8_t* const lvl_mask[2]) {
  const int w = A.res[0][0];
  int lod = 0;
  const uint8_t* parent = tile_mask;
  for (; lod + 1 < A.num_lods && A.res[lod + 1][0] * A.res[lod + 1][1] > kTailPixels; ++lod) {
    LevelMask M{};
    if (parent && lod < 3) {                                             // levels 1 .. 3 by their tiles; the (small) rest in full
      M.parent = parent;
      M.pnx = (A.res[lod][0] + 7) / 8; M.pny = (A.res[lod][1] + 7) / 8; M.cnx = (A.res[lod + 1][0] + 7) / 8; M.cny = (A.res[lod + 1][1] + 7) / 8;
      M.child = lod < 2 ? lvl_mask[lod] : nullptr;
      M.force_last_row = lod + 1 >= 3 ? 1 : 0;
    }
    launch_inpaint_level_masked(st, A, lod, M);
    parent = M.child;
  }
  if (lod + 1 < A.num_lods) hipLaunchKernelGGL(k_inpaint_tail, dim3(1), dim3(1024), 0, st, A, w, lod);
}

__device__ __forceinline__ int mirror_idx(int i, int n) {   // GL_MIRRORED_REPEAT, view_lod.cpp:52-53
  const int p = 2 * n;
  const int m = ((i % p) + p) % p;
  return m < n ? m : p - 1 - m;
}
__device__ __forceinline__ float4 atlas_bilinear(const Atlas& A, float u, float v) {
  const float fx = u * (float)A.aw - 0.5f, fy = v * (float)A.h - 0.5f;
  const float x0f = floorf(fx), y0f = floorf(fy);
  const float ax = fx - x0f, ay = fy - y0f;
  const int x0 = mirror_idx((int)x0f, A.aw), x1 = mirror_idx((int)x0f + 1, A.aw);
  const int y0 = mirror_idx((int)y0f, A.h), y1 = mirror_idx((int)y0f + 1, A.h);
  const float4 t00 = A.color[(size_t)y0 * A.aw + x0], t10 = A.color[(size_t)y0 * A.aw + x1];
  const float4 t01 = A.color[(size_t)y1 * A.aw + x0], t11 = A.color[(size_t)y1 * A.aw + x1];
  return make_float4(lerpf(lerpf(t00.x, t10.x, ax), lerpf(t01.x, t11.x, ax), ay), lerpf(lerpf(t00.y, t10.y, ax), lerpf(t01.y, t11.y, ax), ay),
                     lerpf(lerpf(t00.z, t10.z, ax), lerpf(t01.z, t11.z, ax), ay), lerpf(lerpf(t00.w, t10.w, ax), lerpf(t01.w, t11.w, ax), ay));
}
__device__ __forceinline__ Texel fetch_atlas(const Atlas& A, int x, int y) {
  Texel t;
  if (x < 0 || y < 0 || x >= A.aw || y >= A.h) { t.c = make_float4(0, 0, 0, 0); t.d = 0.0f; return t; }
  t.c = A.color[(size_t)y * A.aw + x]; t.d = A.depth[(size_t)y * A.aw + x];
  return t;
}

// tsdf_colorfill.fs:30-55 with depth func LESS against the cleared framebuffer (:313)
// glColorMask of the anaglyph modes on the write into the "default framebuffer" (recon_integration.cpp:212-216,321-333): masked
// channels keep what the colour buffer held -- the cleared 0 (glClear, kinect_client.cpp:609-610,620) or, when the client cleared
// only the depth buffer before this draw (:627), the previous draw's colour
__device__ __forceinline__ float4 masked_write(float4 out, float4 base, int mask) {
  if (mask == 1) return make_float4(out.x, base.y, base.z, base.w);          // GL_TRUE, GL_FALSE, GL_FALSE, GL_FALSE
  if (mask == 2) return make_float4(base.x, out.y, out.z, base.w);           // GL_FALSE, GL_TRUE, GL_TRUE, GL_FALSE
  return out;
}
__global__ __launch_bounds__(256) void k_colorfill(Atlas A, int w, int h, float4* __restrict__ fb_c, float* __restrict__ fb_d, int mask, int keep,
                                                   const uint8_t* __restrict__ tile_mask) {
  const int px = blockIdx.x * 16 + (threadIdx.x & 15), py = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (px >= w || py >= h) return;
  // a tile no brick has been under for three draws: the framebuffer holds the background there since the last full pass (dirty tiles, above)
  if (tile_mask && !tile_mask[(py >> 3) * ((w + 7) >> 3) + (px >> 3)]) return;
  const float tcx = (float)px / (float)A.res[0][0], tcy = (float)py / (float)A.res[0][1];             // :32
  const size_t o = (size_t)py * w + px;
  // depth comes from level 0 alone (:54) and the fragment only lands if it beats the cleared depth (GL_LESS, :313):
  // background pixels need none of the colour work
  const float d0 = fetch_atlas(A, (int)((float)A.off[0][0] + (float)A.res[0][0] * tcx), (int)((float)A.off[0][1] + (float)A.res[0][1] * tcy)).d;
  const float4 base = keep ? fb_c[o] : make_float4(0, 0, 0, 0);
  if (!(d0 < 1.0f)) {
    fb_c[o] = base;
    fb_d[o] = 1.0f;
    return;
  }
  float4 out = make_float4(0, 0, 0, 0);
  int level = 0;
  for (; level < A.num_lods; ++level) {                                                               // :36-40
    const int cx = (int)((float)A.off[level][0] + (float)A.res[level][0] * tcx);
    const int cy = (int)((float)A.off[level][1] + (float)A.res[level][1] * tcy);
    out = fetch_atlas(A, cx, cy).c;
    if (out.w > 0.0f) break;
  }
  if (level > 0) {                                                                                    // :42-51
    const float ptx = ((float)px + 0.5f) / (float)w, pty = ((float)py + 0.5f) / (float)h;
    float p[2][2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {                          // to_lod_pos2(level + 1), (level + 2); slots past num_lods are zero
      const int l = level + 1 + k;
      const bool in = l < A.num_lods;
      const int o0 = in ? A.off[l][0] : 0, o1 = in ? A.off[l][1] : 0, r0 = in ? A.res[l][0] : 0, r1 = in ? A.res[l][1] : 0;
      p[k][0] = fminf(fmaxf((float)o0 + (float)r0 * ptx, (float)o0 + 0.5f), (float)(o0 + r0) - 0.5f);
      p[k][1] = fminf(fmaxf((float)o1 + (float)r1 * pty, (float)o1 + 0.5f), (float)(o1 + r1) - 0.5f);
    }
    const float rix = 1.0f / (float)A.aw, riy = 1.0f / (float)A.h;                                    // resolution_inv, :497
    const float4 c1 = atlas_bilinear(A, p[0][0] * rix, p[0][1] * riy);
    const float4 c2 = atlas_bilinear(A, p[1][0] * rix, p[1][1] * riy);
    const float w1 = sqrtf(ptx * ptx + pty * pty);                                                    // :47 (Appendix C.7)
    const float w2 = 1.0f - w1;
    out = make_float4((c1.x * w1 + c2.x * w2) / (w1 + w2), (c1.y * w1 + c2.y * w2) / (w1 + w2),
                      (c1.z * w1 + c2.z * w2) / (w1 + w2), (c1.w * w1 + c2.w * w2) / (w1 + w2));
  }
  fb_c[o] = masked_write(out, base, mask);
  fb_d[o] = d0;
}
void launch_colorfill(hipStream_t st, const Atlas& A, int w, int h, float4* fb_color, float* fb_depth, int mask, int keep_color, const uint8_t* tile_mask) {
  dim3 grid((w + 15) / 16, (h + 15) / 16);
  hipLaunchKernelGGL(k_colorfill, grid, dim3(256), 0, st, A, w, h, fb_color, fb_depth, mask, keep_color, tile_mask);
}
// draw() without hole filling but with a colour mask / an uncleared colour buffer: fragments (depth < 1 in the march target) write
// their unmasked channels, everything else keeps the colour buffer; the depth buffer was cleared before the draw
__global__ __launch_bounds__(256) void k_resolve_masked(Atlas A, int w, int h, float4* __restrict__ fb_c, float* __restrict__ fb_d, int mask, int keep) {
  const int px = blockIdx.x * 16 + (threadIdx.x & 15), py = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (px >= w || py >= h) return;
  const size_t o = (size_t)py * w + px, a = (size_t)py * A.aw + px;
  const float d = A.depth[a];
  const float4 base = keep ? fb_c[o] : make_float4(0, 0, 0, 0);
  fb_c[o] = d < 1.0f ? masked_write(A.color[a], base, mask) : base;
  fb_d[o] = d < 1.0f ? d : 1.0f;
}
void launch_resolve_masked(hipStream_t st, const Atlas& A, int w, int h, float4* fb_color, float* fb_depth, int mask, int keep_color) {
  dim3 grid((w + 15) / 16, (h + 15) / 16);
  hipLaunchKernelGGL(k_resolve_masked, grid, dim3(256), 0, st, A, w, h, fb_color, fb_depth, mask, keep_color);
}

__global__ __launch_bounds__(256) void k_clear_image(float4* __restrict__ c, float* __restrict__ d, size_t n, float4 cv, float dv) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    c[i] = cv;
    d[i] = dv;
  }
}
void launch_clear_image(hipStream_t st, float4* color, float* depth, size_t n, float4 c, float d) {
  size_t g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(k_clear_image, dim3((unsigned)g), dim3(256), 0, st, color, depth, n, c, d);
}

}  // namespace rr
#ifdef RR_TAIL_TRACE
extern "C" __attribute__((visibility("default"))) int tsdf_debug_tail_trace(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rr::g_tail_trace), sizeof(unsigned long long) * (size_t)(n < 32 ? n : 32));
}
#endif
