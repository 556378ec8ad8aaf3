// oracle/tsdf_oracle.cpp -- TEST INFRASTRUCTURE ONLY (parity checker + CPU baseline).
//
// A plain CPU restatement of rgbd-recon's TSDF hot path: kinect::ReconIntegration
// (framework/reconstruction/recon_integration.cpp) and the GLSL programs it drives.
// Nothing under rgbd-recon_amd/ may include, link or call this file; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path
//  and its arithmetic is GLSL that cannot be executed here (no GL context),
// so this restatement is pinned only by line-by-line review against the cited sources and by
// hand-computed known-answer tests (tests/test_oracle_*.py).
// Exceptions, pinned by the reference's OWN code compiled from its sources into oracle/_ref (oracle/ref/Makefile):
//   * orc_decode_dxt                      == external/squish DecompressImage   (tests/test_oracle_ingest.py)
//   * the .stream record layout           == framework/io/FileBuffer           (tests/test_oracle_ingest.py)
//   * (product side) .cv_* volume files   == calibration_volume.hpp read/write (tests/test_calib_io.py)
//   * make_view / orc_view_matrices       == draw()'s matrix block on external/gloost Matrix + vendored glm, to 1e-6 of each
//                                            matrix's largest element (tests/test_view_math.py; the product's tsdf_view_matrices too)
// Everything else below -- TSDF path, pre-processing, inverse-LUT builder (CGAL absent), point / triangle-grid back-ends --
// is restated, not run.
//
// Texture sampling follows OpenGL 4.4 §8.14 with the sampler state the reference sets
// (SURVEY.md Appendix A): fp32, lerp(a,b,t) = a + (b-a)*t applied x, then y, then z.
// Build with -ffp-contract=off so no FMA contraction changes a rounding.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(vec3 a) { return sqrtf(dot(a, a)); }
inline vec3 normalize(vec3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
inline float lerp(float a, float b, float t) { return a + (b - a) * t; }
inline float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// column-major 4x4, m[col*4+row] (OpenGL / gloost / glm layout)
struct mat4 { float m[16]; };
inline vec4 mul(const mat4& a, vec4 v) {
  return {a.m[0] * v.x + a.m[4] * v.y + a.m[8] * v.z + a.m[12] * v.w,
          a.m[1] * v.x + a.m[5] * v.y + a.m[9] * v.z + a.m[13] * v.w,
          a.m[2] * v.x + a.m[6] * v.y + a.m[10] * v.z + a.m[14] * v.w,
          a.m[3] * v.x + a.m[7] * v.y + a.m[11] * v.z + a.m[15] * v.w};
}
// matrix helpers run in double and round once to float (documented in DESIGN.md: the reference
// mixes host glm/gloost fp32 and in-shader inverse(); neither is reproducible bit-for-bit).
void mul_d(const double* a, const double* b, double* o) {
  double r[16];
  for (int c = 0; c < 4; ++c)
    for (int rr = 0; rr < 4; ++rr) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a[k * 4 + rr] * b[c * 4 + k];
      r[c * 4 + rr] = s;
    }
  memcpy(o, r, sizeof(r));
}
bool inverse_d(const double* m, double* o) {
  double inv[16];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  if (det == 0) return false;
  det = 1.0 / det;
  for (int i = 0; i < 16; ++i) o[i] = inv[i] * det;
  return true;
}
void to_d(const float* f, double* d) { for (int i = 0; i < 16; ++i) d[i] = f[i]; }
mat4 to_f(const double* d) { mat4 r; for (int i = 0; i < 16; ++i) r.m[i] = (float)d[i]; return r; }

// ---------------------------------------------------------------- GL sampling (Appendix A)
inline void axis_setup(float u, int n, int& i0, int& i1, float& a) {
  float f = u * (float)n - 0.5f;
  float fl = floorf(f);
  a = f - fl;
  int i = (int)fminf(fmaxf(fl, -1.0f), (float)n);   // NaN -> -1, keeps the cast defined
  i0 = clampi(i, 0, n - 1);
  i1 = clampi(i + 1, 0, n - 1);
}
// 3-D LINEAR + CLAMP_TO_EDGE, nc interleaved channels, x fastest (calibration_volume.hpp:57-59)
void tex3d(const float* t, int nc, const uint32_t* res, float u, float v, float w, float* out) {
  int x0, x1, y0, y1, z0, z1; float ax, ay, az;
  int nx = (int)res[0], ny = (int)res[1], nz = (int)res[2];
  axis_setup(u, nx, x0, x1, ax); axis_setup(v, ny, y0, y1, ay); axis_setup(w, nz, z0, z1, az);
  auto T = [&](int x, int y, int z, int c) { return t[((size_t)((size_t)z * ny + y) * nx + x) * nc + c]; };
  for (int c = 0; c < nc; ++c) {
    float c00 = lerp(T(x0, y0, z0, c), T(x1, y0, z0, c), ax);
    float c10 = lerp(T(x0, y1, z0, c), T(x1, y1, z0, c), ax);
    float c01 = lerp(T(x0, y0, z1, c), T(x1, y0, z1, c), ax);
    float c11 = lerp(T(x0, y1, z1, c), T(x1, y1, z1, c), ax);
    out[c] = lerp(lerp(c00, c10, ay), lerp(c01, c11, ay), az);
  }
}
// 2-D array LINEAR + CLAMP_TO_EDGE; layer is already the integer clamp(floor(l + .5))
void tex2d_linear(const float* t, int nc, int w, int h, int layer, float u, float v, float* out) {
  int x0, x1, y0, y1; float ax, ay;
  axis_setup(u, w, x0, x1, ax); axis_setup(v, h, y0, y1, ay);
  const float* b = t + (size_t)layer * w * h * nc;
  for (int c = 0; c < nc; ++c) {
    float r0 = lerp(b[((size_t)y0 * w + x0) * nc + c], b[((size_t)y0 * w + x1) * nc + c], ax);
    float r1 = lerp(b[((size_t)y1 * w + x0) * nc + c], b[((size_t)y1 * w + x1) * nc + c], ax);
    out[c] = lerp(r0, r1, ay);
  }
}
// RGB8 unorm bilinear (NetKinectArray.cpp:147-157 colour array): texels converted to float first
void tex2d_linear_u8(const uint8_t* t, int w, int h, int layer, float u, float v, float* out3) {
  int x0, x1, y0, y1; float ax, ay;
  axis_setup(u, w, x0, x1, ax); axis_setup(v, h, y0, y1, ay);
  const uint8_t* b = t + (size_t)layer * w * h * 3;
  for (int c = 0; c < 3; ++c) {
    float t00 = b[((size_t)y0 * w + x0) * 3 + c] / 255.0f, t10 = b[((size_t)y0 * w + x1) * 3 + c] / 255.0f;
    float t01 = b[((size_t)y1 * w + x0) * 3 + c] / 255.0f, t11 = b[((size_t)y1 * w + x1) * 3 + c] / 255.0f;
    out3[c] = lerp(lerp(t00, t10, ax), lerp(t01, t11, ax), ay);
  }
}
inline int nearest_idx(float u, int n) {
  float f = floorf(u * (float)n);
  return clampi((int)fminf(fmaxf(f, -1.0f), (float)n), 0, n - 1);
}
// 2-D array NEAREST (depth RG32F, NetKinectArray.cpp:181-184), returns channel c
inline float tex2d_nearest(const float* t, int nc, int w, int h, int layer, float u, float v, int c) {
  int x = nearest_idx(u, w), y = nearest_idx(v, h);
  return t[((size_t)layer * w * h + (size_t)y * w + x) * nc + c];
}

}  // namespace

// ================================================================= C API
extern "C" {

struct orc_config {
  float bbox_min[3], bbox_max[3];
  float voxel_size;        // used when res[0] == 0: res = ceil(bbox / voxel_size)   (recon_integration.cpp:340-344)
  uint32_t res[3];         // explicit volume resolution (benchmark configs)
  float brick_size[3];     // world units; reference has one scalar (recon_integration.cpp:53,462-472)
  float limit;
  uint32_t num_streams;
  uint32_t depth_w, depth_h, color_w, color_h;
  uint32_t view_w, view_h;
};

struct orc_brick { uint32_t lo[3], hi[3]; };   // voxel ranges == VolumeSampler::containedVoxels lists

struct orc_ctx {
  orc_config cfg;
  uint32_t res[3];
  vec3 bmin, bsize;                 // bbox min, extent
  vec3 brick;                       // brick size (world)
  uint32_t res_bricks[3];
  std::vector<orc_brick> bricks;
  std::vector<uint32_t> counters;   // Bricks SSBO payload (inc_bricks.glsl:10-16)
  std::vector<uint32_t> occupied;   // Occupied SSBO (inc_bricks.glsl:18-20)
  uint32_t min_voxels = 10;         // recon_integration.cpp:59
  float ratio_occupied = 0;
  bool use_bricks = true, skip_space = true, fill_holes = true;
  float limit;
  std::vector<float> tsdf;
  // borrowed inputs
  const float* xyz_inv[16] = {}; uint32_t xyz_inv_res[16][3];
  const float* uv[16] = {};      uint32_t uv_res[16][3];
  const float* xyz[16] = {};     uint32_t xyz_res[16][3];
  const float *depth = nullptr, *quality = nullptr, *silhouette = nullptr;
  const uint8_t* color = nullptr;
  // pre-processing (NetKinectArray::processTextures, framework/NetKinectArray.cpp:309-426): raw inputs, flags, products
  const float* raw_depth = nullptr;                    // [N][H][W] metres (m_depthArray_raw, R32F, NEAREST)
  float cv_min_d[16], cv_max_d[16];                    // CalibVolumes::getDepthLimits(i)
  vec3 cam_pos[16];                                    // CalibVolumes::getCameraPositions()
  bool compress[16] = {};                              // KinectCalibrationFile::isCompressedDepth(), NetKinectArray.cpp:343
  float dc_near[16] = {}, dc_scale[16] = {}, dc_scaled_near[16] = {};   // uniforms near / scale / scaled_near, :344-349
  bool filter_textures = true, use_processed_depth = true, refine_bound = true;   // NetKinectArray.cpp:63-69
  std::vector<float> pp_depth2, pp_depth_rg, pp_lab, pp_depth_b, pp_sil, pp_normal, pp_quality;
  // view state
  uint32_t vw, vh, aw;                        // view size, atlas width (1.5 w)
  std::vector<uint32_t> lod_off, lod_res;     // ViewLod tables, 2 uints per level
  std::vector<float> atlasA_c, atlasA_d, atlasB_c, atlasB_d;   // RGBA32F + DEPTH32F atlases
  bool target_is_A = true;                    // m_view_inpaint points at A
  std::vector<float> peels;                   // m_view_depth RGBA32F
  std::vector<float> nsamples;                // tex_num_samples
  std::vector<float> fb_c, fb_d;              // "default framebuffer" colour / depth
  float min_length = 0.0125f;                 // KinectCalibrationFile::min_length default (KinectCalibrationFile.cpp:96) -> Reconstruction::m_min_length
  const float* normals = nullptr;             // [N][H][W][3] kinect_normals (NetKinectArray "normal" array) for the point back-end
  int shade_mode = 0;
  // stereo modes (source/kinect_client.cpp:616-669): glViewport origin + viewport_offset uniform (side by side,
  // tsdf_raymarch.fs:70,388-389); m_color_mask_mode (reconstruction.cpp:51-53) + whether the client cleared the colour buffer
  int vp_org[2] = {0, 0};
  float vp_off[2] = {0.0f, 0.0f};
  uint32_t color_mask_mode = 0;
  bool keep_color = false;
};

// glColorMask(GL_TRUE, GL_FALSE, GL_FALSE, GL_FALSE) / (GL_FALSE, GL_TRUE, GL_TRUE, GL_FALSE), recon_integration.cpp:212-216,321-326
static inline void masked_store(const orc_ctx* c, float* dst, const float* src) {
  if (c->color_mask_mode == 1) { dst[0] = src[0]; return; }
  if (c->color_mask_mode == 2) { dst[1] = src[1]; dst[2] = src[2]; return; }
  memcpy(dst, src, 16);
}

static void set_view(orc_ctx* c, uint32_t w, uint32_t h) {
  // ViewLod::setResolution, view_lod.cpp:24-50
  c->vw = w; c->vh = h;
  uint32_t num_lods = 1 + (uint32_t)floorf(log2f((float)std::min(w, h)));
  c->aw = (uint32_t)(w * 1.5f);
  c->lod_off.assign(num_lods * 2, 0); c->lod_res.assign(num_lods * 2, 0);
  uint32_t ox = w, oy = h;
  for (uint32_t i = 0; i < num_lods; ++i) {
    uint32_t rx = (uint32_t)floorf(w / powf(2.0f, (float)i)), ry = (uint32_t)floorf(h / powf(2.0f, (float)i));
    c->lod_res[2 * i] = rx; c->lod_res[2 * i + 1] = ry;
    if (i > 0) { oy -= ry; c->lod_off[2 * i] = ox; c->lod_off[2 * i + 1] = oy; }
  }
  size_t n = (size_t)c->aw * h;
  c->atlasA_c.assign(n * 4, 0.5f); c->atlasA_d.assign(n, 0.5f);   // view_lod.cpp:31-35 initial fill
  c->atlasB_c.assign(n * 4, 0.5f); c->atlasB_d.assign(n, 0.5f);
  c->peels.assign((size_t)w * h * 4, 0.0f);
  c->nsamples.assign((size_t)w * h, 0.0f);
  c->fb_c.assign((size_t)w * h * 4, 0.0f); c->fb_d.assign((size_t)w * h, 1.0f);
}

// divideBox(), recon_integration.cpp:360-406 + VolumeSampler::containedVoxels, volume_sampler.cpp:50-62
static void divide_box(orc_ctx* c) {
  c->bricks.clear();
  vec3 mn = c->bmin, size = c->bsize, start = mn;
  uint32_t rb[3] = {0, 0, 0};
  vec3 step = {1.0f / (float)c->res[0], 1.0f / (float)c->res[1], 1.0f / (float)c->res[2]};
  while (size.z - start.z + mn.z > 0.0f) {
    while (size.y - start.y + mn.y > 0.0f) {
      while (size.x - start.x + mn.x > 0.0f) {
        vec3 rem = size - start + mn;
        vec3 bs = {fminf(c->brick.x, rem.x), fminf(c->brick.y, rem.y), fminf(c->brick.z, rem.z)};
        vec3 pos = {(start.x - mn.x) / size.x, (start.y - mn.y) / size.y, (start.z - mn.z) / size.z};
        vec3 sz = {bs.x / size.x, bs.y / size.y, bs.z / size.z};
        orc_brick b;
        const float p[3] = {pos.x, pos.y, pos.z}, s[3] = {sz.x, sz.y, sz.z}, st[3] = {step.x, step.y, step.z};
        for (int a = 0; a < 3; ++a) {
          unsigned v = (unsigned)(p[a] / st[a]);
          b.lo[a] = v;
          while ((float)v < (p[a] + s[a]) / st[a]) ++v;
          b.hi[a] = std::min(v, c->res[a]);   // indices past the volume would be out-of-range vertices
        }
        c->bricks.push_back(b);
        start.x += c->brick.x;
        if (rb[2] == 0 && rb[1] == 0) ++rb[0];
      }
      start.x = mn.x;
      start.y += c->brick.y;
      if (rb[2] == 0) ++rb[1];
    }
    start.y = mn.y;
    start.z += c->brick.z;
    ++rb[2];
  }
  memcpy(c->res_bricks, rb, sizeof(rb));
  c->counters.assign(c->bricks.size(), 0);
  c->occupied.clear();
}

orc_ctx* orc_create(const orc_config* cfg) {
  orc_ctx* c = new orc_ctx();
  c->cfg = *cfg;
  c->limit = cfg->limit;
  c->bmin = {cfg->bbox_min[0], cfg->bbox_min[1], cfg->bbox_min[2]};
  c->bsize = {cfg->bbox_max[0] - cfg->bbox_min[0], cfg->bbox_max[1] - cfg->bbox_min[1], cfg->bbox_max[2] - cfg->bbox_min[2]};
  const float ext[3] = {c->bsize.x, c->bsize.y, c->bsize.z};
  float vox[3];
  for (int a = 0; a < 3; ++a) {
    if (cfg->res[0] == 0) { c->res[a] = (uint32_t)ceilf(ext[a] / cfg->voxel_size); vox[a] = cfg->voxel_size; }   // setVoxelSize :340-344
    else { c->res[a] = cfg->res[a]; vox[a] = ext[a] / (float)cfg->res[a]; }
  }
  // setBrickSize :463  m_brick_size = voxel * round(size / voxel)
  c->brick = {vox[0] * roundf(cfg->brick_size[0] / vox[0]), vox[1] * roundf(cfg->brick_size[1] / vox[1]), vox[2] * roundf(cfg->brick_size[2] / vox[2])};
  c->tsdf.assign((size_t)c->res[0] * c->res[1] * c->res[2], 0.0f);
  divide_box(c);
  set_view(c, cfg->view_w, cfg->view_h);
  return c;
}
void orc_destroy(orc_ctx* c) { delete c; }
// setVoxelSize(), recon_integration.cpp:340-353: a new resolution, a new (uninitialised -> zero here) volume, and
// setBrickSize(m_brick_size) with the brick size that is CURRENT, i.e. already snapped to the old voxels (:351, :462-464)
void orc_set_voxel_size(orc_ctx* c, float size) {
  const float ext[3] = {c->bsize.x, c->bsize.y, c->bsize.z};
  for (int a = 0; a < 3; ++a) c->res[a] = (uint32_t)ceilf(ext[a] / size);
  c->brick = {size * roundf(c->brick.x / size), size * roundf(c->brick.y / size), size * roundf(c->brick.z / size)};
  c->tsdf.assign((size_t)c->res[0] * c->res[1] * c->res[2], 0.0f);
  divide_box(c);
}

void orc_get_layout(orc_ctx* c, uint32_t* res3, uint32_t* res_bricks3, float* brick3, uint32_t* num_lods) {
  memcpy(res3, c->res, 12); memcpy(res_bricks3, c->res_bricks, 12);
  brick3[0] = c->brick.x; brick3[1] = c->brick.y; brick3[2] = c->brick.z;
  *num_lods = (uint32_t)(c->lod_res.size() / 2);
}
void orc_get_lod_tables(orc_ctx* c, uint32_t* off, uint32_t* res) {
  memcpy(off, c->lod_off.data(), c->lod_off.size() * 4); memcpy(res, c->lod_res.data(), c->lod_res.size() * 4);
}
void orc_get_brick_ranges(orc_ctx* c, uint32_t* out /*6 per brick*/) { memcpy(out, c->bricks.data(), c->bricks.size() * sizeof(orc_brick)); }

void orc_set_calibration(orc_ctx* c, uint32_t i, const float* xyz_inv, const uint32_t* r_inv, const float* uv, const uint32_t* r_uv,
                         const float* xyz, const uint32_t* r_xyz) {
  c->xyz_inv[i] = xyz_inv; memcpy(c->xyz_inv_res[i], r_inv, 12);
  c->uv[i] = uv; if (uv) memcpy(c->uv_res[i], r_uv, 12);
  c->xyz[i] = xyz; if (xyz) memcpy(c->xyz_res[i], r_xyz, 12);
}
void orc_set_frame(orc_ctx* c, const float* depth_rg, const float* quality, const float* silhouette, const uint8_t* color_rgb) {
  c->depth = depth_rg; c->quality = quality; c->silhouette = silhouette; c->color = color_rgb;
}
void orc_set_flags(orc_ctx* c, int use_bricks, int skip_space, int fill_holes, uint32_t min_voxels, int shade_mode) {
  c->use_bricks = use_bricks; c->skip_space = skip_space; c->fill_holes = fill_holes; c->min_voxels = min_voxels; c->shade_mode = shade_mode;
}
void orc_set_limit(orc_ctx* c, float limit) { c->limit = limit; }
void orc_set_stereo(orc_ctx* c, int org_x, int org_y, float off_x, float off_y, uint32_t color_mask_mode, int clear_color) {
  c->vp_org[0] = org_x; c->vp_org[1] = org_y; c->vp_off[0] = off_x; c->vp_off[1] = off_y;
  c->color_mask_mode = color_mask_mode; c->keep_color = clear_color == 0;
}

// ---------------------------------------------------------------- bricks (inc_bricks.glsl)
void orc_clear_occupied(orc_ctx* c) { std::fill(c->counters.begin(), c->counters.end(), 0u); }   // recon_integration.cpp:271-277

// mark_brick(), inc_bricks.glsl:40-58.  Positions whose own brick index falls outside the grid are
// undefined behaviour in the reference (negative float -> uvec3); they are skipped here.
static void mark_brick(orc_ctx* c, vec3 pos) {
  vec3 rel = pos - c->bmin;
  float fx = floorf(rel.x / c->brick.x), fy = floorf(rel.y / c->brick.y), fz = floorf(rel.z / c->brick.z);
  const int rx = (int)c->res_bricks[0], ry = (int)c->res_bricks[1], rz = (int)c->res_bricks[2];
  if (!(fx >= 0 && fy >= 0 && fz >= 0 && fx < (float)rx && fy < (float)ry && fz < (float)rz)) return;
  int ix = (int)fx, iy = (int)fy, iz = (int)fz;
  // to_world(vec3(0.5), index) = vec3(index) * brick_size + bbox_min + 0.5 * brick_size   (:22-24)
  vec3 center = {(float)ix * c->brick.x + c->bmin.x + 0.5f * c->brick.x,
                 (float)iy * c->brick.y + c->bmin.y + 0.5f * c->brick.y,
                 (float)iz * c->brick.z + c->bmin.z + 0.5f * c->brick.z};
  vec3 diff = pos - center;
  vec3 da = {fabsf(diff.x), fabsf(diff.y), fabsf(diff.z)};
  float mv = fmaxf(da.x, fmaxf(da.y, da.z));
  float mcx = da.x < mv ? 0.0f : 1.0f, mcy = da.y < mv ? 0.0f : 1.0f, mcz = da.z < mv ? 0.0f : 1.0f;
  auto sgn = [](float v) { return (v > 0.0f) ? 1 : ((v < 0.0f) ? -1 : 0); };
  int ox = sgn(diff.x * mcx), oy = sgn(diff.y * mcy), oz = sgn(diff.z * mcz);
  int nx = clampi(ix + ox, 0, rx - 1), ny = clampi(iy + oy, 0, ry - 1), nz = clampi(iz + oz, 0, rz - 1);
  c->counters[(size_t)nz * ry * rx + (size_t)ny * rx + nx] += (da.x > c->brick.x * 0.1f) ? 1u : 0u;   // :52 (tests d_abs.x, quirk 2)
  c->counters[(size_t)iz * ry * rx + (size_t)iy * rx + ix] += 1u;                                     // :57
}

// pre_normal.fs:22-33 call site: every pixel with 0 < depth < 1 marks the brick of its world position
void orc_mark_bricks(orc_ctx* c) {
  const int w = (int)c->cfg.depth_w, h = (int)c->cfg.depth_h;
  for (uint32_t l = 0; l < c->cfg.num_streams; ++l)
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        float u = ((float)x + 0.5f) / (float)w, v = ((float)y + 0.5f) / (float)h;   // full-screen triangle texcoord
        float d = tex2d_nearest(c->depth, 2, w, h, (int)l, u, v, 0);
        if (d <= 0.0f || d >= 1.0f) continue;
        float wp[3];
        tex3d(c->xyz[l], 3, c->xyz_res[l], u, v, d, wp);
        mark_brick(c, {wp[0], wp[1], wp[2]});
      }
}

// updateOccupiedBricks(), recon_integration.cpp:430-445
float orc_update_occupied(orc_ctx* c) {
  c->occupied.clear();
  for (uint32_t i = 0; i < c->counters.size(); ++i)
    if (c->counters[i] >= c->min_voxels) c->occupied.push_back(i);
  c->ratio_occupied = (float)c->occupied.size() / (float)c->counters.size();
  return c->ratio_occupied;
}
uint32_t orc_num_bricks(orc_ctx* c) { return (uint32_t)c->counters.size(); }
uint32_t orc_num_occupied(orc_ctx* c) { return (uint32_t)c->occupied.size(); }
void orc_get_counters(orc_ctx* c, uint32_t* out) { memcpy(out, c->counters.data(), c->counters.size() * 4); }
void orc_set_counters(orc_ctx* c, const uint32_t* in) { memcpy(c->counters.data(), in, c->counters.size() * 4); }
void orc_get_occupied(orc_ctx* c, uint32_t* out) { memcpy(out, c->occupied.data(), c->occupied.size() * 4); }

// ---------------------------------------------------------------- K1  tsdf_integration.vs:23-59
static inline float integrate_voxel(const orc_ctx* c, float px, float py, float pz) {
  const float limit = c->limit;
  const int w = (int)c->cfg.depth_w, h = (int)c->cfg.depth_h;
  float weighted_tsd = limit, total_weight = 0.0f;
  for (uint32_t i = 0; i < c->cfg.num_streams; ++i) {
    float pc[4];
    tex3d(c->xyz_inv[i], 4, c->xyz_inv_res[i], px, py, pz, pc);
    float sil;
    tex2d_linear(c->silhouette, 1, w, h, (int)i, pc[0], pc[1], &sil);
    if (sil < 1.0f) {
      if (weighted_tsd >= limit) { weighted_tsd = -limit; continue; }
    }
    float depth = tex2d_nearest(c->depth, 2, w, h, (int)i, pc[0], pc[1], 0);
    float sdist = pc[2] - depth;
    if (sdist <= -limit) {
      weighted_tsd = -limit;
    } else if (sdist >= limit) {
    } else {
      float weight;
      tex2d_linear(c->quality, 1, w, h, (int)i, pc[0], pc[1], &weight);
      weighted_tsd = (weighted_tsd * total_weight + weight * sdist) / (total_weight + weight);
      total_weight += weight;
    }
  }
  return weighted_tsd;
}

// integrate(), recon_integration.cpp:242-269
void orc_integrate(orc_ctx* c) {
  const uint32_t rx = c->res[0], ry = c->res[1], rz = c->res[2];
  std::fill(c->tsdf.begin(), c->tsdf.end(), -c->limit);                  // :249-250
  const float sx = 1.0f / (float)rx, sy = 1.0f / (float)ry, sz = 1.0f / (float)rz;   // volume_sampler.cpp:36-38
  auto do_voxel = [&](uint32_t x, uint32_t y, uint32_t z) {
    float px = ((float)x + 0.5f) * sx, py = ((float)y + 0.5f) * sy, pz = ((float)z + 0.5f) * sz;
    float v = integrate_voxel(c, px, py, pz);
    int ix = (int)(px * (float)rx), iy = (int)(py * (float)ry), iz = (int)(pz * (float)rz);   // :57 ivec3(position * res)
    c->tsdf[((size_t)iz * ry + iy) * rx + ix] = v;
  };
  if (c->use_bricks) {
#pragma omp parallel for schedule(dynamic, 8)
    for (long k = 0; k < (long)c->occupied.size(); ++k) {
      const orc_brick& b = c->bricks[c->occupied[k]];
      for (uint32_t y = b.lo[1]; y < b.hi[1]; ++y)
        for (uint32_t x = b.lo[0]; x < b.hi[0]; ++x)
          for (uint32_t z = b.lo[2]; z < b.hi[2]; ++z) do_voxel(x, y, z);
    }
  } else {
#pragma omp parallel for schedule(static)
    for (long z = 0; z < (long)rz; ++z)
      for (uint32_t y = 0; y < ry; ++y)
        for (uint32_t x = 0; x < rx; ++x) do_voxel(x, y, (uint32_t)z);
  }
}
const float* orc_tsdf(orc_ctx* c) { return c->tsdf.data(); }
void orc_set_tsdf(orc_ctx* c, const float* v) { memcpy(c->tsdf.data(), v, c->tsdf.size() * 4); }

// ---------------------------------------------------------------- view matrices (draw(), :182-205)
struct view_mats {
  mat4 mv, proj, mv_inv, v2w, v2w_inv, img_to_eye, normal, mv_v2w, glnormal_inv;
  vec3 cam_vol, cam_world;
};
static view_mats make_view(const orc_ctx* c, const float* mv16, const float* proj16) {
  view_mats V;
  memcpy(V.mv.m, mv16, 64); memcpy(V.proj.m, proj16, 64);
  double mv[16], pr[16], v2w[16] = {0}, t[16], s[16] = {0}, tr[16] = {0};
  to_d(mv16, mv); to_d(proj16, pr);
  // vol_to_world = translate(bbox_min) * scale(bbox_dim), :66-72
  v2w[0] = c->bsize.x; v2w[5] = c->bsize.y; v2w[10] = c->bsize.z; v2w[15] = 1;
  v2w[12] = c->bmin.x; v2w[13] = c->bmin.y; v2w[14] = c->bmin.z;
  V.v2w = to_f(v2w);
  inverse_d(v2w, t); V.v2w_inv = to_f(t);
  double mvi[16]; inverse_d(mv, mvi); V.mv_inv = to_f(mvi);
  // image_to_eye = inverse(scale(w/2,h/2,.5) * translate(1,1,1) * P), :184-193
  s[0] = c->vw * 0.5; s[5] = c->vh * 0.5; s[10] = 0.5; s[15] = 1;
  tr[0] = tr[5] = tr[10] = tr[15] = 1; tr[12] = tr[13] = tr[14] = 1;
  double a[16], b[16]; mul_d(tr, pr, a); mul_d(s, a, b); inverse_d(b, a); V.img_to_eye = to_f(a);
  // NormalMatrix = inverseTranspose(MV * vol_to_world), :199
  double m[16], mi[16], mit[16]; mul_d(mv, v2w, m); V.mv_v2w = to_f(m); inverse_d(m, mi);
  for (int cc = 0; cc < 4; ++cc) for (int r = 0; r < 4; ++r) mit[cc * 4 + r] = mi[r * 4 + cc];
  V.normal = to_f(mit);
  // gl_NormalMatrix (fixed function) = inverseTranspose(MV); shading.glsl:65 applies its inverse = transpose(MV)
  double mvt[16]; for (int cc = 0; cc < 4; ++cc) for (int r = 0; r < 4; ++r) mvt[cc * 4 + r] = mv[r * 4 + cc];
  V.glnormal_inv = to_f(mvt);
  // CameraPos, :202-205
  vec4 cw = mul(V.mv_inv, {0, 0, 0, 1}); V.cam_world = {cw.x, cw.y, cw.z};
  vec4 cv = mul(V.v2w_inv, cw); V.cam_vol = {cv.x, cv.y, cv.z};
  return V;
}

// screenToVol(), tsdf_raymarch.fs:376-383
static inline vec3 screen_to_vol(const view_mats& V, float fx, float fy, float fz) {
  vec4 p = mul(V.img_to_eye, {fx, fy, fz, 1.0f});
  vec4 es = {p.x / p.w, p.y / p.w, p.z / p.w, 1.0f};
  vec4 ws = mul(V.mv_inv, es);
  vec4 vp = mul(V.v2w_inv, ws);
  return {vp.x, vp.y, vp.z};
}
// Direction of the ray through a pixel centre, in volume space, unnormalised.  Equals
// pass_Position - CameraPos up to scale (tsdf_raymarch.fs:64).
static inline vec3 pixel_dir_vol(const view_mats& V, float fx, float fy) {
  vec4 p = mul(V.img_to_eye, {fx, fy, 1.0f, 1.0f});
  vec4 ed = {p.x / p.w, p.y / p.w, p.z / p.w, 0.0f};       // eye-space direction (camera at the eye origin)
  vec4 wd = mul(V.mv_inv, ed);
  vec4 vd = mul(V.v2w_inv, wd);
  return {vd.x, vd.y, vd.z};
}

// ---------------------------------------------------------------- K5  drawDepthLimits(), :408-428
// bricks.vs/gs/fs: every face of every occupied brick that is not shared with a brick whose counter
// is > 10 (inc_bricks.glsl:60-62; index arithmetic wraps like the shader's uint maths, reads past the
// buffer give 0) is rasterised with MIN blending into (z, -z, frontFacing ? 1 : z); clear (1,0,1,0).
// Rasterisation is restated per pixel centre: intersect the pixel ray with each face rectangle.
static uint32_t neighbour_count(const orc_ctx* c, uint32_t ix, uint32_t iy, uint32_t iz, int axis, int dir) {
  uint32_t i[3] = {ix, iy, iz};
  i[axis] += (uint32_t)dir;                                              // uvec3 + ivec3 wraps
  uint32_t id = i[2] * c->res_bricks[1] * c->res_bricks[0] + i[1] * c->res_bricks[0] + i[0];   // get_id :26-28
  return id < c->counters.size() ? c->counters[id] : 0u;
}
static void depth_limits(orc_ctx* c, const view_mats& V) {
  const uint32_t w = c->vw, h = c->vh;
  const uint32_t rbx = c->res_bricks[0], rby = c->res_bricks[1];
  const float brick[3] = {c->brick.x, c->brick.y, c->brick.z}, bmin[3] = {c->bmin.x, c->bmin.y, c->bmin.z};
  struct face { int axis; float coord; float lo[3], hi[3]; bool outward_pos; };
  std::vector<face> faces;
  for (uint32_t id : c->occupied) {
    uint32_t iz = id / (rbx * rby), rem = id % (rbx * rby), iy = rem / rbx, ix = rem % rbx;   // index_3d :30-38
    const uint32_t idx[3] = {ix, iy, iz};
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {   // to_world(p, index) = index * brick + bbox_min + p * brick, p in {0,1}
      lo[a] = (float)idx[a] * brick[a] + bmin[a] + 0.0f * brick[a];
      hi[a] = (float)idx[a] * brick[a] + bmin[a] + 1.0f * brick[a];
    }
    for (int a = 0; a < 3; ++a)
      for (int d = -1; d <= 1; d += 2) {
        if (neighbour_count(c, ix, iy, iz, a, d) > 10u) continue;        // bricks.gs:26-43
        face f; f.axis = a; f.coord = d < 0 ? lo[a] : hi[a]; f.outward_pos = d > 0;
        memcpy(f.lo, lo, 12); memcpy(f.hi, hi, 12);
        faces.push_back(f);
      }
  }
  // Which faces can a pixel see at all?  A face whose four corners lie in front of the eye projects into the bounding box of its projected
  // corners (the rectangle is convex, the projection of its points is then inside the projected corners' hull); bin the faces into
  // 16x16-pixel screen tiles by that box (two pixels of margin for the rounding of the projection; a face with a corner at or behind the
  // eye plane goes into every tile).  The per-pixel, per-face arithmetic below is unchanged and MIN blending is order independent: the
  // peels are the same bits as with every pixel testing every face -- a rasteriser does not do that either.
  constexpr uint32_t kBin = 16;
  const uint32_t btx = (w + kBin - 1) / kBin, bty = (h + kBin - 1) / kBin;
  std::vector<std::vector<uint32_t>> bins((size_t)btx * bty);
  for (uint32_t fi = 0; fi < faces.size(); ++fi) {
    const face& f = faces[fi];
    const int a = f.axis, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
    float x0 = 1e30f, y0 = 1e30f, x1 = -1e30f, y1 = -1e30f;
    bool all_front = true;
    for (int k = 0; k < 4; ++k) {
      float cp[3]; cp[a] = f.coord; cp[a1] = (k & 1) ? f.hi[a1] : f.lo[a1]; cp[a2] = (k & 2) ? f.hi[a2] : f.lo[a2];
      const vec4 e = mul(V.mv, {cp[0], cp[1], cp[2], 1.0f});
      const vec4 cl = mul(V.proj, e);
      if (!(cl.w > 1e-4f)) { all_front = false; break; }
      const float sx = (cl.x / cl.w * 0.5f + 0.5f) * (float)w, sy = (cl.y / cl.w * 0.5f + 0.5f) * (float)h;
      x0 = fminf(x0, sx); x1 = fmaxf(x1, sx); y0 = fminf(y0, sy); y1 = fmaxf(y1, sy);
    }
    long bx0 = 0, by0 = 0, bx1 = (long)btx - 1, by1 = (long)bty - 1;
    if (all_front && x0 == x0 && x1 == x1 && y0 == y0 && y1 == y1) {
      if (x1 < -2.0f || y1 < -2.0f || x0 > (float)w + 2.0f || y0 > (float)h + 2.0f) continue;          // off screen
      bx0 = std::max(0L, (long)floorf(x0 - 2.0f) / (long)kBin); by0 = std::max(0L, (long)floorf(y0 - 2.0f) / (long)kBin);
      bx1 = std::min((long)btx - 1, (long)floorf(x1 + 2.0f) / (long)kBin); by1 = std::min((long)bty - 1, (long)floorf(y1 + 2.0f) / (long)kBin);
      if (x0 - 2.0f < 0.0f) bx0 = 0;
      if (y0 - 2.0f < 0.0f) by0 = 0;
    }
    for (long by = by0; by <= by1; ++by)
      for (long bx = bx0; bx <= bx1; ++bx) bins[(size_t)by * btx + bx].push_back(fi);
  }
#pragma omp parallel for schedule(dynamic, 4)
  for (long py = 0; py < (long)h; ++py)
    for (uint32_t px = 0; px < w; ++px) {
      float r = 1.0f, g = 0.0f, b = 1.0f;                                // clear colour :144
      vec4 p = mul(V.img_to_eye, {(float)px + 0.5f, (float)py + 0.5f, 1.0f, 1.0f});
      vec4 ed = {p.x / p.w, p.y / p.w, p.z / p.w, 0.0f};
      vec4 wd = mul(V.mv_inv, ed);
      const float o[3] = {V.cam_world.x, V.cam_world.y, V.cam_world.z}, d[3] = {wd.x, wd.y, wd.z};
      for (const uint32_t fi : bins[(size_t)(py / kBin) * btx + px / kBin]) {
        const face& f = faces[fi];
        const int a = f.axis, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
        float t = (f.coord - o[a]) / d[a];
        if (!(t > 0.0f)) continue;
        float q1 = o[a1] + t * d[a1], q2 = o[a2] + t * d[a2];
        if (q1 < f.lo[a1] || q1 > f.hi[a1] || q2 < f.lo[a2] || q2 > f.hi[a2]) continue;
        float hp[3]; hp[a] = f.coord; hp[a1] = q1; hp[a2] = q2;
        vec4 e = mul(V.mv, {hp[0], hp[1], hp[2], 1.0f});
        float zw = (V.proj.m[10] * e.z + V.proj.m[14]) / (-e.z) * 0.5f + 0.5f;
        if (!(zw >= 0.0f && zw <= 1.0f)) continue;                       // near/far clip
        bool front = f.outward_pos ? (o[a] > f.coord) : (o[a] < f.coord);
        r = fminf(r, zw); g = fminf(g, -zw); b = fminf(b, front ? 1.0f : zw);
      }
      float* o4 = &c->peels[((size_t)py * w + px) * 4];
      o4[0] = r; o4[1] = g; o4[2] = b; o4[3] = 0.0f;   // min(clear 0, fragment 1)
    }
}

// ---------------------------------------------------------------- K2  tsdf_raymarch.fs
static const float camera_colors[8][3] = {   // shading.glsl:24-30, extended past 5 streams
    {228 / 255.0f, 26 / 255.0f, 28 / 255.0f}, {55 / 255.0f, 126 / 255.0f, 184 / 255.0f}, {77 / 255.0f, 175 / 255.0f, 74 / 255.0f},
    {152 / 255.0f, 78 / 255.0f, 163 / 255.0f}, {255 / 255.0f, 127 / 255.0f, 0 / 255.0f}, {255 / 255.0f, 255 / 255.0f, 51 / 255.0f},
    {166 / 255.0f, 86 / 255.0f, 40 / 255.0f}, {247 / 255.0f, 129 / 255.0f, 191 / 255.0f}};

static inline float sample_tsdf(const orc_ctx* c, vec3 p) {   // :136-138
  float v; tex3d(c->tsdf.data(), 1, c->res, p.x, p.y, p.z, &v); return v;
}
static vec3 get_gradient(const orc_ctx* c, vec3 pos, float sd) {   // :140-149
  vec3 g = {sample_tsdf(c, pos + vec3{sd, 0, 0}) - sample_tsdf(c, pos - vec3{sd, 0, 0}),
            sample_tsdf(c, pos + vec3{0, sd, 0}) - sample_tsdf(c, pos - vec3{0, sd, 0}),
            sample_tsdf(c, pos + vec3{0, 0, sd}) - sample_tsdf(c, pos - vec3{0, 0, sd})};
  vec3 n = normalize(g);
  return {-n.x, -n.y, -n.z};
}
static vec4 blend_colors(const orc_ctx* c, vec3 sp) {   // :295-330
  const int w = (int)c->cfg.depth_w, h = (int)c->cfg.depth_h, cw = (int)c->cfg.color_w, ch = (int)c->cfg.color_h;
  vec3 tc = {0, 0, 0}, tc2 = {0, 0, 0};
  float tw = 0.0f, tw2 = 0.0f;
  for (uint32_t i = 0; i < c->cfg.num_streams; ++i) {
    float pc[4], pcol[2], col[3];
    tex3d(c->xyz_inv[i], 4, c->xyz_inv_res[i], sp.x, sp.y, sp.z, pc);
    tex3d(c->uv[i], 2, c->uv_res[i], pc[0], pc[1], pc[2], pcol);
    tex2d_linear_u8(c->color, cw, ch, (int)i, pcol[0], pcol[1], col);
    float depth = tex2d_nearest(c->depth, 2, w, h, (int)i, pc[0], pc[1], 0);
    float quality = 0.0f;
    float dist = fabsf(depth - pc[2]);
    if (dist < c->limit) tex2d_linear(c->quality, 1, w, h, (int)i, pc[0], pc[1], &quality);
    vec3 cv = {col[0], col[1], col[2]};
    tc = tc + cv * quality / (dist + 0.01f);
    tw += quality / (dist + 0.01f);
    tc2 = tc2 + cv / dist;
    tw2 += 1.0f / dist;
  }
  if (tw > 0.0f) { tc = tc / tw; return {tc.x, tc.y, tc.z, 1.0f}; }
  tc2 = tc2 / tw2;
  return {tc2.x, tc2.y, tc2.z, -1.0f};
}
static vec3 blend_cameras(const orc_ctx* c, vec3 sp) {   // :346-361 with getWeights :151-166
  const int w = (int)c->cfg.depth_w, h = (int)c->cfg.depth_h;
  vec3 tc = {0, 0, 0}; float tw = 0.0f;
  for (uint32_t i = 0; i < c->cfg.num_streams; ++i) {
    float pc[4];
    tex3d(c->xyz_inv[i], 4, c->xyz_inv_res[i], sp.x, sp.y, sp.z, pc);
    float depth = tex2d_nearest(c->depth, 2, w, h, (int)i, pc[0], pc[1], 0);
    float q = 0.0f;
    if (fabsf(depth - pc[2]) < c->limit) tex2d_linear(c->quality, 1, w, h, (int)i, pc[0], pc[1], &q);
    const float* cc = camera_colors[i % 8];
    tc = tc + vec3{cc[0], cc[1], cc[2]} * q; tw += q;
  }
  tc = tc / tw;
  if (tw <= 0.0f) tc = {1.0f, 1.0f, 1.0f};
  return tc;
}
// shading.glsl:32-69
static vec3 shade(const orc_ctx* c, const view_mats& V, vec3 view_pos, vec3 view_normal, vec3 diffuse) {
  if (c->shade_mode == 0) return diffuse;
  if (c->shade_mode == 1) {
    const vec3 LightPosition = {1.5f, 1.0f, 1.0f}, LightDiffuse = {1.0f, 0.9f, 0.7f};
    const vec3 LightAmbient = LightDiffuse * 0.2f, LightSpecular = {1.0f, 1.0f, 1.0f}, solid = {0.5f, 0.5f, 0.5f};
    const float ks = 0.5f, n = 20.0f;
    float diff = 0.0f, spec = 0.0f;
    vec3 toLight = normalize(LightPosition - view_pos);
    float lightAngle = dot(view_normal, toLight);
    if (!(lightAngle <= 0.0f)) {
      diff = fmaxf(lightAngle, 0.0f);
      vec3 toViewer = normalize(vec3{-view_pos.x, -view_pos.y, -view_pos.z});
      vec3 hv = normalize(toLight + toViewer);
      spec = powf(dot(hv, view_normal), n);
      float a = (1.0f - lightAngle) * (1.0f - lightAngle);
      spec *= 1.0f - a * a * a;
    }
    return LightAmbient * solid + LightDiffuse * solid * diff + LightSpecular * ks * spec;
  }
  if (c->shade_mode == 2) {
    vec4 r = mul(V.glnormal_inv, {view_normal.x, view_normal.y, view_normal.z, 0.0f});
    return {r.x, r.y, r.z};
  }
  return {1.0f, 1.0f, 1.0f};
}

// draw(), :176-240 (+ drawF :151-158): clears, depth limits, raymarch into the hole-filling atlas level 0
// or the framebuffer.
void orc_draw(orc_ctx* c, const float* mv16, const float* proj16) {
  const view_mats V = make_view(c, mv16, proj16);
  const uint32_t w = c->vw, h = c->vh;
  const bool skip = c->skip_space && c->use_bricks;                      // :154, :510-513
  if (skip) depth_limits(c, V);
  std::fill(c->nsamples.begin(), c->nsamples.end(), 0.0f);               // :207-208
  float* tgt_c; float* tgt_d; uint32_t stride;
  if (c->fill_holes) {                                                   // m_view_inpaint->enable(): whole atlas cleared, view_lod.cpp:75-81
    auto& ac = c->target_is_A ? c->atlasA_c : c->atlasB_c; auto& ad = c->target_is_A ? c->atlasA_d : c->atlasB_d;
    for (size_t i = 0; i < ad.size(); ++i) { ac[4 * i] = 0; ac[4 * i + 1] = 1; ac[4 * i + 2] = 0; ac[4 * i + 3] = 0; ad[i] = 1.0f; }
    tgt_c = ac.data(); tgt_d = ad.data(); stride = c->aw;
  } else {
    // the client's glClear before the draw: colour + depth (kinect_client.cpp:609-610,620), or depth alone before the anaglyph's second eye (:627)
    for (size_t i = 0; i < c->fb_d.size(); ++i) { if (!c->keep_color) { c->fb_c[4 * i] = 0; c->fb_c[4 * i + 1] = 0; c->fb_c[4 * i + 2] = 0; c->fb_c[4 * i + 3] = 0; } c->fb_d[i] = 1.0f; }
    tgt_c = c->fb_c.data(); tgt_d = c->fb_d.data(); stride = w;
  }
  const float limit = c->limit, sd = limit * 0.5f;                       // sampleDistance :34
  // rasterised into the window viewport (its origin is part of gl_FragCoord) or, with hole filling, into the pyramid's level-0
  // viewport at (0, 0) (ViewLod::enable, view_lod.cpp:68)
  const int org[2] = {c->fill_holes ? 0 : c->vp_org[0], c->fill_holes ? 0 : c->vp_org[1]};
#pragma omp parallel for schedule(dynamic, 2)
  for (long py = 0; py < (long)h; ++py)
    for (uint32_t px = 0; px < w; ++px) {
      const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
      vec3 step = normalize(pixel_dir_vol(V, fx, fy)) * sd;              // :64
      uint32_t max_n = 0; vec3 pos = {0, 0, 0};
      if (skip) {                                                        // getStartPos(ivec2(gl_FragCoord.xy - viewport_offset)) :70, :384-393
        const float qx = ((float)((long)px + org[0]) + 0.5f) - c->vp_off[0], qy = ((float)(py + org[1]) + 0.5f) - c->vp_off[1];   // gl_FragCoord - viewport_offset
        const int cx = (int)qx, cy = (int)qy;                            // ivec2(): truncation
        static const float zero4[4] = {0, 0, 0, 0};                     // texelFetch out of range -> 0 (Appendix A)
        const float* dm = (cx >= 0 && cy >= 0 && cx < (int)w && cy < (int)h) ? &c->peels[((size_t)cy * w + cx) * 4] : zero4;
        float r = dm[0], g = dm[1], b = dm[2];
        r = (r >= b) ? 0.0f : r;
        vec3 pf = screen_to_vol(V, qx, qy, r), pb = screen_to_vol(V, qx, qy, -g);
        if (r >= 1.0f) pb = pf;
        pos = pf;
        max_n = (uint32_t)ceilf(length(pf - pb) / sd);                   // :73
      } else {                                                           // intersectBox :363-374
        const vec3 o = V.cam_vol;
        vec3 inv = {1.0f / step.x, 1.0f / step.y, 1.0f / step.z};
        vec3 tbot = inv * (vec3{0, 0, 0} - o), ttop = inv * (vec3{1, 1, 1} - o);
        vec3 tmn = {fminf(ttop.x, tbot.x), fminf(ttop.y, tbot.y), fminf(ttop.z, tbot.z)};
        vec3 tmx = {fmaxf(ttop.x, tbot.x), fmaxf(ttop.y, tbot.y), fmaxf(ttop.z, tbot.z)};
        float t0 = fmaxf(fmaxf(tmn.x, tmn.y), fmaxf(tmn.x, tmn.z));
        float t1 = fminf(fminf(tmx.x, tmx.y), fminf(tmx.x, tmx.z));
        if (!(t0 <= t1) || t1 < 0.0f) continue;                          // pixel not covered by the cube: no fragment
        float t_near = t0 < 0.0f ? 0.0f : t0;
        pos = o + step * t_near;
        max_n = (uint32_t)ceilf(fabsf(t1 - t_near));
      }
      float prev = -limit;
      uint32_t n = 0; bool hit = false;
      while (n < max_n) {                                                // :92-110
        n += 1;
        float density = sample_tsdf(c, pos);
        if (density > 0.0f) {
          pos = (pos - step) - step * (prev / (density - prev));
          hit = true; break;
        }
        prev = density;
        pos = pos + step;
      }
      {                                                                  // :395-398: imageStore at ivec2(gl_FragCoord.xy) = viewport origin + pixel
        const long sx = (long)px + org[0], sy = py + org[1];
        if (sx >= 0 && sy >= 0 && sx < (long)w && sy < (long)h) c->nsamples[(size_t)sy * w + sx] = (float)n * 0.0027f;
      }
      if (!hit) continue;                                                // discard
      // submitFragment :116-134
      vec3 gn = get_gradient(c, pos, sd);
      vec4 vn4 = mul(V.normal, {gn.x, gn.y, gn.z, 0.0f});
      vec3 view_normal = normalize(vec3{vn4.x, vn4.y, vn4.z});
      vec4 vp4 = mul(V.mv_v2w, {pos.x, pos.y, pos.z, 1.0f});
      vec3 view_pos = {vp4.x, vp4.y, vp4.z};
      vec4 out;
      if (c->shade_mode == 3) { vec3 bc = blend_cameras(c, pos); out = {bc.x, bc.y, bc.z, 1.0f}; }
      else { vec4 dc = blend_colors(c, pos); vec3 s = shade(c, V, view_pos, view_normal, {dc.x, dc.y, dc.z}); out = {s.x, s.y, s.z, dc.w}; }
      float fd = (V.proj.m[10] * view_pos.z + V.proj.m[14]) / -view_pos.z * 0.5f + 0.5f;   // :133
      size_t o = (size_t)py * stride + px;
      // GL depth test LESS against the cleared 1.0 and the [0,1] depth clamp
      fd = clampf(fd, 0.0f, 1.0f);
      if (!(fd < tgt_d[o])) continue;
      const float o4[4] = {out.x, out.y, out.z, out.w};
      if (c->fill_holes) memcpy(&tgt_c[4 * o], o4, 16);                  // into the pyramid: no mask (:212-216 only without hole filling)
      else masked_store(c, &tgt_c[4 * o], o4);
      tgt_d[o] = fd;
    }
}

// ---------------------------------------------------------------- K5', K3, K4  fillColors(), :279-338
static inline void fetch(const std::vector<float>& col, const std::vector<float>& dep, uint32_t aw, uint32_t ah, int x, int y, float* c4, float* d) {
  if (x < 0 || y < 0 || x >= (int)aw || y >= (int)ah) { c4[0] = c4[1] = c4[2] = c4[3] = 0; *d = 0; return; }   // Appendix A: out-of-range texelFetch -> 0
  size_t o = (size_t)y * aw + x; memcpy(c4, &col[4 * o], 16); *d = dep[o];
}
// framebuffer_transfer.fs:13-17 into viewport (0,0,w,h) of dst after clearing dst (view_lod.cpp:75-81)
static void transfer(orc_ctx* c, const std::vector<float>& sc, const std::vector<float>& sdp, std::vector<float>& dc, std::vector<float>& dd) {
  const uint32_t w = c->vw, h = c->vh, aw = c->aw;
  for (size_t i = 0; i < dd.size(); ++i) { dc[4 * i] = 0; dc[4 * i + 1] = 1; dc[4 * i + 2] = 0; dc[4 * i + 3] = 0; dd[i] = 1.0f; }
  for (uint32_t y = 0; y < h; ++y)
    for (uint32_t x = 0; x < w; ++x) {
      float tu = ((float)x + 0.5f) / (float)w, tv = ((float)y + 0.5f) / (float)h;   // pass_TexCoord, screen_quad.cpp:11-15
      int sx = (int)(tu * (float)aw), sy = (int)(tv * (float)h);                    // ivec2(pass_TexCoord * resolution_tex)
      float c4[4], d; fetch(sc, sdp, aw, h, sx, sy, c4, &d);
      size_t o = (size_t)y * aw + x; memcpy(&dc[4 * o], c4, 16); dd[o] = d;
    }
}
// tsdf_inpaint.fs:34-89, writes level `lod + 1` of dst reading the squeezed copy src
static void inpaint_level(orc_ctx* c, int lod, const std::vector<float>& sc, const std::vector<float>& sdp, std::vector<float>& dc, std::vector<float>& dd) {
  const uint32_t aw = c->aw, h = c->vh;
  const uint32_t *off = c->lod_off.data(), *res = c->lod_res.data();
  const uint32_t ox = off[2 * (lod + 1)], oy = off[2 * (lod + 1) + 1], rx = res[2 * (lod + 1)], ry = res[2 * (lod + 1) + 1];
  for (uint32_t fy = oy; fy < oy + ry; ++fy)
    for (uint32_t fx = ox; fx < ox + rx; ++fx) {
      float tcx = ((float)fx - (float)ox) / (float)rx, tcy = ((float)fy - (float)oy) / (float)ry;      // :37
      int lx = (int)((float)off[2 * lod] + (float)res[2 * lod] * tcx), ly = (int)((float)off[2 * lod + 1] + (float)res[2 * lod + 1] * tcy);   // to_lod_pos :30-32
      int pxi = (int)((float)lx * (float)(2.0 / 3.0)), pyi = (int)((float)ly * 1.0f);                  // :38
      float depth_av = 0.0f; int num = 0; float smp[16][4];
      for (int x = 0; x < 4; ++x)
        for (int y = 0; y < 4; ++y) {
          float c4[4], d; fetch(sc, sdp, aw, h, pxi + x - 1, pyi + y - 1, c4, &d);                     // :45-47
          if (c4[3] <= 0.0f) c4[0] = -1.0f; else { depth_av += d; ++num; }
          float* s = smp[x + y * 4]; s[0] = c4[0]; s[1] = c4[1]; s[2] = c4[2]; s[3] = d;
        }
      size_t o = (size_t)fy * aw + fx;
      if (num == 0) {                                                                                   // :59-68
        float c4[4], d; fetch(sc, sdp, aw, h, pxi, pyi, c4, &d);
        dd[o] = d;
        if (d < 1.0f) { dc[4 * o] = 0; dc[4 * o + 1] = 0; dc[4 * o + 2] = 0; dc[4 * o + 3] = -1.0f; }
        else { dc[4 * o] = 0; dc[4 * o + 1] = 1; dc[4 * o + 2] = 0; dc[4 * o + 3] = 0; }
        continue;
      }
      depth_av /= (float)num;
      float tc[3] = {0, 0, 0}, td = 0.0f, tw = 0.0f;
      for (int i = 0; i < 16; ++i)
        if (smp[i][0] >= 0.0f && smp[i][3] >= depth_av) { tc[0] += smp[i][0] * 1.0f; tc[1] += smp[i][1] * 1.0f; tc[2] += smp[i][2] * 1.0f; td += smp[i][3] * 1.0f; tw += 1.0f; }
      dc[4 * o] = tc[0] / tw; dc[4 * o + 1] = tc[1] / tw; dc[4 * o + 2] = tc[2] / tw; dc[4 * o + 3] = 1.0f;
      dd[o] = td / tw;
    }
}
// bilinear RGBA32F, MIRRORED_REPEAT (view_lod.cpp:52-53)
static void atlas_bilinear(const std::vector<float>& col, uint32_t aw, uint32_t ah, float u, float v, float* out) {
  auto mirror = [](int i, int n) { int p = 2 * n; int m = ((i % p) + p) % p; return m < n ? m : p - 1 - m; };
  float fx = u * (float)aw - 0.5f, fy = v * (float)ah - 0.5f;
  float x0f = floorf(fx), y0f = floorf(fy); float ax = fx - x0f, ay = fy - y0f;
  int x0 = mirror((int)x0f, (int)aw), x1 = mirror((int)x0f + 1, (int)aw), y0 = mirror((int)y0f, (int)ah), y1 = mirror((int)y0f + 1, (int)ah);
  for (int k = 0; k < 4; ++k) {
    float r0 = lerp(col[((size_t)y0 * aw + x0) * 4 + k], col[((size_t)y0 * aw + x1) * 4 + k], ax);
    float r1 = lerp(col[((size_t)y1 * aw + x0) * 4 + k], col[((size_t)y1 * aw + x1) * 4 + k], ax);
    out[k] = lerp(r0, r1, ay);
  }
}
// tsdf_colorfill.fs:30-55 into the framebuffer with depth func LESS (:313)
static void colorfill(orc_ctx* c, const std::vector<float>& sc, const std::vector<float>& sdp) {
  const uint32_t w = c->vw, h = c->vh, aw = c->aw;
  const int num_lods = (int)(c->lod_res.size() / 2);
  auto OFF = [&](int l, int k) { return l < num_lods ? c->lod_off[2 * l + k] : 0u; };   // uniform slots past num_lods are zero (quirk 7)
  auto RES = [&](int l, int k) { return l < num_lods ? c->lod_res[2 * l + k] : 0u; };
  const float rix = 1.0f / (float)aw, riy = 1.0f / (float)h;                            // resolution_inv :497
  for (uint32_t py = 0; py < h; ++py)
    for (uint32_t px = 0; px < w; ++px) {
      float tcx = (float)px / (float)RES(0, 0), tcy = (float)py / (float)RES(0, 1);     // :32
      float out[4] = {0, 0, 0, 0}, d;
      int level = 0;
      for (; level < num_lods; ++level) {                                               // :36-40
        int cx = (int)((float)(int)OFF(level, 0) + (float)(int)RES(level, 0) * tcx), cy = (int)((float)(int)OFF(level, 1) + (float)(int)RES(level, 1) * tcy);
        fetch(sc, sdp, aw, h, cx, cy, out, &d);
        if (out[3] > 0.0f) break;
      }
      if (level > 0) {                                                                  // :42-51
        float ptx = ((float)px + 0.5f) / (float)w, pty = ((float)py + 0.5f) / (float)h; // pass_TexCoord
        auto lod_pos2 = [&](int l, float& ox_, float& oy_) {
          float ofx = (float)OFF(l, 0), ofy = (float)OFF(l, 1), rx_ = (float)RES(l, 0), ry_ = (float)RES(l, 1);
          ox_ = fminf(fmaxf(ofx + rx_ * ptx, ofx + 0.5f), (float)(OFF(l, 0) + RES(l, 0)) - 0.5f);
          oy_ = fminf(fmaxf(ofy + ry_ * pty, ofy + 0.5f), (float)(OFF(l, 1) + RES(l, 1)) - 0.5f);
        };
        float p2x, p2y, p1x, p1y; lod_pos2(level + 2, p2x, p2y); lod_pos2(level + 1, p1x, p1y);
        float c1[4], c2[4];
        atlas_bilinear(sc, aw, h, p1x * rix, p1y * riy, c1);
        atlas_bilinear(sc, aw, h, p2x * rix, p2y * riy, c2);
        float w1 = sqrtf(ptx * ptx + pty * pty);                                        // distance(tc, floor(tc)), tc in (0,1)
        float w2 = 1.0f - w1;
        for (int k = 0; k < 4; ++k) out[k] = (c1[k] * w1 + c2[k] * w2) / (w1 + w2);
      }
      float c0[4], d0;
      fetch(sc, sdp, aw, h, (int)((float)(int)OFF(0, 0) + (float)(int)RES(0, 0) * tcx), (int)((float)(int)OFF(0, 1) + (float)(int)RES(0, 1) * tcy), c0, &d0);   // :54
      size_t o = (size_t)py * w + px;
      if (d0 < c->fb_d[o]) { masked_store(c, &c->fb_c[4 * o], out); c->fb_d[o] = d0; }  // GL_LESS, glColorMask :321-326
    }
}
void orc_fill_colors(orc_ctx* c) {
  // the caller cleared the default framebuffer (kinect_client.cpp:602-612) -- or only its depth (:627)
  for (size_t i = 0; i < c->fb_d.size(); ++i) { if (!c->keep_color) { c->fb_c[4 * i] = c->fb_c[4 * i + 1] = c->fb_c[4 * i + 2] = c->fb_c[4 * i + 3] = 0; } c->fb_d[i] = 1.0f; }
  auto *Tc = c->target_is_A ? &c->atlasA_c : &c->atlasB_c, *Td = c->target_is_A ? &c->atlasA_d : &c->atlasB_d;   // m_view_inpaint
  auto *Sc = c->target_is_A ? &c->atlasB_c : &c->atlasA_c, *Sd = c->target_is_A ? &c->atlasB_d : &c->atlasA_d;   // m_view_inpaint2
  const int num_lods = (int)(c->lod_res.size() / 2);
  transfer(c, *Tc, *Td, *Sc, *Sd);                       // :282-289 ; after the swap m_view_inpaint = S, m_view_inpaint2 = T
  for (int i = 1; i < num_lods; ++i) {
    inpaint_level(c, i - 1, *Sc, *Sd, *Tc, *Td);         // :291-301 reads S, writes level i of T
    transfer(c, *Tc, *Td, *Sc, *Sd);                     // :303-311
  }
  colorfill(c, *Tc, *Td);                                // :315 binds m_view_inpaint2 == T
  c->target_is_A = !c->target_is_A;                      // odd number of swaps per frame
}

// downloads for the tests
void orc_get_view(orc_ctx* c, float* rgba, float* depth, float* nsamples, float* peels) {
  const uint32_t w = c->vw, h = c->vh;
  if (c->fill_holes && (rgba || depth)) {   // level 0 of the current raymarch target
    auto& ac = c->target_is_A ? c->atlasA_c : c->atlasB_c; auto& ad = c->target_is_A ? c->atlasA_d : c->atlasB_d;
    for (uint32_t y = 0; y < h; ++y) {
      if (rgba) memcpy(rgba + (size_t)y * w * 4, &ac[(size_t)y * c->aw * 4], (size_t)w * 16);
      if (depth) memcpy(depth + (size_t)y * w, &ad[(size_t)y * c->aw], (size_t)w * 4);
    }
  } else {
    if (rgba) memcpy(rgba, c->fb_c.data(), c->fb_c.size() * 4);
    if (depth) memcpy(depth, c->fb_d.data(), c->fb_d.size() * 4);
  }
  if (nsamples) memcpy(nsamples, c->nsamples.data(), c->nsamples.size() * 4);
  if (peels) memcpy(peels, c->peels.data(), c->peels.size() * 4);
}
void orc_get_framebuffer(orc_ctx* c, float* rgba, float* depth) {
  memcpy(rgba, c->fb_c.data(), c->fb_c.size() * 4); memcpy(depth, c->fb_d.data(), c->fb_d.size() * 4);
}
// atlas that fillColors() just completed (un-squeezed, all levels); call after orc_fill_colors
void orc_get_atlas(orc_ctx* c, float* rgba, float* depth) {
  auto& ac = c->target_is_A ? c->atlasB_c : c->atlasA_c; auto& ad = c->target_is_A ? c->atlasB_d : c->atlasA_d;
  memcpy(rgba, ac.data(), ac.size() * 4); memcpy(depth, ad.data(), ad.size() * 4);
}
// inject a level-0 image into the current raymarch target (K3/K4 unit tests)
void orc_set_view(orc_ctx* c, const float* rgba, const float* depth) {
  auto& ac = c->target_is_A ? c->atlasA_c : c->atlasB_c; auto& ad = c->target_is_A ? c->atlasA_d : c->atlasB_d;
  for (size_t i = 0; i < ad.size(); ++i) { ac[4 * i] = 0; ac[4 * i + 1] = 1; ac[4 * i + 2] = 0; ac[4 * i + 3] = 0; ad[i] = 1.0f; }
  for (uint32_t y = 0; y < c->vh; ++y) {
    memcpy(&ac[(size_t)y * c->aw * 4], rgba + (size_t)y * c->vw * 4, (size_t)c->vw * 16);
    memcpy(&ad[(size_t)y * c->aw], depth + (size_t)y * c->vw, (size_t)c->vw * 4);
  }
}

// ================================================================= image pre-processing
// Literal restatement of glsl/pre_{morph,depth,boundary,normal,quality}.fs + inc_color.glsl as driven by
// NetKinectArray::processDepth / processTextures (framework/NetKinectArray.cpp:249-288, :309-426).
// Every pass runs per layer over the depth resolution; pass_TexCoord = (pixel + .5) / size; texSizeInv = 1 / size.
void orc_set_raw_frame(orc_ctx* c, const float* raw_depth, const uint8_t* color_rgb) { c->raw_depth = raw_depth; c->color = color_rgb; }
void orc_set_depth_limits(orc_ctx* c, uint32_t i, float mn, float mx) { c->cv_min_d[i] = mn; c->cv_max_d[i] = mx; }
void orc_set_camera_position(orc_ctx* c, uint32_t i, const float* p) { c->cam_pos[i] = {p[0], p[1], p[2]}; }
// NetKinectArray.cpp:343-349: compress = isCompressedDepth(); scale = far - near; scaled_near = scale / 255.0f
// OpenMP threads of every parallel region from now on (bench.py's 1-thread and all-cores CPU baselines); returns the previous maximum
int orc_set_threads(int n) {
#ifdef _OPENMP
  const int old = omp_get_max_threads();
  if (n > 0) omp_set_num_threads(n);
  return old;
#else
  (void)n; return 1;
#endif
}
void orc_set_depth_compression(orc_ctx* c, uint32_t i, int compress, float near_, float far_) {
  const float scale = far_ - near_;
  c->compress[i] = compress != 0; c->dc_near[i] = near_; c->dc_scale[i] = scale; c->dc_scaled_near[i] = scale / 255.0f;
}
void orc_set_preprocess_flags(orc_ctx* c, int filter_textures, int processed_depth, int refine) {
  c->filter_textures = filter_textures; c->use_processed_depth = processed_depth; c->refine_bound = refine;
}

namespace {
// inc_color.glsl:8-46 (the shader divides an already normalised colour by 255 again, :14-16: restated as written)
inline float pivot_rgb(float n) { return (n > 0.04045f ? powf((n + 0.055f) / 1.055f, 2.4f) : n / 12.92f) * 100.0f; }
inline float pivot_xyz(float n) { return n > 0.008856f ? powf(n, (float)(1.0 / 3.0)) : (903.3f * n + 16.0f) / 116.0f; }
inline vec3 rgb_to_lab(vec3 rgb) {
  const float r = pivot_rgb(rgb.x / 255.0f), g = pivot_rgb(rgb.y / 255.0f), b = pivot_rgb(rgb.z / 255.0f);
  const float X = r * 0.4124f + g * 0.3576f + b * 0.1805f, Y = r * 0.2126f + g * 0.7152f + b * 0.0722f, Z = r * 0.0193f + g * 0.1192f + b * 0.9505f;
  const float x = pivot_xyz(X / 95.047f), y = pivot_xyz(Y / 100.000f), z = pivot_xyz(Z / 108.883f);
  return {fmaxf(0.0f, 116.0f * y - 16.0f), 500.0f * (x - y), 200.0f * (y - z)};
}
inline bool in_bbox(const orc_ctx* c, vec3 p) {   // inc_bbox_test.glsl:11-21
  return p.x >= c->cfg.bbox_min[0] && p.y >= c->cfg.bbox_min[1] && p.z >= c->cfg.bbox_min[2] && p.x <= c->cfg.bbox_max[0] && p.y <= c->cfg.bbox_max[1] && p.z <= c->cfg.bbox_max[2];
}
inline float len2(float x, float y) { return sqrtf(x * x + y * y); }
}  // namespace

void orc_process_textures(orc_ctx* c) {
  const int W = (int)c->cfg.depth_w, H = (int)c->cfg.depth_h, N = (int)c->cfg.num_streams, CW = (int)c->cfg.color_w, CH = (int)c->cfg.color_h;
  const size_t P = (size_t)W * H;
  const float tsx = 1.0f / (float)W, tsy = 1.0f / (float)H;                      // texSizeInv, NetKinectArray.cpp:195
  c->pp_depth2.assign(N * P, 0.0f); c->pp_depth_rg.assign(N * P * 2, 0.0f); c->pp_lab.assign(N * P * 3, 0.0f);
  c->pp_depth_b.assign(N * P * 2, 0.0f); c->pp_sil.assign(N * P, 0.0f); c->pp_normal.assign(N * P * 3, 0.0f); c->pp_quality.assign(N * P, 0.0f);
  auto tc = [&](int x, int y, float& u, float& v) { u = ((float)x + 0.5f) / (float)W; v = ((float)y + 0.5f) / (float)H; };

  // ---- "morph": pre_morph.fs mode 0 = dilate(coords, 1) on the raw depth (:73-112, :123-127); mode 1 copies (:130-131)
  const float min_depth = 0.5f, max_depth = 4.5f, max_dist = 0.2f;                 // :32-33, :54
  auto valid_m = [&](float d) { return d > min_depth && d < max_depth; };           // in_bbox(texcoord, depth) returns true (:48)
#pragma omp parallel for collapse(2)
  for (int l = 0; l < N; ++l)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        float u, v; tc(x, y, u, v);
        auto smp = [&](float uu, float vv) { return tex2d_nearest(c->raw_depth, 1, W, H, l, uu, vv, 0); };
        const float depth = smp(u, v);
        float out;
        if (valid_m(depth)) out = depth;
        else {
          float avg = 0.0f, num = 0.0f; bool valid = false;
          for (int dy = -1; dy < 2; ++dy) for (int dx = -1; dx < 2; ++dx) {
            const float ds = smp(u + (float)dx * tsx, v + (float)dy * tsy);
            if (valid_m(ds)) { valid = true; avg += ds; num += 1.0f; }
          }
          if (!valid) out = 0.0f;
          else {
            avg /= num;
            float nd = 0.0f; num = 0.0f; valid = false;
            for (int dy = -1; dy < 2; ++dy) for (int dx = -1; dx < 2; ++dx) {
              const float ds = smp(u + (float)dx * tsx, v + (float)dy * tsy);
              if (valid_m(ds) && fabsf(avg - ds) < max_dist) { valid = true; nd += ds; num += 1.0f; }
            }
            out = valid ? nd / num : 0.0f;
          }
        }
        c->pp_depth2[(size_t)l * P + (size_t)y * W + x] = out;
      }
  // the filter pass reads the processed (dilated) depth when m_use_processed_depth, else the raw array (:286-288, :194)
  const float* fdepth = c->use_processed_depth ? c->pp_depth2.data() : c->raw_depth;

  // ---- "filter": pre_depth.fs main() :129-154, bilateral_filter :85-127
#pragma omp parallel for collapse(2)
  for (int l = 0; l < N; ++l)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        float u, v; tc(x, y, u, v);
        const float mn = c->cv_min_d[l], mx = c->cv_max_d[l];
        // sample(), pre_depth.fs:63-72, with uncompress() :51-61 (sqrt mapping of the 8-bit wire depth)
        auto smp = [&](float uu, float vv) {
          const float t = tex2d_nearest(fdepth, 1, W, H, l, uu, vv, 0);
          if (!c->compress[l]) return t;
          if (t < c->dc_scaled_near[l]) return 0.0f;
          return (t * t + 0.15f * c->dc_scaled_near[l]) * c->dc_scale[l] + c->dc_near[l];
        };
        auto norm = [&](float d) { return (d - mn) / (mx - mn); };
        const float depth = smp(u, v);
        const float dn = norm(depth);
        float wp[3];
        tex3d(c->xyz[l], 3, c->xyz_res[l], u, v, dn, wp);
        const bool is_in_box = in_bbox(c, {wp[0], wp[1], wp[2]});
        // out_Color = rgb_to_lab(get_color(vec3(tc, (dn <= 0 || dn >= 1) ? 1 : dn)))   :136
        float cc[2], col[3];
        tex3d(c->uv[l], 2, c->uv_res[l], u, v, (dn <= 0.0f || dn >= 1.0f) ? 1.0f : dn, cc);
        tex2d_linear_u8(c->color, CW, CH, l, cc[0], cc[1], col);
        const vec3 lab = rgb_to_lab({col[0], col[1], col[2]});
        const size_t o = (size_t)l * P + (size_t)y * W + x;
        c->pp_lab[3 * o] = lab.x; c->pp_lab[3 * o + 1] = lab.y; c->pp_lab[3 * o + 2] = lab.z;
        float od[2] = {0.0f, 0.0f};
        if (is_in_box) {
          if (!c->filter_textures) { od[0] = dn; od[1] = 1.0f; }
          else {
            const float dist_range_max = 0.35f * (depth / 4.5f), dist_range_max_inv = 1.0f / dist_range_max;   // :89-92
            float depth_bf = 0.0f, w = 0.0f, w_range = 0.0f, num = 0.0f;
            for (int dy = -6; dy < 7; ++dy) for (int dx = -6; dx < 7; ++dx) {
              num += 1.0f;
              const float ds = smp(u + (float)dx * tsx, v + (float)dy * tsy);
              const float dr = fabsf(ds - depth);
              if ((ds < mn) || (ds > mx) || (dr > dist_range_max)) continue;                     // is_outside :74-76
              const float gs = 1.0f - len2((float)dx, (float)dy) * (1.0f / 6.0f);               // computeGaussSpace :37-41
              const float gr = 1.0f - fminf(dr, dist_range_max) * dist_range_max_inv;           // computeGaussRange :43-48
              const float ws = gs * gr;
              depth_bf += ws * ds; w += ws; w_range += gr;
            }
            od[0] = norm(depth_bf / w); od[1] = w_range / num;                                   // :124-126
          }
        }
        c->pp_depth_rg[2 * o] = od[0]; c->pp_depth_rg[2 * o + 1] = od[1];
      }

  // ---- "boundary": pre_boundary.fs main() :86-117, get_color_diff :37-55
#pragma omp parallel for collapse(2)
  for (int l = 0; l < N; ++l)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        float u, v; tc(x, y, u, v);
        const size_t o = (size_t)l * P + (size_t)y * W + x;
        float dx_ = tex2d_nearest(c->pp_depth_rg.data(), 2, W, H, l, u, v, 0), dy_ = tex2d_nearest(c->pp_depth_rg.data(), 2, W, H, l, u, v, 1);
        float sil = 1.0f;
        if (dx_ <= 0.0f) { dy_ = 0.0f; sil = 0.0f; }
        else if (!(dy_ > 0.65f)) {                                                               // valid_range :27-30
          sil = 0.0f;
          float color[3];
          tex2d_linear(c->pp_lab.data(), 3, W, H, l, u, v, color);
          float total = 0.0f, num = 0.0f;
          for (int ky = -2; ky < 3; ++ky) for (int kx = -2; kx < 3; ++kx) {
            const float us = u + (float)kx * tsx, vs = v + (float)ky * tsy;
            const float sx = tex2d_nearest(c->pp_depth_rg.data(), 2, W, H, l, us, vs, 0), sy = tex2d_nearest(c->pp_depth_rg.data(), 2, W, H, l, us, vs, 1);
            if (sx > 0.0f && sy > 0.65f) {
              num += 1.0f;
              float cs[3];
              tex2d_linear(c->pp_lab.data(), 3, W, H, l, us, vs, cs);
              total += length(vec3{color[0] - cs[0], color[1] - cs[1], color[2] - cs[2]});
            }
          }
          const float color_dist = (num < 16.0f * 0.5f) ? 1.0f : total / num;                   // total_samples = 16 (:23, :53)
          if (color_dist > 0.5f || !c->refine_bound) { dx_ = -1.0f; dy_ = 0.1f; sil = 0.0f; }
          else dy_ = 1.0f;
        } else dy_ = 0.0f;
        c->pp_depth_b[2 * o] = dx_; c->pp_depth_b[2 * o + 1] = dy_; c->pp_sil[o] = sil;
      }

  // ---- "normal": pre_normal.fs :26-56 (+ mark_brick :33); serial because of the counters
  for (int l = 0; l < N; ++l)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        float u, v; tc(x, y, u, v);
        const size_t o = (size_t)l * P + (size_t)y * W + x;
        auto dsm = [&](float uu, float vv) { return tex2d_nearest(c->pp_depth_b.data(), 2, W, H, l, uu, vv, 0); };
        auto outside = [](float d) { return d <= 0.0f || d >= 1.0f; };
        const float depth = dsm(u, v);
        vec3 n = {0, 0, 0};
        if (!outside(depth)) {
          float wp[3];
          tex3d(c->xyz[l], 3, c->xyz_res[l], u, v, depth, wp);
          mark_brick(c, {wp[0], wp[1], wp[2]});
          float dt = dsm(u, v + tsy), db = dsm(u, v - tsy), dl = dsm(u - tsx, v), dr = dsm(u + tsx, v);
          dt = outside(dt) ? depth : dt; db = outside(db) ? depth : db; dl = outside(dl) ? depth : dl; dr = outside(dr) ? depth : dr;
          float wt[3], wb[3], wl[3], wr[3];
          tex3d(c->xyz[l], 3, c->xyz_res[l], u, v + tsy, dt, wt);
          tex3d(c->xyz[l], 3, c->xyz_res[l], u, v - tsy, db, wb);
          tex3d(c->xyz[l], 3, c->xyz_res[l], u - tsx, v, dl, wl);
          tex3d(c->xyz[l], 3, c->xyz_res[l], u + tsx, v, dr, wr);
          const vec3 a = {wb[0] - wt[0], wb[1] - wt[1], wb[2] - wt[2]}, b = {wl[0] - wr[0], wl[1] - wr[1], wl[2] - wr[2]};
          n = normalize(vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x});
        }
        c->pp_normal[3 * o] = n.x; c->pp_normal[3 * o + 1] = n.y; c->pp_normal[3 * o + 2] = n.z;
      }

  // ---- "quality": pre_quality.fs bilateral_filter :65-119, normal_angle :43-48.  pow() of a negative base is undefined
  // in GLSL; powf is used as is (angle < 0 only for back-facing noise).
#pragma omp parallel for collapse(2)
  for (int l = 0; l < N; ++l)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        float u, v; tc(x, y, u, v);
        const size_t o = (size_t)l * P + (size_t)y * W + x;
        auto dsm = [&](float uu, float vv) { return tex2d_nearest(c->pp_depth_b.data(), 2, W, H, l, uu, vv, 0); };
        auto outside = [](float d) { return d <= 0.0f || d >= 1.0f; };
        const float depth = dsm(u, v);
        float q = 0.0f;
        if (!outside(depth)) {
          const float dist_range_max = 0.35f * (depth / 1.0f), dist_range_max_inv = 1.0f / dist_range_max;
          float w_range = 0.0f, border = 0.0f, num = 0.0f;
          for (int dy = -6; dy < 7; ++dy) for (int dx = -6; dx < 7; ++dx) {
            num += 1.0f;
            const float ds = dsm(u + (float)dx * tsx, v + (float)dy * tsy);
            const float dr = fabsf(ds - depth);
            if (outside(ds) || dr > dist_range_max) { border += 1.0f; continue; }
            w_range += 1.0f - fminf(dr, dist_range_max) * dist_range_max_inv;
          }
          const float lateral = 1.0f - border / num;
          q = powf(lateral, 6.0f);
          q *= powf(w_range / num, 6.0f);
          q /= depth * 6.5f;
          float wn[3], wp[3];
          tex2d_linear(c->pp_normal.data(), 3, W, H, l, u, v, wn);
          tex3d(c->xyz[l], 3, c->xyz_res[l], u, v, depth, wp);
          const vec3 tocam = normalize(c->cam_pos[l] - vec3{wp[0], wp[1], wp[2]});
          const float angle = dot(tocam, vec3{wn[0], wn[1], wn[2]});
          q *= powf(angle, 2.0f);
        }
        c->pp_quality[o] = q;
      }
  // the path now reads the products (texture units 2, 3, 5 of NetKinectArray::bindToTextureUnits, :451-462)
  c->depth = c->pp_depth_b.data(); c->quality = c->pp_quality.data(); c->silhouette = c->pp_sil.data();
}
void orc_get_preprocessed(orc_ctx* c, float* depth2, float* depth_rg, float* lab, float* depth_b, float* sil, float* normal, float* quality) {
  auto cp = [](float* d, const std::vector<float>& s) { if (d) memcpy(d, s.data(), s.size() * 4); };
  cp(depth2, c->pp_depth2); cp(depth_rg, c->pp_depth_rg); cp(lab, c->pp_lab); cp(depth_b, c->pp_depth_b); cp(sil, c->pp_sil); cp(normal, c->pp_normal); cp(quality, c->pp_quality);
}

// ---------------------------------------------------------------- sampling primitives for unit tests
void orc_tex3d(const float* t, int nc, const uint32_t* res, float u, float v, float w, float* out) { tex3d(t, nc, res, u, v, w, out); }
void orc_tex2d_linear(const float* t, int nc, int w, int h, int layer, float u, float v, float* out) { tex2d_linear(t, nc, w, h, layer, u, v, out); }
float orc_tex2d_nearest(const float* t, int nc, int w, int h, int layer, float u, float v, int ch) { return tex2d_nearest(t, nc, w, h, layer, u, v, ch); }
// The matrix block of draw() alone (also cross-checked against the reference's own gloost/glm, oracle/ref/ref_view_math.cpp):
// vol_to_world[16], image_to_eye[16], NormalMatrix[16], CameraPos[3].
void orc_view_matrices(orc_ctx* c, const float* mv, const float* proj, float* out51) {
  view_mats V = make_view(c, mv, proj);
  memcpy(out51, V.v2w.m, 64); memcpy(out51 + 16, V.img_to_eye.m, 64); memcpy(out51 + 32, V.normal.m, 64);
  out51[48] = V.cam_vol.x; out51[49] = V.cam_vol.y; out51[50] = V.cam_vol.z;
}


// ================================================================= frame ingest
// S3TC block decode.  The reference hands DXT1/DXT5 colour to GL as GL_COMPRESSED_RGBA_S3TC_DXT{1,5}_EXT
// (NetKinectArray.cpp:147-154) and decodes the same blocks on the CPU with its vendored squish
// (external/squish, NetKinectArray.cpp:620).  GL leaves the interpolation precision to the implementation; the
// integer decode of squish is the definition used here (colourblock.cpp:160-212, alpha.cpp:297-348: 565 endpoints
// expanded by bit replication, (2a+b)/3 and (a+b)/2 in integer arithmetic; DXT1 three-colour mode -> transparent black).
// blocks: row-major 4x4 blocks, 8 bytes (DXT1) or 16 bytes (DXT5: alpha block then colour block); rgba: [h][w][4].
static void dxt_endpoint(const uint8_t* b, int out[3]) {
  const int v = b[0] | (b[1] << 8);
  const int r = (v >> 11) & 31, g = (v >> 5) & 63, bl = v & 31;
  out[0] = (r << 3) | (r >> 2); out[1] = (g << 2) | (g >> 4); out[2] = (bl << 3) | (bl >> 2);
}
void orc_decode_dxt(const uint8_t* blocks, int w, int h, int format, uint8_t* rgba) {
  const int bpb = format == 1 ? 8 : 16;
  size_t k = 0;
  for (int by = 0; by < h; by += 4)
    for (int bx = 0; bx < w; bx += 4, ++k) {
      const uint8_t* blk = blocks + k * bpb;
      const uint8_t* cb = format == 1 ? blk : blk + 8;
      int e0[3], e1[3];
      dxt_endpoint(cb, e0); dxt_endpoint(cb + 2, e1);
      const int a = cb[0] | (cb[1] << 8), b = cb[2] | (cb[3] << 8);
      const bool three = format == 1 && a <= b;
      uint8_t pal[4][4];
      for (int i = 0; i < 3; ++i) {
        pal[0][i] = (uint8_t)e0[i]; pal[1][i] = (uint8_t)e1[i];
        pal[2][i] = (uint8_t)(three ? (e0[i] + e1[i]) / 2 : (2 * e0[i] + e1[i]) / 3);
        pal[3][i] = (uint8_t)(three ? 0 : (e0[i] + 2 * e1[i]) / 3);
      }
      pal[0][3] = pal[1][3] = pal[2][3] = 255; pal[3][3] = three ? 0 : 255;
      uint8_t alpha[16];
      if (format != 1) {
        const int a0 = blk[0], a1 = blk[1];
        uint8_t at[8];
        at[0] = (uint8_t)a0; at[1] = (uint8_t)a1;
        if (a0 <= a1) { for (int i = 1; i < 5; ++i) at[1 + i] = (uint8_t)(((5 - i) * a0 + i * a1) / 5); at[6] = 0; at[7] = 255; }
        else for (int i = 1; i < 7; ++i) at[1 + i] = (uint8_t)(((7 - i) * a0 + i * a1) / 7);
        uint64_t bits = 0;
        for (int i = 0; i < 6; ++i) bits |= (uint64_t)blk[2 + i] << (8 * i);
        for (int i = 0; i < 16; ++i) alpha[i] = at[(bits >> (3 * i)) & 7];
      }
      for (int py = 0; py < 4; ++py)
        for (int px = 0; px < 4; ++px) {
          const int x = bx + px, y = by + py;
          if (x >= w || y >= h) continue;
          const int idx = (cb[4 + py] >> (2 * px)) & 3;
          uint8_t* o = rgba + 4 * ((size_t)y * w + x);
          o[0] = pal[idx][0]; o[1] = pal[idx][1]; o[2] = pal[idx][2];
          o[3] = format == 1 ? pal[idx][3] : alpha[4 * py + px];
        }
    }
}


// ================================================================= inverse calibration volumes
// kinect::Frustum (framework/calibration/frustum.cpp) from the 8 corner texels of a forward volume
// (getCornerPoints, calibration_inverter.cpp:117-133), restated in glm's fp32 operation order
// (external/glm-0.9.5.3 detail/func_geometric.inl: dot = x+y+z left to right, vec4 dot = (x+y)+(z+w),
// normalize = v * (1/sqrt(dot)), cross as written there).
namespace {
struct Frustum { vec3 corner[8]; float plane[6][4]; };
inline vec3 v_cross(vec3 x, vec3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
inline float v_dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 v_normalize(vec3 v) { const float s = 1.0f / sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); return {v.x * s, v.y * s, v.z * s}; }
inline vec3 v_avg4(vec3 a, vec3 b, vec3 c, vec3 d) { return (a + b + c + d) / 4.0f; }
Frustum make_frustum(const float* xyz, const uint32_t res[3]) {
  Frustum f;
  const uint32_t ex = res[0] - 1, ey = res[1] - 1, ez = res[2] - 1;
  auto at = [&](uint32_t x, uint32_t y, uint32_t z) { const float* t = xyz + 3 * ((size_t)z * res[0] * res[1] + (size_t)y * res[0] + x); return vec3{t[0], t[1], t[2]}; };
  f.corner[0] = at(0, 0, 0); f.corner[1] = at(0, ey, 0); f.corner[2] = at(ex, ey, 0); f.corner[3] = at(ex, 0, 0);
  f.corner[4] = at(0, 0, ez); f.corner[5] = at(0, ey, ez); f.corner[6] = at(ex, ey, ez); f.corner[7] = at(ex, 0, ez);
  const vec3* c = f.corner;
  vec3 e[12];                                                             // getEdgeCenters, frustum.cpp
  for (int i = 0; i < 4; ++i) { e[i] = (c[i] + c[(i + 1) & 3]) * 0.5f; e[4 + i] = (c[4 + i] + c[4 + ((i + 1) & 3)]) * 0.5f; e[8 + i] = (c[i] + c[4 + i]) * 0.5f; }
  const vec3 centre[6] = {v_avg4(c[0], c[1], c[2], c[3]), v_avg4(c[4], c[5], c[6], c[7]), v_avg4(c[0], c[1], c[4], c[5]),
                          v_avg4(c[2], c[3], c[6], c[7]), v_avg4(c[1], c[2], c[5], c[6]), v_avg4(c[0], c[3], c[4], c[7])};   // getSideCenters
  const vec3 normal[6] = {v_normalize(v_cross(e[0] - e[2], e[3] - e[2])), v_normalize(v_cross(e[4] - e[6], e[5] - e[7])),
                          v_normalize(v_cross(e[0] - e[4], e[9] - e[8])), v_normalize(v_cross(e[2] - e[6], e[11] - e[10])),
                          v_normalize(v_cross(e[9] - e[10], e[1] - e[5])), v_normalize(v_cross(e[8] - e[11], e[7] - e[3]))};  // getSideNormals
  for (int i = 0; i < 6; ++i) { f.plane[i][0] = normal[i].x; f.plane[i][1] = normal[i].y; f.plane[i][2] = normal[i].z; f.plane[i][3] = -v_dot(normal[i], centre[i]); }
  return f;
}
inline bool frustum_inside(const Frustum& f, vec3 p) {                    // Frustum::inside: dot(plane, vec4(p, 1)) < 0 -> outside
  for (int i = 0; i < 6; ++i) {
    const float* pl = f.plane[i];
    if ((pl[0] * p.x + pl[1] * p.y) + (pl[2] * p.z + pl[3] * 1.0f) < 0.0f) return false;
  }
  return true;
}
vec3 closest_point(vec3 p, vec3 u, vec3 q, vec3 v) {                      // closestPoint(), frustum.cpp
  const vec3 w0 = p - q;
  const float a = v_dot(u, u), b = v_dot(u, v), c = v_dot(v, v), d = v_dot(u, w0), e = v_dot(v, w0);
  const float sc = (b * e - c * d) / (a * c - b * b), tc = (a * e - b * d) / (a * c - b * b);
  return ((p + u * sc) + (q + v * tc)) * 0.5f;
}
}  // namespace

// planes[6][4], camera[3] (Frustum::getCameraPos -> CalibVolumes::getCameraPositions, CalibVolumes.cpp:224-230)
void orc_frustum(const float* xyz, const uint32_t res[3], float* planes, float* camera) {
  const Frustum f = make_frustum(xyz, res);
  memcpy(planes, f.plane, sizeof(f.plane));
  const vec3* c = f.corner;
  const vec3 cn = v_avg4(c[0], c[1], c[2], c[3]), cf = v_avg4(c[4], c[5], c[6], c[7]), dir = cf - cn;
  const vec3 s = (closest_point(c[0], c[0] - c[4], cn, dir) + closest_point(c[1], c[1] - c[5], cn, dir) +
                  closest_point(c[2], c[2] - c[6], cn, dir) + closest_point(c[3], c[3] - c[7], cn, dir)) / 4.0f;
  camera[0] = s.x; camera[1] = s.y; camera[2] = s.z;
}

// CalibrationInverter::calculateInverseVolumes for ONE sensor, calibration_inverter.cpp:68-115.
// The 8-NN comes from CGAL::Orthogonal_k_neighbor_search (third-party, not vendored: CGAL 4.x as pulled in by
// framework/CMakeLists.txt) with the EPICK kernel: exact k nearest by squared distance in DOUBLE over the float
// coordinates, reported in ascending distance.  CGAL leaves the order of equidistant samples open; here ties go to
// the sample enumerated first by getXyzSamples (:40-55: x outer, z inner).  This restatement is a brute-force scan.
// inverseDistance (:57-66): fp32, weight = 1 / distance (a coincident sample gives inf -> NaN, as in the reference).
void orc_invert_calibration(const float* xyz, const uint32_t res[3], const float bbox_min[3], const float bbox_max[3], const uint32_t rinv[3], float* out) {
  const Frustum fr = make_frustum(xyz, res);
  const size_t n = (size_t)res[0] * res[1] * res[2];
  std::vector<double> px(n), py(n), pz(n);
  std::vector<uint32_t> ix(n), iy(n), iz(n);
  {
    size_t k = 0;                                                         // getXyzSamples enumeration order
    for (uint32_t x = 0; x < res[0]; ++x) for (uint32_t y = 0; y < res[1]; ++y) for (uint32_t z = 0; z < res[2]; ++z, ++k) {
      const float* t = xyz + 3 * ((size_t)z * res[0] * res[1] + (size_t)y * res[0] + x);
      px[k] = t[0]; py[k] = t[1]; pz[k] = t[2]; ix[k] = x; iy[k] = y; iz[k] = z;
    }
  }
  const vec3 dims = {bbox_max[0] - bbox_min[0], bbox_max[1] - bbox_min[1], bbox_max[2] - bbox_min[2]};
  const vec3 vstep = {1.0f / (float)rinv[0], 1.0f / (float)rinv[1], 1.0f / (float)rinv[2]};
  const vec3 sstep = {dims.x * vstep.x, dims.y * vstep.y, dims.z * vstep.z};
  const vec3 start = {bbox_min[0] + sstep.x * 0.5f, bbox_min[1] + sstep.y * 0.5f, bbox_min[2] + sstep.z * 0.5f};
  const vec3 cd = {(float)res[0], (float)res[1], (float)res[2]};
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
  for (uint32_t z = 0; z < rinv[2]; ++z)
    for (uint32_t y = 0; y < rinv[1]; ++y)
      for (uint32_t x = 0; x < rinv[0]; ++x) {
        float* o = out + 4 * ((size_t)z * rinv[0] * rinv[1] + (size_t)y * rinv[0] + x);
        const vec3 p = {start.x + (float)x * sstep.x, start.y + (float)y * sstep.y, start.z + (float)z * sstep.z};
        if (!frustum_inside(fr, p)) { o[0] = o[1] = o[2] = o[3] = -1.0f; continue; }
        double bd[8]; size_t bi[8]; int cnt = 0;
        for (size_t k = 0; k < n; ++k) {
          const double dx = (double)p.x - px[k], dy = (double)p.y - py[k], dz = (double)p.z - pz[k];
          double d = 0.0; d += dx * dx; d += dy * dy; d += dz * dz;
          if (cnt == 8 && !(d < bd[7])) continue;                         // later index loses ties
          int j = cnt < 8 ? cnt++ : 7;
          while (j > 0 && d < bd[j - 1]) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; --j; }
          bd[j] = d; bi[j] = k;
        }
        float tw = 0.0f; vec3 wi = {0.0f, 0.0f, 0.0f};
        for (int j = 0; j < cnt; ++j) {
          const size_t k = bi[j];
          const vec3 dv = {(float)px[k] - p.x, (float)py[k] - p.y, (float)pz[k] - p.z};            // glm::distance = length(p1 - p0)
          const float w = 1.0f / sqrtf(dv.x * dv.x + dv.y * dv.y + dv.z * dv.z);
          wi = {wi.x + w * (float)ix[k], wi.y + w * (float)iy[k], wi.z + w * (float)iz[k]};
          tw += w;
        }
        wi = {wi.x / tw, wi.y / tw, wi.z / tw};
        o[0] = (wi.x + 0.5f) / cd.x; o[1] = (wi.y + 0.5f) / cd.y; o[2] = (wi.z + 0.5f) / cd.z; o[3] = 1.0f;
      }
}


// ================================================================= point back-end (first half)
// kinect::ReconPoints::draw(), framework/reconstruction/recon_points.cpp:71-111, with glsl/points.vs / .gs / .fs:
// one GL point per depth pixel and layer, layers in order, depth test GL_LESS into the cleared default framebuffer.
// Every varying is `flat`, so a pixel shows the colour of the nearest point that covers it; between equal depths the
// point drawn first wins (layer, then row-major pixel order: the order of the vertex buffer, recon_points.cpp:46-52).
// Definitions GL leaves to the implementation (the kernels use the same ones):
//   * P * MV is formed once in double and rounded to float (gl_Position = gl_ProjectionMatrix * gl_ModelViewMatrix * v);
//   * gl_NormalMatrix, redeclared as a mat4 uniform by points.fs, is inverseTranspose(MV) (the fixed-function matrix);
//   * a point is clipped by its centre; gl_PointSize is clamped to [1, 2047] and, with GL_POINT_SPRITE enabled
//     (kinect_client.cpp:258-259), not rounded: it covers the pixels whose centre lies in [c - s/2, c + s/2) per axis;
//   * window z = ndc.z * 0.5 + 0.5 in fp32 (no 24-bit quantisation).
void orc_set_normals(orc_ctx* c, const float* normals) { c->normals = normals; }
void orc_draw_points(orc_ctx* c, const float* mv16, const float* proj16) {
  const view_mats V = make_view(c, mv16, proj16);
  double mvd[16], prd[16], pm[16];
  to_d(mv16, mvd); to_d(proj16, prd); mul_d(prd, mvd, pm);
  const mat4 PMV = to_f(pm);
  const int W = (int)c->cfg.depth_w, H = (int)c->cfg.depth_h, N = (int)c->cfg.num_streams, CW = (int)c->cfg.color_w, CH = (int)c->cfg.color_h;
  const int vw = (int)c->vw, vh = (int)c->vh;
  const float* normals = c->normals ? c->normals : (c->pp_normal.empty() ? nullptr : c->pp_normal.data());
  for (size_t i = 0; i < c->fb_d.size(); ++i) { c->fb_c[4 * i] = c->fb_c[4 * i + 1] = c->fb_c[4 * i + 2] = c->fb_c[4 * i + 3] = 0.0f; c->fb_d[i] = 1.0f; }
  const float stepX = 1.0f / (float)W, stepY = 1.0f / (float)H;                     // recon_points.cpp:44-45
  const float max_size = c->shade_mode == 3 ? 4.0f : 10.0f;                          // points.gs:49-53
  for (int l = 0; l < N; ++l)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const float u = (float)(((double)x + 0.5) * (double)stepX), v = (float)(((double)y + 0.5) * (double)stepY);   // :48
        const float depth = c->depth[((size_t)l * W * H + (size_t)y * W + x) * 2];   // texel (x, y) at its own centre
        float pc[3], tc[2];
        tex3d(c->xyz[l], 3, c->xyz_res[l], u, v, depth, pc);                          // points.vs:27
        tex3d(c->uv[l], 2, c->uv_res[l], u, v, depth, tc);                            // :29
        const vec3 pos_cs = {pc[0], pc[1], pc[2]};
        if (!in_bbox(c, pos_cs) || depth <= 0.0f) continue;                          // points.gs:36-38
        if (tc[0] > 0.99f || tc[0] < 0.01f || tc[1] > 0.99f || tc[1] < 0.01f) continue;   // points.fs:38-41 (flat: the whole point)
        const vec4 pe = mul(V.mv, {pos_cs.x, pos_cs.y, pos_cs.z, 1.0f});
        const vec3 pos_es = {pe.x, pe.y, pe.z};
        const vec4 clip = mul(PMV, {pos_cs.x, pos_cs.y, pos_cs.z, 1.0f});
        if (!(clip.w > 0.0f) || fabsf(clip.x) > clip.w || fabsf(clip.y) > clip.w || fabsf(clip.z) > clip.w) continue;
        const float xw = (clip.x / clip.w * 0.5f + 0.5f) * (float)vw, yw = (clip.y / clip.w * 0.5f + 0.5f) * (float)vh;
        const float zw = clip.z / clip.w * 0.5f + 0.5f;
        if (!(zw < 1.0f)) continue;
        const float size = fminf(fmaxf(max_size / length(pos_es), 1.0f), 2047.0f);   // points.gs:55
        const float half = size * 0.5f;
        int x0 = (int)ceilf((xw - half) - 0.5f), x1 = (int)ceilf((xw + half) - 0.5f) - 1;
        int y0 = (int)ceilf((yw - half) - 0.5f), y1 = (int)ceilf((yw + half) - 0.5f) - 1;
        x0 = x0 < 0 ? 0 : x0; y0 = y0 < 0 ? 0 : y0; x1 = x1 > vw - 1 ? vw - 1 : x1; y1 = y1 > vh - 1 ? vh - 1 : y1;
        vec3 out = {0, 0, 0};
        bool shaded = false;
        for (int py = y0; py <= y1; ++py)
          for (int px = x0; px <= x1; ++px) {
            const size_t o = (size_t)py * vw + px;
            if (!(zw < c->fb_d[o])) continue;                                        // GL_LESS
            if (!shaded) {                                                           // points.fs:64-75
              shaded = true;
              if (c->shade_mode == 3) out = {camera_colors[l & 7][0], camera_colors[l & 7][1], camera_colors[l & 7][2]};
              else {
                float col[3];
                tex2d_linear_u8(c->color, CW, CH, l, tc[0], tc[1], col);
                vec3 n = {0, 0, 0};
                if (normals) { const float* t = normals + ((size_t)l * W * H + (size_t)y * W + x) * 3; n = {t[0], t[1], t[2]}; }
                const float* mi = V.mv_inv.m;                                        // inverseTranspose(MV) * (n, 0): columns of MV^-1 dotted with n
                const vec3 vn = {mi[0] * n.x + mi[1] * n.y + mi[2] * n.z, mi[4] * n.x + mi[5] * n.y + mi[6] * n.z, mi[8] * n.x + mi[9] * n.y + mi[10] * n.z};
                out = shade(c, V, pos_es, vn, {col[0], col[1], col[2]});
              }
            }
            c->fb_c[4 * o] = out.x; c->fb_c[4 * o + 1] = out.y; c->fb_c[4 * o + 2] = out.z; c->fb_c[4 * o + 3] = 1.0f;
            c->fb_d[o] = zw;
          }
      }
}


// ================================================================= triangle-grid back-end (second half)
// kinect::ReconTrigrid::draw(), framework/reconstruction/recon_trigrid.cpp:85-148, with glsl/trigrid_accum.{vs,gs,fs} and
// trigrid_normalize.fs: two triangles per depth-pixel cell and layer; pass 1 z-buffers the surfaces, pass 2 adds up
// quality-weighted shaded colour of every fragment within epsilon (0.075) of the front surface, pass 3 divides.
// Restated literally, including
//   * the vertex buffer's swapped loop bounds (recon_trigrid.cpp:53-54: y < width, x < height): cells x < H, y < W --
//     columns >= H of a W x H image are never drawn and the rows past the image collapse to zero-area triangles;
//   * trigrid_accum.fs:69: gl_FragCoord.xy + 0.5 (one pixel diagonal offset of the reprojected surface position).
// Definitions GL leaves to the implementation (shared with the kernels): P*MV formed in double; gl_NormalMatrix unused here;
// a triangle with a vertex at w <= 0 is dropped; pixel-centre sampling, edge functions in fp32 with a top-left style
// tie-break on exact zeros; window z linear in screen space, fragments outside 0 <= z <= 1 dropped; smooth varyings
// perspective-correct as sum(l_i a_i / w_i) / sum(l_i / w_i); depth and quality images sampled NEAREST (the positions are
// texel centres or clamp to them); additive blending in draw order (fp32).
}  // extern "C" (helpers with templates follow)
namespace {
struct TriVert { vec3 pos_cs, pos_es; float tcx, tcy, depth, quality, xw, yw, zw, iw; bool front; };
struct TriSetup { TriVert v[3]; vec3 normal; float area; bool ok; };

TriVert tri_vertex(const orc_ctx* c, const view_mats& V, const mat4& PMV, int l, int gx, int gy) {
  const int W = (int)c->cfg.depth_w, H = (int)c->cfg.depth_h;
  const float stepX = 1.0f / (float)W, stepY = 1.0f / (float)H;                       // recon_trigrid.cpp:51-52
  const float u = (float)(((double)gx + 0.5) * (double)stepX), v = (float)(((double)gy + 0.5) * (double)stepY);
  TriVert t;
  t.depth = tex2d_nearest(c->depth, 2, W, H, l, u, v, 0);
  t.quality = tex2d_nearest(c->quality, 1, W, H, l, u, v, 0);
  float pc[3], tc[2];
  tex3d(c->xyz[l], 3, c->xyz_res[l], u, v, t.depth, pc);
  tex3d(c->uv[l], 2, c->uv_res[l], u, v, t.depth, tc);
  t.pos_cs = {pc[0], pc[1], pc[2]}; t.tcx = tc[0]; t.tcy = tc[1];
  const vec4 pe = mul(V.mv, {pc[0], pc[1], pc[2], 1.0f});
  t.pos_es = {pe.x, pe.y, pe.z};
  const vec4 clip = mul(PMV, {pc[0], pc[1], pc[2], 1.0f});
  t.front = clip.w > 0.0f;
  t.iw = 1.0f / clip.w;
  t.xw = (clip.x / clip.w * 0.5f + 0.5f) * (float)c->vw;
  t.yw = (clip.y / clip.w * 0.5f + 0.5f) * (float)c->vh;
  t.zw = clip.z / clip.w * 0.5f + 0.5f;
  return t;
}
TriSetup tri_setup(const orc_ctx* c, TriVert a, TriVert b, TriVert d) {                 // trigrid_accum.gs
  TriSetup T; T.v[0] = a; T.v[1] = b; T.v[2] = d; T.ok = false;
  if (a.depth < 0.0f || b.depth < 0.0f || d.depth < 0.0f) return T;                    // validSurface, :31-42
  const float avg = (a.depth + b.depth + d.depth) / 3.0f;
  const float l = c->min_length * avg * 4.0f;
  if (!(length(b.pos_cs - a.pos_cs) < l && length(d.pos_cs - a.pos_cs) < l && length(d.pos_cs - b.pos_cs) < l)) return T;
  if (!(a.front && b.front && d.front)) return T;
  const vec3 ea = b.pos_es - a.pos_es, eb = d.pos_es - a.pos_es;
  T.normal = normalize(vec3{ea.y * eb.z - eb.y * ea.z, ea.z * eb.x - eb.z * ea.x, ea.x * eb.y - eb.x * ea.y});   // normalize(cross(a, b)), :59
  T.area = (b.xw - a.xw) * (d.yw - a.yw) - (d.xw - a.xw) * (b.yw - a.yw);
  if (!(T.area != 0.0f)) return T;
  T.ok = true;
  return T;
}
struct Fragment { float z, tcx, tcy, quality; vec3 pos_es, pos_cs; };
// coverage + interpolation at pixel centre (px + .5, py + .5); false: not covered
inline bool tri_fragment(const TriSetup& T, int px, int py, Fragment& f) {
  const float x = (float)px + 0.5f, y = (float)py + 0.5f;
  const TriVert &a = T.v[0], &b = T.v[1], &d = T.v[2];
  float e0 = (d.xw - b.xw) * (y - b.yw) - (d.yw - b.yw) * (x - b.xw);                    // edge b->d, opposite a
  float e1 = (a.xw - d.xw) * (y - d.yw) - (a.yw - d.yw) * (x - d.xw);                    // edge d->a, opposite b
  float e2 = (b.xw - a.xw) * (y - a.yw) - (b.yw - a.yw) * (x - a.xw);                    // edge a->b, opposite d
  const float sgn = T.area > 0.0f ? 1.0f : -1.0f;
  const float ex[3] = {(d.xw - b.xw) * sgn, (a.xw - d.xw) * sgn, (b.xw - a.xw) * sgn}, ey[3] = {(d.yw - b.yw) * sgn, (a.yw - d.yw) * sgn, (b.yw - a.yw) * sgn};
  const float ee[3] = {e0 * sgn, e1 * sgn, e2 * sgn};
  for (int i = 0; i < 3; ++i) {
    if (ee[i] < 0.0f) return false;
    if (ee[i] == 0.0f && !(ey[i] > 0.0f || (ey[i] == 0.0f && ex[i] < 0.0f))) return false;   // a pixel centre exactly on an edge belongs to one side only
    if (!(ee[i] >= 0.0f)) return false;                                                      // NaN
  }
  const float l0 = e0 / T.area, l1 = e1 / T.area, l2 = e2 / T.area;
  f.z = l0 * a.zw + l1 * b.zw + l2 * d.zw;
  if (!(f.z >= 0.0f && f.z <= 1.0f)) return false;
  const float w0 = l0 * a.iw, w1 = l1 * b.iw, w2 = l2 * d.iw, iw = w0 + w1 + w2;
  auto ip = [&](float p, float q, float r) { return (w0 * p + w1 * q + w2 * r) / iw; };
  f.tcx = ip(a.tcx, b.tcx, d.tcx); f.tcy = ip(a.tcy, b.tcy, d.tcy); f.quality = ip(a.quality, b.quality, d.quality);
  f.pos_es = {ip(a.pos_es.x, b.pos_es.x, d.pos_es.x), ip(a.pos_es.y, b.pos_es.y, d.pos_es.y), ip(a.pos_es.z, b.pos_es.z, d.pos_es.z)};
  f.pos_cs = {ip(a.pos_cs.x, b.pos_cs.x, d.pos_cs.x), ip(a.pos_cs.y, b.pos_cs.y, d.pos_cs.y), ip(a.pos_cs.z, b.pos_cs.z, d.pos_cs.z)};
  return true;
}
// the tests every stage applies (trigrid_accum.fs:44-62); n = -normalize(pass_normal_es)
inline bool tri_fragment_kept(const orc_ctx* c, const TriSetup& T, const Fragment& f, vec3& n) {
  if (!in_bbox(c, f.pos_cs)) return false;
  if (f.tcx > 0.99f || f.tcx < 0.01f || f.tcy > 0.99f || f.tcy < 0.01f) return false;
  const vec3 nn = normalize(T.normal);
  n = {-nn.x, -nn.y, -nn.z};
  if (dot(n, normalize(f.pos_es)) > 0.0f) return false;                                  // backface culling
  return true;
}
template <typename F>
void for_each_triangle(orc_ctx* c, const view_mats& V, const mat4& PMV, F&& body) {
  const int W = (int)c->cfg.depth_w, H = (int)c->cfg.depth_h, N = (int)c->cfg.num_streams, vw = (int)c->vw, vh = (int)c->vh;
  for (int l = 0; l < N; ++l)
    for (int y = 0; y < W; ++y)                                                           // sic: recon_trigrid.cpp:53
      for (int x = 0; x < H; ++x) {                                                       // sic: :54
        const TriVert v00 = tri_vertex(c, V, PMV, l, x, y), v10 = tri_vertex(c, V, PMV, l, x + 1, y), v01 = tri_vertex(c, V, PMV, l, x, y + 1),
                      v11 = tri_vertex(c, V, PMV, l, x + 1, y + 1);
        const TriSetup tris[2] = {tri_setup(c, v00, v10, v01), tri_setup(c, v10, v11, v01)};   // :55-61
        for (const TriSetup& T : tris) {
          if (!T.ok) continue;
          const float minx = fminf(fminf(T.v[0].xw, T.v[1].xw), T.v[2].xw), maxx = fmaxf(fmaxf(T.v[0].xw, T.v[1].xw), T.v[2].xw);
          const float miny = fminf(fminf(T.v[0].yw, T.v[1].yw), T.v[2].yw), maxy = fmaxf(fmaxf(T.v[0].yw, T.v[1].yw), T.v[2].yw);
          if (!(maxx >= 0.0f && maxy >= 0.0f && minx <= (float)vw && miny <= (float)vh)) continue;
          const int x0 = (int)fmaxf(floorf(minx - 0.5f), 0.0f), x1 = (int)fminf(ceilf(maxx - 0.5f), (float)(vw - 1));
          const int y0 = (int)fmaxf(floorf(miny - 0.5f), 0.0f), y1 = (int)fminf(ceilf(maxy - 0.5f), (float)(vh - 1));
          for (int py = y0; py <= y1; ++py)
            for (int px = x0; px <= x1; ++px) {
              Fragment f;
              if (!tri_fragment(T, px, py, f)) continue;
              vec3 n;
              if (!tri_fragment_kept(c, T, f, n)) continue;
              body(l, px, py, f, n);
            }
        }
      }
}
}  // namespace
extern "C" {

void orc_set_min_length(orc_ctx* c, float v) { c->min_length = v; }
void orc_draw_trigrid(orc_ctx* c, const float* mv16, const float* proj16) {
  const view_mats V = make_view(c, mv16, proj16);
  double mvd[16], prd[16], pm[16];
  to_d(mv16, mvd); to_d(proj16, prd); mul_d(prd, mvd, pm);
  const mat4 PMV = to_f(pm);
  const int vw = (int)c->vw, vh = (int)c->vh, CW = (int)c->cfg.color_w, CH = (int)c->cfg.color_h;
  std::vector<float> zbuf((size_t)vw * vh, 1.0f), acc((size_t)vw * vh * 4, 0.0f);
  for_each_triangle(c, V, PMV, [&](int, int px, int py, const Fragment& f, vec3) {         // stage 0: depth only, GL_LESS
    float& z = zbuf[(size_t)py * vw + px];
    if (f.z < z) z = f.z;
  });
  const float epsilon = 0.075f;                                                            // recon_trigrid.cpp:35
  for_each_triangle(c, V, PMV, [&](int l, int px, int py, const Fragment& f, vec3 n) {     // stage 1: blend ONE, ONE
    const float depth_curr = zbuf[(size_t)py * vw + px];
    const vec4 pc = mul(V.img_to_eye, {((float)px + 0.5f) + 0.5f, ((float)py + 0.5f) + 0.5f, depth_curr, 1.0f});   // sic, :69
    const vec3 es = {pc.x / pc.w, pc.y / pc.w, pc.z / pc.w};
    if (epsilon < length(es - f.pos_es)) return;
    vec3 col;
    if (c->shade_mode == 3) col = {camera_colors[l & 7][0], camera_colors[l & 7][1], camera_colors[l & 7][2]};
    else {
      float t[3];
      tex2d_linear_u8(c->color, CW, CH, l, f.tcx, f.tcy, t);
      col = shade(c, V, f.pos_es, n, {t[0], t[1], t[2]});
    }
    float* a = &acc[4 * ((size_t)py * vw + px)];
    a[0] += col.x * f.quality; a[1] += col.y * f.quality; a[2] += col.z * f.quality; a[3] += f.quality;
  });
  for (size_t i = 0; i < (size_t)vw * vh; ++i) {                                           // trigrid_normalize.fs
    const float* a = &acc[4 * i];
    if (a[3] > 0.0f) { c->fb_c[4 * i] = a[0] / a[3]; c->fb_c[4 * i + 1] = a[1] / a[3]; c->fb_c[4 * i + 2] = a[2] / a[3]; c->fb_c[4 * i + 3] = a[3] / a[3]; c->fb_d[i] = zbuf[i]; }
    else { c->fb_c[4 * i] = c->fb_c[4 * i + 1] = c->fb_c[4 * i + 2] = c->fb_c[4 * i + 3] = 0.0f; c->fb_d[i] = 1.0f; }
  }
}

}  // extern "C"
