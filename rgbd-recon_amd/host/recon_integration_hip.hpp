// recon_integration_hip.hpp -- C++ adapter with the reference operator's names, over the C ABI.
//
// kinect::ReconIntegrationHip exposes the public surface of kinect::ReconIntegration
// (framework/reconstruction/recon_integration.hpp:35-64) and of its base kinect::Reconstruction
// (framework/reconstruction/reconstruction.hpp:11-36), so draw3d()-style code
// (source/kinect_client.cpp:569-599,614) compiles against either class:
//
//     recon->clearOccupiedBricks();            // process_textures(), kinect_client.cpp:569-577
//     nka->processTextures();                  //   (the GL pre-process; on the HIP path: markBricks())
//     recon->updateOccupiedBricks();
//     recon->integrate();                      // :595-599
//     recon->drawF();                          // :614
//
// What the GL class pulls out of global GL state is passed explicitly:
//   * CalibVolumes' 3-D textures          -> setCalibration(stream, cv_xyz_inv, cv_uv, cv_xyz)
//   * NetKinectArray's texture arrays     -> uploadFrame(depth, quality, silhouette, colour)
//   * GL_MODELVIEW / GL_PROJECTION        -> setMatrices(mv, proj) before draw()/drawF()
//   * UBO 1 g_shade_mode                  -> setShadeMode()
// Errors: the reference throws / exits; this class throws std::runtime_error carrying tsdf_last_error().
// Header only; link with librgbd_recon_hip.so.  No GL, no torch.
#ifndef RECON_INTEGRATION_HIP_HPP
#define RECON_INTEGRATION_HIP_HPP

#include <array>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rgbd_recon_hip.h"

namespace kinect {

// the pieces of CalibrationFiles (calibration_files.hpp) and gloost::BoundingBox the operator reads
struct ReconInputs {
  unsigned num_kinects = 0;                 // CalibrationFiles::num()
  unsigned depth_width = 640, depth_height = 480;     // getWidth()/getHeight()
  unsigned color_width = 640, color_height = 480;     // getWidthC()/getHeightC()
  std::array<float, 3> bbox_min{{-1.0f, 0.0f, -1.0f}};   // kinect_client.cpp:206-207 default
  std::array<float, 3> bbox_max{{1.0f, 2.2f, 1.0f}};
  int device = 0;
  std::array<unsigned, 3> explicit_res{{0, 0, 0}};    // 0: res = ceil(bbox / voxel_size) like setVoxelSize()
  unsigned slab_z0 = 0, slab_z1 = 0;                  // multi-GPU Z-slab (0,0 = whole volume)
  bool slab_recompute_halo = false;                   // integrate the halo layers locally instead of exchanging them
  unsigned sparse_pool_tiles = 0;                     // > 0: TSDF in a sparse pool of this many 8^3-voxel tiles (needs brick culling)
};

class ReconIntegrationHip {
 public:
  // ReconIntegration(cfs, cv, bbox, limit, size), recon_integration.cpp:30-60
  ReconIntegrationHip(ReconInputs const& in, float limit, float voxel_size, float brick_size = 0.1f, std::size_t width = 1280, std::size_t height = 720)
      : m_in(in), m_brick_size(brick_size) {
    tsdf_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    for (int a = 0; a < 3; ++a) { cfg.bbox_min[a] = in.bbox_min[a]; cfg.bbox_max[a] = in.bbox_max[a]; cfg.res[a] = in.explicit_res[a]; cfg.brick_size[a] = brick_size; }
    cfg.voxel_size = voxel_size;
    cfg.limit = limit;
    cfg.num_streams = in.num_kinects;
    cfg.depth_w = in.depth_width; cfg.depth_h = in.depth_height;
    cfg.color_w = in.color_width; cfg.color_h = in.color_height;
    cfg.view_w = (uint32_t)width; cfg.view_h = (uint32_t)height;
    cfg.device = in.device;
    cfg.slab_z0 = in.slab_z0; cfg.slab_z1 = in.slab_z1;
    cfg.slab_recompute_halo = in.slab_recompute_halo ? 1u : 0u;
    cfg.sparse_pool_tiles = in.sparse_pool_tiles;
    if (tsdf_create(&cfg, &m_ctx) != TSDF_OK) throw std::runtime_error(std::string("ReconIntegrationHip: ") + tsdf_last_error(nullptr));
    const float id[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(m_mv, id, sizeof(id));
    std::memcpy(m_proj, id, sizeof(id));
  }
  ~ReconIntegrationHip() { if (m_ctx) tsdf_destroy(m_ctx); }
  ReconIntegrationHip(ReconIntegrationHip const&) = delete;
  ReconIntegrationHip& operator=(ReconIntegrationHip const&) = delete;

  // ---- explicit inputs (implicit GL state in the reference)
  void setCalibration(unsigned stream, const float* cv_xyz_inv_rgba, const uint32_t res_inv[3], const float* cv_uv_rg, const uint32_t res_uv[3],
                      const float* cv_xyz_rgb, const uint32_t res_xyz[3]) {
    check(tsdf_set_calibration(m_ctx, stream, cv_xyz_inv_rgba, res_inv, cv_uv_rg, res_uv, cv_xyz_rgb, res_xyz));
  }
  void uploadFrame(const float* depth_rg, const float* quality, const float* silhouette, const uint8_t* colour_rgb) {
    check(tsdf_upload_frame(m_ctx, depth_rg, quality, silhouette, colour_rgb));
  }
  // a frame that is already in device memory (a decoder / pre-process on the GPU): one re-layout launch, no copy
  void uploadFrameDev(const float* depth_rg, const float* quality, const float* silhouette, const uint8_t* colour_rgb, bool arrays_complete) {
    check(tsdf_upload_frame_dev(m_ctx, depth_rg, quality, silhouette, colour_rgb, arrays_complete ? TSDF_FRAME_ARRAYS_COMPLETE : 0u));
  }
  // the client's per-frame sequence (kinect_client.cpp:586-599 update + draw: new frame -> clear / mark / update bricks -> integrate() -> drawF())
  // in ONE call into the library: issuing a frame costs the host about as long as the GPU needs for it, every call boundary counts
  void frameDev(const float* depth_rg, const float* quality, const float* silhouette, const uint8_t* colour_rgb) {
    check(tsdf_frame_dev(m_ctx, depth_rg, quality, silhouette, colour_rgb, TSDF_FRAME_ARRAYS_COMPLETE, m_mv, m_proj));
  }
  void setMatrices(const float modelview[16], const float projection[16]) {
    std::memcpy(m_mv, modelview, sizeof(m_mv));
    std::memcpy(m_proj, projection, sizeof(m_proj));
  }
  void markBricks() { check(tsdf_mark_bricks(m_ctx)); }          // mark_brick() runs inside pre_normal.fs in the reference

  // ---- kinect::Reconstruction (reconstruction.hpp:16-23)
  void draw() { check(tsdf_raymarch(m_ctx, m_mv, m_proj)); }
  void drawF() { check(tsdf_draw_f(m_ctx, m_mv, m_proj)); }
  void reload() {}                                               // shader hot reload: nothing to reload
  void resize(std::size_t width, std::size_t height) { check(tsdf_resize(m_ctx, (uint32_t)width, (uint32_t)height)); }
  void setColorMaskMode(unsigned mode) { check(tsdf_set_color_mask_mode(m_ctx, mode)); }       // reconstruction.cpp:51-53 -> glColorMask :212-216,321-333
  void setViewportOffset(float x, float y) { check(tsdf_set_viewport_offset(m_ctx, x, y)); }   // recon_integration.cpp:527 -> tsdf_raymarch.fs:70,388-389
  // GL state the reference reads implicitly: glViewport's origin (gl_FragCoord) and whether glClear included the colour buffer
  void setViewportOrigin(int x, int y) { check(tsdf_set_viewport_origin(m_ctx, x, y)); }
  void setFramebufferClear(bool clear_color) { check(tsdf_set_framebuffer_clear(m_ctx, clear_color ? 1 : 0)); }

  // ---- kinect::ReconIntegration (recon_integration.hpp:42-58)
  void integrate() { check(tsdf_integrate(m_ctx)); }
  void setColorFilling(bool active) { check(tsdf_set_color_filling(m_ctx, active)); }
  void setUseBricks(bool active) { check(tsdf_set_use_bricks(m_ctx, active)); }
  void setSpaceSkip(bool active) { check(tsdf_set_space_skip(m_ctx, active)); }
  void setDrawBricks(bool active) { m_draw_bricks = active; }    // wireframe debug overlay: not part of the HIP path
  void setVoxelSize(float size) { check(tsdf_set_voxel_size(m_ctx, size)); }      // recon_integration.cpp:340-353
  void setTsdfLimit(float limit) { check(tsdf_set_tsdf_limit(m_ctx, limit)); }
  void setBrickSize(float size) { const float s[3] = {size, size, size}; check(tsdf_set_brick_size(m_ctx, s)); m_brick_size = size; }
  unsigned numBricks() const { uint32_t n = 0; tsdf_num_bricks(m_ctx, &n); return n; }
  float occupiedRatio() const { float r = 0.0f; tsdf_occupied_ratio(m_ctx, &r); return r; }
  float getBrickSize() const { float s[3]; tsdf_get_resolution(m_ctx, nullptr, nullptr, s); return s[0]; }
  void clearOccupiedBricks() const { check(tsdf_clear_bricks(m_ctx)); }
  void updateOccupiedBricks() { check(tsdf_update_occupied(m_ctx, nullptr)); }
  void setMinVoxelsPerBrick(unsigned i) { check(tsdf_set_min_voxels_per_brick(m_ctx, i)); }
  void drawOccupiedBricks() const {}
  void setShadeMode(int mode) { check(tsdf_set_shade_mode(m_ctx, mode)); }

  // ---- results (the GL class leaves them in textures / the bound framebuffer)
  void downloadFramebuffer(std::vector<float>& rgba, std::vector<float>& depth, unsigned width, unsigned height) {
    rgba.resize((std::size_t)width * height * 4);
    depth.resize((std::size_t)width * height);
    check(tsdf_download_framebuffer(m_ctx, rgba.data(), depth.data()));
  }
  void downloadVolume(std::vector<float>& tsdf) {
    uint32_t r[3];
    check(tsdf_get_resolution(m_ctx, r, nullptr, nullptr));
    tsdf.resize((std::size_t)r[0] * r[1] * r[2]);
    check(tsdf_download_volume(m_ctx, tsdf.data()));
  }
  // ---- kinect::ReconPoints::draw() (recon_points.cpp:71-111) on the same inputs: the point back-end for A/B comparison
  void uploadNormals(const float* normals_rgb) { check(tsdf_upload_normals(m_ctx, normals_rgb)); }
  void drawPoints() { check(tsdf_draw_points(m_ctx, m_mv, m_proj)); }
  // ---- kinect::ReconTrigrid::draw() (recon_trigrid.cpp:85-148)
  void setMinLength(float v) { check(tsdf_set_min_length(m_ctx, v)); }
  void drawTrigrid() { check(tsdf_draw_trigrid(m_ctx, m_mv, m_proj)); }
  tsdf_ctx* handle() const { return m_ctx; }

 private:
  void check(int32_t rc) const { if (rc != TSDF_OK) throw std::runtime_error(std::string("ReconIntegrationHip: ") + tsdf_last_error(m_ctx)); }
  ReconInputs m_in;
  tsdf_ctx* m_ctx = nullptr;
  float m_mv[16], m_proj[16];
  float m_brick_size;
  bool m_draw_bricks = false;
};

// The input side: kinect::NetKinectArray's public surface (framework/NetKinectArray.h:40-55) over the SAME context, for callers
// that also replace the GL upload / pre-process.  The ZMQ socket stays with the caller: where readLoop() memcpy's the received
// message into the PBOs (NetKinectArray.cpp:513-523), hand it to submit().
class NetKinectArrayHip {
 public:
  // colour_format: CalibrationFiles::isCompressedRGB() (0 RGB8 / 1 DXT1 / 5 DXT5); compressed_depth: isCompressedDepth()
  NetKinectArrayHip(ReconIntegrationHip& recon, unsigned colour_format, bool compressed_depth) : m_ctx(recon.handle()) {
    check(tsdf_set_wire_format(m_ctx, colour_format, compressed_depth ? TSDF_DEPTH_U8 : TSDF_DEPTH_F32));
  }
  // per sensor: KinectCalibrationFile::isCompressedDepth()/getNear()/getFar(), CalibVolumes::getDepthLimits(i)
  void setSensor(unsigned i, bool compressed_depth, float near_m, float far_m, float cv_min_d, float cv_max_d) {
    check(tsdf_set_depth_compression(m_ctx, i, compressed_depth, near_m, far_m));
    check(tsdf_set_depth_limits(m_ctx, i, cv_min_d, cv_max_d));
  }
  std::size_t messageBytes() const { uint64_t n = 0; tsdf_wire_sizes(m_ctx, nullptr, nullptr, &n); return (std::size_t)n; }
  void submit(const void* zmq_message, std::size_t bytes) { check(tsdf_upload_wire_frame(m_ctx, zmq_message, bytes, &m_frametime)); m_dirty = true; }   // readLoop(), :482-529
  bool update() { const bool fresh = m_dirty; m_dirty = false; return fresh; }      // NetKinectArray.cpp:225-236: "a new frame was uploaded"
  void processTextures() { check(tsdf_process_textures(m_ctx)); }                    // :309-426 (marks the bricks itself)
  void filterTextures(bool f) { m_filter = f; apply(); }                             // :463-476: each setter re-runs the passes
  void useProcessedDepths(bool f) { m_processed = f; apply(); }
  void refineBoundary(bool f) { m_refine = f; apply(); }
  double frameTime() const { return m_frametime; }

 private:
  void apply() { check(tsdf_set_preprocess(m_ctx, m_filter, m_processed, m_refine)); }
  void check(int32_t rc) const { if (rc != TSDF_OK) throw std::runtime_error(std::string("NetKinectArrayHip: ") + tsdf_last_error(m_ctx)); }
  tsdf_ctx* m_ctx;
  bool m_dirty = false, m_filter = true, m_processed = true, m_refine = true;
  double m_frametime = 0.0;
};

}  // namespace kinect

#endif  // RECON_INTEGRATION_HIP_HPP
