// recon_integration_hip_gl.hpp -- the HIP operator AS a kinect::Reconstruction, for compilation INSIDE the reference tree.
//
// kinect_client.cpp keeps its render back-ends in `std::vector<std::shared_ptr<kinect::Reconstruction>> g_recons`
// (source/kinect_client.cpp:120, filled :249-253) and draws `g_recons.at(g_recon_mode)->drawF()` (:614, :625-665), calling
// setColorMaskMode / setViewportOffset on the same pointers in the stereo modes (:624,:631,:651,:659).  This class derives from
// the reference's own abstract base (framework/reconstruction/reconstruction.hpp:11-36), overrides its virtuals and carries the
// public surface of kinect::ReconIntegration (recon_integration.hpp:38-64), so that
//
//     g_recon_integration = std::make_shared<kinect::ReconIntegrationHipGL>(*g_calib_files, g_cv.get(), g_bbox, g_tsdf_limit, g_voxel_size, gl);
//     g_recons.emplace_back(g_recon_integration);                       // was: new kinect::ReconIntegration(...), :250-251
//
// compiles and runs with the rest of draw3d() untouched.  The only GL it needs is the state ReconIntegration::draw() reads with
// glGet* (recon_integration.cpp:183,190,197) and a way to put the finished image into the bound framebuffer; both go through the
// four-function GlBridge below, so this header itself includes no GL header (INTEGRATION.md section 2 has the 12-line bridge).
// Include paths: framework/reconstruction (reconstruction.hpp), framework/calibration (calibration_files.hpp), external/gloost.
#ifndef RECON_INTEGRATION_HIP_GL_HPP
#define RECON_INTEGRATION_HIP_GL_HPP

#include <memory>
#include <vector>

#include "reconstruction.hpp"        // kinect::Reconstruction           (reference, framework/reconstruction/reconstruction.hpp)
#include "calibration_files.hpp"     // kinect::CalibrationFiles         (reference, framework/calibration/calibration_files.hpp)

#include "recon_integration_hip.hpp"

namespace kinect {

// what draw() takes from / gives to the GL context of the calling thread
struct GlBridge {
  virtual ~GlBridge() {}
  virtual void modelview(float m[16]) = 0;     // glGetFloatv(GL_MODELVIEW_MATRIX, m)    recon_integration.cpp:197
  virtual void projection(float m[16]) = 0;    // glGetFloatv(GL_PROJECTION_MATRIX, m)   recon_integration.cpp:183
  virtual void viewport(int v[4]) = 0;         // glGetIntegerv(GL_VIEWPORT, v)          reconstruction.cpp:45-49
  // colour RGBA32F + window-space depth of the w x h viewport, bottom row first: into the bound draw framebuffer at the viewport
  virtual void present(const float* rgba, const float* depth, unsigned w, unsigned h) = 0;
};

class ReconIntegrationHipGL : public Reconstruction {
 public:
  // ReconIntegration(cfs, cv, bbox, limit, size), recon_integration.cpp:30-60.  The calibration volumes and the frame images are
  // GL textures in the reference; here they are handed over once / per frame with setCalibration() / uploadFrame() of impl().
  ReconIntegrationHipGL(CalibrationFiles const& cfs, CalibVolumes const* cv, gloost::BoundingBox const& bbox, float limit, float size, GlBridge& gl,
                        float brick_size = 0.1f, std::size_t width = 1280, std::size_t height = 720, int device = 0)
      : Reconstruction(cfs, cv, bbox), m_gl(gl), m_impl(inputs(cfs, bbox, device), limit, size, brick_size, width, height), m_w((unsigned)width), m_h((unsigned)height) {}

  // ---- kinect::Reconstruction virtuals (reconstruction.hpp:16-23)
  void draw() override { render(false); }
  void drawF() override { render(true); }                                 // drawDepthLimits + draw + fillColors, recon_integration.cpp:151-174
  void reload() override {}                                               // shader hot reload (kinect_client.cpp:776-783): no shaders here
  void resize(std::size_t width, std::size_t height) override { m_impl.resize(width, height); m_w = (unsigned)width; m_h = (unsigned)height; }
  void setViewportOffset(float x, float y) override { check(tsdf_set_viewport_offset(m_impl.handle(), x, y)); }   // recon_integration.cpp:527
  // setColorMaskMode() is not virtual in the base: it stores m_color_mask_mode (reconstruction.cpp:51-53), read in render() like the
  // reference reads it in draw() / fillColors() (recon_integration.cpp:212-216,321-333)

  // ---- kinect::ReconIntegration (recon_integration.hpp:42-58), same names
  void integrate() { m_impl.integrate(); }
  void setColorFilling(bool active) { m_impl.setColorFilling(active); }
  void setUseBricks(bool active) { m_impl.setUseBricks(active); }
  void setSpaceSkip(bool active) { m_impl.setSpaceSkip(active); }
  void setDrawBricks(bool active) { m_impl.setDrawBricks(active); }
  void setVoxelSize(float size) { m_impl.setVoxelSize(size); }
  void setTsdfLimit(float limit) { m_impl.setTsdfLimit(limit); }
  void setBrickSize(float size) { m_impl.setBrickSize(size); }
  void setMinVoxelsPerBrick(unsigned i) { m_impl.setMinVoxelsPerBrick(i); }
  float occupiedRatio() const { return m_impl.occupiedRatio(); }
  float getBrickSize() const { return m_impl.getBrickSize(); }
  void clearOccupiedBricks() const { m_impl.clearOccupiedBricks(); }
  void updateOccupiedBricks() { m_impl.updateOccupiedBricks(); }
  void drawOccupiedBricks() const {}                                      // wireframe debug overlay (solid.vs/fs): not part of the path
  // the frame's explicit inputs and everything else of the C ABI
  ReconIntegrationHip& impl() { return m_impl; }

 private:
  static ReconInputs inputs(CalibrationFiles const& cfs, gloost::BoundingBox const& bbox, int device) {
    ReconInputs in;
    in.num_kinects = cfs.num();                                           // reconstruction.cpp:18-20
    in.depth_width = cfs.getWidth(); in.depth_height = cfs.getHeight();
    in.color_width = cfs.getWidthC(); in.color_height = cfs.getHeightC();
    for (int a = 0; a < 3; ++a) { in.bbox_min[a] = bbox.getPMin()[a]; in.bbox_max[a] = bbox.getPMax()[a]; }
    in.device = device;
    return in;
  }
  void render(bool with_fill) {
    float mv[16], pr[16];
    int vp[4];
    m_gl.modelview(mv); m_gl.projection(pr); m_gl.viewport(vp);
    if ((unsigned)vp[2] != m_w || (unsigned)vp[3] != m_h) resize((std::size_t)vp[2], (std::size_t)vp[3]);   // img_to_eye uses GL_VIEWPORT's size, :190
    check(tsdf_set_viewport_origin(m_impl.handle(), vp[0], vp[1]));       // gl_FragCoord's window origin
    check(tsdf_set_color_mask_mode(m_impl.handle(), m_color_mask_mode));
    m_impl.setMatrices(mv, pr);
    if (with_fill) m_impl.drawF(); else m_impl.draw();
    m_impl.downloadFramebuffer(m_rgba, m_depth, m_w, m_h);
    m_gl.present(m_rgba.data(), m_depth.data(), m_w, m_h);
  }
  void check(int32_t rc) const { if (rc != TSDF_OK) throw std::runtime_error(std::string("ReconIntegrationHipGL: ") + tsdf_last_error(m_impl.handle())); }

  GlBridge& m_gl;
  mutable ReconIntegrationHip m_impl;
  unsigned m_w, m_h;
  std::vector<float> m_rgba, m_depth;
};

}  // namespace kinect

#endif  // RECON_INTEGRATION_HIP_GL_HPP
