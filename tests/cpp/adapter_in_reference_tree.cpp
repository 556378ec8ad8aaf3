// Compile-only check (tests/test_host_adapter.py): the Reconstruction-derived adapter against the reference's OWN headers, used the way
// source/kinect_client.cpp uses its back-ends (:120, :249-253, :571-576, :595-599, :614, :624, :651).  Never linked, never run.
#include "recon_integration_hip_gl.hpp"
struct NullBridge : kinect::GlBridge {
  void modelview(float m[16]) override { for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0); }
  void projection(float m[16]) override { for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0); }
  void viewport(int v[4]) override { v[0] = v[1] = 0; v[2] = 1280; v[3] = 720; }
  void present(const float*, const float*, unsigned, unsigned) override {}
};
// source/kinect_client.cpp:120,249-253
std::vector<std::shared_ptr<kinect::Reconstruction>> g_recons;
void init(kinect::CalibrationFiles const& cfs, kinect::CalibVolumes const* cv, gloost::BoundingBox const& bbox, NullBridge& gl) {
  auto hip = std::make_shared<kinect::ReconIntegrationHipGL>(cfs, cv, bbox, 0.01f, 0.01f, gl);
  g_recons.emplace_back(hip);
  g_recons.back()->setColorMaskMode(1);              // :624
  g_recons.back()->setViewportOffset(640.0f, 0.0f);  // :651
  g_recons.back()->drawF();                          // :614
  hip->clearOccupiedBricks(); hip->updateOccupiedBricks(); hip->integrate();   // :571-576, :595-599
}
