// Headless harness: one frame in the reference's call order (source/kinect_client.cpp:569-599,614) through the C++
// adapter, on a hand-made single-stream scene (constant inverse LUT, constant images).  Exit codes: 0 ok,
// 3 no HIP device (the path has no CPU fallback), 1 any other failure.
//   g++ -std=c++17 frame_harness.cpp -o frame_harness -L.. -lrgbd_recon_hip -Wl,-rpath,'$ORIGIN/..'
#include <cmath>
#include <cstdio>
#include <vector>

#include "recon_integration_hip.hpp"

int main() {
  kinect::ReconInputs in;
  in.num_kinects = 1;
  in.depth_width = in.color_width = 8;
  in.depth_height = in.color_height = 8;
  in.bbox_min = {{0.0f, 0.0f, 0.0f}};
  in.bbox_max = {{1.0f, 1.0f, 1.0f}};
  in.explicit_res = {{16, 16, 16}};
  try {
    kinect::ReconIntegrationHip recon(in, /*limit*/ 0.05f, /*voxel*/ 0.0625f, /*brick*/ 0.5f, 32, 32);
    // every voxel projects to (u, v, z) = (0.5, 0.5, 0.52); the depth image says 0.5 -> sdist 0.02 everywhere
    const uint32_t r2[3] = {2, 2, 2};
    std::vector<float> inv(8 * 4), uv(8 * 2), xyz(8 * 3);
    for (int i = 0; i < 8; ++i) {
      inv[4 * i] = 0.5f; inv[4 * i + 1] = 0.5f; inv[4 * i + 2] = 0.52f; inv[4 * i + 3] = 1.0f;
      uv[2 * i] = 0.5f; uv[2 * i + 1] = 0.5f;
      xyz[3 * i] = 0.25f + 0.5f * (i & 1); xyz[3 * i + 1] = 0.25f + 0.5f * ((i >> 1) & 1); xyz[3 * i + 2] = 0.25f + 0.5f * (i >> 2);
    }
    recon.setCalibration(0, inv.data(), r2, uv.data(), r2, xyz.data(), r2);
    std::vector<float> depth(64 * 2, 0.0f), q(64, 1.0f), s(64, 1.0f);
    for (int i = 0; i < 64; ++i) depth[2 * i] = 0.5f;
    std::vector<uint8_t> col(64 * 3, 200);
    recon.uploadFrame(depth.data(), q.data(), s.data(), col.data());
    recon.setMinVoxelsPerBrick(1);
    recon.clearOccupiedBricks();
    recon.markBricks();
    recon.updateOccupiedBricks();
    recon.integrate();
    const float mv[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -0.5f, -0.5f, -3.0f, 1};
    const float f = 1.0f / std::tan(0.5f * 0.6f), n = 0.1f, fa = 50.0f;
    const float pr[16] = {f, 0, 0, 0, 0, f, 0, 0, 0, 0, (fa + n) / (n - fa), -1, 0, 0, 2 * fa * n / (n - fa), 0};
    recon.setMatrices(mv, pr);
    recon.drawF();
    std::vector<float> tsdf, rgba, d;
    recon.downloadVolume(tsdf);
    recon.downloadFramebuffer(rgba, d, 32, 32);
    int band = 0;
    for (float v : tsdf) band += std::fabs(v - 0.02f) < 1e-6f;
    std::printf("occupied ratio %.4f, %u bricks, %d of %zu voxels at sdist 0.02\n", recon.occupiedRatio(), recon.numBricks(), band, tsdf.size());
    return band > 0 ? 0 : 1;
  } catch (std::exception const& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return std::string(e.what()).find("no HIP device") != std::string::npos ? 3 : 1;
  }
}
