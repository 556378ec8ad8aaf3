// calib_inverter -- the offline tool of source/calib_inverter.cpp on the HIP path.
//
//   calib_inverter <file.ks> [-s voxel_size]
//
// Like the reference (source/calib_inverter.cpp:12-74): read the `kinect <calib.yml>` and `bbx` tokens of the .ks file,
// load <calib>.cv_xyz for every sensor, build the inverse volume at res = ceil(bbox / voxel_size) (default 0.007 m) and
// write <calib>.cv_xyz_inv next to the .ks file (CalibrationInverter::writeInverseVolumes, calibration_inverter.cpp:31-38).
// Links only against the C ABI (include/rgbd_recon_hip.h).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/rgbd_recon_hip.h"

int main(int argc, char** argv) {
  std::string ks;
  float voxel_size = 0.007f;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "-s") && i + 1 < argc) voxel_size = (float)atof(argv[++i]);
    else ks = argv[i];
  }
  const size_t dot = ks.find_last_of('.');
  if (ks.empty() || dot == std::string::npos || ks.substr(dot + 1) != "ks") { fprintf(stderr, "No .ks file specified\n"); return 1; }
  const size_t slash = ks.find_last_of("/\\");
  const std::string resource_path = slash == std::string::npos ? std::string("./") : ks.substr(0, slash) + '/';

  std::vector<std::string> calibs;
  float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};
  std::ifstream in(ks);
  if (!in) { fprintf(stderr, "cannot open %s\n", ks.c_str()); return 1; }
  std::string token;
  while (in >> token) {
    if (token == "kinect") {
      in >> token;
      calibs.push_back((token[0] == '/' || (token.size() > 1 && token[1] == ':')) ? token : resource_path + token);
    } else if (token == "bbx") {
      in >> bmin[0] >> bmin[1] >> bmin[2] >> bmax[0] >> bmax[1] >> bmax[2];
    }
  }
  uint32_t res_inv[3];
  if (tsdf_inverse_volume_resolution(bmin, bmax, voxel_size, res_inv) != TSDF_OK) { fprintf(stderr, "bad bounding box / voxel size\n"); return 1; }
  printf("using resolution %u, %u, %u\n", res_inv[0], res_inv[1], res_inv[2]);

  for (const std::string& yml : calibs) {
    const std::string xyz_name = yml.substr(0, yml.size() - 3) + "cv_xyz";          // "<base>.yml" -> "<base>.cv_xyz", calibration_inverter.cpp:18-21
    uint32_t res[3]; float limits[2];
    if (tsdf_calib_volume_info(xyz_name.c_str(), 3, res, limits) != TSDF_OK) { fprintf(stderr, "%s\n", tsdf_calib_last_error()); return 2; }
    printf("loading %s\ndimensions xyz - %u, %u, %u minmax d - %g, %g\n", xyz_name.c_str(), res[0], res[1], res[2], limits[0], limits[1]);
    std::vector<float> xyz((size_t)res[0] * res[1] * res[2] * 3), inv((size_t)res_inv[0] * res_inv[1] * res_inv[2] * 4);
    if (tsdf_calib_volume_read(xyz_name.c_str(), 3, xyz.data(), xyz.size()) != TSDF_OK) { fprintf(stderr, "%s\n", tsdf_calib_last_error()); return 2; }
    float ms = 0.0f;
    const int32_t rc = tsdf_invert_calibration(0, xyz.data(), res, bmin, bmax, res_inv, inv.data(), &ms);
    if (rc != TSDF_OK) { fprintf(stderr, "inversion failed (%d): %s\n", rc, tsdf_calib_last_error()); return 3; }
    const size_t s2 = xyz_name.find_last_of("/\\");
    const std::string out = resource_path + (s2 == std::string::npos ? xyz_name : xyz_name.substr(s2 + 1)) + "_inv";
    const float out_limits[2] = {0.5f, 4.5f};                                       // calibration_inverter.cpp:112
    printf("inverted in %.2f ms; writing to file %s\n", ms, out.c_str());
    if (tsdf_calib_volume_write(out.c_str(), 4, res_inv, out_limits, inv.data()) != TSDF_OK) { fprintf(stderr, "%s\n", tsdf_calib_last_error()); return 2; }
  }
  return 0;
}
