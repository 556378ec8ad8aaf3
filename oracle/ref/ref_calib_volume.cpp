// oracle/ref/ref_calib_volume.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin command-line driver around the REFERENCE's own calibration volume reader/writer:
// it #includes /root/reference/framework/calibration/calibration_volume.hpp and framework/DataTypes.h where
// they lie (header-only, only the vendored glm is needed) -- nothing of the reference is copied into this repository.
// The resulting binary (oracle/_ref/ref_calib_volume, git-ignored) pins this project's .cv_xyz / .cv_uv / .cv_xyz_inv
// file I/O against the real thing:
//   write <xyz|uv|inv> <file> <rx> <ry> <rz> <dmin> <dmax> <seed>   writes a volume of deterministic pseudo-random texels
//   read  <xyz|uv|inv> <file>                                       prints "rx ry rz dmin dmax fnv1a64(payload) first last"
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <glm/glm.hpp>
#include <calibration_volume.hpp>   // -I/root/reference/framework/calibration
#include <DataTypes.h>              // -I/root/reference/framework

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
static float rnd(uint32_t& s) { return (float)(lcg(s) >> 8) * (1.0f / 16777216.0f) * 4.0f - 2.0f; }   // [-2, 2)
static uint64_t fnv(const void* p, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) { h ^= ((const unsigned char*)p)[i]; h *= 1099511628211ull; }
  return h;
}

template <typename T, int C>
static int run(const std::string& mode, int argc, char** argv) {
  static_assert(sizeof(T) == C * sizeof(float), "texel layout");
  if (mode == "write") {
    if (argc != 10) return 2;
    const glm::uvec3 res{(unsigned)atoi(argv[4]), (unsigned)atoi(argv[5]), (unsigned)atoi(argv[6])};
    const glm::fvec2 lim{(float)atof(argv[7]), (float)atof(argv[8])};
    uint32_t seed = (uint32_t)strtoul(argv[9], nullptr, 10);
    std::vector<T> vol((size_t)res.x * res.y * res.z);
    for (auto& t : vol) for (int c = 0; c < C; ++c) ((float*)&t)[c] = rnd(seed);
    kinect::CalibrationVolume<T>(res, lim, vol).write(argv[3]);
    return 0;
  }
  kinect::CalibrationVolume<T> v{std::string(argv[3])};
  const float* f = (const float*)v.volume().data();
  const size_t n = v.volume().size() * C;
  printf("%u %u %u %.9g %.9g %016llx %.9g %.9g\n", v.res().x, v.res().y, v.res().z, v.depthLimits().x, v.depthLimits().y,
         (unsigned long long)fnv(f, n * sizeof(float)), f[0], f[n - 1]);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s write|read xyz|uv|inv file [rx ry rz dmin dmax seed]\n", argv[0]); return 2; }
  const std::string mode = argv[1], type = argv[2];
  if (type == "xyz") return run<kinect::xyz, 3>(mode, argc, argv);     // CalibVolumes.cpp:118 CalibrationVolume<xyz>
  if (type == "uv") return run<kinect::uv, 2>(mode, argc, argv);       // :125 CalibrationVolume<uv>
  if (type == "inv") return run<glm::fvec4, 4>(mode, argc, argv);      // :68  CalibrationVolume<glm::fvec4>
  return 2;
}
