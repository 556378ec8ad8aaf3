// Inverse calibration volumes: host side of k_inverter.hip.
//   kinect::Frustum                       framework/calibration/frustum.cpp
//   CalibrationInverter                   framework/calibration/calibration_inverter.cpp:68-133
//   calib_inverter's resolution rule      source/calib_inverter.cpp:60-63
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>

#include "tsdf_common.hpp"

void rr_set_io_error(const std::string& m);   // file_io.cpp (tsdf_calib_last_error)

namespace {
using rr::InverterGrid;
using rr::InverterQuery;

struct V3 { float x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                        // glm: x + y + z, left to right
inline V3 cross3(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline V3 unit(V3 v) { const float s = 1.0f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); return v * s; }   // glm::normalize = v * inversesqrt
inline V3 mean4(V3 a, V3 b, V3 c, V3 d) { return (a + b + c + d) / 4.0f; }

struct FrustumData { V3 corner[8]; float plane[6][4]; };

// getCornerPoints (calibration_inverter.cpp:117-133) + getPlanes / getSideNormals / getSideCenters / getEdgeCenters (frustum.cpp)
FrustumData frustum_of(const float* xyz, const uint32_t res[3]) {
  FrustumData f;
  const uint32_t ex = res[0] - 1, ey = res[1] - 1, ez = res[2] - 1;
  auto texel = [&](uint32_t x, uint32_t y, uint32_t z) { const float* t = xyz + 3 * (((size_t)z * res[1] + y) * res[0] + x); return V3{t[0], t[1], t[2]}; };
  const uint32_t cx[4] = {0, 0, ex, ex}, cy[4] = {0, ey, ey, 0};
  for (int i = 0; i < 4; ++i) { f.corner[i] = texel(cx[i], cy[i], 0); f.corner[4 + i] = texel(cx[i], cy[i], ez); }
  const V3* c = f.corner;
  V3 edge[12];
  for (int i = 0; i < 4; ++i) {
    const int j = (i + 1) % 4;
    edge[i] = (c[i] + c[j]) * 0.5f; edge[4 + i] = (c[4 + i] + c[4 + j]) * 0.5f; edge[8 + i] = (c[i] + c[4 + i]) * 0.5f;
  }
  const V3 side[6] = {mean4(c[0], c[1], c[2], c[3]), mean4(c[4], c[5], c[6], c[7]), mean4(c[0], c[1], c[4], c[5]),
                      mean4(c[2], c[3], c[6], c[7]), mean4(c[1], c[2], c[5], c[6]), mean4(c[0], c[3], c[4], c[7])};   // near far left right top bottom
  const V3 nrm[6] = {unit(cross3(edge[0] - edge[2], edge[3] - edge[2])), unit(cross3(edge[4] - edge[6], edge[5] - edge[7])),
                     unit(cross3(edge[0] - edge[4], edge[9] - edge[8])), unit(cross3(edge[2] - edge[6], edge[11] - edge[10])),
                     unit(cross3(edge[9] - edge[10], edge[1] - edge[5])), unit(cross3(edge[8] - edge[11], edge[7] - edge[3]))};
  for (int i = 0; i < 6; ++i) { f.plane[i][0] = nrm[i].x; f.plane[i][1] = nrm[i].y; f.plane[i][2] = nrm[i].z; f.plane[i][3] = -dot3(nrm[i], side[i]); }
  return f;
}
V3 closest_between(V3 p, V3 u, V3 q, V3 v) {                              // closestPoint(), frustum.cpp
  const V3 w0 = p - q;
  const float a = dot3(u, u), b = dot3(u, v), c = dot3(v, v), d = dot3(u, w0), e = dot3(v, w0);
  const float sc = (b * e - c * d) / (a * c - b * b), tc = (a * e - b * d) / (a * c - b * b);
  return ((p + u * sc) + (q + v * tc)) * 0.5f;
}
int32_t fail(const std::string& m) { rr_set_io_error(m); return TSDF_ERR_INVALID_ARGUMENT; }
int32_t fail_hip(hipError_t e, const char* what) { rr_set_io_error(std::string(what) + ": " + hipGetErrorString(e)); return e == hipErrorOutOfMemory ? TSDF_ERR_OUT_OF_MEMORY : TSDF_ERR_HIP; }
bool bad_res(const uint32_t r[3]) { return !r || r[0] < 1 || r[1] < 1 || r[2] < 1 || r[0] > 2048 || r[1] > 2048 || r[2] > 2048; }
}  // namespace

extern "C" {

int32_t tsdf_frustum_from_volume(const float* cv_xyz, const uint32_t res[3], float planes[24], float camera_pos[3]) {
  if (!cv_xyz || bad_res(res)) return fail("bad forward volume");
  const FrustumData f = frustum_of(cv_xyz, res);
  if (planes) memcpy(planes, f.plane, sizeof(f.plane));
  if (camera_pos) {                                                      // Frustum::getCameraPos -> CalibVolumes::getCameraPositions, CalibVolumes.cpp:224-230
    const V3* c = f.corner;
    const V3 cn = mean4(c[0], c[1], c[2], c[3]), cf = mean4(c[4], c[5], c[6], c[7]), dir = cf - cn;
    const V3 p = (closest_between(c[0], c[0] - c[4], cn, dir) + closest_between(c[1], c[1] - c[5], cn, dir) +
                  closest_between(c[2], c[2] - c[6], cn, dir) + closest_between(c[3], c[3] - c[7], cn, dir)) / 4.0f;
    camera_pos[0] = p.x; camera_pos[1] = p.y; camera_pos[2] = p.z;
  }
  return TSDF_OK;
}

int32_t tsdf_inverse_volume_resolution(const float bbox_min[3], const float bbox_max[3], float voxel_size, uint32_t res[3]) {
  if (!bbox_min || !bbox_max || !res || !(voxel_size > 0.0f)) return fail("bad argument");
  for (int a = 0; a < 3; ++a) res[a] = (uint32_t)std::ceil((bbox_max[a] - bbox_min[a]) / voxel_size);   // glm::uvec3{glm::ceil(dims / voxel_size)}
  return TSDF_OK;
}

int32_t tsdf_invert_calibration(int32_t device, const float* cv_xyz, const uint32_t res_xyz[3], const float bbox_min[3], const float bbox_max[3],
                                const uint32_t res_inv[3], float* cv_xyz_inv, float* gpu_ms) {
  if (!cv_xyz || !cv_xyz_inv || !bbox_min || !bbox_max || bad_res(res_xyz) || bad_res(res_inv)) return fail("bad argument");
  const uint64_t n64 = (uint64_t)res_xyz[0] * res_xyz[1] * res_xyz[2], nq = (uint64_t)res_inv[0] * res_inv[1] * res_inv[2];
  if (n64 > (1ull << 30) || nq > (1ull << 32)) return fail("volume too large");
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) return fail_hip(e, "hipSetDevice");

  InverterGrid G{};
  G.rx = res_xyz[0]; G.ry = res_xyz[1]; G.rz = res_xyz[2]; G.n = (uint32_t)n64;
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint64_t i = 0; i < n64; ++i)
    for (int a = 0; a < 3; ++a) {
      const double v = cv_xyz[3 * i + a];
      if (!std::isfinite(v)) return fail("forward volume holds a non-finite position");
      lo[a] = std::min(lo[a], v); hi[a] = std::max(hi[a], v);
    }
  // ~2 samples per cell on average, at most 256 cells per axis
  double ext[3], vol = 1.0;
  for (int a = 0; a < 3; ++a) { ext[a] = std::max(hi[a] - lo[a], 1e-6); vol *= ext[a]; }
  const double edge = std::cbrt(vol / std::max<double>(1.0, (double)n64 / 2.0));
  for (int a = 0; a < 3; ++a) {
    G.g[a] = (int)std::min(256.0, std::max(1.0, std::ceil(ext[a] / edge)));
    G.gmin[a] = lo[a]; G.cell[a] = ext[a] / G.g[a]; G.inv[a] = 1.0 / G.cell[a];
  }
  InverterQuery Q{};
  const FrustumData f = frustum_of(cv_xyz, res_xyz);
  memcpy(Q.plane, f.plane, sizeof(Q.plane));
  for (int a = 0; a < 3; ++a) {                                           // calibration_inverter.cpp:69-78
    Q.res[a] = res_inv[a];
    const float dim = bbox_max[a] - bbox_min[a], vstep = 1.0f / (float)res_inv[a];
    Q.step[a] = dim * vstep;
    Q.start[a] = bbox_min[a] + Q.step[a] * 0.5f;
  }

  const size_t cells = (size_t)G.g[0] * G.g[1] * G.g[2], nb = (cells + 1 + 2047) / 2048;
  float* d_xyz = nullptr; uint32_t *d_count = nullptr, *d_start = nullptr, *d_sums = nullptr; float4 *d_sorted = nullptr, *d_out = nullptr;
  hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
  int32_t rc = TSDF_OK;
  auto guard = [&](hipError_t err, const char* what) { if (err != hipSuccess && rc == TSDF_OK) rc = fail_hip(err, what); return err == hipSuccess; };
  if (guard(hipStreamCreate(&st), "hipStreamCreate") && guard(hipEventCreate(&e0), "hipEventCreate") && guard(hipEventCreate(&e1), "hipEventCreate") &&
      guard(hipMalloc(&d_xyz, n64 * 3 * sizeof(float)), "hipMalloc(cv_xyz)") && guard(hipMalloc(&d_count, (cells + 1) * 4), "hipMalloc(cells)") &&
      guard(hipMalloc(&d_start, (cells + 1) * 4), "hipMalloc(cells)") && guard(hipMalloc(&d_sums, nb * 4), "hipMalloc(sums)") &&
      guard(hipMalloc(&d_sorted, n64 * sizeof(float4)), "hipMalloc(sorted samples)") && guard(hipMalloc(&d_out, nq * sizeof(float4)), "hipMalloc(cv_xyz_inv)") &&
      guard(hipMemcpyAsync(d_xyz, cv_xyz, n64 * 3 * sizeof(float), hipMemcpyHostToDevice, st), "upload")) {
    guard(hipEventRecord(e0, st), "hipEventRecord");
    rr::launch_inverter_build(st, G, d_xyz, d_count, d_start, d_sums, d_sorted);
    rr::launch_inverter_query(st, G, Q, d_xyz, d_start, d_sorted, d_out);
    guard(hipGetLastError(), "kernel launch");
    guard(hipEventRecord(e1, st), "hipEventRecord");
    guard(hipMemcpyAsync(cv_xyz_inv, d_out, nq * sizeof(float4), hipMemcpyDeviceToHost, st), "download");
    guard(hipStreamSynchronize(st), "hipStreamSynchronize");
    if (rc == TSDF_OK && gpu_ms) guard(hipEventElapsedTime(gpu_ms, e0, e1), "hipEventElapsedTime");
  }
  (void)hipFree(d_xyz); (void)hipFree(d_count); (void)hipFree(d_start); (void)hipFree(d_sums); (void)hipFree(d_sorted); (void)hipFree(d_out);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}

}  // extern "C"
