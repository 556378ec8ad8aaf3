// Host-only file formats of the reference.  No GPU involved; pinned against the reference's own readers/writers
// (oracle/_ref, tests/test_calib_io.py, tests/test_oracle_ingest.py).
//
// Calibration volume files (*.cv_xyz RGB32F, *.cv_uv RG32F, *.cv_xyz_inv RGBA32F): the on-disk format of
// kinect::CalibrationVolume<T>::read / write (framework/calibration/calibration_volume.hpp:30-38, :62-78):
//   u32 res.x, res.y, res.z; f32 depth_min, depth_max; T volume[res.x * res.y * res.z]   (x fastest, :57-59)
// Host-only; no GPU involved.  Pinned against the reference's own reader/writer by tests/test_calib_io.py.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/rgbd_recon_hip.h"

namespace {
thread_local std::string g_io_error;
struct Header { uint32_t res[3]; float limits[2]; };
static_assert(sizeof(Header) == 20, "20-byte header");

int32_t fail(const std::string& m) { g_io_error = m; return TSDF_ERR_INVALID_ARGUMENT; }
}  // namespace
void rr_set_io_error(const std::string& m) { g_io_error = m; }
namespace {

int32_t read_header(FILE* f, const char* path, uint32_t texel_floats, Header* h) {
  if (fread(h, sizeof(Header), 1, f) != 1) return fail(std::string(path) + ": shorter than the 20-byte header");
  const uint64_t n = (uint64_t)h->res[0] * h->res[1] * h->res[2];
  if (n == 0 || h->res[0] > 4096 || h->res[1] > 4096 || h->res[2] > 4096) return fail(std::string(path) + ": implausible resolution");
  if (fseek(f, 0, SEEK_END) != 0) return fail(std::string(path) + ": not seekable");
  const long size = ftell(f);
  if ((uint64_t)size != sizeof(Header) + n * texel_floats * sizeof(float))
    return fail(std::string(path) + ": file size does not match its header for " + std::to_string(texel_floats) + "-float texels");
  fseek(f, (long)sizeof(Header), SEEK_SET);
  return TSDF_OK;
}
}  // namespace

extern "C" {

const char* tsdf_calib_last_error(void) { return g_io_error.c_str(); }

int32_t tsdf_calib_volume_info(const char* path, uint32_t texel_floats, uint32_t res[3], float depth_limits[2]) {
  if (!path || texel_floats < 1 || texel_floats > 4) return fail("bad argument");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(std::string(path) + ": cannot open");
  Header h;
  const int32_t rc = read_header(f, path, texel_floats, &h);
  fclose(f);
  if (rc) return rc;
  if (res) memcpy(res, h.res, sizeof(h.res));
  if (depth_limits) memcpy(depth_limits, h.limits, sizeof(h.limits));
  return TSDF_OK;
}

int32_t tsdf_calib_volume_read(const char* path, uint32_t texel_floats, float* data, uint64_t capacity_floats) {
  if (!path || !data || texel_floats < 1 || texel_floats > 4) return fail("bad argument");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(std::string(path) + ": cannot open");
  Header h;
  int32_t rc = read_header(f, path, texel_floats, &h);
  if (rc == TSDF_OK) {
    const uint64_t n = (uint64_t)h.res[0] * h.res[1] * h.res[2] * texel_floats;
    if (n > capacity_floats) rc = fail("destination too small");
    else if (fread(data, sizeof(float), n, f) != n) rc = fail(std::string(path) + ": short read");
  }
  fclose(f);
  return rc;
}

int32_t tsdf_calib_volume_write(const char* path, uint32_t texel_floats, const uint32_t res[3], const float depth_limits[2], const float* data) {
  if (!path || !res || !depth_limits || !data || texel_floats < 1 || texel_floats > 4) return fail("bad argument");
  FILE* f = fopen(path, "wb");
  if (!f) return fail(std::string(path) + ": cannot create");
  Header h;
  memcpy(h.res, res, sizeof(h.res));
  memcpy(h.limits, depth_limits, sizeof(h.limits));
  const uint64_t n = (uint64_t)res[0] * res[1] * res[2] * texel_floats;
  const bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(data, sizeof(float), n, f) == n;
  if (fclose(f) != 0 || !ok) return fail(std::string(path) + ": write failed");
  return TSDF_OK;
}

// ---- recordings/<sensor>.stream: sys::FileBuffer (framework/io/FileBuffer.cpp) as driven by NetKinectArray::readFromFiles
// (framework/NetKinectArray.cpp:709-749): records of colorsize + depthsize bytes back to back, no header.
int32_t tsdf_stream_num_frames(const char* path, uint64_t record_bytes, uint64_t* frames) {   // FileBuffer::calcNumFrames, :60-62
  if (!path || !record_bytes || !frames) return fail("bad argument");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(std::string(path) + ": cannot open");
  fseek(f, 0, SEEK_END);
  *frames = (uint64_t)ftell(f) / record_bytes;
  fclose(f);
  return TSDF_OK;
}
int32_t tsdf_stream_read_record(const char* path, uint64_t record_bytes, uint64_t frame, void* out) {
  if (!path || !record_bytes || !out) return fail("bad argument");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(std::string(path) + ": cannot open");
  fseek(f, 0, SEEK_END);
  const uint64_t size = (uint64_t)ftell(f);
  int32_t rc = TSDF_OK;
  // FileBuffer::read returns 0 bytes when the request runs past the end and looping is off (:96-106)
  if ((frame + 1) * record_bytes > size) rc = fail(std::string(path) + ": frame " + std::to_string(frame) + " is past the end of the stream");
  else if (fseek(f, (long)(frame * record_bytes), SEEK_SET) != 0 || fread(out, 1, record_bytes, f) != record_bytes) rc = fail(std::string(path) + ": short read");
  fclose(f);
  return rc;
}

}  // extern "C"
