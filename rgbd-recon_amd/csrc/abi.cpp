// C-ABI implementation (include/rgbd_recon_hip.h): host-side state of the HIP TSDF core.
// Mirrors the host logic of kinect::ReconIntegration (framework/reconstruction/recon_integration.cpp):
// setVoxelSize/setBrickSize/divideBox (:340-406,:462-472), draw() matrix set-up (:182-205),
// ViewLod::setResolution (framework/rendering/view_lod.cpp:24-50), the per-frame call order of
// source/kinect_client.cpp:569-599,614.  All device work goes to one HIP stream.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "tsdf_common.hpp"

using namespace rr;

#include "ctx.hpp"

thread_local std::string g_create_error;

namespace rrhost {       // host-side helpers (shared with comm.cpp through ctx.hpp where declared there)

// n / d == (n * m) >> k for every n < 2^27: with l = ceil(log2 d), k = 27 + l and m = floor(2^k / d) + 1 = (2^k + e) / d, 0 < e <= d <= 2^l,
// the product is n / d + n e / (d 2^k) with n e < 2^27 2^l = 2^k, i.e. less than 1 / d above the true quotient: the floor is the same.
// (tile ids stay below 2^27: resolutions are capped at 4096 = 512 tiles per axis)
FastDiv make_fast_div(uint32_t d) {
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  FastDiv f;
  f.k = 27 + l;
  f.m = (uint32_t)(((1ull << f.k) / d) + 1);
  return f;
}

// ---- small double-precision matrix kit (column-major), results rounded once to float
void mat_mul_d(const double* a, const double* b, double* o) {
  double r[16];
  for (int c = 0; c < 4; ++c)
    for (int row = 0; row < 4; ++row) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a[k * 4 + row] * b[c * 4 + k];
      r[c * 4 + row] = s;
    }
  memcpy(o, r, sizeof(r));
}
// Gauss-Jordan with partial pivoting on [M | I]
bool mat_inv_d(const double* m, double* o) {
  double a[4][8];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) { a[r][c] = m[c * 4 + r]; a[r][4 + c] = (r == c) ? 1.0 : 0.0; }
  for (int col = 0; col < 4; ++col) {
    int piv = col;
    for (int r = col + 1; r < 4; ++r) if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
    if (a[piv][col] == 0.0) return false;
    if (piv != col) for (int k = 0; k < 8; ++k) std::swap(a[piv][k], a[col][k]);
    const double inv = 1.0 / a[col][col];
    for (int k = 0; k < 8; ++k) a[col][k] *= inv;
    for (int r = 0; r < 4; ++r) if (r != col) {
      const double f = a[r][col];
      if (f != 0.0) for (int k = 0; k < 8; ++k) a[r][k] -= f * a[col][k];
    }
  }
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) o[c * 4 + r] = a[r][4 + c];
  return true;
}
Mat4 to_mat4(const double* d) { Mat4 r; for (int i = 0; i < 16; ++i) r.m[i] = (float)d[i]; return r; }

void release_view(tsdf_ctx* c) {
  for (int k = 0; k < 2; ++k) { hipFree(c->atlas_color[k]); hipFree(c->atlas_depth[k]); c->atlas_color[k] = nullptr; c->atlas_depth[k] = nullptr; }
  c->atlas_parity = 0;
  hipFree(c->d_peels); hipFree(c->d_peels_alt); c->d_peels_alt = nullptr; c->last_alt_peels = false; hipFree(c->d_nsamples); hipFree(c->d_fb_c); hipFree(c->d_fb_d);
  hipFree(c->d_long); c->d_long = nullptr;
  hipFree(c->d_tri_z); hipFree(c->d_tri_acc); c->d_tri_z = nullptr; c->d_tri_acc = nullptr;
  for (int k = 0; k < 3; ++k) { hipFree(c->d_touched[k]); c->d_touched[k] = nullptr; }
  for (int k = 0; k < 2; ++k) { hipFree(c->d_fill_mask[k]); hipFree(c->d_lvl_mask[k]); c->d_fill_mask[k] = nullptr; c->d_lvl_mask[k] = nullptr; }
  c->draw_masks_valid = false; c->fb_consistent = false;
  c->tile_history = false; c->touched_idx = 0;
  hipFree(c->d_hits); hipFree(c->d_hit_counters); hipFree(c->d_comp_key); c->d_hits = nullptr; c->d_hit_counters = nullptr; c->d_comp_key = nullptr;
  c->atlas.color = nullptr; c->atlas.depth = nullptr; c->d_peels = nullptr; c->d_nsamples = nullptr; c->d_fb_c = nullptr; c->d_fb_d = nullptr;
}
void release_bricks(tsdf_ctx* c) {
  for (int k = 0; k < 2; ++k) { hipFree(c->d_counters[k]); hipFree(c->d_flags[k]); hipFree(c->d_occupied[k]); c->d_counters[k] = nullptr; c->d_flags[k] = nullptr; c->d_occupied[k] = nullptr; }
  c->br.counters = nullptr; c->br.flags = nullptr; c->br.num_occupied = nullptr; c->br.occupied = nullptr;
  for (int a = 0; a < 3; ++a) {
    hipFree(c->d_vox_first[a]); hipFree(c->d_vox_count[a]); hipFree(c->d_tile_b0[a]); hipFree(c->d_tile_b1[a]); hipFree(c->d_tile_full[a]); c->d_tile_full[a] = nullptr;
    hipFree(c->d_brick_t0[a]); hipFree(c->d_brick_t1[a]);
    c->d_vox_first[a] = nullptr; c->d_vox_count[a] = nullptr; c->d_tile_b0[a] = nullptr; c->d_tile_b1[a] = nullptr;
    c->d_brick_t0[a] = nullptr; c->d_brick_t1[a] = nullptr;
  }
}

// ViewLod::setResolution, view_lod.cpp:24-50; resize(), recon_integration.cpp:482-500
int32_t setup_view(tsdf_ctx* c, uint32_t w, uint32_t h) {
  if (w < 2 || h < 2 || w > 16384 || h > 16384) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "view size %ux%u out of range", w, h);
  release_view(c);
  c->vw = (int)w; c->vh = (int)h;
  Atlas& A = c->atlas;
  A.num_lods = 1 + (int)floorf(log2f((float)std::min(w, h)));
  if (A.num_lods > TSDF_MAX_LODS) A.num_lods = TSDF_MAX_LODS;
  A.aw = (int)((float)w * 1.5f);
  A.h = (int)h;
  int ox = (int)w, oy = (int)h;
  for (int i = 0; i < TSDF_MAX_LODS; ++i) { A.off[i][0] = A.off[i][1] = A.res[i][0] = A.res[i][1] = 0; }
  for (int i = 0; i < A.num_lods; ++i) {
    A.res[i][0] = (int)floorf((float)w / powf(2.0f, (float)i));
    A.res[i][1] = (int)floorf((float)h / powf(2.0f, (float)i));
    if (i > 0) { oy -= A.res[i][1]; A.off[i][0] = ox; A.off[i][1] = oy; }
  }
  const size_t na = (size_t)A.aw * A.h, nv = (size_t)w * h;
  HIP_TRY(c, hipMalloc(&A.color, na * sizeof(float4)));
  HIP_TRY(c, hipMalloc(&A.depth, na * sizeof(float)));
  c->atlas_color[0] = A.color; c->atlas_depth[0] = A.depth; c->atlas_parity = 0;    // (the second pyramid: on the first overlapped draw)
  HIP_TRY(c, hipMalloc(&c->d_peels, nv * sizeof(float4)));
  HIP_TRY(c, hipMalloc(&c->d_peels_alt, nv * sizeof(float4)));
  HIP_TRY(c, hipMalloc(&c->d_nsamples, nv * sizeof(float)));
  HIP_TRY(c, hipMalloc(&c->d_fb_c, nv * sizeof(float4)));
  HIP_TRY(c, hipMalloc(&c->d_fb_d, nv * sizeof(float)));
  HIP_TRY(c, hipMalloc(&c->d_hits, nv * 16));
  HIP_TRY(c, hipMalloc(&c->d_long, nv * sizeof(LongRay)));
  const size_t n_img_tiles = (size_t)((c->vw + 7) / 8) * ((c->vh + 7) / 8);
  for (int k = 0; k < 3; ++k) { HIP_TRY(c, hipMalloc(&c->d_touched[k], n_img_tiles)); HIP_TRY(c, hipMemsetAsync(c->d_touched[k], 0, n_img_tiles, c->stream)); }
  for (int k = 0; k < 2; ++k) { HIP_TRY(c, hipMalloc(&c->d_fill_mask[k], n_img_tiles)); HIP_TRY(c, hipMalloc(&c->d_lvl_mask[k], n_img_tiles)); }
  c->draw_masks_valid = false; c->fb_consistent = false;
  if (const char* e = getenv("RR_FILL_TILES")) c->fill_tiles = atoi(e) != 0;          // A/B and test hook
  c->tile_history = false; c->touched_idx = 0;
  HIP_TRY(c, hipMalloc(&c->d_hit_counters, 4 * sizeof(uint32_t)));
  HIP_TRY(c, hipMemsetAsync(c->d_hit_counters, 0, 4 * sizeof(uint32_t), c->stream));
  if (const char* e = getenv("RR_MARCH_CAP")) c->march_cap = (uint32_t)atoi(e);
  if (const char* e = getenv("RR_MARCH_BOX")) c->march_box = atoi(e);
  if (const char* e = getenv("RR_K1_FORM")) c->k1_form_cap = atoi(e);     // A/B and test hook, read when the context is created
  if (const char* e = getenv("RR_IMAGE_TILES")) c->use_tile_history = atoi(e) != 0;
  c->hit_parity = 0;
  c->own_miss_counts = false;
  // the atlas starts as ViewLod::enable() leaves it (colour (0,1,0,0), depth 1); regions no kernel writes keep that
  launch_clear_image(c->stream, A.color, A.depth, na, make_float4(0.0f, 1.0f, 0.0f, 0.0f), 1.0f);
  launch_clear_image(c->stream, c->d_fb_c, c->d_fb_d, nv, make_float4(0, 0, 0, 0), 1.0f);
  HIP_TRY(c, hipMemsetAsync(c->d_peels, 0, nv * sizeof(float4), c->stream));
  HIP_TRY(c, hipMemsetAsync(c->d_peels_alt, 0, nv * sizeof(float4), c->stream));
  HIP_TRY(c, hipMemsetAsync(c->d_nsamples, 0, nv * sizeof(float), c->stream));
  return TSDF_OK;
}

// setBrickSize + divideBox, recon_integration.cpp:462-472, :360-406, with the per-brick voxel lists of
// VolumeSampler::containedVoxels (volume_sampler.cpp:50-62) kept as per-axis ranges.
int32_t setup_bricks(tsdf_ctx* c, const float req[3]) {
  for (int a = 0; a < 3; ++a) if (!(req[a] > 0.0f)) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "brick size must be > 0");
  release_bricks(c);
  memcpy(c->brick_req, req, sizeof(float) * 3);
  Bricks& B = c->br;
  const float bmin[3] = {c->cfg.bbox_min[0], c->cfg.bbox_min[1], c->cfg.bbox_min[2]};
  float ext[3];
  for (int a = 0; a < 3; ++a) {
    ext[a] = c->cfg.bbox_max[a] - c->cfg.bbox_min[a];
    B.size[a] = c->vox[a] * roundf(req[a] / c->vox[a]);                 // :463
    if (!(B.size[a] > 0.0f)) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "brick size %g rounds to zero voxels", req[a]);
    B.bbox_min[a] = bmin[a];
  }
  // the three nested while loops of divideBox() are separable: per axis, the sequence of brick starts
  std::vector<float> starts[3];
  for (int a = 0; a < 3; ++a) {
    float s = bmin[a];
    while (ext[a] - s + bmin[a] > 0.0f) {                               // :366-368
      starts[a].push_back(s);
      s += B.size[a];                                                   // :373,:379,:385
      if (starts[a].size() > 65535) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "more than 65535 bricks along one axis");
    }
    B.res[a] = (int)starts[a].size();
  }
  B.n = B.res[0] * B.res[1] * B.res[2];
  // containedVoxels(): for (v = pos/step; v < (pos+size)/step; ++v), all in float
  std::vector<uint16_t> first[3];
  std::vector<uint8_t> count[3];
  for (int a = 0; a < 3; ++a) {
    const float step = 1.0f / (float)c->res[a];
    first[a].assign(c->res[a], 0); count[a].assign(c->res[a], 0);
    for (size_t b = 0; b < starts[a].size(); ++b) {
      const float bs = std::min(B.size[a], ext[a] - starts[a][b] + bmin[a]);   // :369 glm::min(brick, size - start + min)
      const float pos = (starts[a][b] - bmin[a]) / ext[a], sz = bs / ext[a];   // :371
      unsigned v = (unsigned)(pos / step);
      for (; (float)v < (pos + sz) / step; ++v) {
        if (v >= (unsigned)c->res[a]) break;
        if (count[a][v] == 0) first[a][v] = (uint16_t)b;
        if (count[a][v] == 255 || (unsigned)first[a][v] + count[a][v] != b) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "degenerate brick/voxel overlap on axis %d", a);
        count[a][v]++;
      }
    }
    HIP_TRY(c, hipMalloc(&c->d_vox_first[a], c->res[a] * sizeof(uint16_t)));
    HIP_TRY(c, hipMalloc(&c->d_vox_count[a], c->res[a] * sizeof(uint8_t)));
    HIP_TRY(c, hipMemcpy(c->d_vox_first[a], first[a].data(), c->res[a] * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_vox_count[a], count[a].data(), c->res[a] * sizeof(uint8_t), hipMemcpyHostToDevice));
    B.vox_first[a] = c->d_vox_first[a];
    B.vox_count[a] = c->d_vox_count[a];
    const int nt = (c->res[a] + 7) / 8;
    std::vector<uint16_t> t0(nt, 1), t1(nt, 0);
    for (int t = 0; t < nt; ++t) {
      int lo = 0x7fffffff, hi = -1;
      for (int v = t * 8; v < std::min(t * 8 + 8, c->res[a]); ++v)
        if (count[a][v]) { lo = std::min(lo, (int)first[a][v]); hi = std::max(hi, (int)first[a][v] + count[a][v] - 1); }
      if (hi >= 0) { t0[t] = (uint16_t)lo; t1[t] = (uint16_t)hi; }
    }
    HIP_TRY(c, hipMalloc(&c->d_tile_b0[a], nt * sizeof(uint16_t)));
    HIP_TRY(c, hipMalloc(&c->d_tile_b1[a], nt * sizeof(uint16_t)));
    HIP_TRY(c, hipMemcpy(c->d_tile_b0[a], t0.data(), nt * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_tile_b1[a], t1.data(), nt * sizeof(uint16_t), hipMemcpyHostToDevice));
    B.tile_b0[a] = c->d_tile_b0[a];
    B.tile_b1[a] = c->d_tile_b1[a];
    std::vector<uint8_t> full(nt, 1);
    for (int t = 0; t < nt; ++t)
      for (int v = t * 8; v < std::min(t * 8 + 8, c->res[a]); ++v) if (!count[a][v]) full[t] = 0;
    HIP_TRY(c, hipMalloc(&c->d_tile_full[a], nt));
    HIP_TRY(c, hipMemcpy(c->d_tile_full[a], full.data(), nt, hipMemcpyHostToDevice));
    B.tile_full[a] = c->d_tile_full[a];
    // and the inverse: the storage tiles a brick's voxel list reaches into
    const int nb = (int)starts[a].size();
    std::vector<uint16_t> bt0(nb, 1), bt1(nb, 0);
    std::vector<int> blo(nb, 0x7fffffff), bhi(nb, -1);
    for (int v = 0; v < c->res[a]; ++v)
      for (int k = 0; k < count[a][v]; ++k) { const int b = first[a][v] + k; blo[b] = std::min(blo[b], v >> 3); bhi[b] = std::max(bhi[b], v >> 3); }
    for (int b = 0; b < nb; ++b) if (bhi[b] >= 0) { bt0[b] = (uint16_t)blo[b]; bt1[b] = (uint16_t)bhi[b]; }
    HIP_TRY(c, hipMalloc(&c->d_brick_t0[a], nb * sizeof(uint16_t)));
    HIP_TRY(c, hipMalloc(&c->d_brick_t1[a], nb * sizeof(uint16_t)));
    HIP_TRY(c, hipMemcpy(c->d_brick_t0[a], bt0.data(), nb * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_brick_t1[a], bt1.data(), nb * sizeof(uint16_t), hipMemcpyHostToDevice));
    B.brick_t0[a] = c->d_brick_t0[a];
    B.brick_t1[a] = c->d_brick_t1[a];
  }
  c->full_classify = true; c->alt.full = true;                           // a new brick grid: walk every tile once
  // tiles == bricks structurally?  (every voxel in exactly one brick per axis, and a tile never straddles two)
  bool uniform = true;
  for (int a = 0; a < 3 && uniform; ++a)
    for (int v = 0; v < c->res[a] && uniform; ++v)
      uniform = count[a][v] == 1 && first[a][v] == first[a][v & ~7];
  c->tiles.uniform = uniform ? 1 : 0;
  c->counter_words = (((size_t)B.n + 3 + 63) / 64) * 64;      // padded: one aligned fill kernel; k_update_occupied reads whole quads
  for (int k = 0; k < 2; ++k) {
    HIP_TRY(c, hipMalloc(&c->d_counters[k], c->counter_words * sizeof(uint32_t)));
    HIP_TRY(c, hipMemset(c->d_counters[k], 0, c->counter_words * sizeof(uint32_t)));
  }
  c->counters_cur = 0; c->spare_clean = true;
  B.counters = c->d_counters[0];
  if (!c->d_occ_counts) HIP_TRY(c, hipMalloc(&c->d_occ_counts, 3 * sizeof(uint32_t)));
  HIP_TRY(c, hipMemset(c->d_occ_counts, 0, 3 * sizeof(uint32_t)));      // a new grid: no occupied list yet
  c->occ_counts_stale = false;
  B.num_occupied = c->d_occ_counts + c->occ_parity;
  for (int k = 0; k < 2; ++k) {
    HIP_TRY(c, hipMalloc(&c->d_flags[k], (size_t)B.n));
    HIP_TRY(c, hipMalloc(&c->d_occupied[k], (size_t)B.n * sizeof(uint32_t)));
    HIP_TRY(c, hipMemset(c->d_flags[k], 0, (size_t)B.n));
  }
  B.flags = c->d_flags[c->occ_parity]; B.occupied = c->d_occupied[c->occ_parity];
  // the fills above run on the NULL stream, asynchronously with respect to the host, and the context's streams are non-blocking: finished
  // before any of them touches the new tables
  HIP_TRY(c, hipDeviceSynchronize());
  return TSDF_OK;
}

void timer_begin(tsdf_ctx* c, const char* name) {
  if (!c->timers_on) return;
  if (!c->timer_filter.empty() && c->timer_filter.find(std::string(",") + name + ",") == std::string::npos) return;
  Timer& t = c->timers[name];
  if (t.used == t.ev.size()) {
    if (t.ev.size() >= kMaxTimerPairs) { t.used = t.ev.size() - 1; }      // saturate: overwrite the last pair
    else { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); t.ev.emplace_back(a, b); }
  }
  hipEventRecord(t.ev[t.used].first, c->stream);
  t.open = true;
}
void timer_end(tsdf_ctx* c, const char* name) {
  if (!c->timers_on) return;
  auto it = c->timers.find(name);
  if (it == c->timers.end() || !it->second.open) return;
  Timer& t = it->second;
  hipEventRecord(t.ev[t.used].second, c->stream);
  t.used++;
  t.open = false;
}

void timer_begin_on(tsdf_ctx* c, const char* name, hipStream_t st) { hipStream_t keep = c->stream; c->stream = st; timer_begin(c, name); c->stream = keep; }
void timer_end_on(tsdf_ctx* c, const char* name, hipStream_t st) { hipStream_t keep = c->stream; c->stream = st; timer_end(c, name); c->stream = keep; }

// ---- the helper thread that issues the fill lane's calls (see tsdf_ctx::fill_worker)
}  // namespace rrhost
struct tsdf_ctx::FillWorker {
  struct Job { Atlas atlas; const uint8_t* tile_mask; uint8_t* lvl_mask[2]; int vw, vh; float4* fb_c; float* fb_d; int mask_mode, keep; hipEvent_t wait_ev, done_ev; };
  static constexpr uint64_t kRing = 16;
  Job ring[kRing];
  std::atomic<uint64_t> submitted{0}, issued{0};
  std::atomic<bool> stop{false}, asleep{false};
  std::atomic<int> hip_error{0};
  std::mutex m; std::condition_variable cv;
  int device = 0; hipStream_t stream = nullptr;
  std::thread th;
  static void relax() { __builtin_ia32_pause(); }
  void run() {
    (void)hipSetDevice(device);
    uint64_t done = 0;
    for (;;) {
      int spins = 0;
      while (submitted.load(std::memory_order_acquire) == done) {
        if (stop.load()) return;
        if (++spins < 20000) { relax(); continue; }                       // ~1 ms of spinning, then sleep until the next job is announced
        std::unique_lock<std::mutex> lk(m);
        asleep.store(true);
        cv.wait_for(lk, std::chrono::milliseconds(10), [&] { return submitted.load() != done || stop.load(); });
        asleep.store(false);
        spins = 0;
      }
      const Job& j = ring[done % kRing];
      hipError_t e = hipStreamWaitEvent(stream, j.wait_ev, 0);
      rr::launch_inpaint_pyramid(stream, j.atlas, j.tile_mask, j.lvl_mask);
      rr::launch_colorfill(stream, j.atlas, j.vw, j.vh, j.fb_c, j.fb_d, j.mask_mode, j.keep, j.tile_mask);
      const hipError_t e2 = hipEventRecord(j.done_ev, stream), e3 = hipGetLastError();
      if (e == hipSuccess) e = e2 != hipSuccess ? e2 : e3;
      if (e != hipSuccess) hip_error.store((int)e);
      ++done;
      issued.store(done, std::memory_order_release);
    }
  }
  uint64_t submit(const Job& j) {
    const uint64_t n = submitted.load(std::memory_order_relaxed);
    while (n - issued.load(std::memory_order_acquire) >= kRing) relax();
    ring[n % kRing] = j;
    submitted.store(n + 1);
    if (asleep.load()) { { std::lock_guard<std::mutex> lk(m); } cv.notify_one(); }
    return n + 1;
  }
  void wait_issued(uint64_t n) const { while (issued.load(std::memory_order_acquire) < n) relax(); }
  void drain() const { wait_issued(submitted.load()); }
};
namespace rrhost {
static hipError_t fill_worker_error(tsdf_ctx* c) {
  if (!c->fill_worker) return hipSuccess;
  const int e = c->fill_worker->hip_error.exchange(0);
  return (hipError_t)e;
}
// stage overlap: GPU-side join (the context's stream waits for the hole filling that is still in flight) and host-side sync of both streams
hipError_t join_fill_of(tsdf_ctx* c, int pyramid) {
  if (!c->fill_pending[pyramid]) return hipSuccess;
  c->fill_pending[pyramid] = false;
  if (c->fill_worker) { c->fill_worker->wait_issued(c->fill_job_no[pyramid]); const hipError_t e = fill_worker_error(c); if (e != hipSuccess) return e; }   // (its record must have been issued before this wait is)
  return hipStreamWaitEvent(c->stream, c->fill_done[pyramid], 0);
}
hipError_t join_fill(tsdf_ctx* c) {
  const hipError_t a = join_fill_of(c, 0), b = join_fill_of(c, 1);
  return a != hipSuccess ? a : b;
}
// ---- the lane ahead (see tsdf_ctx::pre_stream)
bool pipelined(const tsdf_ctx* c) { return c->overlap_fill && !c->pipeline_blocked; }
inline int alt_of(int x) { return x ^ 1; }
bool deep_ok(const tsdf_ctx* c);
// the deferred wait of the gate, now (every lane call that does not defer it itself starts with this)
static void gate_now(tsdf_ctx* c) {
  if (!c->gate_wait_pending) return;
  c->gate_wait_pending = false;
  hipStreamWaitEvent(c->pre_lane, c->gate_wait_ev, 0);
}
hipStream_t pre_enter(tsdf_ctx* c, bool defer_gate) {
  if (!pipelined(c)) return c->stream;
  if (!c->pre_stream) {                                                   // (a context created with RR_OVERLAP_FILL=0 and switched on later)
    if (hipStreamCreateWithFlags(&c->pre_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->pre_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->pre_gate, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->pre_gate_b, hipEventDisableTiming) != hipSuccess) { c->pipeline_blocked = true; return c->stream; }
  }
  if (c->main_since_gate) {                                               // the lane's first call of a new frame
    // which stream: the lane ahead's own -- or, with the integrate lane in use, that one: a frame's preparation and its integrate() then
    // follow each other without a cross-stream hand-over (15-35 us each on this machine) at the price of not
    // overlapping the preparation of frame f + 2 with the integrate of frame f + 1
    c->pre_lane = (c->pre_on_integ && deep_ok(c)) ? c->integ_stream : c->pre_stream;
    hipEvent_t const gate_old = c->gate_flip ? c->pre_gate_b : c->pre_gate, gate_new = c->gate_flip ? c->pre_gate : c->pre_gate_b;
    if (c->gate_wait_pending) { c->gate_wait_pending = false; hipStreamWaitEvent(c->pre_lane, c->gate_wait_ev, 0); }   // (a deferred wait nobody asked for: never skipped)
    if (c->pre_gate_recorded) {                                            // (recorded at the previous frame's first call: the consumers of the frame before that)
      if (defer_gate) { c->gate_wait_pending = true; c->gate_wait_ev = gate_old; }
      else hipStreamWaitEvent(c->pre_lane, gate_old, 0);
    }
    // ... on the context's stream, which waits for an integrate() on the fourth lane only when a DRAW joins it.  Frames integrated back to back with no
    // draw in between leave that integrate out of every gate, while it may still read the frame slot / brick counters / occupancy set this frame is about to
    // overwrite (the copies alternate): the lane waits for the integrate lane itself then.  (In the per-frame order upload .. integrate, draw the flag is
    // clear here -- the draw has joined the lane -- and nothing is added.)
    if (c->integ_pending && c->integ_stream && c->pre_lane != c->integ_stream) {
      hipEventRecord(c->integ_done, c->integ_stream);
      hipStreamWaitEvent(c->pre_lane, c->integ_done, 0);
    }
    hipEventRecord(gate_new, c->stream);
    c->gate_flip = !c->gate_flip;
    c->pre_gate_recorded = true; c->main_since_gate = false;
    c->slot_flipped = c->counters_flipped = c->occ_flipped = false;
    c->counters_zeroed = c->occ_count_zeroed = false;
  } else if (!defer_gate) gate_now(c);
  return c->pre_lane;
}
hipError_t pre_leave(tsdf_ctx* c, hipStream_t lane) {
  if (lane != c->stream) c->pre_pending = true;                          // (the event is recorded once, when a consumer asks: every record costs the lane ~4 us)
  return hipSuccess;
}
hipError_t join_pre(tsdf_ctx* c) {
  gate_now(c);
  c->main_since_gate = true;
  c->slot_in_use = c->counters_in_use = c->occ_in_use = true;
  if (!c->pre_pending) return hipSuccess;
  c->pre_pending = false;
  const hipError_t e = hipEventRecord(c->pre_done, c->pre_lane);
  return e != hipSuccess ? e : hipStreamWaitEvent(c->stream, c->pre_done, 0);
}
// leave the pipelined mode for good (explicit frame-slot calls, the pre-processing path): drain the lanes, everything on the context's stream from now on
hipError_t block_pipeline(tsdf_ctx* c) {
  if (c->pipeline_blocked) return hipSuccess;
  c->pipeline_blocked = true;
  hipError_t e = hipStreamSynchronize(c->stream);
  if (c->pre_stream) { const hipError_t f = hipStreamSynchronize(c->pre_stream); if (e == hipSuccess) e = f; }
  if (c->integ_stream) { const hipError_t f = hipStreamSynchronize(c->integ_stream); if (e == hipSuccess) e = f; c->integ_pending = false; }
  c->pre_pending = false;
  return e;
}
// ---- the fourth lane (see tsdf_ctx::integ_stream)
hipError_t join_integ(tsdf_ctx* c) {
  if (!c->integ_pending) return hipSuccess;
  c->integ_pending = false;
  const hipError_t e = hipEventRecord(c->integ_done, c->integ_stream);
  return e != hipSuccess ? e : hipStreamWaitEvent(c->stream, c->integ_done, 0);
}
bool deep_ok(const tsdf_ctx* c) {
  // (a Z-slab context too, when it recomputes its halo layers itself: an EXCHANGED halo is written into the volume from outside between integrate() and the draw)
  const bool whole = (c->vol.own_tz0 == 0 && c->vol.own_tz1 == (c->res[2] + 7) / 8);
  return c->deep && !c->deep_failed && pipelined(c) && c->integ_stream && (whole || c->cfg.slab_recompute_halo != 0) && !c->vol.slot && c->proj_budget == 0;
}
// exchange the set in use with the other one (host pointers only: kernels already queued keep the pointers they were launched with)
void swap_volume_set(tsdf_ctx* c) {
  tsdf_ctx::VolSet& a = c->alt;
  std::swap(c->vol.data, a.data); std::swap(c->d_cls_all, a.cls_all); std::swap(c->tiles.stamp, a.stamp);
  std::swap(c->d_tile_list[0], a.list[0]); std::swap(c->d_tile_list[1], a.list[1]); std::swap(c->d_tile_counts, a.counts);
  std::swap(c->tile_parity, a.parity); std::swap(c->full_classify, a.full); std::swap(c->frame_stamp, a.stampno);
  c->vol.cls = c->d_cls_all;
  c->tiles.cls = c->d_cls_all + (size_t)(c->vol.int_tz0 - c->vol.tz0) * c->vol.nty * c->vol.ntx;
  c->tiles.list = c->d_tile_list[0]; c->tiles.count = c->d_tile_counts;
  c->vol_set ^= 1;
}
// the second set, initialised on `lane` like setup_volume initialises the first; false = no memory for it (the context stays on one volume)
bool ensure_alt_set(tsdf_ctx* c, hipStream_t lane) {
  tsdf_ctx::VolSet& a = c->alt;
  if (a.data) return true;
  const Volume& V = c->vol;
  const size_t nvox = (size_t)V.n_stored_tiles * TILE_VOX, n = (size_t)c->tiles.n;
  bool ok = hipMalloc(&a.data, nvox * sizeof(float)) == hipSuccess && hipMalloc(&a.stamp, n * sizeof(uint32_t)) == hipSuccess &&
            hipMalloc(&a.cls_all, (size_t)V.n_stored_tiles) == hipSuccess && hipMalloc(&a.list[0], n * sizeof(uint32_t)) == hipSuccess &&
            hipMalloc(&a.list[1], n * sizeof(uint32_t)) == hipSuccess && hipMalloc(&a.counts, 2 * sizeof(uint32_t)) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    hipFree(a.data); hipFree(a.stamp); hipFree(a.cls_all); hipFree(a.list[0]); hipFree(a.list[1]); hipFree(a.counts);
    a = tsdf_ctx::VolSet{}; c->deep_failed = true;
    return false;
  }
  launch_fill_u32(lane, (uint32_t*)a.data, 0u, nvox);
  hipMemsetAsync(a.cls_all, kTileMixed, (size_t)V.n_stored_tiles, lane);
  hipMemsetAsync(a.counts, 0, 2 * sizeof(uint32_t), lane);
  hipMemsetAsync(a.stamp, 0, n * sizeof(uint32_t), lane);
  a.parity = 0; a.full = true; a.stampno = 0;
  return true;
}
hipError_t sync_ctx(tsdf_ctx* c) {
  hipError_t e = hipStreamSynchronize(c->stream);
  if (c->integ_stream) { const hipError_t f = hipStreamSynchronize(c->integ_stream); if (e == hipSuccess) e = f; c->integ_pending = false; c->draw_pending[0] = c->draw_pending[1] = false; }
  if (c->pre_stream) { const hipError_t f = hipStreamSynchronize(c->pre_stream); if (e == hipSuccess) e = f; c->pre_pending = false; }
  if (c->fill_worker) { c->fill_worker->drain(); const hipError_t f = fill_worker_error(c); if (e == hipSuccess) e = f; }
  if (c->fill_stream) { const hipError_t f = hipStreamSynchronize(c->fill_stream); if (e == hipSuccess) e = f; }
  c->fill_pending[0] = c->fill_pending[1] = false;
  return e;
}

// draw() matrix block, recon_integration.cpp:182-205 (+ vol_to_world :66-72)
// The matrices alone (no context): shared by make_view_params and the host-only tsdf_view_matrices.
bool view_matrices(const float* bbox_min, const float* bbox_max, int vw, int vh, const float* mv16, const float* pr16, ViewParams* P, Mat4* v2w_out) {
  double mv[16], pr[16], v2w[16] = {0}, sc[16] = {0}, tr[16] = {0}, t0[16], t1[16];
  for (int i = 0; i < 16; ++i) { mv[i] = mv16[i]; pr[i] = pr16[i]; }
  for (int a = 0; a < 3; ++a) {
    v2w[a * 5] = (double)(bbox_max[a] - bbox_min[a]);    // float subtraction like :66-68
    v2w[12 + a] = bbox_min[a];
  }
  v2w[15] = 1.0;
  if (v2w_out) *v2w_out = to_mat4(v2w);
  memcpy(P->mv.m, mv16, 64); memcpy(P->proj.m, pr16, 64);
  if (!mat_inv_d(v2w, t0)) return false;
  P->v2w_inv = to_mat4(t0);
  double mvi[16];
  if (!mat_inv_d(mv, mvi)) return false;
  P->mv_inv = to_mat4(mvi);
  sc[0] = vw * 0.5; sc[5] = vh * 0.5; sc[10] = 0.5; sc[15] = 1.0;                        // :187-191
  tr[0] = tr[5] = tr[10] = tr[15] = 1.0; tr[12] = tr[13] = tr[14] = 1.0;                 // :184-186
  mat_mul_d(tr, pr, t0); mat_mul_d(sc, t0, t1);
  if (!mat_inv_d(t1, t0)) return false;
  P->img_to_eye = to_mat4(t0);                                                            // :192-194
  double m[16], mi[16], mit[16];
  mat_mul_d(mv, v2w, m);
  P->mv_v2w = to_mat4(m);
  if (!mat_inv_d(m, mi)) return false;
  for (int col = 0; col < 4; ++col) for (int r = 0; r < 4; ++r) mit[col * 4 + r] = mi[r * 4 + col];
  P->normal = to_mat4(mit);                                                               // :199
  double mvt[16];
  for (int col = 0; col < 4; ++col) for (int r = 0; r < 4; ++r) mvt[col * 4 + r] = mv[r * 4 + col];
  P->glnormal_inv = to_mat4(mvt);                    // inverse(inverseTranspose(MV)), shading.glsl:65
  // camera position: inverse(MV) * (0,0,0,1), then inverse(vol_to_world) * that, in fp32 on rounded matrices (:202-205)
  const Mat4& I = P->mv_inv;
  const float cw[4] = {I.m[12], I.m[13], I.m[14], I.m[15]};
  for (int a = 0; a < 3; ++a) P->cam_world[a] = cw[a];
  const Mat4& W = P->v2w_inv;
  for (int r = 0; r < 3; ++r) P->cam_vol[r] = W.m[r] * cw[0] + W.m[4 + r] * cw[1] + W.m[8 + r] * cw[2] + W.m[12 + r] * cw[3];
  return true;
}

bool make_view_params(const tsdf_ctx* c, const float* mv16, const float* pr16, ViewParams* P) {
  if (!view_matrices(c->cfg.bbox_min, c->cfg.bbox_max, c->vw, c->vh, mv16, pr16, P, nullptr)) return false;
  P->w = c->vw; P->h = c->vh;
  P->shade_mode = c->shade_mode;
  P->skip = (c->skip_space && c->use_bricks) ? 1 : 0;                                     // :154, :510-513
  // with hole filling the raymarch is rasterised into the pyramid's level-0 viewport (0, 0, w, h) (ViewLod::enable, view_lod.cpp:68):
  // gl_FragCoord then carries no window origin, while the shader still subtracts viewport_offset ("currently not working",
  // recon_integration.cpp:528-530; the client switches hole filling off for side-by-side stereo, kinect_client.cpp:645-647)
  for (int a = 0; a < 2; ++a) { P->vp_org[a] = c->fill_holes ? 0 : c->vp_org[a]; P->vp_off[a] = c->vp_off[a]; }
  return true;
}

// frame slots: device images of one frame {packed depth/quality/silhouette, depth plane, RGBA8 colour}
int32_t alloc_frame_slot(tsdf_ctx* c, int k) {
  tsdf_ctx::FrameSlot& S = c->slots[k];
  if (S.dqs) return TSDF_OK;
  const size_t np = (size_t)c->cfg.num_streams * c->cfg.depth_w * c->cfg.depth_h, nc = (size_t)c->cfg.num_streams * c->cfg.color_w * c->cfg.color_h;
  HIP_TRY(c, hipMalloc((void**)&S.dqs, np * sizeof(float4)));
  HIP_TRY(c, hipMalloc((void**)&S.depth, np * sizeof(float)));
  HIP_TRY(c, hipMalloc((void**)&S.color, nc * sizeof(uchar4)));
  HIP_TRY(c, hipMalloc((void**)&S.ranges, (size_t)c->cfg.num_streams * ((c->cfg.depth_w + 7) / 8) * ((c->cfg.depth_h + 7) / 8) * sizeof(float4)));
  // (finished before anything else touches the slot: it may be written on another lane right away, and hipMemset itself is asynchronous
  // with respect to the host and not ordered against non-blocking streams)
  HIP_TRY(c, hipMemsetAsync(S.color, 0, nc * sizeof(uchar4), c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipEventCreateWithFlags(&S.ready, hipEventDisableTiming));
  HIP_TRY(c, hipEventCreateWithFlags(&S.released, hipEventDisableTiming));
  return TSDF_OK;
}
void use_frame_slot(tsdf_ctx* c, int k) {
  c->cur_slot = k;
  c->frame.dqs = c->slots[k].dqs; c->frame.depth = c->slots[k].depth; c->frame.color = c->slots[k].color;
  c->frame.ranges = c->use_ranges ? c->slots[k].ranges : nullptr; c->frame.rcw = ((int)c->cfg.depth_w + 7) / 8; c->frame.rch = ((int)c->cfg.depth_h + 7) / 8;
}

int halo_layers_for(float limit, int res_z) { return (int)ceilf((limit * (float)res_z + 2.0f) / 8.0f); }

// draw() without hole filling writes the default framebuffer through glColorMask (recon_integration.cpp:212-216): with a mask, or
// with a colour buffer the client did not clear, the march renders into the (otherwise unused) atlas and a merge pass follows
bool masked_direct(const tsdf_ctx* c) { return !c->fill_holes && (c->color_mask_mode != 0 || c->keep_color); }

RayTarget ray_target(tsdf_ctx* c) {
  RayTarget R{};
  if (masked_direct(c)) { R.color = c->atlas.color; R.depth = c->atlas.depth; R.stride = c->atlas.aw; R.clear[0] = R.clear[1] = R.clear[2] = R.clear[3] = 0; }
  else if (c->fill_holes) { R.color = c->atlas.color; R.depth = c->atlas.depth; R.stride = c->atlas.aw; R.clear[0] = 0; R.clear[1] = 1; R.clear[2] = 0; R.clear[3] = 0; }
  else { R.color = c->d_fb_c; R.depth = c->d_fb_d; R.stride = c->vw; R.clear[0] = R.clear[1] = R.clear[2] = R.clear[3] = 0; }
  R.nsamples = c->d_nsamples;
  R.peels = c->d_peels;
  return R;
}

// setVoxelSize()'s device side, recon_integration.cpp:340-348: the volume for the current c->res (tile-major storage, slot table of a
// sparse pool, per-tile state and work lists).  Called by tsdf_create and tsdf_set_voxel_size; everything it allocates is released first.
void release_volume(tsdf_ctx* c) {
  hipFree(c->vol.data); hipFree(c->vol.slot);                          // (both callers have synchronised the stream)
  hipFree(c->tiles.stamp); hipFree(c->d_cls_all);
  hipFree(c->d_tile_list[0]); hipFree(c->d_tile_list[1]); hipFree(c->d_tile_counts); hipFree(c->d_linear); hipFree(c->d_tile_bounds); hipFree(c->d_pair_masks); c->d_pair_masks = nullptr; hipFree(c->d_work_recs); c->d_work_recs = nullptr;
  hipFree(c->proj.data); hipFree(c->proj.slot); hipFree(c->proj.items); hipFree(c->d_proj_words); hipFree(c->d_item_stats);
  c->proj = ProjCache{}; c->d_proj_words = nullptr; c->d_item_stats = nullptr; c->proj_failed = false; c->last_integrate_cached = false;
  hipFree(c->alt.data); hipFree(c->alt.cls_all); hipFree(c->alt.stamp); hipFree(c->alt.list[0]); hipFree(c->alt.list[1]); hipFree(c->alt.counts);
  c->alt = tsdf_ctx::VolSet{}; c->deep_failed = false;
  c->vol.data = nullptr; c->vol.slot = nullptr; c->tiles.stamp = nullptr; c->d_cls_all = nullptr;
  c->d_tile_list[0] = c->d_tile_list[1] = nullptr; c->d_tile_counts = nullptr; c->d_linear = nullptr; c->d_tile_bounds = nullptr;
  c->tile_bounds_valid = false; c->tile_parity = 0; c->full_classify = true; c->frame_stamp = 0;
}
int32_t setup_volume(tsdf_ctx* c) {
  Volume& V = c->vol;
  for (int a = 0; a < 3; ++a) V.res[a] = c->res[a];
  V.ntx = (c->res[0] + 7) / 8; V.nty = (c->res[1] + 7) / 8;
  V.div_layer = make_fast_div((uint32_t)(V.ntx * V.nty)); V.div_row = make_fast_div((uint32_t)V.ntx);
  const int ntz = (c->res[2] + 7) / 8;
  if (!(V.limit > 0.0f)) V.limit = c->cfg.limit;                       // (re-created by setVoxelSize: keeps the current setTsdfLimit value)
  uint32_t z0 = c->cfg.slab_z0, z1 = c->cfg.slab_z1;
  if (z0 == 0 && z1 == 0) z1 = (uint32_t)c->res[2];
  if (z1 > (uint32_t)c->res[2] || z0 >= z1 || (z0 % 8) != 0 || (z1 % 8 != 0 && z1 != (uint32_t)c->res[2])) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "slab range must be tile (8) aligned and inside the volume");
  V.own_tz0 = (int)z0 / 8; V.own_tz1 = ((int)z1 + 7) / 8;
  // How far past a slab face an owned sample can make this context read (x = limit/2 * res_z voxels = one sampleDistance):
  // the refined hit position lies up to one step behind the owned sample (tsdf_raymarch.fs:99-101), its gradient taps another
  // sampleDistance further (:140-149), and the trilinear footprint of that tap half a voxel + one plane beyond:
  // ceil(2x + 0.5) planes below the face, floor(2x + 0.5) + 1 above.  (limit * res_z + 2) planes cover both.
  c->halo_layers = halo_layers_for(V.limit, c->res[2]);
  const bool whole = (V.own_tz0 == 0 && V.own_tz1 == ntz);
  V.tz0 = whole ? 0 : std::max(0, V.own_tz0 - c->halo_layers);
  V.tz1 = whole ? ntz : std::min(ntz, V.own_tz1 + c->halo_layers);
  V.zlo = V.tz0 * 8; V.zhi = std::min(V.tz1 * 8, c->res[2]) - 1;
  const bool recompute = !whole && c->cfg.slab_recompute_halo != 0;
  V.int_tz0 = recompute ? V.tz0 : V.own_tz0;
  V.int_tz1 = recompute ? V.tz1 : V.own_tz1;
  if (!whole && V.own_tz1 - V.own_tz0 < c->halo_layers) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "slab thinner than its halo");
  const bool sparse = c->cfg.sparse_pool_tiles > 0;
  V.n_stored_tiles = (V.tz1 - V.tz0) * V.nty * V.ntx;
  if ((uint64_t)(V.tz1 - V.tz0) * V.nty * V.ntx >= (1ull << 31)) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "more than 2^31 storage tiles");
  if (sparse && !whole && !recompute) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "a sparse slab context needs slab_recompute_halo (exchanged halo layers have no pool slots)");
  if (sparse && c->cfg.sparse_pool_tiles >= (1u << 23)) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "sparse_pool_tiles must be below 2^23 (16 GiB of tiles)");
  const size_t nvox = sparse ? (size_t)c->cfg.sparse_pool_tiles * TILE_VOX : (size_t)(V.tz1 - V.tz0) * V.nty * V.ntx * TILE_VOX;
  if (nvox >= (1ull << 32)) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "a context stores at most 2^32 voxels (32-bit tap offsets); split the volume into Z-slabs or use a sparse pool");
  int32_t rc;
  auto tryhip = [&](hipError_t e, const char* what) -> int32_t {
    if (e == hipSuccess) return TSDF_OK;
    c->err = std::string(what) + ": " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? TSDF_ERR_OUT_OF_MEMORY : TSDF_ERR_HIP;
  };
  release_volume(c);
  if ((rc = tryhip(hipMalloc(&V.data, nvox * sizeof(float)), "hipMalloc(volume)"))) return rc;
  launch_fill_u32(c->stream, (uint32_t*)V.data, 0u, nvox);
  V.slot = nullptr; V.pool_tiles = 0;
  if (sparse) {
    if ((rc = tryhip(hipMalloc(&V.slot, (size_t)V.n_stored_tiles * sizeof(uint32_t)), "hipMalloc(slot table)"))) return rc;
    if ((rc = tryhip(hipMemsetAsync(V.slot, 0xff, (size_t)V.n_stored_tiles * sizeof(uint32_t), c->stream), "hipMemset(slot table)"))) return rc;
    V.pool_tiles = c->cfg.sparse_pool_tiles;
  }
  TileState& S = c->tiles;
  S.n = (V.int_tz1 - V.int_tz0) * V.nty * V.ntx;
  if ((rc = tryhip(hipMalloc(&S.stamp, (size_t)S.n * sizeof(uint32_t)), "hipMalloc(tiles)"))) return rc;
  if ((rc = tryhip(hipMalloc(&c->d_cls_all, (size_t)V.n_stored_tiles), "hipMalloc(tiles)"))) return rc;
  hipMemsetAsync(c->d_cls_all, kTileMixed, (size_t)V.n_stored_tiles, c->stream);   // halo layers keep this value for good
  V.cls = c->d_cls_all;
  S.cls = c->d_cls_all + (size_t)(V.int_tz0 - V.tz0) * V.nty * V.ntx;
  for (int k = 0; k < 2; ++k)
    if ((rc = tryhip(hipMalloc(&c->d_tile_list[k], (size_t)S.n * sizeof(uint32_t)), "hipMalloc(tiles)"))) return rc;
  if ((rc = tryhip(hipMalloc(&c->d_tile_counts, 2 * sizeof(uint32_t)), "hipMalloc(tiles)"))) return rc;
  hipMemsetAsync(c->d_tile_counts, 0, 2 * sizeof(uint32_t), c->stream);
  hipMemsetAsync(S.stamp, 0, (size_t)S.n * sizeof(uint32_t), c->stream);
  S.list = c->d_tile_list[0]; S.count = c->d_tile_counts;
  return TSDF_OK;
}
// the largest LUT texel box any 8^3 tile of the current volume touches, with the kernel's own fp32 index arithmetic, against the LDS
// budget of the integrate kernels: which form stream i can use (depends on the LUT and on the volume resolution)
void fit_lut_to_volume(tsdf_ctx* c, uint32_t i) {
  const StreamLut& L = c->luts.s[i];
  int worst[3] = {1, 1, 1};
  for (int a = 0; a < 3; ++a) {
    const float step = 1.0f / (float)c->res[a];
    const int n = L.inv_res[a];
    auto idx0 = [&](int v) { float f = ((float)v + 0.5f) * step * (float)n - 0.5f; int k = (int)fminf(fmaxf(floorf(f), -1.0f), (float)n); return std::min(std::max(k, 0), n - 1); };
    auto idx1 = [&](int v) { float f = ((float)v + 0.5f) * step * (float)n - 0.5f; int k = (int)fminf(fmaxf(floorf(f), -1.0f), (float)n); return std::min(std::max(k + 1, 0), n - 1); };
    for (int t = 0; t * 8 < c->res[a]; ++t) worst[a] = std::max(worst[a], idx1(std::min(t * 8 + 7, c->res[a] - 1)) - idx0(t * 8) + 1);
  }
  c->lut_dz[i] = worst[2];
  c->lds_ok[i] = worst[0] * worst[1] * worst[2] > integrate_box_cap() ? 0 : ((worst[1] * worst[2] * 8 <= integrate_row_cap() && worst[2] * 64 <= integrate_box_cap()) ? 2 : 1);
}

}  // namespace rrhost
using namespace rrhost;

extern "C" {

const char* tsdf_last_error(const tsdf_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int32_t tsdf_create(const tsdf_config* cfg, tsdf_ctx** out) {
  if (!cfg || !out) { g_create_error = "null argument"; return TSDF_ERR_INVALID_ARGUMENT; }
  *out = nullptr;
  // struct_size lets the struct grow: a caller built against the layout that ended at slab_recompute_halo is still accepted
  tsdf_config grown{};
  if (cfg->struct_size == offsetof(tsdf_config, sparse_pool_tiles) || cfg->struct_size == offsetof(tsdf_config, proj_cache_mib) || cfg->struct_size == offsetof(tsdf_config, lane_flags)) {
    memcpy(&grown, cfg, cfg->struct_size); grown.struct_size = sizeof(tsdf_config); cfg = &grown;
  }
  if (cfg->struct_size != sizeof(tsdf_config)) { g_create_error = "tsdf_config.struct_size mismatch"; return TSDF_ERR_INVALID_ARGUMENT; }
  if (cfg->num_streams < 1 || cfg->num_streams > TSDF_MAX_STREAMS) { g_create_error = "num_streams out of range"; return TSDF_ERR_INVALID_ARGUMENT; }
  if (!(cfg->limit > 0.0f)) { g_create_error = "limit must be > 0"; return TSDF_ERR_INVALID_ARGUMENT; }
  for (int a = 0; a < 3; ++a)
    if (!(cfg->bbox_max[a] > cfg->bbox_min[a])) { g_create_error = "empty bounding box"; return TSDF_ERR_INVALID_ARGUMENT; }
  if (cfg->depth_w < 1 || cfg->depth_h < 1 || cfg->color_w < 1 || cfg->color_h < 1) { g_create_error = "image size must be >= 1"; return TSDF_ERR_INVALID_ARGUMENT; }
  if (cfg->depth_w > 65536 || cfg->depth_h > 65536 || cfg->color_w > 65536 || cfg->color_h > 65536) { g_create_error = "image sides above 65536 are not supported"; return TSDF_ERR_INVALID_ARGUMENT; }
  if ((uint64_t)cfg->depth_w * cfg->depth_h * cfg->num_streams >= (1ull << 24) * 16 || (uint64_t)cfg->depth_w * cfg->depth_h >= (1ull << 24) ||
      (uint64_t)cfg->color_w * cfg->color_h >= (1ull << 24)) { g_create_error = "images of 2^24 pixels or more are not supported (24-bit index arithmetic)"; return TSDF_ERR_INVALID_ARGUMENT; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_error = "no HIP device visible (the HIP path has no CPU fallback)"; return TSDF_ERR_NO_DEVICE; }
  if (cfg->device < 0 || cfg->device >= ndev) { g_create_error = "device ordinal out of range"; return TSDF_ERR_INVALID_ARGUMENT; }
  tsdf_ctx* c = new tsdf_ctx();
  c->cfg = *cfg;
  c->device = cfg->device;
  auto fail = [&](int32_t code) { g_create_error = c->err; tsdf_destroy(c); return code; };
  if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return fail(TSDF_ERR_HIP); }
  // RR_LANE_PRIORITY (A/B hook): "pre,fill,integ,main" stream priorities: -1 = high, 0 = normal, 1 = low
  // All normal by default.  The lane ahead BELOW the others ("1,0,0,0") is worth 4 % at c2 with one context (115 against 120 us per frame: its re-layout and
  // brick marking otherwise take the machine from the march) -- but a low-priority stream changes how the runtime deals its hardware queues: three contexts
  // in flight fall from 7 120 to 4 210 frames/s, and beside RCCL's kernels the lane starves (1 010 against 3 470 frames/s in the one-rank exchange rehearsal)
  int lo = 0, hi = 0, ppre = cfg->lane_priority[0], pfill = cfg->lane_priority[1], pinteg = cfg->lane_priority[2], pmain = cfg->lane_priority[3];
  hipDeviceGetStreamPriorityRange(&lo, &hi);                             // (least, greatest): numerically greatest <= least
  if (const char* e = getenv("RR_LANE_PRIORITY")) sscanf(e, "%d,%d,%d,%d", &ppre, &pfill, &pinteg, &pmain);
  auto prio = [&](int rel) { return rel < 0 ? hi : (rel > 0 ? lo : (lo + hi) / 2); };
  // (the context's own stream: created WITHOUT a priority unless the hook asks for one -- streams created through the priority call are dealt
  //  their hardware queues differently: several contexts' streams then share one, 4 250 instead of 7 080 frames/s with three contexts in flight)
  // Space sharing of the lanes (round 4): RR_LANE_XCDS = "pre,fill,integ,main", each a hex mask of the XCDs (bit x = XCD x) the lane's stream may run on
  // (0 / absent = all), or RR_LANE_CUS = "a-b,a-b,a-b,a-b": the CU slots [a, b] of EVERY XCD (0 - 31).  Bit i of a HIP CU mask is CU slot i / 8 of XCD i % 8
  // on this part (tools/probes/cumask_probe.hip).
  uint32_t lane_mask[4][8]; bool lane_masked[4] = {false, false, false, false};
  {
    unsigned xm[4] = {0, 0, 0, 0}; int lo4[4] = {0, 0, 0, 0}, hi4[4] = {31, 31, 31, 31};
    const char* ex = getenv("RR_LANE_XCDS"); const char* ec = getenv("RR_LANE_CUS");
    if (ex) sscanf(ex, "%x,%x,%x,%x", &xm[0], &xm[1], &xm[2], &xm[3]);
    if (ec) sscanf(ec, "%d-%d,%d-%d,%d-%d,%d-%d", &lo4[0], &hi4[0], &lo4[1], &hi4[1], &lo4[2], &hi4[2], &lo4[3], &hi4[3]);
    for (int l = 0; l < 4; ++l) {
      lane_masked[l] = (ex && (xm[l] & 0xffu) != 0u && (xm[l] & 0xffu) != 0xffu) || (ec && (lo4[l] > 0 || hi4[l] < 31));
      for (int k = 0; k < 8; ++k) lane_mask[l][k] = 0u;
      for (int i = 0; i < 256; ++i) {
        const int xcd = i & 7, cu = i >> 3;
        const bool on = (!ex || (xm[l] & 0xffu) == 0u || ((xm[l] >> xcd) & 1u)) && cu >= lo4[l] && cu <= hi4[l];
        if (on) lane_mask[l][i >> 5] |= 1u << (i & 31);
      }
    }
  }
  auto make_stream = [&](hipStream_t* st, int lane, int rel, bool with_priority) -> hipError_t {
    if (lane_masked[lane]) return hipExtStreamCreateWithCUMask(st, 8, lane_mask[lane]);
    return with_priority ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio(rel)) : hipStreamCreateWithFlags(st, hipStreamNonBlocking);
  };
  if (make_stream(&c->own_stream, 3, pmain, pmain != 0) != hipSuccess) { c->err = "hipStreamCreate failed"; return fail(TSDF_ERR_HIP); }
  c->stream = c->own_stream;
  // The three lanes of a context (stage overlap) are created together: the HIP runtime deals its hardware queues (4 by default,
  // GPU_MAX_HW_QUEUES) to streams in creation order, and two lanes that share a queue do not overlap at all
  // tsdf_config::lane_flags decides; the RR_* variables of the A/B tools override it when set
  c->overlap_fill = !(cfg->lane_flags & TSDF_LANES_ONE_STREAM); c->deep = !(cfg->lane_flags & TSDF_LANES_NO_INTEGRATE_LANE); c->fill_thread = !(cfg->lane_flags & TSDF_LANES_NO_FILL_THREAD);
  if (const char* e = getenv("RR_OVERLAP_FILL")) c->overlap_fill = atoi(e) != 0;
  if (c->overlap_fill) {
    const bool two_lanes = getenv("RR_LANES") ? atoi(getenv("RR_LANES")) == 2 : (cfg->lane_flags & TSDF_LANES_SHARED_FILL_LANE) != 0;   // the lane ahead and the fill lane share one stream
    if (make_stream(&c->pre_stream, 0, ppre, true) != hipSuccess ||
        (two_lanes ? (c->fill_stream = c->pre_stream, hipSuccess) : make_stream(&c->fill_stream, 1, pfill, true)) != hipSuccess ||
        hipEventCreateWithFlags(&c->pre_done, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->pre_gate, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->pre_gate_b, hipEventDisableTiming) != hipSuccess ||
        make_stream(&c->integ_stream, 2, pinteg, true) != hipSuccess ||
        hipEventCreateWithFlags(&c->draw_done[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->draw_done[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->integ_done, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->integ_gate, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->fill_done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->fill_done[1], hipEventDisableTiming) != hipSuccess) { c->err = "hipStreamCreate failed"; return fail(TSDF_ERR_HIP); }
  }
  if (const char* e = getenv("RR_K1_RANGES")) c->use_ranges = atoi(e) != 0;
  if (const char* e = getenv("RR_FILL_THREAD")) c->fill_thread = atoi(e) != 0;
  if (const char* e = getenv("RR_PRE_ON_INTEG")) c->pre_on_integ = atoi(e) != 0;
  if (const char* e = getenv("RR_DEEP")) c->deep = atoi(e) != 0;          // A/B and test hook: integrate() on the context's stream, one volume
  {
    uint64_t mib = cfg->proj_cache_mib;                                   // 0: off (the default: measured slower than the LUT kernel)
    if (mib == 0) if (const char* e = getenv("RR_PROJ_CACHE_MB")) mib = (uint64_t)atoll(e);   // A/B and test hook
    c->proj_budget = (size_t)(mib << 20);
  }
  if (const char* e = getenv("RR_K1_CULLED_RANGES")) c->culled_ranges = atoi(e) != 0;
  if (const char* e = getenv("RR_K1_REC")) c->use_recs = atoi(e) != 0;    // A/B hook: 0 = the LDS form's own tile head
  // setVoxelSize(), :340-347
  for (int a = 0; a < 3; ++a) {
    const float ext = cfg->bbox_max[a] - cfg->bbox_min[a];
    if (cfg->res[0] == 0) {
      if (!(cfg->voxel_size > 0.0f)) { c->err = "voxel_size must be > 0 when res is not given"; return fail(TSDF_ERR_INVALID_ARGUMENT); }
      c->res[a] = (int)ceilf(ext / cfg->voxel_size);
      c->vox[a] = cfg->voxel_size;
    } else {
      c->res[a] = (int)cfg->res[a];
      c->vox[a] = ext / (float)cfg->res[a];
    }
    if (c->res[a] < 1 || c->res[a] > 4096) { c->err = "volume resolution out of range [1, 4096]"; return fail(TSDF_ERR_INVALID_ARGUMENT); }
  }
  int32_t rc;
  if ((rc = setup_volume(c))) return fail(rc);
  auto tryhip = [&](hipError_t e, const char* what) -> int32_t {
    if (e == hipSuccess) return TSDF_OK;
    c->err = std::string(what) + ": " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? TSDF_ERR_OUT_OF_MEMORY : TSDF_ERR_HIP;
  };
  if ((rc = tryhip(hipHostMalloc((void**)&c->h_num_occupied, sizeof(uint32_t), hipHostMallocDefault), "hipHostMalloc"))) return fail(rc);
  *c->h_num_occupied = 0;
  if ((rc = setup_bricks(c, cfg->brick_size))) return fail(rc);
  // frame images
  FrameImages& F = c->frame;
  F.w = (int)cfg->depth_w; F.h = (int)cfg->depth_h; F.cw = (int)cfg->color_w; F.ch = (int)cfg->color_h;
  const size_t np = (size_t)cfg->num_streams * F.w * F.h, nc = (size_t)cfg->num_streams * F.cw * F.ch;
  if ((rc = alloc_frame_slot(c, 0))) return fail(rc);
  use_frame_slot(c, 0);
  if ((rc = tryhip(hipMalloc(&c->d_stage_depth, np * 8), "hipMalloc(stage)"))) return fail(rc);
  if ((rc = tryhip(hipMalloc(&c->d_stage_q, np * 4), "hipMalloc(stage)"))) return fail(rc);
  if ((rc = tryhip(hipMalloc(&c->d_stage_s, np * 4), "hipMalloc(stage)"))) return fail(rc);
  if ((rc = tryhip(hipMalloc(&c->d_stage_col, nc * 3), "hipMalloc(stage)"))) return fail(rc);
  c->luts.n = (int)cfg->num_streams;
  c->pre.filter_textures = 1; c->pre.refine = 1;                      // NetKinectArray.cpp:63-69
  if ((rc = setup_view(c, cfg->view_w, cfg->view_h))) return fail(rc);
  if ((rc = tryhip(sync_ctx(c), "hipStreamSynchronize"))) return fail(rc);
  *out = c;
  return TSDF_OK;
}

int32_t tsdf_destroy(tsdf_ctx* c) {
  CHECK_CTX(c);
  hipSetDevice(c->device);
  sync_ctx(c);          // (a null handle is the NULL stream: tsdf_adopt_null_stream)
  if (c->copy_stream) hipStreamSynchronize(c->copy_stream);   // an asynchronous upload may still be writing a frame slot
  tsdf_comm_destroy(c);
  release_view(c); release_bricks(c);
  release_volume(c);
  for (auto& sl : c->slots) { hipFree(sl.dqs); hipFree(sl.depth); hipFree(sl.color); hipFree(sl.ranges); if (sl.ready) hipEventDestroy(sl.ready); if (sl.released) hipEventDestroy(sl.released); }
  for (int k = 0; k < 2; ++k) { if (c->h_stage[k]) hipHostFree(c->h_stage[k]); if (c->stage_done[k]) hipEventDestroy(c->stage_done[k]); }
  hipFree(c->d_astage);
  if (c->copy_stream) hipStreamDestroy(c->copy_stream);
  hipFree(c->d_raw); hipFree(c->d_depth2); hipFree(c->d_depth_rg); hipFree(c->d_lab); hipFree(c->d_depth_b); hipFree(c->d_normal); hipFree(c->d_pre_blocks);
  hipFree(c->d_stage_depth); hipFree(c->d_stage_q); hipFree(c->d_stage_s); hipFree(c->d_stage_col);
  for (auto& per : c->lut_alloc) for (void* p : per) hipFree(p);
  if (c->h_num_occupied) hipHostFree(c->h_num_occupied);
  hipFree(c->d_occ_counts);
  for (int k = 0; k < 2; ++k) { if (c->h_wire[k]) hipHostFree(c->h_wire[k]); if (c->wire_done[k]) hipEventDestroy(c->wire_done[k]); }
  hipFree(c->d_wire);
  for (auto& kv : c->timers) for (auto& e : kv.second.ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  if (c->pre_stream && c->pre_stream != c->fill_stream) hipStreamDestroy(c->pre_stream);
  if (c->pre_done) hipEventDestroy(c->pre_done);
  if (c->pre_gate) hipEventDestroy(c->pre_gate);
  if (c->pre_gate_b) hipEventDestroy(c->pre_gate_b);
  if (c->src_ready) hipEventDestroy(c->src_ready);
  if (c->normals_read) hipEventDestroy(c->normals_read);
  if (c->fill_worker) {
    c->fill_worker->stop.store(true);
    { std::lock_guard<std::mutex> lk(c->fill_worker->m); }
    c->fill_worker->cv.notify_one();
    if (c->fill_worker->th.joinable()) c->fill_worker->th.join();
    delete c->fill_worker; c->fill_worker = nullptr;
  }
  if (c->fill_stream) hipStreamDestroy(c->fill_stream);
  if (c->integ_stream) hipStreamDestroy(c->integ_stream);
  for (hipEvent_t e : {c->integ_done, c->integ_gate, c->draw_done[0], c->draw_done[1]}) if (e) hipEventDestroy(e);
  for (hipEvent_t e : c->fill_done) if (e) hipEventDestroy(e);
  if (c->own_stream) hipStreamDestroy(c->own_stream);
  delete c;
  return TSDF_OK;
}

int32_t tsdf_sparse_pool_stats(tsdf_ctx* c, uint32_t* need, uint32_t* cap) {
  CHECK_CTX(c);
  if (!c->vol.slot) FAIL(c, TSDF_ERR_STATE, "not a sparse context");
  HIP_TRY(c, hipSetDevice(c->device));
  uint32_t n = 0;
  HIP_TRY(c, hipMemcpyAsync(&n, c->tiles.count, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, sync_ctx(c));
  if (need) *need = n;
  if (cap) *cap = c->vol.pool_tiles;
  return TSDF_OK;
}
int32_t tsdf_fill_stats(tsdf_ctx* c, uint64_t out[2]) {
  CHECK_CTX(c);
  if (!out) return TSDF_ERR_INVALID_ARGUMENT;
  out[0] = c->n_fills; out[1] = c->n_fills_by_tiles;
  return TSDF_OK;
}
int32_t tsdf_integrate_stats(tsdf_ctx* c, uint32_t out[6]) {
  CHECK_CTX(c);
  if (!out) return TSDF_ERR_INVALID_ARGUMENT;
  for (int k = 0; k < 6; ++k) out[k] = 0;
  if (!c->last_integrate_cached || !c->proj.data) return TSDF_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->d_item_stats) HIP_TRY(c, hipMalloc((void**)&c->d_item_stats, 4 * sizeof(uint32_t)));
  // (the tile list of the last integrate: tsdf_integrate has already flipped the parity)
  TileState S = c->tiles;
  if (c->use_bricks) { const int p = c->tile_parity ^ 1; S.list = c->d_tile_list[p]; S.count = c->d_tile_counts + p; }
  launch_item_stats(c->stream, c->luts, S, c->use_bricks ? 1 : 0, c->d_pair_masks, c->proj, c->d_item_stats);
  HIP_TRY(c, hipMemcpyAsync(out, c->d_item_stats, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(out + 4, c->proj.alloc, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, sync_ctx(c));
  out[4] = std::min(out[4], c->proj.cap); out[5] = c->proj.cap;
  return TSDF_OK;
}
int32_t tsdf_set_stream(tsdf_ctx* c, void* s) {
  CHECK_CTX(c);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_ctx(c));
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return TSDF_OK;
}
// The process's NULL ("legacy default") stream has the handle 0, which tsdf_set_stream reads as "back to the context's own
// stream": adopting it needs an entry point of its own.  torch.cuda.default_stream().cuda_stream IS 0, so a caller that wants
// the context's kernels ordered with torch ops / RCCL collectives issued on torch's default stream must call this one.
int32_t tsdf_adopt_null_stream(tsdf_ctx* c) {
  CHECK_CTX(c);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_ctx(c));
  c->stream = nullptr;                         // hipStream_t 0: every launch / copy / event record below goes to the NULL stream
  return TSDF_OK;
}
int32_t tsdf_sync(tsdf_ctx* c) { CHECK_CTX(c); HIP_TRY(c, sync_ctx(c)); return TSDF_OK; }

int32_t tsdf_set_calibration(tsdf_ctx* c, uint32_t i, const float* inv, const uint32_t ri[3], const float* uv, const uint32_t ru[3], const float* xyz, const uint32_t rx[3]) {
  CHECK_CTX(c);
  if (i >= c->cfg.num_streams) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "stream %u out of range", i);
  if (!inv || !ri) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "cv_xyz_inv is required");
  HIP_TRY(c, hipSetDevice(c->device));
  StreamLut& L = c->luts.s[i];
  auto vol_n = [](const uint32_t r[3]) { return (size_t)r[0] * r[1] * r[2]; };
  // every LUT is indexed with 24-bit multiplies per axis and a 32-bit texel index in the kernels: the same bounds for all three
  auto check_res = [&](const uint32_t r[3], const char* what) -> int32_t {
    for (int a = 0; a < 3; ++a) if (r[a] < 1 || r[a] > 2048) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "%s resolution out of range [1, 2048]", what);
    if ((uint64_t)r[0] * r[1] * r[2] > (1ull << 31)) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "%s larger than 2^31 texels (the kernels index it with 32 bits)", what);
    return TSDF_OK;
  };
  if (int32_t rc = check_res(ri, "cv_xyz_inv")) return rc;
  if (uv) { if (!ru) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "cv_uv resolution missing"); if (int32_t rc = check_res(ru, "cv_uv")) return rc; }
  if (xyz) { if (!rx) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "cv_xyz resolution missing"); if (int32_t rc = check_res(rx, "cv_xyz")) return rc; }
  // re-calibration of a stream: queued kernels may still read the old volumes; wait, then free what this call replaces
  if (c->have_calib[i]) HIP_TRY(c, sync_ctx(c));
  auto replace = [&](int slot, void* fresh) { if (c->lut_alloc[i][slot]) hipFree(c->lut_alloc[i][slot]); c->lut_alloc[i][slot] = fresh; };
  // allocate, copy, and only then swap: a failed copy must leave the stream's old volume (and L.*) in place, never a freed pointer
  auto upload = [&](void** fresh, const void* src, size_t bytes) -> int32_t {
    *fresh = nullptr;
    HIP_TRY(c, hipMalloc(fresh, bytes));
    const hipError_t e = hipMemcpy(*fresh, src, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(*fresh); *fresh = nullptr; FAIL(c, TSDF_ERR_HIP, "hipMemcpy(calibration volume) failed: %s", hipGetErrorString(e)); }
    return TSDF_OK;
  };
  void* fresh = nullptr;
  if (int32_t rc = upload(&fresh, inv, vol_n(ri) * sizeof(float4))) return rc;
  replace(0, fresh);
  L.inv = (const float4*)fresh; for (int a = 0; a < 3; ++a) L.inv_res[a] = (int)ri[a];
  fit_lut_to_volume(c, i);
  c->tile_bounds_valid = false;                                          // the tiles' LUT-box bounds and cached projections belong to the old volume
  if (c->proj.data) {                                                    // (the stream is idle: synchronised above) the pool's slot size follows the LUTs: re-made by the next integrate()
    hipFree(c->proj.data); hipFree(c->proj.slot); hipFree(c->proj.items); hipFree(c->d_proj_words);
    c->proj = ProjCache{}; c->d_proj_words = nullptr; c->last_integrate_cached = false;
  }
  c->proj_failed = false;
  if (uv) {
    if (int32_t rc = upload(&fresh, uv, vol_n(ru) * sizeof(float2))) return rc;
    replace(1, fresh);
    L.uv = (const float2*)fresh; for (int a = 0; a < 3; ++a) L.uv_res[a] = (int)ru[a];
  }
  if (xyz) {
    const size_t n = vol_n(rx);
    std::vector<float> padded(n * 4);
    for (size_t k = 0; k < n; ++k) { padded[4 * k] = xyz[3 * k]; padded[4 * k + 1] = xyz[3 * k + 1]; padded[4 * k + 2] = xyz[3 * k + 2]; padded[4 * k + 3] = 0.0f; }
    if (int32_t rc = upload(&fresh, padded.data(), n * sizeof(float4))) return rc;
    replace(2, fresh);
    L.xyz = (const float4*)fresh; for (int a = 0; a < 3; ++a) L.xyz_res[a] = (int)rx[a];
    // CalibVolumes::addVolume builds the sensor's frustum from this volume (CalibVolumes.cpp:122) and getCameraPositions()
    // (:224-230) feeds the quality pass: default camera position, until tsdf_set_camera_position overrides it
    float cam[3];
    if (tsdf_frustum_from_volume(xyz, rx, nullptr, cam) == TSDF_OK && std::isfinite(cam[0]) && std::isfinite(cam[1]) && std::isfinite(cam[2])) {
      for (int a = 0; a < 3; ++a) c->pre.cam[i][a] = cam[a];
      c->have_cam[i] = true;
    }
  }
  c->have_calib[i] = true;
  return TSDF_OK;
}

// On the lane ahead the re-layout launch of a new frame also clears the brick counters the frame's clearOccupiedBricks() is going to use
// (it flips to them here instead): one launch and one dependent step less on the lane.
static uint32_t* counters_for_upload(tsdf_ctx* c, hipStream_t lane) {
  if (lane == c->stream || c->counters_zeroed || !c->d_counters[0]) return nullptr;
  if (!c->counters_flipped && c->counters_in_use) { c->counters_cur = alt_of(c->counters_cur); c->br.counters = c->d_counters[c->counters_cur]; }
  c->counters_flipped = true; c->counters_in_use = false; c->counters_zeroed = true; c->spare_clean = false;
  return c->br.counters;
}
// The frame slot a new frame is written to.  On the lane ahead: the OTHER slot (the context's stream may still read the current one for
// the previous frame), once per frame of the lane; c->frame then points at it, so everything queued from now on reads the new frame.
static int32_t begin_slot_write(tsdf_ctx* c, hipStream_t lane, bool keep_colour) {
  if (lane == c->stream || c->slot_flipped) return TSDF_OK;
  c->slot_flipped = true;
  if (!c->slot_in_use) return TSDF_OK;                                   // nothing queued reads the current slot (the first frame): in place
  c->slot_in_use = false;
  const int old = c->cur_slot, t = alt_of(old);
  if (int32_t rc = alloc_frame_slot(c, t)) return rc;
  if (keep_colour) {                                                     // "colour may be NULL (keeps the previous one)": the previous one lives in the other slot
    const size_t nc = (size_t)c->cfg.num_streams * c->frame.cw * c->frame.ch;
    HIP_TRY(c, hipMemcpyAsync(c->slots[t].color, c->slots[old].color, nc * sizeof(uchar4), hipMemcpyDeviceToDevice, lane));
  }
  use_frame_slot(c, t);
  return TSDF_OK;
}
int32_t tsdf_upload_frame(tsdf_ctx* c, const float* depth_rg, const float* quality, const float* silhouette, const uint8_t* colour) {
  CHECK_CTX(c);
  if (!depth_rg || !quality || !silhouette) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "depth, quality and silhouette are required");
  HIP_TRY(c, hipSetDevice(c->device));
  const hipStream_t lane = pre_enter(c);
  if (int32_t rc = begin_slot_write(c, lane, colour == nullptr)) return rc;
  const FrameImages& F = c->frame;
  const size_t np = (size_t)c->cfg.num_streams * F.w * F.h, nc = (size_t)c->cfg.num_streams * F.cw * F.ch;
  HIP_TRY(c, hipMemcpyAsync(c->d_stage_depth, depth_rg, np * 8, hipMemcpyHostToDevice, lane));
  HIP_TRY(c, hipMemcpyAsync(c->d_stage_q, quality, np * 4, hipMemcpyHostToDevice, lane));
  HIP_TRY(c, hipMemcpyAsync(c->d_stage_s, silhouette, np * 4, hipMemcpyHostToDevice, lane));
  if (colour) HIP_TRY(c, hipMemcpyAsync(c->d_stage_col, colour, nc * 3, hipMemcpyHostToDevice, lane));
  launch_pack_frame_fused(lane, c->d_stage_depth, c->d_stage_q, c->d_stage_s, (float4*)F.dqs, (float*)c->frame.depth, c->slots[c->cur_slot].ranges,
                          (int)c->cfg.num_streams, F.w, F.h, colour ? c->d_stage_col : nullptr, (uchar4*)F.color, nc, counters_for_upload(c, lane), (uint32_t)c->counter_words);
  HIP_TRY(c, hipGetLastError());
  c->slots[c->cur_slot].have = true;
  HIP_TRY(c, pre_leave(c, lane));
  return TSDF_OK;
}
// The same with the four arrays already in device memory (a producer on the GPU: the pre-processing of another library, a decoder, a
// staging buffer the caller DMA'd himself): no copy, one re-layout launch into a frame slot.  This is what a NEW frame costs the path
// itself (NetKinectArray hands integrate() a new frame every time, kinect_client.cpp:586-599).
//   flags 0                      the arrays were (or are being) written by work queued on the context's stream: the re-layout is ordered behind it
//   TSDF_FRAME_ARRAYS_COMPLETE   the arrays are complete now: the re-layout starts at once on the lane ahead, beside the previous frame's kernels
// Either way the arrays must stay untouched until work queued on the context's stream AFTER the next tsdf_integrate() / draw call runs.
int32_t tsdf_upload_frame_dev(tsdf_ctx* c, const float* depth_rg, const float* quality, const float* silhouette, const uint8_t* colour, uint32_t flags) {
  CHECK_CTX(c);
  if (!depth_rg || !quality || !silhouette) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "depth, quality and silhouette are required");
  if (((uintptr_t)depth_rg & 7u) || ((uintptr_t)quality & 3u) || ((uintptr_t)silhouette & 3u) || ((uintptr_t)colour & 3u))
    FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "device arrays must be aligned to their element size (depth 8 bytes, the others 4)");
  HIP_TRY(c, hipSetDevice(c->device));
  const hipStream_t lane = pre_enter(c);
  if (lane != c->stream && !(flags & TSDF_FRAME_ARRAYS_COMPLETE)) {      // the producer may be work on the context's stream: behind all of it
    if (!c->src_ready) HIP_TRY(c, hipEventCreateWithFlags(&c->src_ready, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->src_ready, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(lane, c->src_ready, 0));
  }
  if (int32_t rc = begin_slot_write(c, lane, colour == nullptr)) return rc;
  const FrameImages& F = c->frame;
  const size_t nc = (size_t)c->cfg.num_streams * F.cw * F.ch;
  timer_begin_on(c, "0repack", lane);
  launch_pack_frame_fused(lane, depth_rg, quality, silhouette, (float4*)F.dqs, (float*)c->frame.depth, c->slots[c->cur_slot].ranges,
                          (int)c->cfg.num_streams, F.w, F.h, colour, (uchar4*)F.color, nc, counters_for_upload(c, lane), (uint32_t)c->counter_words);
  timer_end_on(c, "0repack", lane);
  HIP_TRY(c, hipGetLastError());
  c->slots[c->cur_slot].have = true;
  HIP_TRY(c, pre_leave(c, lane));
  return TSDF_OK;
}

// ---- asynchronous frame upload: NetKinectArray's double PBO (framework/double_pixel_buffer.cpp:18-81; readLoop's memcpy into the
// mapped back buffer NetKinectArray.cpp:516-520; update() = swap + PBO -> texture DMA, :225-236) as two device frame slots, a pinned
// host staging ring and a copy stream.  While the path computes on the current slot the next frame travels into the other one.
static int32_t ensure_async_upload(tsdf_ctx* c) {
  if (c->copy_stream) return TSDF_OK;
  const size_t np = (size_t)c->cfg.num_streams * c->frame.w * c->frame.h, nc = (size_t)c->cfg.num_streams * c->frame.cw * c->frame.ch;
  const size_t bytes = np * 16 + nc * 3;
  for (int k = 0; k < 2; ++k) {                                            // (re-entered after a failed first attempt: keep what exists)
    if (!c->h_stage[k]) HIP_TRY(c, hipHostMalloc((void**)&c->h_stage[k], bytes, hipHostMallocDefault));
    if (!c->stage_done[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->stage_done[k], hipEventDisableTiming));
  }
  if (!c->d_astage) HIP_TRY(c, hipMalloc((void**)&c->d_astage, bytes));
  if (int32_t rc = alloc_frame_slot(c, 0)) return rc;
  if (int32_t rc = alloc_frame_slot(c, 1)) return rc;
  HIP_TRY(c, sync_ctx(c));                             // the new slot's colour memset
  HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  return TSDF_OK;
}
int32_t tsdf_frame_staging(tsdf_ctx* c, float** depth_rg, float** quality, float** silhouette, uint8_t** colour) {
  CHECK_CTX(c);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, block_pipeline(c));                                          // the caller manages the two frame slots himself from here on
  if (int32_t rc = ensure_async_upload(c)) return rc;
  const int k = c->stage_k;
  if (c->stage_busy[k]) { HIP_TRY(c, hipEventSynchronize(c->stage_done[k])); c->stage_busy[k] = false; }   // its last upload has left the buffer
  const size_t np = (size_t)c->cfg.num_streams * c->frame.w * c->frame.h;
  uint8_t* b = c->h_stage[k];
  if (depth_rg) *depth_rg = (float*)b;
  if (quality) *quality = (float*)(b + np * 8);
  if (silhouette) *silhouette = (float*)(b + np * 12);
  if (colour) *colour = b + np * 16;
  return TSDF_OK;
}
int32_t tsdf_upload_frame_async(tsdf_ctx* c, const float* depth_rg, const float* quality, const float* silhouette, const uint8_t* colour, int32_t with_colour) {
  CHECK_CTX(c);
  HIP_TRY(c, hipSetDevice(c->device));
  float *sd, *sq, *ss; uint8_t* sc;
  if (int32_t rc = tsdf_frame_staging(c, &sd, &sq, &ss, &sc)) return rc;   // (waits for the staging buffer)
  const size_t np = (size_t)c->cfg.num_streams * c->frame.w * c->frame.h, nc = (size_t)c->cfg.num_streams * c->frame.cw * c->frame.ch;
  // NULL (or the staging pointer itself): the producer wrote the staging buffer in place, as readLoop() writes the mapped PBO
  if (depth_rg && depth_rg != sd) memcpy(sd, depth_rg, np * 8);
  if (quality && quality != sq) memcpy(sq, quality, np * 4);
  if (silhouette && silhouette != ss) memcpy(ss, silhouette, np * 4);
  if (colour && colour != sc) { memcpy(sc, colour, nc * 3); with_colour = 1; }
  const int k = c->stage_k, t = alt_of(c->cur_slot);
  tsdf_ctx::FrameSlot& S = c->slots[t];
  if (S.in_use) HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, S.released, 0));   // the path's reads of slot t were all queued before `released`
  const size_t bytes = np * 16 + (with_colour ? nc * 3 : 0);
  HIP_TRY(c, hipMemcpyAsync(c->d_astage, c->h_stage[k], bytes, hipMemcpyHostToDevice, c->copy_stream));
  HIP_TRY(c, hipEventRecord(c->stage_done[k], c->copy_stream));
  c->stage_busy[k] = true; c->stage_k ^= 1;
  launch_pack_frame_fused(c->copy_stream, (const float*)c->d_astage, (const float*)(c->d_astage + np * 8), (const float*)(c->d_astage + np * 12), S.dqs, S.depth, S.ranges,
                          (int)c->cfg.num_streams, c->frame.w, c->frame.h, with_colour ? c->d_astage + np * 16 : nullptr, S.color, nc);
  HIP_TRY(c, hipEventRecord(S.ready, c->copy_stream));
  S.pending = true; S.have = true;
  HIP_TRY(c, hipGetLastError());
  return TSDF_OK;
}
int32_t tsdf_select_frame_slot(tsdf_ctx* c, uint32_t slot) {
  CHECK_CTX(c);
  if (slot > 1) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "frame slot must be 0 or 1");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, block_pipeline(c));                                          // the caller manages the two frame slots himself from here on
  if (int32_t rc = alloc_frame_slot(c, (int)slot)) return rc;
  tsdf_ctx::FrameSlot& N = c->slots[slot];
  if ((int)slot != c->cur_slot) {
    tsdf_ctx::FrameSlot& O = c->slots[c->cur_slot];
    if (O.released) { HIP_TRY(c, hipEventRecord(O.released, c->stream)); O.in_use = true; }
  }
  if (N.pending) { HIP_TRY(c, hipStreamWaitEvent(c->stream, N.ready, 0)); N.pending = false; }   // the path waits on the GPU, not on the host
  use_frame_slot(c, (int)slot);
  return TSDF_OK;
}
int32_t tsdf_current_frame_slot(const tsdf_ctx* c, uint32_t* slot) { CHECK_CTX(c); if (!slot) return TSDF_ERR_INVALID_ARGUMENT; *slot = (uint32_t)c->cur_slot; return TSDF_OK; }

// ---- image pre-processing (NetKinectArray::processTextures)
static int32_t ensure_pre_buffers(tsdf_ctx* c) {
  const FrameImages& F = c->frame;
  const size_t np = (size_t)c->cfg.num_streams * F.w * F.h;
  if (!c->d_raw) {
    HIP_TRY(c, hipMalloc(&c->d_raw, np * sizeof(float)));
    HIP_TRY(c, hipMalloc(&c->d_depth2, np * sizeof(float)));
    HIP_TRY(c, hipMalloc(&c->d_depth_rg, np * sizeof(float2)));
    HIP_TRY(c, hipMalloc(&c->d_lab, np * sizeof(float4)));
    HIP_TRY(c, hipMalloc(&c->d_depth_b, np * sizeof(float2)));
    HIP_TRY(c, hipMalloc(&c->d_normal, np * sizeof(float4)));
    const size_t blocks = (size_t)c->cfg.num_streams * ((F.w + 15) / 16) * ((F.h + 15) / 16);
    c->pre_cand_cap = (uint32_t)std::min<size_t>(blocks, 1024);
    if (const char* e = getenv("RR_TEST_PRE_CAND_CAP")) c->pre_cand_cap = (uint32_t)std::min<size_t>(blocks, (size_t)std::max(1, atoi(e)));   // test hook: a short list, so that candidate blocks overflow it
    HIP_TRY(c, hipMalloc(&c->d_pre_blocks, (1 + c->pre_cand_cap + blocks) * sizeof(uint32_t)));
    HIP_TRY(c, hipMemset(c->d_pre_blocks, 0, (1 + c->pre_cand_cap + blocks) * sizeof(uint32_t)));
  }
  return TSDF_OK;
}
// The raw frame (NetKinectArray::update(): depth + colour of every sensor) goes through the lane ahead like a pre-processed one does (round 4): its
// colour is re-laid out into the frame slot the lane writes, its depth is what tsdf_process_textures() -- on the same lane -- starts from.
int32_t tsdf_upload_raw_frame(tsdf_ctx* c, const float* depth_raw, const uint8_t* colour) {
  CHECK_CTX(c);
  if (!depth_raw || !colour) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "raw depth and colour are required");
  HIP_TRY(c, hipSetDevice(c->device));
  if (int32_t rc = ensure_pre_buffers(c)) return rc;
  const hipStream_t lane = pre_enter(c);
  if (int32_t rc = begin_slot_write(c, lane, false)) return rc;
  const FrameImages& F = c->frame;
  const size_t np = (size_t)c->cfg.num_streams * F.w * F.h, nc = (size_t)c->cfg.num_streams * F.cw * F.ch;
  HIP_TRY(c, hipMemcpyAsync(c->d_raw, depth_raw, np * sizeof(float), hipMemcpyHostToDevice, lane));
  HIP_TRY(c, hipMemcpyAsync(c->d_stage_col, colour, nc * 3, hipMemcpyHostToDevice, lane));
  c->pending_rgb = c->d_stage_col;                                       // its RGBA8 re-layout rides along in tsdf_process_textures' first launch
  c->raw_src = c->d_raw; c->have_raw = true; ++c->raw_generation;
  HIP_TRY(c, pre_leave(c, lane));
  return TSDF_OK;
}
// ... and with both arrays in device memory already (a decoder, a camera SDK's buffer): no copy -- the passes read depth_raw where it lies.  flags as
// for tsdf_upload_frame_dev; the arrays must stay untouched until work queued on the context's stream AFTER the next tsdf_integrate() / draw call runs.
static int32_t upload_raw_frame_dev_impl(tsdf_ctx* c, const float* depth_raw, const uint8_t* colour, uint32_t flags, bool defer_gate) {
  CHECK_CTX(c);
  if (!depth_raw || !colour) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "raw depth and colour are required");
  if (((uintptr_t)depth_raw & 3u) || ((uintptr_t)colour & 3u)) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "device arrays must be 4-byte aligned");
  HIP_TRY(c, hipSetDevice(c->device));
  if (int32_t rc = ensure_pre_buffers(c)) return rc;
  const hipStream_t lane = pre_enter(c, defer_gate);                      // (nothing below writes what the previous frames' draws read: flips of host pointers, a wait for the producer)
  if (lane != c->stream && !(flags & TSDF_FRAME_ARRAYS_COMPLETE)) {      // the producer may be work on the context's stream: behind all of it
    if (!c->src_ready) HIP_TRY(c, hipEventCreateWithFlags(&c->src_ready, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->src_ready, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(lane, c->src_ready, 0));
  }
  if (int32_t rc = begin_slot_write(c, lane, false)) return rc;
  c->pending_rgb = colour;                                               // its RGBA8 re-layout rides along in tsdf_process_textures' first launch
  c->raw_src = depth_raw; c->have_raw = true; ++c->raw_generation;
  HIP_TRY(c, pre_leave(c, lane));
  return TSDF_OK;
}
int32_t tsdf_upload_raw_frame_dev(tsdf_ctx* c, const float* depth_raw, const uint8_t* colour, uint32_t flags) { return upload_raw_frame_dev_impl(c, depth_raw, colour, flags, false); }
int32_t tsdf_set_depth_limits(tsdf_ctx* c, uint32_t i, float mn, float mx) {
  CHECK_CTX(c);
  if (i >= c->cfg.num_streams || !(mx > mn)) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "bad stream index or depth limits");
  c->pre.cv_min[i] = mn; c->pre.cv_max[i] = mx; c->have_limits[i] = true;
  return TSDF_OK;
}
int32_t tsdf_set_camera_position(tsdf_ctx* c, uint32_t i, const float xyz[3]) {
  CHECK_CTX(c);
  if (i >= c->cfg.num_streams || !xyz) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "bad stream index");
  for (int a = 0; a < 3; ++a) c->pre.cam[i][a] = xyz[a];
  c->have_cam[i] = true;
  return TSDF_OK;
}

// ---- frame ingest (NetKinectArray::init sizes :118-139, readLoop :482-529, update :225-236)
static void wire_sizes(const tsdf_ctx* c, uint32_t cf, uint32_t df, uint64_t* cs, uint64_t* ds) {
  const uint64_t cp = (uint64_t)c->frame.cw * c->frame.ch, dp = (uint64_t)c->frame.w * c->frame.h;
  *cs = cf == TSDF_COLOR_DXT1 ? cp * 4 / 8 : (cf == TSDF_COLOR_DXT5 ? cp : cp * 3);     // :118-130 (DXT5: the literal 307200 = 640*480)
  *ds = df == TSDF_DEPTH_U8 ? dp : dp * 4;                                               // :134-141
}
int32_t tsdf_set_wire_format(tsdf_ctx* c, uint32_t color_format, uint32_t depth_format) {
  CHECK_CTX(c);
  if (color_format != TSDF_COLOR_RGB8 && color_format != TSDF_COLOR_DXT1 && color_format != TSDF_COLOR_DXT5) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "colour format must be 0 (RGB8), 1 (DXT1) or 5 (DXT5)");
  if (depth_format != TSDF_DEPTH_F32 && depth_format != TSDF_DEPTH_U8) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "depth format must be 0 (float32) or 1 (8 bit)");
  uint64_t cs, ds;
  wire_sizes(c, color_format, depth_format, &cs, &ds);
  if ((cs & 15) || (ds & 15)) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "image sizes give records that are not 16-byte multiples (colour %llu, depth %llu bytes)", (unsigned long long)cs, (unsigned long long)ds);
  if (color_format != TSDF_COLOR_RGB8 && ((c->frame.cw | c->frame.ch) & 3)) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "DXT colour needs a resolution that is a multiple of 4 (the wire size w*h/2 assumes it)");
  c->color_format = color_format; c->depth_format = depth_format;
  return TSDF_OK;
}
int32_t tsdf_wire_sizes(tsdf_ctx* c, uint64_t* colorsize, uint64_t* depthsize, uint64_t* message_bytes) {
  CHECK_CTX(c);
  uint64_t cs, ds;
  wire_sizes(c, c->color_format, c->depth_format, &cs, &ds);
  if (colorsize) *colorsize = cs;
  if (depthsize) *depthsize = ds;
  if (message_bytes) *message_bytes = (cs + ds) * c->cfg.num_streams;
  return TSDF_OK;
}
int32_t tsdf_set_depth_compression(tsdf_ctx* c, uint32_t i, int32_t compressed, float near_, float far_) {
  CHECK_CTX(c);
  if (i >= c->cfg.num_streams) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "stream %u out of range", i);
  const float scale = far_ - near_;                                      // NetKinectArray.cpp:346-349
  c->pre.compress[i] = compressed != 0; c->pre.dc_near[i] = near_; c->pre.dc_scale[i] = scale; c->pre.dc_scaled_near[i] = scale / 255.0f;
  return TSDF_OK;
}
int32_t tsdf_upload_wire_frame(tsdf_ctx* c, const void* message, uint64_t bytes, double* timestamp) {
  CHECK_CTX(c);
  uint64_t cs, ds;
  wire_sizes(c, c->color_format, c->depth_format, &cs, &ds);
  const uint64_t want = (cs + ds) * c->cfg.num_streams;
  if (!message || bytes != want) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "wire message must be %llu bytes (%u x (colour %llu + depth %llu)), got %llu", (unsigned long long)want, c->cfg.num_streams, (unsigned long long)cs, (unsigned long long)ds, (unsigned long long)bytes);
  HIP_TRY(c, hipSetDevice(c->device));
  if (int32_t rc = ensure_pre_buffers(c)) return rc;
  if (c->wire_capacity < want) {
    HIP_TRY(c, sync_ctx(c));
    for (int k = 0; k < 2; ++k) {
      if (c->h_wire[k]) HIP_TRY(c, hipHostFree(c->h_wire[k]));
      c->h_wire[k] = nullptr; c->wire_pending[k] = false;
      HIP_TRY(c, hipHostMalloc((void**)&c->h_wire[k], want, hipHostMallocDefault));
      if (!c->wire_done[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->wire_done[k], hipEventDisableTiming));
    }
    HIP_TRY(c, hipFree(c->d_wire)); c->d_wire = nullptr;
    HIP_TRY(c, hipMalloc(&c->d_wire, want));
    c->wire_capacity = want;
  }
  // double buffer: the copy out of slot k may still be in flight from two frames ago
  const int k = c->wire_slot; c->wire_slot ^= 1;
  if (c->wire_pending[k]) HIP_TRY(c, hipEventSynchronize(c->wire_done[k]));
  memcpy(c->h_wire[k], message, want);                                   // readLoop's memcpy into the mapped PBO, :516-520
  if (timestamp) memcpy(timestamp, c->h_wire[k], sizeof(double));        // the first 8 bytes of the message, :510
  const hipStream_t lane = pre_enter(c);                                 // (round 4) the lane ahead: copy, unpack / DXT decode and the passes that follow
  if (int32_t rc = begin_slot_write(c, lane, false)) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->d_wire, c->h_wire[k], want, hipMemcpyHostToDevice, lane));
  HIP_TRY(c, hipEventRecord(c->wire_done[k], lane));
  c->wire_pending[k] = true;
  WireLayout L{};
  L.msg = c->d_wire; L.rec = (uint32_t)(cs + ds); L.cs = (uint32_t)cs; L.n = (int)c->cfg.num_streams;
  L.cw = c->frame.cw; L.ch = c->frame.ch; L.w = c->frame.w; L.h = c->frame.h;
  L.cfmt = (int)c->color_format; L.dfmt = (int)c->depth_format;
  timer_begin_on(c, "0ingest", lane);
  launch_wire_unpack(lane, L, (uchar4*)c->frame.color, c->d_raw);
  timer_end_on(c, "0ingest", lane);
  HIP_TRY(c, hipGetLastError());
  c->raw_src = c->d_raw; c->have_raw = true; ++c->raw_generation;
  HIP_TRY(c, pre_leave(c, lane));
  return TSDF_OK;
}
// the raw frame's colour waits for tsdf_process_textures' first launch (tsdf_ctx::pending_rgb): whoever reads the frame slot's colour before that asks for it here
static int32_t flush_pending_colour(tsdf_ctx* c) {
  if (!c->pending_rgb) return TSDF_OK;
  const hipStream_t lane = pre_enter(c);
  launch_pack_color(lane, c->pending_rgb, (uchar4*)c->frame.color, (size_t)c->cfg.num_streams * c->frame.cw * c->frame.ch);
  c->pending_rgb = nullptr;
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, pre_leave(c, lane));
  return TSDF_OK;
}
int32_t tsdf_download_raw_frame(tsdf_ctx* c, float* depth_raw, uint8_t* colour_rgba) {
  CHECK_CTX(c);
  if (!c->have_raw) FAIL(c, TSDF_ERR_STATE, "no raw frame uploaded");
  HIP_TRY(c, hipSetDevice(c->device));
  if (int32_t rc = flush_pending_colour(c)) return rc;
  HIP_TRY(c, sync_ctx(c));
  const size_t np = (size_t)c->cfg.num_streams * c->frame.w * c->frame.h, nc = (size_t)c->cfg.num_streams * c->frame.cw * c->frame.ch;
  if (depth_raw) HIP_TRY(c, hipMemcpy(depth_raw, c->raw_src, np * 4, hipMemcpyDeviceToHost));
  if (colour_rgba) HIP_TRY(c, hipMemcpy(colour_rgba, c->frame.color, nc * 4, hipMemcpyDeviceToHost));
  return TSDF_OK;
}
int32_t tsdf_set_preprocess(tsdf_ctx* c, int32_t filter_textures, int32_t processed_depth, int32_t refine) {
  CHECK_CTX(c);
  c->pre.filter_textures = filter_textures != 0; c->use_processed_depth = processed_depth != 0; c->pre.refine = refine != 0;
  return TSDF_OK;
}
static PreBuffers pre_buffers(tsdf_ctx* c) {
  PreBuffers B{};
  B.raw = c->raw_src; B.depth2 = c->d_depth2; B.fdepth = c->use_processed_depth ? c->d_depth2 : c->raw_src;
  B.depth_rg = c->d_depth_rg; B.lab = c->d_lab; B.depth_b = c->d_depth_b; B.normal = c->d_normal;
  B.dqs = (float4*)c->frame.dqs; B.depth_plane = (float*)c->frame.depth;
  B.cand_count = c->d_pre_blocks; B.cand_list = c->d_pre_blocks + 1; B.cand_cap = c->pre_cand_cap; B.blk_flag = c->d_pre_blocks + 1 + c->pre_cand_cap;
  return B;
}
// phase 0: the five passes.  Phases 1 and 2 are tsdf_frame_raw_dev's split of the same work (round 4): the morph and the filter pass read the raw frame and write
// only intermediate images of the lane's own, so they are queued IN FRONT of the lane's wait for the draws of two frames back (the two-frame cycle end of draw(f) ->
// preparation of f + 2 -> integrate(f + 2) -> draw(f + 2) that bounds the frame from raw data loses their 36 us); phase 2 -- behind that wait -- re-lays the colour out
// into the frame slot and runs the three passes that write what draws read (depth plane, packed texel, range cells, brick counters).
static int32_t process_textures_impl(tsdf_ctx* c, int phase) {
  CHECK_CTX(c);
  if (!c->have_raw) FAIL(c, TSDF_ERR_STATE, "no raw frame uploaded (tsdf_upload_raw_frame)");
  for (uint32_t i = 0; i < c->cfg.num_streams; ++i) {
    if (!c->have_calib[i] || !c->luts.s[i].xyz || !c->luts.s[i].uv) FAIL(c, TSDF_ERR_STATE, "stream %u needs cv_xyz and cv_uv (tsdf_set_calibration)", i);
    if (!c->have_limits[i] || !c->have_cam[i]) FAIL(c, TSDF_ERR_STATE, "stream %u needs tsdf_set_depth_limits and tsdf_set_camera_position", i);
  }
  HIP_TRY(c, hipSetDevice(c->device));
  // Round 4: on the lane ahead, like markBricks() -- the passes read the raw frame and write only what the lane owns (the frame slot it flipped to,
  // the brick counters clearOccupiedBricks() flipped to, its own intermediate images), so they run beside the previous frames' integrate and draw.
  const hipStream_t lane = pre_enter(c, phase == 1);
  if (int32_t rc = begin_slot_write(c, lane, true)) return rc;          // (already done by the raw upload of this frame; a re-run of an old raw frame takes the colour along)
  if (phase != 1 && c->normals_read_pending) {                           // the point / triangle-grid back-ends read the normal image this call rewrites
    if (lane != c->stream) HIP_TRY(c, hipStreamWaitEvent(lane, c->normals_read, 0));
    c->normals_read_pending = false;
  }
  PreParams& P = c->pre;
  P.W = c->frame.w; P.H = c->frame.h; P.N = (int)c->cfg.num_streams;
  for (int a = 0; a < 3; ++a) { P.bbox_min[a] = c->cfg.bbox_min[a]; P.bbox_max[a] = c->cfg.bbox_max[a]; }
  const PreBuffers B = pre_buffers(c);
  c->pre_generation = c->raw_generation; c->pre_processed_depth = c->use_processed_depth;
  if (phase != 2) timer_begin_on(c, "1preprocess", lane);
  const size_t ncol = (size_t)c->cfg.num_streams * c->frame.cw * c->frame.ch;
  float4* const ranges = c->slots[c->cur_slot].ranges;
  if (phase == 1) {
    launch_preprocess(lane, P, B, c->luts, c->frame, c->br, ranges, nullptr, nullptr, 0, nullptr, 0, 1);
    launch_preprocess(lane, P, B, c->luts, c->frame, c->br, ranges, nullptr, nullptr, 0, nullptr, 0, 2);
  } else if (phase == 2) {
    if (c->pending_rgb) launch_preprocess(lane, P, B, c->luts, c->frame, c->br, ranges, c->pending_rgb, (uchar4*)c->frame.color, ncol, nullptr, 0, 6);
    for (int k = 3; k <= 5; ++k) launch_preprocess(lane, P, B, c->luts, c->frame, c->br, ranges, nullptr, nullptr, 0, nullptr, 0, k);
  } else if (c->timers_on && c->timer_filter.find(",k_pre_") != std::string::npos) {   // each pass between its own pair of events, when the timer filter NAMES them (a pair costs the lane ~7 us)
    static const char* const names[5] = {"k_pre_morph", "k_pre_filter", "k_pre_boundary", "k_pre_normal", "k_pre_quality"};
    for (int k = 1; k <= 5; ++k) {
      timer_begin_on(c, names[k - 1], lane);
      launch_preprocess(lane, P, B, c->luts, c->frame, c->br, ranges, c->pending_rgb, (uchar4*)c->frame.color, ncol, nullptr, 0, k);
      timer_end_on(c, names[k - 1], lane);
    }
  } else launch_preprocess(lane, P, B, c->luts, c->frame, c->br, ranges, c->pending_rgb, (uchar4*)c->frame.color, ncol);
  HIP_TRY(c, hipGetLastError());
  if (phase != 1) {
    c->pending_rgb = nullptr;
    timer_end_on(c, "1preprocess", lane);
    c->slots[c->cur_slot].have = true;
  }
  HIP_TRY(c, pre_leave(c, lane));
  return TSDF_OK;
}
int32_t tsdf_process_textures(tsdf_ctx* c) { return process_textures_impl(c, 0); }
int32_t tsdf_download_preprocessed(tsdf_ctx* c, float* depth2, float* depth_rg, float* lab, float* depth_b, float* sil, float* normals, float* quality) {
  CHECK_CTX(c);
  if (!c->have_raw) FAIL(c, TSDF_ERR_STATE, "nothing was pre-processed yet");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_ctx(c));
  const size_t np = (size_t)c->cfg.num_streams * c->frame.w * c->frame.h;
  if (depth2) HIP_TRY(c, hipMemcpy(depth2, c->d_depth2, np * 4, hipMemcpyDeviceToHost));
  if (depth_rg) HIP_TRY(c, hipMemcpy(depth_rg, c->d_depth_rg, np * 8, hipMemcpyDeviceToHost));
  if (depth_b) HIP_TRY(c, hipMemcpy(depth_b, c->d_depth_b, np * 8, hipMemcpyDeviceToHost));
  std::vector<float4> tmp;
  auto fetch4 = [&](const void* src) -> int32_t { tmp.resize(np); HIP_TRY(c, hipMemcpy(tmp.data(), src, np * 16, hipMemcpyDeviceToHost)); return TSDF_OK; };
  int32_t rc;
  if (lab) {
    // the Lab image of the filter pass: the passes evaluate it only where the boundary pass reads it (k_pre_boundary); the whole image is produced here, from the
    // inputs of the frame that was processed -- which must still be the resident ones
    if (c->pre_generation != c->raw_generation || c->pre_processed_depth != c->use_processed_depth)
      FAIL(c, TSDF_ERR_STATE, "the Lab image is produced on request from the processed frame's inputs, and a newer raw frame has replaced them (download before the next upload)");
    launch_pre_lab(c->stream, c->pre, pre_buffers(c), c->luts, c->frame);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  if (lab) { if ((rc = fetch4(c->d_lab))) return rc; for (size_t i = 0; i < np; ++i) { lab[3 * i] = tmp[i].x; lab[3 * i + 1] = tmp[i].y; lab[3 * i + 2] = tmp[i].z; } }
  if (normals) { if ((rc = fetch4(c->d_normal))) return rc; for (size_t i = 0; i < np; ++i) { normals[3 * i] = tmp[i].x; normals[3 * i + 1] = tmp[i].y; normals[3 * i + 2] = tmp[i].z; } }
  if (sil || quality) { if ((rc = fetch4(c->frame.dqs))) return rc; for (size_t i = 0; i < np; ++i) { if (quality) quality[i] = tmp[i].y; if (sil) sil[i] = tmp[i].z; } }
  return TSDF_OK;
}

static int32_t require_inputs(tsdf_ctx* c, bool need_xyz, bool need_uv) {
  if (!c->slots[c->cur_slot].have) FAIL(c, TSDF_ERR_STATE, "no frame uploaded (tsdf_upload_frame)");
  for (uint32_t i = 0; i < c->cfg.num_streams; ++i) {
    if (!c->have_calib[i]) FAIL(c, TSDF_ERR_STATE, "stream %u has no calibration (tsdf_set_calibration)", i);
    if (need_xyz && !c->luts.s[i].xyz) FAIL(c, TSDF_ERR_STATE, "stream %u has no cv_xyz volume", i);
    if (need_uv && !c->luts.s[i].uv) FAIL(c, TSDF_ERR_STATE, "stream %u has no cv_uv volume", i);
  }
  return TSDF_OK;
}

int32_t tsdf_clear_bricks(tsdf_ctx* c) {
  CHECK_CTX(c);
  HIP_TRY(c, hipSetDevice(c->device));
  const hipStream_t lane = pre_enter(c);
  timer_begin_on(c, "bricks", lane);
  if (lane != c->stream) {                                               // the lane ahead: the other counter buffer (the previous frame's draw may still read this one)
    if (!c->counters_flipped && c->counters_in_use) { c->counters_cur = alt_of(c->counters_cur); c->br.counters = c->d_counters[c->counters_cur]; }
    c->counters_flipped = true; c->counters_in_use = false;
    c->spare_clean = false;
    if (c->counters_zeroed) c->counters_zeroed = false;                  // the frame's re-layout launch cleared them (a second clear of the frame fills again)
    else HIP_TRY(c, hipMemsetAsync(c->br.counters, 0, c->counter_words * sizeof(uint32_t), lane));
  } else if (c->spare_clean) {                                           // the other buffer was zeroed by the last integrate(): swap
    c->counters_cur = alt_of(c->counters_cur);
    c->br.counters = c->d_counters[c->counters_cur];
    c->spare_clean = false;
  } else HIP_TRY(c, hipMemsetAsync(c->br.counters, 0, c->counter_words * sizeof(uint32_t), c->stream));
  HIP_TRY(c, pre_leave(c, lane));
  return TSDF_OK;
}
int32_t tsdf_mark_bricks(tsdf_ctx* c) {
  CHECK_CTX(c);
  int32_t rc = require_inputs(c, true, false);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  const hipStream_t lane = pre_enter(c);
  uint32_t* zero_word = nullptr;
  if (lane != c->stream) {                                               // the count of the occupancy set the coming update flips to (or stays on)
    c->occ_zeroed_word = (!c->occ_flipped && c->occ_in_use) ? alt_of(c->occ_parity) : c->occ_parity;
    zero_word = c->d_occ_counts + c->occ_zeroed_word;
    c->occ_count_zeroed = true;
  }
  // the peel tiles the coming draw would reset first -- those the draw before the previous one touched, in the peel image that draw used (two alternate
  // while the lanes are on) --: reset here, on the lane ahead, beside the previous frame's kernels
  PeelClear pc{};
  if (lane != c->stream && c->use_bricks && c->skip_space && c->use_tile_history && c->tile_history && c->last_alt_peels && c->d_peels_alt) {
    pc.peels = (uint4*)c->d_peels_alt; pc.touched_prev = c->d_touched[(c->touched_idx + 1) % 3];
    pc.w = c->vw; pc.h = c->vh; pc.ntx = (c->vw + 7) / 8; pc.n_tiles = pc.ntx * ((c->vh + 7) / 8);
    c->peels_cleared = true;
  }
  launch_mark_bricks(lane, c->luts, c->frame, c->br, zero_word, pc.peels ? &pc : nullptr);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, pre_leave(c, lane));
  return TSDF_OK;
}
int32_t tsdf_update_occupied(tsdf_ctx* c, float* ratio) {
  CHECK_CTX(c);
  HIP_TRY(c, hipSetDevice(c->device));
  const hipStream_t lane = pre_enter(c);
  if (lane != c->stream) {
    // the lane ahead: the other occupancy set (flags, list, count) -- the previous frame's integrate / draw may still read this one --, its
    // count zeroed here instead of by the previous update (which would zero the word the context's stream is reading)
    if (!c->occ_flipped && c->occ_in_use) c->occ_parity = alt_of(c->occ_parity);
    c->occ_flipped = true; c->occ_in_use = false; c->occ_counts_stale = true;
    c->br.num_occupied = c->d_occ_counts + c->occ_parity; c->br.flags = c->d_flags[c->occ_parity]; c->br.occupied = c->d_occupied[c->occ_parity];
    // the frame's marking launch cleared a count word -- THE one this update lands on, unless a call between the two (a draw, tsdf_occupied_ratio, an
    // integrate on the context's stream) joined the lane and made this update flip after all: then, and for a second update of the frame, fill again
    const bool cleared = c->occ_count_zeroed && c->occ_zeroed_word == c->occ_parity;
    c->occ_count_zeroed = false;
    if (!cleared) HIP_TRY(c, hipMemsetAsync(c->br.num_occupied, 0, sizeof(uint32_t), lane));
    launch_update_occupied(lane, c->br, c->min_voxels, c->d_occ_counts + 2);          // (a third word takes the kernel's re-arming store)
  } else {
    c->occ_parity = alt_of(c->occ_parity);
    c->br.num_occupied = c->d_occ_counts + c->occ_parity;            // zero since the previous update (or creation) re-armed it ...
    if (c->occ_counts_stale) {                                           // ... unless the lane ahead has used the words in between (it zeroes its own and re-arms none)
      HIP_TRY(c, hipMemsetAsync(c->br.num_occupied, 0, sizeof(uint32_t), c->stream));
      c->occ_counts_stale = false;
    }
    c->br.flags = c->d_flags[c->occ_parity]; c->br.occupied = c->d_occupied[c->occ_parity];
    launch_update_occupied(c->stream, c->br, c->min_voxels, c->d_occ_counts + alt_of(c->occ_parity));
  }
  HIP_TRY(c, hipGetLastError());
  timer_end_on(c, "bricks", lane);
  HIP_TRY(c, pre_leave(c, lane));
  if (ratio) return tsdf_occupied_ratio(c, ratio);                     // the reference reads the count back every frame (:432-440); here only on request
  return TSDF_OK;
}
int32_t tsdf_occupied_ratio(tsdf_ctx* c, float* ratio) {
  CHECK_CTX(c);
  if (!ratio) return TSDF_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, join_pre(c));
  HIP_TRY(c, hipMemcpyAsync(c->h_num_occupied, c->br.num_occupied, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, sync_ctx(c));
  *ratio = (float)*c->h_num_occupied / (float)c->br.n;                 // :440
  return TSDF_OK;
}

int32_t tsdf_integrate(tsdf_ctx* c) {
  CHECK_CTX(c);
  int32_t rc = require_inputs(c, false, false);
  if (rc) return rc;
  if (c->vol.slot && !c->use_bricks) FAIL(c, TSDF_ERR_STATE, "a sparse tile pool needs brick culling (setUseBricks(true)): without it every tile is active");
  HIP_TRY(c, hipSetDevice(c->device));
  // the lane: the fourth one and the volume set the previous draw is NOT reading (stage overlap), or the context's stream
  hipStream_t lane = c->stream;
  if (deep_ok(c) && ensure_alt_set(c, c->integ_stream)) {
    lane = c->integ_stream;
    if (c->draw_unrecorded) {                                            // a draw without hole filling behind it: mark its end now
      if (c->fill_worker) c->fill_worker->wait_issued(c->draw_wait_job[c->vol_set]);
      HIP_TRY(c, hipEventRecord(c->draw_done[c->vol_set], c->stream));
      c->draw_pending[c->vol_set] = true; c->draw_unrecorded = false;
    }
    HIP_TRY(c, join_integ(c));                                           // (bookkeeping only: the lane is in order, and a draw has normally consumed it)
    swap_volume_set(c);
    if (c->draw_pending[c->vol_set]) { HIP_TRY(c, hipStreamWaitEvent(lane, c->draw_done[c->vol_set], 0)); c->draw_pending[c->vol_set] = false; }   // the draw two frames back read this set
    // the frame's images and brick state: from the lane ahead (both this lane and the context's stream wait for it), or from work on the context's stream
    c->main_since_gate = true;
    c->slot_in_use = c->counters_in_use = c->occ_in_use = true;
    if (c->pre_pending) {
      c->pre_pending = false;
      HIP_TRY(c, hipEventRecord(c->pre_done, c->pre_lane));
      if (c->pre_lane != lane) HIP_TRY(c, hipStreamWaitEvent(lane, c->pre_done, 0));
      HIP_TRY(c, hipStreamWaitEvent(c->stream, c->pre_done, 0));
    } else {
      HIP_TRY(c, hipEventRecord(c->integ_gate, c->stream));
      HIP_TRY(c, hipStreamWaitEvent(lane, c->integ_gate, 0));
    }
    c->integ_pending = true;
  } else {
    HIP_TRY(c, join_integ(c));
    HIP_TRY(c, join_pre(c));
  }
  const bool deep = lane != c->stream;
  timer_begin_on(c, "2integrate", lane);
  int lds = 2;
  for (uint32_t i = 0; i < c->cfg.num_streams; ++i) lds = std::min(lds, c->lds_ok[i]);
  const bool want_cache = lds >= 2 && c->k1_form_cap >= 3 && c->proj_budget > 0 && !c->vol.slot && !c->proj_failed;
  lds = std::min(lds, c->k1_form_cap);
  if (c->use_bricks) {
    // this frame's list / count, the previous integrate()'s (trusted unless something else may have written the volume)
    TileState& S = c->tiles;
    const int p = c->tile_parity;
    S.list = c->d_tile_list[p]; S.count = c->d_tile_counts + p;
    S.prev_list = c->d_tile_list[p ^ 1]; S.prev_count = c->d_tile_counts + (p ^ 1); S.next_count = c->d_tile_counts + (p ^ 1);
    if (++c->frame_stamp == 0) { c->frame_stamp = 1; c->full_classify = true; HIP_TRY(c, hipMemsetAsync(S.stamp, 0, (size_t)S.n * sizeof(uint32_t), lane)); }
  }
  // the draw that follows would first reset the peel tiles its predecessor touched: let the classify launch do it
  PeelClear pc{};
  {
    const bool whole = (c->vol.own_tz0 == 0 && c->vol.own_tz1 == (c->res[2] + 7) / 8);
    if (!deep && !pipelined(c) && c->use_bricks && !c->full_classify && c->skip_space && whole && c->use_tile_history && c->tile_history && !c->last_alt_peels && c->d_peels) {   // (with the lanes on the reset rides on the lane ahead: tsdf_mark_bricks)
      pc.peels = (uint4*)c->d_peels; pc.touched_prev = c->d_touched[(c->touched_idx + 2) % 3];     // the previous draw's tiles
      pc.w = c->vw; pc.h = c->vh; pc.ntx = (c->vw + 7) / 8; pc.n_tiles = pc.ntx * ((c->vh + 7) / 8);
      c->peels_cleared = true;
    }
  }
  if (c->use_bricks && !c->full_classify && !c->spare_clean && !pipelined(c)) {   // ... and zero the spare counter buffer for the next clearOccupiedBricks() (the lane ahead clears its own)
    pc.zero = c->d_counters[alt_of(c->counters_cur)]; pc.zero_words = (uint32_t)c->counter_words;
    c->spare_clean = true;
  }
  launch_integrate(lane, c->luts, c->frame, c->vol, c->br, c->tiles, c->use_bricks ? 1 : 0, lds, c->full_classify ? 1 : 0, c->frame_stamp, 1, &pc);
  // dense launches: the static half of the uniform-pair shortcut (k_integrate.hip), built once per calibration
  const float4* bounds = nullptr;
  const bool culled_ranges = c->use_bricks && c->culled_ranges && !c->vol.slot && (size_t)c->vol.n_stored_tiles * c->cfg.num_streams * 32 <= ((size_t)512 << 20);
  if ((!c->use_bricks || culled_ranges) && lds >= 2 && c->frame.ranges) {
    if (!c->d_tile_bounds) HIP_TRY(c, hipMalloc((void**)&c->d_tile_bounds, (size_t)c->vol.n_stored_tiles * c->cfg.num_streams * 2 * sizeof(float4)));
    if (!c->d_pair_masks) HIP_TRY(c, hipMalloc((void**)&c->d_pair_masks, (size_t)c->tiles.n * sizeof(uint32_t)));
    if (c->use_recs && !c->d_work_recs) HIP_TRY(c, hipMalloc((void**)&c->d_work_recs, (size_t)c->tiles.n * (sizeof(uint4) + 8 * sizeof(unsigned long long))));   // per tile: the 16-byte record, then (after all records) its 8 x 64 per-voxel bits
    if (!c->tile_bounds_valid) { launch_tile_bounds(lane, c->luts, c->vol, c->d_tile_bounds); c->tile_bounds_valid = true; }
    bounds = c->d_tile_bounds;
  }
  // projection cache: the pool and its tables, on the first integrate() that can use them.  Capacity = the budget, at most one slot per
  // stored tile; a failed allocation (another context holds the memory) just leaves this context on the LUT path
  const ProjCache* proj = nullptr;
  if (bounds && want_cache) {
    if (!c->proj.data) {
      int dz_max = 1;
      for (uint32_t i = 0; i < c->cfg.num_streams; ++i) dz_max = std::max(dz_max, c->lut_dz[i]);
      const size_t slot_bytes = (size_t)c->cfg.num_streams * dz_max * 64 * 3 * sizeof(float);
      const size_t cap = std::min<size_t>((size_t)c->vol.n_stored_tiles, c->proj_budget / slot_bytes);
      bool ok = cap > 0;
      ok = ok && hipMalloc((void**)&c->proj.slot, (size_t)c->vol.n_stored_tiles * sizeof(uint32_t)) == hipSuccess;
      ok = ok && hipMalloc((void**)&c->proj.items, (size_t)c->tiles.n * sizeof(uint32_t)) == hipSuccess;
      ok = ok && hipMalloc((void**)&c->d_proj_words, 4 * sizeof(uint32_t)) == hipSuccess;
      ok = ok && hipMalloc((void**)&c->proj.data, cap * slot_bytes) == hipSuccess;
      if (!ok) {
        (void)hipGetLastError();
        hipFree(c->proj.slot); hipFree(c->proj.items); hipFree(c->d_proj_words); hipFree(c->proj.data);
        c->proj = ProjCache{}; c->d_proj_words = nullptr; c->proj_failed = true;
      } else {
        HIP_TRY(c, hipMemsetAsync(c->proj.slot, 0xff, (size_t)c->vol.n_stored_tiles * sizeof(uint32_t), c->stream));
        HIP_TRY(c, hipMemsetAsync(c->d_proj_words, 0, 4 * sizeof(uint32_t), c->stream));
        c->proj.cap = (uint32_t)cap; c->proj.slot_floats = (uint32_t)(slot_bytes / sizeof(float)); c->proj.alloc = c->d_proj_words; c->proj.dz_max = (uint32_t)dz_max;
        c->proj_parity = 0;
      }
    }
    if (c->proj.data) {
      for (uint32_t i = 0; i < c->cfg.num_streams; ++i) c->proj.inv_rz[i] = c->luts.s[i].inv_res[2];
      c->proj.n_slow = c->d_proj_words + 1 + c->proj_parity; c->proj.n_slow_next = c->d_proj_words + 1 + (c->proj_parity ^ 1);
      c->proj_parity ^= 1;
      proj = &c->proj;
    }
  }
  c->last_integrate_cached = proj != nullptr;
  if (bounds) {                                                        // this frame's (tile, stream) pair classes (+ which work items are cached)
    timer_begin_on(c, "k_pair_masks", lane);
    launch_integrate(lane, c->luts, c->frame, c->vol, c->br, c->tiles, c->use_bricks ? 1 : 0, lds, 0, c->frame_stamp, 3, nullptr, bounds, c->d_pair_masks, proj, c->use_recs ? c->d_work_recs : nullptr);
    timer_end_on(c, "k_pair_masks", lane);
  }
  timer_begin_on(c, "k_integrate_tiles", lane);                                // the kernel(s) alone (bench.py's roofline)
  launch_integrate(lane, c->luts, c->frame, c->vol, c->br, c->tiles, c->use_bricks ? 1 : 0, lds, 0, c->frame_stamp, 4, nullptr, bounds, bounds ? c->d_pair_masks : nullptr, proj, bounds && c->use_recs ? c->d_work_recs : nullptr);
  timer_end_on(c, "k_integrate_tiles", lane);
  if (c->use_bricks) { c->tile_parity ^= 1; c->full_classify = false; }
  else c->full_classify = true;                                       // a dense pass wrote every tile: the next culled frame must look at all of them
  timer_end_on(c, "2integrate", lane);
  HIP_TRY(c, hipGetLastError());
  return TSDF_OK;
}

static int32_t raymarch_impl(tsdf_ctx* c, const float* mv, const float* pr, bool outer_timer) {
  if (!mv || !pr) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "null matrix");
  int32_t rc = require_inputs(c, false, c->shade_mode != 3);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  ViewParams P;
  if (!make_view_params(c, mv, pr, &P)) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "singular modelview / projection matrix");
  HIP_TRY(c, join_pre(c));
  if (outer_timer) timer_begin(c, "3recon");
  const bool partial = !(c->vol.own_tz0 == 0 && c->vol.own_tz1 == (c->res[2] + 7) / 8);
  const bool shifted = P.vp_org[0] != 0 || P.vp_org[1] != 0 || P.vp_off[0] != 0.0f || P.vp_off[1] != 0.0f;
  if (shifted && partial) FAIL(c, TSDF_ERR_STATE, "a viewport origin / offset is not available on a slab context (the composite indexes pixels)");
  if (shifted) HIP_TRY(c, hipMemsetAsync(c->d_nsamples, 0, (size_t)c->vw * c->vh * sizeof(float), c->stream));   // clearImage of tex_num_samples, :207-208: the stores land at origin + pixel
  // (slab contexts too: a tile without a brick under it holds clear values after the march AND after a composite into this target --
  // the brick tables are replicated, so no rank can hit there)
  const bool use_tiles = P.skip && c->use_tile_history && !shifted && !masked_direct(c);
  if (P.skip) {
    timer_begin(c, "brickdraw");
    // two peel images alternate per tiled draw while the lanes are on (tsdf_ctx::d_peels_alt); a change of that mode starts a new tile history
    const bool alt_peels = use_tiles && pipelined(c) && c->d_peels_alt;
    if (alt_peels != c->last_alt_peels) c->tile_history = false;
    if (use_tiles && !c->tile_history) {
      const size_t n_img_tiles = (size_t)((c->vw + 7) / 8) * ((c->vh + 7) / 8);
      for (int k = 0; k < 3; ++k) HIP_TRY(c, hipMemsetAsync(c->d_touched[k], 0, n_img_tiles, c->stream));
      if (alt_peels) launch_clear_peels(c->stream, c->d_peels_alt, c->vw * c->vh);   // (the other image: all clear, like the one this draw resets in full)
      c->peels_cleared = false;
    }
    if (alt_peels && c->tile_history) std::swap(c->d_peels, c->d_peels_alt);          // the image of the draw before the previous one: its tiles are in the oldest mask
    c->last_alt_peels = alt_peels;
    launch_depth_limits(c->stream, P, c->br, c->d_peels, use_tiles ? c->d_touched[c->touched_idx] : nullptr,
                        use_tiles && c->tile_history ? c->d_touched[(c->touched_idx + (alt_peels ? 1 : 2)) % 3] : nullptr, use_tiles && c->tile_history && c->peels_cleared ? 1 : 0);
    timer_end(c, "brickdraw");
  }
  c->peels_cleared = false;                                              // consumed (or void: this draw did its own reset)
  // two pyramids alternate while the hole filling runs beside the next frame (stage overlap): this draw takes the other one and only has
  // to wait for the hole filling of the draw BEFORE the previous one -- long finished -- instead of the previous draw's
  const bool two_pyramids = c->fill_holes && c->overlap_fill && !masked_direct(c);
  if (two_pyramids) {
    const int p = c->atlas_parity ^ 1;
    if (!c->atlas_color[p]) {
      const size_t na = (size_t)c->atlas.aw * c->atlas.h;
      HIP_TRY(c, hipMalloc(&c->atlas_color[p], na * sizeof(float4)));
      HIP_TRY(c, hipMalloc(&c->atlas_depth[p], na * sizeof(float)));
      launch_clear_image(c->stream, c->atlas_color[p], c->atlas_depth[p], na, make_float4(0.0f, 1.0f, 0.0f, 0.0f), 1.0f);   // ViewLod::enable's clear
      c->tile_history = false;
    }
    c->atlas_parity = p;
    c->atlas.color = c->atlas_color[p]; c->atlas.depth = c->atlas_depth[p];
    HIP_TRY(c, join_fill_of(c, p));
  } else HIP_TRY(c, join_fill(c));                                       // the previous draw's hole filling still reads level 0 / writes the framebuffer
  if (!c->tile_history) c->tiled_draws = 0;
  RayTarget RT = ray_target(c);
  if (use_tiles) {
    const int cur = c->touched_idx, prev = (cur + 2) % 3, oldest = (cur + 1) % 3;
    RT.touched_cur = c->d_touched[cur]; RT.touched_prev = c->d_touched[prev];
    RT.touched_prev_target = c->d_touched[two_pyramids ? oldest : prev];
    RT.touched_recycle = c->d_touched[oldest];
    RT.rewrite_all = c->tiled_draws >= 1 ? 0 : 1;
    RT.rewrite_target = c->tiled_draws >= (two_pyramids ? 2 : 1) ? 0 : 1;
    // the hole filling of this draw may keep to the tiles of this draw and the two before, once three tiled draws in a row have left
    // nothing else in the pyramid it fills and in the framebuffer
    RT.fill_mask = c->fill_holes ? c->d_fill_mask[c->atlas_parity] : nullptr;
    c->draw_masks_valid = c->fill_holes && c->tiled_draws >= 2 && !partial;
    c->touched_idx = (cur + 1) % 3;
    c->tile_history = true;
    c->tiled_draws = std::min(2, c->tiled_draws + 1);
  } else { c->tile_history = false; c->draw_masks_valid = false; }
  if (!c->fill_holes) c->fb_consistent = false;                          // the march (or the masked merge below) writes the framebuffer itself
  HIP_TRY(c, join_integ(c));                                             // the volume: from here on (the depth limits above needed the bricks only)
  timer_begin(c, "draw");
  timer_begin(c, "k_march");
  launch_raymarch(c->stream, P, c->luts, c->frame, c->vol, RT, partial ? 1 : 0, c->d_hits, c->d_hit_counters, c->hit_parity, 2, c->d_long, c->march_cap ? c->march_cap : 0xffffffffu, c->march_box);
  timer_end(c, "k_march");
  launch_raymarch(c->stream, P, c->luts, c->frame, c->vol, RT, partial ? 1 : 0, c->d_hits, c->d_hit_counters, c->hit_parity, 3, c->d_long, c->march_cap ? c->march_cap : 0xffffffffu, c->march_box);
  c->hit_parity ^= 1;
  c->last_two_pass = !partial && P.skip && c->d_long && c->march_cap != 0;          // launch_raymarch's own condition
  c->own_miss_counts = true;
  if (masked_direct(c)) launch_resolve_masked(c->stream, c->atlas, c->vw, c->vh, c->d_fb_c, c->d_fb_d, (int)c->color_mask_mode, c->keep_color ? 1 : 0);
  timer_end(c, "draw");
  c->draw_unrecorded = c->integ_stream != nullptr;                       // (draw_done[set]: recorded by fillColors(), or by the next integrate())
  HIP_TRY(c, hipGetLastError());
  return TSDF_OK;
}
// ---- the point back-end behind the same Reconstruction interface (kinect::ReconPoints, recon_points.cpp)
int32_t tsdf_upload_normals(tsdf_ctx* c, const float* normals_rgb) {
  CHECK_CTX(c);
  if (!normals_rgb) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "null normals");
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t np = (size_t)c->cfg.num_streams * c->frame.w * c->frame.h;
  if (!c->d_normal) HIP_TRY(c, hipMalloc(&c->d_normal, np * sizeof(float4)));
  std::vector<float4> padded(np);
  for (size_t i = 0; i < np; ++i) padded[i] = make_float4(normals_rgb[3 * i], normals_rgb[3 * i + 1], normals_rgb[3 * i + 2], 0.0f);
  HIP_TRY(c, sync_ctx(c));
  HIP_TRY(c, hipMemcpy(c->d_normal, padded.data(), np * sizeof(float4), hipMemcpyHostToDevice));
  return TSDF_OK;
}
int32_t tsdf_view_matrices(const float* mv, const float* pr, uint32_t vw, uint32_t vh, const float* bbox_min, const float* bbox_max, float* out) {
  if (!mv || !pr || !bbox_min || !bbox_max || !out || !vw || !vh) return TSDF_ERR_INVALID_ARGUMENT;
  ViewParams P;
  Mat4 v2w;
  if (!view_matrices(bbox_min, bbox_max, (int)vw, (int)vh, mv, pr, &P, &v2w)) return TSDF_ERR_INVALID_ARGUMENT;
  memcpy(out, v2w.m, 64); memcpy(out + 16, P.img_to_eye.m, 64); memcpy(out + 32, P.normal.m, 64);
  for (int a = 0; a < 3; ++a) out[48 + a] = P.cam_vol[a];
  return TSDF_OK;
}

int32_t tsdf_draw_points(tsdf_ctx* c, const float* mv, const float* pr) {
  CHECK_CTX(c);
  if (!mv || !pr) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "null matrix");
  int32_t rc = require_inputs(c, true, true);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  ViewParams P;
  if (!make_view_params(c, mv, pr, &P)) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "singular modelview / projection matrix");
  PointParams Q{};
  double mvd[16], prd[16], pm[16];
  for (int i = 0; i < 16; ++i) { mvd[i] = mv[i]; prd[i] = pr[i]; }
  mat_mul_d(prd, mvd, pm);
  for (int i = 0; i < 16; ++i) Q.pmv.m[i] = (float)pm[i];
  for (int a = 0; a < 3; ++a) { Q.bbox_min[a] = c->cfg.bbox_min[a]; Q.bbox_max[a] = c->cfg.bbox_max[a]; }
  Q.normals = c->d_normal;
  if (!c->d_comp_key) HIP_TRY(c, hipMalloc(&c->d_comp_key, (size_t)c->vw * c->vh * sizeof(unsigned long long)));
  FrameImages F = c->frame;
  F.depth = (float*)c->frame.depth;
  HIP_TRY(c, join_pre(c));
  timer_begin(c, "points");
  HIP_TRY(c, join_fill(c));
  c->fb_consistent = false;
  launch_draw_points(c->stream, P, Q, c->luts, F, c->d_comp_key, c->d_fb_c, c->d_fb_d);
  timer_end(c, "points");
  if (c->d_normal && pipelined(c)) {                                     // the next tsdf_process_textures() on the lane ahead rewrites the normal image
    if (!c->normals_read) HIP_TRY(c, hipEventCreateWithFlags(&c->normals_read, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->normals_read, c->stream));
    c->normals_read_pending = true;
  }
  HIP_TRY(c, hipGetLastError());
  return TSDF_OK;
}
// kinect::ReconTrigrid (recon_trigrid.cpp)
int32_t tsdf_set_min_length(tsdf_ctx* c, float v) { CHECK_CTX(c); if (!(v > 0.0f)) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "min_length must be positive"); c->min_length = v; return TSDF_OK; }
int32_t tsdf_draw_trigrid(tsdf_ctx* c, const float* mv, const float* pr) {
  CHECK_CTX(c);
  if (!mv || !pr) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "null matrix");
  int32_t rc = require_inputs(c, true, true);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  ViewParams P;
  if (!make_view_params(c, mv, pr, &P)) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "singular modelview / projection matrix");
  PointParams Q{};
  double mvd[16], prd[16], pm[16];
  for (int i = 0; i < 16; ++i) { mvd[i] = mv[i]; prd[i] = pr[i]; }
  mat_mul_d(prd, mvd, pm);
  for (int i = 0; i < 16; ++i) Q.pmv.m[i] = (float)pm[i];
  for (int a = 0; a < 3; ++a) { Q.bbox_min[a] = c->cfg.bbox_min[a]; Q.bbox_max[a] = c->cfg.bbox_max[a]; }
  const size_t nv = (size_t)c->vw * c->vh;
  if (!c->d_tri_z) { HIP_TRY(c, hipMalloc(&c->d_tri_z, nv * sizeof(uint32_t))); HIP_TRY(c, hipMalloc(&c->d_tri_acc, nv * sizeof(float4))); }
  HIP_TRY(c, join_pre(c));
  timer_begin(c, "trigrid");
  HIP_TRY(c, join_fill(c));
  c->fb_consistent = false;
  launch_draw_trigrid(c->stream, P, Q, c->luts, c->frame, c->min_length, c->d_tri_z, c->d_tri_acc, c->d_fb_c, c->d_fb_d);
  timer_end(c, "trigrid");
  HIP_TRY(c, hipGetLastError());
  return TSDF_OK;
}
int32_t tsdf_raymarch(tsdf_ctx* c, const float* mv, const float* pr) {
  CHECK_CTX(c);
  int32_t rc = raymarch_impl(c, mv, pr, true);
  if (rc == TSDF_OK) timer_end(c, "3recon");
  return rc;
}
// fillColors(): with stage overlap on its own stream behind an event of the context's stream; *used = the stream it was queued on
static int32_t fill_colors_impl(tsdf_ctx* c, hipStream_t* used) {
  hipStream_t fs = c->stream;
  if (c->overlap_fill) {
    if (!c->fill_stream) {                                                // (a context created with RR_OVERLAP_FILL=0 and switched on later)
      HIP_TRY(c, hipStreamCreateWithFlags(&c->fill_stream, hipStreamNonBlocking));
      for (hipEvent_t& e : c->fill_done) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    for (hipEvent_t& e : c->draw_done) if (!e) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (c->fill_worker) c->fill_worker->wait_issued(c->draw_wait_job[c->vol_set]);   // (a job that waits for the event's PREVIOUS record must have issued that wait)
    HIP_TRY(c, hipEventRecord(c->draw_done[c->vol_set], c->stream));      // everything the caller queued so far: the march / composite into level 0
    c->draw_pending[c->vol_set] = true; c->draw_unrecorded = false;
    fs = c->fill_stream;
  }
  const bool by_tiles = c->fill_tiles && c->draw_masks_valid && c->fb_consistent && c->color_mask_mode == 0 && !c->keep_color;
  const uint8_t* tile_mask = by_tiles ? c->d_fill_mask[c->atlas_parity] : nullptr;
  ++c->n_fills; c->n_fills_by_tiles += by_tiles ? 1 : 0;
  c->fb_consistent = c->color_mask_mode == 0 && !c->keep_color;         // the framebuffer is this pass's now: background wherever no tile was dirty
  c->draw_masks_valid = false;                                           // (consumed: a second fillColors() of the same draw, e.g. after a composite, goes through every tile)
  if (c->overlap_fill && c->fill_thread && !c->timers_on) {              // the helper thread issues the lane's calls (tsdf_ctx::fill_worker)
    if (!c->fill_worker) {
      c->fill_worker = new tsdf_ctx::FillWorker();
      c->fill_worker->device = c->device; c->fill_worker->stream = c->fill_stream;
      c->fill_worker->th = std::thread([w = c->fill_worker] { w->run(); });
    }
    tsdf_ctx::FillWorker::Job j{c->atlas, tile_mask, {c->d_lvl_mask[0], c->d_lvl_mask[1]}, c->vw, c->vh, c->d_fb_c, c->d_fb_d, (int)c->color_mask_mode, c->keep_color ? 1 : 0,
                                c->draw_done[c->vol_set], c->fill_done[c->atlas_parity]};
    c->fill_job_no[c->atlas_parity] = c->draw_wait_job[c->vol_set] = c->fill_worker->submit(j);
    c->fill_pending[c->atlas_parity] = true;
    if (used) *used = c->fill_stream;
    return TSDF_OK;
  }
  if (c->fill_worker) { c->fill_worker->drain(); HIP_TRY(c, fill_worker_error(c)); }   // (earlier jobs first: the lane is in order)
  if (c->overlap_fill) HIP_TRY(c, hipStreamWaitEvent(c->fill_stream, c->draw_done[c->vol_set], 0));
  timer_begin_on(c, "holefill", fs);
  launch_inpaint_pyramid(fs, c->atlas, tile_mask, c->d_lvl_mask);
  launch_colorfill(fs, c->atlas, c->vw, c->vh, c->d_fb_c, c->d_fb_d, (int)c->color_mask_mode, c->keep_color ? 1 : 0, tile_mask);
  timer_end_on(c, "holefill", fs);
  if (c->overlap_fill) { HIP_TRY(c, hipEventRecord(c->fill_done[c->atlas_parity], fs)); c->fill_pending[c->atlas_parity] = true; }
  HIP_TRY(c, hipGetLastError());
  if (used) *used = fs;
  return TSDF_OK;
}
int32_t tsdf_fill_colors(tsdf_ctx* c) {
  CHECK_CTX(c);
  if (!c->fill_holes) FAIL(c, TSDF_ERR_STATE, "colour filling is off: the raymarch did not render into the pyramid");
  HIP_TRY(c, hipSetDevice(c->device));
  return fill_colors_impl(c, nullptr);
}
int32_t tsdf_draw_f(tsdf_ctx* c, const float* mv, const float* pr) {
  CHECK_CTX(c);
  int32_t rc = raymarch_impl(c, mv, pr, true);
  if (rc) return rc;
  hipStream_t last = c->stream;
  if (c->fill_holes && (rc = fill_colors_impl(c, &last))) return rc;
  timer_end_on(c, "3recon", last);
  return TSDF_OK;
}
int32_t tsdf_frame_dev(tsdf_ctx* c, const float* depth_rg, const float* quality, const float* silhouette, const uint8_t* colour, uint32_t flags, const float* mv, const float* pr) {
  CHECK_CTX(c);
  int32_t rc;
  static const bool prof = getenv("RR_HOST_PROFILE") != nullptr;          // (measurement hook: host time of each step, printed every 1000 frames)
  if (prof) {
    using clk = std::chrono::steady_clock;
    static double acc[6] = {0, 0, 0, 0, 0, 0}; static int n = 0;
    auto t0 = clk::now();
    auto lap = [&](int k) { auto t1 = clk::now(); acc[k] += std::chrono::duration<double, std::micro>(t1 - t0).count(); t0 = t1; };
    if (depth_rg && (rc = tsdf_upload_frame_dev(c, depth_rg, quality, silhouette, colour, flags))) return rc;
    lap(0);
    if ((rc = tsdf_clear_bricks(c))) return rc;
    if ((rc = tsdf_mark_bricks(c))) return rc;
    lap(1);
    if ((rc = tsdf_update_occupied(c, nullptr))) return rc;
    lap(2);
    if ((rc = tsdf_integrate(c))) return rc;
    lap(3);
    if ((rc = raymarch_impl(c, mv, pr, true))) return rc;
    lap(4);
    hipStream_t last = c->stream;
    if (c->fill_holes && (rc = fill_colors_impl(c, &last))) return rc;
    timer_end_on(c, "3recon", last);
    lap(5);
    if (++n % 1000 == 0) { fprintf(stderr, "host us per frame: upload %.1f clear+mark %.1f update %.1f integrate %.1f draw %.1f fill %.1f\n", acc[0] / 1000, acc[1] / 1000, acc[2] / 1000, acc[3] / 1000, acc[4] / 1000, acc[5] / 1000); for (double& a : acc) a = 0; }
    return TSDF_OK;
  }
  if (depth_rg && (rc = tsdf_upload_frame_dev(c, depth_rg, quality, silhouette, colour, flags))) return rc;
  if ((rc = tsdf_clear_bricks(c)) || (rc = tsdf_mark_bricks(c)) || (rc = tsdf_update_occupied(c, nullptr)) || (rc = tsdf_integrate(c))) return rc;
  return tsdf_draw_f(c, mv, pr);
}
// The same from the RAW frame: NetKinectArray::update() + processTextures() in front of the path (kinect_client.cpp:569-577)
int32_t tsdf_frame_raw_dev(tsdf_ctx* c, const float* depth_raw, const uint8_t* colour, uint32_t flags, const float* mv, const float* pr) {
  CHECK_CTX(c);
  int32_t rc;
  const bool split = depth_raw && rrhost::pipelined(c) && !(c->timers_on && c->timer_filter.find(",k_pre_") != std::string::npos);   // (the per-pass timers keep the five passes in one piece)
  if (depth_raw && (rc = upload_raw_frame_dev_impl(c, depth_raw, colour, flags, split))) return rc;
  if (split) {                                                           // morph + filter in front of the lane's gate, the rest behind it (process_textures_impl)
    if ((rc = process_textures_impl(c, 1)) || (rc = tsdf_clear_bricks(c)) || (rc = process_textures_impl(c, 2))) return rc;
  } else if ((rc = tsdf_clear_bricks(c)) || (rc = tsdf_process_textures(c))) return rc;
  if ((rc = tsdf_update_occupied(c, nullptr)) || (rc = tsdf_integrate(c))) return rc;
  return tsdf_draw_f(c, mv, pr);
}
int32_t tsdf_set_stage_overlap(tsdf_ctx* c, int32_t on) {
  CHECK_CTX(c);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_ctx(c));
  c->overlap_fill = on != 0;
  c->tile_history = false;                                               // one pyramid or two alternating: what the march targets hold changes
  return TSDF_OK;
}

// ---- setters
int32_t tsdf_set_tsdf_limit(tsdf_ctx* c, float limit) {
  CHECK_CTX(c);
  if (!(limit > 0.0f)) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "limit must be > 0");
  const bool whole = (c->vol.own_tz0 == 0 && c->vol.own_tz1 == (c->res[2] + 7) / 8);
  if (!whole && halo_layers_for(limit, c->res[2]) > c->halo_layers) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "limit needs a wider slab halo than this context allocated");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_ctx(c));
  c->vol.limit = limit;
  launch_mark_all_mixed(c->stream, c->tiles);    // the clear value changed: no tile is known to hold it
  c->full_classify = true;
  if (c->alt.data) {                                                     // ... in either volume set
    TileState other = c->tiles;
    other.cls = c->alt.cls_all + (size_t)(c->vol.int_tz0 - c->vol.tz0) * c->vol.nty * c->vol.ntx;
    launch_mark_all_mixed(c->stream, other);
    c->alt.full = true;
  }
  HIP_TRY(c, sync_ctx(c));
  return TSDF_OK;
}
// setVoxelSize(), recon_integration.cpp:340-353: res = ceil(bbox / size), a new volume, and the brick grid re-snapped to the new
// voxels from the CURRENT (already snapped) brick size -- the reference passes m_brick_size, not the originally requested value.
int32_t tsdf_set_voxel_size(tsdf_ctx* c, float size) {
  CHECK_CTX(c);
  if (!(size > 0.0f)) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "voxel size must be > 0");
  if (c->cfg.slab_z0 != 0 || c->cfg.slab_z1 != 0) FAIL(c, TSDF_ERR_STATE, "setVoxelSize on a slab context: the slab range is in voxel planes of the old grid (re-create the contexts)");
  int res[3];
  for (int a = 0; a < 3; ++a) {
    res[a] = (int)ceilf((c->cfg.bbox_max[a] - c->cfg.bbox_min[a]) / size);
    if (res[a] < 1 || res[a] > 4096) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "volume resolution out of range [1, 4096]");
  }
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_ctx(c));
  const int old_res[3] = {c->res[0], c->res[1], c->res[2]};
  const float old_vox[3] = {c->vox[0], c->vox[1], c->vox[2]};
  const float brick[3] = {c->br.size[0], c->br.size[1], c->br.size[2]};
  for (int a = 0; a < 3; ++a) { c->res[a] = res[a]; c->vox[a] = size; }
  int32_t rc = setup_volume(c);
  if (rc == TSDF_OK) rc = setup_bricks(c, brick);
  if (rc != TSDF_OK) {                                                   // leave a usable context behind: back to the old grid
    const std::string why = c->err;
    for (int a = 0; a < 3; ++a) { c->res[a] = old_res[a]; c->vox[a] = old_vox[a]; }
    if (setup_volume(c) == TSDF_OK) setup_bricks(c, brick);
    c->err = why;
    return rc;
  }
  for (uint32_t i = 0; i < c->cfg.num_streams; ++i) if (c->have_calib[i]) fit_lut_to_volume(c, i);
  HIP_TRY(c, sync_ctx(c));
  return TSDF_OK;
}
int32_t tsdf_set_use_bricks(tsdf_ctx* c, int32_t a) { CHECK_CTX(c); c->use_bricks = a != 0; return TSDF_OK; }
int32_t tsdf_set_space_skip(tsdf_ctx* c, int32_t a) { CHECK_CTX(c); c->skip_space = a != 0; return TSDF_OK; }
int32_t tsdf_set_color_filling(tsdf_ctx* c, int32_t a) { CHECK_CTX(c); c->fill_holes = a != 0; c->tile_history = false; return TSDF_OK; }   // the march target changes
int32_t tsdf_set_min_voxels_per_brick(tsdf_ctx* c, uint32_t n) { CHECK_CTX(c); c->min_voxels = n; return TSDF_OK; }
int32_t tsdf_set_shade_mode(tsdf_ctx* c, int32_t m) {
  CHECK_CTX(c);
  if (m < 0 || m > 3) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "shade mode must be 0..3");
  c->shade_mode = m;
  return TSDF_OK;
}
// Reconstruction::setViewportOffset -> uniform viewport_offset (recon_integration.cpp:527); the GL viewport origin is GL state the
// reference reads implicitly through gl_FragCoord (glViewport(x, y, ..), kinect_client.cpp:650,658)
int32_t tsdf_set_viewport_offset(tsdf_ctx* c, float x, float y) { CHECK_CTX(c); c->vp_off[0] = x; c->vp_off[1] = y; c->tile_history = false; return TSDF_OK; }
int32_t tsdf_set_viewport_origin(tsdf_ctx* c, int32_t x, int32_t y) {
  CHECK_CTX(c);
  if (x < -(1 << 20) || x > (1 << 20) || y < -(1 << 20) || y > (1 << 20)) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "viewport origin out of range");
  c->vp_org[0] = x; c->vp_org[1] = y; c->tile_history = false;
  return TSDF_OK;
}
// Reconstruction::setColorMaskMode (reconstruction.cpp:51-53; used recon_integration.cpp:212-216,321-333) and whether the client
// cleared the colour buffer before this draw (glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT), kinect_client.cpp:609-610,620) or
// only the depth buffer (the anaglyph's second eye, :627)
int32_t tsdf_set_color_mask_mode(tsdf_ctx* c, uint32_t mode) {
  CHECK_CTX(c);
  if (mode > 2) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "colour mask mode must be 0 (all), 1 (red) or 2 (green + blue)");
  c->color_mask_mode = mode; c->tile_history = false;
  return TSDF_OK;
}
int32_t tsdf_set_framebuffer_clear(tsdf_ctx* c, int32_t clear_color) { CHECK_CTX(c); c->keep_color = clear_color == 0; c->tile_history = false; return TSDF_OK; }
int32_t tsdf_set_brick_size(tsdf_ctx* c, const float size[3]) {
  CHECK_CTX(c);
  if (!size) return TSDF_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_ctx(c));
  return setup_bricks(c, size);
}
int32_t tsdf_resize(tsdf_ctx* c, uint32_t w, uint32_t h) {
  CHECK_CTX(c);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_ctx(c));
  int32_t rc = setup_view(c, w, h);
  if (rc) return rc;
  HIP_TRY(c, sync_ctx(c));
  return TSDF_OK;
}

// ---- getters
int32_t tsdf_get_resolution(const tsdf_ctx* c, uint32_t res[3], uint32_t rb[3], float bs[3]) {
  CHECK_CTX(c);
  for (int a = 0; a < 3; ++a) { if (res) res[a] = (uint32_t)c->res[a]; if (rb) rb[a] = (uint32_t)c->br.res[a]; if (bs) bs[a] = c->br.size[a]; }
  return TSDF_OK;
}
int32_t tsdf_num_bricks(const tsdf_ctx* c, uint32_t* n) { CHECK_CTX(c); if (!n) return TSDF_ERR_INVALID_ARGUMENT; *n = (uint32_t)c->br.n; return TSDF_OK; }
int32_t tsdf_num_lods(const tsdf_ctx* c, uint32_t* n) { CHECK_CTX(c); if (!n) return TSDF_ERR_INVALID_ARGUMENT; *n = (uint32_t)c->atlas.num_lods; return TSDF_OK; }

// ---- downloads / uploads
static int32_t need_linear(tsdf_ctx* c) {
  if (!c->d_linear) HIP_TRY(c, hipMalloc(&c->d_linear, (size_t)c->res[0] * c->res[1] * c->res[2] * sizeof(float)));
  return TSDF_OK;
}
int32_t tsdf_download_volume(tsdf_ctx* c, float* out) {
  CHECK_CTX(c);
  if (!out) return TSDF_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  int32_t rc = need_linear(c);
  if (rc) return rc;
  HIP_TRY(c, join_integ(c));
  launch_volume_to_linear(c->stream, c->vol, c->d_linear);
  HIP_TRY(c, hipMemcpyAsync(out, c->d_linear, (size_t)c->res[0] * c->res[1] * c->res[2] * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, sync_ctx(c));
  return TSDF_OK;
}
int32_t tsdf_upload_volume(tsdf_ctx* c, const float* in) {
  CHECK_CTX(c);
  if (!in) return TSDF_ERR_INVALID_ARGUMENT;
  if (c->vol.slot) FAIL(c, TSDF_ERR_STATE, "tsdf_upload_volume is not available with a sparse tile pool (tiles get storage from integrate())");
  HIP_TRY(c, hipSetDevice(c->device));
  int32_t rc = need_linear(c);
  if (rc) return rc;
  HIP_TRY(c, sync_ctx(c));
  HIP_TRY(c, hipMemcpyAsync(c->d_linear, in, (size_t)c->res[0] * c->res[1] * c->res[2] * sizeof(float), hipMemcpyHostToDevice, c->stream));
  launch_volume_from_linear(c->stream, c->vol, c->d_linear);
  launch_mark_all_mixed(c->stream, c->tiles);
  c->full_classify = true;
  HIP_TRY(c, sync_ctx(c));
  return TSDF_OK;
}
// the storage tiles the last culled integrate() computed (its compacted work list): ids are x-fastest tile indices relative to
// the first integrated tile layer; grid = {tiles along x, tiles along y, first integrated tile layer, integrated tiles}
int32_t tsdf_download_active_tiles(tsdf_ctx* c, uint32_t* ids, uint32_t capacity, uint32_t* count, uint32_t grid[4]) {
  CHECK_CTX(c);
  if (!count) return TSDF_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_ctx(c));
  const int p = c->tile_parity ^ 1;                                       // integrate() flipped the parity after its launches
  uint32_t n = 0;
  HIP_TRY(c, hipMemcpy(&n, c->d_tile_counts + p, sizeof(uint32_t), hipMemcpyDeviceToHost));
  *count = n;
  if (grid) { grid[0] = (uint32_t)c->vol.ntx; grid[1] = (uint32_t)c->vol.nty; grid[2] = (uint32_t)c->vol.int_tz0; grid[3] = (uint32_t)c->tiles.n; }
  if (ids && capacity) HIP_TRY(c, hipMemcpy(ids, c->d_tile_list[p], (size_t)std::min(n, capacity) * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return TSDF_OK;
}
int32_t tsdf_download_bricks(tsdf_ctx* c, uint32_t* counters, uint8_t* flags) {
  CHECK_CTX(c);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_ctx(c));
  if (counters) HIP_TRY(c, hipMemcpy(counters, c->br.counters, (size_t)c->br.n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (flags) HIP_TRY(c, hipMemcpy(flags, c->br.flags, (size_t)c->br.n, hipMemcpyDeviceToHost));
  return TSDF_OK;
}
int32_t tsdf_upload_brick_counters(tsdf_ctx* c, const uint32_t* counters) {
  CHECK_CTX(c);
  if (!counters) return TSDF_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_ctx(c));
  HIP_TRY(c, hipMemcpy(c->br.counters, counters, (size_t)c->br.n * sizeof(uint32_t), hipMemcpyHostToDevice));
  return TSDF_OK;
}
int32_t tsdf_download_image(tsdf_ctx* c, float* rgba, float* depth, float* ns, float* peels) {
  CHECK_CTX(c);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_ctx(c));
  const RayTarget R = ray_target(c);
  const size_t w = (size_t)c->vw, h = (size_t)c->vh;
  if (rgba) HIP_TRY(c, hipMemcpy2D(rgba, w * 16, R.color, (size_t)R.stride * 16, w * 16, h, hipMemcpyDeviceToHost));
  if (depth) HIP_TRY(c, hipMemcpy2D(depth, w * 4, R.depth, (size_t)R.stride * 4, w * 4, h, hipMemcpyDeviceToHost));
  if (ns) HIP_TRY(c, hipMemcpy(ns, c->d_nsamples, w * h * 4, hipMemcpyDeviceToHost));
  if (peels) {
    HIP_TRY(c, hipMemcpy(peels, c->d_peels, w * h * 16, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < w * h; ++i) { peels[4 * i + 1] = -peels[4 * i + 1]; peels[4 * i + 3] = 0.0f; }   // device keeps max z; reference keeps min(-z)
  }
  return TSDF_OK;
}
int32_t tsdf_upload_image(tsdf_ctx* c, const float* rgba, const float* depth) {
  CHECK_CTX(c);
  if (!rgba || !depth) return TSDF_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_ctx(c));
  const RayTarget R = ray_target(c);
  const size_t w = (size_t)c->vw, h = (size_t)c->vh;
  HIP_TRY(c, hipMemcpy2D(R.color, (size_t)R.stride * 16, rgba, w * 16, w * 16, h, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy2D(R.depth, (size_t)R.stride * 4, depth, w * 4, w * 4, h, hipMemcpyHostToDevice));
  c->tile_history = false; c->draw_masks_valid = false;                  // the march target no longer holds what the last march left
  return TSDF_OK;
}
int32_t tsdf_download_framebuffer(tsdf_ctx* c, float* rgba, float* depth) {
  CHECK_CTX(c);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_ctx(c));
  const size_t n = (size_t)c->vw * c->vh;
  if (rgba) HIP_TRY(c, hipMemcpy(rgba, c->d_fb_c, n * 16, hipMemcpyDeviceToHost));
  if (depth) HIP_TRY(c, hipMemcpy(depth, c->d_fb_d, n * 4, hipMemcpyDeviceToHost));
  return TSDF_OK;
}
int32_t tsdf_download_atlas(tsdf_ctx* c, float* rgba, float* depth) {
  CHECK_CTX(c);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_ctx(c));
  const size_t n = (size_t)c->atlas.aw * c->atlas.h;
  if (rgba) HIP_TRY(c, hipMemcpy(rgba, c->atlas.color, n * 16, hipMemcpyDeviceToHost));
  if (depth) HIP_TRY(c, hipMemcpy(depth, c->atlas.depth, n * 4, hipMemcpyDeviceToHost));
  return TSDF_OK;
}

// ---- multi-GPU hooks
int32_t tsdf_halo_info(const tsdf_ctx* c, uint32_t* layers, uint64_t* bytes) {
  CHECK_CTX(c);
  if (layers) *layers = (uint32_t)c->halo_layers;
  if (bytes) *bytes = (uint64_t)c->halo_layers * c->vol.nty * c->vol.ntx * TILE_VOX * sizeof(float);
  return TSDF_OK;
}
int32_t tsdf_halo_pack_dev(tsdf_ctx* c, void* lo, void* hi) {
  CHECK_CTX(c);
  if (c->vol.slot) FAIL(c, TSDF_ERR_STATE, "halo exchange is not available with a sparse tile pool (use slab_recompute_halo)");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, join_integ(c));                                             // (the volume: behind an integrate() in flight on the integrate lane)
  const Volume& V = c->vol;
  const size_t layer = (size_t)V.nty * V.ntx * TILE_VOX, n = (size_t)c->halo_layers * layer * sizeof(float);
  if (lo) HIP_TRY(c, hipMemcpyAsync(lo, V.data + (size_t)(V.own_tz0 - V.tz0) * layer, n, hipMemcpyDeviceToDevice, c->stream));
  if (hi) HIP_TRY(c, hipMemcpyAsync(hi, V.data + (size_t)(V.own_tz1 - c->halo_layers - V.tz0) * layer, n, hipMemcpyDeviceToDevice, c->stream));
  return TSDF_OK;
}
int32_t tsdf_halo_unpack_dev(tsdf_ctx* c, const void* below, const void* above) {
  CHECK_CTX(c);
  if (c->vol.slot) FAIL(c, TSDF_ERR_STATE, "halo exchange is not available with a sparse tile pool (use slab_recompute_halo)");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, join_integ(c));
  const Volume& V = c->vol;
  const size_t layer = (size_t)V.nty * V.ntx * TILE_VOX;
  if (below && V.tz0 < V.own_tz0) {
    const int have = V.own_tz0 - V.tz0;        // < halo_layers only at the volume boundary: take the top `have` layers
    HIP_TRY(c, hipMemcpyAsync(V.data, (const float*)below + (size_t)(c->halo_layers - have) * layer, (size_t)have * layer * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  }
  if (above && V.tz1 > V.own_tz1) {
    const int have = V.tz1 - V.own_tz1;
    HIP_TRY(c, hipMemcpyAsync(V.data + (size_t)(V.own_tz1 - V.tz0) * layer, above, (size_t)have * layer * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  }
  return TSDF_OK;
}
int32_t tsdf_export_partial_dev(tsdf_ctx* c, void* dst) {
  CHECK_CTX(c);
  if (!dst) return TSDF_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, join_fill(c));
  launch_export_partial(c->stream, ray_target(c), c->vw, c->vh, dst);
  HIP_TRY(c, hipGetLastError());
  return TSDF_OK;
}
int32_t tsdf_composite_dev(tsdf_ctx* c, const void* gathered, uint32_t n) {
  CHECK_CTX(c);
  if (!gathered || n < 1) return TSDF_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, join_fill(c));
  c->draw_masks_valid = false;                                           // the composite writes every pixel of the march target
  launch_composite(c->stream, gathered, (int)n, ray_target(c), c->vw, c->vh);
  HIP_TRY(c, hipGetLastError());
  return TSDF_OK;
}

int32_t tsdf_export_hits_dev(tsdf_ctx* c, void* dst, uint32_t capacity) {
  CHECK_CTX(c);
  if (!dst || capacity < 1) return TSDF_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  // the raymarch that just ran used counter (hit_parity ^ 1): raymarch_impl flips the parity after its launch
  const int p = c->hit_parity ^ 1;
  HIP_TRY(c, join_fill(c));
  launch_export_hits(c->stream, ray_target(c), c->vw, c->d_hits, c->d_hit_counters + p, c->last_two_pass ? c->d_long : nullptr, c->d_hit_counters + 2 + p, dst, capacity);
  HIP_TRY(c, hipGetLastError());
  return TSDF_OK;
}
int32_t tsdf_composite_hits_dev(tsdf_ctx* c, const void* gathered, uint32_t n, uint64_t stride_bytes) {
  CHECK_CTX(c);
  if (!gathered || n < 1 || n > 32 || stride_bytes < 32) return TSDF_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->d_comp_key) HIP_TRY(c, hipMalloc(&c->d_comp_key, (size_t)c->vw * c->vh * sizeof(unsigned long long)));
  // a compositing context that did not march this frame (dedicated compositor, multigpu.py) has no miss counts of its own: 0 then
  HIP_TRY(c, join_fill(c));
  c->draw_masks_valid = false;                                           // the composite writes every pixel of the march target
  launch_composite_hits(c->stream, gathered, (size_t)stride_bytes, (int)n, ray_target(c), c->vw, c->vh, c->d_comp_key, c->own_miss_counts ? 1 : 0);
  HIP_TRY(c, hipGetLastError());
  return TSDF_OK;
}

int32_t tsdf_set_march_cap(tsdf_ctx* c, uint32_t samples) { CHECK_CTX(c); c->march_cap = samples; c->tile_history = false; return TSDF_OK; }

// ---- timers
int32_t tsdf_enable_timers(tsdf_ctx* c, int32_t a) { CHECK_CTX(c); c->timers_on = a != 0; return TSDF_OK; }
int32_t tsdf_set_timer_filter(tsdf_ctx* c, const char* names) {
  CHECK_CTX(c);
  c->timer_filter = (names && *names) ? "," + std::string(names) + "," : std::string();
  return TSDF_OK;
}
int32_t tsdf_timer_ms(tsdf_ctx* c, const char* name, float* ms) {
  CHECK_CTX(c);
  if (!name || !ms) return TSDF_ERR_INVALID_ARGUMENT;
  auto it = c->timers.find(name);
  if (it == c->timers.end() || it->second.used == 0) FAIL(c, TSDF_ERR_STATE, "timer '%s' has not run", name);
  auto& e = it->second.ev[it->second.used - 1];
  HIP_TRY(c, hipEventSynchronize(e.second));
  HIP_TRY(c, hipEventElapsedTime(ms, e.first, e.second));
  return TSDF_OK;
}
// event pairs are created on first use; creating them inside a measured loop costs host time per frame: make n available up front
int32_t tsdf_timer_reserve(tsdf_ctx* c, const char* name, uint32_t n) {
  CHECK_CTX(c);
  if (!name) return TSDF_ERR_INVALID_ARGUMENT;
  HIP_TRY(c, hipSetDevice(c->device));
  Timer& t = c->timers[name];
  while (t.ev.size() < std::min<size_t>(n, kMaxTimerPairs)) {
    hipEvent_t a, b;
    HIP_TRY(c, hipEventCreate(&a)); HIP_TRY(c, hipEventCreate(&b));
    t.ev.emplace_back(a, b);
  }
  return TSDF_OK;
}
// caller-defined intervals on the context's stream (bench.py brackets whole frames with them)
int32_t tsdf_timer_begin(tsdf_ctx* c, const char* name) { CHECK_CTX(c); if (!name) return TSDF_ERR_INVALID_ARGUMENT; HIP_TRY(c, hipSetDevice(c->device)); timer_begin(c, name); return TSDF_OK; }
int32_t tsdf_timer_end(tsdf_ctx* c, const char* name) { CHECK_CTX(c); if (!name) return TSDF_ERR_INVALID_ARGUMENT; timer_end(c, name); return TSDF_OK; }
// the end event behind the hole filling that is still in flight on its own stream (stage overlap): begin ... end_after_fill spans a whole
// frame from its first kernel to its framebuffer -- the frame's latency, where tsdf_timer_end would stop at the march
int32_t tsdf_timer_end_after_fill(tsdf_ctx* c, const char* name) {
  CHECK_CTX(c);
  if (!name) return TSDF_ERR_INVALID_ARGUMENT;
  if (c->fill_worker) c->fill_worker->drain();
  timer_end_on(c, name, (c->fill_pending[0] || c->fill_pending[1]) ? c->fill_stream : c->stream);
  return TSDF_OK;
}
// the individual samples recorded since the last tsdf_timer_stats / tsdf_timer_samples of this timer (and resets it)
int32_t tsdf_timer_samples(tsdf_ctx* c, const char* name, float* out_ms, uint32_t capacity, uint32_t* count) {
  CHECK_CTX(c);
  if (!name || !count || (!out_ms && capacity)) return TSDF_ERR_INVALID_ARGUMENT;
  *count = 0;
  auto it = c->timers.find(name);
  if (it == c->timers.end()) return TSDF_OK;
  Timer& t = it->second;
  const size_t n = std::min<size_t>(t.used, capacity);
  for (size_t i = 0; i < n; ++i) {
    HIP_TRY(c, hipEventSynchronize(t.ev[i].second));
    HIP_TRY(c, hipEventElapsedTime(&out_ms[i], t.ev[i].first, t.ev[i].second));
  }
  *count = (uint32_t)n;
  t.used = 0;
  return TSDF_OK;
}
int32_t tsdf_timer_spans(tsdf_ctx* c, const char* name, const char* origin, float* begin_ms, float* end_ms, uint32_t capacity, uint32_t* count) {
  CHECK_CTX(c);
  if (!name || !origin || !count || ((!begin_ms || !end_ms) && capacity)) return TSDF_ERR_INVALID_ARGUMENT;
  *count = 0;
  auto it = c->timers.find(name), io = c->timers.find(origin);
  if (it == c->timers.end() || io == c->timers.end() || io->second.used == 0) return TSDF_OK;
  Timer& t = it->second;
  const hipEvent_t zero = io->second.ev[0].first;
  const size_t n = std::min<size_t>(t.used, capacity);
  for (size_t i = 0; i < n; ++i) {
    HIP_TRY(c, hipEventSynchronize(t.ev[i].second));
    HIP_TRY(c, hipEventElapsedTime(&begin_ms[i], zero, t.ev[i].first));
    HIP_TRY(c, hipEventElapsedTime(&end_ms[i], zero, t.ev[i].second));
  }
  *count = (uint32_t)n;
  return TSDF_OK;
}
int32_t tsdf_timer_stats(tsdf_ctx* c, const char* name, uint32_t* count, float* total_ms) {
  CHECK_CTX(c);
  if (!name || !count || !total_ms) return TSDF_ERR_INVALID_ARGUMENT;
  *count = 0; *total_ms = 0.0f;
  auto it = c->timers.find(name);
  if (it == c->timers.end()) return TSDF_OK;
  Timer& t = it->second;
  double sum = 0.0;
  for (size_t i = 0; i < t.used; ++i) {
    float ms = 0.0f;
    HIP_TRY(c, hipEventSynchronize(t.ev[i].second));
    HIP_TRY(c, hipEventElapsedTime(&ms, t.ev[i].first, t.ev[i].second));
    sum += ms;
  }
  *count = (uint32_t)t.used; *total_ms = (float)sum;
  t.used = 0;
  return TSDF_OK;
}

}  // extern "C"
