// Native multi-GPU exchange behind the C ABI (SURVEY.md section 8b "tsdf_halo_exchange(ctx) / tsdf_composite(ctx) ... one context per rank
// sharing an RCCL communicator", section 8e).  The reference's caller is C++ (source/kinect_client.cpp:569-614): with these entry
// points it drives N GPUs itself -- one process (or thread) per GPU, one context each, RCCL over xGMI on the context's own stream:
//   tsdf_comm_unique_id / tsdf_comm_init / tsdf_comm_destroy   communicator life time (the caller carries the 128-byte id to the ranks)
//   tsdf_broadcast_frame    the frame arrives in ONE process (NetKinectArray's reader, NetKinectArray.cpp:482-529): broadcast of the four
//                           arrays as delivered, then every rank re-lays its copy out (tsdf_upload_frame_dev's launch)
//   tsdf_halo_exchange      all-gather of every slab's two boundary tile layers, before the raymarch
//   tsdf_composite_gather   hit records of every slab -> rank 0, nearest hit per pixel, hole filling; sizes without a host sync per frame
//   tsdf_composite_finish   before the result is read: repairs a gather that turned out too small
// RCCL is bound at run time (dlopen): a process that never calls tsdf_comm_* needs no librccl, and inside a torch process the copy torch
// has already loaded is the one used (two RCCLs in one process would each bring their own device state).
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstring>

#include <rccl/rccl.h>

#include "ctx.hpp"

using namespace rrhost;

namespace {
constexpr int kLag = 2;                       // frames between a hit count and its use as a gather size: its pinned copy has long arrived
constexpr int kRing = kLag + 1;

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};
// Thread-safe (C++11 magic static: the object is complete before any caller sees it; "one process (or thread) per GPU" makes concurrent first
// calls the normal case).  RR_TEST_NO_RCCL (test hook): behave as if the library could not be loaded.
static Rccl load_rccl() {
  Rccl R;
  if (getenv("RR_TEST_NO_RCCL")) { R.why = "librccl.so not loaded (RR_TEST_NO_RCCL is set)"; return R; }
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (int pass = 0; pass < 2 && !R.lib; ++pass)                        // first a copy that is already in the process (torch's), then the system's
    for (const char* n : names)
      if (!R.lib) R.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
  if (!R.lib) { const char* e = dlerror(); R.why = std::string("librccl.so not found: ") + (e ? e : "no loader message"); return R; }   // (dlerror() clears itself: one call)
  bool ok = true;
  auto sym = [&](const char* n) { void* p = dlsym(R.lib, n); if (!p) { ok = false; R.why = std::string("librccl lacks ") + n; } return p; };
  R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
  R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
  R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
  R.AllGather = (decltype(R.AllGather))sym("ncclAllGather");
  R.Broadcast = (decltype(R.Broadcast))sym("ncclBroadcast");
  R.Send = (decltype(R.Send))sym("ncclSend");
  R.Recv = (decltype(R.Recv))sym("ncclRecv");
  R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
  R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
  R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
  if (!ok) R.lib = nullptr;
  return R;
}
static Rccl& rccl_state() { static Rccl R = load_rccl(); return R; }
Rccl* rccl() { Rccl& R = rccl_state(); return R.lib ? &R : nullptr; }
const char* rccl_why() { return rccl_state().why.c_str(); }            // why rccl() is null

#define NCCL_TRY(c, expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) { FAIL(c, TSDF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, rccl()->GetErrorString(_r), __FILE__, __LINE__); } } while (0)
// inside ncclGroupStart / ncclGroupEnd: an error closes the group before it returns (an open group would swallow every later collective of the thread)
#define NCCL_TRY_IN_GROUP(c, expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) { rccl()->GroupEnd(); FAIL(c, TSDF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, rccl()->GetErrorString(_r), __FILE__, __LINE__); } } while (0)
#define NEED_COMM(c) do { CHECK_CTX(c); if (!(c)->comm.comm) FAIL(c, TSDF_ERR_STATE, "no communicator (tsdf_comm_init)"); } while (0)

bool is_worker(const tsdf_ctx* c) { return !(c->comm.dedicated && c->comm.rank == 0); }
int first_worker(const tsdf_ctx* c) { return c->comm.dedicated ? 1 : 0; }

void release_comm_buffers(tsdf_ctx* c) {
  tsdf_ctx::Comm& M = c->comm;
  hipFree(M.d_halo_send); hipFree(M.d_halo_gath); hipFree(M.d_hitbuf); hipFree(M.d_hitparts); hipFree(M.d_counts); hipFree(M.d_frame_stage);
  for (int k = 0; k < kRing; ++k) { if (M.h_counts[k]) hipHostFree(M.h_counts[k]); if (M.counts_evt[k]) hipEventDestroy(M.counts_evt[k]); }
  tsdf_ctx::Comm fresh;                                                 // the communicator, the roles and the statistics stay
  fresh.comm = M.comm; fresh.rank = M.rank; fresh.world = M.world; fresh.dedicated = M.dedicated;
  fresh.regathers = M.regathers; fresh.overflowed_frames = M.overflowed_frames; fresh.min_capacity = M.min_capacity; fresh.max_capacity = M.max_capacity;
  M = fresh;
}

// the all-gathered [written, hit] counts of frame f on the host (waits for their pinned copy: for f <= now - kLag it arrived long ago)
const int32_t* counts_of(tsdf_ctx* c, uint64_t f) {
  tsdf_ctx::Comm& M = c->comm;
  const int slot = (int)(f % kRing);
  if (M.counts_rec[slot]) hipEventSynchronize(M.counts_evt[slot]);
  return M.h_counts[slot];
}
uint32_t max_hits(tsdf_ctx* c, uint64_t f) {
  const int32_t* h = counts_of(c, f);
  int32_t m = 0;
  for (int r = 0; r < c->comm.world; ++r) m = std::max(m, h[2 * r + 1]);
  return (uint32_t)m;
}
void set_verdict(tsdf_ctx* c, uint64_t f, bool truncated) {
  tsdf_ctx::Comm& M = c->comm;
  const int k = (int)(f % tsdf_ctx::Comm::kVerdicts);
  M.verdict_frame[k] = f; M.verdict[k] = truncated ? 1 : 0; M.verdict_set[k] = true;
}
// records gathered per rank for frame f: 1.5 x the largest per-rank hit count of frame f - kLag (every rank computes the same number).
// The frame whose counts are read here was gathered with caps[...]: if it hit more rays than that and was not the latest frame when
// tsdf_composite_finish ran, it was composited from truncated lists -- counted, so that a caller can tell.
uint32_t capacity_for(tsdf_ctx* c, uint64_t f) {
  tsdf_ctx::Comm& M = c->comm;
  const uint32_t npx = M.max_capacity ? std::min(M.max_capacity, (uint32_t)(c->vw * c->vh)) : (uint32_t)(c->vw * c->vh);
  if (f < (uint64_t)kLag) return npx;                                   // no history yet: a slab cannot hit more rays than there are pixels
  const uint32_t m = max_hits(c, f - kLag);
  const bool truncated = m > M.caps[(f - kLag) % kRing];               // (tsdf_composite_finish raises the capacity of a frame it repairs)
  if (truncated) ++M.overflowed_frames;
  set_verdict(c, f - kLag, truncated);
  const uint32_t cap = std::max(M.min_capacity, ((m * 3u) / 2u + 1024u + 1023u) / 1024u * 1024u);
  return std::min(cap, npx);
}

// Rank 0's own records never travel: it exports them -- ALL of them, whatever the capacity of the gather -- straight into its row of the
// gather buffer, once per frame.  (They must not be exported again for a repeated gather: the first composite has overwritten the march
// target they are read from.)
int32_t exchange_hits(tsdf_ctx* c, uint32_t cap, bool first, uint64_t f) {
  tsdf_ctx::Comm& M = c->comm;
  Rccl* R = rccl();
  float* const own = M.rank == 0 ? M.d_hitparts : M.d_hitbuf;
  if (is_worker(c) && (first || M.rank != 0)) {                         // (a compositor's header stays {0 records, 0 hits})
    if (int32_t rc = tsdf_export_hits_dev(c, own, M.rank == 0 ? (uint32_t)(c->vw * c->vh) : cap)) return rc;
  }
  const bool record_counts = first;
  if (record_counts) {
    NCCL_TRY(c, R->AllGather(own, M.d_counts, 2, ncclInt32, (ncclComm_t)M.comm, c->stream));
    const int slot = (int)(f % kRing);
    HIP_TRY(c, hipMemcpyAsync(M.h_counts[slot], M.d_counts, (size_t)M.world * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipEventRecord(M.counts_evt[slot], c->stream));
    M.counts_rec[slot] = true;
    M.caps[slot] = cap;
  }
  const size_t n = 8 + (size_t)cap * 8;                                 // floats: header + records
  if (M.world > 1) {
    NCCL_TRY(c, R->GroupStart());
    if (M.rank == 0) {
      for (int r = 1; r < M.world; ++r) NCCL_TRY_IN_GROUP(c, R->Recv(M.d_hitparts + (size_t)r * M.hit_floats, n, ncclFloat, r, (ncclComm_t)M.comm, c->stream));
    } else {
      NCCL_TRY_IN_GROUP(c, R->Send(M.d_hitbuf, n, ncclFloat, 0, (ncclComm_t)M.comm, c->stream));
    }
    NCCL_TRY(c, R->GroupEnd());
  }
  if (M.rank == 0) {
    if (int32_t rc = tsdf_composite_hits_dev(c, M.d_hitparts, (uint32_t)M.world, (uint64_t)M.hit_floats * sizeof(float))) return rc;
    if (c->fill_holes) { if (int32_t rc = tsdf_fill_colors(c)) return rc; }
  }
  return TSDF_OK;
}
}  // namespace

extern "C" {

int32_t tsdf_comm_unique_id(uint8_t id[TSDF_COMM_ID_BYTES]) {
  if (!id) return TSDF_ERR_INVALID_ARGUMENT;
  static_assert(TSDF_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is RCCL's");
  Rccl* R = rccl();
  if (!R) { g_create_error = std::string("RCCL is not available in this process: ") + rccl_why(); return TSDF_ERR_STATE; }   // (no context here: tsdf_last_error(NULL) returns it)
  ncclUniqueId u;
  if (R->GetUniqueId(&u) != ncclSuccess) return TSDF_ERR_HIP;
  memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return TSDF_OK;
}

int32_t tsdf_comm_init(tsdf_ctx* c, const uint8_t id[TSDF_COMM_ID_BYTES], uint32_t rank, uint32_t world, uint32_t flags) {
  CHECK_CTX(c);
  if (!id || world < 1 || world > 32 || rank >= world) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "rank / world out of range (1 <= world <= 32)");
  const bool dedicated = (flags & TSDF_COMM_DEDICATED_COMPOSITOR) != 0;
  if (dedicated && world < 2) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "a dedicated compositor needs at least one worker rank");
  if (c->comm.comm) FAIL(c, TSDF_ERR_STATE, "the context already has a communicator (tsdf_comm_destroy first)");
  Rccl* R = rccl();
  if (!R) FAIL(c, TSDF_ERR_STATE, "RCCL is not available in this process: %s", rccl_why());
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, sync_ctx(c));
  ncclUniqueId u;
  memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  NCCL_TRY(c, R->CommInitRank(&comm, (int)world, u, (int)rank));        // (collective: every rank of the world calls it, each on its own device)
  tsdf_ctx::Comm& M = c->comm;
  M = tsdf_ctx::Comm{};
  M.comm = comm; M.rank = (int)rank; M.world = (int)world; M.dedicated = dedicated;
  return TSDF_OK;
}

int32_t tsdf_comm_destroy(tsdf_ctx* c) {
  CHECK_CTX(c);
  if (!c->comm.comm) return TSDF_OK;
  hipSetDevice(c->device);
  sync_ctx(c);
  release_comm_buffers(c);
  if (Rccl* R = rccl()) R->CommDestroy((ncclComm_t)c->comm.comm);
  c->comm = tsdf_ctx::Comm{};
  return TSDF_OK;
}

// bounds of the lagged guess of a frame's gather size: at least min_records (default 4096), at most max_records (0 = one per view pixel).  The
// result never depends on the guess -- tsdf_composite_finish repairs a gather that was too small --; tests force that path with a small maximum.
int32_t tsdf_comm_set_capacity_limits(tsdf_ctx* c, uint32_t min_records, uint32_t max_records) {
  NEED_COMM(c);
  if (min_records < 1 || (max_records && max_records < min_records)) return TSDF_ERR_INVALID_ARGUMENT;
  c->comm.min_capacity = min_records; c->comm.max_capacity = max_records;
  return TSDF_OK;
}
// Was frame `frame` (0 = the first tsdf_composite_gather of this communicator's buffers) composited from TRUNCATED record lists and left that way?
// A gather is sized from the hit counts of two frames earlier; a frame that hit more rays than that is repaired by tsdf_composite_finish only while
// it is the latest frame.  *truncated: 0 = complete (or repaired), 1 = pixels of that frame may be missing (a caller that kept it must redraw).
// The verdict of a frame is known once its counts have reached the host: at once for the latest three frames (the call waits for the pinned copy),
// from a ring of the last 64 frames otherwise; TSDF_ERR_STATE: not gathered yet, or too old.
int32_t tsdf_comm_frame_status(tsdf_ctx* c, uint64_t frame, int32_t* truncated) {
  NEED_COMM(c);
  if (!truncated) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "null argument");
  tsdf_ctx::Comm& M = c->comm;
  if (frame >= M.frame_no) FAIL(c, TSDF_ERR_STATE, "frame %llu has not been gathered yet (%llu frames so far)", (unsigned long long)frame, (unsigned long long)M.frame_no);
  const int k = (int)(frame % tsdf_ctx::Comm::kVerdicts);
  if (M.verdict_set[k] && M.verdict_frame[k] == frame) { *truncated = M.verdict[k]; return TSDF_OK; }
  if (frame + kRing >= M.frame_no) {                                     // its counts are still in the ring of pinned copies
    HIP_TRY(c, hipSetDevice(c->device));
    const bool t = max_hits(c, frame) > M.caps[frame % kRing];
    *truncated = t ? 1 : 0;                                              // (not stored: the latest frame may still be repaired by tsdf_composite_finish)
    return TSDF_OK;
  }
  FAIL(c, TSDF_ERR_STATE, "the verdict of frame %llu is no longer kept (the last %d frames are)", (unsigned long long)frame, tsdf_ctx::Comm::kVerdicts);
}
int32_t tsdf_comm_stats(tsdf_ctx* c, uint32_t* regathers, uint32_t* overflowed_frames) {
  NEED_COMM(c);
  if (regathers) *regathers = c->comm.regathers;
  if (overflowed_frames) *overflowed_frames = c->comm.overflowed_frames;
  return TSDF_OK;
}

// SURVEY.md section 8e "Per-frame images are replicated: uploaded to every rank, or ncclBroadcast from the receiving rank".  The root passes the
// frame's four arrays (host or device memory), the others NULL.
int32_t tsdf_broadcast_frame(tsdf_ctx* c, uint32_t root, const float* depth_rg, const float* quality, const float* silhouette, const uint8_t* colour) {
  NEED_COMM(c);
  tsdf_ctx::Comm& M = c->comm;
  if ((int)root >= M.world) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "root out of range");
  HIP_TRY(c, hipSetDevice(c->device));
  const FrameImages& F = c->frame;
  const size_t np = (size_t)c->cfg.num_streams * F.w * F.h, nc = (size_t)c->cfg.num_streams * F.cw * F.ch;
  const size_t bytes = np * 16 + ((nc * 3 + 15) & ~(size_t)15);         // [depth_rg 8 B][quality 4 B][silhouette 4 B] per pixel, then RGB8
  if (!M.d_frame_stage) { HIP_TRY(c, hipMalloc((void**)&M.d_frame_stage, bytes)); M.frame_stage_bytes = bytes; }
  uint8_t* st = (uint8_t*)M.d_frame_stage;
  if ((int)root == M.rank) {
    if (!depth_rg || !quality || !silhouette || !colour) FAIL(c, TSDF_ERR_INVALID_ARGUMENT, "the root passes all four arrays");
    // (hipMemcpyDefault: the root's arrays may lie in host memory -- pinned, for a copy that really is asynchronous -- or in device memory)
    HIP_TRY(c, hipMemcpyAsync(st, depth_rg, np * 8, hipMemcpyDefault, c->stream));
    HIP_TRY(c, hipMemcpyAsync(st + np * 8, quality, np * 4, hipMemcpyDefault, c->stream));
    HIP_TRY(c, hipMemcpyAsync(st + np * 12, silhouette, np * 4, hipMemcpyDefault, c->stream));
    HIP_TRY(c, hipMemcpyAsync(st + np * 16, colour, nc * 3, hipMemcpyDefault, c->stream));
  }
  if (M.world > 1) NCCL_TRY(c, rccl()->Broadcast(st, st, bytes, ncclUint8, (int)root, (ncclComm_t)M.comm, c->stream));
  if (!is_worker(c)) return TSDF_OK;                                    // a compositor without a slab reads no frame
  return tsdf_upload_frame_dev(c, (const float*)st, (const float*)(st + np * 8), (const float*)(st + np * 12), st + np * 16, 0);   // (behind the broadcast on the context's stream)
}

int32_t tsdf_halo_exchange(tsdf_ctx* c) {
  NEED_COMM(c);
  tsdf_ctx::Comm& M = c->comm;
  HIP_TRY(c, hipSetDevice(c->device));
  uint32_t layers = 0; uint64_t face_bytes = 0;
  tsdf_halo_info(c, &layers, &face_bytes);
  const size_t n = (size_t)(face_bytes / sizeof(float));
  if (!M.d_halo_send || M.halo_floats != n) {
    HIP_TRY(c, sync_ctx(c));
    hipFree(M.d_halo_send); hipFree(M.d_halo_gath); M.d_halo_send = M.d_halo_gath = nullptr;
    HIP_TRY(c, hipMalloc((void**)&M.d_halo_send, 2 * n * sizeof(float)));
    HIP_TRY(c, hipMalloc((void**)&M.d_halo_gath, (size_t)M.world * 2 * n * sizeof(float)));
    HIP_TRY(c, hipMemsetAsync(M.d_halo_send, 0, 2 * n * sizeof(float), c->stream));       // (a compositor sends these zeros: nobody reads them)
    M.halo_floats = n;
  }
  if (is_worker(c)) { if (int32_t rc = tsdf_halo_pack_dev(c, M.d_halo_send, M.d_halo_send + n)) return rc; }
  NCCL_TRY(c, rccl()->AllGather(M.d_halo_send, M.d_halo_gath, 2 * n, ncclFloat, (ncclComm_t)M.comm, c->stream));
  if (!is_worker(c)) return TSDF_OK;                                    // (took part in the collective; has no slab faces of its own)
  const float* below = M.rank > first_worker(c) ? M.d_halo_gath + ((size_t)(M.rank - 1) * 2 + 1) * n : nullptr;   // the lower neighbour's HIGH face
  const float* above = M.rank < M.world - 1 ? M.d_halo_gath + ((size_t)(M.rank + 1) * 2 + 0) * n : nullptr;       // the upper neighbour's LOW face
  return tsdf_halo_unpack_dev(c, below, above);
}

int32_t tsdf_composite_gather(tsdf_ctx* c) {
  NEED_COMM(c);
  tsdf_ctx::Comm& M = c->comm;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t npx = (size_t)c->vw * c->vh, hf = 8 + npx * 8;
  if (!(M.d_hitbuf || M.d_hitparts) || M.hit_floats != hf) {
    HIP_TRY(c, sync_ctx(c));
    release_comm_buffers(c);
    if (M.rank != 0) {
      HIP_TRY(c, hipMalloc((void**)&M.d_hitbuf, hf * sizeof(float)));
      HIP_TRY(c, hipMemsetAsync(M.d_hitbuf, 0, hf * sizeof(float), c->stream));
    }
    if (M.rank == 0) { HIP_TRY(c, hipMalloc((void**)&M.d_hitparts, (size_t)M.world * hf * sizeof(float))); HIP_TRY(c, hipMemsetAsync(M.d_hitparts, 0, (size_t)M.world * hf * sizeof(float), c->stream)); }
    HIP_TRY(c, hipMalloc((void**)&M.d_counts, (size_t)M.world * 2 * sizeof(int32_t)));
    for (int k = 0; k < kRing; ++k) {
      HIP_TRY(c, hipHostMalloc((void**)&M.h_counts[k], (size_t)M.world * 2 * sizeof(int32_t), hipHostMallocDefault));
      memset(M.h_counts[k], 0, (size_t)M.world * 2 * sizeof(int32_t));
      HIP_TRY(c, hipEventCreateWithFlags(&M.counts_evt[k], hipEventDisableTiming));
    }
    M.hit_floats = hf; M.frame_no = 0; M.have_last = false;
    for (bool& b : M.verdict_set) b = false;
  }
  const uint64_t f = M.frame_no;
  const uint32_t cap = capacity_for(c, f);
  if (int32_t rc = exchange_hits(c, cap, true, f)) return rc;
  M.have_last = true; M.last_frame = f; M.last_cap = cap;
  M.frame_no = f + 1;
  return TSDF_OK;
}

// Completes the latest frame: call before reading its result (and at the end of a timed region).  A COLLECTIVE when the gather of that
// frame turned out too small -- every rank sees the same counts and re-gathers together (the hit list stays valid until the next draw).
int32_t tsdf_composite_finish(tsdf_ctx* c, uint32_t* regathered) {
  NEED_COMM(c);
  tsdf_ctx::Comm& M = c->comm;
  HIP_TRY(c, hipSetDevice(c->device));
  if (regathered) *regathered = 0;
  if (M.have_last) {
    M.have_last = false;
    const uint32_t m = max_hits(c, M.last_frame);
    if (m > M.last_cap) {
      ++M.regathers;
      if (regathered) *regathered = 1;
      const uint32_t cap = std::min((uint32_t)(c->vw * c->vh), m);
      if (int32_t rc = exchange_hits(c, cap, false, M.last_frame)) return rc;
      M.caps[M.last_frame % kRing] = cap;
    }
    set_verdict(c, M.last_frame, false);                                 // complete: gathered in full, or repaired just now
  }
  HIP_TRY(c, sync_ctx(c));
  return TSDF_OK;
}

}  // extern "C"
