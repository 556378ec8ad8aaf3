// Host-side state of one context (private to the library: abi.cpp, comm.cpp).
#pragma once
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "tsdf_common.hpp"

using namespace rr;

extern thread_local std::string g_create_error;

// One named timer = a pool of event pairs, one pair per invocation since the last tsdf_timer_stats()
struct Timer { std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; size_t used = 0; bool open = false; };
constexpr size_t kMaxTimerPairs = 8192;

struct BrickRange { uint32_t lo[3], hi[3]; };

struct tsdf_ctx {
  tsdf_config cfg{};
  std::string err;
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  // volume
  int res[3]{};
  float vox[3]{};
  Volume vol{};
  TileState tiles{};
  uint8_t* d_cls_all = nullptr;      // tile class of every stored tile (owned + halo)
  int halo_layers = 0;
  // bricks
  float brick_req[3]{};          // requested size (setBrickSize argument)
  Bricks br{};
  std::vector<BrickRange> ranges;
  uint16_t* d_vox_first[3]{};
  uint8_t* d_vox_count[3]{};
  uint16_t* d_tile_b0[3]{};
  uint16_t* d_tile_b1[3]{};
  uint8_t* d_tile_full[3]{};
  uint16_t* d_brick_t0[3]{};
  uint16_t* d_brick_t1[3]{};
  // active-tile lists of this and the previous integrate() (k_classify_lists), their two device counts, and what decides
  // whether the next integrate() may trust the previous list
  uint32_t* d_tile_list[2]{}; uint32_t* d_tile_counts = nullptr; int tile_parity = 0; bool full_classify = true; uint32_t frame_stamp = 0;
  uint32_t* d_occ_counts = nullptr; int occ_parity = 0;   // two occupied-brick counts used alternately (see Bricks::num_occupied)
  uint32_t min_voxels = 10;      // recon_integration.cpp:59
  size_t counter_words = 0;
  // two counter buffers: while frame f uses one, integrate(f)'s classify launch zeroes the other (part D of k_classify_lists), and
  // clearOccupiedBricks() of frame f + 1 is a pointer swap instead of a fill launch; spare_clean says whether that happened
  uint32_t* d_counters[2]{}; int counters_cur = 0; bool spare_clean = false;
  uint32_t* h_num_occupied = nullptr;   // pinned
  // calibration + frame
  StreamTable luts{};
  void* lut_alloc[TSDF_MAX_STREAMS][3]{};   // per stream: cv_xyz_inv, cv_uv, cv_xyz device copies (freed when the stream is re-calibrated)
  bool have_calib[TSDF_MAX_STREAMS]{};
  // the stream's per-tile LUT box against the integrate kernel's LDS budget: 0 = does not fit (global-memory kernel), 1 = the box fits
  // (direct 8-tap form), 2 = the separable passes' rows and planes fit as well (the fastest form)
  int lut_dz[TSDF_MAX_STREAMS]{};   // the most z-planes of the stream's inverse LUT any 8^3 tile of the volume touches
  int lds_ok[TSDF_MAX_STREAMS]{};
  int k1_form_cap = 3;           // RR_K1_FORM: 3 = no cap (separable LDS form; + the projection cache where a budget was given), 2 = separable LDS form, 1 = direct 8-tap form, 0 = global-memory kernel
  // projection cache (ProjCache, tsdf_common.hpp): pool + slot table allocated by the first integrate() that can use it, dropped with
  // the volume, invalidated by tsdf_set_calibration
  ProjCache proj{}; uint32_t* d_proj_words = nullptr; int proj_parity = 0; size_t proj_budget = 0; bool proj_failed = false; uint32_t* d_item_stats = nullptr;
  bool last_integrate_cached = false;
  FrameImages frame{};           // the CURRENT frame slot's images (what mark / integrate / draw read)
  // Two frame slots (the reference's double PBO + texture arrays, NetKinectArray.cpp:225-236): while the path computes on slot
  // `cur_slot`, tsdf_upload_frame_async fills the other one on a copy stream; tsdf_select_frame_slot makes it current.
  struct FrameSlot { float4* dqs = nullptr; float* depth = nullptr; uchar4* color = nullptr; float4* ranges = nullptr; bool have = false;
                     hipEvent_t ready = nullptr; bool pending = false;      // recorded on the copy stream after the slot's upload + pack
                     hipEvent_t released = nullptr; bool in_use = false; }; // recorded on the compute stream when the slot stopped being current
  FrameSlot slots[2];
  int cur_slot = 0;
  hipStream_t copy_stream = nullptr;
  uint8_t* h_stage[2]{}; hipEvent_t stage_done[2]{}; bool stage_busy[2]{}; int stage_k = 0;   // pinned host staging ring of the async upload
  float* d_stage_depth = nullptr; float* d_stage_q = nullptr; float* d_stage_s = nullptr; uint8_t* d_stage_col = nullptr;
  uint8_t* d_astage = nullptr;   // device staging of the async upload (its own: the copy stream runs beside the compute stream)
  // pre-processing state (NetKinectArray side)
  PreParams pre{};
  float* d_raw = nullptr; float* d_depth2 = nullptr; float2* d_depth_rg = nullptr; float4* d_lab = nullptr; float2* d_depth_b = nullptr; float4* d_normal = nullptr;
  uint32_t* d_pre_blocks = nullptr; uint32_t pre_cand_cap = 0;   // [count | cand_list[cap] | blk_flag[blocks]] (PreBuffers)
  bool have_raw = false, use_processed_depth = true;
  const uint8_t* pending_rgb = nullptr;   // RGB8 colour of the raw frame uploaded last, still to be re-laid out into the frame slot (rides along in processTextures' first launch)
  uint64_t raw_generation = 0, pre_generation = 0; bool pre_processed_depth = true;   // which raw upload the products of processTextures() belong to (the Lab image is produced on request from its inputs)
  const float* raw_src = nullptr;   // the raw depth the passes read: d_raw (host upload, wire unpack) or the caller's device array (tsdf_upload_raw_frame_dev)
  hipEvent_t normals_read = nullptr; bool normals_read_pending = false;   // recorded behind a point / triangle-grid draw: the lane ahead rewrites d_normal
  bool have_limits[TSDF_MAX_STREAMS]{}, have_cam[TSDF_MAX_STREAMS]{};
  // frame ingest (readLoop / update): wire formats, pinned double buffer (the reference's double_pbo), device copy of the message
  uint32_t color_format = TSDF_COLOR_RGB8, depth_format = TSDF_DEPTH_F32;
  uint8_t* h_wire[2]{}; hipEvent_t wire_done[2]{}; bool wire_pending[2]{}; int wire_slot = 0;
  uint8_t* d_wire = nullptr; size_t wire_capacity = 0;
  // view
  int vw = 0, vh = 0;
  Atlas atlas{};
  float4* d_peels = nullptr; float* d_nsamples = nullptr;
  // While the lanes are on, TWO peel images alternate per (tiled) draw: the one the coming draw uses was last written two draws ago, so its touched tiles can
  // be reset on the lane ahead (a block range of the brick marking launch, k_mark_bricks) instead of by a launch of its own on the context's stream --
  // 12 + 6 us of the lane that bounds the frame.  d_peels is the latest draw's (what tsdf_download_image returns)
  float4* d_peels_alt = nullptr; bool last_alt_peels = false;
  void* d_hits = nullptr; uint32_t* d_hit_counters = nullptr; int hit_parity = 0;
  // image-space dirty tiles (k_raymarch.hip): three masks in rotation -- d_touched[touched_idx] is the coming draw's, (idx + 2) % 3 the
  // previous draw's (peels, sample counts), (idx + 1) % 3 the one before (the other pyramid when two alternate; recycled by the march).
  // The history is dropped whenever something else writes the march target; tiled_draws = consecutive draws since then (saturates at 2).
  uint8_t* d_touched[3]{}; int touched_idx = 0; bool tile_history = false; int tiled_draws = 0;
  // two pyramids, alternating per draw while stage overlap is on (round 3): the march of frame f + 1 writes level 0 of the OTHER one while
  // the hole filling of frame f still reads this one (the reference's m_view_inpaint / m_view_inpaint2, for another reason: it swaps them
  // between its transfer passes, recon_integration.cpp:279-338).  c->atlas points at the one the latest draw used.
  float4* atlas_color[2]{}; float* atlas_depth[2]{}; int atlas_parity = 0;
  // hole filling by dirty tiles (k_inpaint.hip): per pyramid the tile byte mask the march leaves, scratch masks of levels 1 / 2, and
  // whether the masks of the latest draw may be trusted / the framebuffer holds the background outside them
  uint8_t* d_fill_mask[2]{}; uint8_t* d_lvl_mask[2]{}; bool draw_masks_valid = false, fb_consistent = false, fill_tiles = true; uint64_t n_fills = 0, n_fills_by_tiles = 0;
  uint32_t* d_tri_z = nullptr; float4* d_tri_acc = nullptr; float min_length = 0.0125f;   // triangle-grid back-end; KinectCalibrationFile.cpp:96 default
  bool use_tile_history = true;   // RR_IMAGE_TILES=0 turns it off (A/B)
  bool peels_cleared = false;     // integrate() already reset the peel tiles the coming draw would reset (part C of k_classify_lists)
  uint32_t* d_pair_masks = nullptr;   // per work item of the integrate launch: this frame's (tile, stream) pair classes (k_pair_masks)
  uint4* d_work_recs = nullptr; bool use_recs = true;   // per work item of the integrate launch: the 16-byte record k_pair_masks leaves for k_integrate_tiles_rec (RR_K1_REC=0: off, A/B)
  float4* d_tile_bounds = nullptr; bool tile_bounds_valid = false;   // static per (stored tile, stream) LUT-box bounds, built on the first dense integrate after a calibration
  bool culled_ranges = true;      // RR_K1_CULLED_RANGES=0: no uniform-pair shortcut in culled launches (A/B hook; dense storage with a bounds table of at most 512 MiB only)
  bool use_ranges = true;         // RR_K1_RANGES=0: the dense integrate evaluates every voxel of every stream (A/B and test hook, read at creation)
  int march_box = 1;              // the dense march: 1 = LDS voxel boxes, 2 = two boxes per wave with the next one prefetched (round 4), 0 = gathers from global memory as in round 1; RR_MARCH_BOX overrides (A/B and test hook, read at creation)
  void* d_long = nullptr; uint32_t march_cap = 24;   // rays still running after march_cap samples go to the wave-per-ray pass (RR_MARCH_CAP, 0 = off)
  bool last_two_pass = false;     // the last march handed its long rays to the wave-per-ray pass (they are not on the hit list)
  bool own_miss_counts = false;   // this context has marched at this view size: its sample-count image holds the miss counts (-count, or count after a composite)
  unsigned long long* d_comp_key = nullptr;   // per-pixel bid of the compact composite (rank 0, allocated on first use)   // raymarch hit list (k_march -> k_shade)
  float4* d_fb_c = nullptr; float* d_fb_d = nullptr;
  float* d_linear = nullptr;     // scratch for volume up/download
  // flags (recon_integration.cpp:54-57)
  bool fill_holes = true, use_bricks = true, skip_space = true;
  int shade_mode = 0;
  // stereo modes of the client (source/kinect_client.cpp:616-669): viewport origin + viewport_offset uniform (side by side),
  // colour mask + "colour buffer not cleared before this draw" (anaglyph)
  int vp_org[2]{}; float vp_off[2]{};
  uint32_t color_mask_mode = 0; bool keep_color = false;
  // Stage overlap (round 3): the hole filling of draw f runs on a stream of its own beside whatever the caller queues next -- the brick
  // passes and the integrate of frame f + 1 do not touch the pyramid or the framebuffer --, tied to the context's stream by two events:
  // draw_done (below: the fill waits for the march) and fill_done (the next writer / reader of the pyramid or the framebuffer waits for it).
  hipStream_t fill_stream = nullptr; hipEvent_t fill_done[2] = {nullptr, nullptr}; bool fill_pending[2] = {false, false};   // (per pyramid)
  // The fill lane's calls are ISSUED by a helper thread (abi.cpp: FillWorker): issuing a c2 frame costs the calling thread ~100 us of HIP runtime
  // calls -- as long as the device needs for the frame --, 28 of them for the hole filling's 6 launches and 2 event operations, and nothing the caller
  // does next depends on them having been issued.  fillColors() records draw_done on the context's stream itself and hands the rest over as a job;
  // whoever needs fill_done[p] first waits (host side, spinning) until the helper has issued that job's record.  Off while timers are on (their
  // bookkeeping is the caller's thread's) and with RR_FILL_THREAD=0.
  struct FillWorker; FillWorker* fill_worker = nullptr; uint64_t fill_job_no[2] = {0, 0}, draw_wait_job[2] = {0, 0}; bool fill_thread = true;
  // ... and a fourth lane (round 3): integrate() of frame f + 1 on `integ_stream` beside the draw of frame f on the context's stream.  Everything
  // integrate() writes and the draw reads -- the volume, its tile classes, the lists / stamps / counts of the incremental classification --
  // exists twice and alternates per integrate(): `alt` holds the set not in use (allocated on the first such integrate), and each set evolves
  // exactly like the single volume of rounds 1 / 2, seeing every other frame (the reference rebuilds the TSDF from scratch every frame,
  // recon_integration.cpp:249-250: no state is carried from frame to frame).  Events: integ_done (the draw waits for its integrate),
  // draw_done[set] (recorded behind the draw that read the set: the hole filling waits for it, and so does the integrate two frames later
  // that overwrites the set), integ_gate (work queued on the context's stream that the lane must not overtake).  Whole-volume contexts with
  // dense storage only; off with stage overlap off, with the projection cache, and with RR_DEEP=0.
  struct VolSet { float* data = nullptr; uint8_t* cls_all = nullptr; uint32_t* stamp = nullptr; uint32_t* list[2] = {nullptr, nullptr}; uint32_t* counts = nullptr;
                  int parity = 0; bool full = true; uint32_t stampno = 0; } alt;
  hipStream_t integ_stream = nullptr; hipEvent_t integ_done = nullptr, integ_gate = nullptr, draw_done[2] = {nullptr, nullptr};
  bool integ_pending = false, draw_pending[2] = {false, false}, draw_unrecorded = false, deep = true, deep_failed = false;
  int vol_set = 0;
  hipStream_t pre_lane = nullptr; bool pre_on_integ = false;   // the stream the current frame's preparation runs on (pre_stream, or integ_stream: RR_PRE_ON_INTEG)
  bool overlap_fill = true;      // RR_OVERLAP_FILL=0 / tsdf_set_stage_overlap(ctx, 0): everything on the one stream, as in rounds 1 and 2
  // ... and a third lane AHEAD of the context's stream (round 3): what a new frame needs before integrate() can run -- its re-layout and the
  // brick passes (clear / mark / update) -- reads only the new frame and writes state nobody else writes, so it runs on `pre_stream` while
  // the context's stream still works on the previous frame.  Everything the lane writes exists twice and alternates: the frame slots (flip at
  // the first upload of a frame), the brick counters (flip at clearOccupiedBricks), flags + occupied list + count (flip at
  // updateOccupiedBricks).  Two events tie the lanes: pre_done (integrate / draw wait for the lane) and pre_gate (recorded on the context's
  // stream at the lane's first call of a frame, waited for at its first call of the NEXT frame: what the lane overwrites then was last read
  // two frames ago).  (A third copy of everything, letting the lane run two frames ahead, was built and measured: the lanes then crowd each
  // other -- every stage stretches -- and the frame takes 140 instead of 121 us: DESIGN.md section 5.)  Off with stage overlap off, after an explicit frame-slot call (tsdf_select_frame_slot, tsdf_upload_frame_async) and for
  // the pre-processing path.
  hipStream_t pre_stream = nullptr; hipEvent_t pre_done = nullptr, pre_gate = nullptr, pre_gate_b = nullptr, src_ready = nullptr;
  // (round 4) the gate's wait can be DEFERRED inside tsdf_frame_raw_dev: the first two pre-processing passes write nothing the previous draws read, so they run in front of it.
  // Two gate events alternate so that a deferred wait still refers to the record of the previous frame after this frame's record has been made.
  hipEvent_t gate_wait_ev = nullptr; bool gate_wait_pending = false, gate_flip = false;
  bool pre_pending = false, pre_gate_recorded = false, main_since_gate = true, pipeline_blocked = false;
  bool slot_flipped = false, counters_flipped = false, occ_flipped = false;     // once per frame of the lane ...
  bool slot_in_use = false, counters_in_use = false, occ_in_use = false;        // ... and only when a consumer has been queued since the buffer was last written
  int occ_zeroed_word = 0;                                                      // which of the two count words the marking launch cleared
  bool counters_zeroed = false, occ_count_zeroed = false;                       // the re-layout launch / the marking launch has cleared them already (no fill launch of its own)
  uint8_t* d_flags[2]{}; uint32_t* d_occupied[2]{};                             // the two occupancy sets (Bricks::flags / occupied point at the latest update's)
  bool occ_counts_stale = false;                                                // the lane ahead has used the count words: the context's stream zeroes its word itself once
  // native multi-GPU exchange (comm.cpp): one RCCL communicator per context, every collective on the context's stream
  struct Comm {
    void* comm = nullptr;                     // ncclComm_t
    int rank = 0, world = 1;
    bool dedicated = false;                   // rank 0 holds no slab: it only receives, composites and fills holes
    float* d_halo_send = nullptr; float* d_halo_gath = nullptr; size_t halo_floats = 0;   // 2 faces / world x 2 faces
    float* d_hitbuf = nullptr; float* d_hitparts = nullptr; size_t hit_floats = 0;        // 32-byte header + one 32-byte record per view pixel; rank 0: x world
    int32_t* d_counts = nullptr;              // [world][2]: records written, rays hit
    int32_t* h_counts[3] = {nullptr, nullptr, nullptr}; hipEvent_t counts_evt[3] = {nullptr, nullptr, nullptr}; bool counts_rec[3] = {false, false, false};
    uint32_t caps[3] = {0, 0, 0};             // capacity each of the ring's frames was gathered with
    uint64_t frame_no = 0; bool have_last = false; uint64_t last_frame = 0; uint32_t last_cap = 0;
    uint32_t regathers = 0, overflowed_frames = 0, min_capacity = 4096, max_capacity = 0;   // max_capacity: 0 = one record per pixel
    // per-frame verdict of the compact composite (tsdf_comm_frame_status): 1 = the frame was composited from truncated record lists and not repaired
    static constexpr int kVerdicts = 64;
    uint64_t verdict_frame[kVerdicts] = {}; uint8_t verdict[kVerdicts] = {}; bool verdict_set[kVerdicts] = {};
    float* d_frame_stage = nullptr; size_t frame_stage_bytes = 0;                          // tsdf_broadcast_frame: the four arrays as delivered
  } comm;
  bool timers_on = false;
  std::string timer_filter;      // ",name,name," or empty = all
  std::map<std::string, Timer> timers;
};

#define CHECK_CTX(c) do { if (!(c)) return TSDF_ERR_INVALID_ARGUMENT; } while (0)
#define FAIL(c, code, ...) do { char _b[512]; snprintf(_b, sizeof(_b), __VA_ARGS__); (c)->err = _b; return (code); } while (0)
#define HIP_TRY(c, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { FAIL(c, _e == hipErrorOutOfMemory ? TSDF_ERR_OUT_OF_MEMORY : TSDF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); } } while (0)

// helpers defined in abi.cpp
namespace rrhost {
hipStream_t pre_enter(tsdf_ctx* c, bool defer_gate = false);     // (defer_gate: tsdf_frame_raw_dev's first calls; see gate_wait_pending) the lane a frame-preparing call queues its work on (the context's stream when the lane is off); opens the lane's frame
hipError_t pre_leave(tsdf_ctx* c, hipStream_t lane);   // ... and after queuing it
hipError_t join_pre(tsdf_ctx* c);       // GPU side: the context's stream waits for the lane; called by every consumer of a frame's images / brick state
void timer_begin(tsdf_ctx* c, const char* name);
void timer_end(tsdf_ctx* c, const char* name);
hipError_t join_fill(tsdf_ctx* c);      // GPU side: the context's stream waits for the hole filling in flight on the second stream
hipError_t join_integ(tsdf_ctx* c);     // GPU side: the context's stream waits for the integrate in flight on the fourth lane
hipError_t sync_ctx(tsdf_ctx* c);       // host side: both streams
RayTarget ray_target(tsdf_ctx* c);
}  // namespace rrhost
