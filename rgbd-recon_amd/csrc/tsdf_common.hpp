// Shared host/device declarations of the HIP TSDF core (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rgbd_recon_hip.h"

namespace rr {

constexpr int TILE = 8;            // storage tile edge: the TSDF lives in HBM as 8x8x8 tiles of 2 KiB
constexpr int TILE_VOX = 512;

// One calibrated stream: its three LUT volumes (CalibVolumes.cpp:64-80,132-144)
struct StreamLut {
  const float4* inv;   // cv_xyz_inv RGBA32F
  const float2* uv;    // cv_uv      RG32F
  const float4* xyz;   // cv_xyz     RGB32F padded to 16 B texels on upload
  int inv_res[3], uv_res[3], xyz_res[3];
};
struct StreamTable {
  StreamLut s[TSDF_MAX_STREAMS];
  int n;
};

// Per-frame images as resident in HBM: one 16-B texel {depth.r, quality, silhouette, 0} per depth pixel
// (the three arrays are always fetched at the same coordinate, tsdf_integration.vs:32-50) and RGBA8 colour.
struct FrameImages {
  const float4* dqs;      // [N][H][W]
  const float* depth;     // [N][H][W] depth.r alone (brick marking reads nothing else)
  const uchar4* color;    // [N][Hc][Wc]
  int w, h, cw, ch;
  // per 8x8-pixel cell {min depth, max depth, min silhouette, max silhouette} of the packed image, [N][rch][rcw] (k_frame_ranges): lets the
  // dense integrate decide for a whole tile and stream that every voxel takes the same branch of the fusion rule.  nullptr: not built.
  const float4* ranges;
  int rcw, rch;
};

// TSDF volume, tile-major: tile (tx,ty,tz) at ((tz - tz0) * nty + ty) * ntx + tx, voxel (x&7,y&7,z&7) inside
// n / d for n < 2^27 by one 64-bit multiply and a shift (m = floor(2^k / d) + 1, k = 27 + ceil(log2 d): exact, see make_fast_div).
// A tile id is workgroup-uniform in the integrate kernels: this form stays on the scalar unit, an integer division does not.
struct FastDiv { uint32_t m, k; };
struct Volume {
  float* data;
  int res[3];          // logical resolution
  int ntx, nty;        // tiles per axis (x, y); z tiles stored: [tz0, tz1)
  FastDiv div_layer, div_row;   // tile id / (ntx * nty), (tile id within a layer) / ntx
  int tz0, tz1;        // stored tile layers (owned slab + halo)
  int own_tz0, own_tz1;  // owned tile layers: the samples of the raymarch this context is responsible for
  int int_tz0, int_tz1;  // tile layers integrate() computes: the owned ones, plus the halo when it is recomputed locally
  int zlo, zhi;        // stored voxel planes [zlo, zhi]: every Z tap is clamped into them (slab contexts)
  // Sparse tile pool (tsdf_config::sparse_pool_tiles): `data` is a pool of pool_tiles tiles and slot[stored tile index] is the
  // tile's position in it, or kNoSlot (every voxel of the tile reads -limit).  The TSDF is rebuilt from scratch every frame,
  // so a tile's slot is simply its position in this frame's compacted active list: no free list, no fragmentation.
  uint32_t* slot;      // nullptr: dense storage
  uint32_t pool_tiles;
  const uint8_t* cls;  // tile class of every STORED tile, index ((tz - tz0) * nty + ty) * ntx + tx; halo layers stay kTileMixed
  int n_stored_tiles;
  float limit;
};

// Per-tile bookkeeping of the integrated tiles (index = tile id, x fastest, int_tz0 first):
//   active  this frame: some voxel of the tile is in the voxel list of an occupied brick
//   cls     what the tile's 2 KiB in HBM hold: kTileMinus = every voxel is -limit (the clear value), kTileMixed = anything else / unknown
// integrate() clears only inactive tiles with cls != kTileMinus and computes only active ones, so a frame's volume traffic
// follows the occupied bricks instead of the whole volume (the reference clears everything, :249-250).
// (An empty-space pyramid over these classes for the raymarch was built and measured: it loses)
constexpr uint8_t kTileMinus = 0, kTileMixed = 2;
constexpr uint32_t kNoSlot = 0xffffffffu;
struct TileState {
  uint32_t* stamp;     // per integrated tile: the frame number that last put it on the active list (dedupes the scatter of k_classify_lists)
  uint8_t* cls;        // owned tiles only (points into Volume::cls at the first owned layer)
  uint32_t* list;      // compacted active tile ids of THIS frame (unordered)
  uint32_t* count;     // device scalar
  const uint32_t* prev_list;   // the active list of the previous integrate(): the only tiles that can hold anything but the clear value
  const uint32_t* prev_count;
  uint32_t* next_count;        // = prev_count's word: re-armed (zeroed) by the integrate kernel for the next frame
  int n;               // owned tiles
  int uniform;         // every tile lies inside exactly one brick's voxel list: tile active => all its voxels drawn
};

// Occupancy bricks (inc_bricks.glsl:10-20) plus the voxel -> brick tables that restate
// VolumeSampler::containedVoxels (volume_sampler.cpp:50-62): along axis a voxel v lies in bricks
// [first[a][v], first[a][v] + count[a][v]).
struct Bricks {
  uint32_t* counters;       // per brick
  uint8_t* flags;           // counter >= min_voxels (recon_integration.cpp:436)
  uint32_t* num_occupied;   // device scalar: the count of the latest updateOccupiedBricks() (one of two words used alternately, so that
                            // clearOccupiedBricks() leaves the list of the last update intact, as the reference's host vector is)
  uint32_t* occupied;       // compacted ids of the occupied bricks (Occupied SSBO, inc_bricks.glsl:18-20), unordered
  const uint16_t* vox_first[3];
  const uint8_t* vox_count[3];
  const uint16_t* tile_b0[3];   // per storage tile index along an axis: first / last brick whose voxel list reaches into it
  const uint16_t* tile_b1[3];   // (b0 > b1: none)
  const uint8_t* tile_full[3];  // per storage tile index along an axis: 1 = every voxel coordinate of the tile is in at least one brick's list
  const uint16_t* brick_t0[3];  // the inverse: per brick index along an axis, first / last storage tile its voxel list reaches into
  const uint16_t* brick_t1[3];  // (t0 > t1: the brick holds no voxel)
  int res[3];               // brick grid
  int n;
  float size[3];            // world brick size
  float bbox_min[3];
};

// Projection cache (round 3).  texture(cv_xyz_inv[i], voxel centre).xyz -- the (u, v, z) a voxel projects to in stream i,
// tsdf_integration.vs:31 -- depends on the calibration volume and the voxel grid only, never on the frame.  The LDS form of the
// integrate kernel recomputes it every frame (texel box -> LDS -> X pass -> Y pass -> z lerp: ~8 dependent round trips and ~17
// workgroup barriers per tile); with 288 GB of HBM it is cheaper to keep most of it.  What is kept per (tile, stream) is the state
// after the Y pass: the tile's x- and y-filtered LUT planes, dz_max planes of 8 x 8 float3 (the z filter of a voxel then is ONE lerp
// of two of them -- the same operands, the same operation as the LDS form's last step, so the same bits).  dz planes instead of 8
// voxel planes: 2.2 - 3.7 KiB instead of 6 KiB per tile and stream at the BASELINE sizes.  Slots are dealt on first use
// (slot[stored tile]) and filled by the LDS kernel itself while it integrates the tile (write-through); from then on the tile is the
// cached kernel's: per wave one z-plane of 64 voxels -- coalesced 768-byte plane reads, image gathers, fusion rule, 256-byte store.
// No LUT texel, no LDS, no barrier.
constexpr uint32_t kItemFresh = 0x80000000u;   // items[w] >= kItemFresh: not (yet) cached -- the LDS kernel's work; kItemFresh | slot: ... and it fills `slot`
constexpr uint32_t kItemNone = 0xffffffffu;    // no slot (pool exhausted)
struct ProjCache {
  float* data;             // pool: slot s at s * slot_floats; inside: [stream][plane < dz_max][8 x 8 voxels (x fastest)][3]
  uint32_t* slot;          // per STORED tile: its slot, or kNoSlot
  uint32_t* alloc;         // device scalar: slots handed out so far
  uint32_t cap;            // pool capacity in slots
  uint32_t slot_floats;    // N * dz_max * 192
  uint32_t dz_max;         // the most LUT z-planes any tile of the volume touches in any stream (fit_lut_to_volume, abi.cpp)
  uint32_t* items;         // per work item of this frame's launch: slot (cached), kItemFresh | slot (to be filled now), kItemNone
  uint32_t* n_slow;        // device scalar: work items of this frame the LDS kernel must take (0: it returns at once)
  uint32_t* n_slow_next;   // the other of the two alternating words, zeroed by this frame's pair-mask pass for the next frame
  int inv_rz[TSDF_MAX_STREAMS];   // z resolution of each stream's inverse LUT (the cached kernel's only use of the calibration)
};

struct Mat4 { float m[16]; };   // column-major
struct ViewParams {
  Mat4 mv, proj, mv_inv, v2w_inv, img_to_eye, normal, mv_v2w, glnormal_inv;
  float cam_vol[3], cam_world[3];
  int w, h;
  int shade_mode;
  int skip;
  // side-by-side stereo (kinect_client.cpp:637-664): the GL viewport's origin (gl_FragCoord = origin + pixel + 0.5) and the shader's
  // viewport_offset uniform (tsdf_raymarch.fs:70,388-389; setViewportOffset, recon_integration.cpp:527)
  int vp_org[2];
  float vp_off[2];
};

// ViewLod atlas (view_lod.cpp:24-61)
struct Atlas {
  float4* color;
  float* depth;
  int aw, h;          // atlas size (1.5 w, h)
  int num_lods;
  int off[TSDF_MAX_LODS][2];
  int res[TSDF_MAX_LODS][2];
};

// image pre-processing (k_preprocess.hip)
struct PreParams {
  int W, H, N;
  float cv_min[TSDF_MAX_STREAMS], cv_max[TSDF_MAX_STREAMS];   // CalibVolumes::getDepthLimits
  float cam[TSDF_MAX_STREAMS][3];                              // CalibVolumes::getCameraPositions
  float bbox_min[3], bbox_max[3];
  int filter_textures, refine;
  // 8-bit wire depth: compress = isCompressedDepth(); uniforms near / scale / scaled_near (NetKinectArray.cpp:343-349)
  int compress[TSDF_MAX_STREAMS];
  float dc_near[TSDF_MAX_STREAMS], dc_scale[TSDF_MAX_STREAMS], dc_scaled_near[TSDF_MAX_STREAMS];
};
struct PreBuffers {
  const float* raw;       // [N][H][W] metres
  float* depth2;          // morph output
  const float* fdepth;    // what the filter pass samples: depth2 or raw (use_processed_depth)
  float2* depth_rg;       // filter output (normalised depth, range quality)
  float4* lab;            // filter output, Lab colour
  float2* depth_b;        // boundary output
  float4* normal;         // normal output
  float4* dqs;            // packed {depth.r, quality, silhouette, 0}
  float* depth_plane;
  // (round 4) the 16 x 16 blocks that hold a boundary candidate: the filter pass numbers them (blk_flag = 1 + position in cand_list, 0 = none; cand_count zeroed by the
  // morph launch), the boundary pass runs them FIRST (their chain -- Lab tile, then the colour comparisons -- is what its launch waits for).  Null: no list.
  uint32_t* blk_flag; uint32_t* cand_list; uint32_t* cand_count; uint32_t cand_cap;
};
// point back-end (k_points.hip)
struct PointParams {
  Mat4 pmv;                     // P * MV, formed in double, rounded once
  float bbox_min[3], bbox_max[3];
  const float4* normals;        // [N][H][W] kinect_normals (xyz), may be null
};
void launch_draw_points(hipStream_t st, const ViewParams& P, const PointParams& Q, const StreamTable& T, const FrameImages& F, unsigned long long* key,
                        float4* fb_c, float* fb_d);

void launch_draw_trigrid(hipStream_t st, const ViewParams& P, const PointParams& Q, const StreamTable& T, const FrameImages& F, float min_length, uint32_t* zbuf,
                         float4* acc, float4* fb_c, float* fb_d);

// inverse calibration volume builder (k_inverter.hip)
struct InverterGrid {
  uint32_t rx, ry, rz, n;     // forward volume resolution, sample count
  int g[3];                   // uniform grid over the samples
  double gmin[3], cell[3], inv[3];
};
struct InverterQuery {
  uint32_t res[3];            // output resolution
  float start[3], step[3];    // sample_start, sample_step (calibration_inverter.cpp:74-78)
  float plane[6][4];          // kinect::Frustum planes
};
void launch_inverter_build(hipStream_t st, const InverterGrid& G, const float* xyz, uint32_t* count, uint32_t* start, uint32_t* sums, float4* sorted);
void launch_inverter_query(hipStream_t st, const InverterGrid& G, const InverterQuery& Q, const float* xyz, const uint32_t* start, const float4* sorted, float4* out);

// one wire message in HBM: per sensor [colour: cs bytes][depth], rec bytes per sensor (k_ingest.hip)
struct WireLayout {
  const uint8_t* msg;
  uint32_t rec, cs;
  int n, cw, ch, w, h;
  int cfmt, dfmt;         // TSDF_COLOR_* / TSDF_DEPTH_*
};
void launch_wire_unpack(hipStream_t st, const WireLayout& L, uchar4* rgba, float* raw);
// ranges: the frame slot's 8x8-pixel range cells (written by the last pass), or null; rgb -> rgba (n_color_px pixels) and zero (zero_words words, a multiple of
// 4): the frame's colour re-layout and the brick counters' clear riding along in the first launch, or null
void launch_preprocess(hipStream_t st, const PreParams& P, const PreBuffers& B, const StreamTable& T, const FrameImages& F, const Bricks& BR, float4* ranges,
                       const uint8_t* rgb = nullptr, uchar4* rgba = nullptr, size_t n_color_px = 0, uint32_t* zero = nullptr, uint32_t zero_words = 0,
                       int only = 0);   // only: 0 = the five passes, 1 .. 5 = that pass alone (morph, filter, boundary, normal, quality: per-kernel timers)
void launch_pre_lab(hipStream_t st, const PreParams& P, const PreBuffers& B, const StreamTable& T, const FrameImages& F);   // PreBuffers::lab, on request

// launchers (one per kernel family, defined in the .hip files)
void launch_pack_color(hipStream_t st, const uint8_t* rgb, uchar4* rgba, size_t n);
void launch_frame_ranges(hipStream_t st, const float4* dqs, int n_streams, int w, int h, float4* ranges);
// pack_frame + frame_ranges (+ pack_color when rgb != nullptr) as one launch; rgb / rgba must be readable / writable up to a multiple of 4 pixels
void launch_pack_frame_fused(hipStream_t st, const float* depth_rg, const float* quality, const float* silhouette, float4* dqs, float* depth, float4* ranges,
                             int n_streams, int w, int h, const uint8_t* rgb, uchar4* rgba, size_t n_color_px,
                             uint32_t* zero = nullptr, uint32_t zero_words = 0);   // zero: a word buffer (multiple of 4 words) the launch clears as well (the coming frame's brick counters)
void launch_fill_u32(hipStream_t st, uint32_t* p, uint32_t v, size_t n);
struct PeelClear;
void launch_mark_bricks(hipStream_t st, const StreamTable& T, const FrameImages& F, const Bricks& B, uint32_t* zero_word = nullptr, const PeelClear* pc = nullptr);   // zero_word: a device word the launch clears as well; pc: peel tiles to reset (one more layer of blocks)
void launch_update_occupied(hipStream_t st, const Bricks& B, uint32_t min_voxels, uint32_t* next_count);
// full_classify: 1 = walk every tile (first frame, after anything that may have left non-clear data outside the previous active
// list); 0 = scatter from the occupied bricks + check the previous list only (work follows the scene, not the volume)
// PeelClear: the peel-tile reset of the coming draw (k_raymarch.hip's k_clear_peel_tiles) rides along in the k_classify_lists launch
// ... and so does the zeroing of the spare brick-counter buffer (`zero`, in 16-byte units of zero_words / 4)
struct PeelClear { uint4* peels; const uint8_t* touched_prev; int w, h, ntx, n_tiles; uint32_t* zero; uint32_t zero_words; };   // null pointers: nothing to do
void launch_integrate(hipStream_t st, const StreamTable& T, const FrameImages& F, const Volume& V, const Bricks& B, const TileState& S, int use_bricks, int lds_ok,
                      int full_classify, uint32_t frame_stamp, int phase = 0, const PeelClear* pc = nullptr,
                      const float4* tile_bounds = nullptr,   // per (stored tile, stream) 2 x float4 LUT-box bounds (launch_tile_bounds), or null
                      uint32_t* pair_masks = nullptr,         // per work item: the frame's pair classes (written by the launch itself), or null
                      const ProjCache* proj = nullptr,        // projection cache (needs tile_bounds / pair_masks: its work items are classified by the pair-mask pass), or null
                      uint4* work_recs = nullptr);            // per work item: the 16-byte record of k_integrate_tiles_rec (written by the pair-mask pass), or null
// [work items, cached items, (tile, stream) pairs of cached items evaluated per voxel, items taken by the LDS kernel] of the last launch -> out[4] (device)
void launch_item_stats(hipStream_t st, const StreamTable& T, const TileState& S, int use_bricks, const uint32_t* pair_masks, const ProjCache& PC, uint32_t* out);
void launch_tile_bounds(hipStream_t st, const StreamTable& T, const Volume& V, float4* bounds);
int integrate_box_cap();
int integrate_row_cap();
void launch_mark_all_mixed(hipStream_t st, const TileState& S);
void launch_volume_to_linear(hipStream_t st, const Volume& V, float* linear);
void launch_volume_from_linear(hipStream_t st, const Volume& V, const float* linear);
void launch_clear_peels(hipStream_t st, float4* peels, int n);   // every peel to the clear value (1,0,1,0)
void launch_depth_limits(hipStream_t st, const ViewParams& P, const Bricks& B, float4* peels, uint8_t* touched_cur = nullptr, const uint8_t* touched_prev = nullptr,
                         int already_cleared = 0);
struct LongRay { uint32_t pix, n, max_n; float prev; float x, y, z, pad; };   // state of a ray handed to k_march_long
struct RayTarget {
  float4* color; float* depth; int stride; float* nsamples; const float4* peels; float clear[4];
  // 8x8-pixel tile history (k_raymarch.hip); null = none.  touched_prev: the previous draw's tiles (its sample counts), touched_prev_target:
  // the tiles of the draw that last wrote THIS target (the previous one, or the one before when two pyramids alternate), touched_recycle:
  // the oldest mask, zeroed for the next draw; rewrite_target / rewrite_all: no valid history for the target / for the sample counts
  const uint8_t* touched_cur; const uint8_t* touched_prev; const uint8_t* touched_prev_target; uint8_t* touched_recycle; int rewrite_all, rewrite_target;
  uint8_t* fill_mask;   // (nullable) per tile: touched by this draw or one of the two before -- what the hole filling of this draw has to look at
};
void launch_raymarch(hipStream_t st, const ViewParams& P, const StreamTable& T, const FrameImages& F, const Volume& V, const RayTarget& R, int partial,
                     void* hit_list, uint32_t* hit_counters, int parity, int phase = 0, void* long_list = nullptr, uint32_t cap = 0xffffffffu,
                     int box_march = 1);   // dense whole-volume march without depth limits: 1 = through LDS voxel boxes with tile-class leaps (k_march_box), 0 = gather march
// hit_list: 16 B per view pixel; long_list: 32 B per view pixel (rays handed to the wave-per-ray pass after `cap` samples);
// hit_counters: 4 words [hit, hit', long, long'], the primed ones re-armed for the next frame by k_shade
void launch_inpaint_level(hipStream_t st, const Atlas& A, int lod);
// tile_mask (nullable): one byte per 8x8-pixel level-0 tile -- only tiles it flags (and what depends on them) are computed; lvl_mask: scratch
void launch_inpaint_pyramid(hipStream_t st, const Atlas& A, const uint8_t* tile_mask = nullptr, uint8_t* const lvl_mask[2] = nullptr);
// mask: Reconstruction::m_color_mask_mode (0 all channels, 1 red only, 2 green + blue only: glColorMask, recon_integration.cpp:321-333);
// keep_color: the colour buffer was NOT cleared before this draw (the anaglyph's second eye, kinect_client.cpp:627)
void launch_colorfill(hipStream_t st, const Atlas& A, int w, int h, float4* fb_color, float* fb_depth, int mask = 0, int keep_color = 0, const uint8_t* tile_mask = nullptr);
// fill_holes off with a colour mask / an uncleared colour buffer: the march renders into the atlas' level-0 region and this merges it
void launch_resolve_masked(hipStream_t st, const Atlas& A, int w, int h, float4* fb_color, float* fb_depth, int mask, int keep_color);
void launch_clear_image(hipStream_t st, float4* color, float* depth, size_t n, float4 c, float d);
void launch_export_partial(hipStream_t st, const RayTarget& R, int w, int h, void* dst);
void launch_composite(hipStream_t st, const void* gathered, int n, const RayTarget& R, int w, int h);
void launch_export_hits(hipStream_t st, const RayTarget& R, int w, const void* hit_list, const uint32_t* hit_count, const void* long_list, const uint32_t* long_count,
                        void* dst, uint32_t capacity);   // long_list: the second-pass rays of a two-pass march (null: none)
void launch_composite_hits(hipStream_t st, const void* gathered, size_t stride_bytes, int n, const RayTarget& R, int w, int h, unsigned long long* key, int own_counts);

}  // namespace rr
