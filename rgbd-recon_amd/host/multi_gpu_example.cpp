// What the reference's C++ host adds to drive N MI355X with one volume in Z-slabs: one process per GPU, every
// process runs this loop on its own context; `carry` is the host's own channel for the 128-byte communicator id (its ZMQ socket, MPI, a
// shared file -- anything).  Compile check: tests/test_host_adapter.py; INTEGRATION.md section 3 walks through it.
#include <cstdint>
#include <cstdio>

#include "../../include/rgbd_recon_hip.h"

struct Frame { const float* depth_rg; const float* quality; const float* silhouette; const uint8_t* colour_rgb; };

// rank 0 receives the frames (NetKinectArray's reader thread, NetKinectArray.cpp:482-529); the others pass nullptr
int run_rank(tsdf_config cfg, uint32_t rank, uint32_t world, void (*carry)(uint8_t id[TSDF_COMM_ID_BYTES], uint32_t rank), bool (*next_frame)(Frame*),
             const float modelview[16], const float projection[16]) {
  const uint32_t layers = (cfg.res[2] + 7) / 8, lo = rank * layers / world, hi = (rank + 1) * layers / world;
  cfg.device = 0;                                   // one visible device per process (ROCR_VISIBLE_DEVICES)
  cfg.slab_z0 = lo * 8; cfg.slab_z1 = hi * 8 < cfg.res[2] ? hi * 8 : cfg.res[2];
  cfg.slab_recompute_halo = 1;                      // K1's voxels are independent: a rank integrates its own halo layers, no halo exchange
  tsdf_ctx* ctx = nullptr;
  if (tsdf_create(&cfg, &ctx) != TSDF_OK) { fprintf(stderr, "rank %u: %s\n", rank, tsdf_last_error(nullptr)); return 1; }
  // ... tsdf_set_calibration(ctx, i, ...) for every stream, as on one GPU ...
  uint8_t id[TSDF_COMM_ID_BYTES];
  if (rank == 0 && tsdf_comm_unique_id(id) != TSDF_OK) return 1;
  carry(id, rank);                                  // rank 0 sends, the others receive
  if (tsdf_comm_init(ctx, id, rank, world, 0) != TSDF_OK) { fprintf(stderr, "rank %u: %s\n", rank, tsdf_last_error(ctx)); return 1; }
  Frame f{};
  while (rank != 0 || next_frame(&f)) {
    int32_t rc = tsdf_broadcast_frame(ctx, 0, f.depth_rg, f.quality, f.silhouette, f.colour_rgb);      // RCCL broadcast + re-layout
    if (rc == TSDF_OK) rc = tsdf_clear_bricks(ctx);
    if (rc == TSDF_OK) rc = tsdf_mark_bricks(ctx);
    if (rc == TSDF_OK) rc = tsdf_update_occupied(ctx, nullptr);
    if (rc == TSDF_OK) rc = tsdf_integrate(ctx);
    if (rc == TSDF_OK) rc = tsdf_raymarch(ctx, modelview, projection);                                  // this slab's ray segments
    if (rc == TSDF_OK) rc = tsdf_composite_gather(ctx);                                                 // -> rank 0: nearest hit + fillColors()
    if (rc != TSDF_OK) { fprintf(stderr, "rank %u: %s\n", rank, tsdf_last_error(ctx)); break; }
    // rank 0, when it wants the picture: tsdf_composite_finish(ctx, nullptr); tsdf_download_framebuffer(ctx, rgba, depth);
  }
  uint32_t regathered = 0;
  tsdf_composite_finish(ctx, &regathered);
  tsdf_comm_destroy(ctx);
  tsdf_destroy(ctx);
  return 0;
}
