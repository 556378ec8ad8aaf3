#!/usr/bin/env python3
"""bench.py -- frames/s of integrate() + drawF() on the synthetic scene of SURVEY.md §8d.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c1|c3|c4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one frame of the reference's per-frame call order (source/kinect_client.cpp:569-599,614):
clearOccupiedBricks -> mark_brick (K6) -> updateOccupiedBricks -> integrate (K0+K1) -> drawF
(K5 depth limits, K2 raymarch, K3/K4 hole filling), with the frame images already resident in HBM.

Scene.  Two frames of the same rig -- A = the scene of SURVEY.md §8d, B = the same objects moved -- lie in HBM as the arrays the
producer delivers (depth RG32F, quality, silhouette, colour RGB8).  Every timed step takes the OTHER one as a NEW frame, as the
reference integrates only new frames (kinect_client.cpp:586-599): tsdf_upload_frame_dev re-lays it out for the kernels (one launch)
and the frame is computed; the scene moves, so the incremental bookkeeping pays its worst case (every active tile of the previous
frame goes stale and is reset, the image-space dirty tiles change).  `value` is that rate.  Beside it: `static` (frame A arrives
every step: nothing moves) and `resident_frames` (round 1 / 2's definition: two already re-laid-out frames alternate, no
per-frame re-layout).

N > 1: the north-star partition -- ONE volume split into Z-slabs over the ranks (strong scaling: `value` = frames of the one
volume per second), halo tile layers recomputed locally (default) or RCCL all-gathered, nearest-hit gather of one 32-byte record
per hit ray to rank 0 (rgbd-recon_amd/multigpu.py).  Before anything is timed rank 0 checks the composite of both frames
against an unpartitioned context, bit for bit, and the run fails if they differ.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[2]: the configuration the metric is quoted on (512^3 x 4 streams, cull + inpaint)
    "c2": dict(res=(512, 512, 512), streams=4, use_bricks=True, skip_space=True, fill_holes=True,
               name="512^3 TSDF x 4 streams 640x480, 1280x720 view, brick cull (8^3-voxel bricks) + inpaint"),
    # BASELINE.json configs[3] / configs[4] (quoted on 8 GPUs; they also fit one MI355X: 1024^3 is a 4 GiB volume)
    "c3": dict(res=(512, 512, 512), streams=8, use_bricks=True, skip_space=True, fill_holes=True,
               name="512^3 TSDF x 8 streams 640x480, 1280x720 view, brick cull (8^3-voxel bricks) + inpaint"),
    "c4": dict(res=(1024, 1024, 1024), streams=8, use_bricks=True, skip_space=True, fill_holes=True,
               name="1024^3 TSDF x 8 streams 640x480, 1280x720 view, brick cull (8^3-voxel bricks) + inpaint"),
    # BASELINE.json configs[1]
    "c1": dict(res=(256, 256, 256), streams=4, use_bricks=False, skip_space=False, fill_holes=False,
               name="256^3 TSDF x 4 streams 640x480, 1280x720 view, dense integrate + raymarch"),
}
VIEW = (1280, 720)
LUT = 128
LIMIT = 0.01
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "traffic.json")   # per-kernel HBM bytes from separate rocprofv3 --pmc passes
MOVED = dict(sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2))  # frame B: both objects somewhere else (every tile churns)


def measured_traffic(config, kernel, key="hbm_bytes"):
    """HBM bytes (or, key="valu_insts", VALU wave-instructions) per launch of `kernel` from the committed PMC summary
    (profiles/traffic.json, made by profiles/summarize_pmc.py from separate FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU passes with the
    gfx950 corrections); None if absent."""
    try:
        with open(TRAFFIC_JSON) as f:
            return json.load(f).get(config, {}).get(kernel, {}).get(key)
    except (OSError, ValueError):
        return None


def dense_bytes(res, n_streams):
    """BASELINE.md §3 / SURVEY.md §8d dense byte counts per launch: every voxel, every LUT texel, every image pixel once."""
    V = res[0] * res[1] * res[2]
    L, P, R = LUT ** 3, 640 * 480, VIEW[0] * VIEW[1]
    return dict(integrate=4 * V + n_streams * 16 * L + n_streams * 16 * P, march=4 * V + 24 * R, inpaint=67 * R)


def culled_integrate_bytes(np, hip, scenes, res, n_streams):
    """Algorithmic bytes of ONE culled integrate launch = what its work units -- the active 8^3-voxel tiles -- must move at least:
         2 KiB stored per active tile
       + 16 B x N x (distinct inverse-LUT texels inside the tiles' texel boxes; all streams share the LUT grid)
       + 16 B x (valid depth pixels of the frame's N images: the packed {depth, quality, silhouette} texels under the surface)
    averaged over the frames of the timed region.  (the dense formula only applies with use_bricks off.)"""
    tot, tiles_n = 0.0, []
    for k, sc in enumerate(scenes):
        hip.upload_frame(sc)
        hip.clearOccupiedBricks(); hip.markBricks(); hip.updateOccupiedBricks(False); hip.integrate()
        tiles, _ = hip.active_tiles()
        touched = np.zeros((LUT, LUT, LUT), bool)
        rng = []
        for a in range(3):                               # the kernel's own fp32 index arithmetic (axis_linear, sampling.hpp)
            n = np.float32(LUT)
            step = np.float32(1.0) / np.float32(res[a])
            v0 = np.minimum(tiles[:, a] * 8, res[a] - 1).astype(np.float32)
            v1 = np.minimum(tiles[:, a] * 8 + 7, res[a] - 1).astype(np.float32)
            lo = np.clip(np.floor((v0 + np.float32(0.5)) * step * n - np.float32(0.5)), 0, LUT - 1).astype(np.int64)
            hi = np.clip(np.floor((v1 + np.float32(0.5)) * step * n - np.float32(0.5)) + 1, 0, LUT - 1).astype(np.int64)
            rng.append((lo, hi))
        for i in range(tiles.shape[0]):
            touched[rng[2][0][i]:rng[2][1][i] + 1, rng[1][0][i]:rng[1][1][i] + 1, rng[0][0][i]:rng[0][1][i] + 1] = True
        valid_px = int((sc["depth"][..., 0] > 0).sum())
        tot += 2048.0 * tiles.shape[0] + 16.0 * n_streams * int(touched.sum()) + 16.0 * valid_px
        tiles_n.append(int(tiles.shape[0]))
    return tot / len(scenes), tiles_n


def cpu_baseline(scene, cfg, limit, brick):
    """The oracle (kind "port": the reference has no CPU path) on the same workload at full size, with all host cores
    and with ONE thread: clear/mark/update bricks + integrate() + drawF() at the bench view; all cores: the fastest of
    three frames (the first pays the first touch of the volume), one thread: one frame after that."""
    from oracle.oracle import OracleRecon, set_threads
    import rgbd_recon_amd as rr
    cores = os.cpu_count() or 1
    o = OracleRecon(scene, res=cfg["res"], brick_size=brick, limit=limit, view=VIEW)
    o.setUseBricks(cfg["use_bricks"]); o.setSpaceSkip(cfg["skip_space"]); o.setColorFilling(cfg["fill_holes"])
    mv, pr = rr.scene.default_view(*VIEW)

    def one():
        t0 = time.perf_counter()
        o.clearOccupiedBricks(); o.markBricks(); o.updateOccupiedBricks()
        o.integrate()
        t1 = time.perf_counter()
        o.drawF(mv, pr)
        t2 = time.perf_counter()
        return (t2 - t0, t1 - t0, t2 - t1)

    threads = set_threads(0)                             # what OpenMP uses by default on this box
    best = min(one() for _ in range(3))
    set_threads(1)
    single = one()
    set_threads(threads)
    return {"value": 1.0 / best[0], "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"oracle/libtsdf_oracle.so, OpenMP {threads} threads ({cores} logical CPUs), fastest of 3 full-size frames of scene A: bricks+integrate {best[1]:.2f} s, "
                      f"drawF at {VIEW[0]}x{VIEW[1]} {best[2]:.2f} s",
            "one_thread": {"value": 1.0 / single[0], "unit": "frames/s", "cores": 1,
                           "sample": f"the same frame with one thread: bricks+integrate {single[1]:.2f} s, drawF {single[2]:.2f} s"}}


def same(np, a, b):
    return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--halo", default="recompute", choices=["recompute", "exchange"], help="N > 1: integrate the halo layers locally, or RCCL all-gather them")
    ap.add_argument("--composite", default="compact", choices=["compact", "dense"], help="N > 1: gather hit records, or whole partial images")
    ap.add_argument("--compositor", default="auto", choices=["auto", "shared", "dedicated"],
                    help="N > 1: 'dedicated' = rank 0 only receives, composites and fills holes and the volume is split over ranks 1..N-1 "
                         "(the gathering rank's image-space tail is longer than a slab's share of the volume work, so an even split makes "
                         "rank 0 every frame's critical path); 'shared' = N slabs, rank 0 composites as well.  auto = dedicated")
    ap.add_argument("--exchange", default="torch", choices=["torch", "native"],
                    help="N > 1: the collectives through torch.distributed (RCCL), or from inside the library (tsdf_halo_exchange / tsdf_composite_gather / "
                         "tsdf_broadcast_frame: RCCL called from C++ on the context's stream, what a C++ host drives; compact composite only)")
    ap.add_argument("--frames", default="broadcast", choices=["broadcast", "resident"],
                    help="N > 1: every new frame arrives on rank 0 and is broadcast to the slab ranks inside the step (default: the frame reaches ONE process "
                         "in the reference, NetKinectArray.cpp:482-529), or both bench frames lie in every rank's HBM")
    ap.add_argument("--partition", default="even", choices=["even", "balanced"],
                    help="N > 1: slabs of equal thickness, or boundaries by the occupied bricks per tile layer of the bench frames (measured no better: "
                         "a slab's march cost follows its position along the view, not its bricks -- DESIGN.md section 6)")
    ap.add_argument("--scene", default="moving", choices=["moving", "static"], help="alternate two resident frames in the timed region (default), or repeat one")
    ap.add_argument("--frames-in-flight", type=int, default=1, choices=[1, 2, 3, 4],
                    help="single GPU: process this many frames concurrently, one context + HIP stream per slot (every frame rebuilds the "
                         "volume from scratch, so frames are independent and the frame's small latency-bound kernels overlap well). "
                         "A throughput mode: the latency of a frame does not improve, and per-kernel times (roofline) are measured under "
                         "contention.  Default 1")
    ap.add_argument("--sparse-pool", type=int, default=0, metavar="TILES",
                    help="store the TSDF in a sparse pool of this many 8^3-voxel tiles (2 KiB each) instead of a dense array "
                         "(BASELINE.json configs[4] 'sparse-brick allocation'); needs a culled configuration")
    ap.add_argument("--preprocess", action="store_true", help="also run the image pre-processing passes (f1) every frame, from the raw depth/colour (static scene)")
    ap.add_argument("--ingest", default=None, choices=["f32-rgb8", "f32-dxt1", "u8-rgb8", "u8-dxt1", "u8-dxt5"],
                    help="also measure the wire path (f2): every frame arrives as one host message, is copied through the pinned double "
                         "buffer, unpacked/decoded on the GPU and pre-processed (implies --preprocess); reported beside `value`, never as it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-timers", action="store_true", help="leave the per-kernel HIP event timers off")
    ap.add_argument("--no-c1", action="store_true", help="skip the dense configs[1] kernel timings (`roofline_c1`)")
    ap.add_argument("--long-steps", type=int, default=1000, help="steps of the extra long pass reported as `long_run` (0 = skip)")
    args = ap.parse_args()
    if args.ingest:
        args.preprocess = True
    if args.ingest or args.frames_in_flight > 1:
        args.scene = "static"

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner from inside
    # communicator creation on rank 0), so file descriptor 1 is pointed at stderr for the whole run and the JSON line goes to
    # the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch                      # first: the HIP runtime torch bundles is the one the library binds to
    import torch.distributed as dist
    import rgbd_recon_amd as rr
    from importlib import import_module
    mg = import_module("rgbd-recon_amd.multigpu")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if "RR_BENCH_DEVICE" not in os.environ and world > torch.cuda.device_count():
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} device(s) visible: one rank per GPU (RCCL refuses two ranks on one device)")
    if "RR_BENCH_DEVICE" not in os.environ and local >= torch.cuda.device_count():
        raise SystemExit(f"LOCAL_RANK {local} has no device (visible: {torch.cuda.device_count()})")
    # rehearsal hooks for a box with ONE GPU (never set by the driver): all ranks on device RR_BENCH_DEVICE, gloo instead of RCCL
    backend = os.environ.get("RR_BENCH_BACKEND", "nccl")
    if "RR_BENCH_DEVICE" in os.environ:
        local = int(os.environ["RR_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    # RR_BENCH_EXCHANGE_ALONE=1 (rehearsal, never set by the driver): one rank runs the whole slab exchange -- pack, RCCL
    # collectives of a world of one, composite -- so that the cost of the collective path itself can be read on a 1-GPU box
    alone = world == 1 and os.environ.get("RR_BENCH_EXCHANGE_ALONE") == "1"
    if alone:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    slabs_mode = world > 1 or alone
    if slabs_mode and (args.frames_in_flight > 1 or args.ingest):
        raise SystemExit("--frames-in-flight / --ingest are single-GPU options")

    cfg = CONFIGS[args.config]
    n_streams = cfg["streams"]
    mk = dict(n_streams=n_streams, width=640, height=480, lut_res=LUT, inv_res=LUT)
    scene = rr.scene.make_scene(**mk)
    scenes = [scene] + ([rr.scene.make_scene(**mk, **MOVED)] if args.scene == "moving" else [])
    ext = scene["bbox_max"] - scene["bbox_min"]
    brick = [float(ext[a]) / cfg["res"][a] * 8 for a in range(3)]          # 8^3 voxels per brick
    dedicated = world > 1 and args.compositor != "shared"
    slab = (mg.worker_slab_range if dedicated else mg.slab_range)(cfg["res"][2], rank, world) if slabs_mode else (0, 0)
    partition_note = "equal thickness"

    def balanced_ranges(n_slabs):
        """slab boundaries from the occupied bricks per tile layer of the bench frames (every rank computes the same ones: the brick tables
        depend on the frames alone) -- a layer's weight is its occupied bricks plus a constant for what a layer costs however empty it is"""
        t = rr.ReconIntegrationHip(scene, res=cfg["res"], brick_size=brick, limit=LIMIT, view=(64, 36), device=local, slab=mg.slab_range(cfg["res"][2], 0, 4), recompute_halo=True)
        layers = (cfg["res"][2] + 7) // 8
        w = np.zeros(layers)
        rb = t.res_bricks
        for sc in scenes:
            t.upload_frame(sc)
            t.clearOccupiedBricks(); t.markBricks(); t.updateOccupiedBricks(False)
            fl = t.bricks()[1].reshape(rb[2], rb[1], rb[0])
            per_z = fl.sum((1, 2)).astype(float)
            if rb[2] != layers:                           # bricks are not 8 voxels thick: spread a brick row over the layers it covers
                per_z = np.interp((np.arange(layers) + 0.5) / layers, (np.arange(rb[2]) + 0.5) / rb[2], per_z) * rb[2] / layers
            w += per_z
        t.close()
        w += 0.5 * w.sum() / layers
        return mg.balanced_slab_ranges(w, n_slabs, cfg["res"][2])

    if world > 1 and args.partition == "balanced" and cfg["use_bricks"]:
        if dedicated:
            slab = balanced_ranges(world - 1)[max(rank - 1, 0)]
        else:
            slab = balanced_ranges(world)[rank]
        partition_note = "balanced by occupied bricks per tile layer"
    # RR_BENCH_ALONE_SLAB="k/n" (with RR_BENCH_EXCHANGE_ALONE=1; rehearsal, never set by the driver): the one rank holds slab k of n --
    # what ONE worker rank of an n-slab partition does per frame, measured on the one GPU (its composite is of course not the frame)
    alone_slab = os.environ.get("RR_BENCH_ALONE_SLAB") if alone else None
    if alone_slab:
        parts = alone_slab.split("/")                    # "k/n" (equal thickness) or "k/n/balanced"
        k_, n_ = int(parts[0]), int(parts[1])
        slab = balanced_ranges(n_)[k_] if len(parts) > 2 else mg.slab_range(cfg["res"][2], k_, n_)

    # the frames as they arrive lie in HBM before anything is timed and every step re-lays one of them out (tsdf_upload_frame_dev) --
    # except in the pre-processing / ingest modes (they produce the images themselves) and the frames-in-flight throughput mode
    repack = not (args.preprocess or args.frames_in_flight > 1)
    # --preprocess (round 4): the RAW frames (depth in metres, RGB8) lie in HBM and every step hands the other one to tsdf_frame_raw_dev -- the moving
    # scene through processTextures() on the lane ahead, as `value`'s loop does with pre-processed frames
    raw_repack = args.preprocess and not args.ingest and args.frames_in_flight == 1 and not (world > 1 or alone)

    def make_ctx(slab=(0, 0), recompute=False, sparse=0, lane_flags=0):
        h = rr.ReconIntegrationHip(scene, res=cfg["res"], brick_size=brick, limit=LIMIT, view=VIEW, device=local, slab=slab,
                                   recompute_halo=recompute, sparse_pool_tiles=sparse, lane_flags=lane_flags)
        h.setUseBricks(cfg["use_bricks"]); h.setSpaceSkip(cfg["skip_space"]); h.setColorFilling(cfg["fill_holes"])
        if args.frames_in_flight > 1:
            h.set_stage_overlap(False)                    # several contexts already overlap whole frames: one stream each (three lanes each would fight over the hardware queues)
        if not repack and not raw_repack:                 # the second resident frame (modes without a per-frame re-layout alternate the two frame slots;
            for k, sc in enumerate(scenes[1:], 1):        #  an explicit frame-slot call switches the context's lane ahead off for good)
                h.select_frame_slot(k); h.upload_frame(sc)
            h.select_frame_slot(0)
        return h

    mem_free_before = torch.cuda.mem_get_info(local)[0]
    hip = make_ctx(slab, args.halo == "recompute" and slabs_mode, args.sparse_pool)
    # ONE explicit torch stream carries the context's kernels, the HIP event timers and the collectives (multigpu.py: the handle
    # of torch's default stream is 0 and cannot be handed over)
    stream = torch.cuda.Stream()
    if args.preprocess:
        hip.upload_raw_frame(scene)
        hip.sync()
    drv = mg.SlabDriver(hip, rank, world, f"cuda:{local}", view=VIEW, halo=args.halo, composite=args.composite,
                        preprocess=args.preprocess, exchange_when_alone=alone, stream=stream, compositor="dedicated" if dedicated else "shared",
                        native=slabs_mode and args.exchange == "native")
    mv, pr = rr.scene.default_view(*VIEW)
    nsc = len(scenes)
    raw = []
    if repack:
        for sc in scenes:
            ts = [torch.from_numpy(np.ascontiguousarray(sc[k])).to(f"cuda:{local}") for k in ("depth", "quality", "silhouette", "color")]
            raw.append((ts, tuple(t.data_ptr() for t in ts)))
        torch.cuda.synchronize()

    raw_dev = []
    if raw_repack:
        for sc in scenes:
            ts = [torch.from_numpy(np.ascontiguousarray(sc["depth_raw"], np.float32)).to(f"cuda:{local}"), torch.from_numpy(np.ascontiguousarray(sc["color"], np.uint8)).to(f"cuda:{local}")]
            raw_dev.append((ts, tuple(t.data_ptr() for t in ts)))
        torch.cuda.synchronize()

    bcast = slabs_mode and repack and args.frames == "broadcast"

    def step(d, i):
        if bcast and d is drv:                             # the frame arrived on rank 0: broadcast, re-layout on the slab ranks, then the frame
            d.broadcast_frame(0, raw[i % nsc][0] if rank == 0 else None)
            d.frame(mv, pr)
            return
        if repack:
            d.frame(mv, pr, new_frame=raw[i % nsc][1])
            return
        if raw_repack:
            d.frame(mv, pr, new_frame=raw_dev[i % nsc][1])
            return
        if nsc > 1:
            d.b.select_frame_slot(i % nsc)
        d.frame(mv, pr)

    # extra frame slots (throughput mode): independent contexts on their own streams, fed round robin in the timed loop
    slots = [drv]
    if args.frames_in_flight > 1:
        for _ in range(args.frames_in_flight - 1):
            h2 = make_ctx(sparse=args.sparse_pool)
            if args.preprocess:
                h2.upload_raw_frame(scene)
            slots.append(mg.SlabDriver(h2, 0, 1, f"cuda:{local}", view=VIEW, preprocess=args.preprocess))
        for k, d in enumerate(slots * 5):
            step(d, k)

    def barrier():
        for d in slots:
            d.finish()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- N > 1: the partition must reproduce the unpartitioned frame, bit for bit, before it is worth timing
    slab_check = None
    if alone_slab:
        slab_check = f"skipped: one slab ({alone_slab}, voxel planes {list(slab)}) alone is not the frame"
    elif slabs_mode:
        ok = True
        got = []
        for k in range(nsc):
            for _ in range(mg.SlabDriver.LAG + 1):         # also past the frames that gather the full capacity
                step(drv, k)
            drv.finish()
            if rank == 0:
                got.append((hip.view_images()[:3], hip.framebuffer()))
        if rank == 0:
            whole = make_ctx()
            for k in range(nsc):
                if repack:
                    whole.upload_frame(scenes[k])
                else:
                    whole.select_frame_slot(k)
                whole.clearOccupiedBricks(); whole.markBricks(); whole.updateOccupiedBricks(False); whole.integrate(); whole.drawF(mv, pr)
                (wa, wd, wn, _), (wc, wdd) = whole.view_images(), whole.framebuffer()
                (sa, sd, sn), (sc, sdd) = got[k]
                if dedicated and args.composite == "compact":
                    # the compositor did not march: the write-only sample-count image holds 0 where no slab hit (multigpu.py)
                    hitpx = wd < 1
                    ok &= same(np, sn[hitpx], wn[hitpx]) and bool(((sn == 0) | (sn == wn))[~hitpx].all())
                else:
                    ok &= same(np, sn, wn)
                ok &= same(np, sa, wa) and same(np, sd, wd) and same(np, sc, wc) and same(np, sdd, wdd) and int((wd < 1).sum()) > 1000
            whole.close()
            del whole
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=f"cuda:{local}" if backend == "nccl" else "cpu")
        if world > 1:
            dist.broadcast(flag, src=0)
        if int(flag.item()) != 1:
            raise SystemExit("slab partition does NOT reproduce the unpartitioned frame: refusing to time it")
        slab_check = (f"rank 0: raymarch colour/depth/sample counts and the hole-filled framebuffer of {nsc} frame(s) bit-identical to an unpartitioned context"
                      + (" (sample counts: at the hit pixels; the compositor does not march, the write-only count image holds 0 or the exact count elsewhere)" if dedicated and args.composite == "compact" else ""))

    # ---- N = 1: what is timed is checked too (VERDICT r03 "next" 6).  Two fresh contexts -- the lanes as shipped, and everything on one stream
    # (tsdf_set_stage_overlap(0)) -- are fed the same 8 moving frames from the start; the lanes' frames are queued back to back with no read; after
    # frame 6 and after frame 7 volume, raymarch images and the hole-filled framebuffer must be equal bit for bit, or nothing is timed.
    lane_check = None
    if world == 1 and not alone and args.frames_in_flight == 1 and (repack or raw_repack) and os.environ.get("RR_OVERLAP_FILL", "1") != "0":
        def frame_on(b, i):
            if repack:
                b.frame_dev(mv, pr, raw[i % nsc][1])
            else:
                b.frame_raw_dev(mv, pr, raw_dev[i % nsc][1])
        la, se = make_ctx(sparse=args.sparse_pool), make_ctx(sparse=args.sparse_pool)
        se.set_stage_overlap(False)
        if raw_repack:
            la.set_preprocess_calibration(scene); se.set_preprocess_calibration(scene)
        whole_volume = cfg["res"][0] * cfg["res"][1] * cfg["res"][2] <= (1 << 28)      # (a 1024^3 volume is 4 GiB per download: images only)
        ok, n_done = True, 0
        for upto in (7, 8):
            for i in range(n_done, upto):
                frame_on(la, i); frame_on(se, i)
            n_done = upto
            (a_c, a_d, a_n, _), (a_fc, a_fd) = la.view_images(), la.framebuffer()
            (b_c, b_d, b_n, _), (b_fc, b_fd) = se.view_images(), se.framebuffer()
            ok &= same(np, a_c, b_c) and same(np, a_d, b_d) and same(np, a_n, b_n) and same(np, a_fc, b_fc) and same(np, a_fd, b_fd) and int((a_fd < 1).sum()) > 1000
            if whole_volume:
                ok &= same(np, la.tsdf(), se.tsdf())
        la.close(); se.close()
        del la, se
        if not ok:
            raise SystemExit("the four-lane frame loop does NOT reproduce the one-stream frames: refusing to time it")
        lane_check = ("frames 6 and 7 of 8 moving frames queued back to back through the lanes: " + ("volume, " if whole_volume else "") +
                      "raymarch colour / depth / sample counts and the hole-filled framebuffer bit-identical to a context with every kernel on one stream")

    # host -> device frame upload, outside the timed region (value = HBM-resident rate); reported for the PCIe-inclusive figure
    hip.sync()
    if not repack and not raw_repack:
        hip.select_frame_slot(0)    # scene A's slot (the slab check above leaves the LAST slot current: uploading A there would overwrite frame B)
    tu0 = time.perf_counter()
    for _ in range(5):
        hip.upload_frame(scene)
    hip.sync()
    upload_ms = (time.perf_counter() - tu0) / 5 * 1e3
    warmup_effective = max(800, args.warmup)
    for i in range(args.warmup):
        step(drv, i)
    # The HIP runtime has a one-time stall of ~18 ms a few thousand launches into a process (measured: 600 timed steps after 20
    # warm-up frames ran 12 % slower than after 700; 200 or 30 000 steps did not show it).  Whatever W the caller asks for, run
    # the launch path into its steady state before anything is timed.
    for i in range(max(0, 800 - args.warmup)):
        step(drv, i)
    barrier()

    # Stage breakdown, OUTSIDE the timed region: every recorded HIP event costs a few microseconds of stream time, so the full set
    # of timers runs on its own short pass ...
    # The stage and kernel timers -- and with them the roofline -- come from NON-overlapped passes: by default fillColors() of frame f runs on
    # its own stream beside the brick passes and the integrate of frame f + 1 (stage overlap), which stretches every kernel it shares the
    # machine with; tsdf_set_stage_overlap(0) queues everything on the one stream for these passes, the timed region runs as shipped
    overlap = os.environ.get("RR_OVERLAP_FILL", "1") != "0" and args.frames_in_flight == 1      # (without hole filling -- c1 -- the lane ahead and the integrate lane still apply)
    every = slots if not slabs_mode else [drv]
    stages = {}
    dom = None
    serial = None
    dom_ms_serial = None
    if not args.no_timers:
        for d in every:
            d.b.set_stage_overlap(False)
        hip.set_timer_filter(None)
        hip.enable_timers(True)
        nb = max(10, min(50, args.steps))
        for i in range(nb):
            step(drv, i)
        barrier()
        hip.enable_timers(False)
        for name in ("0ingest", "0repack", "1preprocess", "bricks", "2integrate", "k_pair_masks", "k_integrate_tiles", "brickdraw", "draw", "k_march", "holefill", "3recon"):
            n, ms = hip.timer_stats(name)
            if n:
                stages[name] = ms / n
        cands = [k for k in ("k_integrate_tiles", "k_march") if k in stages]
        dom = max(cands, key=lambda k: stages[k]) if cands else None
        if world > 1 and "k_integrate_tiles" in stages:
            dom = "k_integrate_tiles"       # N > 1: every slab rank times the same kernel (a thin slab's march can outlast its integrate launch)
        # ... the dominant kernel alone (two events per frame) over args.steps non-overlapped frames: the roofline's measurement, and the
        # rate of the frame without stage overlap ...
        if dom:
            hip.timer_reserve(dom, 2 * args.steps)       # no hipEventCreate inside the timed loops
            hip.set_timer_filter([dom])
        barrier()
        stride_s = max(1, args.steps // 200)
        ts0 = time.perf_counter()
        for i in range(args.steps):
            hip.enable_timers(bool(dom) and i % stride_s == 0)
            step(drv, i)
        barrier()
        dts = time.perf_counter() - ts0
        hip.enable_timers(False)
        if dom:
            n, ms = hip.timer_stats(dom)
            dom_ms_serial = ms / n if n else None
        serial = {"value": args.steps / dts, "ms_per_step": dts / args.steps * 1e3,
                  "note": "stage overlap off (tsdf_set_stage_overlap(0)): every kernel of the frame on one stream, as in rounds 1 and 2; the roofline's kernel time comes from this pass"}
        for d in every:
            d.b.set_stage_overlap(overlap)
        for i in range(60):                               # back into the overlapped steady state (the switch drops the image-space tile history)
            step(drv, i)
    barrier()
    # ---- the timed region: exactly --steps frames as shipped (stage overlap on), no event recorded inside (an event pair per frame on the
    # context's stream costs the frame 4-8 us: the kernel timers run in the serial pass above and in the overlapped pass below)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(slots[i % len(slots)], i // len(slots) if len(slots) > 1 else i)
    t_issued = time.perf_counter() - t0                   # (the host's share: every call of the loop has returned; the device may still be working)
    barrier()
    dt = time.perf_counter() - t0
    dom_ms_insitu = None
    if not args.no_timers:                                # the dominant kernel in situ (beside the other lanes' kernels), a pass of its own
        if dom:                                           # (every rank runs the loop -- its steps hold collectives --, whether it times a kernel or not)
            hip.timer_stats(dom)                          # (drops the serial pass's samples)
        stride = max(1, args.steps // 200)
        for i in range(args.steps):
            hip.enable_timers(bool(dom) and i % stride == 0)
            step(drv, i)
        barrier()
        hip.enable_timers(False)
        if dom:
            n, ms = hip.timer_stats(dom)
            dom_ms_insitu = ms / n if n else None
    hip.enable_timers(False)
    hip.set_timer_filter(None)

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device=f"cuda:{local}" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    dt = max_over_ranks(dt)
    dom_ms = dom_ms_serial if dom_ms_serial else dom_ms_insitu

    def timed(n_steps, sel):
        barrier()
        t = time.perf_counter()
        for i in range(n_steps):
            sel(i)
        barrier()
        return max_over_ranks(time.perf_counter() - t)

    # the spread of `value`: four more windows of --steps frames of the same loop (a 20-step window is a 2.5 ms sample)
    spread = [args.steps / dt] + [args.steps / timed(args.steps, lambda i: step(slots[i % len(slots)], i // len(slots) if len(slots) > 1 else i)) for _ in range(4)]
    value_spread = {"windows": 5, "min": min(spread), "median": sorted(spread)[2], "max": max(spread),
                    "note": "frames/s of the timed region and of four more windows of --steps frames right after it"}
    # the static scene (best case of the incremental bookkeeping: nothing churns), same number of steps
    static = None
    resident = None
    if nsc > 1:
        def st(i):
            drv.frame(mv, pr, new_frame=raw[0][1] if repack else (raw_dev[0][1] if raw_repack else None))
        if not repack and not raw_repack:
            hip.select_frame_slot(0)
        timed(20, st)
        ds = timed(args.steps, st)
        static = {"value": args.steps / ds, "ms_per_step": ds / args.steps * 1e3,
                  "note": "frame A arrives every step (re-laid out every time): no tile goes stale, the dirty-tile history is a no-op"}
    if nsc > 1 and repack and not slabs_mode:
        # rounds 1 / 2 defined `value` without the per-frame re-layout: two already re-laid-out frames in the two frame slots alternate
        # (a context of its own: explicit frame-slot calls switch a context's lane ahead off)
        rh = rr.ReconIntegrationHip(scene, res=cfg["res"], brick_size=brick, limit=LIMIT, view=VIEW, device=local, sparse_pool_tiles=args.sparse_pool)
        rh.setUseBricks(cfg["use_bricks"]); rh.setSpaceSkip(cfg["skip_space"]); rh.setColorFilling(cfg["fill_holes"])
        for k, sc in enumerate(scenes):
            rh.select_frame_slot(k); rh.upload_frame(sc)

        def rs(i):
            rh.select_frame_slot(i % nsc)
            rh.clearOccupiedBricks(); rh.markBricks(); rh.updateOccupiedBricks(False); rh.integrate(); rh.drawF(mv, pr)
        with torch.cuda.stream(stream):
            for i in range(300):
                rs(i)
            rh.sync()
            tr0 = time.perf_counter()
            for i in range(args.steps):
                rs(i)
            rh.sync()
            dr = time.perf_counter() - tr0
        rh.close()
        resident = {"value": args.steps / dr, "ms_per_step": dr / args.steps * 1e3,
                    "note": "two frames already in the kernels' layout alternate (tsdf_select_frame_slot; brick passes on the context's stream): the definition of `value` in rounds 1 and 2"}
    # the same loop without the integrate lane (RR_DEEP=0 at creation: integrate() on the context's stream, one volume -- three lanes)
    three_lanes = None
    if nsc > 1 and repack and not slabs_mode and overlap and os.environ.get("RR_DEEP", "1") != "0":
        th = make_ctx(sparse=args.sparse_pool, lane_flags=rr.LANES_NO_INTEGRATE_LANE)
        for i in range(300):
            th.frame_dev(mv, pr, raw[i % nsc][1])
        th.sync()
        tt0 = time.perf_counter()
        for i in range(args.steps):
            th.frame_dev(mv, pr, raw[i % nsc][1])
        th.sync()
        dtl = time.perf_counter() - tt0
        th.close()
        three_lanes = {"value": args.steps / dtl, "ms_per_step": dtl / args.steps * 1e3,
                       "note": "a context created with tsdf_config::lane_flags = TSDF_LANES_NO_INTEGRATE_LANE: lane ahead + the context's stream (integrate, march, shade) + fill lane, one volume"}
    long_run = None
    if args.long_steps and args.frames_in_flight == 1:
        dl = timed(args.long_steps, lambda i: step(drv, i))
        long_run = {"steps": args.long_steps, "value": args.long_steps / dl, "ms_per_step": dl / args.long_steps * 1e3,
                    "note": "a second, longer pass of the same loop (the contract's timed region above is exactly --steps)"}
    # ---- the same loop from the RAW frames (f1, SURVEY.md section 8f: NetKinectArray::processTextures() in front of the path), an extra key, never `value`
    with_pre = None
    if world == 1 and not alone and repack and args.frames_in_flight == 1 and not args.sparse_pool:
        ph = make_ctx()
        ph.set_preprocess_calibration(scene)
        rd = []
        for sc in scenes:
            ts = [torch.from_numpy(np.ascontiguousarray(sc["depth_raw"], np.float32)).to(f"cuda:{local}"), torch.from_numpy(np.ascontiguousarray(sc["color"], np.uint8)).to(f"cuda:{local}")]
            rd.append((ts, tuple(t.data_ptr() for t in ts)))
        torch.cuda.synchronize()
        for i in range(300):
            ph.frame_raw_dev(mv, pr, rd[i % nsc][1])
        ph.sync()
        tp0 = time.perf_counter()
        for i in range(args.steps):
            ph.frame_raw_dev(mv, pr, rd[i % nsc][1])
        ph.sync()
        dtp = time.perf_counter() - tp0
        pre_ms, pre_kernels = None, None
        if not args.no_timers:
            names = ["k_pre_morph", "k_pre_filter", "k_pre_boundary", "k_pre_normal", "k_pre_quality"]
            ph.set_stage_overlap(False)
            ph.set_timer_filter(["1preprocess"]); ph.enable_timers(True)
            for i in range(30):
                ph.frame_raw_dev(mv, pr, rd[i % nsc][1])
            ph.sync(); ph.enable_timers(False)
            n, ms = ph.timer_stats("1preprocess")
            pre_ms = ms / n if n else None
            ph.set_timer_filter(names); ph.enable_timers(True)               # (each pass between its own events: a pass of its own, the events cost the lane time)
            for i in range(30):
                ph.frame_raw_dev(mv, pr, rd[i % nsc][1])
            ph.sync(); ph.enable_timers(False)
            # a roofline object per pass (VERDICT r03 "next" 3).  Bytes: the arrays a pass has to read and write once (N streams, P depth pixels, Pc colour
            # pixels, L texels of a 128^3 LUT); the two bilateral passes are arithmetic, not traffic: their 13 x 13 taps are counted as vector operations
            pp = ph.preprocessed()
            N_, P_, Pc_, L_ = n_streams, 640 * 480, 640 * 480, LUT ** 3
            n_box = int((pp["depth_rg"][..., 1] > 0).sum())                 # pixels the filter pass ran its taps for (inside the bounding box)
            n_valid = int(((pp["depth_b"][..., 0] > 0) & (pp["depth_b"][..., 0] < 1)).sum())   # ... and the quality pass
            # LUT bytes: the distinct texels the passes can touch.  A depth image is a surface in the LUT: the trilinear taps of a stream's pixels lie in the two
            # texel layers around it (2 x LUT^2 texels, however many pixels tap them), background pixels in the two slices at the clamped end -- never more than the LUT
            lut = lambda texel_bytes, px: min(N_ * texel_bytes * L_, N_ * texel_bytes * LUT * LUT * (2 + (2 if px else 0)))
            # (round 4: the Lab colour is evaluated by the boundary pass, for the 22 x 22 neighbourhood of every 16 x 16 block that holds a candidate pixel)
            n_cand_blocks = int(((pp["depth_rg"][..., 0] > 0) & ~(pp["depth_rg"][..., 1] > 0.65)).reshape(N_, 480 // 16, 16, 640 // 16, 16).any(axis=(2, 4)).sum())
            alg = {"k_pre_morph": (4 + 4) * N_ * P_ + (3 + 4) * N_ * Pc_,    # raw depth in, dilated depth out; RGB8 in, RGBA8 out (rides along)
                   "k_pre_filter": (4 + 8) * N_ * P_ + lut(16, n_box),       # morph image in, {filtered depth, range quality} out; cv_xyz
                   "k_pre_boundary": (8 + 8 + 4) * N_ * P_ + n_cand_blocks * 22 * 22 * (4 + 4 * 4) + lut(8, n_cand_blocks * 22 * 22),   # + per candidate block: depth, 4 colour taps and cv_uv of its Lab tile
                   "k_pre_normal": (4 + 16) * N_ * P_ + lut(16, 5 * n_valid),
                   "k_pre_quality": (4 + 8 + 16) * N_ * P_ + 16 * n_valid + lut(16, n_valid)}   # depth plane + depth_b (the silhouette) in, the packed texel out
            ops = {"k_pre_filter": 169 * 15 * n_box, "k_pre_quality": 169 * 11 * n_valid}    # vector operations (lane-instructions): taps x instructions per tap
            pre_kernels = {}
            for kname in names:
                kn, kms = ph.timer_stats(kname)
                if not kn:
                    continue
                t = kms / kn
                o = {"avg_launch_ms": t, "algorithmic_bytes": alg[kname], "bound": "hbm", "achieved": alg[kname] / (t * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg[kname] / (t * 1e-3) / 1e9 / HBM_PEAK_GBS}
                if kname in ops:                                            # 1024 SIMDs x 2.4 GHz x 32 lanes per cycle = 78.6 T lane-operations / s (half the 157.3 TFLOP/s FMA peak)
                    o.update({"bound": "valu", "vector_ops": ops[kname], "achieved": ops[kname] / (t * 1e-3) / 1e12, "peak": 78.6, "unit": "T lane-ops/s",
                              "frac": ops[kname] / (t * 1e-3) / 1e12 / 78.6, "hbm_frac": alg[kname] / (t * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "note": "13 x 13 bilateral: taps x vector instructions per tap over the pixels that run them (the scene's %d / %d); only whole waves issue, so the SIMDs' real "
                                      "issue share is higher (valu_issue_frac from the counter pass where profiles/traffic.json holds one)" % (n_box if kname == "k_pre_filter" else n_valid, N_ * P_)})
                    vi = measured_traffic(args.config + "_preprocess", kname, "valu_insts")
                    if vi:
                        o["valu_issue_frac"] = vi * 2.0 / (1024 * 2.4e9 * t * 1e-3)
                tr = measured_traffic(args.config + "_preprocess", kname)
                if tr:
                    o["traffic"] = tr
                pre_kernels[kname] = o
        ph.close()
        del ph, rd
        with_pre = {"value": args.steps / dtp, "ms_per_step": dtp / args.steps * 1e3, "preprocess_ms": pre_ms, "kernels": pre_kernels,
                    "note": "tsdf_frame_raw_dev: every step takes the other RAW frame (depth in metres + RGB8, resident in HBM) through pre_morph / pre_depth (13 x 13 bilateral) / "
                            "pre_boundary (+ RGB -> Lab where it compares colours) / pre_normal (marks the bricks) / pre_quality on the lane ahead, then the frame as in `value`; preprocess_ms: the five "
                            "passes + the range cells on one stream (HIP events)"}
    device_mem_bytes = int(mem_free_before - torch.cuda.mem_get_info(local)[0])
    # per-frame device time distribution (SURVEY.md section 8d asks for median and p95): one event pair per frame, own short pass
    frame_ms = None
    if not args.no_timers:
        hip.timer_reserve("frame", 100)
        hip.set_timer_filter(["frame"])
        hip.enable_timers(True)
        for i in range(100):
            if nsc > 1 and not repack and not raw_repack:
                hip.select_frame_slot(i % nsc)
            hip.timer_begin("frame"); step(drv, i); hip.timer_end_after_fill("frame")
        barrier()
        hip.enable_timers(False)
        hip.set_timer_filter(None)
        smp = np.sort(hip.timer_samples("frame"))
        if smp.size:
            frame_ms = {"frames": int(smp.size), "median": float(np.median(smp)), "p95": float(smp[min(smp.size - 1, int(0.95 * smp.size))]),
                        "min": float(smp[0]), "max": float(smp[-1]),
                        "note": "a frame's LATENCY on rank 0: HIP events from its first kernel to the end of its hole filling (which, with stage overlap, runs on its own stream "
                                "beside the next frame's brick passes and integrate: latency > 1 / rate); separate pass after the timed region"}
    ratio = hip.occupiedRatio()
    stages_rank0 = None
    if dedicated:
        # rank 0 neither marks bricks nor integrates: the volume-side stage times and the occupancy come from the first worker
        infos = [None] * world
        dist.all_gather_object(infos, {"stage_ms": stages, "ratio": ratio})
        stages_rank0, stages, ratio = stages, infos[1]["stage_ms"], infos[1]["ratio"]
    out = {
        "metric": "frames/sec (integrate+raymarch) at %d^3 x %d streams" % (cfg["res"][0], n_streams),
        "value": args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_effective": warmup_effective,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["name"], "config": args.config, "streams": n_streams, "res": list(cfg["res"]),
                   "view": list(VIEW), "limit": LIMIT, "occupied_brick_ratio": ratio, "preprocess": bool(args.preprocess),
                   "scene": (("a NEW frame every step (two frames resident in HBM as delivered -- depth RG32F, quality, silhouette, RGB8 -- alternate; each is re-laid "
                              "out by tsdf_upload_frame_dev inside the step); objects moved: every active tile churns") if repack else
                             "two resident frames alternating every step (objects moved: every active tile churns)") if nsc > 1 else "one static frame",
                   "frames_in_flight": args.frames_in_flight,
                   "storage": ("sparse pool: %d of %d tiles in use" % hip.sparse_pool_stats()) if args.sparse_pool else "dense",
                   "parallelism": ((f"single GPU holding slab {alone_slab} of the volume, " if alone_slab else "single GPU, ") + "slab exchange over RCCL with one rank (rehearsal)" if alone else "single GPU") if world == 1 else
                                  (f"ONE volume in {world - 1} Z-slab(s) on ranks 1..{world - 1} + rank 0 as dedicated compositor (receive, composite, hole filling)" if dedicated else
                                   f"ONE volume in {world} Z-slabs") + f" (strong scaling; slab boundaries {partition_note}; this rank's voxel planes {list(slab)}), halo {args.halo}, RCCL {args.composite} hit gather to rank 0, no host sync per frame; "
                                  + ("collectives called from inside the library (comm.cpp)" if args.exchange == "native" else "collectives through torch.distributed")
                                  + ("; every new frame is broadcast from rank 0 inside the step" if bcast else "; both frames resident on every rank")},
        "stage_overlap": bool(overlap),
        "lanes": ("lane ahead (re-layout + brick passes of frame f+2) | integrate lane (classify, pair masks, integrate of f+1; two alternating volume sets) | "
                  "the context's stream (depth limits, march, shade of f) | fill lane (hole filling of f-1)") if overlap and os.environ.get("RR_DEEP", "1") != "0" and not slabs_mode
                 else ("lane ahead | the context's stream (integrate, depth limits, march, shade) | fill lane" if overlap else "one stream"),
        "host_issue_ms_per_step": t_issued / args.steps * 1e3,
        "hole_filling": dict(zip(("passes", "by_dirty_tiles"), hip.fill_stats())),
        "value_spread": value_spread,
        "lane_check": lane_check,
        "with_preprocess": with_pre,
        "device_mem_bytes": device_mem_bytes,
        "device_mem_note": "device memory taken since just before the context was created (hipMemGetInfo): the context -- with the lanes on, TWO volume sets (tsdf_config / "
                           "tsdf_set_stage_overlap), two frame slots, two pyramids -- plus this script's resident input frames",
        "three_lanes": three_lanes,
        "serial": serial,
        "static": static,
        "resident_frames": resident,
        "long_run": long_run,
        "stage_ms": stages,
        **({"stage_ms_compositor": stages_rank0} if stages_rank0 is not None else {}),
        "frame_device_ms": frame_ms,
        "stage_ms_note": "per-stage device time from a separate all-timers pass before the timed region (every recorded event costs ~1 us of stream time, so their sum "
                         "exceeds ms_per_step a little): 0repack + bricks + 2integrate + 3recon = the frame; k_pair_masks / k_integrate_tiles lie inside 2integrate, "
                         "brickdraw / draw / k_march / holefill inside 3recon",
        "upload_ms_per_frame": upload_ms,
        "pcie_inclusive_frames_per_s": 1e3 / (upload_ms + dt / args.steps * 1e3),
    }
    if slabs_mode:
        out["slab_check"] = slab_check
        out["regathers"] = drv.regathers
        out["overflowed_frames"] = drv.overflowed_frames   # frames composited from truncated record lists and not repaired: must be 0 (exit code below)
    # N > 1, an extra key, never `value`: what the same N GPUs deliver as N independent single-GPU frame streams (every frame rebuilds
    # the volume from scratch, so frames are independent units: no exchange at all, weak scaling by construction)
    if world > 1:
        rep = make_ctx()
        rdrv = mg.SlabDriver(rep, 0, 1, f"cuda:{local}", view=VIEW)
        for i in range(50):
            step(rdrv, i)
        rdrv.finish()
        barrier()
        tr0 = time.perf_counter()
        for i in range(args.steps):
            step(rdrv, i)
        rdrv.finish()
        barrier()
        dtr = max_over_ranks(time.perf_counter() - tr0)
        out["frame_replicas"] = {"value": world * args.steps / dtr, "unit": "frames/s", "scaling": "weak",
                                 "note": f"{world} unpartitioned contexts, one per GPU, each fusing its own frames: no data-path collective. Reported beside `value` "
                                         "(the Z-slab partition of ONE volume), never as it"}
        rep.close()

    # ---- roofline of the dominant kernel: ALGORITHMIC bytes of the units the launch really processes / its measured time
    def roofline(kname, alg, ms, cfgname, note):
        ach = alg / (ms * 1e-3) / 1e9
        kn = "k_integrate_tiles_lds" if kname == "k_integrate_tiles" else "k_march"
        traffic = measured_traffic(cfgname, kn)
        valu = measured_traffic(cfgname, kn, "valu_insts")
        return {"bound": "hbm", "kernel": "k_integrate_tiles_rec (the LDS form, record head)" if kn == "k_integrate_tiles_lds" else kn, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic, "algorithmic_bytes": alg, "avg_launch_ms": ms,
                "traffic_frac": (traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                # share of the SIMDs' vector issue capacity: a SIMD-32 issues one wave64 VALU instruction per 2 cycles (MI355X_MICROARCH.md,
                # "Wave scheduling"), 1024 SIMDs at 2.4 GHz
                "valu_issue_frac": (valu * 2.0 / (1024 * 2.4e9 * ms * 1e-3)) if valu else None,
                "timing": "HIP events around this kernel alone, recorded on the launch stream, over a pass of --steps frames with stage overlap off (the timed region "
                          "itself records no event: a pair per frame costs it 4-8 us)", "note": note}

    db = dense_bytes(cfg["res"], n_streams)
    if dom_ms and world == 1 and not alone:
        if dom == "k_integrate_tiles" and cfg["use_bricks"]:
            alg, tiles_n = culled_integrate_bytes(np, hip, scenes, cfg["res"], n_streams)
            note = (f"units = active 8^3-voxel tiles of the launch ({tiles_n} in the timed frames): 2 KiB stored per tile + 16 B x N x distinct inverse-LUT "
                    f"texels of their boxes + 16 B x valid depth pixels; the DENSE figure of BASELINE.md section 3 ({db['integrate']} B) does not apply to a culled launch")
        elif dom == "k_integrate_tiles":
            alg, note = db["integrate"], "dense launch: 4V + N*16*L + N*16*P (BASELINE.md section 3)"
        else:
            alg, note = db["march"], "dense march: 4V + 24R (BASELINE.md section 3); with depth limits the rays sample only inside occupied bricks, so this is an upper figure"
        out["roofline"] = roofline(dom, alg, dom_ms, args.config, note)
        out["roofline"]["avg_launch_ms_overlapped"] = dom_ms_insitu           # in situ: beside the other lanes' kernels (stage overlap on), a pass of its own after the timed region
        if dom == "k_integrate_tiles" and "k_pair_masks" in stages:
            # the launch reads its (tile, stream) pair classes from the pair-mask pass that runs right before it: kernel + helper together
            both = dom_ms + stages["k_pair_masks"]
            out["roofline"]["with_helper"] = {"kernels": "k_pair_masks + k_integrate_tiles_lds", "ms": both, "frac": alg / (both * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              "note": f"k_pair_masks {stages['k_pair_masks'] * 1e3:.1f} us from the all-timers pass (its two events included)"}
        if cfg["use_bricks"] and cfg["fill_holes"] and args.frames_in_flight == 1:
            # the whole frame: algorithmic bytes of every stage (with the culled launches counted by the units they process)
            if dom != "k_integrate_tiles":
                alg_i, tiles_n = culled_integrate_bytes(np, hip, scenes, cfg["res"], n_streams)
            else:
                alg_i = alg
            P, Pc, R = 640 * 480, 640 * 480, VIEW[0] * VIEW[1]
            T = sum(tiles_n) / len(tiles_n)
            parts = {"repack": ((16 * P + 3 * Pc) + (20 * P + 4 * Pc)) * n_streams if repack else 0,      # arrays in, packed texel + depth plane + RGBA8 out
                     "bricks": 4 * n_streams * P,                                                        # K6 reads the depth plane
                     "integrate": alg_i,
                     "raymarch": 2048 * T + 24 * R + 15 * n_streams * P,                                 # 4V of SURVEY 8d -> the active tiles' 2 KiB
                     "inpaint_colorfill": 67 * R}
            fb = float(sum(parts.values()))
            ms = dt / args.steps * 1e3
            out["roofline_frame"] = {"bound": "hbm", "bytes": fb, "stage_bytes": parts, "ms_per_step": ms, "achieved": fb / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": fb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "traffic": measured_traffic(args.config, "frame"), "traffic_if_streaming": measured_traffic(args.config, "frame", "hbm_bytes_if_streaming"),
                                     "note": "sum of the stages' algorithmic bytes (SURVEY.md section 8d; culled launches by the units they process) / the timed step; "
                                             "traffic = FETCH_SIZE + WRITE_SIZE summed over the frame's kernels (separate --pmc passes on one stream, profiles/traffic.json; "
                                             "FETCH_SIZE counts half of a coalesced stream: traffic_if_streaming doubles it) -- below the algorithmic figure because the hole "
                                             "filling keeps to the dirty screen tiles while the formula counts every pixel of every level"}
    if world > 1:
        # N > 1: the same object for the slowest slab launch (every rank that owns a slab measures its own; counters -- `traffic` -- were
        # only collected on one GPU).  Only culled integrate launches: the units are the slab's active tiles (halo layers it recomputes included)
        mine = None
        owns_slab = not (dedicated and rank == 0)
        if owns_slab and dom_ms and dom == "k_integrate_tiles" and cfg["use_bricks"]:
            alg, tiles_n = culled_integrate_bytes(np, hip, scenes, cfg["res"], n_streams)
            mine = {"rank": rank, "ms": dom_ms, "alg": alg, "tiles": tiles_n, "slab": list(slab)}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        allr = [r for r in allr if r]
        if allr:
            r = max(allr, key=lambda r: r["ms"])
            out["roofline"] = roofline("k_integrate_tiles", r["alg"], r["ms"], "none", f"the slowest slab launch of the {len(allr)} slab ranks: rank {r['rank']}, voxel planes {r['slab']}, "
                                       f"{r['tiles']} active tiles in the timed frames (the halo layers it recomputes included); bytes as at N = 1 with the frame's valid depth pixels "
                                       "counted in full for every slab (an upper figure); other ranks: " + ", ".join(f"rank {q['rank']} {q['ms'] * 1e3:.1f} us / {q['alg'] / 1e6:.1f} MB" for q in allr if q is not r))
    # ---- the dense configuration (configs[1]): both kernels against the dense byte counts, in the same line
    if world == 1 and not alone and not args.no_c1 and not args.no_timers and args.config != "c1" and args.frames_in_flight == 1:
        c1 = CONFIGS["c1"]
        sc1 = scene if n_streams == c1["streams"] else rr.scene.make_scene(n_streams=c1["streams"], width=640, height=480, lut_res=LUT, inv_res=LUT)
        ext1 = sc1["bbox_max"] - sc1["bbox_min"]
        h1 = rr.ReconIntegrationHip(sc1, res=c1["res"], brick_size=[float(ext1[a]) / c1["res"][a] * 8 for a in range(3)], limit=LIMIT, view=VIEW, device=local)
        h1.setUseBricks(False); h1.setSpaceSkip(False); h1.setColorFilling(False)

        def f1():
            h1.clearOccupiedBricks(); h1.markBricks(); h1.updateOccupiedBricks(False); h1.integrate(); h1.drawF(mv, pr)
        with torch.cuda.stream(stream):
            for _ in range(30):
                f1()
            h1.set_timer_filter(["k_integrate_tiles", "k_march"]); h1.enable_timers(True)
            for _ in range(100):
                f1()
            h1.sync()
            h1.enable_timers(False)
            h1.sync()
            tc0 = time.perf_counter()
            for _ in range(100):
                f1()
            h1.sync()
            c1_ms = (time.perf_counter() - tc0) / 100 * 1e3
        d1 = dense_bytes(c1["res"], c1["streams"])
        r1 = {"workload": c1["name"], "ms_per_frame": c1_ms, "frames_per_s": 1e3 / c1_ms}
        for tname, key in (("k_integrate_tiles", "integrate"), ("k_march", "march")):
            n, ms = h1.timer_stats(tname)
            if n:
                r1[key] = roofline(tname, d1[key], ms / n, "c1", "dense launch, dense bytes of BASELINE.md section 3")
        out["roofline_c1"] = r1
        h1.close()
    if args.ingest and world == 1:
        dfmt, cfmt = args.ingest.split("-")
        cf, df = {"rgb8": rr.COLOR_RGB8, "dxt1": rr.COLOR_DXT1, "dxt5": rr.COLOR_DXT5}[cfmt], {"f32": rr.DEPTH_F32, "u8": rr.DEPTH_U8}[dfmt]
        msg = rr.scene.make_wire_message(scene, cf, df, timestamp=1.0)
        hip.setWireFormat(cf, df)
        for i in range(n_streams):
            hip.setDepthCompression(i, df == rr.DEPTH_U8, 0.5, 4.5)
        if df == rr.DEPTH_U8:
            # pre_morph.fs validates the normalised 8-bit codes against 0.5..4.5 "metres" (reference quirk):
            # with processed depth on, everything nearer than code 128 is dropped.  Run the 8-bit path the way it works.
            hip.setPreprocess(processed_depth=False)
        k = max(10, args.steps // 4)
        for _ in range(3):
            hip.upload_wire_frame(msg, scene); drv.frame(mv, pr)
        barrier()
        hip.set_timer_filter(["0ingest"])
        hip.enable_timers(not args.no_timers)
        t0 = time.perf_counter()
        for _ in range(k):
            hip.upload_wire_frame(msg)
            drv.frame(mv, pr)
        barrier()
        dti = (time.perf_counter() - t0) / k
        hip.enable_timers(False)
        n_u, ms_u = hip.timer_stats("0ingest") if not args.no_timers else (0, 0.0)
        out["ingest"] = {"format": args.ingest, "message_bytes": len(msg), "frames": k, "wire_to_frame_ms": dti * 1e3,
                         "wire_inclusive_frames_per_s": 1.0 / dti, "gpu_unpack_ms": (ms_u / n_u) if n_u else None,
                         "note": "host message -> pinned copy -> H2D -> GPU unpack/DXT decode -> pre-process -> integrate -> drawF, "
                                 "one frame in flight; `value` above stays the HBM-resident rate"}
    # ---- PCIe-inclusive rate with the upload of frame f + 1 overlapping the compute of frame f (double-buffered frame slots,
    # pinned staging filled in place as the reference's reader thread fills the mapped PBO): never `value`
    if world == 1 and not alone and not args.preprocess and args.frames_in_flight == 1:
        with torch.cuda.stream(stream):
            for k in range(2):                            # both staging buffers hold a frame (the producer's job)
                st_d, st_q, st_s, st_c = hip.frame_staging()
                sc = scenes[k % nsc]
                st_d[...] = sc["depth"]; st_q[...] = sc["quality"]; st_s[...] = sc["silhouette"]; st_c[...] = sc["color"]
                hip.upload_frame_async(None)
                hip.select_frame_slot(hip.current_frame_slot() ^ 1)
            hip.sync()
            ko = max(50, args.steps)
            for phase in range(2):
                t0 = time.perf_counter()
                for i in range(ko):
                    hip.upload_frame_async(None)          # frame f + 1 starts travelling ...
                    drv.frame(mv, pr)                      # ... while frame f is computed
                    hip.select_frame_slot(hip.current_frame_slot() ^ 1)
                hip.sync()
                dto = (time.perf_counter() - t0) / ko
        out["pcie_overlapped"] = {"frames_per_s": 1.0 / dto, "ms_per_frame": dto * 1e3, "bytes_per_frame": int(scene["depth"].nbytes + scene["quality"].nbytes + scene["silhouette"].nbytes + scene["color"].nbytes),
                                  "note": "tsdf_upload_frame_async of the next frame (pinned staging -> copy stream -> other frame slot) overlapped with the compute of the current one"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scene, cfg, LIMIT, brick)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or alone:
        dist.destroy_process_group()
    if slabs_mode and drv.overflowed_frames:
        raise SystemExit(f"{drv.overflowed_frames} frame(s) were composited from truncated hit lists (SlabDriver.frame_status tells which): the rate above counts wrong frames")


if __name__ == "__main__":
    main()
