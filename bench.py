#!/usr/bin/env python3
"""bench.py -- frames/s of integrate() + drawF() on the synthetic scene of SURVEY.md §8d.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one frame of the reference's per-frame call order (source/kinect_client.cpp:569-599,614):
clearOccupiedBricks -> mark_brick (K6) -> updateOccupiedBricks -> integrate (K0+K1) -> drawF
(K5 depth limits, K2 raymarch, K3/K4 hole filling), with the frame images already resident in HBM.
N > 1, --parallel auto (default): the reference rebuilds the volume from scratch every frame, so frames are independent
units -- when the volume fits one GPU (every BASELINE configuration does) the ranks each fuse their own frames of the stream,
no data-path collective, weak scaling; a volume that does not fit one GPU is Z-slab partitioned instead.  --parallel slabs
forces the north-star partition: the SAME volume split into Z-slabs over the ranks (strong scaling), halo layers recomputed
or RCCL all-gathered before the raymarch, nearest-hit gather of the partial images (rgbd-recon_amd/multigpu.py; DESIGN.md
section 6 has the measured cost floor of that exchange against the 0.22 ms frame).

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
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "traffic.json")   # per-kernel HBM bytes from separate rocprofv3 --pmc passes


def measured_traffic(config, kernel, key="hbm_bytes"):
    """HBM bytes (or, key="valu_insts", VALU wave-instructions) per launch of `kernel` from the committed PMC summary
    (profiles/traffic.json, made by profiles/summarize_pmc.py from separate FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU passes with the
    gfx950 corrections); None if absent."""
    try:
        with open(TRAFFIC_JSON) as f:
            return json.load(f).get(config, {}).get(kernel, {}).get(key)
    except (OSError, ValueError):
        return None


def algorithmic_bytes(cfg, n_streams, world):
    """BASELINE.md §3 / SURVEY.md §8d dense byte counts, per launch of each kernel on ONE rank."""
    V = cfg["res"][0] * cfg["res"][1] * cfg["res"][2] // world
    L = LUT ** 3 // world
    P = 640 * 480
    R = VIEW[0] * VIEW[1]
    integrate = 4 * V + n_streams * 16 * L + n_streams * 16 * P
    raymarch = 4 * V + 24 * R + n_streams * 15 * P
    march = 4 * V + 24 * R                     # k_march alone: the volume + the per-pixel peel/hit records (k_shade reads the images)
    inpaint = 67 * R
    return dict(integrate=integrate, raymarch=raymarch, march=march, inpaint=inpaint)


def cpu_baseline(scene, cfg, limit, brick):
    """The oracle (kind "port": the reference has no CPU path) on the same workload at full size: three frames of
    clear/mark/update bricks + integrate() + drawF() at the bench view, the fastest one reported (the first frame pays the
    first touch of the volume)."""
    from oracle.oracle import OracleRecon
    import rgbd_recon_amd as rr
    cores = os.cpu_count() or 1
    o = OracleRecon(scene, res=cfg["res"], brick_size=brick, limit=limit, view=VIEW)
    o.setUseBricks(cfg["use_bricks"]); o.setSpaceSkip(cfg["skip_space"]); o.setColorFilling(cfg["fill_holes"])
    mv, pr = rr.scene.default_view(*VIEW)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        o.clearOccupiedBricks(); o.markBricks(); o.updateOccupiedBricks()
        o.integrate()
        t1 = time.perf_counter()
        o.drawF(mv, pr)
        t2 = time.perf_counter()
        if best is None or t2 - t0 < best[0]:
            best = (t2 - t0, t1 - t0, t2 - t1)
    return {"value": 1.0 / best[0], "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle/libtsdf_oracle.so, OpenMP {cores} threads, fastest of 3 full-size frames: bricks+integrate {best[1]:.2f} s, "
                      f"drawF at {VIEW[0]}x{VIEW[1]} {best[2]:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--halo", default="recompute", choices=["recompute", "exchange"], help="N > 1: integrate the halo layers locally, or RCCL all-gather them")
    ap.add_argument("--composite", default="compact", choices=["compact", "dense"], help="N > 1: gather hit records, or whole partial images")
    ap.add_argument("--parallel", default="auto", choices=["auto", "slabs", "frames"],
                    help="N > 1: 'frames' = every GPU fuses its own frames of the stream (each frame rebuilds the volume from scratch, so frames "
                         "are independent: no exchange at all, weak scaling); 'slabs' = ONE volume split into Z-slabs with the RCCL exchange "
                         "(the north-star partition, strong scaling); 'auto' = frames while the dense volume fits half of one GPU's HBM, else slabs")
    ap.add_argument("--frames-in-flight", type=int, default=1, choices=[1, 2, 3, 4],
                    help="single GPU: process this many frames concurrently, one context + HIP stream per slot (every frame rebuilds the "
                         "volume from scratch, so frames are independent and the frame's 17 small latency-bound kernels overlap well). "
                         "A throughput mode: the latency of a frame does not improve, and per-kernel times (roofline) are measured under "
                         "contention.  Default 1")
    ap.add_argument("--sparse-pool", type=int, default=0, metavar="TILES",
                    help="store the TSDF in a sparse pool of this many 8^3-voxel tiles (2 KiB each) instead of a dense array "
                         "(BASELINE.json configs[4] 'sparse-brick allocation'); needs a culled configuration")
    ap.add_argument("--preprocess", action="store_true", help="also run the image pre-processing passes (f1) every frame, from the raw depth/colour")
    ap.add_argument("--ingest", default=None, choices=["f32-rgb8", "f32-dxt1", "u8-rgb8", "u8-dxt1", "u8-dxt5"],
                    help="also measure the wire path (f2): every frame arrives as one host message, is copied through the pinned double "
                         "buffer, unpacked/decoded on the GPU and pre-processed (implies --preprocess); reported beside `value`, never as it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-timers", action="store_true", help="leave the per-kernel HIP event timers off")
    args = ap.parse_args()
    if args.ingest:
        args.preprocess = True

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

    cfg = CONFIGS[args.config]
    limit = 0.01
    scene = rr.scene.make_scene(n_streams=cfg["streams"], width=640, height=480, lut_res=LUT, inv_res=LUT)
    ext = scene["bbox_max"] - scene["bbox_min"]
    brick = [float(ext[a]) / cfg["res"][a] * 8 for a in range(3)]          # 8^3 voxels per brick
    if alone:
        args.parallel = "slabs"
    if args.parallel == "auto":                # 4 B per voxel against half of the 288 GB of one MI355X: every BASELINE configuration -> frames
        fits = 4 * cfg["res"][0] * cfg["res"][1] * cfg["res"][2] <= 144e9
        args.parallel = "frames" if fits else "slabs"
    frames_mode = world > 1 and args.parallel == "frames"
    slab = mg.slab_range(cfg["res"][2], rank, world) if ((world > 1 and not frames_mode) or alone) else (0, 0)
    hip = rr.ReconIntegrationHip(scene, res=cfg["res"], brick_size=brick, limit=limit, view=VIEW, device=local, slab=slab,
                                 recompute_halo=(args.halo == "recompute"), sparse_pool_tiles=args.sparse_pool)
    hip.setUseBricks(cfg["use_bricks"]); hip.setSpaceSkip(cfg["skip_space"]); hip.setColorFilling(cfg["fill_holes"])
    stream = torch.cuda.current_stream()
    hip.set_stream(stream.cuda_stream)         # kernels, HIP event timers and the collectives share one stream
    if args.preprocess:
        hip.upload_raw_frame(scene)
    drv = mg.SlabDriver(hip, 0 if frames_mode else rank, 1 if frames_mode else world, f"cuda:{local}", view=VIEW, halo=args.halo, composite=args.composite,
                        preprocess=args.preprocess, exchange_when_alone=alone)
    mv, pr = rr.scene.default_view(*VIEW)
    # extra frame slots (throughput mode): independent contexts on their own streams, fed round robin in the timed loop
    slots = [drv]
    if args.frames_in_flight > 1:
        if world > 1 or args.ingest:
            raise SystemExit("--frames-in-flight is a single-GPU option (and not combined with --ingest)")
        for _ in range(args.frames_in_flight - 1):
            h2 = rr.ReconIntegrationHip(scene, res=cfg["res"], brick_size=brick, limit=limit, view=VIEW, device=local, sparse_pool_tiles=args.sparse_pool)
            h2.setUseBricks(cfg["use_bricks"]); h2.setSpaceSkip(cfg["skip_space"]); h2.setColorFilling(cfg["fill_holes"])
            s2 = torch.cuda.Stream()
            h2.set_stream(s2.cuda_stream)
            if args.preprocess:
                h2.upload_raw_frame(scene)
            slots.append(mg.SlabDriver(h2, 0, 1, f"cuda:{local}", view=VIEW, preprocess=args.preprocess))
            slots[-1]._stream = s2                  # keep the torch stream alive
        for d in slots * 5:
            d.frame(mv, pr)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # host -> device frame upload, outside the timed region (value = HBM-resident rate); reported for the PCIe-inclusive figure
    hip.sync()
    tu0 = time.perf_counter()
    for _ in range(5):
        hip.upload_frame(scene)
    hip.sync()
    upload_ms = (time.perf_counter() - tu0) / 5 * 1e3
    for _ in range(args.warmup):
        drv.frame(mv, pr)
    # The HIP runtime has a one-time stall of ~18 ms a few thousand launches into a process (measured: 600 timed steps after 20
    # warm-up frames ran 12 % slower than after 700; 200 or 30 000 steps did not show it).  Whatever W the caller asks for, run
    # the launch path into its steady state before anything is timed.
    for _ in range(max(0, 800 - args.warmup)):
        drv.frame(mv, pr)
    barrier()

    # Stage breakdown, OUTSIDE the timed region: every recorded HIP event costs a few microseconds of stream time (17 launches
    # and 9 nested timers per frame add ~15 % to a 0.29 ms frame), so the full set of timers runs on its own short pass ...
    stages = {}
    dom = None
    if not args.no_timers:
        hip.set_timer_filter(None)
        hip.enable_timers(True)
        nb = max(10, min(50, args.steps))
        for _ in range(nb):
            drv.frame(mv, pr)
        barrier()
        hip.enable_timers(False)
        for name in ("0ingest", "1preprocess", "bricks", "2integrate", "k_integrate_tiles", "brickdraw", "draw", "k_march", "holefill", "3recon"):
            n, ms = hip.timer_stats(name)
            if n:
                stages[name] = ms / n
        cands = [k for k in ("k_integrate_tiles", "k_march") if k in stages]
        dom = max(cands, key=lambda k: stages[k]) if cands else None
        # ... and the timed region records only the dominant kernel's two events per frame (the roofline's live measurement)
        if dom:
            hip.timer_reserve(dom, args.steps)       # no hipEventCreate inside the timed loop
            hip.set_timer_filter([dom])
            hip.enable_timers(True)
    barrier()
    # at most ~200 event pairs in flight: many hundreds of un-synchronised events slow the launch path down (600 steps with an
    # event pair each ran 12 % slower than 200), so long runs time the dominant kernel on every stride-th frame
    stride = max(1, args.steps // 200)
    timing = bool(dom)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if timing and stride > 1:
            hip.enable_timers(i % stride == 0)
        slots[i % len(slots)].frame(mv, pr)
    barrier()
    dt = time.perf_counter() - t0
    hip.enable_timers(False)
    hip.set_timer_filter(None)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    dom_ms = None
    if dom:
        n, ms = hip.timer_stats(dom)
        dom_ms = ms / n if n else None
    # per-frame device time distribution (SURVEY.md section 8d asks for median and p95): one event pair per frame, own short pass
    frame_ms = None
    if not args.no_timers:
        hip.timer_reserve("frame", 100)
        hip.set_timer_filter(["frame"])
        hip.enable_timers(True)
        for _ in range(100):
            hip.timer_begin("frame"); drv.frame(mv, pr); hip.timer_end("frame")
        barrier()
        hip.enable_timers(False)
        hip.set_timer_filter(None)
        smp = np.sort(hip.timer_samples("frame"))
        if smp.size:
            frame_ms = {"frames": int(smp.size), "median": float(np.median(smp)), "p95": float(smp[min(smp.size - 1, int(0.95 * smp.size))]),
                        "min": float(smp[0]), "max": float(smp[-1]), "note": "HIP events around whole frames on rank 0, separate pass after the timed region"}
    ratio = hip.occupiedRatio()
    ab = algorithmic_bytes(cfg, cfg["streams"], 1 if frames_mode else world)
    out = {
        "metric": "frames/sec (integrate+raymarch) at %d^3 x %d streams" % (cfg["res"][0], cfg["streams"]),
        "value": (world if frames_mode else 1) * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if args.parallel == "frames" else "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["name"], "config": args.config, "streams": cfg["streams"], "res": list(cfg["res"]),
                   "view": list(VIEW), "limit": limit, "occupied_brick_ratio": ratio, "preprocess": bool(args.preprocess),
                   "frames_in_flight": args.frames_in_flight,
                   "storage": ("sparse pool: %d of %d tiles in use" % hip.sparse_pool_stats()) if args.sparse_pool else "dense",
                   "parallelism": ("single GPU, slab exchange over RCCL with one rank (rehearsal)" if alone else "single GPU") if world == 1 else (f"{world} GPUs, frame-parallel: each rank fuses its own frames of the stream, no data-path collective (--parallel slabs = Z-slab partition of one volume)" if frames_mode else
                                                                    f"{world} Z-slabs, halo {args.halo}, RCCL {args.composite} hit gather to rank 0")},
        "stage_ms": stages,
        "frame_device_ms": frame_ms,
        "stage_ms_note": "per-stage device time from a separate all-timers pass before the timed region (not part of `value`)",
        "upload_ms_per_frame": upload_ms,
        "pcie_inclusive_frames_per_s": 1e3 / (upload_ms + dt / args.steps * 1e3),
    }
    if dom_ms:
        key = "integrate" if dom == "k_integrate_tiles" else "march"
        ach = ab[key] / (dom_ms * 1e-3) / 1e9
        kname = "k_integrate_tiles_lds" if key == "integrate" else "k_march"
        traffic = measured_traffic(args.config, kname)
        valu = measured_traffic(args.config, kname, "valu_insts")
        out["roofline"] = {"bound": "hbm", "kernel": kname,
                           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": traffic, "algorithmic_bytes": ab[key], "avg_launch_ms": dom_ms,
                           "traffic_frac": (traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                           # what actually binds this kernel: VALU wave-instructions (PMC) x 4 issue cycles over the
                           # 1024 SIMDs' cycles at 2.4 GHz during the launch
                           # share of the SIMDs' vector issue capacity: a SIMD-32 issues one wave64 VALU instruction per 2 cycles (MI355X_MICROARCH.md,
                           # "Wave scheduling"), 1024 SIMDs at 2.4 GHz
                           "valu_issue_frac": (valu * 2.0 / (1024 * 2.4e9 * dom_ms * 1e-3)) if valu else None,
                           "timing": "HIP events around this kernel alone, recorded on the launch stream in every frame of the timed region",
                           "note": "algorithmic bytes are the DENSE figures of BASELINE.md section 3; with brick culling the launch touches "
                                   "only occupied tiles (occupied_brick_ratio), so achieved may exceed what HBM really moved (traffic)"}
        frame_bytes = ab["integrate"] + ab["raymarch"] + (ab["inpaint"] if cfg["fill_holes"] else 0)
        out["frame_roofline"] = {"algorithmic_bytes": frame_bytes, "achieved": frame_bytes / (dt / args.steps) / 1e9,
                                 "frac": frame_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s"}
    if args.ingest and world == 1:
        dfmt, cfmt = args.ingest.split("-")
        cf, df = {"rgb8": rr.COLOR_RGB8, "dxt1": rr.COLOR_DXT1, "dxt5": rr.COLOR_DXT5}[cfmt], {"f32": rr.DEPTH_F32, "u8": rr.DEPTH_U8}[dfmt]
        msg = rr.scene.make_wire_message(scene, cf, df, timestamp=1.0)
        hip.setWireFormat(cf, df)
        for i in range(cfg["streams"]):
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
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scene, cfg, limit, brick)
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or alone:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
