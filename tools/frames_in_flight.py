"""Experiment: throughput with several independent frames in flight on ONE GPU (one context + stream per frame slot).
Every frame rebuilds the volume from scratch, so consecutive frames are independent; the frame's 17 kernels are small and
latency bound, so two frames overlap well.  Latency per frame does not improve -- this is a throughput mode."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rgbd_recon_amd as rr
VIEW = (1280, 720)
scene = rr.scene.make_scene(n_streams=4, width=640, height=480, lut_res=128, inv_res=128)
ext = scene["bbox_max"] - scene["bbox_min"]
res = (512, 512, 512)
brick = [float(ext[a]) / res[a] * 8 for a in range(3)]
mv, pr = rr.scene.default_view(*VIEW)
for k in (1, 2, 3, 4):
    ctxs, streams = [], []
    for i in range(k):
        h = rr.ReconIntegrationHip(scene, res=res, brick_size=brick, limit=0.01, view=VIEW)
        s = torch.cuda.Stream()
        h.set_stream(s.cuda_stream)
        ctxs.append(h); streams.append(s)
    def frame(h):
        h.clearOccupiedBricks(); h.markBricks(); h.updateOccupiedBricks(False); h.integrate(); h.drawF(mv, pr)
    for i in range(40): frame(ctxs[i % k])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 1200
    for i in range(n): frame(ctxs[i % k])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{k} frame(s) in flight: {dt * 1e3:.4f} ms per frame, {1 / dt:.0f} frames/s", flush=True)
    del ctxs, streams
