"""Experiment: does capturing the c2 frame (17 dependent launches) in a HIP graph shorten it?  Not part of the product."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rgbd_recon_amd as rr

VIEW = (1280, 720)
scene = rr.scene.make_scene(n_streams=4, width=640, height=480, lut_res=128, inv_res=128)
ext = scene["bbox_max"] - scene["bbox_min"]
res = (512, 512, 512)
brick = [float(ext[a]) / res[a] * 8 for a in range(3)]
hip = rr.ReconIntegrationHip(scene, res=res, brick_size=brick, limit=0.01, view=VIEW)
s = torch.cuda.Stream()
hip.set_stream(s.cuda_stream)
mv, pr = rr.scene.default_view(*VIEW)

def frame():
    hip.clearOccupiedBricks(); hip.markBricks(); hip.updateOccupiedBricks(False); hip.integrate(); hip.drawF(mv, pr)

with torch.cuda.stream(s):
    for _ in range(20): frame()
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): frame()
    s.synchronize()
    print("eager ms/frame", (time.perf_counter() - t0) / 200 * 1e3)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            frame(); frame()                      # two frames: the hit-list parity alternates
        for _ in range(10): g.replay()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(100): g.replay()
        s.synchronize()
        print("graph ms/frame", (time.perf_counter() - t0) / 200 * 1e3)
        fb1 = hip.framebuffer()
        frame(); frame(); s.synchronize()
        fb2 = hip.framebuffer()
        print("graph result equals eager:", all(np.array_equal(a, b, equal_nan=True) for a, b in zip(fb1, fb2)))
    except Exception as e:
        print("capture failed:", repr(e)[:500])
