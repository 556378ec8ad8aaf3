"""What a frame would cost if the re-layout of the new frame and the brick passes ran on a lane of their own, a frame ahead: the period of
[integrate + drawF] alone (static brick state; results are those of a static scene) against the full step.  python tools/lane_potential.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rgbd_recon_amd as rr
import bench
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
mk = dict(n_streams=cfg["streams"], width=640, height=480, lut_res=bench.LUT, inv_res=bench.LUT)
a = rr.scene.make_scene(**mk)
ext = a["bbox_max"] - a["bbox_min"]
hip = rr.ReconIntegrationHip(a, res=cfg["res"], brick_size=[float(ext[k]) / cfg["res"][k] * 8 for k in range(3)], limit=bench.LIMIT, view=bench.VIEW)
hip.setUseBricks(cfg["use_bricks"]); hip.setSpaceSkip(cfg["skip_space"]); hip.setColorFilling(cfg["fill_holes"])
mv, pr = rr.scene.default_view(*bench.VIEW)
raw = [torch.from_numpy(np.ascontiguousarray(a[k])).cuda() for k in ("depth", "quality", "silhouette", "color")]
ptr = [t.data_ptr() for t in raw]
def full():
    hip.upload_frame_dev(*ptr, complete=True); hip.clearOccupiedBricks(); hip.markBricks(); hip.updateOccupiedBricks(False); hip.integrate(); hip.drawF(mv, pr)
def main_only():
    hip.integrate(); hip.drawF(mv, pr)
def block():
    hip.select_frame_slot(hip.current_frame_slot())          # an explicit frame-slot call: the lane ahead is switched off for good
    full()
for name, fn in (("full step", full), ("integrate + drawF only", main_only), ("full step", full), ("full step, lane ahead off", block), ("full step, lane ahead off", full)):
    for _ in range(300): fn()
    hip.sync(); t = time.perf_counter()
    for _ in range(1000): fn()
    hip.sync(); dt = (time.perf_counter() - t) / 1000
    print(f"{name}: {dt * 1e6:.1f} us/frame, {1 / dt:.0f} frames/s", flush=True)
