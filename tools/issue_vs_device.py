"""tools/issue_vs_device.py : is the host the limit?  The c2 frame loop as in bench.py's timed region: host time to ISSUE n frames against the
time until the device has finished them (the HIP queues take thousands of packets: the host runs ahead unless it is the slower side)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rr = importlib.import_module("rgbd-recon_amd")
mk = dict(n_streams=4, width=640, height=480, lut_res=128, inv_res=128)
scenes = [rr.scene.make_scene(**mk), rr.scene.make_scene(sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2), **mk)]
ext = scenes[0]["bbox_max"] - scenes[0]["bbox_min"]
res = 512
hip = rr.ReconIntegrationHip(scenes[0], res=(res,) * 3, brick_size=[float(ext[a]) / res * 8 for a in range(3)], limit=0.01, view=(1280, 720))
raw = [[torch.from_numpy(np.ascontiguousarray(sc[k])).cuda() for k in ("depth", "quality", "silhouette", "color")] for sc in scenes]
ptr = [[t.data_ptr() for t in r] for r in raw]
torch.cuda.synchronize()
mv, pr = rr.scene.default_view(1280, 720)


ONE = os.environ.get("ONE_CALL", "1") != "0"           # tsdf_frame_dev: the frame in one call into the library


def step(i):
    if ONE:
        hip.frame_dev(mv, pr, ptr[i & 1])
        return
    hip.upload_frame_dev(*ptr[i & 1], complete=True)
    hip.clearOccupiedBricks(); hip.markBricks(); hip.updateOccupiedBricks(False); hip.integrate(); hip.drawF(mv, pr)


for i in range(800):
    step(i)
hip.sync()
for n in (10, 20, 40, 200, 1000, 3000):
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    t1 = time.perf_counter()
    hip.sync()
    t2 = time.perf_counter()
    print(f"{'one call' if ONE else 'seven calls'}, {n} frames: issued in {1e6 * (t1 - t0) / n:.1f} us/frame, finished after {1e6 * (t2 - t0) / n:.1f} us/frame", flush=True)
