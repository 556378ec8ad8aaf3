"""Where the host spends its time while it issues c2 frames: mean wall time of every call of the frame loop (no synchronisation in the loop).
A call whose mean approaches the frame period is one the runtime blocks in.  python tools/host_calls.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rgbd_recon_amd as rr
import bench
cfg = bench.CONFIGS["c2"]
mk = dict(n_streams=cfg["streams"], width=640, height=480, lut_res=bench.LUT, inv_res=bench.LUT)
scs = [rr.scene.make_scene(**mk), rr.scene.make_scene(**mk, **bench.MOVED)]
ext = scs[0]["bbox_max"] - scs[0]["bbox_min"]
hip = rr.ReconIntegrationHip(scs[0], res=cfg["res"], brick_size=[float(ext[k]) / cfg["res"][k] * 8 for k in range(3)], limit=bench.LIMIT, view=bench.VIEW)
mv, pr = rr.scene.default_view(*bench.VIEW)
raw = [[torch.from_numpy(np.ascontiguousarray(sc[k])).cuda() for k in ("depth", "quality", "silhouette", "color")] for sc in scs]
ptr = [[t.data_ptr() for t in r] for r in raw]
torch.cuda.synchronize()
calls = [("upload_frame_dev", lambda i: hip.upload_frame_dev(*ptr[i & 1], complete=True)), ("clear", lambda i: hip.clearOccupiedBricks()), ("mark", lambda i: hip.markBricks()),
         ("update", lambda i: hip.updateOccupiedBricks(False)), ("integrate", lambda i: hip.integrate()), ("drawF", lambda i: hip.drawF(mv, pr))]
for i in range(1500):
    for _, fn in calls: fn(i)
hip.sync()
acc = {n: 0.0 for n, _ in calls}
N = 2000
t0 = time.perf_counter()
for i in range(N):
    for n, fn in calls:
        t = time.perf_counter(); fn(i); acc[n] += time.perf_counter() - t
t1 = time.perf_counter(); hip.sync(); t2 = time.perf_counter()
print({n: round(v / N * 1e6, 1) for n, v in acc.items()}, "issue", round((t1 - t0) / N * 1e6, 1), "us/frame; drained after", round((t2 - t1) * 1e6, 1), "us", flush=True)
