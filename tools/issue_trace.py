"""tools/issue_trace.py : host time of each of 60 consecutive tsdf_frame_dev calls after a sync (how far may the host run ahead of the device?)"""
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
for i in range(800):
    hip.frame_dev(mv, pr, ptr[i & 1])
hip.sync()
ts = [time.perf_counter()]
for i in range(60):
    hip.frame_dev(mv, pr, ptr[i & 1])
    ts.append(time.perf_counter())
hip.sync()
te = time.perf_counter()
d = [1e6 * (b - a) for a, b in zip(ts, ts[1:])]
print("per-call host us:", " ".join(f"{x:.0f}" for x in d))
print(f"all issued after {1e6 * (ts[-1] - ts[0]):.0f} us, device done after {1e6 * (te - ts[0]):.0f} us")
