"""tools/host_calls2.py : host time of each call of the c2 frame loop (perf_counter around the binding calls), fourth lane on / off (RR_DEEP)."""
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
calls = [("upload", lambda i: hip.upload_frame_dev(*ptr[i & 1], complete=True)), ("clear", lambda i: hip.clearOccupiedBricks()), ("mark", lambda i: hip.markBricks()),
         ("update", lambda i: hip.updateOccupiedBricks(False)), ("integrate", lambda i: hip.integrate()), ("draw", lambda i: hip.draw(mv, pr)), ("fill", lambda i: hip.fillColors())]
for i in range(800):
    for _, f in calls:
        f(i)
hip.sync()
tot = {n: 0.0 for n, _ in calls}
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40       # (short: the queues stay far from full)
for rep in range(10):
    for i in range(N):
        for n, f in calls:
            t = time.perf_counter(); f(i); tot[n] += time.perf_counter() - t
    hip.sync()
print(N, "frames per burst; host us per call:", {n: round(1e6 * v / (10 * N), 1) for n, v in tot.items()}, "sum", round(1e6 * sum(tot.values()) / (10 * N), 1))
