"""Host cost of issuing one frame (Python + ctypes + HIP launches / event operations): the frame loop on a problem so small that the GPU is never
the limit.  python tools/host_cost2.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rgbd_recon_amd as rr
sc = rr.scene.make_scene(n_streams=4, width=32, height=24, lut_res=8, inv_res=8)
hip = rr.ReconIntegrationHip(sc, res=(32, 32, 32), brick_size=[2.0 / 4, 2.2 / 4, 2.0 / 4], limit=0.05, view=(64, 36))
mv, pr = rr.scene.default_view(64, 36)
raw = [torch.from_numpy(np.ascontiguousarray(sc[k])).cuda() for k in ("depth", "quality", "silhouette", "color")]
ptr = [t.data_ptr() for t in raw]
torch.cuda.synchronize()
def full():
    hip.upload_frame_dev(*ptr, complete=True); hip.clearOccupiedBricks(); hip.markBricks(); hip.updateOccupiedBricks(False); hip.integrate(); hip.drawF(mv, pr)
for label in ("three lanes", "one stream"):
    if label == "one stream":
        hip.set_stage_overlap(False)
    for _ in range(2000): full()
    hip.sync(); t = time.perf_counter()
    for _ in range(5000): full()
    t1 = time.perf_counter(); hip.sync(); t2 = time.perf_counter()
    print(f"{label}: issue {1e6 * (t1 - t) / 5000:.1f} us/frame, with the final sync {1e6 * (t2 - t) / 5000:.1f} us/frame", flush=True)
