"""N moving-scene frames of c2 exactly as bench.py's timed step issues them (new frame from device memory every step): profiling target"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rgbd_recon_amd as rr
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = bench.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "c2"]
mk = dict(n_streams=cfg["streams"], width=640, height=480, lut_res=bench.LUT, inv_res=bench.LUT)
scs = [rr.scene.make_scene(**mk), rr.scene.make_scene(**mk, **bench.MOVED)]
ext = scs[0]["bbox_max"] - scs[0]["bbox_min"]
hip = rr.ReconIntegrationHip(scs[0], res=cfg["res"], brick_size=[float(ext[k]) / cfg["res"][k] * 8 for k in range(3)], limit=bench.LIMIT, view=bench.VIEW)
hip.setUseBricks(cfg["use_bricks"]); hip.setSpaceSkip(cfg["skip_space"]); hip.setColorFilling(cfg["fill_holes"])
mv, pr = rr.scene.default_view(*bench.VIEW)
raw = [[torch.from_numpy(np.ascontiguousarray(sc[k])).cuda() for k in ("depth", "quality", "silhouette", "color")] for sc in scs]
ptr = [[t.data_ptr() for t in r] for r in raw]
torch.cuda.synchronize()
for i in range(N):
    hip.upload_frame_dev(*ptr[i & 1], complete=True); hip.clearOccupiedBricks(); hip.markBricks(); hip.updateOccupiedBricks(False); hip.integrate(); hip.drawF(mv, pr)
hip.sync()
