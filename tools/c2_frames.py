"""N moving-scene frames of a bench configuration, nothing else (profiling target): python tools/c2_frames.py [N] [config]
RGBDR_LIB selects a variant library (tools/build_variant.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import rgbd_recon_amd as rr
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = bench.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "c2"]
mk = dict(n_streams=cfg["streams"], width=640, height=480, lut_res=bench.LUT, inv_res=bench.LUT)
a, b = rr.scene.make_scene(**mk), rr.scene.make_scene(**mk, **bench.MOVED)
ext = a["bbox_max"] - a["bbox_min"]
hip = rr.ReconIntegrationHip(a, res=cfg["res"], brick_size=[float(ext[k]) / cfg["res"][k] * 8 for k in range(3)], limit=bench.LIMIT, view=bench.VIEW)
hip.setUseBricks(cfg["use_bricks"]); hip.setSpaceSkip(cfg["skip_space"]); hip.setColorFilling(cfg["fill_holes"])
hip.select_frame_slot(1); hip.upload_frame(b); hip.select_frame_slot(0)
mv, pr = rr.scene.default_view(*bench.VIEW)
for i in range(N):
    hip.select_frame_slot(i & 1)
    hip.clearOccupiedBricks(); hip.markBricks(); hip.updateOccupiedBricks(False); hip.integrate(); hip.drawF(mv, pr)
hip.sync()
print(hip.integrate_stats())
