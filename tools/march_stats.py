"""Per-wave timing of the slab march (instrumented library: -DRR_MARCH_STATS): python tools/march_stats.py k n   (slab k of n, configs[2])"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import rgbd_recon_amd as rr
from importlib import import_module
mg = import_module("rgbd-recon_amd.multigpu")
k, n = int(sys.argv[1]), int(sys.argv[2])
scene = rr.scene.make_scene(n_streams=4, width=640, height=480, lut_res=128, inv_res=128, **(dict(sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2)) if len(sys.argv) > 3 else {}))   # third argument: bench.py's frame B
ext = scene["bbox_max"] - scene["bbox_min"]
VIEW = (1280, 720)
hip = rr.ReconIntegrationHip(scene, res=(512,) * 3, brick_size=[float(ext[a]) / 512 * 8 for a in range(3)], limit=0.01, view=VIEW, slab=mg.slab_range(512, k, n), recompute_halo=True)
mv, pr = rr.scene.default_view(*VIEW)
L = hip._L
out = (C.c_ulonglong * 8)()
for f in range(6):
    hip.clearOccupiedBricks(); hip.markBricks(); hip.updateOccupiedBricks(False); hip.integrate()
    hip.sync(); L.tsdf_debug_march_stats(out, 1)
    hip.draw(mv, pr); hip.sync()
L.tsdf_debug_march_stats(out, 0)
v = [int(x) for x in out]
pix = v[0] & 0xffffffff
print(f"slab {k}/{n}: slowest wave {(v[0] >> 32) / 100:.1f} us at pixel ({pix % VIEW[0]}, {pix // VIEW[0]}); max pre-run steps {v[1]}, max batches {v[2]}, max max_n {v[3]}, working waves {v[4]}, mean wave {v[5] / max(v[4], 1) / 100:.2f} us")
