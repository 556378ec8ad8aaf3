"""Ray length histogram of the c2 bench frame (diagnostic, not part of the product)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rgbd_recon_amd as rr
VIEW = (1280, 720)
scene = rr.scene.make_scene(n_streams=4, width=640, height=480, lut_res=128, inv_res=128)
ext = scene["bbox_max"] - scene["bbox_min"]
res = (512, 512, 512)
brick = [float(ext[a]) / res[a] * 8 for a in range(3)]
hip = rr.ReconIntegrationHip(scene, res=res, brick_size=brick, limit=0.01, view=VIEW)
mv, pr = rr.scene.default_view(*VIEW)
hip.clearOccupiedBricks(); hip.markBricks(); hip.updateOccupiedBricks(); hip.integrate(); hip.drawF(mv, pr)
rgba, d, ns, pe = hip.view_images()
n = np.rint(np.abs(ns) / 0.0027).astype(int)
print("pixels", n.size, "with samples", (n > 0).sum(), "total samples", n.sum(), "max", n.max(), "hits", (d < 1).sum())
for lo, hi in [(1, 8), (9, 16), (17, 24), (25, 32), (33, 48), (49, 64), (65, 128), (129, 10000)]:
    m = (n >= lo) & (n <= hi)
    print(f"{lo:4d}-{hi:5d}: {m.sum():7d} rays, {n[m].sum():8d} samples")
# per 8x8 wave tile: max ray length
H, W = n.shape
t = n[:H // 8 * 8, :W // 8 * 8].reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))
print("wave tiles", t.size, "active", (t > 0).sum(), "sum of per-wave max", t.sum(), "tiles with max>16:", (t > 16).sum(), ">32:", (t > 32).sum(), ">64:", (t > 64).sum())
