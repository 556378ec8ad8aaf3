"""One-off: the bench scene in a 4096^3 volume (256 GiB of dense voxels) through the sparse tile pool."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rgbd_recon_amd as rr
VIEW = (1280, 720)
scene = rr.scene.make_scene(n_streams=4, width=640, height=480, lut_res=128, inv_res=128)
ext = scene["bbox_max"] - scene["bbox_min"]
brick = [float(ext[a]) / 512 * 8 for a in range(3)]
mv, pr = rr.scene.default_view(*VIEW)
for res, pool in ((1024, 65536), (2048, 400000), (4096, 2600000)):
    hip = rr.ReconIntegrationHip(scene, res=(res,) * 3, brick_size=brick, limit=0.01, view=VIEW, sparse_pool_tiles=pool)
    def frame():
        hip.clearOccupiedBricks(); hip.markBricks(); hip.updateOccupiedBricks(False); hip.integrate(); hip.drawF(mv, pr)
    for _ in range(3): frame()
    hip.sync(); t0 = time.perf_counter()
    for _ in range(20): frame()
    hip.sync(); dt = (time.perf_counter() - t0) / 20
    need, cap = hip.sparse_pool_stats()
    hits = int((hip.framebuffer()[1] < 1).sum())
    print(f"{res}^3: dense {res**3 * 4 / 2**30:.0f} GiB -> {need} tiles = {need * 2048 / 2**30:.2f} GiB in use (pool {cap}); {dt * 1e3:.3f} ms/frame; {hits} covered pixels", flush=True)
    del hip
