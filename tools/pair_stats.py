"""How many (tile, stream) pairs of an integrate launch the uniform-pair shortcut skips (instrumented library: -DRR_PAIR_STATS): python tools/pair_stats.py [c1|c2|c3|c4]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import rgbd_recon_amd as rr
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
res, n, dense = {"c1": (256, 4, True), "c2": (512, 4, False), "c3": (512, 8, False), "c4": (1024, 8, False)}[cfg]
mk = dict(n_streams=n, width=640, height=480, lut_res=128, inv_res=128)
for name, extra in (("frame A", {}), ("frame B", dict(sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2)))):
    scene = rr.scene.make_scene(**mk, **extra)
    ext = scene["bbox_max"] - scene["bbox_min"]
    hip = rr.ReconIntegrationHip(scene, res=(res,) * 3, brick_size=[float(ext[a]) / res * 8 for a in range(3)], limit=0.01, view=(64, 36))
    hip.setUseBricks(not dense)
    out = (C.c_ulonglong * 4)()
    hip._L.tsdf_debug_pair_stats(out, 1)
    hip.clearOccupiedBricks(); hip.markBricks(); hip.updateOccupiedBricks(False); hip.integrate(); hip.sync()
    hip._L.tsdf_debug_pair_stats(out, 0)
    it, pr, un, al = (int(x) for x in out)
    print(f"{cfg} {name}: {it} work items, {pr} pairs, {un} uniform ({100.0 * un / max(pr, 1):.1f} %), {al} items with every pair uniform ({100.0 * al / max(it, 1):.1f} %)")
    hip.close()
