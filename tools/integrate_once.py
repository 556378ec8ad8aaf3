"""configs[1] dense integrate() x N (profiling target): python tools/integrate_once.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import rgbd_recon_amd as rr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
scene = rr.scene.make_scene(n_streams=4, width=640, height=480, lut_res=128, inv_res=128)
ext = scene["bbox_max"] - scene["bbox_min"]
hip = rr.ReconIntegrationHip(scene, res=(256,) * 3, brick_size=[float(ext[a]) / 256 * 8 for a in range(3)], limit=0.01, view=(1280, 720))
hip.setUseBricks(False); hip.setSpaceSkip(False); hip.setColorFilling(False)
for _ in range(N):
    hip.integrate()
hip.sync()
