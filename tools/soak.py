"""Long-run determinism soak: two different c2 frames (scene, view) alternating N times; the TSDF and the framebuffer must hash identically every CHECK frames
(the per-frame state -- alternating tile lists, re-armed device counters, image-space dirty tiles, double-buffered brick
counters -- is self-cleaning; a race or a stale counter would show as a drifting hash).   python tools/soak.py [N] [CHECK]"""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from importlib import import_module
rr = import_module("rgbd-recon_amd")

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
CHECK = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
VIEW = (1280, 720)
scene = rr.scene.make_scene(n_streams=4, width=640, height=480, lut_res=128, inv_res=128)
ext = scene["bbox_max"] - scene["bbox_min"]
hip = rr.ReconIntegrationHip(scene, res=(512, 512, 512), brick_size=[float(ext[a]) / 512 * 8 for a in range(3)], limit=0.01, view=VIEW)
hip.setUseBricks(True); hip.setSpaceSkip(True); hip.setColorFilling(True)
mv, pr = rr.scene.default_view(*VIEW)
scene_b = rr.scene.make_scene(n_streams=4, width=640, height=480, lut_res=128, inv_res=128, seed=77)     # another object, another view
mv_b = rr.scene.gl_flat(rr.scene.look_at((1.6, 1.4, 2.4), (0.0, 1.1, 0.0)))


def digest():
    c, d = hip.framebuffer()
    return hashlib.sha1(c.tobytes() + d.tobytes() + hip.tsdf().tobytes()).hexdigest()


ref = {}
t0 = time.perf_counter()
for f in range(1, N + 1):
    which = f & 1
    hip.upload_frame(scene if which else scene_b)
    hip.clearOccupiedBricks(); hip.markBricks(); hip.updateOccupiedBricks(False); hip.integrate(); hip.drawF(mv if which else mv_b, pr)
    if f <= 2 or f % CHECK in (0, 1):
        h = digest()
        ref.setdefault(which, h)
        print(f"frame {f} ({'A' if which else 'B'}): {h[:16]} {'ok' if h == ref[which] else 'DIFFERENT'}  ({time.perf_counter() - t0:.1f} s)", flush=True)
        assert h == ref[which], f"frame {f} differs from the first frame of its kind"
print("soak ok:", N, "frames")
