"""Long-run determinism soak: two DIFFERENT frames (objects moved, so every tile churns; another view) alternating N times in one
context; TSDF + framebuffer must hash identically to the first frame of their kind every CHECK frames (the per-frame state --
alternating tile lists, re-armed device counters, image-space dirty tiles, double-buffered brick counters, exact tile classes, frame
slots -- is self-cleaning; a race or a stale counter would show as a drifting hash).
    python tools/soak.py [N] [CHECK] [c2|c1]"""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from importlib import import_module
rr = import_module("rgbd-recon_amd")

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
CHECK = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
CFG = sys.argv[3] if len(sys.argv) > 3 else "c2"
VIEW = (1280, 720)
res = 512 if CFG == "c2" else 256
mk = dict(n_streams=4, width=640, height=480, lut_res=128, inv_res=128)
scene = rr.scene.make_scene(**mk)
scene_b = rr.scene.make_scene(sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2), **mk)      # bench.py's frame B
ext = scene["bbox_max"] - scene["bbox_min"]
hip = rr.ReconIntegrationHip(scene, res=(res,) * 3, brick_size=[float(ext[a]) / res * 8 for a in range(3)], limit=0.01, view=VIEW)
dense = CFG != "c2"
hip.setUseBricks(not dense); hip.setSpaceSkip(not dense); hip.setColorFilling(not dense)
# round 3: every frame ARRIVES (device arrays -> tsdf_upload_frame_dev), so the three lanes of the context -- the lane ahead (re-layout +
# brick passes of frame f + 1), the context's stream (integrate / march / shade of frame f) and the fill lane (hole filling of frame f - 1) --
# run against each other for the whole soak; SOAK_SLOTS=1 selects round 2's flow (two resident frames, explicit frame slots: no lane ahead)
SLOTS = os.environ.get("SOAK_SLOTS") == "1"
RAW = os.environ.get("SOAK_RAW") == "1"                  # round 4: the RAW frames through tsdf_frame_raw_dev (pre-processing on the lane ahead, its first two passes in front of the gate)
if RAW:
    hip.set_preprocess_calibration(scene)
    raw = [[torch.from_numpy(np.ascontiguousarray(sc["depth_raw"], np.float32)).cuda(), torch.from_numpy(np.ascontiguousarray(sc["color"], np.uint8)).cuda()] for sc in (scene_b, scene)]
    ptr = [[t.data_ptr() for t in r] for r in raw]
    torch.cuda.synchronize()
elif SLOTS:
    hip.select_frame_slot(1); hip.upload_frame(scene_b); hip.select_frame_slot(0)
else:
    raw = [[torch.from_numpy(np.ascontiguousarray(sc[k])).cuda() for k in ("depth", "quality", "silhouette", "color")] for sc in (scene_b, scene)]
    ptr = [[t.data_ptr() for t in r] for r in raw]
    torch.cuda.synchronize()
mv, pr = rr.scene.default_view(*VIEW)
mv_b = rr.scene.gl_flat(rr.scene.look_at((1.6, 1.4, 2.4), (0.0, 1.1, 0.0)))


def digest():
    c, d = hip.framebuffer()
    return hashlib.sha1(c.tobytes() + d.tobytes() + hip.tsdf().tobytes()).hexdigest()


ref = {}
t0 = time.perf_counter()
for f in range(1, N + 1):
    which = ((f + 1) >> 1) & 1                            # A A B B ...: with two volume sets alternating per integrate() (the fourth lane) each set still sees A, B, A, B
    if RAW:
        hip.frame_raw_dev(mv if which else mv_b, pr, new_frame=(ptr[which][0], ptr[which][1]), complete=True)
    else:
        if SLOTS:
            hip.select_frame_slot(0 if which else 1)
        else:
            hip.upload_frame_dev(*ptr[which], complete=True)
        hip.clearOccupiedBricks(); hip.markBricks(); hip.updateOccupiedBricks(False); hip.integrate(); hip.drawF(mv if which else mv_b, pr)
    if f <= 4 or f % CHECK in (0, 1, 2, 3):
        h = digest()
        ref.setdefault(which, h)
        print(f"frame {f} ({'A' if which else 'B'}): {h[:16]} {'ok' if h == ref[which] else 'DIFFERENT'}  ({time.perf_counter() - t0:.1f} s)", flush=True)
        assert h == ref[which], f"frame {f} differs from the first frame of its kind"
print("soak ok:", N, "frames", CFG)
