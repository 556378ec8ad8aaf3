"""tools/lane_spans.py [frames] : where the stages of consecutive c2 frames lie on the device clock in the shipped mode (all lanes), from the
context's own HIP-event timers (no profiler: under rocprofv3 the host cannot keep the lanes fed).  Prints the stages of three steady-state
frames, sorted by begin, and the frame period."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rr = importlib.import_module("rgbd-recon_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
mk = dict(n_streams=4, width=640, height=480, lut_res=128, inv_res=128)
scenes = [rr.scene.make_scene(**mk), rr.scene.make_scene(sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2), **mk)]
ext = scenes[0]["bbox_max"] - scenes[0]["bbox_min"]
res = int(os.environ.get("RES", 512))
hip = rr.ReconIntegrationHip(scenes[0], res=(res,) * 3, brick_size=[float(ext[a]) / res * 8 for a in range(3)], limit=0.01, view=(1280, 720))
dense = os.environ.get("DENSE") == "1"
hip.setUseBricks(not dense); hip.setSpaceSkip(not dense); hip.setColorFilling(not dense)
raw = [[torch.from_numpy(np.ascontiguousarray(sc[k])).cuda() for k in ("depth", "quality", "silhouette", "color")] for sc in scenes]
ptr = [[t.data_ptr() for t in r] for r in raw]
torch.cuda.synchronize()
mv, pr = rr.scene.default_view(1280, 720)


RAW = os.environ.get("RAW", "0")           # 1: raw frames through tsdf_frame_raw_dev (its first two passes in front of the lane's gate); 2: the same frames through the separate calls (gate first)
if RAW != "0":
    hip.set_preprocess_calibration(scenes[0])
    rawf = [[torch.from_numpy(np.ascontiguousarray(sc["depth_raw"], np.float32)).cuda(), torch.from_numpy(np.ascontiguousarray(sc["color"], np.uint8)).cuda()] for sc in scenes]
    rptr = [[t.data_ptr() for t in r] for r in rawf]
    torch.cuda.synchronize()


def step(i):
    if RAW == "1":
        hip.frame_raw_dev(mv, pr, new_frame=(rptr[i & 1][0], rptr[i & 1][1]), complete=True)
        return
    if RAW == "2":
        hip.upload_raw_frame_dev(rptr[i & 1][0], rptr[i & 1][1], complete=True); hip.clearOccupiedBricks(); hip.processTextures()
        hip.updateOccupiedBricks(False); hip.integrate(); hip.drawF(mv, pr)
        return
    hip.upload_frame_dev(*ptr[i & 1], complete=True)
    hip.clearOccupiedBricks(); hip.markBricks(); hip.updateOccupiedBricks(False); hip.integrate(); hip.drawF(mv, pr)


for i in range(400):
    step(i)
hip.sync()
names = os.environ.get("SPANS", "1preprocess,bricks,2integrate,brickdraw,draw,holefill" if RAW != "0" else "0repack,bricks,2integrate,brickdraw,draw,holefill").split(",")
hip.set_timer_filter(names)
hip.enable_timers(True)
for n in names:
    hip.timer_reserve(n, N + 8)
for i in range(N):
    step(i)
hip.sync()
spans = []
for n in names:
    b, e = hip.timer_spans(n, names[0])
    for k, (x, y) in enumerate(zip(b, e)):
        spans.append((float(x) * 1e3, float(y) * 1e3, n, k))
spans.sort()
f0 = N - 8
t0 = [s for s in spans if s[2] == names[0] and s[3] == f0][0][0]
for b, e, n, k in spans:
    if f0 <= k < f0 + 4 and b >= t0:
        lane = {"0repack": 0, "1preprocess": 0, "bricks": 0, "2integrate": 1, "k_pair_masks": 1, "k_integrate_tiles": 1, "brickdraw": 2, "draw": 2, "k_march": 2, "holefill": 3}[n]
        print(f"{b - t0:8.1f} {e - t0:8.1f} {e - b:7.1f}   {'    ' * lane}{n}({k})")
rb = sorted(s[0] for s in spans if s[2] == names[0])
print("frame period us (last 10 frames):", (rb[-1] - rb[-11]) / 10.0)
