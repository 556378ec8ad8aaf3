"""Host enqueue time per c2 frame vs the wall time of the frame loop (is the loop host-bound?): python tools/host_cost.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import rgbd_recon_amd as rr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
scene = rr.scene.make_scene(n_streams=4, width=640, height=480, lut_res=128, inv_res=128)
moved = rr.scene.make_scene(n_streams=4, width=640, height=480, lut_res=128, inv_res=128, sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2))
ext = scene["bbox_max"] - scene["bbox_min"]
hip = rr.ReconIntegrationHip(scene, res=(512,) * 3, brick_size=[float(ext[a]) / 512 * 8 for a in range(3)], limit=0.01, view=(1280, 720))
hip.upload_frame_async(moved); hip.sync()
mv, pr = rr.scene.default_view(1280, 720)
def frame(i):
    hip.select_frame_slot(i & 1)
    hip.clearOccupiedBricks(); hip.markBricks(); hip.updateOccupiedBricks(False); hip.integrate(); hip.drawF(mv, pr)
for i in range(50): frame(i)
hip.sync()
t0 = time.perf_counter()
for i in range(N): frame(i)
t1 = time.perf_counter()
hip.sync()
t2 = time.perf_counter()
hip.sync()
best = 1e9
for rep in range(20):                      # short bursts after a sync: the queue never fills, so this is the host's own cost
    t3 = time.perf_counter()
    for i in range(20): frame(i)
    best = min(best, (time.perf_counter() - t3) / 20)
    hip.sync()
print(f"host cost of a frame's calls (queue empty): {1e6 * best:.1f} us")
print(f"overlap={os.environ.get('RR_STAGE_OVERLAP', '1')} host enqueue {1e6 * (t1 - t0) / N:.1f} us/frame, loop incl. final sync {1e6 * (t2 - t0) / N:.1f} us/frame, tail wait {1e3 * (t2 - t1):.2f} ms")
res, rb, bs = hip.res, hip.res_bricks, hip.brick_size
print("res", res, "bricks", rb, "brick size", bs)
