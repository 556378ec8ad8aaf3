"""Timing of the point back-end at the bench's frame size (4 x 640x480 points into 1280x720)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rgbd_recon_amd as rr
VIEW = (1280, 720)
scene = rr.scene.make_scene(n_streams=4, width=640, height=480, lut_res=128, inv_res=128)
hip = rr.ReconIntegrationHip(scene, res=(64, 64, 64), brick_size=0.25, limit=0.04, view=VIEW)
hip.upload_normals(scene["normals"]); hip.setShadeMode(1)
mv, pr = rr.scene.default_view(*VIEW)
for _ in range(10): hip.drawPoints(mv, pr)
hip.sync(); t0 = time.perf_counter()
for _ in range(200): hip.drawPoints(mv, pr)
hip.sync(); print("drawPoints ms", (time.perf_counter() - t0) / 200 * 1e3, "covered pixels", (hip.framebuffer()[1] < 1).sum())
for _ in range(5): hip.drawTrigrid(mv, pr)
hip.sync(); t0 = time.perf_counter()
for _ in range(100): hip.drawTrigrid(mv, pr)
hip.sync(); print("drawTrigrid ms", (time.perf_counter() - t0) / 100 * 1e3, "covered pixels", (hip.framebuffer()[1] < 1).sum())
