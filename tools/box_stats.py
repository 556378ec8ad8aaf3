"""What a dense march (k_march_box) does per frame at c1: batches, boxes, samples from LDS / global, leaps (tools/build_k1_variant.sh boxstats "-DRR_BOX_STATS" k_raymarch).
    RGBDR_LIB=build_variants/lib_boxstats.so python tools/box_stats.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import rgbd_recon_amd as rr
import bench
cfg = bench.CONFIGS["c1"]
sc = rr.scene.make_scene(n_streams=cfg["streams"], width=640, height=480, lut_res=bench.LUT, inv_res=bench.LUT)
hip = rr.ReconIntegrationHip(sc, res=cfg["res"], limit=bench.LIMIT, view=bench.VIEW)
hip.setUseBricks(cfg["use_bricks"]); hip.setSpaceSkip(cfg["skip_space"]); hip.setColorFilling(cfg["fill_holes"])
hip.set_stage_overlap(False)
mv, pr = rr.scene.default_view(*bench.VIEW)
hip.clearOccupiedBricks(); hip.markBricks(); hip.updateOccupiedBricks(False); hip.integrate()
L = rr.load_library()
out = (C.c_uint64 * 8)()
hip.draw(mv, pr); hip.sync()
L.tsdf_debug_box_stats(out, 1)
N = 5
for _ in range(N):
    hip.draw(mv, pr)
hip.sync()
assert L.tsdf_debug_box_stats(out, 0) == 0
names = ["batches", "batches with a box", "sum of S over boxes", "sample rounds from LDS (x kSub)", "sample rounds from global (x kSub)", "box floats", "retries (S halved)", "samples skipped (all-clear boxes + leaps)"]
waves = (bench.VIEW[0] // 8) * (bench.VIEW[1] // 8)
for n, v in zip(names, out):
    print(f"{n:45s} {v / N:12.0f} per frame   {v / N / waves:8.2f} per wave")
