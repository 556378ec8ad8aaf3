"""Where do the ~16 us of a tile go?  An instrumented build of the culled LDS integrate kernel (tools/build_k1_variant.sh trace "-DRR_K1_TRACE") stamps
s_memtime at the phase boundaries of the LAST tile each workgroup processes (lane 0 of the workgroup); this runs c2 frames through it and prints the mean
span of every segment over the workgroups, in shader-clock cycles and in us at 2.4 GHz... the clock s_memtime counts is 100 MHz on this part: see the output.
    RGBDR_LIB=build_variants/lib_trace.so RR_K1_FORM=2 python tools/k1_phase_trace.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import rgbd_recon_amd as rr
res = (512, 512, 512)
mk = dict(n_streams=4, width=640, height=480, lut_res=128, inv_res=128)
a = rr.scene.make_scene(**mk)
ext = a["bbox_max"] - a["bbox_min"]
hip = rr.ReconIntegrationHip(a, res=res, brick_size=[float(ext[k]) / res[k] * 8 for k in range(3)], limit=0.01, view=(1280, 720))
hip.set_stage_overlap(False)
mv, pr = rr.scene.default_view(1280, 720)
for _ in range(20):
    hip.clearOccupiedBricks(); hip.markBricks(); hip.updateOccupiedBricks(False); hip.integrate()
hip.sync()
L = rr.load_library()
n = 2048 * 16
buf = (C.c_uint64 * n)()
assert L.tsdf_debug_k1_trace(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 16).astype(np.float64)
ok = t[:, 0] > 0
t = t[ok]
names = ["head", "A own", "A barrier", "s0 box stored", "s0 B barrier", "s0 X+Y", "s0 Z", "s1 box stored", "s1 B barrier", "s1 X+Y", "s1 Z", "s2 box", "s2 B barrier", "to stores", "stores issued", "class barrier"]
print(f"{t.shape[0]} workgroups; tile span mean {np.mean(t[:, 15] - t[:, 0]):.0f} ticks, median {np.median(t[:, 15] - t[:, 0]):.0f}")
prev = t[:, 0]
for k in range(1, 16):
    cur = t[:, k]
    valid = cur >= prev
    d = (cur - prev)[valid]
    if d.size:
        print(f"{names[k]:16s} n={d.size:5d} mean {d.mean():8.1f} median {np.median(d):8.1f} p90 {np.percentile(d, 90):8.1f}")
    prev = np.where(valid, cur, prev)

h = (t[:, 11] >= t[:, 0]) & (t[:, 12] >= t[:, 11]) & (t[:, 12] <= t[:, 1])
if h.any():
    print(f"three back-to-back stamps at the loop top (n={int(h.sum())}): 0 -> 11: mean {np.mean(t[h, 11] - t[h, 0]):.0f} median {np.median(t[h, 11] - t[h, 0]):.0f}; 11 -> 12: mean {np.mean(t[h, 12] - t[h, 11]):.0f} median {np.median(t[h, 12] - t[h, 11]):.0f}; 12 -> phase A own part done: mean {np.mean(t[h, 1] - t[h, 12]):.0f}")
