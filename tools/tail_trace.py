"""Where do the ~20 us of k_inpaint_tail go?  (tools/build_k1_variant.sh tailtrace "-DRR_TAIL_TRACE" k_inpaint)  s_memtime stamps of thread 0 at the level
boundaries of the last launch.
    RGBDR_LIB=build_variants/lib_tailtrace.so python tools/tail_trace.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import rgbd_recon_amd as rr
res = (128, 128, 128)
a = rr.scene.make_scene(n_streams=4, width=640, height=480, lut_res=128, inv_res=128)
hip = rr.ReconIntegrationHip(a, res=res, limit=0.01, view=(1280, 720))
hip.set_stage_overlap(False)
hip.setColorFilling(True)
mv, pr = rr.scene.default_view(1280, 720)
hip.integrate()
for _ in range(10):
    hip.drawF(mv, pr)
hip.sync()
L = rr.load_library()
buf = (C.c_uint64 * 32)()
assert L.tsdf_debug_tail_trace(buf, 32) == 0
t = np.frombuffer(buf, dtype=np.uint64).astype(np.float64)
print("k_inpaint_tail: shader-clock cycles (s_memtime) of thread 0 from kernel entry")
names = ["entry", "entry (second stamp)"] + [f"level +{k // 2 + 1} {'computed' if k % 2 == 0 else 'barrier'}" for k in range(12)]
for k in range(1, 14):
    if t[k] >= t[k - 1] and t[k] > 0:
        print(f"{names[k]:22s} +{t[k] - t[k - 1]:6.0f}  (at {t[k] - t[0]:6.0f})")
