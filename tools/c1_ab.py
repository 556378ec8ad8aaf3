"""A/B of the dense c1 integrate across variant libraries: kernel time (HIP events).  python tools/c1_ab.py default lib1.so ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
sys.path.insert(0, %r)
import torch  # noqa
import rgbd_recon_amd as rr
scene = rr.scene.make_scene(n_streams=4, width=640, height=480, lut_res=128, inv_res=128)
ext = scene["bbox_max"] - scene["bbox_min"]
hip = rr.ReconIntegrationHip(scene, res=(256,) * 3, brick_size=[float(ext[a]) / 256 * 8 for a in range(3)], limit=0.01, view=(1280, 720))
hip.setUseBricks(False); hip.setSpaceSkip(False); hip.setColorFilling(False)
for _ in range(30): hip.integrate()
hip.set_timer_filter(["k_integrate_tiles"]); hip.enable_timers(True)
for _ in range(100): hip.integrate()
hip.sync(); n, ms = hip.timer_stats("k_integrate_tiles")
print(json.dumps({"k_integrate_us": ms / n * 1e3}))
''' % ROOT
for lib in sys.argv[1:]:
    env = dict(os.environ)
    if lib != "default":
        env["RGBDR_LIB"] = os.path.abspath(lib)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    print(lib, line[-1] if line else p.stderr[-800:], flush=True)
