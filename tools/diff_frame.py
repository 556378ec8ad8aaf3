"""Debug aid: one frame of a configuration through the HIP path and the oracle, and where the two differ.
   python tools/diff_frame.py <streams> <res> [dense]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import rgbd_recon_amd as rr
from oracle.oracle import OracleRecon

n, res = int(sys.argv[1]), int(sys.argv[2])
dense = len(sys.argv) > 3
VIEW = (1280, 720)
scene = rr.scene.make_scene(n_streams=n, width=640, height=480, lut_res=128, inv_res=128)
ext = scene["bbox_max"] - scene["bbox_min"]
kw = dict(res=(res,) * 3, brick_size=[float(ext[a]) / res * 8 for a in range(3)], limit=0.01, view=VIEW)
mv, pr = rr.scene.default_view(*VIEW)
hip, orc = rr.ReconIntegrationHip(scene, **kw), OracleRecon(scene, **kw)
for o in (hip, orc):
    if dense:
        o.setUseBricks(False); o.setSpaceSkip(False)
    o.clearOccupiedBricks(); o.markBricks(); o.updateOccupiedBricks(); o.integrate(); o.draw(mv, pr)
same = lambda a, b: (a == b) | (np.isnan(a) & np.isnan(b))
(ha, hd, hn, hp), (oa, od, on, op) = hip.view_images(), orc.view_images()
for name, a, b in (("peels", hp, op), ("nsamples", hn, on), ("depth", hd, od), ("colour", ha, oa)):
    m = ~same(a, b)
    if m.ndim == 3:
        m = m.any(-1)
    print(name, "differ:", int(m.sum()))
    ys, xs = np.nonzero(m)
    for y, x in list(zip(ys, xs))[:8]:
        print("   px", x, y, "hip", a[y, x], "orc", b[y, x], "| ns", hn[y, x] / 0.0027, on[y, x] / 0.0027, "depth", hd[y, x], od[y, x])
