"""How close to bit-identical is the draw path?  c2 at full size, HIP vs oracle: share of pixels whose values differ at all,
and the largest difference, per output.  (Not a test: a measurement for DESIGN.md section 2.)   python tools/exactness.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from importlib import import_module
rr = import_module("rgbd-recon_amd")
from oracle.oracle import OracleRecon

VIEW = (1280, 720)
res = (512, 512, 512) if len(sys.argv) < 2 else (int(sys.argv[1]),) * 3
scene = rr.scene.make_scene(n_streams=4, width=640, height=480, lut_res=128, inv_res=128)
ext = scene["bbox_max"] - scene["bbox_min"]
kw = dict(res=res, brick_size=[float(ext[a]) / res[a] * 8 for a in range(3)], limit=0.01, view=VIEW)
hip, orc = rr.ReconIntegrationHip(scene, **kw), OracleRecon(scene, **kw)
mv, pr = rr.scene.default_view(*VIEW)
for o in (hip, orc):
    o.setUseBricks(True); o.setSpaceSkip(True); o.setColorFilling(True)
    o.clearOccupiedBricks(); o.markBricks(); o.updateOccupiedBricks(); o.integrate(); o.draw(mv, pr)


def report(name, a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    with np.errstate(invalid="ignore"):
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    d[same] = 0
    print("%-28s differing %9d of %9d (%.4f %%)  max abs diff %.3g" % (name, (~same).sum(), same.size, 100.0 * (~same).mean(), np.nanmax(d)))


report("tsdf", hip.tsdf(), orc.tsdf())
(ha, hd, hn, hp), (oa, od, on, op) = hip.view_images(), orc.view_images()
report("depth peels", hp, op)
report("march sample counts", hn, on)
report("march depth", hd, od)
report("march colour rgba", ha, oa)
for o in (hip, orc):
    o.fillColors()                       # (the atlas is compared before this: the oracle ping-pongs two atlases literally, the HIP path keeps one)
(hc, hdd), (oc, odd) = hip.framebuffer(), orc.framebuffer()
report("framebuffer depth", hdd, odd)
report("framebuffer colour", hc, oc)
with np.errstate(invalid="ignore"):
    bad = np.argwhere(~((hc == oc) | (np.isnan(hc) & np.isnan(oc))).all(axis=2))
print("framebuffer pixels that differ:", len(bad), "first:", bad[:5].tolist())
for y, x in bad[:5]:
    print("  ", (int(y), int(x)), hc[y, x], oc[y, x], "level-0 alpha", ha[y, x, 3], oa[y, x, 3])

(hac, had), (oac, oad) = hip.atlas(), orc.atlas()
off, lres = orc.lod_tables()
for l in range(len(off)):
    x0, y0, rx, ry = int(off[l][0]), int(off[l][1]), int(lres[l][0]), int(lres[l][1])
    report("atlas level %d colour" % l, hac[y0:y0 + ry, x0:x0 + rx], oac[y0:y0 + ry, x0:x0 + rx])
    report("atlas level %d depth" % l, had[y0:y0 + ry, x0:x0 + rx], oad[y0:y0 + ry, x0:x0 + rx])
