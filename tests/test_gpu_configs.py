"""-m gpu: the BASELINE.json configurations at their own sizes against the oracle (the oracle needs a second or two per
frame on the GPU box's host cores at these sizes):
  configs[0]  128^3, 1 stream 640x480, integration                       -> TSDF bit-identical
  configs[1]  256^3, 4 streams, dense integrate + raymarch, 1280x720     -> TSDF and the rendered frame bit-identical
  configs[2]  512^3, 4 streams, brick cull + inpaint, 1280x720           -> bricks, TSDF, depth peels, sample counts, raymarch colour/depth,
                                                                            every pyramid level of the atlas and the hole-filled frame bit-identical
The draw path restates the oracle's fp32 operation order exactly (-ffp-contract=off on both sides), so the bar is equality, not a
tolerance: a single differing value fails.
(configs[3]/[4] are the multi-GPU shapes: slab == whole is covered at full size in test_gpu_sequences.py.)"""
import numpy as np
import pytest

from oracle.oracle import OracleRecon

pytestmark = pytest.mark.gpu

VIEW = (1280, 720)
LIMIT = 0.01


def same(a, b):
    return (a == b) | (np.isnan(a) & np.isnan(b))


def frame_identical(hip, orc):
    """raymarch target (before hole filling) and framebuffer: every value equal"""
    (ha, hd, hn, hp), (oa, od, on, op) = hip.view_images(), orc.view_images()
    assert same(hn, on).all() and same(hd, od).all() and same(ha, oa).all()
    (fc, fd), (gc, gd) = hip.framebuffer(), orc.framebuffer()
    assert same(fd, gd).all() and same(fc, gc).all()
    assert (fd < 1).sum() > 20000


def build(rr, n_streams, res, use_bricks, skip, fill):
    scene = rr.scene.make_scene(n_streams=n_streams, width=640, height=480, lut_res=128, inv_res=128)
    ext = scene["bbox_max"] - scene["bbox_min"]
    kw = dict(res=res, brick_size=[float(ext[a]) / res[a] * 8 for a in range(3)], limit=LIMIT, view=VIEW)
    hip, orc = rr.ReconIntegrationHip(scene, **kw), OracleRecon(scene, **kw)
    for o in (hip, orc):
        o.setUseBricks(use_bricks); o.setSpaceSkip(skip); o.setColorFilling(fill)
    return scene, hip, orc


def test_config0_128cubed_one_stream_integration(rr):
    _, hip, orc = build(rr, 1, (128, 128, 128), False, False, False)
    for o in (hip, orc):
        o.clearOccupiedBricks(); o.markBricks(); o.updateOccupiedBricks(); o.integrate()
    a, b = hip.tsdf(), orc.tsdf()
    assert same(a, b).all()
    assert (np.abs(a) < LIMIT).mean() > 1e-3                                 # a surface band exists


def test_config1_256cubed_dense_integrate_and_raymarch(rr):
    _, hip, orc = build(rr, 4, (256, 256, 256), False, False, False)
    mv, pr = rr.scene.default_view(*VIEW)
    for o in (hip, orc):
        o.clearOccupiedBricks(); o.markBricks(); o.updateOccupiedBricks(); o.integrate(); o.drawF(mv, pr)
    assert same(hip.tsdf(), orc.tsdf()).all()
    frame_identical(hip, orc)


def test_config2_512cubed_cull_and_inpaint(rr):
    _, hip, orc = build(rr, 4, (512, 512, 512), True, True, True)
    mv, pr = rr.scene.default_view(*VIEW)
    ratios = []
    for o in (hip, orc):
        o.clearOccupiedBricks(); o.markBricks(); ratios.append(o.updateOccupiedBricks()); o.integrate(); o.draw(mv, pr)
    assert ratios[0] == ratios[1] and 0.002 < ratios[0] < 0.05
    np.testing.assert_array_equal(hip.bricks()[0], orc.counters())
    assert same(hip.tsdf(), orc.tsdf()).all()
    (ha, hd, hn, hp), (oa, od, on, op) = hip.view_images(), orc.view_images()
    assert same(hp, op).all()                                                # depth peels (min-z of the occupied bricks' faces)
    assert same(hn, on).all()                                                # the two-pass march against the oracle's serial loop: sample counts
    assert same(hd, od).all() and same(ha, oa).all()                         # first zero crossing and its shading
    assert (hd < 1).sum() > 20000
    for o in (hip, orc):
        o.fillColors()
    (hac, had), (oac, oad) = hip.atlas(), orc.atlas()
    off, lres = orc.lod_tables()
    for l in range(len(off)):                                                # every level of the push-pull pyramid
        x0, y0, rx, ry = int(off[l][0]), int(off[l][1]), int(lres[l][0]), int(lres[l][1])
        assert same(hac[y0:y0 + ry, x0:x0 + rx], oac[y0:y0 + ry, x0:x0 + rx]).all(), l
        assert same(had[y0:y0 + ry, x0:x0 + rx], oad[y0:y0 + ry, x0:x0 + rx]).all(), l
    (fc, fd), (gc, gd) = hip.framebuffer(), orc.framebuffer()
    assert same(fd, gd).all() and same(fc, gc).all()
    assert ((fd < 1) & (hd >= 1)).sum() > 10                                # hole filling did fill pixels the march missed
