"""-m gpu: the BASELINE.json configurations at their own sizes against the oracle (the oracle needs a second or two per
frame on the GPU box's host cores at these sizes):
  configs[0]  128^3, 1 stream 640x480, integration                       -> TSDF bit-identical
  configs[1]  256^3, 4 streams, dense integrate + raymarch, 1280x720     -> TSDF bit-identical, frame within the path's tolerances
  configs[2]  512^3, 4 streams, brick cull + inpaint, 1280x720           -> bricks exact, TSDF bit-identical, frame within tolerances
(configs[3]/[4] are the multi-GPU shapes: slab == whole is covered at full size in test_gpu_sequences.py.)"""
import numpy as np
import pytest

from oracle.oracle import OracleRecon

pytestmark = pytest.mark.gpu

VIEW = (1280, 720)
LIMIT = 0.01


def same(a, b):
    return (a == b) | (np.isnan(a) & np.isnan(b))


def frame_close(hip, orc):
    (fc, fd), (gc, gd) = hip.framebuffer(), orc.framebuffer()
    assert ((fd < 1) != (gd < 1)).mean() <= 2e-3
    both = (fd < 1) & (gd < 1)
    assert both.sum() > 20000
    assert (np.abs(fd[both] - gd[both]) > 1e-4).mean() <= 2e-3
    with np.errstate(invalid="ignore"):
        assert (np.abs(fc[both] - gc[both]) > 2e-3).mean() <= 1e-2
    return both.sum()


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
    frame_close(hip, orc)


def test_config2_512cubed_cull_and_inpaint(rr):
    _, hip, orc = build(rr, 4, (512, 512, 512), True, True, True)
    mv, pr = rr.scene.default_view(*VIEW)
    ratios = []
    for o in (hip, orc):
        o.clearOccupiedBricks(); o.markBricks(); ratios.append(o.updateOccupiedBricks()); o.integrate(); o.drawF(mv, pr)
    assert ratios[0] == ratios[1] and 0.002 < ratios[0] < 0.05
    np.testing.assert_array_equal(hip.bricks()[0], orc.counters())
    assert same(hip.tsdf(), orc.tsdf()).all()
    frame_close(hip, orc)
    # the ray bookkeeping of the two-pass march against the oracle's serial loop: sample counts per pixel
    ns_h, ns_o = hip.view_images()[2], orc.view_images()[2]
    assert (ns_h != ns_o).mean() <= 2e-3
