"""-m gpu: the triangle-grid back-end (tsdf_draw_trigrid) against the oracle's in-order rasteriser.  Coverage and the depth
image are bit-exact (integer atomicMin on the z bit pattern); colours are sums of fp32 atomics whose order differs from GL's
draw order: 2e-5 absolute on values in [0, 1]."""
import numpy as np
import pytest

from oracle.oracle import OracleRecon

pytestmark = pytest.mark.gpu

KW = dict(res=(64, 64, 64), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.04, view=(320, 180))


@pytest.fixture(scope="module")
def scene(rr):
    return rr.scene.make_scene(n_streams=3, width=320, height=240, lut_res=32, inv_res=16)   # 8.8 mm pixel pitch at 2.5 m: inside min_length


def views(rr, w, h):
    pr = rr.scene.gl_flat(rr.scene.perspective(50.0, w / float(h), 0.1, 200.0))
    return [(rr.scene.gl_flat(rr.scene.look_at(e, (0.0, 1.1, 0.0))), pr) for e in [(0.0, 1.1, 3.0), (1.6, 1.4, 2.4)]]


@pytest.mark.parametrize("mode", [0, 1, 3])
def test_trigrid_matches_oracle(rr, scene, mode):
    hip, orc = rr.ReconIntegrationHip(scene, **KW), OracleRecon(scene, **KW)
    for o in (hip, orc):
        o.setShadeMode(mode)
    for mv, pr in views(rr, *KW["view"]):
        hip.drawTrigrid(mv, pr); orc.drawTrigrid(mv, pr)
        (fc, fd), (gc, gd) = hip.framebuffer(), orc.framebuffer()
        np.testing.assert_array_equal(fd, gd)                              # coverage + depth: exact
        assert np.abs(fc - gc).max() <= 2e-5
        assert (gd < 1).sum() > 2000 and gc[gd < 1][:, 3].min() == 1.0


def test_min_length_controls_the_mesh(rr, scene):
    hip, orc = rr.ReconIntegrationHip(scene, **KW), OracleRecon(scene, **KW)
    mv, pr = views(rr, *KW["view"])[0]
    counts = []
    for ml in (0.0125, 0.004):                                             # the second one rejects the diagonals of most cells
        for o in (hip, orc):
            o.setMinLength(ml); o.setShadeMode(3)
            o.drawTrigrid(mv, pr)
        (fc, fd), (gc, gd) = hip.framebuffer(), orc.framebuffer()
        np.testing.assert_array_equal(fd, gd)
        counts.append((gd < 1).sum())
    assert counts[1] < counts[0] * 0.7
    with pytest.raises(rr.TsdfError):
        hip.setMinLength(0.0)
