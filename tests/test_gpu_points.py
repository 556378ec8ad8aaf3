"""-m gpu: the point back-end (tsdf_draw_points: 64-bit atomicMin z-buffer + one shading pass) against the oracle's
in-order GL_LESS rasteriser.  Winner ids and window depths are integer/bit work: the depth image is bit-exact; colours go
through shade(): bit-exact too, except mode 1 (Phong calls pow(): 1e-6)."""
import numpy as np
import pytest

from oracle.oracle import OracleRecon

pytestmark = pytest.mark.gpu

KW = dict(res=(64, 64, 64), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.04, view=(160, 90))


def views(rr, w, h):
    pr = rr.scene.gl_flat(rr.scene.perspective(50.0, w / float(h), 0.1, 200.0))
    return [(rr.scene.gl_flat(rr.scene.look_at(e, (0.0, 1.1, 0.0))), pr) for e in [(0.0, 1.1, 3.0), (1.6, 1.4, 2.4), (0.3, 2.6, 0.4)]]


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_points_match_oracle(rr, small_scene, mode):
    hip, orc = rr.ReconIntegrationHip(small_scene, **KW), OracleRecon(small_scene, **KW)
    for o in (hip, orc):
        o.upload_normals(small_scene["normals"])
        o.setShadeMode(mode)
    covered = []
    for mv, pr in views(rr, *KW["view"]):
        hip.drawPoints(mv, pr); orc.drawPoints(mv, pr)
        (fc, fd), (gc, gd) = hip.framebuffer(), orc.framebuffer()
        np.testing.assert_array_equal(fd, gd)
        if mode == 1:
            assert np.abs(fc - gc).max() <= 1e-6                            # Phong: pow()
        else:
            np.testing.assert_array_equal(fc, gc)
        covered.append((gd < 1).sum())
    assert min(covered) > 500


def test_points_after_process_textures_use_its_normals(rr, small_scene):
    """Raw frame -> processTextures() -> drawPoints(): depth and normals are the ones the pre-processing produced."""
    hip, orc = rr.ReconIntegrationHip(small_scene, **KW), OracleRecon(small_scene, **KW)
    mv, pr = views(rr, *KW["view"])[0]
    for o in (hip, orc):
        o.upload_raw_frame(small_scene)
        o.clearOccupiedBricks(); o.processTextures()
        o.setShadeMode(1)
        o.drawPoints(mv, pr)
    (fc, fd), (gc, gd) = hip.framebuffer(), orc.framebuffer()
    np.testing.assert_array_equal(fd, gd)
    with np.errstate(invalid="ignore"):
        ok = (np.abs(fc - gc) <= 2e-6) | (np.isnan(fc) & np.isnan(gc))          # degenerate normals (0/0 in pre_normal.fs) shade to NaN on both sides
    assert ok.all() and (gd < 1).sum() > 500 and np.isnan(gc).mean() < 0.05


def test_big_sprites_close_to_the_eye(rr, small_scene):
    """An eye inside the scene: sprites of tens of pixels, heavy overdraw -- the atomic z-buffer stays exact."""
    hip, orc = rr.ReconIntegrationHip(small_scene, **KW), OracleRecon(small_scene, **KW)
    mv = rr.scene.gl_flat(rr.scene.look_at((0.0, 1.1, 0.9), (0.0, 1.1, 0.0)))
    pr = rr.scene.gl_flat(rr.scene.perspective(50.0, 160 / 90.0, 0.1, 200.0))
    for o in (hip, orc):
        o.upload_normals(small_scene["normals"]); o.setShadeMode(3)
        o.drawPoints(mv, pr)
    (fc, fd), (gc, gd) = hip.framebuffer(), orc.framebuffer()
    np.testing.assert_array_equal(fd, gd)
    np.testing.assert_array_equal(fc, gc)
    assert (gd < 1).mean() > 0.2
