"""Known-answer tests of the oracle's triangle-grid back-end (kinect::ReconTrigrid::draw, recon_trigrid.cpp:85-148 +
glsl/trigrid_accum.* / trigrid_normalize.fs)."""
import numpy as np

from helpers import tiny_scene
from oracle.oracle import OracleRecon

ORTHO = np.array([[2, 0, 0, -1], [0, 2, 0, -1], [0, 0, -1, 0], [0, 0, 0, 1]], np.float32).T.reshape(-1)   # window = (x, y) * view, z_w = 0.5 - z/2
IDENT = np.eye(4, dtype=np.float32).reshape(-1)
CAM = np.array([[228, 26, 28], [55, 126, 184]], np.float32) / np.float32(255)


def plane(depths, n=8, view=(16, 16), quality=None):
    k = len(depths)
    sc = tiny_scene([(0.5, 0.5, 0.5)] * k, depths, quality or [1.0] * k, [1.0] * k, w=n, h=n, lut=8)
    o = OracleRecon(sc, res=(8, 8, 8), brick_size=0.5, view=view)
    o.setMinLength(0.1)                                        # l = 0.1 * depth * 4: the 1/8 grid (diagonal 0.177) passes at depth 0.5
    o.setShadeMode(3)
    return o


def test_a_plane_is_covered_exactly_once():
    o = plane([0.5])
    o.drawTrigrid(IDENT, ORTHO)
    c, z = o.framebuffer()
    hit = z < 1
    # vertices at (i + .5)/8 -> window 1, 3, ..., 15: pixel centres 1.5 .. 14.5 are inside, each in exactly one triangle
    assert hit.sum() == 14 * 14 and hit[1:15, 1:15].all()
    np.testing.assert_allclose(z[hit], 0.25, rtol=1e-6)
    np.testing.assert_allclose(c[hit][:, :3], np.broadcast_to(CAM[0], (196, 3)), rtol=1e-6)
    assert (c[hit][:, 3] == 1).all() and (c[~hit] == 0).all()


def test_quality_weighted_blend_of_two_sensors_within_epsilon():
    o = plane([0.5, 0.52], quality=[1.0, 3.0])                 # 0.02 apart: both within epsilon = 0.075 of the front surface
    o.drawTrigrid(IDENT, ORTHO)
    c, z = o.framebuffer()
    np.testing.assert_allclose(z[8, 8], 0.5 - 0.52 / 2, rtol=1e-6)         # the z-buffer holds the nearer (larger d) plane
    np.testing.assert_allclose(c[8, 8, :3], (CAM[0] * 1 + CAM[1] * 3) / 4, rtol=1e-5)
    o = plane([0.3, 0.7], quality=[1.0, 3.0])                  # 0.4 apart: the farther surface is occluded
    o.drawTrigrid(IDENT, ORTHO)
    c, z = o.framebuffer()
    np.testing.assert_allclose(c[8, 8, :3], CAM[1], rtol=1e-5)


def test_long_edges_and_back_faces_are_dropped():
    o = plane([0.5])
    o.setMinLength(0.05)                                       # l = 0.1 < grid spacing 0.125: validSurface fails everywhere
    o.drawTrigrid(IDENT, ORTHO)
    assert (o.framebuffer()[1] == 1).all()
    o = plane([0.5])
    flip = np.diag([1, 1, -1, 1]).astype(np.float32)           # look from behind: the triangle normals face away
    o.drawTrigrid(flip.T.reshape(-1), ORTHO)
    assert (o.framebuffer()[1] == 1).all()


def test_swapped_loop_bounds_of_the_vertex_buffer():
    """recon_trigrid.cpp:53-54 iterates y < width, x < height: on a 8 x 4 image only columns x < 4 (+1) are drawn."""
    sc = tiny_scene([(0.5, 0.5, 0.5)], [0.5], [1.0], [1.0], w=8, h=4, lut=8)
    o = OracleRecon(sc, res=(8, 8, 8), brick_size=0.5, view=(16, 16))
    o.setMinLength(0.2); o.setShadeMode(3)
    o.drawTrigrid(IDENT, ORTHO)
    z = o.framebuffer()[1]
    cols = np.flatnonzero((z < 1).any(axis=0))
    # cells x = 0..3 span u from 0.5/8 to 4.5/8 -> window x 1 .. 9: pixel centres 1.5 .. 8.5
    assert cols.min() == 1 and cols.max() == 8
