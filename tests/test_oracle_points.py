"""Known-answer tests of the oracle's point back-end (kinect::ReconPoints::draw, recon_points.cpp:71-111 + glsl/points.*)."""
import numpy as np

from helpers import tiny_scene
from oracle.oracle import OracleRecon

# clip = (2x-1, 2y-1, -z, 1): window = (x, y) * view, window z = 0.5 - z/2
ORTHO = np.array([[2, 0, 0, -1], [0, 2, 0, -1], [0, 0, -1, 0], [0, 0, 0, 1]], np.float32).T.reshape(-1)   # column-major
IDENT = np.eye(4, dtype=np.float32).reshape(-1)
CAM0 = np.array([228, 26, 28], np.float32) / np.float32(255)


def recon(depths, view=(16, 16), w=1, h=1, **kw):
    sc = tiny_scene([(0.5, 0.5, 0.5)] * len(depths), depths, [1.0] * len(depths), [1.0] * len(depths), w=w, h=h, lut=2, **kw)
    o = OracleRecon(sc, res=(8, 8, 8), brick_size=0.5, view=view)
    return o, sc


def test_one_point_covers_the_pixels_whose_centres_lie_in_its_square():
    d = float(np.sqrt(np.float32(0.5)))                       # |(0.5, 0.5, d)| = 1 -> gl_PointSize = 4 / 1 in shade mode 3
    o, _ = recon([d])
    o.setShadeMode(3)
    o.drawPoints(IDENT, ORTHO)
    c, z = o.framebuffer()
    hit = z < 1
    assert hit.sum() == 16 and hit[6:10, 6:10].all()           # window centre (8, 8), half size 2: centres 6.5 .. 9.5
    np.testing.assert_allclose(z[hit], 0.5 - d / 2, rtol=1e-6)
    np.testing.assert_allclose(c[8, 8], [*CAM0, 1.0], rtol=1e-6)
    assert (c[~hit] == 0).all()


def test_nearest_point_wins_and_the_first_drawn_wins_ties():
    d0, d1 = 0.3, 0.7                                          # window z = 0.5 - d/2: layer 1 (0.15) is in front of layer 0 (0.35); the 2-texel LUT is linear on [0.25, 0.75]
    o, _ = recon([d0, d1])
    o.setShadeMode(3)
    o.drawPoints(IDENT, ORTHO)
    c, z = o.framebuffer()
    cam1 = np.array([55, 126, 184], np.float32) / np.float32(255)
    np.testing.assert_allclose(c[8, 8, :3], cam1, rtol=1e-6)
    front = np.abs(z - (0.5 - d1 / 2)) < 1e-6
    back = np.abs(z - (0.5 - d0 / 2)) < 1e-6
    assert back.sum() > 0 and front.sum() > 0 and front[8, 8]
    assert back.sum() + front.sum() == (z < 1).sum()           # the farther point is the nearer-to-the-eye-origin one: bigger sprite, a ring survives
    o, _ = recon([0.4, 0.4])
    o.setShadeMode(3)
    o.drawPoints(IDENT, ORTHO)
    np.testing.assert_allclose(o.framebuffer()[0][8, 8, :3], CAM0, rtol=1e-6)    # GL_LESS: equal depth does not replace


def test_culling_rules():
    o, _ = recon([0.0])                                        # depth <= 0: points.gs:36
    o.drawPoints(IDENT, ORTHO)
    assert (o.framebuffer()[1] == 1).all()
    sc = tiny_scene([(0.5, 0.5, 0.5)], [0.5], [1.0], [1.0], w=1, h=1, lut=2)
    sc["cv_uv"][:] = 0.995                                     # colour coordinate at the border of the RGB image: points.fs:38-41
    o = OracleRecon(sc, res=(8, 8, 8), brick_size=0.5, view=(16, 16))
    o.drawPoints(IDENT, ORTHO)
    assert (o.framebuffer()[1] == 1).all()
    sc = tiny_scene([(0.5, 0.5, 0.5)], [0.5], [1.0], [1.0], w=1, h=1, lut=2)
    sc["bbox_max"] = np.array([1.0, 1.0, 0.4], np.float32)     # world position outside the bounding box: inc_bbox_test.glsl
    o = OracleRecon(sc, res=(8, 8, 8), brick_size=0.5, view=(16, 16))
    o.drawPoints(IDENT, ORTHO)
    assert (o.framebuffer()[1] == 1).all()


def test_unshaded_colour_is_the_bilinear_colour_image():
    col = np.zeros((1, 2, 2, 3), np.uint8)
    col[0, :, :, 0] = [[0, 200], [0, 200]]
    o, sc = recon([0.5], w=2, h=2, color=col)
    o.setShadeMode(0)
    o.upload_normals(np.zeros((1, 2, 2, 3), np.float32))
    o.drawPoints(IDENT, ORTHO)
    c, z = o.framebuffer()
    # four points at (0.25|0.75, 0.25|0.75); cv_uv is the identity, so each samples the colour image at its own (u, v):
    # u = 0.25 is texel 0's centre -> 0, u = 0.75 texel 1's centre -> 200/255
    assert c[4, 4, 0] == 0 and abs(c[4, 12, 0] - 200 / 255) < 1e-6 and c[4, 4, 3] == 1
