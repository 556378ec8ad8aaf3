"""K3/K4 on small atlases with planted holes (CPU only): ViewLod layout (view_lod.cpp:24-50), the squeeze mapping of
framebuffer_transfer.fs:13-17 composed with tsdf_inpaint.fs:34-89, and the pull of tsdf_colorfill.fs:30-55."""
import numpy as np

from helpers import tiny_scene
from oracle.oracle import OracleRecon

f32 = np.float32
W, H = 48, 32


def recon(w=W, h=H):
    sc = tiny_scene([(0.5, 0.5, 0.5)], [0.5], [1.0], [1.0])
    return OracleRecon(sc, res=(8, 8, 8), brick_size=[0.5] * 3, limit=0.05, view=(w, h))


def test_viewlod_layout_1280x720():
    o = recon(1280, 720)
    off, res = o.lod_tables()
    assert o.num_lods == 10 and len(off) == 10                       # 1 + floor(log2(720))
    assert (res[:4] == [[1280, 720], [640, 360], [320, 180], [160, 90]]).all() and tuple(res[9]) == (2, 1)
    assert tuple(off[0]) == (0, 0) and tuple(off[1]) == (1280, 360) and tuple(off[2]) == (1280, 180) and tuple(off[3]) == (1280, 90)
    assert o.atlas()[0].shape == (720, 1920, 4)


def test_fully_valid_image_levels_are_window_means():
    o = recon()
    rng = np.random.default_rng(3)
    rgba = np.concatenate([rng.random((H, W, 3), dtype=f32), np.ones((H, W, 1), f32)], -1)
    depth = np.full((H, W), 0.5, f32)                                 # equal depths: every sample passes `>= mean`
    o.set_view_images(rgba, depth)
    o.fillColors()
    ac, ad = o.atlas()
    off, res = o.lod_tables()
    np.testing.assert_array_equal(ac[:, :W], rgba)                     # level 0 untouched
    # level-1 pixel (i, j): pos = ivec2(vec2(floor(48 * i/24), floor(32 * j/16)) * (2/3, 1)); the 4x4 window reads squeezed
    # columns s = pos.x-1..pos.x+2, i.e. atlas columns floor(1.5 * (s + .5)), rows pos.y-1..pos.y+2
    i, j = 7, 5
    px, py = int(f32(int(f32(48) * (f32(i) / f32(24)))) * f32(2.0 / 3.0)), int(f32(32) * (f32(j) / f32(16)))
    cols = [int(1.5 * (s + 0.5)) for s in range(px - 1, px + 3)]
    acc = np.zeros(3, f32)                                             # samples[x + 4*y] are summed in index order (:75-85)
    for k in range(16):
        x, y = k % 4, k // 4
        acc = (acc + rgba[py - 1 + y, cols[x], :3] * f32(1)).astype(f32)
    want = (acc / f32(16)).astype(f32)
    got = ac[off[1][1] + j, off[1][0] + i]
    np.testing.assert_array_equal(got[:3], want)
    assert got[3] == 1.0 and ad[off[1][1] + j, off[1][0] + i] == f32(0.5)
    # colorfill: every pixel valid at level 0 -> output == input, depth == input
    fc, fd = o.framebuffer()
    np.testing.assert_array_equal(fc, rgba)
    np.testing.assert_array_equal(fd, depth)


def test_background_stays_untouched_and_holes_get_filled():
    o = recon()
    rgba = np.zeros((H, W, 4), f32); rgba[..., 1] = 1.0                # clear colour (0,1,0,0), view_lod.cpp:76
    depth = np.ones((H, W), f32)
    rgba[8:24, 12:36] = (0.2, 0.4, 0.6, 1.0); depth[8:24, 12:36] = 0.7 # a valid surface patch ...
    rgba[14:18, 20:26] = (9.0, 9.0, 9.0, -1.0)                         # ... with a hole: surface hit but no valid colour (alpha -1)
    o.set_view_images(rgba, depth)
    o.fillColors()
    fc, fd = o.framebuffer()
    bg = depth >= 1
    assert (fd[bg] == 1).all() and (fc[bg] == 0).all()                 # depth func LESS: background fragments fail (:313)
    np.testing.assert_array_equal(fd[~bg], depth[~bg])                 # depth always comes from level 0 (:54)
    valid = rgba[..., 3] > 0
    np.testing.assert_array_equal(fc[valid], rgba[valid])
    hole = rgba[..., 3] < 0
    assert np.allclose(fc[hole][:, :3], [0.2, 0.4, 0.6], atol=1e-5)    # pulled from the coarser levels, all of which average the same colour
    assert not np.isnan(fc).any()


def test_far_surface_wins_in_the_pyramid():
    """tsdf_inpaint.fs:75-85 keeps only samples at or behind the mean depth."""
    o = recon()
    rgba = np.zeros((H, W, 4), f32); rgba[..., 3] = 1.0
    depth = np.full((H, W), 0.9, f32)
    rgba[..., 0] = 1.0                                                 # far surface is red ...
    rgba[:, ::2] = (0.0, 0.0, 1.0, 1.0); depth[:, ::2] = 0.2           # ... every other column is a near, blue surface
    o.set_view_images(rgba, depth)
    o.fillColors()
    ac, ad = o.atlas()
    off, res = o.lod_tables()
    lvl1 = ac[off[1][1] + 2: off[1][1] + res[1][1] - 2, off[1][0] + 2: off[1][0] + res[1][0] - 2]
    assert (lvl1[..., 0] == 1).all() and (lvl1[..., 2] == 0).all()
    d1 = ad[off[1][1] + 2: off[1][1] + res[1][1] - 2, off[1][0] + 2: off[1][0] + res[1][0] - 2]
    assert np.abs(d1 - f32(0.9)).max() < 1e-6                          # sum of k far depths / k, fp32 rounding only
