"""GL sampling semantics of SURVEY.md Appendix A, against hand-computed values (CPU only)."""
import numpy as np
import pytest

from oracle import oracle as orc

f32 = np.float32


def test_trilinear_at_texel_centres_returns_texels():
    t = np.random.default_rng(0).random((3, 4, 5, 2)).astype(f32)       # [z][y][x][c]
    for (z, y, x) in [(0, 0, 0), (2, 3, 4), (1, 2, 3)]:
        got = orc.tex3d(t, (x + 0.5) / 5, (y + 0.5) / 4, (z + 0.5) / 3)
        np.testing.assert_array_equal(got, t[z, y, x])


def test_trilinear_lerp_order_and_weights():
    t = np.zeros((2, 2, 2, 1), f32)
    t[0, 0, 1, 0] = 1.0; t[0, 1, 0, 0] = 2.0; t[1, 0, 0, 0] = 4.0
    # u*n - .5 = .25 -> weights (.25, .5, .75) on (x, y, z)
    u, v, w = (0.25 + 0.5) / 2, (0.5 + 0.5) / 2, (0.75 + 0.5) / 2
    ax, ay, az = f32(0.25), f32(0.5), f32(0.75)
    lerp = lambda a, b, k: f32(a + f32(f32(b - a) * k))
    c00, c10, c01, c11 = lerp(f32(0), f32(1), ax), lerp(f32(2), f32(0), ax), lerp(f32(4), f32(0), ax), f32(0)
    want = lerp(lerp(c00, c10, ay), lerp(c01, c11, ay), az)
    assert orc.tex3d(t, u, v, w)[0] == want


def test_clamp_to_edge_and_constant_exactness():
    t = np.full((4, 4, 4, 1), 0.01, f32)
    for p in [(-3.0, 0.5, 0.5), (0.5, 7.0, 0.5), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (0.37, 0.51, 0.93)]:
        assert orc.tex3d(t, *p)[0] == f32(0.01)                 # lerp(a, a, t) == a: constants survive filtering
    ramp = np.arange(4, dtype=f32).reshape(1, 1, 4, 1).repeat(4, 0).repeat(4, 1)
    assert orc.tex3d(ramp, -1.0, 0.5, 0.5)[0] == 0.0 and orc.tex3d(ramp, 2.0, 0.5, 0.5)[0] == 3.0


def test_bilinear_array_and_layer():
    t = np.zeros((2, 2, 2, 1), f32)                              # [layer][y][x][c]
    t[1] = np.array([[1, 2], [3, 4]], f32).reshape(2, 2, 1)
    assert orc.tex2d_linear(t, 1, 0.5, 0.5)[0] == f32(2.5)
    assert orc.tex2d_linear(t, 1, 0.25, 0.25)[0] == 1.0
    assert orc.tex2d_linear(t, 0, 0.5, 0.5)[0] == 0.0
    ones = np.ones((1, 3, 3, 1), f32)
    assert orc.tex2d_linear(ones, 0, 0.4321, 0.777)[0] == 1.0   # silhouette interior stays exactly 1 (tsdf_integration.vs:33)


@pytest.mark.parametrize("u,x", [(0.0, 0), (0.2499, 0), (0.25, 1), (0.999, 3), (1.0, 3), (-0.5, 0), (7.0, 3)])
def test_nearest_is_floor_clamped(u, x):
    t = np.arange(4, dtype=f32).reshape(1, 1, 4, 1)
    assert orc.tex2d_nearest(t, 0, u, 0.5) == f32(x)


def test_nan_coordinate_is_defined():
    t = np.ones((2, 2, 2, 1), f32)
    assert np.isnan(orc.tex3d(t, np.nan, 0.5, 0.5)[0])
