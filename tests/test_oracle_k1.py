"""Known-answer tests of the oracle's K1 (glsl/tsdf_integration.vs:23-59): every branch, by hand (CPU only)."""
import numpy as np

from helpers import tiny_scene
from oracle.oracle import OracleRecon

f32 = np.float32
L = f32(0.01)
KW = dict(res=(4, 4, 4), brick_size=[0.5, 0.5, 0.5], limit=float(L), view=(8, 8))


def run(pos_calib, depth, quality, sil):
    o = OracleRecon(tiny_scene(pos_calib, depth, quality, sil), **KW)
    o.setUseBricks(False)
    o.integrate()
    v = o.tsdf()
    assert (v == v.flat[0]).all() or np.isnan(v).all()
    return v.flat[0]


def test_outside_silhouette_carves():                    # :33-38  silhouette < 1 and nothing written yet
    assert run([(0.5, 0.5, 0.5)], [0.5], [1.0], [0.0]) == -L


def test_in_front_of_surface_is_minus_limit():           # :42-44  sdist <= -limit
    assert run([(0.5, 0.5, 0.30)], [0.5], [1.0], [1.0]) == -L


def test_behind_surface_keeps_plus_limit():              # :46-48  sdist >= limit -> untouched initial +limit
    assert run([(0.5, 0.5, 0.70)], [0.5], [1.0], [1.0]) == L


def test_band_single_stream_is_sdist():                  # :49-54  (limit*0 + w*sd) / (0 + w)
    z, d, w = f32(0.504), f32(0.5), f32(0.8)
    sd = f32(z - d)
    want = f32(f32(f32(L * f32(0)) + f32(w * sd)) / f32(f32(0) + w))
    assert run([(0.5, 0.5, float(z))], [float(d)], [float(w)], [1.0]) == want


def test_two_streams_weighted_average_in_order():
    z0, z1, d, w0, w1 = f32(0.504), f32(0.497), f32(0.5), f32(0.8), f32(0.3)
    sd0, sd1 = f32(z0 - d), f32(z1 - d)
    t = f32(f32(f32(L * f32(0)) + f32(w0 * sd0)) / f32(f32(0) + w0))
    W = f32(f32(0) + w0)
    want = f32(f32(f32(t * W) + f32(w1 * sd1)) / f32(W + w1))
    got = run([(0.5, 0.5, float(z0)), (0.5, 0.5, float(z1))], [float(d)] * 2, [float(w0), float(w1)], [1.0, 1.0])
    assert got == want


def test_stream_order_matters():                         # the loop over streams is not commutative
    a = run([(0.5, 0.5, 0.504), (0.5, 0.5, 0.5)], [0.5, 0.5], [1.0, 1.0], [1.0, 0.0])   # band first, then outside silhouette
    b = run([(0.5, 0.5, 0.5), (0.5, 0.5, 0.504)], [0.5, 0.5], [1.0, 1.0], [0.0, 1.0])   # carve first, then band
    sd = f32(f32(0.504) - f32(0.5))
    # a: stream 1 is outside its silhouette but tsd < limit already, so :33-38 falls through to the depth test (sdist 0, in band)
    assert a == f32(f32(f32(sd * f32(1)) + f32(f32(1) * f32(0))) / f32(f32(1) + f32(1)))
    # b: stream 0 carves (-limit, weight stays 0), stream 1 then overwrites with its own sdist
    assert b == f32(f32(f32(-L * f32(0)) + f32(f32(1) * sd)) / f32(f32(0) + f32(1))) and a != b
    c = run([(0.5, 0.5, 0.30), (0.5, 0.5, 0.70)], [0.5, 0.5], [1.0, 1.0], [1.0, 1.0])   # carve, then "behind": stays -limit
    d = run([(0.5, 0.5, 0.70), (0.5, 0.5, 0.30)], [0.5, 0.5], [1.0, 1.0], [1.0, 1.0])   # behind, then carve
    assert c == -L and d == -L
    e = run([(0.5, 0.5, 0.504), (0.5, 0.5, 0.30)], [0.5, 0.5], [1.0, 1.0], [1.0, 1.0])  # band then carve: overwritten
    g = run([(0.5, 0.5, 0.30), (0.5, 0.5, 0.504)], [0.5, 0.5], [1.0, 1.0], [1.0, 1.0])  # carve then band: (-L*0 + sd)/1
    assert e == -L and g == f32(f32(0.504) - f32(0.5)) and e != g


def test_zero_quality_gives_nan_like_the_shader():       # :52  0/0 (SURVEY §7 NaN hazard)
    assert np.isnan(run([(0.5, 0.5, 0.5)], [0.5], [0.0], [1.0]))


def test_store_index_and_layout():
    """voxel (x,y,z) is stored at z*ry*rx + y*rx + x (tsdf_integration.vs:57, volume_sampler.cpp:39-45)."""
    sc = tiny_scene([(0.5, 0.5, 0.5)], [0.5], [1.0], [1.0], lut=4)
    inv = sc["cv_xyz_inv"].reshape(1, 4, 4, 4, 4)
    inv[0, ..., 2] = 0.30                                 # everything carved ...
    inv[0, 3, 1, 2, 2] = 0.504                            # ... except LUT texel x=2,y=1,z=3
    o = OracleRecon(sc, res=(4, 4, 4), brick_size=[0.5] * 3, limit=float(L), view=(8, 8))
    o.setUseBricks(False)
    o.integrate()
    v = o.tsdf()
    assert v[3, 1, 2] == f32(f32(0.504) - f32(0.5)) and (np.delete(v.ravel(), 3 * 16 + 1 * 4 + 2) == -L).all()
