"""Known-answer tests of the oracle's restatement of the inverse-LUT builder:
kinect::Frustum (framework/calibration/frustum.cpp) and CalibrationInverter::calculateInverseVolumes
(framework/calibration/calibration_inverter.cpp:57-115)."""
import numpy as np

from oracle import oracle as orc


def lattice(rx, ry, rz, origin=(0, 0, 0), step=(1, 1, 1)):
    """Forward volume whose texel (x, y, z) sits at origin + (x, y, z) * step: [rz][ry][rx][3]."""
    z, y, x = np.meshgrid(np.arange(rz), np.arange(ry), np.arange(rx), indexing="ij")
    return (np.stack([x, y, z], -1) * np.asarray(step) + np.asarray(origin)).astype(np.float32)


def test_frustum_of_a_box_lattice():
    planes, cam = orc.frustum(lattice(5, 3, 4))                   # corners (0,0,0)..(4,2,3)
    want = np.array([[0, 0, 1, 0], [0, 0, -1, 3], [1, 0, 0, 0], [-1, 0, 0, 4], [0, -1, 0, 2], [0, 1, 0, 0]], np.float32)   # near far left right top bottom
    np.testing.assert_allclose(planes, want, atol=1e-6)
    assert np.isnan(cam).all()                                    # parallel side edges never meet: 0/0 in closestPoint, as in the reference


def test_frustum_camera_position_of_a_pyramid():
    """Texel (x, y, z) on the ray through the image point (x-2, y-1) at depth z+1 from an eye at (1, 2, 3)."""
    rz, ry, rx = 4, 3, 5
    z, y, x = np.meshgrid(np.arange(rz), np.arange(ry), np.arange(rx), indexing="ij")
    d = (z + 1.0)
    vol = np.stack([1 + (x - 2) * 0.25 * d, 2 + (y - 1) * 0.25 * d, 3 + d], -1).astype(np.float32)
    planes, cam = orc.frustum(vol)
    np.testing.assert_allclose(cam, [1, 2, 3], atol=1e-5)
    inside = lambda p: all((pl[0] * p[0] + pl[1] * p[1]) + (pl[2] * p[2] + pl[3]) >= 0 for pl in planes)
    assert inside((1, 2, 5)) and not inside((1, 2, 3.5)) and not inside((1, 2, 7.5)) and not inside((3, 2, 5))


def test_inverse_of_an_identity_lattice():
    """Sensor lattice = world lattice with spacing 1: the inverse maps a world point to (its lattice coordinate + 0.5) / dims."""
    vol = lattice(6, 6, 6)
    inv = orc.invert_calibration(vol, (0, 0, 0), (5, 5, 5), (5, 5, 5))     # voxel centres at 0.5, 1.5, ... 4.5: cell centres
    assert (inv[..., 3] == 1).all()
    z, y, x = np.meshgrid(np.arange(5), np.arange(5), np.arange(5), indexing="ij")
    want = (np.stack([x, y, z], -1) + 0.5 + 0.5) / 6.0                     # 8 equidistant corners average to the cell centre
    np.testing.assert_allclose(inv[..., :3], want, atol=1e-6)


def test_outside_the_frustum_is_minus_one_and_coincident_samples_are_nan():
    vol = lattice(4, 4, 4)                                                 # frustum = box (0..3)^3
    inv = orc.invert_calibration(vol, (-2, -2, -2), (4, 4, 4), (3, 3, 3))  # voxel centres at -1, 1, 3
    assert (inv[0, 0, 0] == -1).all() and (inv[0, 1, 1] == -1).all()
    assert np.isnan(inv[1, 1, 1, :3]).all() and inv[1, 1, 1, 3] == 1       # centre (1,1,1) IS sample (1,1,1): weight 1/0
    assert np.isnan(inv[2, 2, 2, :3]).all()                                # (3,3,3) lies on the far planes: dot == 0 counts as inside


def test_inverse_distance_weights_by_hand():
    """2x2x2 samples; a point on the x axis between two corners: only the distance order and the 1/d weights matter."""
    vol = lattice(2, 2, 2)
    inv = orc.invert_calibration(vol, (0, 0, 0), (1, 1, 1), (4, 1, 1))     # x = 0.125, 0.375, 0.625, 0.875; y = z = 0.5
    f = np.float32
    for i, xq in enumerate([0.125, 0.375, 0.625, 0.875]):
        # four samples at x=0 (distance a), four at x=1 (distance b), all with |dy| = |dz| = 0.5
        a = np.sqrt(f(xq) * f(xq) + f(0.25) + f(0.25), dtype=f)
        b = np.sqrt(f(1 - xq) * f(1 - xq) + f(0.25) + f(0.25), dtype=f)
        wa, wb = f(1) / a, f(1) / b
        ix = (4 * wb) / (4 * wa + 4 * wb)
        np.testing.assert_allclose(inv[0, 0, i, 0], (ix + 0.5) / 2, rtol=1e-6)
        np.testing.assert_allclose(inv[0, 0, i, 1:3], (0.5 + 0.5) / 2, rtol=1e-6)


def test_scene_luts_are_consistent_with_the_builder(small_scene):
    """The synthetic scene's analytic cv_xyz_inv and the builder's IDW inverse agree to a fraction of a LUT cell."""
    sc = small_scene
    # the synthetic camera's (u, v, d) lattice is left-handed (image v runs down); kinect::Frustum's plane normals assume
    # the handedness of real calibration volumes, so flip v -- an equally valid lattice for the same camera
    xyz = np.ascontiguousarray(sc["cv_xyz"][0].reshape(32, 32, 32, 3)[:, ::-1])
    res = (12, 12, 12)
    inv = orc.invert_calibration(xyz, sc["bbox_min"], sc["bbox_max"], res)
    ok = inv[..., 3] > 0
    assert 0.05 < ok.mean() <= 1.0
    # forward lookup of the returned coordinate lands on the voxel centre
    ext = sc["bbox_max"] - sc["bbox_min"]
    z, y, x = np.meshgrid(*(np.arange(n) for n in res[::-1]), indexing="ij")
    centre = sc["bbox_min"] + (np.stack([x, y, z], -1) + 0.5) / np.array(res) * ext
    pts = inv[ok][:, :3]
    back = np.array([orc.tex3d(xyz, *p) for p in pts[::7]])
    err = np.linalg.norm(back - centre[ok][::7], axis=1)
    assert np.median(err) < 0.02 and err.max() < 0.1


def test_product_frustum_equals_the_oracle_bit_for_bit(rr, small_scene):
    """tsdf_frustum_from_volume is host code (no GPU): planes and camera position against the oracle on several volumes."""
    vols = [lattice(5, 3, 4), lattice(7, 7, 2, origin=(-1, 0.5, 2), step=(0.3, 0.2, 1.7))]
    vols += [np.ascontiguousarray(small_scene["cv_xyz"][i].reshape(32, 32, 32, 3)[:, ::-1]) for i in range(small_scene["n"])]
    for v in vols:
        (pa, ca), (pb, cb) = rr.frustum_from_volume(v), orc.frustum(v)
        assert (pa.view(np.uint32) == pb.view(np.uint32)).all()
        assert ((ca == cb) | (np.isnan(ca) & np.isnan(cb))).all()
    # and the camera positions are the ones the scene was rendered from (what tsdf_set_camera_position needs)
    for i in range(small_scene["n"]):
        _, cam = rr.frustum_from_volume(small_scene["cv_xyz"][i].reshape(32, 32, 32, 3))
        np.testing.assert_allclose(cam, small_scene["camera_positions"][i], atol=2e-5)
    assert rr.inverse_volume_resolution((-1, 0, -1), (1, 2.2, 1), 0.007) == (286, 315, 286)   # source/calib_inverter.cpp:60-63
