"""Brick tables and mark_brick(): glsl/inc_bricks.glsl:22-58, recon_integration.cpp:360-406,430-445,462-472 (CPU only)."""
import numpy as np

from helpers import tiny_scene
from oracle.oracle import OracleRecon

f32 = np.float32


def make(res=None, voxel_size=0.01, brick=0.1, pos=(0.5, 0.5, 0.5), depth=0.5, bbox=None, w=2, h=2):
    sc = tiny_scene([pos], [depth], [1.0], [1.0], w=w, h=h)
    if bbox is not None:
        sc["bbox_min"], sc["bbox_max"] = np.float32(bbox[0]), np.float32(bbox[1])
    return OracleRecon(sc, res=res, voxel_size=voxel_size, brick_size=brick, limit=0.01, view=(8, 8)), sc


def test_reference_default_grid():
    """bbox (-1,0,-1)..(1,2.2,1), voxel 0.01, brick 0.1 (kinect_client.cpp:206-207, recon_integration.cpp:53).
    ceil() runs in fp32: 2.2f / 0.01f = 220.00002 -> 221 planes in y (recon_integration.cpp:342-344)."""
    o, _ = make(bbox=((-1, 0, -1), (1, 2.2, 1)))
    assert o.res == (int(np.ceil(f32(2) / f32(0.01))), int(np.ceil(f32(f32(2.2) - f32(0)) / f32(0.01))), 200)
    assert o.res == (200, 221, 200)
    assert o.res_bricks == (20, 22, 20) and o.numBricks() == 8800
    r = o.brick_ranges()
    # brick id = z*ry*rx + y*rx + x (inc_bricks.glsl:26-28): id 1 is the next brick in x
    assert (r[1][:3] == [10, 0, 0]).all() and (r[20][:3] == [0, 10, 0]).all() and (r[440][:3] == [0, 0, 10]).all()
    # containedVoxels() runs its bounds in fp32 (volume_sampler.cpp:53-58): (pos + size) / step lands just above an
    # integer for most bricks, so neighbouring voxel lists OVERLAP by one plane (brick 1 lists x in [10, 21)), and the
    # 221 y-planes spread over 22 bricks of 10.045 voxels.  Literal reference behaviour; harmless (same value written twice).
    assert (r[0] == [0, 0, 0, 10, 11, 10]).all() and (r[1] == [10, 0, 0, 21, 11, 10]).all()
    cover = np.zeros((200, 221, 200), np.int8)
    for lo0, lo1, lo2, hi0, hi1, hi2 in r:
        cover[lo2:hi2, lo1:hi1, lo0:hi0] += 1
    assert cover.min() == 1 and cover.max() == 8                     # no voxel is orphaned; corners are listed by 8 bricks


def test_brick_size_snaps_to_whole_voxels():               # :463  m_brick_size = voxel * round(size / voxel)
    o, _ = make(voxel_size=0.04, brick=0.21, bbox=((0, 0, 0), (1, 1, 1)))
    assert np.float32(o.brick_size[0]) == f32(0.04) * f32(5)
    # divideBox() accumulates `start += brick` in fp32 (:373-385): 5 * 0.19999999 < 1, so a sixth sliver brick appears
    assert o.res == (25, 25, 25) and o.res_bricks == (6, 6, 6)
    r = o.brick_ranges().reshape(6, 6, 6, 6)[0, 0]
    assert (r[:, 0] == [0, 5, 10, 14, 20, 25]).all() and (r[:, 3] == [5, 10, 15, 20, 25, 25]).all()   # 0.59999996/0.04 -> 14; sliver is empty


def test_partial_last_brick_and_full_cover():
    o, _ = make(res=(20, 22, 20), brick=[8 / 20, 8 / 22, 8 / 20], bbox=((0, 0, 0), (1, 1, 1)))
    assert o.res_bricks == (3, 3, 3)
    r = o.brick_ranges().reshape(3, 3, 3, 6)
    assert (r[0, 0, :, 0] == [0, 8, 16]).all() and (r[0, 0, :, 3] == [8, 16, 20]).all()
    cover = np.zeros((20, 22, 20), int)
    for lo0, lo1, lo2, hi0, hi1, hi2 in o.brick_ranges():
        cover[lo2:hi2, lo1:hi1, lo0:hi0] += 1
    assert (cover == 1).all()                                        # every voxel is drawn by exactly one brick


def test_mark_brick_counts_and_neighbour_rule():
    """One valid pixel at world (0.30, 0.55, 0.55), 4^3 bricks of 0.25 in a unit bbox: own brick (1,2,2) gets +1; the
    dominant axis of pos - centre(0.375,0.625,0.625) = (-.075,-.075,-.075) is a three-way tie, so all three components of
    `offset` are -1 (inc_bricks.glsl:45-50) and the diagonal neighbour (0,1,1) gets +1 because |dx| > 0.1 * brick (:52)."""
    o, sc = make(res=(16, 16, 16), brick=0.25, w=1, h=1)
    sc["cv_xyz"][0, :, :] = (0.30, 0.55, 0.55)
    o.clearOccupiedBricks(); o.markBricks()
    c = o.counters().reshape(4, 4, 4)
    assert c[2, 2, 1] == 1 and c[1, 1, 0] == 1 and c.sum() == 2


def test_mark_brick_x_axis_quirk_and_threshold():
    """d_abs.x is tested whatever the dominant axis (quirk 2): dominant y, tiny dx -> the neighbour add is 0."""
    o, sc = make(res=(16, 16, 16), brick=0.25, w=1, h=1)
    sc["cv_xyz"][0, :, :] = (0.380, 0.52, 0.62)                       # centre (.375,.625,.625): d = (.005, -.105, -.005)
    o.clearOccupiedBricks(); o.markBricks()
    c = o.counters().reshape(4, 4, 4)
    assert c[2, 2, 1] == 1 and c.sum() == 1
    # updateOccupiedBricks: >= min_voxels (recon_integration.cpp:436)
    o.setMinVoxelsPerBrick(1)
    assert o.updateOccupiedBricks() == f32(1) / f32(64) and list(o.occupied()) == [2 * 16 + 2 * 4 + 1]
    o.setMinVoxelsPerBrick(2)
    assert o.updateOccupiedBricks() == 0.0


def test_invalid_depth_marks_nothing():                   # pre_normal.fs:22-24  d <= 0 or d >= 1
    for d in (0.0, 1.0, -1.0):
        o, _ = make(res=(16, 16, 16), brick=0.25, depth=d)
        o.clearOccupiedBricks(); o.markBricks()
        assert o.counters().sum() == 0


def test_integrate_only_touches_occupied_brick_voxel_lists():
    """integrate() with bricks: cleared to -limit, then only the voxel lists of occupied bricks are drawn (:249-258)."""
    o, _ = make(res=(8, 8, 8), brick=0.5, pos=(0.5, 0.5, 0.504), bbox=((0, 0, 0), (1, 1, 1)))
    cnt = np.zeros(8, np.uint32)
    cnt[5] = 10                                                       # brick (x=1, y=0, z=1)
    o.set_counters(cnt)
    assert o.updateOccupiedBricks() == 0.125
    o.integrate()
    v = o.tsdf()
    band = f32(f32(0.504) - f32(0.5))
    assert (v[4:8, 0:4, 4:8] == band).all()
    v[4:8, 0:4, 4:8] = -0.01
    assert (v == f32(-0.01)).all()
