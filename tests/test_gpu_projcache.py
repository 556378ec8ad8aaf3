"""-m gpu: the opt-in projection cache of the integrate kernel (round 3; ProjCache in csrc/tsdf_common.hpp, k_integrate_cached;
tsdf_config::proj_cache_mib -- off by default: measured slower than the LUT kernel on MI355X).

texture(cv_xyz_inv[i], voxel centre).xyz (tsdf_integration.vs:31) depends on the calibration and the voxel grid only.  The first
integrate() of a tile computes its x/y-filtered LUT planes with the LUT kernel and writes them through to a pool slot; every later
integrate() of that tile reads them back (k_integrate_cached).  Both must give the oracle's volume bit for bit, in every state the
cache can be in:

  * first frame (all tiles filled by the LUT kernel), second frame (all tiles cached), culled and dense, tile-aligned or not;
  * a moving scene: cached, fresh and stale tiles in one launch;
  * bricks that do not coincide with storage tiles (the cached kernel's per-voxel "is it in an occupied brick's list" test);
  * odd inputs: NaN / infinite / out-of-range LUT texels, NaN depths, silhouettes that are neither 0 nor 1, zero quality;
  * a pool smaller than the scene (part of the tiles stays on the LUT path for good), the cache switched off;
  * re-calibration (cached coordinates of the old LUT must not survive), setVoxelSize (another grid), setTsdfLimit;
  * Z-slab contexts with recomputed halo layers.
"""
import numpy as np
import pytest

from helpers import assert_same
from oracle.oracle import OracleRecon

pytestmark = pytest.mark.gpu

MOVED = dict(sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2))


def bricks_and_integrate(o):
    o.clearOccupiedBricks(); o.markBricks(); o.updateOccupiedBricks(); o.integrate()


def scene_pair(rr, **kw):
    return rr.scene.make_scene(**kw), rr.scene.make_scene(**kw, **MOVED)


@pytest.mark.parametrize("res,brick_div", [((64, 64, 64), 8), ((40, 56, 72), 5), ((96, 96, 96), 12)])
@pytest.mark.parametrize("use_bricks", [True, False])
def test_first_frame_fills_second_frame_reads_the_cache(rr, small_scene, res, brick_div, use_bricks):
    kw = dict(res=res, brick_size=[2.0 / brick_div, 2.2 / brick_div, 2.0 / brick_div], limit=0.05, view=(64, 36))
    hip, orc = rr.ReconIntegrationHip(small_scene, proj_cache_mib=256, **kw), OracleRecon(small_scene, **kw)
    for o in (hip, orc):
        o.setUseBricks(use_bricks)
    bricks_and_integrate(orc)
    want = orc.tsdf()
    for f in range(3):
        bricks_and_integrate(hip)
        st = hip.integrate_stats()
        assert st["items"] > 0 and st["slots"] > 0, st
        if f == 0:
            assert st["cached"] == 0 and st["lut_items"] == st["items"], st      # nothing cached yet: the LUT kernel fills every slot
        else:
            assert st["cached"] == st["items"] and st["lut_items"] == 0, st      # steady state
            assert 0 < st["full_pairs"] <= st["items"] * small_scene["n"], st
        assert_same(hip.tsdf(), want, f"frame {f} (res {res}, use_bricks {use_bricks})")
    assert (np.abs(want) < kw["limit"]).sum() > 300


def test_moving_scene_mixes_cached_fresh_and_stale_tiles(rr):
    mk = dict(n_streams=4, width=160, height=120, lut_res=32, inv_res=32)
    a, b = scene_pair(rr, **mk)
    kw = dict(res=(96, 96, 96), brick_size=[2.0 / 12, 2.2 / 12, 2.0 / 12], limit=0.04, view=(64, 36))
    hip, orc = rr.ReconIntegrationHip(a, proj_cache_mib=256, **kw), OracleRecon(a, **kw)
    seen_mixed = False
    for f, sc in enumerate([a, b, a, b, b, a]):
        for o in (hip, orc):
            o.upload_frame(sc)
            bricks_and_integrate(o)
        st = hip.integrate_stats()
        seen_mixed |= 0 < st["cached"] < st["items"]
        assert_same(hip.tsdf(), orc.tsdf(), f"frame {f}")
    assert seen_mixed, "no launch had both cached and uncached tiles: the test does not cover the mixed case"
    assert hip.integrate_stats()["lut_items"] == 0


def test_bricks_that_do_not_coincide_with_tiles(rr, small_scene):
    """10-voxel bricks on 8-voxel tiles (the reference's default geometry): the voxel lists of neighbouring bricks overlap, a tile
    reaches into up to 27 bricks, and the cached kernel has to ask per voxel whether an occupied brick lists it."""
    kw = dict(res=(80, 88, 80), brick_size=0.25, limit=0.05, view=(64, 36))
    hip, orc = rr.ReconIntegrationHip(small_scene, proj_cache_mib=256, **kw), OracleRecon(small_scene, **kw)
    for mv_ in (10, 40, 10, 150):
        for o in (hip, orc):
            o.setMinVoxelsPerBrick(mv_)
            bricks_and_integrate(o)
        assert_same(hip.tsdf(), orc.tsdf(), f"min voxels {mv_}")
    assert hip.integrate_stats()["cached"] > 0


def test_odd_inputs_through_the_cache(rr):
    base = rr.scene.make_scene(n_streams=4, width=160, height=120, lut_res=32, inv_res=32)
    rng = np.random.default_rng(11)
    odd = dict(base)
    inv = odd["cv_xyz_inv"].copy()
    n = inv.shape[1]
    for i, vals in enumerate(([np.nan, 0.3, 0.5], [np.inf, -np.inf, 0.4], [7.5, -3.25, 0.45], [0.5, 0.5, -2.0e5])):
        inv[i, rng.choice(n, n // 60, replace=False), :3] = np.array(vals, np.float32)
    odd["cv_xyz_inv"] = inv
    d = odd["depth"].copy(); q = odd["quality"].copy(); s = odd["silhouette"].copy()
    d[0, 10:14, 20:40, 0] = np.nan; d[1, 50:60, 70:90, 0] = -0.3; d[2, 80:82, :, 0] = np.inf
    s[1, 30:50, 30:50] = 0.5; s[3, 60:70, 10:30] = np.nan; s[2, 5:9, 100:140] = 2.0
    q[0] = 0.0
    odd["depth"], odd["quality"], odd["silhouette"] = d, q, s
    for use_bricks in (False, True):
        kw = dict(res=(64, 64, 64), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.04, view=(64, 36))
        hip, orc = rr.ReconIntegrationHip(odd, proj_cache_mib=256, **kw), OracleRecon(odd, **kw)
        for f in range(2):
            for o in (hip, orc):
                o.setUseBricks(use_bricks)
                bricks_and_integrate(o)
            assert_same(hip.tsdf(), orc.tsdf(), f"odd inputs, use_bricks {use_bricks}, frame {f}")
        assert hip.integrate_stats()["cached"] > 0
        assert np.isnan(hip.tsdf()).sum() > 0


def test_small_pool_and_no_pool(rr, small_scene):
    kw = dict(res=(96, 96, 96), brick_size=[2.0 / 12, 2.2 / 12, 2.0 / 12], limit=0.04, view=(64, 36))
    orc = OracleRecon(small_scene, **kw)
    bricks_and_integrate(orc)
    want = orc.tsdf()
    # 1 MiB holds a few dozen slots of 4 streams x dz planes x 768 B: most tiles never get one
    small = rr.ReconIntegrationHip(small_scene, proj_cache_mib=1, **kw)
    for f in range(3):
        bricks_and_integrate(small)
        assert_same(small.tsdf(), want, f"small pool, frame {f}")
    st = small.integrate_stats()
    assert 10 < st["slots"] < 200 and st["slots_used"] == st["slots"] and st["cached"] == st["slots"] and st["lut_items"] == st["items"] - st["slots"], st
    off = rr.ReconIntegrationHip(small_scene, **kw)
    for f in range(2):
        bricks_and_integrate(off)
        assert_same(off.tsdf(), want, f"no pool, frame {f}")
    assert off.integrate_stats()["slots"] == 0


def test_recalibration_voxel_size_and_limit_changes(rr):
    mk = dict(n_streams=3, width=128, height=96, lut_res=24, inv_res=32)
    a = rr.scene.make_scene(**mk)
    kw = dict(res=(64, 64, 64), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.04, view=(64, 36))
    hip, orc = rr.ReconIntegrationHip(a, proj_cache_mib=256, **kw), OracleRecon(a, **kw)
    for f in range(2):
        for o in (hip, orc):
            bricks_and_integrate(o)
    assert_same(hip.tsdf(), orc.tsdf(), "before")
    assert hip.integrate_stats()["cached"] > 0
    # another calibration: the inverse LUTs of the streams rotated by one (stream i now projects like stream i + 1 did)
    c = dict(a)
    for k in ("cv_xyz_inv", "cv_xyz", "cv_uv"):
        c[k] = np.ascontiguousarray(np.roll(a[k], 1, axis=0))
    hip.set_calibration(c)
    orc2 = OracleRecon(c, **kw)
    for f in range(2):
        bricks_and_integrate(hip)
        if f == 0:
            assert hip.integrate_stats()["cached"] == 0, "cached coordinates survived a re-calibration"
    orc2.upload_frame(a)
    bricks_and_integrate(orc2)
    hip.upload_frame(a)
    bricks_and_integrate(hip)
    assert_same(hip.tsdf(), orc2.tsdf(), "after re-calibration")
    # setTsdfLimit: the cached coordinates do not depend on it
    for o in (hip, orc2):
        o.setTsdfLimit(0.06)
        bricks_and_integrate(o)
    assert_same(hip.tsdf(), orc2.tsdf(), "after setTsdfLimit")
    assert hip.integrate_stats()["cached"] > 0
    # setVoxelSize: a new grid, a new cache
    for o in (hip, orc2):
        o.setVoxelSize(0.025)
        bricks_and_integrate(o)
        bricks_and_integrate(o)
    assert_same(hip.tsdf(), orc2.tsdf(), "after setVoxelSize")
    assert hip.integrate_stats()["cached"] > 0


def test_slab_contexts_with_recomputed_halo(rr, small_scene):
    kw = dict(res=(64, 64, 96), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 12], limit=0.04, view=(64, 36))
    orc = OracleRecon(small_scene, **kw)
    bricks_and_integrate(orc)
    want = orc.tsdf()
    for z0, z1 in ((0, 32), (32, 64), (64, 96)):
        s = rr.ReconIntegrationHip(small_scene, slab=(z0, z1), recompute_halo=True, proj_cache_mib=256, **kw)
        for f in range(2):
            bricks_and_integrate(s)
        assert s.integrate_stats()["cached"] > 0 or not (np.abs(want[z0:z1]) < 0.04).any()
        got = s.tsdf()
        assert_same(got[z0:z1], want[z0:z1], f"slab {z0}:{z1}")
