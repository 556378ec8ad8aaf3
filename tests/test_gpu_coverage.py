"""-m gpu: configurations and state changes around the hot path, each against the oracle on identical inputs:
stream counts 1 and 8 (the reference hard-codes 5), unequal LUT / colour resolutions, every setter the operator has,
and a sequence of different frames (the tile bookkeeping must reset tiles that stop being occupied)."""
import os

import numpy as np
import pytest

from helpers import assert_frames_identical, assert_same
from oracle.oracle import OracleRecon

pytestmark = pytest.mark.gpu

LIMIT = 0.04
KW = dict(res=(64, 64, 64), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=LIMIT, view=(160, 90))


def frame(o, mv, pr):
    o.clearOccupiedBricks(); o.markBricks(); r = o.updateOccupiedBricks()
    o.integrate()
    o.drawF(mv, pr)
    return r


def compare(hip, orc, limit=LIMIT):
    assert_same(hip.tsdf(), orc.tsdf(), "tsdf")
    return assert_frames_identical(hip, orc)


@pytest.mark.parametrize("n_streams", [1, 8])
def test_stream_counts(rr, n_streams):
    sc = rr.scene.make_scene(n_streams=n_streams, width=96, height=72, lut_res=16, inv_res=24, color_width=64, color_height=40)
    hip, orc = rr.ReconIntegrationHip(sc, **KW), OracleRecon(sc, **KW)
    mv, pr = rr.scene.default_view(*KW["view"])
    assert frame(hip, mv, pr) == frame(orc, mv, pr)
    assert compare(hip, orc) > 200


def test_lut_finer_than_volume_takes_the_global_path(rr):
    """inverse LUT 64^3 over a 32^3 TSDF: a tile's texel box (18^3) exceeds the LDS budget -> generic kernel, same answer."""
    sc = rr.scene.make_scene(n_streams=2, width=96, height=72, lut_res=16, inv_res=64)
    kw = dict(KW, res=(32, 32, 32), limit=0.08)
    hip, orc = rr.ReconIntegrationHip(sc, **kw), OracleRecon(sc, **kw)
    for o in (hip, orc):
        o.setUseBricks(False)
        o.integrate()
    assert_same(hip.tsdf(), orc.tsdf(), "tsdf (global-memory integrate)")


@pytest.mark.parametrize("res,inv_res", [((40, 40, 40), 28), ((40, 48, 56), 28), ((24, 24, 24), 20)])
def test_lds_box_near_its_capacity(rr, res, inv_res):
    """LUT nearly as fine as the volume: a tile's texel box is 6..7 texels per axis (up to 343 of the 384 the LDS path holds),
    boxes of different shapes per axis, partial tiles at the volume border -- still the LDS kernel, dense and culled."""
    sc = rr.scene.make_scene(n_streams=3, width=96, height=72, lut_res=16, inv_res=inv_res)
    kw = dict(KW, res=res, brick_size=[2.0 / 5, 2.2 / 5, 2.0 / 5], limit=0.08)
    hip, orc = rr.ReconIntegrationHip(sc, **kw), OracleRecon(sc, **kw)
    for use_bricks in (False, True):
        for o in (hip, orc):
            o.setUseBricks(use_bricks)
            o.clearOccupiedBricks(); o.markBricks(); o.updateOccupiedBricks(); o.integrate()
        a = hip.tsdf()
        assert_same(a, orc.tsdf(), f"tsdf (use_bricks={use_bricks})")
        assert (np.abs(a) < 0.08).sum() > 500


@pytest.mark.parametrize("form", ["3", "2", "1", "0"])
def test_all_three_integrate_kernels_give_the_same_volume(rr, small_scene, form, monkeypatch):
    """RR_K1_FORM caps the kernel choice when a context is created: 3 = no cap, here with the opt-in projection cache on (RR_PROJ_CACHE_MB:
    tiles integrated before are read back from the pool by k_integrate_cached), 2 = separable LDS passes stream by stream (the default
    form), 1 = direct 8-tap LDS form, 0 = every tap from global memory.  The choice is normally made from the LUT box size; each must be
    bit-identical."""
    monkeypatch.setenv("RR_K1_FORM", form)
    if form == "3":
        monkeypatch.setenv("RR_PROJ_CACHE_MB", "64")
    hip, orc = rr.ReconIntegrationHip(small_scene, **KW), OracleRecon(small_scene, **KW)
    for use_bricks in (True, False):
        for o in (hip, orc):
            o.setUseBricks(use_bricks)
            o.clearOccupiedBricks(); o.markBricks(); o.updateOccupiedBricks(); o.integrate()
        assert_same(hip.tsdf(), orc.tsdf(), f"tsdf (form {form}, use_bricks={use_bricks})")


def test_nan_and_out_of_range_lut_coordinates_sample_like_the_oracle(rr):
    """The inverse LUT may hold anything (the reference blends its -1 "invalid" marker into neighbouring texels): NaN, infinities
    and coordinates far outside [0, 1] must pick the same image texels as the oracle's GL clamp rule (the kernels clamp the
    integer-valued float index with v_med3_f32; the oracle with fmin / fmax and integer clamps) -- in all three integrate kernels
    and in the shading's LUT chain."""
    sc = dict(rr.scene.make_scene(n_streams=3, width=96, height=72, lut_res=16, inv_res=24))
    inv = sc["cv_xyz_inv"].copy()
    rng = np.random.default_rng(5)
    n = inv.shape[1]
    for i, vals in enumerate(([np.nan, np.nan, 0.5], [np.inf, -np.inf, 0.4], [7.5, -3.25, 0.45])):
        idx = rng.choice(n, n // 50, replace=False)
        inv[i, idx, :3] = np.array(vals, np.float32)
    sc["cv_xyz_inv"] = inv
    mv, pr = rr.scene.default_view(*KW["view"])
    for form in ("3", "2", "1", "0"):
        os.environ["RR_K1_FORM"] = form
        if form == "3":
            os.environ["RR_PROJ_CACHE_MB"] = "64"            # the opt-in projection cache: the second integrate() reads the pool
        try:
            hip, orc = rr.ReconIntegrationHip(sc, **KW), OracleRecon(sc, **KW)
        finally:
            del os.environ["RR_K1_FORM"]
            os.environ.pop("RR_PROJ_CACHE_MB", None)
        for o in (hip, orc):
            o.setUseBricks(False)
            o.integrate()
            o.integrate()
            o.drawF(mv, pr)
        assert_same(hip.tsdf(), orc.tsdf(), f"tsdf (form {form})")
        assert_frames_identical(hip, orc, f"frame (form {form})", min_hits=0)


def test_setters_between_frames(rr, small_scene):
    hip, orc = rr.ReconIntegrationHip(small_scene, **KW), OracleRecon(small_scene, **KW)
    mv, pr = rr.scene.default_view(*KW["view"])
    frame(hip, mv, pr); frame(orc, mv, pr)
    compare(hip, orc)
    # setTsdfLimit: the clear value and the raymarch step change (recon_integration.cpp:456-460)
    for o in (hip, orc):
        o.setTsdfLimit(0.06)
    frame(hip, mv, pr); frame(orc, mv, pr)
    compare(hip, orc, 0.06)
    # setMinVoxelsPerBrick + setUseBricks / setSpaceSkip / setColorFilling toggles
    for o in (hip, orc):
        o.setMinVoxelsPerBrick(40)
    assert frame(hip, mv, pr) == frame(orc, mv, pr)
    compare(hip, orc, 0.06)
    for o in (hip, orc):
        o.setSpaceSkip(False); o.setColorFilling(False)
    frame(hip, mv, pr); frame(orc, mv, pr)
    compare(hip, orc, 0.06)
    for o in (hip, orc):
        o.setUseBricks(False); o.setColorFilling(True)
    frame(hip, mv, pr); frame(orc, mv, pr)
    compare(hip, orc, 0.06)


def test_set_brick_size_and_resize(rr, small_scene):
    hip = rr.ReconIntegrationHip(small_scene, **KW)
    mv, pr = rr.scene.default_view(*KW["view"])
    frame(hip, mv, pr)
    hip.setBrickSize([2.0 / 16, 2.2 / 16, 2.0 / 16])                    # setBrickSize -> divideBox (:462-472)
    hip.resize(96, 64)                                                  # resize (:482-500)
    kw = dict(KW, brick_size=[2.0 / 16, 2.2 / 16, 2.0 / 16], view=(96, 64))
    orc = OracleRecon(small_scene, **kw)
    assert hip.res_bricks == orc.res_bricks == (16, 16, 16) and hip.num_lods == orc.num_lods
    mv, pr = rr.scene.default_view(96, 64)
    assert frame(hip, mv, pr) == frame(orc, mv, pr)
    assert compare(hip, orc) > 100


def test_frame_sequence_resets_tiles_that_empty_out(rr):
    """Three different frames through ONE context vs a fresh oracle per frame: tiles occupied in frame k but not in k+1
    must read -limit again (the reference clears the whole volume every frame, recon_integration.cpp:249-250)."""
    kw = dict(n_streams=3, width=128, height=96, lut_res=24, inv_res=32)
    frames = [rr.scene.make_scene(**kw), rr.scene.make_scene(sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2), **kw),
              rr.scene.make_scene(sphere_c=(-0.45, 1.5, 0.4), box_c=(0.2, 0.3, -0.6), **kw)]
    hip = rr.ReconIntegrationHip(frames[0], **KW)
    mv, pr = rr.scene.default_view(*KW["view"])
    occupied = []
    for sc in frames + [frames[0]]:
        hip.upload_frame(sc)
        orc = OracleRecon(sc, **KW)
        assert frame(hip, mv, pr) == frame(orc, mv, pr)
        compare(hip, orc)
        occupied.append(set(np.flatnonzero(hip.bricks()[1])))
    assert occupied[0] - occupied[1] and occupied[1] - occupied[2]      # bricks really do empty out between the frames
    assert occupied[0] == occupied[3]


def test_older_config_struct_is_still_accepted(rr, small_scene):
    """tsdf_config grew three times (sparse_pool_tiles, proj_cache_mib, the lane fields): a caller built against any earlier layout still works."""
    import ctypes as C
    L = rr.load_library()
    hip = rr.ReconIntegrationHip(small_scene, **KW)                  # reference context for the geometry
    cfg = rr.TsdfConfig()
    cfg.struct_size = rr.TsdfConfig.sparse_pool_tiles.offset         # the layout that ended at slab_recompute_halo
    cfg.bbox_min[:] = [float(x) for x in small_scene["bbox_min"]]
    cfg.bbox_max[:] = [float(x) for x in small_scene["bbox_max"]]
    cfg.voxel_size = 0.05
    cfg.res[:] = [64, 64, 64]
    cfg.brick_size[:] = [0.25, 0.275, 0.25]
    cfg.limit = 0.04
    cfg.num_streams = small_scene["n"]
    cfg.depth_w, cfg.depth_h = small_scene["width"], small_scene["height"]
    cfg.color_w, cfg.color_h = small_scene["color_width"], small_scene["color_height"]
    cfg.view_w, cfg.view_h = 160, 90
    cfg.sparse_pool_tiles = 12345                                    # lies beyond struct_size: must be ignored
    ctx = C.c_void_p()
    assert L.tsdf_create(C.byref(cfg), C.byref(ctx)) == 0
    need, cap = C.c_uint32(), C.c_uint32()
    assert L.tsdf_sparse_pool_stats(ctx, C.byref(need), C.byref(cap)) != 0       # a dense context
    assert L.tsdf_destroy(ctx) == 0
    cfg.struct_size = rr.TsdfConfig.proj_cache_mib.offset            # the layout that ended at sparse_pool_tiles: the field counts now
    cfg.sparse_pool_tiles = 0
    cfg.proj_cache_mib = 0xfffffff                                   # lies beyond struct_size: must be ignored
    assert L.tsdf_create(C.byref(cfg), C.byref(ctx)) == 0
    assert L.tsdf_destroy(ctx) == 0
    cfg.struct_size = rr.TsdfConfig.lane_flags.offset                # round 3's layout (ended at proj_cache_mib)
    cfg.proj_cache_mib = 0
    cfg.lane_flags = 0xffffffff                                      # lies beyond struct_size: must be ignored (all four lanes)
    assert L.tsdf_create(C.byref(cfg), C.byref(ctx)) == 0
    assert L.tsdf_destroy(ctx) == 0
    cfg.struct_size = 8
    assert L.tsdf_create(C.byref(cfg), C.byref(ctx)) != 0


def test_caller_timers_and_samples(rr, small_scene):
    hip = rr.ReconIntegrationHip(small_scene, **KW)
    mv, pr = rr.scene.default_view(*KW["view"])
    hip.timer_reserve("frame", 8)
    hip.set_timer_filter(["frame"])
    hip.enable_timers(True)
    for _ in range(5):
        hip.timer_begin("frame"); frame(hip, mv, pr); hip.timer_end("frame")
    hip.enable_timers(False)
    smp = hip.timer_samples("frame")
    assert smp.shape == (5,) and (smp > 0).all() and np.median(smp) < 5.0      # (a single sample can catch a one-off runtime stall of tens of ms)
    assert hip.timer_samples("frame").size == 0                      # reading resets
    assert hip.timer_stats("k_integrate_tiles")[0] == 0              # filtered out while the filter was set
    hip.set_timer_filter(None)


def test_integrate_without_update_uses_the_last_occupied_list(rr):
    """clearOccupiedBricks() zeroes the counters only; until the next updateOccupiedBricks() the reference keeps integrating and
    drawing with the host list of the last update (recon_integration.cpp:242-277,430-445)."""
    kw = dict(n_streams=3, width=128, height=96, lut_res=24, inv_res=32)
    a, b = rr.scene.make_scene(**kw), rr.scene.make_scene(sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2), **kw)
    hip, orc = rr.ReconIntegrationHip(a, **KW), OracleRecon(a, **KW)
    mv, pr = rr.scene.default_view(*KW["view"])
    for o in (hip, orc):
        frame(o, mv, pr)                                                 # frame 1: scene a, list of a
        o.upload_frame(b)
        o.clearOccupiedBricks(); o.markBricks()                          # counters of b, but no update: the list is still a's
        o.integrate(); o.drawF(mv, pr)
    compare(hip, orc)
    assert (hip.bricks()[0] != 0).sum() > 0
    for o in (hip, orc):
        o.updateOccupiedBricks(); o.integrate(); o.drawF(mv, pr)         # now b's list
    compare(hip, orc)


def test_contexts_release_their_device_memory(rr, small_scene):
    """tsdf_destroy and setVoxelSize must give the TSDF volume back (a 512^3 volume is 512 MiB: a client that re-creates
    contexts or changes the voxel size would run the device out of memory).  Free device memory after a create / integrate /
    setVoxelSize / destroy loop equals what it was after the first such cycle."""
    import torch

    def cycle():
        h = rr.ReconIntegrationHip(small_scene, res=(160, 160, 160), brick_size=0.2, limit=0.04, view=(64, 36), proj_cache_mib=32)
        h.clearOccupiedBricks(); h.markBricks(); h.updateOccupiedBricks(); h.integrate(); h.integrate()
        h.setVoxelSize(0.0125)                       # 160 x 176 x 160: a second, larger volume replaces the first
        h.clearOccupiedBricks(); h.markBricks(); h.updateOccupiedBricks(); h.integrate()
        h.close()

    cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(6):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (8 << 20), f"{(free0 - free1) >> 20} MiB of device memory lost over 6 context cycles"
