"""-m gpu: round 3.

  * tsdf_upload_frame_dev: a frame whose arrays are already in device memory goes through ONE re-layout launch (packed texel, depth
    plane, per-cell ranges, RGBA8 colour) -- the result must be what the host upload leaves, for image sizes that do not fill the
    8x8 cells / 4-pixel colour quads, and the frame computed from it must be the oracle's.
"""
import numpy as np
import pytest

from helpers import assert_same
from oracle.oracle import OracleRecon

pytestmark = pytest.mark.gpu


def frame(o, mv, pr):
    o.clearOccupiedBricks(); o.markBricks(); o.updateOccupiedBricks(); o.integrate(); o.drawF(mv, pr)


@pytest.mark.parametrize("w,h,cw,ch,n", [(160, 120, 160, 120, 4), (131, 97, 67, 45, 3), (64, 48, 33, 21, 1)])
def test_device_resident_frame_upload_equals_host_upload(rr, w, h, cw, ch, n):
    import torch
    mk = dict(n_streams=n, width=w, height=h, lut_res=24, inv_res=32, color_width=cw, color_height=ch)
    a = rr.scene.make_scene(**mk)
    b = rr.scene.make_scene(**mk, sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2))
    kw = dict(res=(64, 64, 64), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.04, view=(96, 54))
    mv, pr = rr.scene.default_view(*kw["view"])
    dev, host, orc = rr.ReconIntegrationHip(a, **kw), rr.ReconIntegrationHip(a, **kw), OracleRecon(a, **kw)
    for use_bricks in (True, False):                      # the dense launch reads the per-cell ranges (uniform-pair shortcut)
        for o in (dev, host, orc):
            o.setUseBricks(use_bricks)
        for k, sc in enumerate((b, a, b)):
            t = [torch.from_numpy(np.ascontiguousarray(sc[key])).cuda() for key in ("depth", "quality", "silhouette", "color")]
            torch.cuda.synchronize()
            dev.upload_frame_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr() if k != 1 else 0, complete=(k == 2))   # colour is optional
            if k == 1:
                dev.sync(); dev.upload_frame(sc)                       # (frame 1: colour through the host path, same result)
            host.upload_frame(sc); orc.upload_frame(sc)
            for o in (dev, host, orc):
                frame(o, mv, pr)
            dev.sync()
            assert_same(dev.tsdf(), host.tsdf(), f"tsdf dev vs host upload (frame {k}, use_bricks {use_bricks})")
            assert_same(dev.tsdf(), orc.tsdf(), f"tsdf vs oracle (frame {k})")
            (dc, dd), (hc, hd), (oc, od) = dev.framebuffer(), host.framebuffer(), orc.framebuffer()
            assert_same(dd, hd, "framebuffer depth"); assert_same(dc, hc, "framebuffer colour")
            assert_same(dd, od, "framebuffer depth vs oracle"); assert_same(dc, oc, "framebuffer colour vs oracle")
    assert (dd < 1).sum() > 50


def test_stage_overlap_of_hole_filling_changes_nothing(rr):
    """fillColors() of draw f runs on a second stream beside the next frame's upload re-layout, brick passes and integrate (stage overlap,
    the default).  Frames queued back to back without any read in between -- the case in which the streams really overlap -- must
    leave what a context without overlap leaves, and both what the oracle leaves; toggling between frames included."""
    import torch
    mk = dict(n_streams=3, width=160, height=120, lut_res=24, inv_res=32)
    scs = [rr.scene.make_scene(**mk), rr.scene.make_scene(**mk, sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2)), rr.scene.make_scene(**mk, sphere_c=(-0.3, 1.3, 0.2))]
    kw = dict(res=(96, 96, 96), brick_size=[2.0 / 12, 2.2 / 12, 2.0 / 12], limit=0.03, view=(320, 180))
    pr = rr.scene.gl_flat(rr.scene.perspective(50.0, 16.0 / 9.0, 0.1, 200.0))
    eyes = [(0.0, 1.1, 3.0), (1.6, 1.4, 2.4), (-2.2, 0.6, 1.2)]
    mvs = [rr.scene.gl_flat(rr.scene.look_at(e, (0.0, 1.1, 0.0))) for e in eyes]
    over, plain, orc = rr.ReconIntegrationHip(scs[0], **kw), rr.ReconIntegrationHip(scs[0], **kw), OracleRecon(scs[0], **kw)
    plain.set_stage_overlap(False)
    raw = [[torch.from_numpy(np.ascontiguousarray(sc[k])).cuda() for k in ("depth", "quality", "silhouette", "color")] for sc in scs]
    torch.cuda.synchronize()
    order = [0, 1, 2, 1, 0, 2, 2, 1]
    for rounds in range(2):
        for n, k in enumerate(order):                     # eight frames queued without a single host read
            for o in (over, plain):
                o.upload_frame_dev(*[t.data_ptr() for t in raw[k]], complete=True)
                frame_nosync(o, mvs[n % 3], pr)
        if rounds == 0:
            over.set_stage_overlap(False); over.set_stage_overlap(True)       # (synchronises, drops and re-arms the second stream's state)
    orc.upload_frame(scs[order[-1]])
    frame(orc, mvs[(len(order) - 1) % 3], pr)
    (oc, od), (pc, pd), (rc_, rd) = over.framebuffer(), plain.framebuffer(), orc.framebuffer()
    assert_same(od, pd, "framebuffer depth, overlap vs one stream"); assert_same(oc, pc, "framebuffer colour, overlap vs one stream")
    assert_same(od, rd, "framebuffer depth vs oracle"); assert_same(oc, rc_, "framebuffer colour vs oracle")
    (oa, odd) = over.atlas(); (pa, pdd) = plain.atlas()
    assert_same(oa, pa, "pyramid colour"); assert_same(odd, pdd, "pyramid depth")
    assert (od < 1).sum() > 200
    # a read right behind a draw waits for the hole filling in flight
    for k in (1, 0):
        for o in (over, orc):
            o.upload_frame(scs[k])
            frame(o, mvs[k], pr)
        (oc, od), (rc_, rd) = over.framebuffer(), orc.framebuffer()
        assert_same(od, rd, "depth right behind the draw"); assert_same(oc, rc_, "colour right behind the draw")


def frame_nosync(o, mv, pr):
    o.clearOccupiedBricks(); o.markBricks(); o.updateOccupiedBricks(False); o.integrate(); o.drawF(mv, pr)


def compare_images(hip, orc, what):
    (hc, hd), (oc, od) = hip.framebuffer(), orc.framebuffer()
    assert_same(hd, od, f"framebuffer depth, {what}"); assert_same(hc, oc, f"framebuffer colour, {what}")
    (ha, hdd), (oa, odd) = hip.atlas(), orc.atlas()
    assert_same(hdd, odd, f"pyramid depth, {what}"); assert_same(ha, oa, f"pyramid colour, {what}")
    return int((od < 1).sum())


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("view", [(320, 180), (203, 117), (1280, 720)])
def test_hole_filling_by_dirty_tiles_equals_the_full_pass(rr, overlap, view):
    """fillColors() keeps to the screen tiles the last three draws touched once three culled draws in a row have left nothing else in
    the pyramid and the framebuffer (k_inpaint.hip, "dirty tiles").  Every level of the pyramid and the framebuffer must stay the
    oracle's -- which fills every pixel of every level, every frame -- while the content moves under a moving camera, and through
    everything that breaks the three-draw history: a draw without hole filling, a point draw, a colour mask, an uncleared colour buffer,
    a shifted viewport, an uploaded image, dense draws, a resize."""
    mk = dict(n_streams=3, width=160, height=120, lut_res=24, inv_res=32)
    scs = [rr.scene.make_scene(**mk), rr.scene.make_scene(**mk, sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2)), rr.scene.make_scene(**mk, sphere_c=(-0.3, 1.3, 0.2))]
    kw = dict(res=(96, 96, 96), brick_size=[2.0 / 12, 2.2 / 12, 2.0 / 12], limit=0.03, view=view)
    pr = rr.scene.gl_flat(rr.scene.perspective(50.0, view[0] / view[1], 0.1, 200.0))
    eyes = [(0.0, 1.1, 3.0), (1.6, 1.4, 2.4), (-2.2, 0.6, 1.2), (0.2, 3.4, 0.4), (0.0, 1.1, 6.5)]
    mvs = [rr.scene.gl_flat(rr.scene.look_at(e, (0.0, 1.1, 0.0))) for e in eyes]
    hip, orc = rr.ReconIntegrationHip(scs[0], **kw), OracleRecon(scs[0], **kw)
    hip.set_stage_overlap(overlap)
    n = [0]

    def both(fn):
        for o in (hip, orc):
            fn(o)

    def run(frames, what, check_every=True):
        seen = 0
        for _ in range(frames):
            k = n[0]; n[0] += 1
            both(lambda o: (o.upload_frame(scs[k % 3]), frame(o, mvs[(k * 2) % 5], pr)))
            if check_every or _ == frames - 1:
                seen = compare_images(hip, orc, f"{what}, frame {k}")
        return seen

    assert run(6, "steady state") > 50
    fills, by_tiles = hip.fill_stats()
    assert fills == 6 and by_tiles == 4, (fills, by_tiles)               # the first two draws of a history fill every tile
    run(5, "no read between the frames", check_every=False)
    assert hip.fill_stats() == (11, 9)
    # a draw without hole filling writes the framebuffer itself
    both(lambda o: o.setColorFilling(False)); run(2, "filling off"); both(lambda o: o.setColorFilling(True))
    run(4, "filling on again")
    # colour masks / an uncleared colour buffer (the anaglyph pair): full passes, and a full pass after them
    both(lambda o: (o.setColorMaskMode(1), o.setFramebufferClear(False))); run(2, "red mask, colour kept")
    both(lambda o: o.setColorMaskMode(2)); run(2, "cyan mask")
    both(lambda o: (o.setColorMaskMode(0), o.setFramebufferClear(True))); run(4, "mask off")
    # a shifted viewport marches every pixel
    both(lambda o: o.setViewportOffset(0.37, -0.21)); run(2, "shifted")
    both(lambda o: o.setViewportOffset(0.0, 0.0)); run(4, "shift off")
    # an image uploaded over the march target, then filled
    rgba, depth = orc.view_images()[:2]
    rng = np.random.default_rng(5)
    depth = np.where(rng.random(depth.shape) < 0.02, np.float32(0.5), np.float32(1.0)).astype(np.float32)
    rgba = np.where(depth[..., None] < 1, rng.random(rgba.shape, dtype=np.float32), np.float32([0, 1, 0, 0])).astype(np.float32)
    both(lambda o: (o.set_view_images(rgba, depth), o.fillColors()))
    compare_images(hip, orc, "uploaded image")
    run(4, "after the uploaded image")
    # dense draws (no brick culling, no tiles), back to culled
    both(lambda o: o.setUseBricks(False)); run(2, "dense")
    both(lambda o: o.setUseBricks(True)); run(4, "culled again")
    both(lambda o: o.setSpaceSkip(False)); run(2, "no space skipping")
    both(lambda o: o.setSpaceSkip(True)); run(3, "space skipping again")
    # another picture size
    small = (view[0] // 2 + 3, view[1] // 2 + 5)
    hip.resize(*small)
    orc2 = OracleRecon(scs[0], **dict(kw, view=small))
    pr2 = rr.scene.gl_flat(rr.scene.perspective(50.0, small[0] / small[1], 0.1, 200.0))
    for k in range(5):
        for o in (hip, orc2):
            o.upload_frame(scs[k % 3]); frame(o, mvs[(k * 2) % 5], pr2)
        compare_images(hip, orc2, f"resized, frame {k}")
    fills, by_tiles = hip.fill_stats()
    assert by_tiles >= 20 and fills - by_tiles >= 20, (fills, by_tiles)


def test_hole_filling_by_dirty_tiles_under_a_point_draw(rr, small_scene):
    kw = dict(res=(64, 64, 64), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.04, view=(160, 90))
    mv, pr = rr.scene.default_view(*kw["view"])
    hip, orc = rr.ReconIntegrationHip(small_scene, **kw), OracleRecon(small_scene, **kw)
    for k in range(4):
        for o in (hip, orc):
            frame(o, mv, pr)
    compare_images(hip, orc, "before")
    assert hip.fill_stats() == (4, 2)
    for o in (hip, orc):
        o.upload_normals(small_scene["normals"])
        o.drawPoints(mv, pr)                                              # writes the framebuffer outside any tile mask
    (hc, hd), (oc, od) = hip.framebuffer(), orc.framebuffer()
    assert_same(hd, od, "points depth"); assert_same(hc, oc, "points colour")
    for k in range(3):
        for o in (hip, orc):
            frame(o, mv, pr)
        compare_images(hip, orc, f"after the point draw, frame {k}")
    assert hip.fill_stats() == (7, 4)                                     # the first fill after it went through every tile


@pytest.mark.parametrize("overlap", [True, False])
def test_dirty_tile_masks_reach_as_far_as_the_hole_filling_spreads(rr, overlap):
    """Small objects that move in a large, still picture: the tile masks hug the bricks, while every level of the hole filling spreads
    the content three more pixels of its parent level (10 x 5 window, tsdf_inpaint.fs) -- into tiles whose own parents are clean --
    and levels from the third on read past their last row.  A pass that forgot either (checked with such builds: they fail here)
    keeps what an earlier frame spread there."""
    mk = dict(n_streams=3, width=160, height=120, lut_res=24, inv_res=32)
    scs = [rr.scene.make_scene(**mk), rr.scene.make_scene(**mk, sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2)), rr.scene.make_scene(**mk, sphere_c=(-0.3, 1.3, 0.2))]
    seen = 0
    for view, eye, target in [((640, 360), (0.0, 1.1, 9.0), (0.0, 1.1, 0.0)), ((640, 360), (2.0, 5.0, 7.0), (1.5, -1.0, 0.0)),
                              ((491, 277), (-6.0, 1.5, 5.0), (0.4, 2.4, 0.0)), ((640, 360), (0.0, 0.9, 8.0), (0.0, 3.2, 0.0)),
                              ((333, 555), (5.0, 2.0, 5.0), (0.0, -0.6, 0.0)),
                              # the objects cut by the first / the last row of the picture, in its left half: what levels >= 3 pick up past their last row
                              ((640, 360), (3.0, 4.3, 8.0), (3.0, 4.3, 0.0)), ((640, 360), (3.0, -2.1, 8.0), (3.0, -2.1, 0.0)),
                              ((512, 512), (1.5, 3.2, 5.0), (1.5, 3.2, 0.0)), ((512, 512), (1.5, -1.0, 5.0), (1.5, -1.0, 0.0))]:
        kw = dict(res=(96, 96, 96), brick_size=[2.0 / 24, 2.2 / 24, 2.0 / 24], limit=0.012, view=view)
        pr = rr.scene.gl_flat(rr.scene.perspective(50.0, view[0] / view[1], 0.1, 200.0))
        mv = rr.scene.gl_flat(rr.scene.look_at(eye, target))
        hip, orc = rr.ReconIntegrationHip(scs[0], **kw), OracleRecon(scs[0], **kw)
        hip.set_stage_overlap(overlap)
        order = [0, 1, 1, 0, 0, 2, 1, 1, 2, 0]
        for k, i in enumerate(order):
            for o in (hip, orc):
                o.upload_frame(scs[i]); frame(o, mv, pr)
            seen += compare_images(hip, orc, f"view {view} from {eye}, frame {k}")
        assert hip.fill_stats() == (len(order), len(order) - 2)
    assert seen > 9000


def make_pair(rr, scene, deep, monkeypatch, **kw):
    """a context with the fourth lane (integrate() of frame f + 1 beside the draw of frame f, two volume sets) and one without"""
    return rr.ReconIntegrationHip(scene, lane_flags=0 if deep else rr.LANES_NO_INTEGRATE_LANE, **kw)   # tsdf_config::lane_flags (round 4; RR_DEEP in round 3)


def test_two_volume_sets_alternate_without_a_trace(rr, monkeypatch):
    """integrate() of frame f + 1 runs on a lane of its own beside the draw of frame f, into the volume set the draw is NOT reading;
    the two sets (volume, tile classes, incremental tile lists) alternate per integrate() and each sees every other frame.  Volume,
    active tiles, framebuffer and pyramid must be the oracle's after every frame of a moving scene -- whichever set holds it -- and
    through everything that touches a set from outside: setTsdfLimit, an uploaded volume, dense passes, a new brick grid, min-voxel
    changes, integrate() twice in a row, a draw without integrate(), the occupied ratio read back between update and integrate."""
    mk = dict(n_streams=3, width=160, height=120, lut_res=24, inv_res=32)
    scs = [rr.scene.make_scene(**mk), rr.scene.make_scene(**mk, sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2)), rr.scene.make_scene(**mk, sphere_c=(-0.3, 1.3, 0.2))]
    kw = dict(res=(96, 96, 96), brick_size=[2.0 / 12, 2.2 / 12, 2.0 / 12], limit=0.03, view=(320, 180))
    pr = rr.scene.gl_flat(rr.scene.perspective(50.0, 16.0 / 9.0, 0.1, 200.0))
    mvs = [rr.scene.gl_flat(rr.scene.look_at(e, (0.0, 1.1, 0.0))) for e in [(0.0, 1.1, 3.0), (1.6, 1.4, 2.4), (-2.2, 0.6, 1.2)]]
    hip, orc = make_pair(rr, scs[0], True, monkeypatch, **kw), OracleRecon(scs[0], **kw)
    n = [0]

    def both(fn):
        for o in (hip, orc):
            fn(o)

    def check(what):
        assert_same(hip.tsdf(), orc.tsdf(), f"volume, {what}")
        compare_images(hip, orc, what)

    def run(frames, what, ratio=False):
        got = []
        for _ in range(frames):
            k = n[0]; n[0] += 1
            for o in (hip, orc):
                o.upload_frame(scs[[0, 1, 1, 2, 0, 2][k % 6]])
                o.clearOccupiedBricks(); o.markBricks()
                r = o.updateOccupiedBricks(ratio) if o is hip else o.updateOccupiedBricks()
                if ratio:
                    got.append(r)
                o.integrate(); o.drawF(mvs[k % 3], pr)
            if ratio:
                assert got[-2] == got[-1] and got[-1] > 0
            check(f"{what}, frame {k}")

    run(7, "moving scene")
    run(3, "occupied ratio read back", ratio=True)
    hip.set_stage_overlap(False); run(4, "everything on one stream (after the lanes have used the rotating buffers)")
    hip.set_stage_overlap(True); run(5, "lanes again")
    both(lambda o: o.setTsdfLimit(0.045)); run(4, "another limit")
    both(lambda o: o.setTsdfLimit(0.03)); run(3, "the first limit again")
    vol = orc.tsdf().copy(); vol[20:40, 30:50, 10:30] = np.float32(0.01)
    both(lambda o: (o.set_tsdf(vol), o.drawF(mvs[0], pr))); check("an uploaded volume, drawn")
    run(4, "after the uploaded volume")
    both(lambda o: o.setUseBricks(False)); run(3, "dense")
    both(lambda o: o.setUseBricks(True)); run(4, "culled again")
    both(lambda o: o.setMinVoxelsPerBrick(60)); run(3, "more voxels per brick")
    both(lambda o: o.setMinVoxelsPerBrick(10)); run(3, "fewer again")
    both(lambda o: (o.integrate(), o.integrate(), o.drawF(mvs[1], pr))); check("integrate twice, draw")
    both(lambda o: o.drawF(mvs[2], pr)); check("a draw without integrate")
    both(lambda o: (o.integrate(), o.drawF(mvs[0], pr), o.drawF(mvs[1], pr))); check("two draws of one volume")
    run(4, "and on")
    hip.setBrickSize([2.0 / 8, 2.2 / 8, 2.0 / 8])
    orc2 = OracleRecon(scs[0], **dict(kw, brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8]))
    for k in range(5):
        for o in (hip, orc2):
            o.upload_frame(scs[k % 3]); frame(o, mvs[k % 3], pr)
        assert_same(hip.tsdf(), orc2.tsdf(), f"volume, new brick grid, frame {k}")
        compare_images(hip, orc2, f"new brick grid, frame {k}")


@pytest.mark.parametrize("use_bricks", [True, False])
def test_fourth_lane_back_to_back_frames(rr, monkeypatch, use_bricks):
    """Frames queued back to back with no host read in between -- integrate(f + 1) really runs beside march / shade(f) -- leave what a
    context without the lane leaves, every frame's framebuffer included (kept on the device by a copy queued on the context's stream)."""
    import torch
    mk = dict(n_streams=3, width=160, height=120, lut_res=24, inv_res=32)
    scs = [rr.scene.make_scene(**mk), rr.scene.make_scene(**mk, sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2)), rr.scene.make_scene(**mk, sphere_c=(-0.3, 1.3, 0.2))]
    kw = dict(res=(128, 128, 128), brick_size=[2.0 / 16, 2.2 / 16, 2.0 / 16], limit=0.03, view=(640, 360))
    pr = rr.scene.gl_flat(rr.scene.perspective(50.0, 16.0 / 9.0, 0.1, 200.0))
    mvs = [rr.scene.gl_flat(rr.scene.look_at(e, (0.0, 1.1, 0.0))) for e in [(0.0, 1.1, 3.0), (1.6, 1.4, 2.4), (-2.2, 0.6, 1.2)]]
    deep, flat = make_pair(rr, scs[0], True, monkeypatch, **kw), make_pair(rr, scs[0], False, monkeypatch, **kw)
    orc = OracleRecon(scs[0], **kw)
    raw = [[torch.from_numpy(np.ascontiguousarray(sc[k])).cuda() for k in ("depth", "quality", "silhouette", "color")] for sc in scs]
    torch.cuda.synchronize()
    order = [0, 1, 1, 2, 0, 2, 1, 0, 0, 2, 1, 2] * 3
    for o in (deep, flat, orc):
        o.setUseBricks(use_bricks); o.setSpaceSkip(use_bricks)
    for rounds in range(2):
        for n, k in enumerate(order):
            for o in (deep, flat):
                o.upload_frame_dev(*[t.data_ptr() for t in raw[k]], complete=True)
                frame_nosync(o, mvs[n % 3], pr)
        last = (order[-1], mvs[(len(order) - 1) % 3])
        (dc, dd), (fc, fd) = deep.framebuffer(), flat.framebuffer()
        assert_same(dd, fd, f"framebuffer depth, round {rounds}"); assert_same(dc, fc, f"framebuffer colour, round {rounds}")
        assert_same(deep.tsdf(), flat.tsdf(), f"volume, round {rounds}")
    orc.upload_frame(scs[last[0]]); frame(orc, last[1], pr)
    assert_same(deep.tsdf(), orc.tsdf(), "volume vs oracle")
    (dc, dd), (oc, od) = deep.framebuffer(), orc.framebuffer()
    assert_same(dd, od, "framebuffer depth vs oracle"); assert_same(dc, oc, "framebuffer colour vs oracle")
    assert (od < 1).sum() > 500


def test_frame_in_one_call_equals_the_separate_calls(rr):
    """tsdf_frame_dev (upload_frame_dev + clear / mark / update + integrate + drawF in one call into the library) against the calls
    one by one and against the oracle; tsdf_timer_spans puts the stages of those frames on one clock."""
    import torch
    mk = dict(n_streams=3, width=160, height=120, lut_res=24, inv_res=32)
    scs = [rr.scene.make_scene(**mk), rr.scene.make_scene(**mk, sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2))]
    kw = dict(res=(96, 96, 96), brick_size=[2.0 / 12, 2.2 / 12, 2.0 / 12], limit=0.03, view=(320, 180))
    mv, pr = rr.scene.default_view(*kw["view"])
    one, sep, orc = rr.ReconIntegrationHip(scs[0], **kw), rr.ReconIntegrationHip(scs[0], **kw), OracleRecon(scs[0], **kw)
    raw = [[torch.from_numpy(np.ascontiguousarray(sc[k])).cuda() for k in ("depth", "quality", "silhouette", "color")] for sc in scs]
    torch.cuda.synchronize()
    one.enable_timers(True)
    for k in (0, 1, 1, 0, 1):
        ptrs = [t.data_ptr() for t in raw[k]]
        one.frame_dev(mv, pr, ptrs)
        sep.upload_frame_dev(*ptrs, complete=True); frame_nosync(sep, mv, pr)
    for a_, b_ in zip(one.bricks(), sep.bricks()):
        assert_same(a_, b_, "brick counters / flags, one call vs separate calls")
    # the two ways mixed on ONE context
    for n, k in enumerate((1, 0, 0, 1, 0, 1, 1)):
        ptrs = [t.data_ptr() for t in raw[k]]
        if n in (1, 2, 5):
            one.upload_frame_dev(*ptrs, complete=True); frame_nosync(one, mv, pr)
        else:
            one.frame_dev(mv, pr, ptrs)
        sep.upload_frame_dev(*ptrs, complete=True); frame_nosync(sep, mv, pr)
        if n in (2, 4, 6):
            for a_, b_ in zip(one.bricks(), sep.bricks()):
                assert_same(a_, b_, f"brick counters / flags, mixed calls, frame {n}")
    one.frame_dev(mv, pr)                                  # no new frame: the bricks, the volume and the picture again from the frame in place
    frame_nosync(sep, mv, pr)
    orc.upload_frame(scs[1]); frame(orc, mv, pr)
    assert_same(one.tsdf(), sep.tsdf(), "volume, one call vs separate calls"); assert_same(one.tsdf(), orc.tsdf(), "volume vs oracle")
    compare_images(one, orc, "one call")
    (b0, e0), (b1, e1), (b2, e2) = one.timer_spans("0repack", "0repack"), one.timer_spans("2integrate", "0repack"), one.timer_spans("holefill", "0repack")
    assert len(b0) == 12 and len(b1) == 13 and len(b2) == 13 and b0[0] == 0.0
    assert (e0 > b0).all() and (e1 > b1).all() and (b1[:12] >= e0).all() and (b2 >= e1).all()      # re-layout -> integrate -> hole filling of a frame, in that order
    assert (np.diff(b1) > 0).all()


def test_fill_lane_issued_by_the_helper_thread(rr, monkeypatch):
    """The hole filling's launches are issued by a helper thread of the context (not while timers are on: then the calling thread issues
    them itself).  Frames queued back to back, switching between the two ways every few frames, reads and re-draws in between, against a
    context without the thread (RR_FILL_THREAD=0) and the oracle; creating and destroying contexts with a live helper."""
    import torch
    mk = dict(n_streams=3, width=160, height=120, lut_res=24, inv_res=32)
    scs = [rr.scene.make_scene(**mk), rr.scene.make_scene(**mk, sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2)), rr.scene.make_scene(**mk, sphere_c=(-0.3, 1.3, 0.2))]
    kw = dict(res=(96, 96, 96), brick_size=[2.0 / 12, 2.2 / 12, 2.0 / 12], limit=0.03, view=(320, 180))
    pr = rr.scene.gl_flat(rr.scene.perspective(50.0, 16.0 / 9.0, 0.1, 200.0))
    mvs = [rr.scene.gl_flat(rr.scene.look_at(e, (0.0, 1.1, 0.0))) for e in [(0.0, 1.1, 3.0), (1.6, 1.4, 2.4), (-2.2, 0.6, 1.2)]]
    raw = [[torch.from_numpy(np.ascontiguousarray(sc[k])).cuda() for k in ("depth", "quality", "silhouette", "color")] for sc in scs]
    torch.cuda.synchronize()
    monkeypatch.setenv("RR_FILL_THREAD", "0")
    plain = rr.ReconIntegrationHip(scs[0], **kw)
    monkeypatch.setenv("RR_FILL_THREAD", "1")
    thr, orc = rr.ReconIntegrationHip(scs[0], **kw), OracleRecon(scs[0], **kw)
    k = 0
    for rounds in range(6):
        thr.enable_timers(rounds % 2 == 1)                 # odd rounds: the calling thread issues the lane's calls (behind the helper's pending jobs)
        for _ in range(5):
            ptrs = [t.data_ptr() for t in raw[k % 3]]
            for o in (thr, plain):
                o.frame_dev(mvs[k % 3], pr, ptrs)
            if k % 4 == 3:                                 # a second draw of the same volume (two pyramids, one volume set), then a read right behind it
                for o in (thr, plain):
                    o.drawF(mvs[(k + 1) % 3], pr)
                (tc, td), (pc, pd) = thr.framebuffer(), plain.framebuffer()
                assert_same(td, pd, f"depth, frame {k}"); assert_same(tc, pc, f"colour, frame {k}")
            k += 1
    orc.upload_frame(scs[(k - 1) % 3]); frame(orc, mvs[(k - 1) % 3], pr)
    assert_same(thr.tsdf(), orc.tsdf(), "volume vs oracle")
    compare_images(thr, orc, "helper thread")
    compare_images(plain, orc, "no helper thread")
    for _ in range(3):                                     # contexts that come and go with a helper that has work behind it
        t2 = rr.ReconIntegrationHip(scs[0], **kw)
        for n in range(4):
            t2.frame_dev(mvs[n % 3], pr, [t.data_ptr() for t in raw[n % 3]])
        t2.close()
