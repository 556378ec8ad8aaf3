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
