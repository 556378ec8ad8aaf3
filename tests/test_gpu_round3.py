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
            dev.upload_frame_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr() if k != 1 else 0)   # colour is optional
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
