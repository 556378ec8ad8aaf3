"""-m gpu: the real slab driver (rgbd-recon_amd/multigpu.py) with TWO processes, both on the one GPU of the box, over gloo
(RCCL refuses two ranks on one device; the collective is staged through host memory, everything else -- contexts,
exchange hooks, partial raymarch, composite -- is the production path).  Rank 0 compares with the unpartitioned frame."""
import os
import socket
from importlib import import_module

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KW = dict(res=(64, 64, 64), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.04, view=(160, 90))


def _worker(rank, world, port, q, halo, composite):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import rgbd_recon_amd as rr
    mgpu = import_module("rgbd-recon_amd.multigpu")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        scene = rr.scene.make_scene(n_streams=4, width=160, height=120, lut_res=32, inv_res=32)
        mv, pr = rr.scene.default_view(*KW["view"])
        hip = rr.ReconIntegrationHip(scene, slab=mgpu.slab_range(KW["res"][2], rank, world), recompute_halo=(halo == "recompute"), **KW)
        hip.set_stream(torch.cuda.current_stream().cuda_stream)
        drv = mgpu.SlabDriver(hip, rank, world, "cuda:0", view=KW["view"], halo=halo, composite=composite)
        for _ in range(2):                       # two frames: buffers are reused
            drv.frame(mv, pr)
        torch.cuda.synchronize()
        ok = True
        if rank == 0:
            whole = rr.ReconIntegrationHip(scene, **KW)
            whole.clearOccupiedBricks(); whole.markBricks(); whole.updateOccupiedBricks()
            whole.integrate(); whole.drawF(mv, pr)
            (wa, wd, wn, _), (sa, sd, sn, _) = whole.view_images(), hip.view_images()
            ok &= bool((sd == wd).all()) and bool((sn == wn).all())
            ok &= bool(((sa == wa) | (np.isnan(sa) & np.isnan(wa))).all())
            (wc, wdd), (sc, sdd) = whole.framebuffer(), hip.framebuffer()
            ok &= bool((sdd == wdd).all()) and bool(((sc == wc) | (np.isnan(sc) & np.isnan(wc))).all())
            ok &= int((wd < 1).sum()) > 300
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("halo,composite", [("exchange", "dense"), ("recompute", "compact")])
def test_two_rank_slab_driver_matches_single_context(halo, composite):
    import torch                     # first import on a fresh box takes minutes: pay it here, not in both children at once
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, halo, composite)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0
    assert dict(q.get(timeout=5) for _ in range(2)) == {0: True, 1: True}
