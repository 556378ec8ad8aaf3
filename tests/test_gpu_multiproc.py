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


def _worker(rank, world, port, q, halo, composite, tiny_capacity=False, compositor="shared"):
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
        dedicated = compositor == "dedicated"
        slab = (mgpu.worker_slab_range if dedicated else mgpu.slab_range)(KW["res"][2], rank, world)
        hip = rr.ReconIntegrationHip(scene, slab=slab, recompute_halo=(halo == "recompute"), **KW)
        drv = mgpu.SlabDriver(hip, rank, world, "cuda:0", view=KW["view"], halo=halo, composite=composite, compositor=compositor)      # (creates and hands over its own torch stream)
        if tiny_capacity:
            drv._capacity = lambda f: 64             # every gather is too small: finish() has to repair each frame it is asked about
        for _ in range(5):                           # past the LAG frames that gather the full capacity; buffers are reused
            drv.frame(mv, pr)
        drv.finish()
        torch.cuda.synchronize()
        ok = True
        if tiny_capacity:
            ok &= drv.regathers == 1
        if rank == 0:
            whole = rr.ReconIntegrationHip(scene, **KW)
            whole.clearOccupiedBricks(); whole.markBricks(); whole.updateOccupiedBricks()
            whole.integrate(); whole.drawF(mv, pr)
            (wa, wd, wn, _), (sa, sd, sn, _) = whole.view_images(), hip.view_images()
            if dedicated and composite == "compact":     # the compositor does not march: the write-only count image holds 0 where no slab hit
                ok &= bool((sn[wd < 1] == wn[wd < 1]).all()) and bool(((sn == 0) | (sn == wn))[~(wd < 1)].all())   # (a long ray that missed ships its count)
            else:
                ok &= bool((sn == wn).all())
            ok &= bool((sd == wd).all())
            ok &= bool(((sa == wa) | (np.isnan(sa) & np.isnan(wa))).all())
            (wc, wdd), (sc, sdd) = whole.framebuffer(), hip.framebuffer()
            ok &= bool((sdd == wdd).all()) and bool(((sc == wc) | (np.isnan(sc) & np.isnan(wc))).all())
            ok &= int((wd < 1).sum()) > 300
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("halo,composite,tiny", [("exchange", "dense", False), ("recompute", "compact", False), ("recompute", "compact", True)])
def test_two_rank_slab_driver_matches_single_context(halo, composite, tiny):
    import torch                     # first import on a fresh box takes minutes: pay it here, not in both children at once
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, halo, composite, tiny)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0
    assert dict(q.get(timeout=5) for _ in range(2)) == {0: True, 1: True}


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("world,halo,composite,tiny", [(2, "recompute", "compact", False), (3, "recompute", "compact", False), (3, "exchange", "dense", False),
                                                       (3, "recompute", "compact", True)])
def test_dedicated_compositor_rank_matches_single_context(world, halo, composite, tiny):
    """compositor="dedicated": rank 0 holds no slab -- it takes part in the collectives, composites and fills holes -- and ranks
    1 .. world-1 split the volume (world 2: one worker with the whole volume -- a two-pass march whose long rays are exported with the hit list; world 3: two slabs).  The
    composite on rank 0 equals the unpartitioned frame; with the compact gather the write-only sample-count image is 0 at the
    pixels no slab hit (the compositor does not march)."""
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, halo, composite, tiny, "dedicated")) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0
    assert dict(q.get(timeout=5) for _ in range(world)) == {r: True for r in range(world)}


def _rccl_alone(port, q, halo, composite):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import rgbd_recon_amd as rr
    mgpu = import_module("rgbd-recon_amd.multigpu")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))      # exactly bench.py's call
    try:
        scene = rr.scene.make_scene(n_streams=4, width=160, height=120, lut_res=32, inv_res=32)
        mv, pr = rr.scene.default_view(*KW["view"])
        hip = rr.ReconIntegrationHip(scene, slab=mgpu.slab_range(KW["res"][2], 0, 1), recompute_halo=(halo == "recompute"), **KW)
        drv = mgpu.SlabDriver(hip, 0, 1, "cuda:0", view=KW["view"], halo=halo, composite=composite, exchange_when_alone=True)
        for _ in range(5):
            drv.frame(mv, pr)
        drv.finish()
        dist.barrier()
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0")                          # bench.py's max-over-ranks reduction
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        whole = rr.ReconIntegrationHip(scene, **KW)
        whole.clearOccupiedBricks(); whole.markBricks(); whole.updateOccupiedBricks()
        whole.integrate(); whole.drawF(mv, pr)
        (wc, wdd), (sc, sdd) = whole.framebuffer(), hip.framebuffer()
        ok = bool((sdd == wdd).all()) and bool(((sc == wc) | (np.isnan(sc) & np.isnan(wc))).all()) and int((wdd < 1).sum()) > 300
        ok &= float(t.item()) == 1.5
        q.put(bool(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("halo,composite", [("exchange", "dense"), ("recompute", "compact")])
def test_slab_exchange_over_rccl_with_one_rank(halo, composite):
    """The collectives of the slab driver through RCCL itself (backend "nccl"), as far as one GPU allows: a world of one rank
    runs pack -> all_gather_into_tensor -> unpack, partial march, export -> (counts all-gather +) gather -> composite, with the
    same tensor views, dtypes and stream as N ranks would.  (Two ranks on one device are refused by RCCL; the two-rank
    tests above use gloo.)"""
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_alone, args=(port, q, halo, composite))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    assert q.get(timeout=5) is True


# ------------------------------------------------------------------------------------------------ native RCCL exchange (comm.cpp), world of one
def _native_alone(q, halo, force_regather):
    import torch                                           # first: librccl of the process is torch's, the library binds to it
    import rgbd_recon_amd as rr
    mgpu = import_module("rgbd-recon_amd.multigpu")
    torch.cuda.set_device(0)
    same = lambda a, b: bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())
    scene = rr.scene.make_scene(n_streams=4, width=160, height=120, lut_res=32, inv_res=32)
    moved = rr.scene.make_scene(n_streams=4, width=160, height=120, lut_res=32, inv_res=32, sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2))
    mv, pr = rr.scene.default_view(*KW["view"])
    hip = rr.ReconIntegrationHip(scene, slab=mgpu.slab_range(KW["res"][2], 0, 1), recompute_halo=(halo == "recompute"), **KW)
    drv = mgpu.SlabDriver(hip, 0, 1, "cuda:0", view=KW["view"], halo=halo, composite="compact", exchange_when_alone=True, native=True,
                          min_capacity=64 if force_regather else 4096, max_capacity=64 if force_regather else 0)
    whole = rr.ReconIntegrationHip(scene, **KW)
    ok = True
    why = []
    for k, sc in enumerate((scene, moved, scene, moved, moved)):
        hip.broadcast_frame(0, sc)                          # the frame arrives on the root: RCCL broadcast (of one) + re-layout on every rank
        drv.frame(mv, pr)
        whole.upload_frame(sc)
        whole.clearOccupiedBricks(); whole.markBricks(); whole.updateOccupiedBricks(); whole.integrate(); whole.drawF(mv, pr)
        if k >= 3 or not force_regather:
            drv.finish()
            (wc, wdd), (sc_, sdd) = whole.framebuffer(), hip.framebuffer()
            good = same(sdd, wdd) and same(sc_, wc) and int((wdd < 1).sum()) > 300
            if not good:
                why.append(f"frame {k}: depth differs at {int((sdd != wdd).sum())} pixels, colour at {int((sc_ != wc).sum())} values")
            ok &= good
    st = hip.comm_stats()
    good = (st["regathers"] >= 1) if force_regather else (st["regathers"] == 0 and st["overflowed_frames"] == 0)
    if not good:
        why.append(f"stats {st}")
    ok &= good
    hip.comm_destroy()
    hip.close()
    q.put("ok" if ok else "; ".join(why))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("halo,force_regather", [("exchange", False), ("recompute", False), ("recompute", True)])
def test_native_rccl_exchange_with_one_rank(halo, force_regather):
    """tsdf_comm_init / tsdf_broadcast_frame / tsdf_halo_exchange / tsdf_composite_gather / tsdf_composite_finish: RCCL called from inside the
    library on the context's stream (what a C++ host drives), a world of one rank on the one GPU: the frames equal an unpartitioned
    context's bit for bit, no host synchronisation inside the frame loop; with 64 records per gather the first gathers are too small
    and tsdf_composite_finish repairs the latest one (the earlier truncated frames are counted)."""
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_native_alone, args=(q, halo, force_regather))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"
