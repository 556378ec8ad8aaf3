"""N > 1 path on CPU: world_size 2 over gloo.  The slab driver (rgbd-recon_amd/multigpu.py) is exercised with a numpy
stand-in that implements the C-ABI hook contract (tsdf_halo_*_dev, tsdf_export_partial_dev, tsdf_composite_dev) on host
pointers, so the neighbour indexing, buffer sizing and gather order of the real exchange are what is tested."""
import ctypes
import os
import socket
from importlib import import_module

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

TILE_LAYER = 2 * 2 * 512          # 2 x 2 tiles per layer
VIEW = (8, 4)


class FakeSlab:
    """Owns `own` tile layers plus `halo` layers either side (where a neighbour exists), like tsdf_ctx."""

    def __init__(self, rank, world, own=3, halo=1):
        self.rank, self.world, self.own, self.halo = rank, world, own, halo
        self.lo = halo if rank > 0 else 0
        self.hi = halo if rank < world - 1 else 0
        self.vol = np.full((self.lo + own + self.hi, TILE_LAYER), np.nan, np.float32)
        self.calls = []
        self.result = None

    # per-frame operator surface (no-ops except integrate/draw)
    def clearOccupiedBricks(self): self.calls.append("clear")
    def markBricks(self): self.calls.append("mark")
    def updateOccupiedBricks(self, want_ratio=True): self.calls.append("update")

    def integrate(self):
        for l in range(self.own):                       # value encodes (global layer, voxel)
            self.vol[self.lo + l] = (self.rank * self.own + l) * 1000.0 + np.arange(TILE_LAYER) % 7
        self.calls.append("integrate")

    def halo_info(self): return self.halo, self.halo * TILE_LAYER * 4

    def halo_pack_dev(self, lo_ptr, hi_ptr):
        n = self.halo * TILE_LAYER * 4
        ctypes.memmove(lo_ptr, self.vol[self.lo:self.lo + self.halo].ctypes.data, n)
        ctypes.memmove(hi_ptr, self.vol[self.lo + self.own - self.halo:self.lo + self.own].ctypes.data, n)

    def halo_unpack_dev(self, below_ptr, above_ptr):
        n = self.halo * TILE_LAYER * 4
        if below_ptr and self.lo:
            ctypes.memmove(self.vol[0:self.lo].ctypes.data, below_ptr, n)
        if above_ptr and self.hi:
            ctypes.memmove(self.vol[self.lo + self.own:].ctypes.data, above_ptr, n)

    def draw(self, mv, proj):
        npx = VIEW[0] * VIEW[1]
        p = np.arange(npx)
        owner_hits = (p % 3 != 2)                       # a third of the pixels miss everywhere
        ns = np.where(owner_hits & (p % self.world == self.rank), (p + 1) * 0.0027, -(40 * 0.0027)).astype(np.float32)
        late = owner_hits & (p % self.world != self.rank) & (p % 5 == 0)
        ns[late] = (p[late] + 50) * 0.0027             # a later crossing further along the same ray: must lose
        self.partial = np.concatenate([np.repeat(np.float32(self.rank + 1), npx * 4), np.where(ns > 0, 0.25 + 0.1 * self.rank, 1.0).astype(np.float32), ns])
        self.calls.append("draw")

    def export_partial_dev(self, ptr):
        ctypes.memmove(ptr, self.partial.ctypes.data, self.partial.nbytes)

    def composite_dev(self, ptr, n):
        npx = VIEW[0] * VIEW[1]
        buf = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_float)), shape=(n, npx * 6)).copy()
        ns = buf[:, npx * 5:]
        pos = np.where(ns > 0, ns, np.inf)
        best = pos.argmin(0)
        any_hit = np.isfinite(pos.min(0))
        self.result = dict(rank_of_pixel=np.where(any_hit, best, -1), ns=np.where(any_hit, pos.min(0), -ns.min(0)),
                           colour=np.where(any_hit, buf[best, np.arange(npx) * 4], 0.0))
        self.calls.append("composite")

    def fillColors(self): self.calls.append("fill")
    def drawF(self, mv, proj): self.calls.append("drawF")


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mgpu = import_module("rgbd-recon_amd.multigpu")
        fake = FakeSlab(rank, world)
        drv = mgpu.SlabDriver(fake, rank, world, "cpu", view=VIEW)
        drv.frame(None, None)
        own = fake.own
        ok = True
        if rank > 0:      # halo below == the lower neighbour's top layer
            ok &= bool((fake.vol[0] == ((rank * own - 1) * 1000.0 + np.arange(TILE_LAYER) % 7)).all())
        if rank < world - 1:
            ok &= bool((fake.vol[-1] == (((rank + 1) * own) * 1000.0 + np.arange(TILE_LAYER) % 7)).all())
        ok &= not np.isnan(fake.vol).any()
        res = None
        if rank == 0:
            r = fake.result
            p = np.arange(VIEW[0] * VIEW[1])
            hits = p % 3 != 2
            ok &= bool((r["rank_of_pixel"][hits] == (p % world)[hits]).all()) and bool((r["rank_of_pixel"][~hits] == -1).all())
            ok &= bool(np.allclose(r["ns"][hits], ((p + 1) * 0.0027)[hits])) and bool(np.allclose(r["ns"][~hits], 40 * 0.0027))
            ok &= bool((r["colour"][hits] == (p % world + 1)[hits]).all())
            ok &= fake.calls == ["clear", "mark", "update", "integrate", "draw", "composite", "fill"]
        else:
            ok &= fake.calls == ["clear", "mark", "update", "integrate", "draw"]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(900)
def test_slab_driver_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    got = dict(q.get(timeout=5) for _ in range(2))
    assert got == {0: True, 1: True}


def test_slab_range_partitions_tile_layers():
    mgpu = import_module("rgbd-recon_amd.multigpu")
    for res_z, world in [(512, 8), (512, 2), (1024, 8), (200, 4), (221, 3), (64, 1)]:
        r = [mgpu.slab_range(res_z, k, world) for k in range(world)]
        assert r[0][0] == 0 and r[-1][1] == res_z
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert all(lo % 8 == 0 for lo, _ in r) and all(hi % 8 == 0 for _, hi in r[:-1])
        sizes = [(hi - lo + 7) // 8 for lo, hi in r]
        assert max(sizes) - min(sizes) <= 1
    assert mgpu.slab_range(512, 3, 8) == (192, 256)
