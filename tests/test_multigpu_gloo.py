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

    def __init__(self, rank, world, own=3, halo=1, layer0=None, view=None):
        self.rank, self.world, self.own, self.halo = rank, world, own, halo
        self.layer0 = rank * own if layer0 is None else layer0      # first global tile layer of the slab (uneven slabs pass it)
        self.view = VIEW if view is None else view
        self.lo = halo if rank > 0 else 0
        self.hi = halo if rank < world - 1 else 0
        self.vol = np.full((self.lo + own + self.hi, TILE_LAYER), np.nan, np.float32)
        self.calls = []
        self.result = None
        self.exports = []
        self.extra_hits = 0

    # per-frame operator surface (no-ops except integrate/draw)
    def clearOccupiedBricks(self): self.calls.append("clear")
    def markBricks(self): self.calls.append("mark")
    def updateOccupiedBricks(self, want_ratio=True): self.calls.append("update")

    def integrate(self):
        for l in range(self.own):                       # value encodes (global layer, voxel)
            self.vol[self.lo + l] = (self.layer0 + l) * 1000.0 + np.arange(TILE_LAYER) % 7
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
        npx = self.view[0] * self.view[1]
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
        npx = self.view[0] * self.view[1]
        buf = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_float)), shape=(n, npx * 6)).copy()
        ns = buf[:, npx * 5:]
        pos = np.where(ns > 0, ns, np.inf)
        best = pos.argmin(0)
        any_hit = np.isfinite(pos.min(0))
        self.result = dict(rank_of_pixel=np.where(any_hit, best, -1), ns=np.where(any_hit, pos.min(0), -ns.min(0)),
                           colour=np.where(any_hit, buf[best, np.arange(npx) * 4], 0.0))
        self.calls.append("composite")

    # compact exchange (tsdf_export_hits_dev / tsdf_composite_hits_dev): 32-byte header {written, hit, overflow} + 32-byte records
    def export_hits_dev(self, ptr, cap):
        npx = self.view[0] * self.view[1]
        ns = self.partial[npx * 5:]
        pix = np.nonzero(ns > 0)[0]
        if self.extra_hits:                                # frames with many more hits than the history predicts
            pix = np.concatenate([pix, np.repeat(pix[:1], self.extra_hits)])
        n = min(pix.size, cap)
        buf = np.zeros(8 + cap * 8, np.float32)
        hdr = buf[:8].view(np.uint32)
        hdr[0], hdr[1], hdr[2] = n, pix.size, int(pix.size > cap)
        rec = buf[8:].reshape(cap, 8)
        rec[:n, 0] = pix[:n].astype(np.uint32).view(np.float32)
        rec[:n, 1] = ns[pix[:n]]
        rec[:n, 2] = self.partial[npx * 4:npx * 5][pix[:n]]
        rec[:n, 4:] = self.partial[:npx * 4].reshape(npx, 4)[pix[:n]]
        ctypes.memmove(ptr, buf.ctypes.data, buf.nbytes)
        self.exports.append((cap, int(pix.size)))

    def composite_hits_dev(self, ptr, n, stride_bytes):
        npx = self.view[0] * self.view[1]
        best = np.full(npx, np.inf)
        rank_of = np.full(npx, -1)
        colour = np.zeros(npx)
        for r in range(n):
            base = ctypes.cast(ptr + r * stride_bytes, ctypes.POINTER(ctypes.c_float))
            hdr = np.ctypeslib.as_array(base, shape=(8,)).view(np.uint32)
            cnt = int(hdr[0])
            rec = np.ctypeslib.as_array(base, shape=(8 + cnt * 8,))[8:].reshape(cnt, 8)
            for k in range(cnt):
                p = int(rec[k, 0:1].view(np.uint32)[0])
                if rec[k, 1] < best[p]:
                    best[p], rank_of[p], colour[p] = rec[k, 1], r, rec[k, 4]
        self.result = dict(rank_of_pixel=rank_of, ns=best, colour=colour)
        self.calls.append("composite")

    def sync(self): pass
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


def _compact_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mgpu = import_module("rgbd-recon_amd.multigpu")
        fake = FakeSlab(rank, world)
        drv = mgpu.SlabDriver(fake, rank, world, "cpu", view=VIEW, halo="recompute", composite="compact", min_capacity=4)
        npx = VIEW[0] * VIEW[1]
        p = np.arange(npx)
        hits = p % 3 != 2

        def check():
            r = fake.result
            return bool((r["rank_of_pixel"][hits] == (p % world)[hits]).all()) and bool((r["rank_of_pixel"][~hits] == -1).all()) and \
                bool(np.allclose(r["ns"][hits], ((p + 1) * 0.0027)[hits])) and bool((r["colour"][hits] == (p % world + 1)[hits]).all())

        ok = True
        for f in range(6):
            drv.frame(None, None)
        drv.finish()
        # frames 0, 1 (no history) gather the full capacity; from frame LAG on: max(min_capacity, 1.5 x hits of frame f - LAG + 1024 rounded), capped at npx
        caps = [c for c, _ in fake.exports]
        ok &= caps[:2] == [npx, npx] and all(c == npx for c in caps)          # the 1024-record margin exceeds this 32-pixel view: full size, never a regather
        ok &= drv.regathers == 0
        if rank == 0:
            ok &= check()
        # a capacity rule that under-sizes every gather (what a frame with many more hits than two frames ago looks like): the
        # result is only right once finish() has re-gathered -- a collective decision taken from the all-gathered counts
        drv._capacity = lambda f: 2
        n0 = len(fake.exports)
        for f in range(3):
            drv.frame(None, None)
        if rank == 0:
            ok &= not check()                                                 # two records per rank are not the frame
        drv.finish()
        # three frames + ONE re-export with the exact size -- except on rank 0, whose own records never travel: it exports all of them once per
        # frame straight into its row of the gather buffer and must NOT export again (the first composite overwrote the march target)
        exact = max(int(((p % world == r) & hits).sum()) + int(((p % world != r) & hits & (p % 5 == 0)).sum()) for r in range(world))
        ok &= drv.regathers == 1 and len(fake.exports) == n0 + (3 if rank == 0 else 4)
        ok &= fake.exports[-1][0] == (npx if rank == 0 else exact)
        if rank == 0:
            ok &= check()
        drv.finish()                                                          # idempotent
        ok &= drv.regathers == 1
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_compact_composite_sizes_its_gather_without_a_host_sync_and_repairs_overflow():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_compact_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert dict(q.get(timeout=5) for _ in range(2)) == {0: True, 1: True}


def test_capacity_rule():
    """1.5 x the largest per-rank hit count of LAG frames ago + 1024, rounded up to 1024, at least min_capacity, at most every pixel"""
    mgpu = import_module("rgbd-recon_amd.multigpu")

    class D(mgpu.SlabDriver):
        def __init__(self, counts, npx, min_capacity):
            self.npx, self.min_capacity, self._c = npx, min_capacity, counts
            self.caps, self.overflowed_frames, self.verdicts = {}, 0, {}

        def _counts_of(self, f):
            return torch.tensor(self._c[f], dtype=torch.int32)

    d = D({0: [[10, 10], [7000, 9000]], 1: [[0, 0], [0, 0]], 5: [[1, 900000], [1, 5]]}, 921600, 4096)
    assert d._capacity(0) == 921600 and d._capacity(1) == 921600              # no history
    assert d._capacity(2) == 15360                                            # 1.5 * 9000 + 1024 = 14524 -> the next multiple of 1024
    assert d._capacity(3) == 4096                                             # nothing hit two frames ago: the floor
    assert d._capacity(7) == 921600                                           # capped at one record per pixel
    assert d.overflowed_frames == 0
    # a frame that hit more rays than were gathered for it, and was no longer the latest when finish() ran, is counted
    d.caps[5] = 4096                                                          # frame 5 was gathered with 4096 records; rank 0 hit 900000 rays
    d._capacity(7)
    assert d.overflowed_frames == 1


# ------------------------------------------------------------------------------------------------ dedicated compositor rank
def _dedicated_worker(rank, world, port, q, halo, composite):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mgpu = import_module("rgbd-recon_amd.multigpu")
        workers = world - 1
        # rank 0 owns no slab: the stand-in for its context is a slab nobody integrates or draws; workers are slabs 0 .. workers-1
        fake = FakeSlab(max(rank - 1, 0), workers)
        drv = mgpu.SlabDriver(fake, rank, world, "cpu", view=VIEW, halo=halo, composite=composite, min_capacity=4, compositor="dedicated")
        for _ in range(4):
            drv.frame(None, None)
        drv.finish()
        ok = True
        frame_calls = fake.calls[-(7 if rank == 0 else 5):]
        if rank == 0:
            ok &= set(fake.calls) == {"composite", "fill"}                   # never marks, integrates or draws
            r = fake.result
            p = np.arange(VIEW[0] * VIEW[1])
            hits = p % 3 != 2
            # a gathered part's index is the RANK (worker index + 1); part 0 (the compositor's own) never wins a pixel
            ok &= bool((r["rank_of_pixel"][hits] == (p % workers + 1)[hits]).all()) and bool((r["rank_of_pixel"][~hits] == -1).all())
            ok &= bool(np.allclose(r["ns"][hits], ((p + 1) * 0.0027)[hits])) and bool((r["colour"][hits] == (p % workers + 1)[hits]).all())
            if composite == "dense":
                ok &= bool(np.allclose(r["ns"][~hits], 40 * 0.0027))          # whole partial images carry the miss counts
        else:
            ok &= frame_calls == ["clear", "mark", "update", "integrate", "draw"]
            if halo == "exchange":                                            # neighbours are worker ranks only: rank 1 has none below
                own, w = fake.own, rank - 1
                if w > 0:
                    ok &= bool((fake.vol[0] == ((w * own - 1) * 1000.0 + np.arange(TILE_LAYER) % 7)).all())
                if w < workers - 1:
                    ok &= bool((fake.vol[-1] == (((w + 1) * own) * 1000.0 + np.arange(TILE_LAYER) % 7)).all())
                ok &= not np.isnan(fake.vol).any()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("halo,composite", [("recompute", "compact"), ("exchange", "dense")])
def test_dedicated_compositor_world_size_3_gloo(halo, composite):
    """compositor="dedicated": rank 0 takes part in every collective but owns no slab; ranks 1 and 2 are slabs 0 and 1."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dedicated_worker, args=(r, 3, port, q, halo, composite)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert dict(q.get(timeout=5) for _ in range(3)) == {0: True, 1: True, 2: True}


def test_worker_slab_range():
    mgpu = import_module("rgbd-recon_amd.multigpu")
    for res_z, world in [(512, 8), (512, 2), (1024, 8), (221, 3)]:
        r = [mgpu.worker_slab_range(res_z, k, world) for k in range(world)]
        assert r[0] == r[1]                                                   # the compositor's placeholder: a valid slab (never integrated)
        assert r[1:] == [mgpu.slab_range(res_z, k, world - 1) for k in range(world - 1)]
    with pytest.raises(AssertionError):
        mgpu.SlabDriver(None, 0, 1, "cpu", compositor="dedicated")


def test_balanced_slab_ranges():
    mgpu = import_module("rgbd-recon_amd.multigpu")
    assert mgpu.balanced_slab_ranges([1] * 64, 8, 512) == [mgpu.slab_range(512, k, 8) for k in range(8)]
    w = [0.2] * 20 + [5] * 10 + [10] * 8 + [5] * 10 + [0.2] * 16
    for world in (2, 3, 7, 8):
        r = mgpu.balanced_slab_ranges(w, world, 512)
        assert r[0][0] == 0 and r[-1][1] == 512 and all(a[1] == b[0] for a, b in zip(r, r[1:])) and all(hi > lo and lo % 8 == 0 for lo, hi in r)
        cost = [sum(w[lo // 8:(hi + 7) // 8]) for lo, hi in r]
        assert max(cost) <= 1.6 * sum(w) / world                              # no slab much heavier than its share
    assert mgpu.balanced_slab_ranges([0] * 63 + [1], 4, 512) == [(0, 488), (488, 496), (496, 504), (504, 512)]   # every slab at least one layer
    assert mgpu.balanced_slab_ranges([1, 1, 1], 3, 20) == [(0, 8), (8, 16), (16, 20)]


# ------------------------------------------------------------------------------------------------ the shape of the first real run: world = 8
VIEW8 = (128, 64)          # 8192 pixels: large enough that the capacity rule (1.5 x hits + 1024, rounded to 1024) gathers LESS than every pixel


def _world8_worker(rank, world, port, q, halo, layers):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mgpu = import_module("rgbd-recon_amd.multigpu")
        workers = world - 1
        z0, z1 = mgpu.worker_slab_range(layers * 8, rank, world)        # uneven: 64 or 128 tile layers do not divide by 7
        own = (z1 - z0) // 8
        fake = FakeSlab(max(rank - 1, 0), workers, own=own, layer0=z0 // 8, view=VIEW8)
        drv = mgpu.SlabDriver(fake, rank, world, "cpu", view=VIEW8, halo=halo, composite="compact", min_capacity=4, compositor="dedicated")
        npx = VIEW8[0] * VIEW8[1]
        p = np.arange(npx)
        hits = p % 3 != 2

        def check():
            r = fake.result
            return bool((r["rank_of_pixel"][hits] == (p % workers + 1)[hits]).all()) and bool((r["rank_of_pixel"][~hits] == -1).all()) and \
                bool(np.allclose(r["ns"][hits], ((p + 1) * 0.0027)[hits])) and bool((r["colour"][hits] == (p % workers + 1)[hits]).all())

        ok = True
        for _ in range(4):
            drv.frame(None, None)
        drv.finish()
        caps = [c for c, _ in fake.exports]
        if rank > 0:
            ok &= caps[:2] == [npx, npx] and all(c < npx for c in caps[2:])   # from frame LAG on the gather is sized from the history
            ok &= fake.calls[-5:] == ["clear", "mark", "update", "integrate", "draw"]
            if halo == "exchange":                                            # neighbours are worker ranks only
                w = rank - 1
                if w > 0:
                    ok &= bool((fake.vol[0] == ((fake.layer0 - 1) * 1000.0 + np.arange(TILE_LAYER) % 7)).all())
                if w < workers - 1:
                    ok &= bool((fake.vol[-1] == ((fake.layer0 + own) * 1000.0 + np.arange(TILE_LAYER) % 7)).all())
                ok &= not np.isnan(fake.vol).any()
        else:
            ok &= check() and set(fake.calls) == {"composite", "fill"}
        ok &= drv.regathers == 0 and drv.overflowed_frames == 0 and not any(drv.frame_status(f) for f in range(4))
        # frame 4: worker rank 3 hits far more rays than the history predicts.  Not the latest when finish() runs -> composited from truncated lists:
        # counted AND marked (VERDICT r03 "next" 5: no silently wrong frame).  Frame 7 overflows too but IS the latest: finish() re-gathers it.
        for f in range(4, 8):
            fake.extra_hits = 4000 if (rank == 3 and f in (4, 7)) else 0
            drv.frame(None, None)
        drv.finish()
        ok &= drv.regathers == 1
        if rank == 0:
            ok &= check()                                                     # the repaired frame 7
        fake.extra_hits = 0
        for f in range(8, 11):
            drv.frame(None, None)
        drv.finish()
        ok &= drv.overflowed_frames == 1 and drv.regathers == 1
        ok &= [drv.frame_status(f) for f in range(4, 11)] == [True, False, False, False, False, False, False]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("halo,layers", [("recompute", 64), ("exchange", 128)])
def test_world_8_dedicated_compositor_uneven_slabs_gloo(halo, layers):
    """world = 8 as the first real run will have it: rank 0 composites, ranks 1-7 hold uneven Z-slabs of the 64 (512^3) / 128 (1024^3) tile
    layers; both halo modes; a frame that overflows its gather while it is not the latest is counted and marked, one that is the latest is repaired."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_world8_worker, args=(r, 8, port, q, halo, layers)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(1200)
        assert p.exitcode == 0
    assert dict(q.get(timeout=5) for _ in range(8)) == {r: True for r in range(8)}
