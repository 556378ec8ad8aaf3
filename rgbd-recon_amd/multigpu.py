"""Z-slab partition of one TSDF volume over the GPUs of a node.

One process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm).  The reference has
no multi-GPU path at all; what is partitioned here is the reference's single volume:

  * K1 (integrate) shards by voxels: rank g owns Z planes [z0_g, z1_g) -- no communication.
  * brick occupancy (K6) is replicated: every rank marks all bricks from the (replicated) frame images;
    2 atomics per depth pixel are cheaper than any collective.
  * K2 needs the voxels just across a slab face (trilinear + gradient taps).  Either each rank's boundary tile layers
    are all-gathered and the neighbours' faces land in the local halo (halo="exchange"), or every rank simply integrates
    its halo layers itself (halo="recompute": K1's voxels are independent, one extra tile layer per face costs less than
    any collective).
  * every rank steps the SAME global ray (identical fp32 position sequence) and samples only the
    positions inside its slab; the partial results are gathered on rank 0 -- as full images (composite="dense") or as
    one 32-byte record per hit ray (composite="compact") -- and the hit with the smallest sample count wins, which is
    exactly the single-GPU first zero crossing.
  * K3/K4 (hole filling) are image space and tiny: rank 0 only.
  * compositor="dedicated": rank 0 holds NO slab.  The image-space tail of a frame on the gathering rank (receive, composite, hole
    filling: ~70 us at 1280x720) is longer than a slab's share of the volume work at any of the BASELINE configurations, so with an
    even split rank 0 is the frame's critical path and the other ranks wait for it in every gather.  With a dedicated compositor the
    volume is split over ranks 1 .. N-1 and rank 0 only receives, composites and fills: while it finishes frame f the workers are
    already integrating frame f + 1 (each rank's work is queued on its stream; the gather is the only meeting point), and the frame
    period is max(worker, compositor) instead of their sum.  With the compact composite the compositor does not march, so the
    write-only diagnostic image of sample counts (tex_num_samples, recon_integration.cpp:207-209: cleared and bound for imageStore,
    read by nothing) holds 0 at the pixels no exported ray covers instead of the ray's total sample count; colour, depth, the sample counts of
    hit pixels and the hole-filled framebuffer are those of the single-GPU frame, bit for bit.

The exchange works on plain device pointers across the C ABI (tsdf_halo_*_dev, tsdf_export_partial_dev,
tsdf_composite_dev); torch only owns the buffers and the collective.

Stream order.  The backend's kernels and torch's collectives must share ONE stream: the driver creates an explicit
`torch.cuda.Stream`, hands its (non-zero) handle to the backend and issues every collective under
`torch.cuda.stream(...)`.  (torch's default stream has the handle 0, which `tsdf_set_stream` reads as "back to the
context's own stream": a driver that passed it left the kernels on a private non-blocking stream, unordered against the
collectives.  binding.set_stream now maps 0 to tsdf_adopt_null_stream, and this driver does not rely on it.)

No host synchronisation per frame.  The compact composite gathers a FIXED number of records per rank, decided from the
all-gathered hit counts of the frame `LAG` frames earlier (identical on every rank, read from a pinned copy whose event
completed long ago) times 1.5.  A frame whose slab hit more rays than that is detected from the same counts: `finish()`
-- the call that precedes any read of the result -- re-exports and re-gathers that frame with the exact size, so results
never depend on the guess.  All ranks see the same counts, so they take the same decision at the same frame.
"""
from __future__ import annotations

import contextlib

import torch
import torch.distributed as dist


def worker_slab_range(res_z: int, rank: int, world: int):
    """Slab of rank `rank` when rank 0 is a dedicated compositor: ranks 1 .. world-1 split the volume; rank 0 gets the first worker's
    slab as a placeholder (its context needs SOME valid volume -- a slab may not be thinner than its halo --; it is never integrated
    or marched)."""
    return slab_range(res_z, max(rank - 1, 0), world - 1)


def balanced_slab_ranges(layer_weights, world: int, res_z: int):
    """Contiguous tile-aligned split of the Z axis into `world` slabs of about equal total weight (every slab at least one tile layer):
    the occupied bricks of a scene are not spread evenly over z, and the frame period of the partition is its slowest slab's.
    `layer_weights[l]` is the cost of tile layer l (e.g. its occupied bricks plus a constant for the per-layer floor); every rank
    must pass the same weights.  Returns [(z0, z1)] in voxel planes."""
    w = [float(x) for x in layer_weights]
    layers = len(w)
    assert layers == (res_z + 7) // 8 and 1 <= world <= layers
    cum = [0.0]
    for x in w:
        cum.append(cum[-1] + x)
    bounds = [0]
    for k in range(1, world):
        target = cum[-1] * k / world
        b = bounds[-1] + 1
        while b < layers and cum[b] < target:
            b += 1
        if b > bounds[-1] + 1 and abs(cum[b - 1] - target) <= abs(cum[b] - target):
            b -= 1                                       # the boundary nearest to the target
        b = max(bounds[-1] + 1, min(b, layers - (world - k)))
        bounds.append(b)
    bounds.append(layers)
    return [(bounds[k] * 8, min(bounds[k + 1] * 8, res_z)) for k in range(world)]


def slab_range(res_z: int, rank: int, world: int):
    """Tile-aligned (8 planes) contiguous split of the Z axis; the last rank takes the remainder."""
    layers = (res_z + 7) // 8
    base, extra = divmod(layers, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo * 8, min(hi * 8, res_z)


class SlabDriver:
    """Runs the per-frame call order of source/kinect_client.cpp:569-599,614 on one slab.

    `backend` is a ReconIntegrationHip created with slab=slab_range(...) (tests inject a stand-in with the same hook
    methods).  `buf_device` is where exchange buffers live ("cuda:N"; "cpu" for the stand-in); with a gloo group and
    CUDA buffers the collective is staged through host memory.

    halo       "exchange": all-gather of the boundary tile layers before the raymarch (the backend must NOT have been
               created with recompute_halo); "recompute": the backend integrates its own halo layers, no collective.
    composite  "dense": every rank ships 24 B per view pixel to rank 0; "compact": one 32-byte record per ray that hit in
               the slab, a fixed number per rank and frame (see the module docstring).
    exchange_when_alone   world == 1 normally draws the frame directly; True runs the whole exchange (pack, collectives, composite)
               with the one rank anyway -- used to drive the RCCL calls on a box with a single GPU.
    stream     the torch stream everything runs on (default: a new one).  CUDA buffers only.
    min_capacity  smallest number of hit records gathered per rank (compact composite).
    compositor "shared": every rank owns a slab, rank 0 composites as well; "dedicated" (world >= 2): rank 0 owns no slab -- its
               backend only needs the view (worker_slab_range() gives the workers' slabs) -- and never integrates or marches.
    """

    LAG = 2            # frames between a hit count and its use as the gather size: its pinned copy has long arrived, the host never waits

    def __init__(self, backend, rank, world, buf_device, group=None, view=(1280, 720), halo="exchange", composite="dense", preprocess=False,
                 exchange_when_alone=False, stream=None, min_capacity=4096, compositor="shared", native=False, max_capacity=0):
        """native: the two exchanges run inside the library (tsdf_halo_exchange / tsdf_composite_gather / tsdf_composite_finish, comm.cpp: RCCL
        called from C++ on the context's stream) and this class only issues the frame's calls -- what a C++ host does; torch.distributed is
        then used once, to carry the communicator's 128-byte id from rank 0 to the others.  Compact composite only."""
        assert halo in ("exchange", "recompute") and composite in ("dense", "compact") and compositor in ("shared", "dedicated")
        self.native = bool(native)
        assert not self.native or composite == "compact", "the native exchange gathers hit records (compact composite)"
        assert compositor == "shared" or world >= 2, "a dedicated compositor needs at least one worker rank"
        self.dedicated = compositor == "dedicated"
        self.is_worker = not (self.dedicated and rank == 0)          # owns a slab: marks bricks, integrates, marches, exports
        self.first_worker = 1 if self.dedicated else 0
        self.b, self.rank, self.world, self.dev, self.group = backend, rank, world, torch.device(buf_device), group
        self.view, self.halo, self.composite = view, halo, composite
        self.preprocess = preprocess          # frames start from the raw sensor images: processTextures() instead of markBricks()
        self.exchanging = world > 1 or exchange_when_alone
        self.stage_cpu = self.exchanging and not self.native and dist.get_backend(group) == "gloo" and self.dev.type == "cuda"
        # the collectives are torch's, so the context's kernels go to a torch stream -- but only when there IS an exchange: a context that
        # keeps its own stream has its three lanes (stage overlap) on consecutively created streams, i.e. on different hardware queues
        self.stream = None
        if self.dev.type == "cuda" and self.exchanging and not self.native:
            self.stream = stream if stream is not None else torch.cuda.Stream(self.dev)
            assert self.stream.cuda_stream != 0, "the NULL stream cannot be handed to tsdf_set_stream"
            backend.set_stream(self.stream.cuda_stream)
        self.frame_no = 0
        self.caps = {}                        # frame -> capacity it was gathered with, until its counts have been looked at
        self.last = None                      # (frame number, gathered capacity) of the latest compact frame, until finish() has checked it
        self.regathers = 0                    # frames whose first gather was too small (finish() repaired them)
        self.min_capacity = int(min_capacity)
        self.overflowed_frames = 0            # frames that were composited from truncated record lists and were no longer the latest when finish() ran
        self.verdicts = {}                    # frame -> was it left truncated (frame_status)
        if self.exchanging and self.native:
            ids = [backend.comm_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(ids, src=0, group=group)
            with self._on_stream():
                backend.comm_init(ids[0], rank, world, dedicated_compositor=self.dedicated)
                backend.comm_set_capacity_limits(self.min_capacity, max_capacity)
            self.stage_cpu = False
        elif self.exchanging:
            npx = view[0] * view[1]
            self.npx = npx
            with self._on_stream():
                if halo == "exchange":
                    layers, nbytes = backend.halo_info()
                    n = nbytes // 4
                    self.send = torch.empty((2, n), dtype=torch.float32, device=self.dev)
                    self.gath = torch.empty((world, 2, n), dtype=torch.float32, device=self.dev)
                if composite == "dense":
                    # 24 B / pixel; a compositor without a slab contributes "no hit, no samples" for every pixel (sample count 0)
                    self.part = (torch.empty if self.is_worker else torch.zeros)(npx * 6, dtype=torch.float32, device=self.dev)
                    self.parts = torch.empty((world, npx * 6), dtype=torch.float32, device=self.dev) if rank == 0 else None
                else:
                    self.hitbuf = torch.zeros(8 + npx * 8, dtype=torch.float32, device=self.dev)     # 32 B header + 32 B records; a slab cannot hit more rays than there are pixels
                    self.hitparts = torch.zeros((world, 8 + npx * 8), dtype=torch.float32, device=self.dev) if rank == 0 else None
                    self.counts = torch.zeros((world, 2), dtype=torch.int32, device=self.dev)        # per rank: [records written, rays hit]
                    ring = self.LAG + 1
                    self.counts_host = [torch.zeros((world, 2), dtype=torch.int32, pin_memory=(self.dev.type == "cuda")) for _ in range(ring)]
                    self.counts_evt = [None] * ring
            if self.stream is not None:
                self.stream.synchronize()

    # ------------------------------------------------------------------ plumbing
    def _on_stream(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _all_gather(self, out, inp):
        if self.stage_cpu:
            o = torch.empty(out.numel(), dtype=out.dtype)
            dist.all_gather_into_tensor(o, inp.reshape(-1).cpu(), group=self.group)
            out.view(-1).copy_(o)
        else:
            dist.all_gather_into_tensor(out.view(-1), inp.reshape(-1), group=self.group)

    def _gather0(self, out_rows, inp):
        """gather `inp` (1-D) of every rank into out_rows[r] (1-D views of equal length) on rank 0"""
        if self.stage_cpu:
            lst = [torch.empty(inp.shape, dtype=inp.dtype) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(inp.cpu(), lst, dst=0, group=self.group)
            if self.rank == 0:
                for row, t in zip(out_rows, lst):
                    row.copy_(t)
        else:
            dist.gather(inp, out_rows if self.rank == 0 else None, dst=0, group=self.group)

    def exchange_halo(self):
        b = self.b
        if self.is_worker:
            b.halo_pack_dev(self.send[0].data_ptr(), self.send[1].data_ptr())
        self._all_gather(self.gath, self.send)
        if not self.is_worker:
            return                                                   # (took part in the collective; has no slab faces of its own)
        below = self.gath[self.rank - 1, 1].data_ptr() if self.rank > self.first_worker else 0
        above = self.gath[self.rank + 1, 0].data_ptr() if self.rank < self.world - 1 else 0
        b.halo_unpack_dev(below, above)

    # ------------------------------------------------------------------ compact composite: sizes without a host sync
    def _counts_of(self, f):
        """the all-gathered [written, hit] counts of frame f on the host (waits for their pinned copy: for f <= now - LAG it arrived long ago)"""
        slot = f % (self.LAG + 1)
        if self.counts_evt[slot] is not None:
            self.counts_evt[slot].synchronize()
        return self.counts_host[slot]

    def _capacity(self, f):
        """records gathered per rank for frame f: 1.5 x the largest per-rank hit count of frame f - LAG (the same number on every rank)"""
        if f < self.LAG:
            return self.npx                                # no history yet: a slab cannot hit more rays than there are pixels
        m = int(self._counts_of(f - self.LAG)[:, 1].max())
        truncated = m > self.caps.get(f - self.LAG, self.npx)   # that frame hit more rays than were gathered (finish() repairs the LATEST frame only)
        if truncated:
            self.overflowed_frames += 1
        self._set_verdict(f - self.LAG, truncated)
        self.caps.pop(f - self.LAG, None)
        cap = max(self.min_capacity, ((m * 3) // 2 + 1024 + 1023) // 1024 * 1024)
        return min(cap, self.npx)

    def _set_verdict(self, f, truncated):
        self.verdicts[f] = bool(truncated)
        self.verdicts.pop(f - 64, None)                    # the last 64 frames are kept (comm.cpp keeps the same ring)

    def frame_status(self, f):
        """True: frame f (counted from 0) of the compact composite was built from truncated record lists and not repaired -- a caller that kept
        that frame must redraw it.  Known at once for the latest LAG + 1 frames (waits for their counts), from the last 64 verdicts otherwise;
        KeyError: not gathered yet / too old.  (tsdf_comm_frame_status is the same rule inside the library.)"""
        if self.exchanging and self.native:
            return self.b.comm_frame_status(f)
        if f in self.verdicts:
            return self.verdicts[f]
        if f < self.frame_no and f + self.LAG + 1 >= self.frame_no:
            cap = self.caps.get(f, self.last[1] if self.last is not None and self.last[0] == f else self.npx)
            return int(self._counts_of(f)[:, 1].max()) > cap
        raise KeyError(f"no verdict for frame {f} ({self.frame_no} frames gathered)")

    def _exchange_hits(self, cap, record_counts_of=None):
        """record_counts_of: the frame number on the frame's first exchange, None on a repeated one (finish()).  Rank 0's own records never
        travel: it exports them -- ALL of them, whatever the capacity of the gather -- straight into its row of the gather buffer, once
        per frame (a repeated gather must not export them again: the first composite has overwritten the march target they are read from)."""
        b = self.b
        first = record_counts_of is not None
        own = self.hitparts[0] if self.rank == 0 else self.hitbuf
        if self.is_worker and (first or self.rank != 0):
            b.export_hits_dev(own.data_ptr(), self.npx if self.rank == 0 else cap)     # (a compositor's header stays {0 records, 0 hits})
        if record_counts_of is not None:
            self._all_gather(self.counts, own[:2].view(torch.int32))
            slot = record_counts_of % (self.LAG + 1)
            self.counts_host[slot].copy_(self.counts, non_blocking=True)
            if self.stream is not None:
                ev = self.counts_evt[slot] or torch.cuda.Event()
                ev.record(self.stream)
                self.counts_evt[slot] = ev
        n = 8 + cap * 8
        if self.world > 1:      # (rank 0's contribution to the collective is a header-sized dummy: its row is in place already)
            if self.rank == 0 and getattr(self, "selfrow", None) is None:
                self.selfrow = torch.zeros_like(self.hitbuf)
            self._gather0([self.selfrow[:n]] + [self.hitparts[r, :n] for r in range(1, self.world)] if self.rank == 0 else None, self.hitbuf[:n])
        if self.rank == 0:
            b.composite_hits_dev(self.hitparts.data_ptr(), self.world, self.hitparts.stride(0) * 4)
            b.fillColors()

    # ------------------------------------------------------------------ frame distribution ("ncclBroadcast from the receiving rank")
    def broadcast_frame(self, root, tensors=None):
        """The frame arrived on `root` (there: its four arrays as device tensors -- depth RG32F, quality, silhouette, colour RGB8 --, elsewhere
        None): every rank gets a copy and the ranks that own a slab re-lay it out.  native: tsdf_broadcast_frame (RCCL called from C++);
        else ONE torch.distributed broadcast of the packed arrays, then tsdf_upload_frame_dev."""
        with self._on_stream():
            if self.native:
                self.b.broadcast_frame(root, dev_ptrs=[t.data_ptr() for t in tensors] if self.rank == root else None)
                return
            n, (h, w, ch, cw) = self.b.n, self.b._dims
            npx, ncp = n * h * w, n * ch * cw
            offs = (0, npx * 8, npx * 12, npx * 16)
            if getattr(self, "frame_stage", None) is None:
                self.frame_stage = torch.empty(npx * 16 + ((ncp * 3 + 15) // 16) * 16, dtype=torch.uint8, device=self.dev)
            st = self.frame_stage
            if self.rank == root:
                for off, t in zip(offs, tensors):
                    b = t.contiguous().view(torch.uint8).reshape(-1)
                    st[off:off + b.numel()].copy_(b)
            if self.world > 1:
                if self.stage_cpu:
                    t = st.cpu()
                    dist.broadcast(t, src=root, group=self.group)
                    st.copy_(t)
                else:
                    dist.broadcast(st, src=root, group=self.group)
            if self.is_worker:
                p = st.data_ptr()
                self.b.upload_frame_dev(p + offs[0], p + offs[1], p + offs[2], p + offs[3])

    # ------------------------------------------------------------------ per frame
    def frame(self, mv, proj, new_frame=None):
        """new_frame: device pointers (depth_rg, quality, silhouette, colour) of a frame that has just arrived in device memory -- the
        ranks that own a slab re-lay it out first (tsdf_upload_frame_dev: what a new frame costs the path itself)"""
        with self._on_stream():
            self._frame(mv, proj, new_frame)

    def _frame(self, mv, proj, new_frame=None):
        b = self.b
        if self.is_worker and not self.exchanging and not self.preprocess:   # one rank, nothing to exchange: the whole frame in one call into the library
            b.frame_dev(mv, proj, new_frame)
            return
        if self.is_worker and not self.exchanging and self.preprocess and new_frame is not None:   # ... and from the RAW frame (depth_raw, colour)
            b.frame_raw_dev(mv, proj, new_frame)
            return
        if self.is_worker and new_frame is not None:
            if self.preprocess:
                b.upload_raw_frame_dev(*new_frame, complete=True)
            else:
                b.upload_frame_dev(*new_frame, complete=True)
        if self.is_worker:
            b.clearOccupiedBricks()
            if self.preprocess:
                b.processTextures()
            else:
                b.markBricks()
            b.updateOccupiedBricks(False)
            b.integrate()
        if not self.exchanging:
            b.drawF(mv, proj)
            return
        if self.native:
            if self.halo == "exchange":
                b.halo_exchange()
            if self.is_worker:
                b.draw(mv, proj)
            b.composite_gather()
            self.last = True
            return
        if self.halo == "exchange":
            self.exchange_halo()
        if self.is_worker:
            b.draw(mv, proj)
        if self.composite == "dense":
            if self.is_worker:
                b.export_partial_dev(self.part.data_ptr())
            self._gather0(list(self.parts.unbind(0)) if self.rank == 0 else None, self.part)
            if self.rank == 0:
                b.composite_dev(self.parts.data_ptr(), self.world)
                b.fillColors()
            return
        f = self.frame_no
        cap = self._capacity(f)
        self._exchange_hits(cap, record_counts_of=f)
        self.caps[f] = cap
        self.last = (f, cap)
        self.frame_no = f + 1

    def finish(self):
        """Completes the latest frame: call before reading its result (and at the end of a timed region).  A COLLECTIVE when
        the compact gather of that frame turned out too small -- every rank sees the same counts and re-gathers together."""
        if self.exchanging and self.native:
            if self.last is not None:
                self.last = None
                with self._on_stream():
                    self.b.composite_finish()
                st = self.b.comm_stats()
                self.regathers, self.overflowed_frames = st["regathers"], st["overflowed_frames"]
        elif self.exchanging and self.composite == "compact" and self.last is not None:
            f, cap = self.last
            self.last = None
            m = int(self._counts_of(f)[:, 1].max())
            if m > cap:
                self.regathers += 1
                self.caps[f] = min(self.npx, m)
                with self._on_stream():
                    self._exchange_hits(min(self.npx, m))
            self._set_verdict(f, False)                      # complete: gathered in full, or repaired just now
        if self.stream is not None:
            self.stream.synchronize()
        elif hasattr(self.b, "sync"):
            self.b.sync()


def frame_slabs_on_one_device(backends, mv, proj, device, halo="exchange", composite="dense"):
    """The same partition with ONE visible device (SURVEY.md §8e fallback): the slabs run one after another and the
    "collectives" are device-to-device copies.  Used to check slab results against the unpartitioned volume on a 1-GPU box.
    backends[k] owns slab k; the composite lands in backends[0]."""
    world = len(backends)
    npx = backends[0].view[0] * backends[0].view[1]
    if halo == "exchange":
        layers, nbytes = backends[0].halo_info()
        faces = torch.empty((world, 2, nbytes // 4), dtype=torch.float32, device=device)
    for k, b in enumerate(backends):
        b.clearOccupiedBricks(); b.markBricks(); b.updateOccupiedBricks(False)
        b.integrate()
        if halo == "exchange":
            b.halo_pack_dev(faces[k, 0].data_ptr(), faces[k, 1].data_ptr())
        b.sync()
    parts = torch.zeros((world, npx * 6 if composite == "dense" else 8 + npx * 8), dtype=torch.float32, device=device)
    torch.cuda.synchronize()                      # the contexts run on their own streams: the buffers must exist before they write them
    for k, b in enumerate(backends):
        if halo == "exchange":
            below = faces[k - 1, 1].data_ptr() if k > 0 else 0
            above = faces[k + 1, 0].data_ptr() if k < world - 1 else 0
            b.halo_unpack_dev(below, above)
        b.draw(mv, proj)
        if composite == "dense":
            b.export_partial_dev(parts[k].data_ptr())
        else:
            b.export_hits_dev(parts[k].data_ptr(), npx)
        b.sync()
    if composite == "dense":
        backends[0].composite_dev(parts.data_ptr(), world)
    else:
        backends[0].composite_hits_dev(parts.data_ptr(), world, parts.stride(0) * 4)
    backends[0].fillColors()
    backends[0].sync()
