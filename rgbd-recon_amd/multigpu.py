"""Z-slab partition of one TSDF volume over the GPUs of a node.

One process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm).  The reference has
no multi-GPU path at all; what is partitioned here is the reference's single volume:

  * K1 (integrate) shards by voxels: rank g owns Z planes [z0_g, z1_g) -- no communication.
  * brick occupancy (K6) is replicated: every rank marks all bricks from the (replicated) frame images;
    2 atomics per depth pixel are cheaper than any collective.
  * K2 needs ONE exchange before it: trilinear + gradient taps reach across a slab face, so each rank's
    boundary tile layers are all-gathered and the two neighbours' faces land in the local halo.
  * every rank steps the SAME global ray (identical fp32 position sequence) and samples only the
    positions inside its slab; the partial images are gathered on rank 0 and the hit with the smallest
    sample count wins, which is exactly the single-GPU first zero crossing.
  * K3/K4 (hole filling) are image space and tiny: rank 0 only.

The exchange works on plain device pointers across the C ABI (tsdf_halo_*_dev, tsdf_export_partial_dev,
tsdf_composite_dev); torch only owns the buffers and the collective.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def slab_range(res_z: int, rank: int, world: int):
    """Tile-aligned (8 planes) contiguous split of the Z axis; the last rank takes the remainder."""
    layers = (res_z + 7) // 8
    base, extra = divmod(layers, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo * 8, min(hi * 8, res_z)


class SlabDriver:
    """Runs the per-frame call order of source/kinect_client.cpp:569-599,614 on one slab.

    `backend` is a ReconIntegrationHip created with slab=slab_range(...) (tests inject an oracle-backed fake
    with the same hook methods).  `buf_device` is where exchange buffers live ("cuda:N"; "cpu" for the fake);
    with a gloo group and CUDA buffers the collective is staged through host memory.
    """

    def __init__(self, backend, rank, world, buf_device, group=None, view=(1280, 720)):
        self.b, self.rank, self.world, self.dev, self.group = backend, rank, world, torch.device(buf_device), group
        self.view = view
        self.stage_cpu = world > 1 and dist.get_backend(group) == "gloo" and self.dev.type == "cuda"
        if world > 1:
            layers, nbytes = backend.halo_info()
            n = nbytes // 4
            self.send = torch.empty((2, n), dtype=torch.float32, device=self.dev)
            self.gath = torch.empty((world, 2, n), dtype=torch.float32, device=self.dev)
            npx = view[0] * view[1]
            self.part = torch.empty(npx * 6, dtype=torch.float32, device=self.dev)          # 24 B / pixel
            self.parts = torch.empty((world, npx * 6), dtype=torch.float32, device=self.dev) if rank == 0 else None

    def _all_gather(self, out, inp):
        if self.stage_cpu:
            o = torch.empty(out.numel(), dtype=out.dtype)
            dist.all_gather_into_tensor(o, inp.reshape(-1).cpu(), group=self.group)
            out.view(-1).copy_(o)
        else:
            dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=self.group)

    def _gather0(self, out, inp):
        if self.stage_cpu:
            lst = [torch.empty(inp.shape, dtype=inp.dtype) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(inp.cpu(), lst, dst=0, group=self.group)
            if self.rank == 0:
                out.copy_(torch.stack(lst))
        else:
            lst = list(out.unbind(0)) if self.rank == 0 else None
            dist.gather(inp, lst, dst=0, group=self.group)

    def exchange_halo(self):
        b = self.b
        b.halo_pack_dev(self.send[0].data_ptr(), self.send[1].data_ptr())
        self._all_gather(self.gath, self.send)
        below = self.gath[self.rank - 1, 1].data_ptr() if self.rank > 0 else 0
        above = self.gath[self.rank + 1, 0].data_ptr() if self.rank < self.world - 1 else 0
        b.halo_unpack_dev(below, above)

    def frame(self, mv, proj):
        b = self.b
        b.clearOccupiedBricks()
        b.markBricks()
        b.updateOccupiedBricks(False)
        b.integrate()
        if self.world == 1:
            b.drawF(mv, proj)
            return
        self.exchange_halo()
        b.draw(mv, proj)
        b.export_partial_dev(self.part.data_ptr())
        self._gather0(self.parts, self.part)
        if self.rank == 0:
            b.composite_dev(self.parts.data_ptr(), self.world)
            b.fillColors()


def frame_slabs_on_one_device(backends, mv, proj, device):
    """The same partition with ONE visible device (SURVEY.md §8e fallback): the slabs run one after another and the
    "collectives" are device-to-device copies.  Used to check slab results against the unpartitioned volume on a 1-GPU box.
    backends[k] owns slab k; the composite lands in backends[0]."""
    world = len(backends)
    layers, nbytes = backends[0].halo_info()
    n = nbytes // 4
    faces = torch.empty((world, 2, n), dtype=torch.float32, device=device)
    npx = backends[0].view[0] * backends[0].view[1]
    parts = torch.empty((world, npx * 6), dtype=torch.float32, device=device)
    for k, b in enumerate(backends):
        b.clearOccupiedBricks(); b.markBricks(); b.updateOccupiedBricks(False)
        b.integrate()
        b.halo_pack_dev(faces[k, 0].data_ptr(), faces[k, 1].data_ptr())
        b.sync()
    for k, b in enumerate(backends):
        below = faces[k - 1, 1].data_ptr() if k > 0 else 0
        above = faces[k + 1, 0].data_ptr() if k < world - 1 else 0
        b.halo_unpack_dev(below, above)
        b.draw(mv, proj)
        b.export_partial_dev(parts[k].data_ptr())
        b.sync()
    backends[0].composite_dev(parts.data_ptr(), world)
    backends[0].fillColors()
    backends[0].sync()
