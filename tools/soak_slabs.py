"""Long-run determinism soak of the slab partition with a dedicated compositor: N ranks on ONE GPU over gloo (RCCL refuses two ranks per
device), two frames (objects moved) and two views alternating; every CHECK frames the composite on rank 0 (raymarch target + hole-filled
framebuffer) must hash like the first frame of its kind.  A race between the ranks' streams, a stale lagged capacity or a dropped record
would show as a drifting hash.
    RR_BENCH_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29650 tools/soak_slabs.py [FRAMES] [CHECK]"""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from importlib import import_module
rr = import_module("rgbd-recon_amd")
mg = import_module("rgbd-recon_amd.multigpu")

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
CHECK = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.cuda.set_device(int(os.environ.get("RR_BENCH_DEVICE", "0")))
VIEW, RES = (1280, 720), 512
mk = dict(n_streams=4, width=640, height=480, lut_res=128, inv_res=128)
scene = rr.scene.make_scene(**mk)
scene_b = rr.scene.make_scene(sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2), **mk)
ext = scene["bbox_max"] - scene["bbox_min"]
hip = rr.ReconIntegrationHip(scene, res=(RES,) * 3, brick_size=[float(ext[a]) / RES * 8 for a in range(3)], limit=0.01, view=VIEW,
                             slab=mg.worker_slab_range(RES, rank, world), recompute_halo=True)
# round 3: every frame ARRIVES (device arrays -> tsdf_upload_frame_dev on the ranks that own a slab), so each worker's lanes -- lane ahead, integrate lane on two
# volume sets of the slab, the context's stream -- and the compositor's fill lane run against each other for the whole soak; SOAK_SLOTS=1: round 2's flow
# (two resident frames, explicit frame slots: one stream)
SLOTS = os.environ.get("SOAK_SLOTS") == "1"
if SLOTS:
    hip.select_frame_slot(1); hip.upload_frame(scene_b); hip.select_frame_slot(0)
else:
    raw = [[torch.from_numpy(np.ascontiguousarray(sc[k])).cuda() for k in ("depth", "quality", "silhouette", "color")] for sc in (scene, scene_b)]
    ptr = [[t.data_ptr() for t in r] for r in raw]
    torch.cuda.synchronize()
drv = mg.SlabDriver(hip, rank, world, "cuda:0", view=VIEW, halo="recompute", composite="compact", compositor="dedicated")
views = [rr.scene.default_view(*VIEW), (rr.scene.gl_flat(rr.scene.look_at((1.6, 1.4, 2.4), (0.0, 1.1, 0.0))), rr.scene.default_view(*VIEW)[1])]


def digest():
    h = hashlib.sha1()
    for a in hip.view_images()[:2] + hip.framebuffer():
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


first, bad, t0 = {}, 0, time.time()
for f in range(N):
    kind = ((f >> 1) & 1, (f >> 2) & 1)                   # (frame, view): A A B B ... so that each of a worker's two volume sets sees both frames
    if SLOTS:
        hip.select_frame_slot(kind[0])
        drv.frame(*views[kind[1]])
    else:
        drv.frame(*views[kind[1]], new_frame=ptr[kind[0]])
    if f % CHECK < 8 or f == N - 1:                        # eight consecutive frames = all four kinds, on either volume set
        drv.finish()
        if rank == 0:
            d = digest()
            ok = first.setdefault(kind, d) == d
            bad += not ok
            print(f"frame {f} kind {kind}: {d} {'ok' if ok else 'MISMATCH'}  ({time.time() - t0:.1f} s, regathers {drv.regathers})", flush=True)
drv.finish()
flag = torch.tensor([bad])
dist.broadcast(flag, src=0)
if rank == 0:
    print(f"soak {'ok' if bad == 0 else 'FAILED'}: {N} frames, {world - 1} slabs + compositor, {len(first)} kinds", flush=True)
dist.destroy_process_group()
sys.exit(1 if int(flag.item()) else 0)
