"""-m gpu: the sparse tile pool (tsdf_config::sparse_pool_tiles, BASELINE.json configs[4] "sparse-brick allocation") against
dense storage: identical volumes and frames over a sequence of different frames (slots are re-assigned every frame), pool
exhaustion degrades to missing tiles instead of faults, slabs work with recomputed halos, and a 2048^3 volume -- 32 GiB dense
-- runs in a 0.6 GiB pool."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KW = dict(res=(64, 64, 64), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.04, view=(160, 90))


def frame(o, mv, pr):
    o.clearOccupiedBricks(); o.markBricks(); o.updateOccupiedBricks(False); o.integrate(); o.drawF(mv, pr)


def same(a, b):
    return ((a == b) | (np.isnan(a) & np.isnan(b))).all()


def test_sparse_pool_equals_dense_over_a_frame_sequence(rr):
    kw = dict(n_streams=3, width=128, height=96, lut_res=24, inv_res=32)
    frames = [rr.scene.make_scene(**kw), rr.scene.make_scene(sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2), **kw),
              rr.scene.make_scene(sphere_c=(-0.45, 1.5, 0.4), box_c=(0.2, 0.3, -0.6), **kw)]
    dense, sparse = rr.ReconIntegrationHip(frames[0], **KW), rr.ReconIntegrationHip(frames[0], sparse_pool_tiles=512, **KW)
    mv, pr = rr.scene.default_view(*KW["view"])
    used = []
    for sc in frames + [frames[0]]:
        for o in (dense, sparse):
            o.upload_frame(sc)
            frame(o, mv, pr)
        assert same(dense.tsdf(), sparse.tsdf())
        for a, b in zip(dense.view_images()[:3] + dense.framebuffer(), sparse.view_images()[:3] + sparse.framebuffer()):
            assert same(a, b)
        need, cap = sparse.sparse_pool_stats()
        assert 0 < need <= cap == 512
        used.append(need)
    assert len(set(used)) > 1 and used[0] == used[3]


def test_pool_exhaustion_drops_tiles_instead_of_faulting(rr, small_scene):
    dense, tiny = rr.ReconIntegrationHip(small_scene, **KW), rr.ReconIntegrationHip(small_scene, sparse_pool_tiles=16, **KW)
    mv, pr = rr.scene.default_view(*KW["view"])
    frame(dense, mv, pr); frame(tiny, mv, pr)
    need, cap = tiny.sparse_pool_stats()
    assert need > cap == 16
    a, b = dense.tsdf(), tiny.tsdf()
    kept = (a == b) | (np.isnan(a) & np.isnan(b))
    assert 0.5 < kept.mean() < 1.0 and (b[~kept] == np.float32(-0.04)).all()      # a dropped tile reads as the clear value
    with pytest.raises(rr.TsdfError):
        tiny.setUseBricks(False); tiny.integrate()


def test_sparse_slabs_with_recomputed_halo(rr, small_scene):
    import torch  # noqa: F401
    from importlib import import_module
    mgpu = import_module("rgbd-recon_amd.multigpu")
    mv, pr = rr.scene.default_view(*KW["view"])
    whole = rr.ReconIntegrationHip(small_scene, **KW)
    frame(whole, mv, pr)
    slabs = [rr.ReconIntegrationHip(small_scene, slab=mgpu.slab_range(64, k, 2), recompute_halo=True, sparse_pool_tiles=512, **KW) for k in range(2)]
    mgpu.frame_slabs_on_one_device(slabs, mv, pr, "cuda:0", halo="recompute", composite="compact")
    for a, b in zip(whole.framebuffer(), slabs[0].framebuffer()):
        assert same(a, b)
    with pytest.raises(rr.TsdfError):
        rr.ReconIntegrationHip(small_scene, slab=mgpu.slab_range(64, 0, 2), recompute_halo=False, sparse_pool_tiles=512, **KW)


def test_2048_cubed_in_a_sparse_pool(rr):
    """2048^3 = 32 GiB of dense voxels.  Same world-space bricks and truncation as the 512^3 bench configuration, so the
    occupied region is the same 1.5 % of the volume: ~260 k tiles, 0.5 GiB of pool."""
    scene = rr.scene.make_scene(n_streams=4, width=640, height=480, lut_res=128, inv_res=128)
    ext = scene["bbox_max"] - scene["bbox_min"]
    view = (640, 360)
    mv, pr = rr.scene.default_view(*view)
    brick = [float(ext[a]) / 512 * 8 for a in range(3)]

    def run(res, pool):
        o = rr.ReconIntegrationHip(scene, res=(res,) * 3, brick_size=brick, limit=0.01, view=view, sparse_pool_tiles=pool)
        frame(o, mv, pr)
        return o

    big = run(2048, 300000)
    need, cap = big.sparse_pool_stats()
    assert 100000 < need <= cap
    fc, fd = big.framebuffer()
    ref = run(512, 8192)
    assert ref.sparse_pool_stats()[0] <= 8192
    gc, gd = ref.framebuffer()
    # the same continuous field sampled 4x finer: the same picture up to a voxel of the coarse volume
    assert ((fd < 1) != (gd < 1)).mean() < 0.01
    both = (fd < 1) & (gd < 1)
    assert both.sum() > 10000 and np.median(np.abs(fd[both] - gd[both])) < 1e-3
    print(f"\n[sparse] 2048^3: {need} of {cap} pool tiles in use ({need * 2048 / 2**30:.2f} GiB of voxels instead of 32 GiB)")
