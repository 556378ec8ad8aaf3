"""-m gpu: what round 1 left untested (VERDICT r01 "What's missing" 2, "Next round" 3 / 9):

  * BASELINE.json configs[3] at its own size: 512^3 x 8 streams -- whole volume == oracle, 8 sequential Z-slabs == whole volume;
  * configs[4] at its own size: 1024^3 x 8 streams, dense storage and a sparse tile pool -- volume and frame == oracle (the
    oracle draws the voxel lists of the occupied bricks and leaves the clear value elsewhere), 8 slabs == whole volume;
  * NaN voxels (quality 0 on the first contributing stream: 0 / 0, tsdf_integration.vs:52) and with them NaN hit positions in
    the shading taps -- whole-volume, sparse and slab instantiations of the march;
  * a slab halo that has to be wider than one sampleDistance + footprint (ADVICE: limit/2 * res_z > 3.75 voxels);
  * the context's stream really orders its kernels with torch work (ADVICE: the NULL-stream handle);
  * asynchronous frame upload / frame slots == plain upload;
  * occupied fp32 sliver bricks (divideBox's extra last brick) and overlapping voxel lists over an incremental frame sequence.
"""
from importlib import import_module

import numpy as np
import pytest

from helpers import assert_same, tsdf_close
from oracle.oracle import OracleRecon

pytestmark = pytest.mark.gpu

VIEW = (1280, 720)


def same(a, b):
    return (a == b) | (np.isnan(a) & np.isnan(b))


def run_frame(o, mv, pr):
    o.clearOccupiedBricks(); o.markBricks(); o.updateOccupiedBricks(False)
    o.integrate()
    o.drawF(mv, pr)


def assert_frame_equal(a, b, what):
    (aa, ad, an, _), (ba, bd, bn, _) = a.view_images(), b.view_images()
    assert same(ad, bd).all(), f"{what}: raymarch depth"
    assert same(an, bn).all(), f"{what}: sample counts"
    assert same(aa, ba).all(), f"{what}: raymarch colour"
    (ac, adp), (bc, bdp) = a.framebuffer(), b.framebuffer()
    assert same(adp, bdp).all() and same(ac, bc).all(), f"{what}: framebuffer"
    return int((ad < 1).sum())


def frame_vs_oracle(hip, orc, mv, pr, what, peels=True):
    """the frame of both sides stage by stage: brick update + integrate + draw() -> raymarch target, then fillColors() -> framebuffer
    (the oracle's literal two-atlas fillColors() swaps its atlases, so its raymarch target is read before the hole filling)"""
    for o in (hip, orc):
        o.clearOccupiedBricks(); o.markBricks(); o.updateOccupiedBricks(); o.integrate(); o.draw(mv, pr)
    (aa, ad, an, ap), (ba, bd, bn, bp) = hip.view_images(), orc.view_images()
    assert not peels or same(ap, bp).all(), f"{what}: depth peels"
    assert same(an, bn).all(), f"{what}: sample counts"
    assert same(ad, bd).all() and same(aa, ba).all(), f"{what}: raymarch depth / colour"
    for o in (hip, orc):
        o.fillColors()
    (ac, adp), (bc, bdp) = hip.framebuffer(), orc.framebuffer()
    assert same(adp, bdp).all() and same(ac, bc).all(), f"{what}: framebuffer"
    return int((ad < 1).sum())


def big(rr, n_streams, res):
    scene = rr.scene.make_scene(n_streams=n_streams, width=640, height=480, lut_res=128, inv_res=128)
    ext = scene["bbox_max"] - scene["bbox_min"]
    kw = dict(res=res, brick_size=[float(ext[a]) / res[a] * 8 for a in range(3)], limit=0.01, view=VIEW)
    mv, pr = rr.scene.default_view(*VIEW)
    return scene, kw, mv, pr


def slabs_frame(rr, scene, kw, mv, pr, n, halo, composite, **extra):
    import torch  # noqa: F401
    mgpu = import_module("rgbd-recon_amd.multigpu")
    slabs = [rr.ReconIntegrationHip(scene, slab=mgpu.slab_range(kw["res"][2], k, n), recompute_halo=(halo == "recompute"), **kw, **extra) for k in range(n)]
    mgpu.frame_slabs_on_one_device(slabs, mv, pr, "cuda:0", halo=halo, composite=composite)
    return slabs


# ------------------------------------------------------------------------------------------------ configs[3]
@pytest.mark.timeout(1500)
def test_config3_512cubed_8_streams_whole_equals_oracle_and_8_slabs_equal_whole(rr):
    scene, kw, mv, pr = big(rr, 8, (512, 512, 512))
    hip, orc = rr.ReconIntegrationHip(scene, **kw), OracleRecon(scene, **kw)
    assert frame_vs_oracle(hip, orc, mv, pr, "c3 whole vs oracle") > 20000
    np.testing.assert_array_equal(hip.bricks()[0], orc.counters())
    whole_tsdf = hip.tsdf()
    assert same(whole_tsdf, orc.tsdf()).all()
    del orc
    for halo, composite in (("recompute", "compact"), ("exchange", "dense")):
        slabs = slabs_frame(rr, scene, kw, mv, pr, 8, halo, composite)
        assert_frame_equal(slabs[0], hip, f"c3 8 slabs ({halo}, {composite}) vs whole")
        if halo == "recompute":
            for k, s in enumerate(slabs):                                   # every rank's owned planes are the whole volume's
                z0, z1 = k * 64, (k + 1) * 64
                assert same(s.tsdf()[z0:z1], whole_tsdf[z0:z1]).all(), f"slab {k} volume"
        for s in slabs:
            s.close()


# ------------------------------------------------------------------------------------------------ configs[4]
@pytest.mark.timeout(2400)
def test_config4_1024cubed_8_streams_dense_and_sparse_equal_oracle_and_8_slabs_equal_whole(rr):
    scene, kw, mv, pr = big(rr, 8, (1024, 1024, 1024))
    hip, orc = rr.ReconIntegrationHip(scene, **kw), OracleRecon(scene, **kw)
    assert frame_vs_oracle(hip, orc, mv, pr, "c4 dense vs oracle") > 20000
    assert 0.0005 < hip.occupiedRatio() < 0.05
    np.testing.assert_array_equal(hip.bricks()[0], orc.counters())
    a, b = hip.tsdf(), orc.tsdf()                                           # 2^30 voxels each: inside the occupied bricks' voxel lists the
    assert a.shape == (1024, 1024, 1024)                                    # fused values, the clear value everywhere else (:249-258)
    for z in range(0, 1024, 128):                                           # compared in chunks: no 1 GiB temporaries
        assert same(a[z:z + 128], b[z:z + 128]).all(), f"planes {z}.."
    assert (a != np.float32(-0.01)).mean() > 1e-4
    del b
    del orc
    # the same volume in a sparse tile pool (BASELINE.json configs[4] "sparse-brick allocation")
    sp = rr.ReconIntegrationHip(scene, sparse_pool_tiles=1 << 17, **kw)
    run_frame(sp, mv, pr)
    need, cap = sp.sparse_pool_stats()
    assert 0 < need <= cap
    s = sp.tsdf()
    for z in range(0, 1024, 128):
        assert same(a[z:z + 128], s[z:z + 128]).all(), f"sparse planes {z}.."
    del s, a
    assert_frame_equal(sp, hip, "c4 sparse vs dense")
    sp.close()
    # 8 Z-slabs of 128 planes (+ 2 halo tile layers per face at limit * res_z = 10.24 voxels)
    slabs = slabs_frame(rr, scene, kw, mv, pr, 8, "recompute", "compact")
    assert slabs[0].halo_info()[0] == 2
    assert_frame_equal(slabs[0], hip, "c4 8 slabs vs whole")
    for s in slabs:
        s.close()
    slabs = slabs_frame(rr, scene, kw, mv, pr, 8, "recompute", "compact", sparse_pool_tiles=1 << 15)
    assert_frame_equal(slabs[0], hip, "c4 8 sparse slabs vs whole")


# ------------------------------------------------------------------------------------------------ NaN voxels / NaN hit positions
KW = dict(res=(64, 64, 64), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.04, view=(160, 90))


def test_nan_voxels_and_nan_hit_positions_in_every_march_instantiation(rr, small_scene):
    """quality 0 on the first stream that contributes to a voxel gives 0 / 0 (tsdf_integration.vs:52) and every later stream keeps
    the NaN: a NaN shell around the surface.  The march steps through it (NaN > 0 is false, prev_density becomes NaN), hits the
    +limit voxels behind, and refines to a NaN position whose gradient / colour taps must clamp like the oracle's: whole-volume
    (kWhole), sparse and slab kernels."""
    sc = dict(small_scene)
    q = sc["quality"].copy()
    q[0] = 0.0
    q[2, :, : q.shape[2] // 2] = 0.0
    sc["quality"] = q
    mv, pr = rr.scene.default_view(*KW["view"])
    orc = OracleRecon(sc, **KW)
    for use_bricks, skip in ((True, True), (False, False)):
        hips = {"whole": rr.ReconIntegrationHip(sc, **KW)}
        if use_bricks:
            hips["sparse"] = rr.ReconIntegrationHip(sc, sparse_pool_tiles=4096, **KW)
        for o in [orc] + list(hips.values()):
            o.setUseBricks(use_bricks); o.setSpaceSkip(skip)
        for name, h in hips.items():
            assert frame_vs_oracle(h, orc, mv, pr, f"{name}, bricks={use_bricks}", peels=skip) > 300
            t = orc.tsdf()
            assert_same(h.tsdf(), t, f"tsdf ({name}, bricks={use_bricks})")
        assert np.isnan(t).sum() > 500                                       # the NaN shell exists
        orc.draw(mv, pr)
        (oc, od, on, _) = orc.view_images()
        assert np.isnan(oc[od < 1]).any()                                    # ... and rays were shaded at NaN positions
        if use_bricks:
            run_frame(hips["whole"], mv, pr)
            for halo, composite in (("recompute", "compact"), ("exchange", "dense")):
                slabs = slabs_frame(rr, sc, KW, mv, pr, 4, halo, composite)
                assert_frame_equal(slabs[0], hips["whole"], f"4 slabs ({halo})")


# ------------------------------------------------------------------------------------------------ halo wider than one step
@pytest.mark.parametrize("halo,composite", [("recompute", "compact"), ("exchange", "dense")])
def test_slab_halo_covers_refined_hit_gradient_taps(rr, small_scene, halo, composite):
    """limit/2 * res_z = 5.12 voxels (configs[4]'s ratio): the gradient taps at the REFINED hit position reach
    ceil(2 * 5.12 + 0.5) = 11 planes past a slab face, so 2 halo tile layers are needed (one sampleDistance + footprint = 1
    layer was the round-1 sizing).  Views along z, where the reach is largest."""
    kw = dict(res=(64, 64, 256), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 32], limit=0.04, view=(160, 90))
    pr = rr.scene.gl_flat(rr.scene.perspective(50.0, 160 / 90.0, 0.1, 200.0))
    for eye in ((0.0, 1.1, 3.0), (0.0, 1.1, -3.0), (0.6, 1.4, 2.8)):
        mv = rr.scene.gl_flat(rr.scene.look_at(eye, (0.0, 1.1, 0.0)))
        whole = rr.ReconIntegrationHip(small_scene, **kw)
        run_frame(whole, mv, pr)
        slabs = slabs_frame(rr, small_scene, kw, mv, pr, 4, halo, composite)
        assert slabs[1].halo_info()[0] == 2
        assert assert_frame_equal(slabs[0], whole, f"eye {eye}") > 300


# ------------------------------------------------------------------------------------------------ stream order
def test_context_kernels_are_ordered_with_torch_work_on_the_adopted_stream(rr, small_scene):
    """set_stream(torch stream) must put the context's kernels ON that stream: a torch copy issued right after an export, with no
    synchronisation in between, has to see the exported image -- for torch's default stream (handle 0 = the NULL stream,
    tsdf_adopt_null_stream) as well as for an explicit one."""
    import torch
    kw = dict(res=(256, 256, 256), brick_size=[2.0 / 32, 2.2 / 32, 2.0 / 32], limit=0.01, view=(640, 360))
    mv, pr = rr.scene.default_view(640, 360)
    npx = 640 * 360
    for which in ("default", "explicit"):
        s = torch.cuda.default_stream() if which == "default" else torch.cuda.Stream()
        assert (s.cuda_stream == 0) == (which == "default")
        hip = rr.ReconIntegrationHip(small_scene, **kw)
        hip.setUseBricks(False); hip.setSpaceSkip(False); hip.setColorFilling(False)
        hip.set_stream(s.cuda_stream)
        with torch.cuda.stream(s):
            buf = torch.full((npx * 6,), -7.0, dtype=torch.float32, device="cuda:0")
            for _ in range(3):
                hip.integrate(); hip.draw(mv, pr)                            # ~1 ms of dense kernels queued ...
            hip.export_partial_dev(buf.data_ptr())                           # ... then the export, then torch reads at once
            got = buf.clone()
            s.synchronize()
        assert not bool((got == -7.0).any()), f"{which} stream: torch read the buffer before the context's kernels had written it"
        assert bool((got == buf).all())
        hip.close()


# ------------------------------------------------------------------------------------------------ async upload / frame slots
def test_async_upload_and_frame_slots_equal_plain_upload(rr):
    kw = dict(n_streams=3, width=128, height=96, lut_res=24, inv_res=32)
    scenes = [rr.scene.make_scene(**kw), rr.scene.make_scene(sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2), **kw),
              rr.scene.make_scene(sphere_c=(-0.3, 1.3, 0.4), **kw)]
    mv, pr = rr.scene.default_view(*KW["view"])
    ref = []
    for sc in scenes:
        h = rr.ReconIntegrationHip(sc, **KW)
        run_frame(h, mv, pr)
        ref.append((h.tsdf(), h.framebuffer()))
    one = rr.ReconIntegrationHip(scenes[0], **KW)
    order = [1, 2, 0, 2, 1, 0, 1]
    one.upload_frame_async(scenes[order[0]])
    for i, k in enumerate(order):
        one.select_frame_slot(one.current_frame_slot() ^ 1)                  # frame k becomes current (GPU-side wait on its upload)
        if i + 1 < len(order):
            if i % 2:                                                        # the producer fills the pinned staging in place ...
                d, q, s, c = one.frame_staging()
                nx = scenes[order[i + 1]]
                d[...] = nx["depth"]; q[...] = nx["quality"]; s[...] = nx["silhouette"]; c[...] = nx["color"]
                one.upload_frame_async(None)
            else:                                                            # ... or hands over its own buffers
                one.upload_frame_async(scenes[order[i + 1]])
        run_frame(one, mv, pr)                                               # computes frame k while frame k + 1 travels
        assert_same(one.tsdf(), ref[k][0], f"step {i} tsdf")
        fc, fd = one.framebuffer()
        assert_same(fd, ref[k][1][1], f"step {i} depth"); assert_same(fc, ref[k][1][0], f"step {i} colour")
    # two resident frames, switched without any upload
    two = rr.ReconIntegrationHip(scenes[0], **KW)
    two.select_frame_slot(1); two.upload_frame(scenes[1]); two.select_frame_slot(0)
    for i in range(5):
        two.select_frame_slot(i & 1)
        run_frame(two, mv, pr)
        assert_same(two.tsdf(), ref[i & 1][0], f"slot switch {i}")


# ------------------------------------------------------------------------------------------------ occupied sliver brick
def test_occupied_sliver_bricks_over_an_incremental_sequence(rr):
    """72 voxels along y in bricks of 4: divideBox()'s fp32 loop (recon_integration.cpp:366-372) yields 19 bricks, not 18 -- a
    last sliver brick that shares voxel plane 71 with its neighbour -- and several interior bricks overlap by one plane
    (containedVoxels' float bounds, volume_sampler.cpp:50-62).  The sliver layer's counters are forced above the threshold, a
    different subset every frame; culled frames -- the first (full classification) and the following (list-based
    classification: brick -> tile scatter against the per-tile brick spans) -- must equal the oracle."""
    kw = dict(KW, res=(64, 72, 64), brick_size=[2.0 / 16, 2.2 / 18, 2.0 / 16])
    scene = rr.scene.make_scene(n_streams=4, width=160, height=120, lut_res=32, inv_res=32, sphere_c=(0.0, 1.9, 0.0))   # the sphere crosses the top of the box
    hip, orc = rr.ReconIntegrationHip(scene, **kw), OracleRecon(scene, **kw)
    assert hip.res_bricks == (16, 19, 16)
    ranges = orc.brick_ranges().reshape(16, 19, 16, 6)
    assert (ranges[:, 18, :, 1] == 71).all() and (ranges[:, 18, :, 4] == 72).all()      # the sliver layer: voxel plane 71 only
    assert (ranges[:, 17, :, 4] == 72).all()                                               # ... which brick 17 lists as well
    ids = np.arange(16 * 19 * 16).reshape(16, 19, 16)
    sliver = ids[:, 18, :].reshape(-1)
    mv, pr = rr.scene.default_view(*KW["view"])
    for f in range(4):
        for o in (hip, orc):
            o.clearOccupiedBricks(); o.markBricks()
        cnt = orc.counters().copy()
        np.testing.assert_array_equal(hip.bricks()[0], cnt)
        cnt[sliver[f::3]] = 1000                                             # a different set of sliver bricks "occupied" each frame
        for o in (hip, orc):
            o.set_counters(cnt)
            o.updateOccupiedBricks(); o.integrate(); o.draw(mv, pr)
        a = hip.tsdf()
        assert_same(a, orc.tsdf(), f"frame {f} tsdf")
        assert (a[:, 60:70, :] != np.float32(-0.04)).sum() > 1000            # the sphere's surface band lies in the overlapping bricks below
        (aa, ad, an, ap), (ba, bd, bn, bp) = hip.view_images(), orc.view_images()
        assert same(ap, bp).all() and same(an, bn).all() and same(ad, bd).all() and same(aa, ba).all(), f"frame {f} raymarch"
        for o in (hip, orc):
            o.fillColors()
        (ac, adp), (bc, bdp) = hip.framebuffer(), orc.framebuffer()
        assert same(adp, bdp).all() and same(ac, bc).all(), f"frame {f} framebuffer"


# ------------------------------------------------------------------------------------------------ dense march through LDS boxes
def _dense(o):
    o.setUseBricks(False); o.setSpaceSkip(False); o.setColorFilling(False)


@pytest.mark.parametrize("res", [(64, 64, 64), (40, 56, 72), (37, 61, 50)])
def test_box_march_equals_gather_march_and_oracle(rr, small_scene, res, monkeypatch):
    """k_march_box (voxel boxes in LDS, all-clear boxes, tile-class leaps) against the round-1 gather march (RR_MARCH_BOX=0) and the
    oracle: eyes far away, grazing, INSIDE the volume (rays of one 8x8 tile diverge: boxes that do not fit, the global path) and on
    an axis; volumes that are not multiples of the 8-voxel tile; a view that is not a multiple of the 8x8 ray tile."""
    kw = dict(res=res, brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.04, view=(173, 99))
    pr = rr.scene.gl_flat(rr.scene.perspective(50.0, 173 / 99.0, 0.1, 200.0))
    box = rr.ReconIntegrationHip(small_scene, **kw)
    monkeypatch.setenv("RR_MARCH_BOX", "0")
    gather = rr.ReconIntegrationHip(small_scene, **kw)
    monkeypatch.setenv("RR_MARCH_BOX", "2")                  # (round 4) two boxes per wave, the next batch's box prefetched: opt-in, measured slower, kept bit-identical
    box2 = rr.ReconIntegrationHip(small_scene, **kw)
    monkeypatch.delenv("RR_MARCH_BOX")
    orc = OracleRecon(small_scene, **kw)
    for o in (box, gather, box2, orc):
        _dense(o)
        o.integrate()
    for eye, at in (((0.0, 1.1, 3.0), (0.0, 1.1, 0.0)), ((2.6, 2.0, 2.2), (0.0, 1.0, 0.0)), ((0.2, 1.0, 0.3), (0.0, 1.1, -1.0)),
                    ((0.05, 1.1, 0.05), (0.9, 1.3, 0.7)), ((0.0, 5.0, 0.001), (0.0, 0.0, 0.0)), ((-3.5, 1.1, 0.0), (0.0, 1.1, 0.0))):
        mv = rr.scene.gl_flat(rr.scene.look_at(eye, at))
        for o in (box, gather, box2, orc):
            o.draw(mv, pr)
        (ba, bd, bn, _), (ga, gd, gn, _), (oa, od, on, _) = box.view_images(), gather.view_images(), orc.view_images()
        (pa, pd, pn, _) = box2.view_images()
        assert same(bn, pn).all() and same(bd, pd).all() and same(ba, pa).all(), f"box vs prefetched boxes, eye {eye}"
        assert same(bn, gn).all() and same(bd, gd).all() and same(ba, ga).all(), f"box vs gather, eye {eye}"
        assert same(bn, on).all() and same(bd, od).all() and same(ba, oa).all(), f"box vs oracle, eye {eye}"
        assert (bn > 0).sum() > 500


def test_box_march_on_uploaded_volumes(rr, small_scene):
    """tsdf_upload_volume leaves every tile class 'mixed' (no leaps) -- and the classes must come back exact with the next integrate.
    Random densities, exact -limit regions, NaN and infinite voxels."""
    kw = dict(res=(48, 40, 56), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.04, view=(160, 90))
    mv, pr = rr.scene.default_view(160, 90)
    hip, orc = rr.ReconIntegrationHip(small_scene, **kw), OracleRecon(small_scene, **kw)
    for o in (hip, orc):
        _dense(o)
    rng = np.random.default_rng(3)
    lim = np.float32(0.04)
    for case in range(4):
        v = np.full((56, 40, 48), -lim, np.float32)
        if case == 0:
            v[20:40, 10:30, 12:36] = rng.uniform(-0.04, 0.04, (20, 20, 24)).astype(np.float32)
        elif case == 1:
            v[:] = rng.uniform(-0.04, 0.001, v.shape).astype(np.float32)
        elif case == 2:
            v[24:32] = np.nan; v[40:44, :, :] = np.float32(0.02); v[10, 10, 10] = np.inf
        else:
            v[:] = -lim                                                        # nothing to hit: every ray runs its full length
        for o in (hip, orc):
            o.set_tsdf(v)
            o.draw(mv, pr)
        (ha, hd, hn, _), (oa, od, on, _) = hip.view_images(), orc.view_images()
        assert same(hn, on).all() and same(hd, od).all() and same(ha, oa).all(), f"uploaded volume case {case}"
    for o in (hip, orc):                                                       # back to integrated volumes: classes exact again, leaps on
        o.integrate(); o.draw(mv, pr)
    (ha, hd, hn, _), (oa, od, on, _) = hip.view_images(), orc.view_images()
    assert same(hn, on).all() and same(hd, od).all() and same(ha, oa).all() and (hd < 1).sum() > 300


# ------------------------------------------------------------------------------------------------ uniform (tile, stream) pairs in the dense integrate
@pytest.mark.parametrize("form", ["2", "3"])
def test_dense_integrate_uniform_pair_shortcut_equals_per_voxel_evaluation(rr, monkeypatch, form):
    """The dense integrate proves per tile and stream, from the LUT box bounds and the per-cell depth / silhouette ranges of the
    frame, that every voxel takes the same branch of the fusion rule, and then skips the per-voxel gathers.  Against the same kernel
    with the shortcut off (RR_K1_RANGES=0) and against the oracle: plain frames, a frame with NaN / infinite / out-of-range LUT
    texels, NaN and negative depths, silhouettes that are neither 0 nor 1, zero quality (NaN voxels), and the raw-frame path."""
    monkeypatch.setenv("RR_K1_FORM", form)      # 2: the LDS form; 3: + the opt-in projection cache (the second integrate() of a context reads the pool)
    if form == "3":
        monkeypatch.setenv("RR_PROJ_CACHE_MB", "64")
    base = rr.scene.make_scene(n_streams=4, width=160, height=120, lut_res=32, inv_res=32)
    rng = np.random.default_rng(11)
    scenes = [base, rr.scene.make_scene(n_streams=4, width=160, height=120, lut_res=32, inv_res=32, sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2))]
    odd = dict(base)
    inv = odd["cv_xyz_inv"].copy()
    n = inv.shape[1]
    for i, vals in enumerate(([np.nan, 0.3, 0.5], [np.inf, -np.inf, 0.4], [7.5, -3.25, 0.45], [0.5, 0.5, -2.0e5])):
        inv[i, rng.choice(n, n // 60, replace=False), :3] = np.array(vals, np.float32)
    odd["cv_xyz_inv"] = inv
    d = odd["depth"].copy(); q = odd["quality"].copy(); s = odd["silhouette"].copy()
    d[0, 10:14, 20:40, 0] = np.nan; d[1, 50:60, 70:90, 0] = -0.3; d[2, 80:82, :, 0] = np.inf
    s[1, 30:50, 30:50] = 0.5; s[3, 60:70, 10:30] = np.nan; s[2, 5:9, 100:140] = 2.0
    q[0] = 0.0
    odd["depth"], odd["quality"], odd["silhouette"] = d, q, s
    scenes.append(odd)
    for res in ((64, 64, 64), (40, 56, 72), (128, 128, 128)):
        kw = dict(res=res, brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.04 if res[0] < 100 else 0.02, view=(64, 36))
        for k, sc in enumerate(scenes):
            fast = rr.ReconIntegrationHip(sc, **kw)
            monkeypatch.setenv("RR_K1_RANGES", "0")
            slow = rr.ReconIntegrationHip(sc, **kw)
            monkeypatch.delenv("RR_K1_RANGES")
            orc = OracleRecon(sc, **kw)
            for o in (fast, slow, orc):
                o.setUseBricks(False)
                o.integrate()
            fast.integrate()
            a, b, c = fast.tsdf(), slow.tsdf(), orc.tsdf()
            assert same(a, b).all(), f"res {res} scene {k}: shortcut vs per-voxel"
            assert same(a, c).all(), f"res {res} scene {k}: shortcut vs oracle"
            assert (a == np.float32(-kw["limit"])).mean() > 0.3 and (np.abs(a) < kw["limit"]).sum() > 100
    # the pre-processing path produces the packed image (and its ranges) on the device
    kw = dict(res=(64, 64, 64), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.04, view=(64, 36))
    hip, orc = rr.ReconIntegrationHip(base, **kw), OracleRecon(base, **kw)
    hip.upload_raw_frame(base); orc.upload_raw_frame(base)
    for o in (hip, orc):
        o.setUseBricks(False)
        o.clearOccupiedBricks(); o.processTextures(); o.updateOccupiedBricks(); o.integrate()
    assert tsdf_close(hip.tsdf(), orc.tsdf(), 0.04).all()


# ------------------------------------------------------------------------------------------------ setVoxelSize
def test_set_voxel_size_recreates_the_volume_like_the_reference(rr, small_scene):
    """setVoxelSize(), recon_integration.cpp:340-353: resolution = ceil(bbox / size), a fresh volume, and the brick grid re-snapped from
    the CURRENT (already snapped) brick size.  A sequence of sizes through one context and one oracle, culled and dense, each frame
    against the oracle and against a context constructed with that size."""
    kw = dict(voxel_size=0.05, brick_size=0.22, limit=0.04, view=(160, 90))
    hip, orc = rr.ReconIntegrationHip(small_scene, **kw), OracleRecon(small_scene, **kw)
    mv, pr = rr.scene.default_view(160, 90)
    assert hip.res == orc.res == (40, 44, 40)
    for size, dense in ((0.05, False), (0.031, False), (0.04, True), (0.07, False), (0.025, True)):
        for o in (hip, orc):
            o.setVoxelSize(size)
            o.setUseBricks(not dense); o.setSpaceSkip(not dense)
        assert hip.res == orc.res and hip.res_bricks == orc.res_bricks, (size, hip.res, orc.res, hip.res_bricks, orc.res_bricks)
        assert np.allclose(hip.brick_size, orc.brick_size, rtol=0, atol=0)
        assert frame_vs_oracle(hip, orc, mv, pr, f"voxel size {size}", peels=not dense) > 300
        assert_same(hip.tsdf(), orc.tsdf(), f"voxel size {size} tsdf")
    fresh = rr.ReconIntegrationHip(small_scene, **dict(kw, voxel_size=0.025))
    assert fresh.res == hip.res                                               # (the brick grids differ: 0.22 snapped once vs through the sequence)
    with pytest.raises(rr.TsdfError):
        hip.setVoxelSize(0.0)
    mgpu = import_module("rgbd-recon_amd.multigpu")
    slab = rr.ReconIntegrationHip(small_scene, res=(64, 64, 64), brick_size=0.25, limit=0.04, view=(160, 90), slab=mgpu.slab_range(64, 0, 2))
    with pytest.raises(rr.TsdfError):
        slab.setVoxelSize(0.05)


# ------------------------------------------------------------------------------------------------ compact export of a two-pass march
def test_compact_export_of_a_whole_context_ships_the_long_ray_pass_too(rr, small_scene):
    """tsdf_export_hits_dev on a whole-volume context: the rays its second (wave-per-ray) march pass finished are not on the hit list,
    so they are exported from the long-ray list -- every one, a miss as (clear colour, depth 1, its sample count).  Composited into a
    second context that never marched, the records reproduce the frame: colour and depth everywhere, the sample counts wherever a
    record landed.  The same with the second pass switched off (tsdf_set_march_cap(0): everything on the hit list) and with a
    capacity that is too small (overflow flag, no write past the capacity)."""
    import torch
    kw = dict(res=(64, 64, 64), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.04, view=(320, 180))
    mv, pr = rr.scene.default_view(*kw["view"])
    npx = kw["view"][0] * kw["view"][1]
    src, dst = rr.ReconIntegrationHip(small_scene, **kw), rr.ReconIntegrationHip(small_scene, **kw)
    for cap in (2, 24, 0):                                                  # 2: almost every ray goes to the second pass
        src.setMarchCap(cap)
        src.clearOccupiedBricks(); src.markBricks(); src.updateOccupiedBricks(); src.integrate(); src.draw(mv, pr)
        buf = torch.zeros(8 + npx * 8, dtype=torch.float32, device="cuda:0")
        src.export_hits_dev(buf.data_ptr(), npx); src.sync()
        hdr = buf[:8].cpu().numpy().view(np.uint32)
        assert hdr[0] == hdr[1] and hdr[2] == 0 and 300 < hdr[0] <= npx
        dst.composite_hits_dev(buf.data_ptr(), 1, buf.numel() * 4); dst.sync()
        (sa, sd, sn, _), (da, dd, dn, _) = src.view_images(), dst.view_images()
        assert same(sd, dd).all() and same(sa, da).all() and int((sd < 1).sum()) > 300, f"cap {cap}"
        rec = buf[8:8 + int(hdr[0]) * 8].cpu().numpy().reshape(-1, 8)
        pix = rec[:, 0].copy().view(np.uint32)
        assert np.unique(pix).size == pix.size                                # a ray is on one list only
        assert same(sn.reshape(-1)[pix], dn.reshape(-1)[pix]).all() and (dn.reshape(-1)[np.setdiff1d(np.arange(npx), pix)] == 0).all()
        if cap == 2:
            first = hdr[0]
        # too small a capacity: exactly `capacity` records, the overflow flag, nothing written behind them
        small = torch.full((8 + 64 * 8 + 8,), -7.0, dtype=torch.float32, device="cuda:0")
        src.export_hits_dev(small.data_ptr(), 64); src.sync()
        h2 = small[:8].cpu().numpy().view(np.uint32)
        assert h2[0] == 64 and h2[1] == hdr[1] and h2[2] == 1 and bool((small[8 + 64 * 8:] == -7.0).all())
    assert first > 0


def test_culled_integrate_uniform_pair_shortcut_equals_per_voxel_evaluation(rr, monkeypatch):
    """The same shortcut in the culled launch (dense storage): per active tile and stream, from the static LUT-box bounds and the frame's
    per-cell ranges.  Against the launch without it (RR_K1_CULLED_RANGES=0) and the oracle, on the plain frames and on the frame with
    NaN / infinite / out-of-range LUT texels, NaN and negative depths and silhouettes that are neither 0 nor 1; slab contexts too."""
    base = rr.scene.make_scene(n_streams=4, width=160, height=120, lut_res=32, inv_res=32)
    rng = np.random.default_rng(12)
    odd = dict(base)
    inv = odd["cv_xyz_inv"].copy()
    n = inv.shape[1]
    for i, vals in enumerate(([np.nan, 0.3, 0.5], [np.inf, -np.inf, 0.4], [7.5, -3.25, 0.45], [0.5, 0.5, -2.0e5])):
        inv[i, rng.choice(n, n // 60, replace=False), :3] = np.array(vals, np.float32)
    odd["cv_xyz_inv"] = inv
    d = odd["depth"].copy(); q = odd["quality"].copy(); s = odd["silhouette"].copy()
    d[0, 10:14, 20:40, 0] = np.nan; d[1, 50:60, 70:90, 0] = -0.3; d[2, 80:82, :, 0] = np.inf
    s[1, 30:50, 30:50] = 0.5; s[3, 60:70, 10:30] = np.nan; s[2, 5:9, 100:140] = 2.0
    odd["depth"], odd["quality"], odd["silhouette"] = d, q, s
    moved = rr.scene.make_scene(n_streams=4, width=160, height=120, lut_res=32, inv_res=32, sphere_c=(0.4, 0.7, -0.3), box_c=(-0.5, 1.5, 0.2))
    for res in ((64, 64, 64), (128, 128, 128)):
        kw = dict(res=res, brick_size=[2.0 / res[0] * 8, 2.2 / res[1] * 8, 2.0 / res[2] * 8], limit=0.04 if res[0] < 100 else 0.02, view=(64, 36))
        for k, sc in enumerate((base, moved, odd)):
            fast = rr.ReconIntegrationHip(sc, **kw)
            monkeypatch.setenv("RR_K1_CULLED_RANGES", "0")
            slow = rr.ReconIntegrationHip(sc, **kw)
            monkeypatch.delenv("RR_K1_CULLED_RANGES")
            orc = OracleRecon(sc, **kw)
            for o in (fast, slow, orc):
                o.clearOccupiedBricks(); o.markBricks(); o.updateOccupiedBricks(); o.integrate()
            a, b, c = fast.tsdf(), slow.tsdf(), orc.tsdf()
            assert same(a, b).all(), f"res {res} scene {k}: shortcut vs per-voxel"
            assert same(a, c).all(), f"res {res} scene {k}: shortcut vs oracle"
            assert (np.abs(a) < kw["limit"]).sum() > 100
    mgpu = import_module("rgbd-recon_amd.multigpu")
    kw = dict(res=(64, 64, 64), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.04, view=(64, 36))
    whole = rr.ReconIntegrationHip(odd, **kw)
    whole.clearOccupiedBricks(); whole.markBricks(); whole.updateOccupiedBricks(); whole.integrate()
    w = whole.tsdf().reshape(64, 64, 64)
    for r in range(3):
        z0, z1 = mgpu.slab_range(64, r, 3)
        sl = rr.ReconIntegrationHip(odd, slab=(z0, z1), recompute_halo=True, **kw)
        sl.clearOccupiedBricks(); sl.markBricks(); sl.updateOccupiedBricks(); sl.integrate()
        assert same(sl.tsdf().reshape(64, 64, 64)[z0:z1], w[z0:z1]).all(), f"slab {r}"
