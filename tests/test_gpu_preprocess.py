"""-m gpu: the HIP pre-processing passes (tsdf_process_textures) against the oracle's restatement of
glsl/pre_{morph,depth,boundary,normal,quality}.fs on the same raw frame, and the whole raw-frame pipeline.

Tolerances: everything without pow() is bit-exact (same fp32 operations in the same order: depth2, filtered depth + range
quality, boundary depth / silhouette, normals, brick counters); Lab colour and quality go through powf (device vs glibc):
Lab 1e-6 abs, quality 1e-5 relative."""
import numpy as np
import pytest

from helpers import tsdf_close
from oracle.oracle import OracleRecon

pytestmark = pytest.mark.gpu

KW = dict(res=(64, 64, 64), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.04, view=(160, 90))


def same(a, b):
    return (a == b) | (np.isnan(a) & np.isnan(b))


@pytest.mark.parametrize("flags", [dict(), dict(filter_textures=False), dict(processed_depth=False, refine=False)])
def test_passes_match_oracle(rr, small_scene, flags):
    hip, orc = rr.ReconIntegrationHip(small_scene, **KW), OracleRecon(small_scene, **KW)
    for o in (hip, orc):
        o.upload_raw_frame(small_scene)
        o.setPreprocess(**flags)
        o.clearOccupiedBricks()
        o.processTextures()
    a, b = hip.preprocessed(), orc.preprocessed()
    for k in ("depth2", "depth_rg", "depth_b", "silhouette", "normals"):
        assert same(a[k], b[k]).all(), f"{k}: {(~same(a[k], b[k])).sum()} of {a[k].size} differ"
    assert np.abs(a["lab"] - b["lab"]).max() <= 1e-6
    with np.errstate(invalid="ignore"):
        ok = (np.abs(a["quality"] - b["quality"]) <= 1e-5 * np.maximum(np.abs(b["quality"]), 1e-3)) | (np.isnan(a["quality"]) & np.isnan(b["quality"]))
    assert ok.all()
    np.testing.assert_array_equal(hip.bricks()[0], orc.counters())          # mark_brick inside the normal pass
    assert (b["silhouette"] > 0).sum() > 500 and ((small_scene["depth_raw"] == 0) & (b["depth2"] > 0)).sum() > 50


def test_raw_frame_pipeline_matches_oracle(rr, small_scene):
    """process_textures() -> integrate() -> drawF() from the RAW frame (source/kinect_client.cpp:569-599,614)."""
    hip, orc = rr.ReconIntegrationHip(small_scene, upload=False, **KW), OracleRecon(small_scene, **KW)
    hip.set_calibration(small_scene)
    mv, pr = rr.scene.default_view(*KW["view"])
    ratios = []
    for o in (hip, orc):
        o.upload_raw_frame(small_scene)
        o.clearOccupiedBricks()
        o.processTextures()
        ratios.append(o.updateOccupiedBricks())
        o.integrate()
        o.drawF(mv, pr)
    assert ratios[0] == ratios[1] > 0
    assert tsdf_close(hip.tsdf(), orc.tsdf(), KW["limit"]).all()
    (fc, fd), (gc, gd) = hip.framebuffer(), orc.framebuffer()
    assert ((fd < 1) != (gd < 1)).mean() <= 2e-3
    both = (fd < 1) & (gd < 1)
    assert both.sum() > 200 and (np.abs(fd[both] - gd[both]) > 1e-4).mean() <= 2e-3
    with np.errstate(invalid="ignore"):
        assert (np.abs(fc[both] - gc[both]) > 2e-3).mean() <= 1e-2


def test_process_textures_needs_its_inputs(rr, small_scene):
    hip = rr.ReconIntegrationHip(small_scene, **KW)
    with pytest.raises(rr.TsdfError) as e:
        hip.processTextures()
    assert e.value.code == -4
