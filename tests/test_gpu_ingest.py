"""-m gpu: frame ingest (tsdf_upload_wire_frame) against the oracle (whose DXT decode is pinned against the reference's
squish, tests/test_oracle_ingest.py).  The unpack is byte/integer work: bit-exact.  The 8-bit depth path is then followed
through tsdf_process_textures with the tolerances of tests/test_gpu_preprocess.py."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from oracle.oracle import OracleRecon

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "oracle", "_ref", "ref_wire_tool")
KW = dict(res=(64, 64, 64), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.04, view=(160, 90))
FORMATS = [(0, 0), (1, 0), (5, 0), (0, 1), (1, 1), (5, 1)]


def same(a, b):
    return (a == b) | (np.isnan(a) & np.isnan(b))


@pytest.mark.parametrize("cf,df", FORMATS)
def test_unpack_is_bit_exact(rr, small_scene, cf, df):
    msg = rr.scene.make_wire_message(small_scene, cf, df, timestamp=42.125)
    hip = rr.ReconIntegrationHip(small_scene, **KW)
    hip.setWireFormat(cf, df)
    n, w, h = small_scene["n"], small_scene["width"], small_scene["height"]
    cs, ds = orc.wire_sizes(w, h, w, h, cf, df)
    assert hip.wireSizes() == (cs, ds, (cs + ds) * n) and len(msg) == (cs + ds) * n
    assert hip.upload_wire_frame(msg) == 42.125
    depth, col = hip.raw_frame()
    ts, cols, deps = orc.wire_split(msg, n, cs, ds)
    for i in range(n):
        if cf == 0:
            want = np.concatenate([cols[i].reshape(h, w, 3), np.full((h, w, 1), 255, np.uint8)], -1)
        else:
            want = orc.decode_dxt(cols[i], w, h, cf)
        np.testing.assert_array_equal(col[i], want)
        wd = deps[i].view(np.float32).reshape(h, w) if df == 0 else (deps[i].astype(np.float32) / np.float32(255.0)).reshape(h, w)
        np.testing.assert_array_equal(depth[i], wd)
    if df == 0 and cf == 0:                                       # the float path is the raw-frame path, byte for byte
        ref = rr.ReconIntegrationHip(small_scene, **KW)
        ref.upload_raw_frame(small_scene)
        d2, c2 = ref.raw_frame()
        np.testing.assert_array_equal(depth, d2)
        assert (col.reshape(-1, 4)[3:] == c2.reshape(-1, 4)[3:]).all()      # the first 8 colour bytes carry the timestamp


@pytest.mark.parametrize("fmt,name", [(1, "dxt1"), (5, "dxt5")])
def test_random_blocks_decode_like_the_reference_squish(rr, small_scene, tmp_path, fmt, name):
    """Every decoder mode (three-colour + transparent, five-step alpha) straight against the reference's codec."""
    if not os.path.exists(TOOL):
        pytest.skip("oracle/_ref not built")
    n, w, h = small_scene["n"], small_scene["width"], small_scene["height"]
    hip = rr.ReconIntegrationHip(small_scene, **KW)
    hip.setWireFormat(fmt, 1)
    cs, ds, total = hip.wireSizes()
    rng = np.random.default_rng(5)
    msg = rng.integers(0, 256, total, dtype=np.uint8)
    hip.upload_wire_frame(msg.tobytes())
    _, col = hip.raw_frame()
    for i in range(n):
        (tmp_path / "b.dxt").write_bytes(msg[i * (cs + ds): i * (cs + ds) + cs].tobytes())
        subprocess.check_call([TOOL, "decompress", name, str(tmp_path / "b.dxt"), str(w), str(h), str(tmp_path / "o.rgba")])
        np.testing.assert_array_equal(col[i], np.fromfile(tmp_path / "o.rgba", np.uint8).reshape(h, w, 4))


@pytest.mark.parametrize("cf,df,flags", [(1, 1, dict(processed_depth=False)), (0, 1, dict()), (5, 0, dict())])
def test_wire_frame_through_process_textures(rr, small_scene, cf, df, flags):
    msg = rr.scene.make_wire_message(small_scene, cf, df)
    hip, o = rr.ReconIntegrationHip(small_scene, **KW), OracleRecon(small_scene, **KW)
    hip.setWireFormat(cf, df)
    hip.upload_wire_frame(msg, small_scene)
    o.upload_raw_frame(small_scene)                                # sets depth limits + camera positions
    o.upload_wire_frame(msg, cf, df)
    for x in (hip, o):
        for i in range(small_scene["n"]):
            x.setDepthCompression(i, df == 1, 0.5, 4.5)
        x.setPreprocess(**flags)
        x.clearOccupiedBricks()
        x.processTextures()
    a, b = hip.preprocessed(), o.preprocessed()
    for k in ("depth2", "depth_rg", "depth_b", "silhouette", "normals"):
        assert same(a[k], b[k]).all(), f"{k}: {(~same(a[k], b[k])).sum()} of {a[k].size} differ"
    assert np.abs(a["lab"] - b["lab"]).max() <= 1e-6
    with np.errstate(invalid="ignore"):
        ok = (np.abs(a["quality"] - b["quality"]) <= 1e-5 * np.maximum(np.abs(b["quality"]), 1e-3)) | (np.isnan(a["quality"]) & np.isnan(b["quality"]))
    assert ok.all()
    np.testing.assert_array_equal(hip.bricks()[0], o.counters())
    if not (df == 1 and flags.get("processed_depth", True)):
        assert (b["silhouette"] > 0).sum() > 500                   # a real frame came through
    # (8-bit depth + processed depth: pre_morph.fs validates the NORMALISED codes against 0.5..4.5 m, so the
    #  reference itself keeps only codes > 127 -- restated literally)


def test_wire_message_errors(rr, small_scene):
    hip = rr.ReconIntegrationHip(small_scene, **KW)
    with pytest.raises(rr.TsdfError):
        hip.setWireFormat(3, 0)
    hip.setWireFormat(1, 1)
    with pytest.raises(rr.TsdfError) as e:
        hip.upload_wire_frame(b"\0" * 100)
    assert "wire message must be" in str(e.value)
