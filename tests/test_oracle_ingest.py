"""Frame ingest: the oracle's DXT decode and record split pinned against the REFERENCE's own code
(oracle/_ref/ref_wire_tool = the reference's vendored squish + framework/io/FileBuffer.cpp, built by oracle/ref/Makefile),
plus known-answer tests of the 8-bit depth mapping (pre_depth.fs:51-61)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "oracle", "_ref", "ref_wire_tool")
needs_ref = pytest.mark.skipif(not os.path.exists(TOOL), reason="oracle/_ref not built (needs /root/reference: __graft_entry__.build())")


def picture(w, h, seed=0):
    """Smooth gradients + an edge + noise: blocks of every kind (flat, two-colour, gradients)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([x * 255 // max(w - 1, 1), y * 255 // max(h - 1, 1), (x + y) % 256, 255 * ((x // 8 + y // 8) % 2)], -1).astype(np.int32)
    img[h // 3: h // 2] += rng.integers(-40, 40, (h // 2 - h // 3, w, 4))
    img[:, w // 2:, :3] = img[:, w // 2:, :3] // 3
    return np.clip(img, 0, 255).astype(np.uint8)


@needs_ref
@pytest.mark.parametrize("fmt,name,w,h", [(1, "dxt1", 64, 48), (5, "dxt5", 64, 48), (1, "dxt1", 30, 18), (5, "dxt5", 22, 9)])
def test_dxt_decode_equals_the_reference_squish_on_compressed_pictures(tmp_path, fmt, name, w, h):
    src = picture(w, h, seed=fmt)
    if fmt == 1:
        src[..., 3] = 255
        src[2:7, 3:11, 3] = 0                                   # punch-through alpha: forces three-colour DXT1 blocks
    (tmp_path / "in.rgba").write_bytes(src.tobytes())
    subprocess.check_call([TOOL, "compress", name, str(tmp_path / "in.rgba"), str(w), str(h), str(tmp_path / "b.dxt")])
    subprocess.check_call([TOOL, "decompress", name, str(tmp_path / "b.dxt"), str(w), str(h), str(tmp_path / "out.rgba")])
    blocks = np.fromfile(tmp_path / "b.dxt", np.uint8)
    want = np.fromfile(tmp_path / "out.rgba", np.uint8).reshape(h, w, 4)
    got = orc.decode_dxt(blocks, w, h, fmt)
    np.testing.assert_array_equal(got, want)
    assert np.abs(got[..., :3].astype(int) - src[..., :3].astype(int))[src[..., 3] > 0].mean() < 12   # and it is the picture, not garbage


@needs_ref
@pytest.mark.parametrize("fmt,name", [(1, "dxt1"), (5, "dxt5")])
def test_dxt_decode_equals_the_reference_squish_on_random_blocks(tmp_path, fmt, name):
    """Random bytes reach every mode: c0 <= c1 (three colours + transparent), a0 <= a1 (five alphas + 0/255), all indices."""
    w, h = 128, 64
    rng = np.random.default_rng(11 + fmt)
    blocks = rng.integers(0, 256, (w // 4) * (h // 4) * (8 if fmt == 1 else 16), dtype=np.uint8)
    blocks[:16] = 0                                              # degenerate equal endpoints
    blocks[16:32] = 255
    (tmp_path / "b.dxt").write_bytes(blocks.tobytes())
    subprocess.check_call([TOOL, "decompress", name, str(tmp_path / "b.dxt"), str(w), str(h), str(tmp_path / "out.rgba")])
    want = np.fromfile(tmp_path / "out.rgba", np.uint8).reshape(h, w, 4)
    np.testing.assert_array_equal(orc.decode_dxt(blocks, w, h, fmt), want)


def test_dxt1_block_by_hand():
    # c0 = pure red (31,0,0) = 0xF800, c1 = pure blue 0x001F; c0 > c1: four-colour mode
    blk = np.array([0x00, 0xF8, 0x1F, 0x00, 0b11100100, 0, 0, 0], np.uint8)       # row 0 indices 0,1,2,3
    out = orc.decode_dxt(blk, 4, 4, 1)
    assert out[0, 0].tolist() == [255, 0, 0, 255] and out[0, 1].tolist() == [0, 0, 255, 255]
    assert out[0, 2].tolist() == [170, 0, 85, 255] and out[0, 3].tolist() == [85, 0, 170, 255]    # (2a+b)/3, (a+2b)/3 in integers
    # swapped endpoints: three-colour mode, index 2 = (a+b)/2, index 3 = transparent black
    blk = np.array([0x1F, 0x00, 0x00, 0xF8, 0b11100100, 0, 0, 0], np.uint8)
    out = orc.decode_dxt(blk, 4, 4, 1)
    assert out[0, 2].tolist() == [127, 0, 127, 255] and out[0, 3].tolist() == [0, 0, 0, 0]
    # 565 expansion is bit replication: g = 0b100001 -> 0b10000110
    blk = np.array([0x20, 0x04, 0x20, 0x04, 0, 0, 0, 0], np.uint8)                # 0x0420: r=0, g=33, b=0
    assert orc.decode_dxt(blk, 4, 4, 1)[0, 0].tolist() == [0, 134, 0, 255]


def test_wire_message_layout():
    """readLoop(): per sensor [colour][depth]; the double timestamp overlays the first 8 colour bytes (offset starts at 0)."""
    n, cs, ds = 3, 24, 16
    msg = bytearray(np.arange(n * (cs + ds), dtype=np.uint8).tobytes())
    msg[:8] = np.float64(1234.5).tobytes()
    ts, cols, deps = orc.wire_split(bytes(msg), n, cs, ds)
    assert ts == 1234.5
    assert cols[1][0] == 40 and cols[1].size == cs and deps[1][0] == 64 and deps[2][-1] == 119
    assert orc.wire_sizes(640, 480, 640, 480, 0, 0) == (921600, 1228800)
    assert orc.wire_sizes(640, 480, 640, 480, 1, 1) == (153600, 307200)          # DXTCompressor storage, NetKinectArray.cpp:118-121
    assert orc.wire_sizes(640, 480, 640, 480, 5, 1) == (307200, 307200)          # the literal 307200 of :125


@needs_ref
def test_stream_file_records_are_what_the_reference_filebuffer_reads(rr, tmp_path):
    """recordings/<sensor>.stream: raw records [colour][depth] back to back (readFromFiles, NetKinectArray.cpp:709-749)."""
    cs, ds = 48, 20
    rng = np.random.default_rng(3)
    recs = [rng.integers(0, 256, cs + ds, dtype=np.uint8) for _ in range(4)]
    path = str(tmp_path / "23.stream")
    for r in recs:                                               # written by the reference's FileBuffer::write
        (tmp_path / "r.bin").write_bytes(r.tobytes())
        subprocess.check_call([TOOL, "stream-append", path, str(tmp_path / "r.bin")])
    for k in (0, 2, 3):
        subprocess.check_call([TOOL, "stream-read", path, str(cs), str(ds), str(k), str(tmp_path / "o.bin")])
        np.testing.assert_array_equal(np.fromfile(tmp_path / "o.bin", np.uint8), recs[k])
        np.testing.assert_array_equal(rr.read_stream_record(path, cs + ds, k), recs[k])           # the product's reader
    assert subprocess.call([TOOL, "stream-read", path, str(cs), str(ds), "4", str(tmp_path / "o.bin")], stderr=subprocess.DEVNULL) == 3
    with pytest.raises(rr.TsdfError):
        rr.read_stream_record(path, cs + ds, 4)                  # past the end, looping off: the reference reads 0 bytes
    assert rr.stream_num_frames(path, cs + ds) == 4              # FileBuffer::calcNumFrames


def test_uncompress_known_answers(rr):
    """pre_depth.fs:51-61 through the filter pass with filter_textures off: out = normalised uncompress(d_c)."""
    from helpers import tiny_scene
    sc = tiny_scene([(0.5, 0.5, 0.5)], [0.5], [1.0], [1.0], w=4, h=1, lut=2)
    sc["depth_limits"] = np.array([0.0, 8.0], np.float32)
    sc["camera_positions"] = np.zeros((1, 3), np.float32)
    o = orc.OracleRecon(sc, res=(8, 8, 8), brick_size=0.5, view=(8, 8))
    near, far = 0.5, 4.5
    o.setDepthCompression(0, True, near, far)
    o.setPreprocess(filter_textures=False, processed_depth=False, refine=True)
    codes = np.array([0, 1, 128, 255], np.uint8)
    msg = np.zeros(4 * 3, np.uint8).tobytes() + codes.tobytes()
    o.upload_wire_frame(msg, color_format=0, depth_format=1)
    o._L.orc_set_depth_limits(o._c, 0, orc.C.c_float(0.0), orc.C.c_float(8.0))
    o._L.orc_set_camera_position(o._c, 0, orc._p(np.zeros(3, np.float32)))
    o.processTextures()
    got = o.preprocessed()["depth_rg"][0, 0, :, 0] * np.float32(8.0)
    scale = np.float32(far - near)
    sn = scale / np.float32(255.0)
    f = codes.astype(np.float32) / np.float32(255.0)
    want = np.where(f < sn, np.float32(0.0), (f * f + np.float32(0.15) * sn) * scale + np.float32(near))
    # code 0 (< scaled_near = 4/255 -> 0.0157) is "no measurement" -> 0; code 255 maps just beyond far: 4.5 + 0.15*4/255*4
    assert want[0] == 0.0 and want[1] == 0.0 and abs(want[3] - (4.5 + 0.15 * 4 / 255 * 4)) < 1e-5
    np.testing.assert_allclose(got, want, rtol=2e-7, atol=1e-7)
