"""Calibration volume file I/O against the REFERENCE's own reader/writer (oracle/_ref/ref_calib_volume is compiled from
/root/reference/framework/calibration/calibration_volume.hpp by oracle/ref/Makefile; the binary travels, the sources do not).
This is the one part of the reference that builds in this image, so this row is pinned by the reference itself."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "oracle", "_ref", "ref_calib_volume")
pytestmark = pytest.mark.skipif(not os.path.exists(TOOL), reason="oracle/_ref not built (needs /root/reference: __graft_entry__.build())")

KINDS = {"xyz": ("cv_xyz", 3), "uv": ("cv_uv", 2), "inv": ("cv_xyz_inv", 4)}


def fnv1a64(b):
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.parametrize("kind", sorted(KINDS))
def test_reference_written_file_is_read_back_identically(rr, tmp_path, kind):
    ext, c = KINDS[kind]
    path = str(tmp_path / f"ref.{ext}")
    subprocess.check_call([TOOL, "write", kind, path, "5", "3", "4", "0.5", "4.5", "1234"])
    vol, lim = rr.read_calib_volume(path, ext)
    assert vol.shape == (4, 3, 5, c) and lim == (0.5, 4.5)
    said = subprocess.check_output([TOOL, "read", kind, path]).split()
    assert int(said[5], 16) == fnv1a64(vol.tobytes())
    assert float(said[6]) == pytest.approx(vol.flat[0], rel=1e-8) and float(said[7]) == pytest.approx(vol.flat[-1], rel=1e-8)


@pytest.mark.parametrize("kind", sorted(KINDS))
def test_file_written_here_is_what_the_reference_reads(rr, tmp_path, kind):
    ext, c = KINDS[kind]
    rng = np.random.default_rng(7)
    vol = rng.standard_normal((6, 2, 7, c)).astype(np.float32)          # [rz][ry][rx][c], x fastest (calibration_volume.hpp:57-59)
    path = str(tmp_path / f"ours.{ext}")
    rr.write_calib_volume(path, ext, vol, (0.25, 3.75))
    said = subprocess.check_output([TOOL, "read", kind, path]).split()
    assert [int(x) for x in said[:3]] == [7, 2, 6] and float(said[3]) == 0.25 and float(said[4]) == 3.75
    assert int(said[5], 16) == fnv1a64(vol.tobytes())
    # and byte-for-byte what the reference writer produces for the same texels: round trip through our reader
    back, lim = rr.read_calib_volume(path, ext)
    np.testing.assert_array_equal(back, vol)
    assert os.path.getsize(path) == 20 + vol.size * 4


def test_truncated_or_mistyped_files_are_errors_not_asserts(rr, tmp_path):
    path = str(tmp_path / "t.cv_xyz")
    subprocess.check_call([TOOL, "write", "xyz", path, "4", "4", "4", "0.5", "4.5", "1"])
    with pytest.raises(rr.TsdfError):
        rr.read_calib_volume(path, "cv_uv")                            # 3-float texels read as 2-float texels
    with open(path, "r+b") as f:
        f.truncate(100)
    with pytest.raises(rr.TsdfError):
        rr.read_calib_volume(path, "cv_xyz")
    with pytest.raises(rr.TsdfError):
        rr.read_calib_volume(str(tmp_path / "missing.cv_xyz"), "cv_xyz")
