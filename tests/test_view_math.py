"""The host matrix block of ReconIntegration::draw() (recon_integration.cpp:66-72,182-205) against the REFERENCE's own
matrix libraries: oracle/_ref/ref_view_math is compiled from external/gloost/{Matrix,Vector3,Point3,Ray}.cpp and the vendored
glm where they lie under /root/reference (oracle/ref/Makefile; built by __graft_entry__.build(), travels to the GPU box).

The reference forms these matrices in fp32 (gloost Gauss-Jordan inverse, glm cofactor inverse); the library and the oracle form
them in double and round once, so agreement is to fp32 rounding of the reference's own arithmetic -- the tolerance below is
relative to the largest element of each matrix; the library and the oracle agree to the last fp32 bit or two."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "oracle", "_ref", "ref_view_math")
needs_ref = pytest.mark.skipif(not os.path.exists(TOOL), reason="oracle/_ref not built (needs /root/reference: __graft_entry__.build())")

BBOX = (np.array([-1.0, 0.0, -1.0], np.float32), np.array([1.0, 2.2, 1.0], np.float32))      # kinect_client.cpp:206-207


def views(rr):
    s = rr.scene
    out = [("bench", (1280, 720)) + tuple(s.default_view(1280, 720))]
    rng = np.random.default_rng(99)
    for k in range(6):
        eye = rng.uniform(-3, 3, 3) + np.array([0, 1.1, 0]); eye[2] += 2.0 if abs(eye[2]) < 0.5 else 0.0
        at = rng.uniform(-0.5, 0.5, 3) + np.array([0, 1.1, 0])
        w, h = [(640, 480), (1920, 1080), (333, 777)][k % 3]
        mv = s.gl_flat(s.look_at(tuple(eye), tuple(at)))
        pr = s.gl_flat(s.perspective(float(rng.uniform(25, 90)), w / float(h), float(rng.uniform(0.05, 0.5)), float(rng.uniform(20, 500))))
        out.append((f"random{k}", (w, h), mv, pr))
    return out


def reference(mv, pr, view, bbox):
    args = [TOOL] + ["%.9g" % v for v in np.concatenate([mv, pr])] + [str(view[0]), str(view[1])] + ["%.9g" % v for v in np.concatenate(bbox)]
    v = np.array(subprocess.run(args, check=True, capture_output=True, text=True).stdout.split(), np.float64)
    assert v.size == 51
    return {"vol_to_world": v[:16], "image_to_eye": v[16:32], "normal_matrix": v[32:48], "camera_pos": v[48:]}


def close(a, b, rel):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.max(np.abs(a - b)) <= rel * max(np.max(np.abs(b)), 1e-30)


@needs_ref
def test_library_matrices_match_the_reference_libraries(rr):
    for name, view, mv, pr in views(rr):
        ref = reference(mv, pr, view, BBOX)
        got = rr.view_matrices(mv, pr, view, *BBOX)
        assert np.array_equal(got["vol_to_world"], ref["vol_to_world"].astype(np.float32)), name          # exact: no rounding involved
        assert close(got["image_to_eye"], ref["image_to_eye"], 1e-6), name   
        assert close(got["normal_matrix"], ref["normal_matrix"], 1e-6), name
        assert close(got["camera_pos"], ref["camera_pos"], 1e-6), name


@needs_ref
def test_oracle_matrices_match_the_reference_libraries(rr):
    from oracle.oracle import OracleRecon
    sc = rr.scene.make_scene(n_streams=1, width=32, height=24, lut_res=8, inv_res=8)
    for name, view, mv, pr in views(rr):
        o = OracleRecon(sc, res=(8, 8, 8), brick_size=0.5, limit=0.01, view=view)
        ref = reference(mv, pr, view, (sc["bbox_min"].astype(np.float32), sc["bbox_max"].astype(np.float32)))
        got = o.view_matrices_all(mv, pr)
        assert np.array_equal(got["vol_to_world"], ref["vol_to_world"].astype(np.float32)), name
        assert close(got["image_to_eye"], ref["image_to_eye"], 1e-6), name
        assert close(got["normal_matrix"], ref["normal_matrix"], 1e-6), name
        assert close(got["camera_pos"], ref["camera_pos"], 1e-6), name
        lib = rr.view_matrices(mv, pr, view, sc["bbox_min"], sc["bbox_max"])
        for k in got:                                                           # library vs oracle: both double, rounded once (two inverse routines)
            assert close(got[k], lib[k], 2e-7), (name, k)


def test_singular_matrices_are_an_error(rr):
    mv, pr = rr.scene.default_view(64, 36)
    with pytest.raises(rr.TsdfError):
        rr.view_matrices(np.zeros(16, np.float32), pr, (64, 36), *BBOX)
    with pytest.raises(rr.TsdfError):
        rr.view_matrices(mv, np.zeros(16, np.float32), (64, 36), *BBOX)
