"""-m gpu: seeded random configurations (odd resolutions, brick sizes unrelated to the tiles, 1-6 streams, odd view sizes, cameras
anywhere around the scene, random option toggles between frames) -- every frame of every case against the oracle, with the path's
tolerances.  The cases are fixed by their seeds; a failure prints the seed's configuration."""
import numpy as np
import pytest

from helpers import POW_ATOL, assert_frames_identical, assert_same
from oracle.oracle import OracleRecon

pytestmark = pytest.mark.gpu


def make_case(rr, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 7))
    w, h = int(rng.choice([64, 96, 128, 160])), int(rng.choice([48, 72, 96, 120]))
    scene_kw = dict(n_streams=n, width=w, height=h, lut_res=int(rng.choice([16, 24, 32])), inv_res=int(rng.choice([16, 24, 32])), seed=int(seed))
    res = tuple(int(x) for x in rng.integers(20, 72, 3))
    brick = [float(x) for x in rng.uniform(0.12, 0.5, 3)]
    limit = float(rng.uniform(0.03, 0.08))
    view = (int(rng.integers(48, 200)), int(rng.integers(32, 120)))
    kw = dict(res=res, brick_size=brick, limit=limit, view=view)
    return rng, scene_kw, kw


def random_view(rr, rng, view):
    ang, elev, dist = rng.uniform(0, 2 * np.pi), rng.uniform(-0.3, 0.9), rng.uniform(2.2, 4.0)
    eye = (dist * np.cos(ang) * np.cos(elev), 1.1 + dist * np.sin(elev), dist * np.sin(ang) * np.cos(elev))
    mv = rr.scene.gl_flat(rr.scene.look_at(eye, (rng.uniform(-0.2, 0.2), 1.1 + rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2))))
    pr = rr.scene.gl_flat(rr.scene.perspective(rng.uniform(35.0, 65.0), view[0] / float(view[1]), 0.1, 200.0))
    return mv, pr


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16, 17, 18, 19, 20])
def test_random_configuration_matches_oracle(rr, seed):
    rng, scene_kw, kw = make_case(rr, seed)
    scenes = [rr.scene.make_scene(**scene_kw),
              rr.scene.make_scene(sphere_c=tuple(rng.uniform([-0.5, 0.6, -0.5], [0.5, 1.6, 0.5])), box_c=tuple(rng.uniform([-0.6, 0.3, -0.6], [0.6, 1.7, 0.6])), **scene_kw)]
    info = f"seed {seed}: {scene_kw} {kw}"
    hip, orc = rr.ReconIntegrationHip(scenes[0], **kw), OracleRecon(scenes[0], **kw)
    state = dict(use_bricks=True, skip=True, fill=True, shade=0, min_voxels=10)
    for f in range(5):
        sc = scenes[f % 2]
        if f:                                                          # toggle something between frames
            what = rng.integers(0, 6)
            if what == 0: state["use_bricks"] = not state["use_bricks"]
            elif what == 1: state["skip"] = not state["skip"]
            elif what == 2: state["fill"] = not state["fill"]
            elif what == 3: state["shade"] = int(rng.integers(0, 4))
            elif what == 4: state["min_voxels"] = int(rng.integers(1, 30))
        mv, pr = random_view(rr, rng, kw["view"])
        ratios = []
        for o in (hip, orc):
            o.upload_frame(sc)
            o.setUseBricks(state["use_bricks"]); o.setSpaceSkip(state["skip"]); o.setColorFilling(state["fill"])
            o.setShadeMode(state["shade"]); o.setMinVoxelsPerBrick(state["min_voxels"])
            o.clearOccupiedBricks(); o.markBricks(); ratios.append(o.updateOccupiedBricks()); o.integrate(); o.drawF(mv, pr)
        assert ratios[0] == ratios[1], info
        np.testing.assert_array_equal(hip.bricks()[0], orc.counters(), err_msg=info)
        tag = f"{info} frame {f} {state}"
        assert_same(hip.tsdf(), orc.tsdf(), tag + " tsdf")
        assert_frames_identical(hip, orc, tag, min_hits=0, colour_atol=POW_ATOL if state["shade"] == 1 else 0.0)
