"""Generates tests/golden/*.npz from the CPU oracle (run in the build container: python tests/golden/make_golden.py).

The reference ships no golden vectors for this path and its GLSL cannot run here, so
these fixtures pin the ORACLE's behaviour on seeded synthetic inputs: a later change to oracle/ that
alters any value fails tests/test_golden.py, and the HIP path is compared with the same files on the
GPU box (where /root/reference and a rebuilt oracle are not needed).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import rgbd_recon_amd as rr  # noqa: E402
from oracle.oracle import OracleRecon  # noqa: E402

SCENE_KW = dict(n_streams=2, width=96, height=72, lut_res=16, inv_res=24, seed=1234)
RECON_KW = dict(res=(32, 32, 32), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.08, view=(64, 36))


def generate():
    scene = rr.scene.make_scene(**SCENE_KW)
    o = OracleRecon(scene, **RECON_KW)
    mv, pr = rr.scene.default_view(*RECON_KW["view"])
    o.clearOccupiedBricks(); o.markBricks()
    ratio = o.updateOccupiedBricks()
    o.integrate()
    tsdf_bricks = o.tsdf()
    counters = o.counters()
    o.draw(mv, pr)
    rgba, depth, ns, peels = o.view_images()
    o.fillColors()
    atlas_c, atlas_d = o.atlas()
    fb_c, fb_d = o.framebuffer()
    o.setUseBricks(False)
    o.integrate()
    tsdf_dense = o.tsdf()
    return dict(counters=counters, ratio=np.float32(ratio), tsdf_bricks=tsdf_bricks, tsdf_dense=tsdf_dense,
                rgba=rgba, depth=depth, nsamples=ns, peels=peels, atlas_c=atlas_c, atlas_d=atlas_d, fb_c=fb_c, fb_d=fb_d,
                mv=mv, pr=pr)


# ---- the rows around the path: pre-processing, ingest decode, inverse LUT, point / triangle-grid frames
WIDE_SCENE_KW = dict(n_streams=2, width=96, height=72, lut_res=16, inv_res=8, seed=4321)
WIDE_RECON_KW = dict(res=(32, 32, 32), brick_size=[2.0 / 8, 2.2 / 8, 2.0 / 8], limit=0.08, view=(160, 90))
WIDE_MIN_LENGTH = 0.1                                        # 96 x 72 depth images: ~3 cm pixel pitch


def wide_inputs():
    scene = rr.scene.make_scene(**WIDE_SCENE_KW)
    rng = np.random.default_rng(99)
    dxt1 = rng.integers(0, 256, (16 // 4) * (8 // 4) * 8, dtype=np.uint8)          # 16 x 8 pixels of random blocks: every decoder mode
    dxt5 = rng.integers(0, 256, (16 // 4) * (8 // 4) * 16, dtype=np.uint8)
    xyz = np.ascontiguousarray(scene["cv_xyz"][0].reshape(16, 16, 16, 3)[:, ::-1])   # v flipped: kinect::Frustum's handedness
    return scene, dxt1, dxt5, xyz


def generate_wide():
    from oracle import oracle as orc
    scene, dxt1, dxt5, xyz = wide_inputs()
    o = OracleRecon(scene, **WIDE_RECON_KW)
    mv, pr = rr.scene.default_view(*WIDE_RECON_KW["view"])
    o.upload_raw_frame(scene)
    o.clearOccupiedBricks(); o.processTextures()
    pp = o.preprocessed()
    counters = o.counters()
    out = {"pp_" + k: v for k, v in pp.items()}
    out["pp_counters"] = counters
    out["dxt1_blocks"], out["dxt5_blocks"] = dxt1, dxt5
    out["dxt1_rgba"], out["dxt5_rgba"] = orc.decode_dxt(dxt1, 16, 8, 1), orc.decode_dxt(dxt5, 16, 8, 5)
    out["inv"] = orc.invert_calibration(xyz, scene["bbox_min"], scene["bbox_max"], (10, 11, 10))
    planes, cam = orc.frustum(xyz)
    out["frustum_planes"], out["frustum_camera"] = planes, cam
    o.setShadeMode(1)
    o.drawPoints(mv, pr)
    out["points_c"], out["points_d"] = o.framebuffer()
    o.setMinLength(WIDE_MIN_LENGTH)
    o.drawTrigrid(mv, pr)
    out["trigrid_c"], out["trigrid_d"] = o.framebuffer()
    out["mv"], out["pr"] = mv, pr
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for name, gen in (("frame_2streams_32cubed.npz", generate), ("widening_2streams.npz", generate_wide)):
        out = os.path.join(here, name)
        np.savez_compressed(out, **gen())
        print(out, os.path.getsize(out), "bytes")
