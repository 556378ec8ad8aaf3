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


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "frame_2streams_32cubed.npz")
    np.savez_compressed(out, **generate())
    print(out, os.path.getsize(out), "bytes")
