import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rr():
    # torch bundles its own HIP runtime: when a process uses both torch and the library (the multi-GPU tests do), torch has
    # to be imported FIRST so that librgbd_recon_hip.so binds to the runtime that is already loaded (bench.py does the same)
    import torch  # noqa: F401
    import rgbd_recon_amd
    return rgbd_recon_amd


@pytest.fixture(scope="session")
def small_scene(rr):
    """4 streams, 160x120 images, 32^3 LUTs: the oracle finishes every stage in well under a second."""
    return rr.scene.make_scene(n_streams=4, width=160, height=120, lut_res=32, inv_res=32)
