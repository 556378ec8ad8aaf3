"""Import alias: ``import rgbd_recon_amd`` -> the package in ``rgbd-recon_amd/`` (hyphenated directory)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
_pkg = importlib.import_module("rgbd-recon_amd")
sys.modules[__name__] = _pkg
