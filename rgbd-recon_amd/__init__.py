"""rgbd-recon_amd -- MI355X-native TSDF fusion core behind rgbd-recon's ReconIntegration surface.

  csrc/      hand-written HIP kernels (gfx950) + the C ABI of include/rgbd_recon_hip.h
  host/      C++ adapter with the reference's class/method names (drop-in for the GL operator)
  binding.py ctypes binding + Python mirror of the operator (used by tests/, bench.py)
  scene.py   synthetic calibrated RGB-D scene (harness input)
  multigpu.py  Z-slab partition driver: one process per GPU, torch.distributed (RCCL) exchange

The directory name carries a hyphen (project naming); import it as ``rgbd_recon_amd`` (alias module at
the repository root) or with importlib.
"""
from .binding import (LANES_NO_FILL_THREAD, LANES_NO_INTEGRATE_LANE, LANES_ONE_STREAM, LANES_SHARED_FILL_LANE, ReconIntegrationHip, TsdfConfig, TsdfError, build_library, declared_symbols,  # noqa: F401
                      load_library, LIB_PATH, HEADER_PATH, read_calib_volume, write_calib_volume,
                      read_stream_record, stream_num_frames, frustum_from_volume, view_matrices, inverse_volume_resolution, invert_calibration, COLOR_RGB8, COLOR_DXT1, COLOR_DXT5, DEPTH_F32, DEPTH_U8)
from . import scene  # noqa: F401
