"""ctypes binding of librgbd_recon_hip.so (include/rgbd_recon_hip.h) and a Python mirror of the
reference operator surface, ``kinect::ReconIntegration``
(framework/reconstruction/recon_integration.hpp:35-103): same method names, same call order
(source/kinect_client.cpp:569-599,614), GL implicit state passed explicitly.

There is NO CPU fallback: if the HIP library is missing or no device is visible the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.environ.get("RGBDR_LIB", os.path.join(_PKG, "librgbd_recon_hip.so"))   # override: A/B builds of the kernels
HEADER_PATH = os.path.join(_ROOT, "include", "rgbd_recon_hip.h")

TSDF_MAX_STREAMS = 16


class TsdfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"tsdf error {code}: {msg}")
        self.code = code


class TsdfConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3),
                ("voxel_size", C.c_float), ("res", C.c_uint32 * 3), ("brick_size", C.c_float * 3),
                ("limit", C.c_float), ("num_streams", C.c_uint32), ("depth_w", C.c_uint32), ("depth_h", C.c_uint32),
                ("color_w", C.c_uint32), ("color_h", C.c_uint32), ("view_w", C.c_uint32), ("view_h", C.c_uint32),
                ("device", C.c_int32), ("slab_z0", C.c_uint32), ("slab_z1", C.c_uint32), ("slab_recompute_halo", C.c_uint32),
                ("sparse_pool_tiles", C.c_uint32), ("proj_cache_mib", C.c_uint32), ("lane_flags", C.c_uint32), ("lane_priority", C.c_int32 * 4)]


LANES_ONE_STREAM, LANES_NO_INTEGRATE_LANE, LANES_NO_FILL_THREAD, LANES_SHARED_FILL_LANE = 1, 2, 4, 8   # tsdf_config::lane_flags


def build_library():
    """hipcc cross-compiles gfx950 without a GPU."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_PKG, "csrc")])


def declared_symbols():
    """Every entry point include/rgbd_recon_hip.h declares."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tsdf_[a-z0-9_]+)\s*\(", text)))


_lib = None


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} is missing: run __graft_entry__.build() (there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.tsdf_last_error.restype = C.c_char_p
        L.tsdf_last_error.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


# ---------------------------------------------------------------------- calibration volume files (host only)
_TEXEL = {"cv_xyz": 3, "cv_uv": 2, "cv_xyz_inv": 4}


def read_calib_volume(path, kind):
    """Reads a *.cv_xyz / *.cv_uv / *.cv_xyz_inv file (calibration_volume.hpp:62-78) -> (array [rz][ry][rx][c], depth limits)."""
    L = load_library()
    L.tsdf_calib_last_error.restype = C.c_char_p
    c = _TEXEL[kind]
    res, lim = (C.c_uint32 * 3)(), (C.c_float * 2)()
    if L.tsdf_calib_volume_info(path.encode(), c, res, lim) != 0:
        raise TsdfError(-1, L.tsdf_calib_last_error().decode())
    out = np.empty((res[2], res[1], res[0], c), np.float32)
    if L.tsdf_calib_volume_read(path.encode(), c, _fp(out), C.c_uint64(out.size)) != 0:
        raise TsdfError(-1, L.tsdf_calib_last_error().decode())
    return out, (lim[0], lim[1])


def write_calib_volume(path, kind, volume, depth_limits):
    """Writes volume [rz][ry][rx][c] in the reference's on-disk format (calibration_volume.hpp:30-38)."""
    L = load_library()
    L.tsdf_calib_last_error.restype = C.c_char_p
    v = _f32(volume)
    assert v.ndim == 4 and v.shape[3] == _TEXEL[kind]
    res = (C.c_uint32 * 3)(v.shape[2], v.shape[1], v.shape[0])
    lim = (C.c_float * 2)(*[float(x) for x in depth_limits])
    if L.tsdf_calib_volume_write(path.encode(), _TEXEL[kind], res, lim, _fp(v)) != 0:
        raise TsdfError(-1, L.tsdf_calib_last_error().decode())


# ---------------------------------------------------------------------- draw() host matrices (recon_integration.cpp:66-72,182-205)
def view_matrices(mv, proj, view, bbox_min, bbox_max):
    """-> dict(vol_to_world [16], image_to_eye [16], normal_matrix [16], camera_pos [3]), column major; host only (no GPU)."""
    L = load_library()
    out = np.zeros(51, np.float32)
    rc = L.tsdf_view_matrices(_fp(_f32(mv).reshape(-1)), _fp(_f32(proj).reshape(-1)), C.c_uint32(int(view[0])), C.c_uint32(int(view[1])),
                              _fp(_f32(bbox_min)), _fp(_f32(bbox_max)), _fp(out))
    if rc != 0:
        raise TsdfError(rc, "singular modelview / projection matrix")
    return {"vol_to_world": out[:16].copy(), "image_to_eye": out[16:32].copy(), "normal_matrix": out[32:48].copy(), "camera_pos": out[48:].copy()}


# ---------------------------------------------------------------------- inverse calibration volumes (source/calib_inverter.cpp)
def frustum_from_volume(cv_xyz):
    """cv_xyz [rz][ry][rx][3] -> (planes [6][4], camera position): kinect::Frustum of the volume's corner texels (host only)."""
    L = load_library()
    L.tsdf_calib_last_error.restype = C.c_char_p
    v = _f32(cv_xyz)
    assert v.ndim == 4 and v.shape[3] == 3
    planes, cam = np.zeros((6, 4), np.float32), np.zeros(3, np.float32)
    if L.tsdf_frustum_from_volume(_fp(v), (C.c_uint32 * 3)(v.shape[2], v.shape[1], v.shape[0]), _fp(planes), _fp(cam)) != 0:
        raise TsdfError(-1, L.tsdf_calib_last_error().decode())
    return planes, cam


def inverse_volume_resolution(bbox_min, bbox_max, voxel_size=0.007):
    L = load_library()
    res = (C.c_uint32 * 3)()
    if L.tsdf_inverse_volume_resolution(_fp(_f32(bbox_min)), _fp(_f32(bbox_max)), C.c_float(voxel_size), res) != 0:
        raise TsdfError(-1, "bad argument")
    return tuple(res)


def invert_calibration(cv_xyz, bbox_min, bbox_max, res_inv, device=0):
    """CalibrationInverter::calculateInverseVolumes for one sensor on the GPU -> ([rz][ry][rx][4], device milliseconds)."""
    L = load_library()
    L.tsdf_calib_last_error.restype = C.c_char_p
    v = _f32(cv_xyz)
    assert v.ndim == 4 and v.shape[3] == 3
    out = np.empty((int(res_inv[2]), int(res_inv[1]), int(res_inv[0]), 4), np.float32)
    ms = C.c_float()
    rc = L.tsdf_invert_calibration(int(device), _fp(v), (C.c_uint32 * 3)(v.shape[2], v.shape[1], v.shape[0]), _fp(_f32(bbox_min)), _fp(_f32(bbox_max)),
                                   (C.c_uint32 * 3)(*[int(x) for x in res_inv]), _fp(out), C.byref(ms))
    if rc != 0:
        raise TsdfError(rc, L.tsdf_calib_last_error().decode())
    return out, ms.value


def stream_num_frames(path, record_bytes):
    """FileBuffer::calcNumFrames for recordings/<sensor>.stream."""
    L = load_library()
    L.tsdf_calib_last_error.restype = C.c_char_p
    n = C.c_uint64()
    if L.tsdf_stream_num_frames(path.encode(), C.c_uint64(record_bytes), C.byref(n)) != 0:
        raise TsdfError(-1, L.tsdf_calib_last_error().decode())
    return n.value


def read_stream_record(path, record_bytes, frame):
    """Record `frame` ([colour][depth]) of one sensor's .stream file (NetKinectArray::readFromFiles)."""
    L = load_library()
    L.tsdf_calib_last_error.restype = C.c_char_p
    out = np.empty(record_bytes, np.uint8)
    if L.tsdf_stream_read_record(path.encode(), C.c_uint64(record_bytes), C.c_uint64(frame), out.ctypes.data_as(C.c_void_p)) != 0:
        raise TsdfError(-1, L.tsdf_calib_last_error().decode())
    return out


COLOR_RGB8, COLOR_DXT1, COLOR_DXT5 = 0, 1, 5
DEPTH_F32, DEPTH_U8 = 0, 1


class ReconIntegrationHip:
    """HIP drop-in for kinect::ReconIntegration.  `scene` supplies what CalibrationFiles / CalibVolumes /
    NetKinectArray hold in the reference (rgbd-recon_amd/scene.py layout)."""

    def __init__(self, scene, res=None, voxel_size=0.01, brick_size=0.1, limit=0.01, view=(1280, 720),
                 device=0, slab=(0, 0), upload=True, recompute_halo=False, sparse_pool_tiles=0, proj_cache_mib=0, lane_flags=0, lane_priority=(0, 0, 0, 0)):
        self._L = load_library()
        self._c = None
        cfg = TsdfConfig()
        cfg.struct_size = C.sizeof(TsdfConfig)
        cfg.bbox_min[:] = [float(x) for x in scene["bbox_min"]]
        cfg.bbox_max[:] = [float(x) for x in scene["bbox_max"]]
        cfg.voxel_size = voxel_size
        cfg.res[:] = list(res) if res is not None else [0, 0, 0]
        cfg.brick_size[:] = [brick_size] * 3 if np.isscalar(brick_size) else list(brick_size)
        cfg.limit = limit
        cfg.num_streams = scene["n"]
        cfg.depth_w, cfg.depth_h = scene["width"], scene["height"]
        cfg.color_w, cfg.color_h = scene["color_width"], scene["color_height"]
        cfg.view_w, cfg.view_h = view
        cfg.device = device
        cfg.slab_z0, cfg.slab_z1 = slab
        cfg.slab_recompute_halo = int(bool(recompute_halo))
        cfg.sparse_pool_tiles = int(sparse_pool_tiles)
        cfg.proj_cache_mib = int(proj_cache_mib or 0)   # opt-in projection cache of the integrate kernel, MiB (0 / None: off)
        cfg.lane_flags = int(lane_flags)
        cfg.lane_priority[:] = [int(x) for x in lane_priority]
        ctx = C.c_void_p()
        rc = self._L.tsdf_create(C.byref(cfg), C.byref(ctx))
        if rc != 0:
            raise TsdfError(rc, self._L.tsdf_last_error(None).decode())
        self._c = ctx
        self.view = tuple(view)
        self.n = scene["n"]
        self._dims = (scene["height"], scene["width"], scene["color_height"], scene["color_width"])
        r3, b3, s3 = (C.c_uint32 * 3)(), (C.c_uint32 * 3)(), (C.c_float * 3)()
        self._ck(self._L.tsdf_get_resolution(self._c, r3, b3, s3))
        self.res, self.res_bricks, self.brick_size = tuple(r3), tuple(b3), tuple(s3)
        n = C.c_uint32()
        self._ck(self._L.tsdf_num_lods(self._c, C.byref(n)))
        self.num_lods = n.value
        if upload:
            self.set_calibration(scene)
            self.upload_frame(scene)

    # ------------------------------------------------------------------ plumbing
    def _ck(self, rc):
        if rc != 0:
            raise TsdfError(rc, self._L.tsdf_last_error(self._c).decode())

    def close(self):
        if self._c:
            self._L.tsdf_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream_ptr):
        """Adopt a caller-owned HIP stream.  None: back to the context's own stream.  0 is the handle of the process's NULL
        ("legacy default") stream -- what torch.cuda.default_stream().cuda_stream returns -- and is adopted as such
        (tsdf_adopt_null_stream); passed to tsdf_set_stream it would mean "back to own" and leave the context's kernels
        unordered against torch ops and RCCL collectives on the default stream."""
        if hip_stream_ptr is None:
            self._ck(self._L.tsdf_set_stream(self._c, None))
        elif int(hip_stream_ptr) == 0:
            self._ck(self._L.tsdf_adopt_null_stream(self._c))
        else:
            self._ck(self._L.tsdf_set_stream(self._c, C.c_void_p(int(hip_stream_ptr))))

    def sync(self):
        self._ck(self._L.tsdf_sync(self._c))

    def set_calibration(self, scene):
        ri = (C.c_uint32 * 3)(*[int(x) for x in scene["inv_res"]])
        rl = (C.c_uint32 * 3)(*[int(x) for x in scene["lut_res"]])
        for i in range(self.n):
            self._ck(self._L.tsdf_set_calibration(self._c, i, _fp(_f32(scene["cv_xyz_inv"][i])), ri,
                                                  _fp(_f32(scene["cv_uv"][i])), rl, _fp(_f32(scene["cv_xyz"][i])), rl))

    def upload_frame(self, scene):
        col = np.ascontiguousarray(scene["color"], np.uint8)
        self._ck(self._L.tsdf_upload_frame(self._c, _fp(_f32(scene["depth"])), _fp(_f32(scene["quality"])),
                                           _fp(_f32(scene["silhouette"])), col.ctypes.data_as(C.POINTER(C.c_uint8))))

    def upload_frame_dev(self, depth_ptr, quality_ptr, silhouette_ptr, colour_ptr=0, complete=False):
        """the frame's arrays are in device memory already (raw device pointers, e.g. torch tensors' data_ptr()): one re-layout launch.
        complete: the arrays are finished when the call is made (TSDF_FRAME_ARRAYS_COMPLETE: the launch need not wait for the context's stream)"""
        self._ck(self._L.tsdf_upload_frame_dev(self._c, C.c_void_p(int(depth_ptr)), C.c_void_p(int(quality_ptr)), C.c_void_p(int(silhouette_ptr)),
                                               C.c_void_p(int(colour_ptr)) if colour_ptr else None, C.c_uint32(1 if complete else 0)))

    # asynchronous upload into the frame slot that is not current (double PBO analog) + slot switch
    def frame_staging(self):
        """numpy views of the pinned staging buffer of the next upload_frame_async (depth_rg, quality, silhouette, colour)"""
        h, w, ch, cw = self._dims
        p = [C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.POINTER(C.c_uint8)()]
        self._ck(self._L.tsdf_frame_staging(self._c, C.byref(p[0]), C.byref(p[1]), C.byref(p[2]), C.byref(p[3])))
        return (np.ctypeslib.as_array(p[0], shape=(self.n, h, w, 2)), np.ctypeslib.as_array(p[1], shape=(self.n, h, w)),
                np.ctypeslib.as_array(p[2], shape=(self.n, h, w)), np.ctypeslib.as_array(p[3], shape=(self.n, ch, cw, 3)))

    def upload_frame_async(self, scene=None, with_colour=True):
        """scene None: the staging buffer was filled in place (frame_staging)"""
        if scene is None:
            self._ck(self._L.tsdf_upload_frame_async(self._c, None, None, None, None, int(with_colour)))
            return
        col = np.ascontiguousarray(scene["color"], np.uint8) if with_colour else None
        self._ck(self._L.tsdf_upload_frame_async(self._c, _fp(_f32(scene["depth"])), _fp(_f32(scene["quality"])), _fp(_f32(scene["silhouette"])),
                                                 col.ctypes.data_as(C.POINTER(C.c_uint8)) if col is not None else None, int(with_colour)))

    def select_frame_slot(self, slot): self._ck(self._L.tsdf_select_frame_slot(self._c, int(slot)))

    def current_frame_slot(self):
        s = C.c_uint32()
        self._ck(self._L.tsdf_current_frame_slot(self._c, C.byref(s)))
        return s.value

    # ------------------------------------------------------------------ NetKinectArray side: raw frame -> processTextures()
    def upload_raw_frame(self, scene):
        col = np.ascontiguousarray(scene["color"], np.uint8)
        self._ck(self._L.tsdf_upload_raw_frame(self._c, _fp(_f32(scene["depth_raw"])), col.ctypes.data_as(C.POINTER(C.c_uint8))))
        self.set_preprocess_calibration(scene)

    def set_preprocess_calibration(self, scene):
        """what processTextures() needs beside the LUTs: CalibVolumes::getDepthLimits / getCameraPositions of every sensor"""
        for i in range(self.n):
            self._ck(self._L.tsdf_set_depth_limits(self._c, i, C.c_float(float(scene["depth_limits"][0])), C.c_float(float(scene["depth_limits"][1]))))
            self._ck(self._L.tsdf_set_camera_position(self._c, i, _fp(_f32(scene["camera_positions"][i]))))
        self._pp_shape = (self.n, scene["height"], scene["width"])

    # ------------------------------------------------------------------ frame ingest: wire message -> raw frame (readLoop / update)
    def setWireFormat(self, color_format=COLOR_RGB8, depth_format=DEPTH_F32):
        self._ck(self._L.tsdf_set_wire_format(self._c, int(color_format), int(depth_format)))

    def wireSizes(self):
        cs, ds, total = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._ck(self._L.tsdf_wire_sizes(self._c, C.byref(cs), C.byref(ds), C.byref(total)))
        return cs.value, ds.value, total.value

    def setDepthCompression(self, stream, compressed, near, far):
        self._ck(self._L.tsdf_set_depth_compression(self._c, int(stream), int(compressed), C.c_float(near), C.c_float(far)))

    def upload_wire_frame(self, message, scene=None):
        """message: bytes-like, per sensor [colour][depth]; returns the timestamp in its first 8 bytes."""
        m = np.frombuffer(message, np.uint8)
        ts = C.c_double()
        self._ck(self._L.tsdf_upload_wire_frame(self._c, m.ctypes.data_as(C.c_void_p), C.c_uint64(m.size), C.byref(ts)))
        if scene is not None:
            for i in range(self.n):
                self._ck(self._L.tsdf_set_depth_limits(self._c, i, C.c_float(float(scene["depth_limits"][0])), C.c_float(float(scene["depth_limits"][1]))))
                self._ck(self._L.tsdf_set_camera_position(self._c, i, _fp(_f32(scene["camera_positions"][i]))))
            self._pp_shape = (self.n, scene["height"], scene["width"])
        return ts.value

    def raw_frame(self):
        """(raw depth [N][H][W], colour RGBA8 [N][ch][cw][4]) as unpacked on the GPU."""
        h, w, ch, cw = self._dims
        d = np.zeros((self.n, h, w), np.float32)
        col = np.zeros((self.n, ch, cw, 4), np.uint8)
        self._ck(self._L.tsdf_download_raw_frame(self._c, _fp(d), col.ctypes.data_as(C.POINTER(C.c_uint8))))
        return d, col

    def setPreprocess(self, filter_textures=True, processed_depth=True, refine=True):
        self._ck(self._L.tsdf_set_preprocess(self._c, int(filter_textures), int(processed_depth), int(refine)))

    def processTextures(self): self._ck(self._L.tsdf_process_textures(self._c))

    def preprocessed(self, lab=True):
        """the products of processTextures(); lab=False leaves out the Lab image, which the library produces on request from the processed frame's inputs
        (an error once a newer raw frame has been uploaded)"""
        n, h, w = self._pp_shape
        out = dict(depth2=np.zeros((n, h, w), np.float32), depth_rg=np.zeros((n, h, w, 2), np.float32), lab=np.zeros((n, h, w, 3), np.float32),
                   depth_b=np.zeros((n, h, w, 2), np.float32), silhouette=np.zeros((n, h, w), np.float32),
                   normals=np.zeros((n, h, w, 3), np.float32), quality=np.zeros((n, h, w), np.float32))
        if not lab:
            del out["lab"]
        self._ck(self._L.tsdf_download_preprocessed(self._c, _fp(out["depth2"]), _fp(out["depth_rg"]), _fp(out["lab"]) if lab else None, _fp(out["depth_b"]),
                                                    _fp(out["silhouette"]), _fp(out["normals"]), _fp(out["quality"])))
        return out

    # ------------------------------------------------------------------ reference operator surface
    def clearOccupiedBricks(self): self._ck(self._L.tsdf_clear_bricks(self._c))
    def markBricks(self): self._ck(self._L.tsdf_mark_bricks(self._c))

    def updateOccupiedBricks(self, want_ratio=True):
        if not want_ratio:
            self._ck(self._L.tsdf_update_occupied(self._c, None))
            return None
        r = C.c_float()
        self._ck(self._L.tsdf_update_occupied(self._c, C.byref(r)))
        return r.value

    def integrate(self): self._ck(self._L.tsdf_integrate(self._c))
    def draw(self, mv, proj): self._ck(self._L.tsdf_raymarch(self._c, _fp(_f32(mv)), _fp(_f32(proj))))
    # kinect::ReconPoints (recon_points.cpp): the point back-end
    def upload_normals(self, normals):
        self._ck(self._L.tsdf_upload_normals(self._c, _fp(_f32(normals))))

    def drawPoints(self, mv, proj):
        self._ck(self._L.tsdf_draw_points(self._c, _fp(_f32(mv)), _fp(_f32(proj))))

    # kinect::ReconTrigrid (recon_trigrid.cpp): the triangle-grid back-end
    def setMinLength(self, v): self._ck(self._L.tsdf_set_min_length(self._c, C.c_float(v)))

    def drawTrigrid(self, mv, proj):
        self._ck(self._L.tsdf_draw_trigrid(self._c, _fp(_f32(mv)), _fp(_f32(proj))))

    def fillColors(self): self._ck(self._L.tsdf_fill_colors(self._c))
    def drawF(self, mv, proj): self._ck(self._L.tsdf_draw_f(self._c, _fp(_f32(mv)), _fp(_f32(proj))))

    def frame_dev(self, mv, proj, new_frame=None, complete=True):
        """one frame of the client's loop in ONE call (tsdf_frame_dev): [the re-layout of a frame that arrived in device memory -- new_frame =
        (depth_rg, quality, silhouette[, colour]) device pointers --,] clear / mark / update of the bricks, integrate, drawF"""
        d, q, s, col = (tuple(new_frame) + (0,))[:4] if new_frame is not None else (0, 0, 0, 0)
        self._ck(self._L.tsdf_frame_dev(self._c, C.c_void_p(d), C.c_void_p(q), C.c_void_p(s), C.c_void_p(col), C.c_uint32(1 if complete else 0),
                                        _fp(_f32(mv)), _fp(_f32(proj))))

    def upload_raw_frame_dev(self, depth_raw_ptr, colour_ptr, complete=False):
        """the RAW frame (depth in metres [N][H][W] float32, colour RGB8) is in device memory already: processTextures() reads it where it lies"""
        self._ck(self._L.tsdf_upload_raw_frame_dev(self._c, C.c_void_p(int(depth_raw_ptr)), C.c_void_p(int(colour_ptr)), C.c_uint32(1 if complete else 0)))

    def frame_raw_dev(self, mv, proj, new_frame=None, complete=True):
        """one frame of the client's loop from the RAW frame in ONE call (tsdf_frame_raw_dev): [new_frame = (depth_raw, colour) device pointers,]
        clearOccupiedBricks, processTextures (marks the bricks), updateOccupiedBricks, integrate, drawF"""
        d, col = tuple(new_frame) if new_frame is not None else (0, 0)
        self._ck(self._L.tsdf_frame_raw_dev(self._c, C.c_void_p(d), C.c_void_p(col), C.c_uint32(1 if complete else 0), _fp(_f32(mv)), _fp(_f32(proj))))

    def setTsdfLimit(self, v): self._ck(self._L.tsdf_set_tsdf_limit(self._c, C.c_float(v)))

    def setVoxelSize(self, size):
        self._ck(self._L.tsdf_set_voxel_size(self._c, C.c_float(size)))
        r3, b3, s3 = (C.c_uint32 * 3)(), (C.c_uint32 * 3)(), (C.c_float * 3)()
        self._ck(self._L.tsdf_get_resolution(self._c, r3, b3, s3))
        self.res, self.res_bricks, self.brick_size = tuple(r3), tuple(b3), tuple(s3)
    def setUseBricks(self, a): self._ck(self._L.tsdf_set_use_bricks(self._c, int(a)))
    def setSpaceSkip(self, a): self._ck(self._L.tsdf_set_space_skip(self._c, int(a)))
    def setColorFilling(self, a): self._ck(self._L.tsdf_set_color_filling(self._c, int(a)))
    def setMinVoxelsPerBrick(self, n): self._ck(self._L.tsdf_set_min_voxels_per_brick(self._c, int(n)))
    def setMarchCap(self, samples): self._ck(self._L.tsdf_set_march_cap(self._c, C.c_uint32(int(samples))))
    def setShadeMode(self, m): self._ck(self._L.tsdf_set_shade_mode(self._c, int(m)))
    # stereo modes of the client (kinect_client.cpp:616-669): Reconstruction::setViewportOffset / setColorMaskMode + the GL state around them
    def setViewportOffset(self, x, y): self._ck(self._L.tsdf_set_viewport_offset(self._c, C.c_float(x), C.c_float(y)))
    def setViewportOrigin(self, x, y): self._ck(self._L.tsdf_set_viewport_origin(self._c, int(x), int(y)))
    def setColorMaskMode(self, m): self._ck(self._L.tsdf_set_color_mask_mode(self._c, int(m)))
    def setFramebufferClear(self, clear_color): self._ck(self._L.tsdf_set_framebuffer_clear(self._c, int(bool(clear_color))))

    def setBrickSize(self, size):
        s = (C.c_float * 3)(*([size] * 3 if np.isscalar(size) else list(size)))
        self._ck(self._L.tsdf_set_brick_size(self._c, s))
        r3, b3, s3 = (C.c_uint32 * 3)(), (C.c_uint32 * 3)(), (C.c_float * 3)()
        self._ck(self._L.tsdf_get_resolution(self._c, r3, b3, s3))
        self.res_bricks, self.brick_size = tuple(b3), tuple(s3)

    def resize(self, w, h):
        self._ck(self._L.tsdf_resize(self._c, w, h))
        self.view = (w, h)
        n = C.c_uint32()
        self._ck(self._L.tsdf_num_lods(self._c, C.byref(n)))
        self.num_lods = n.value

    def occupiedRatio(self):
        r = C.c_float()
        self._ck(self._L.tsdf_occupied_ratio(self._c, C.byref(r)))
        return r.value

    def getBrickSize(self): return self.brick_size

    def numBricks(self):
        n = C.c_uint32()
        self._ck(self._L.tsdf_num_bricks(self._c, C.byref(n)))
        return n.value

    # ------------------------------------------------------------------ state transfer (tests, harness)
    def tsdf(self):
        out = np.empty((self.res[2], self.res[1], self.res[0]), np.float32)
        self._ck(self._L.tsdf_download_volume(self._c, _fp(out)))
        return out

    def set_tsdf(self, v):
        v = _f32(v)
        assert v.size == self.res[0] * self.res[1] * self.res[2]
        self._ck(self._L.tsdf_upload_volume(self._c, _fp(v)))

    def bricks(self):
        n = self.numBricks()
        cnt, fl = np.empty(n, np.uint32), np.empty(n, np.uint8)
        self._ck(self._L.tsdf_download_bricks(self._c, cnt.ctypes.data_as(C.POINTER(C.c_uint32)), fl.ctypes.data_as(C.POINTER(C.c_uint8))))
        return cnt, fl

    def active_tiles(self):
        """(tile coordinates [n][3] (x, y, z in units of 8 voxels), number of integrated tiles) of the last culled integrate()"""
        n, grid = C.c_uint32(), (C.c_uint32 * 4)()
        self._ck(self._L.tsdf_download_active_tiles(self._c, None, 0, C.byref(n), grid))
        ids = np.zeros(max(1, n.value), np.uint32)
        self._ck(self._L.tsdf_download_active_tiles(self._c, ids.ctypes.data_as(C.POINTER(C.c_uint32)), ids.size, C.byref(n), grid))
        ids = ids[:n.value].astype(np.int64)
        ntx, nty, tz0 = int(grid[0]), int(grid[1]), int(grid[2])
        return np.stack([ids % ntx, (ids // ntx) % nty, tz0 + ids // (ntx * nty)], -1), int(grid[3])

    def set_counters(self, a):
        a = np.ascontiguousarray(a, np.uint32)
        assert a.size == self.numBricks()
        self._ck(self._L.tsdf_upload_brick_counters(self._c, a.ctypes.data_as(C.POINTER(C.c_uint32))))

    def view_images(self):
        w, h = self.view
        rgba, d, ns, pe = np.empty((h, w, 4), np.float32), np.empty((h, w), np.float32), np.empty((h, w), np.float32), np.empty((h, w, 4), np.float32)
        self._ck(self._L.tsdf_download_image(self._c, _fp(rgba), _fp(d), _fp(ns), _fp(pe)))
        return rgba, d, ns, pe

    def set_view_images(self, rgba, depth):
        self._ck(self._L.tsdf_upload_image(self._c, _fp(_f32(rgba)), _fp(_f32(depth))))

    def framebuffer(self):
        w, h = self.view
        rgba, d = np.empty((h, w, 4), np.float32), np.empty((h, w), np.float32)
        self._ck(self._L.tsdf_download_framebuffer(self._c, _fp(rgba), _fp(d)))
        return rgba, d

    def atlas(self):
        w, h = self.view
        aw = int(np.float32(w) * np.float32(1.5))
        rgba, d = np.empty((h, aw, 4), np.float32), np.empty((h, aw), np.float32)
        self._ck(self._L.tsdf_download_atlas(self._c, _fp(rgba), _fp(d)))
        return rgba, d

    # ------------------------------------------------------------------ timers / multi-GPU hooks
    def enable_timers(self, on=True): self._ck(self._L.tsdf_enable_timers(self._c, int(on)))

    def timer_ms(self, name):
        ms = C.c_float()
        self._ck(self._L.tsdf_timer_ms(self._c, name.encode(), C.byref(ms)))
        return ms.value

    def set_timer_filter(self, names=None):
        self._ck(self._L.tsdf_set_timer_filter(self._c, (",".join(names)).encode() if names else None))

    def sparse_pool_stats(self):
        need, cap = C.c_uint32(), C.c_uint32()
        self._ck(self._L.tsdf_sparse_pool_stats(self._c, C.byref(need), C.byref(cap)))
        return need.value, cap.value

    def integrate_stats(self):
        """dict of the last integrate() launch: work items (tiles), tiles served from the projection cache, (tile, stream) pairs of those
        evaluated per voxel, tiles taken by the LUT kernel, cache slots in use / capacity (all 0 without the cache)"""
        out = (C.c_uint32 * 6)()
        self._ck(self._L.tsdf_integrate_stats(self._c, out))
        return dict(zip(("items", "cached", "full_pairs", "lut_items", "slots_used", "slots"), [int(v) for v in out]))

    def fill_stats(self):
        """(hole-filling passes so far, of them restricted to the dirty screen tiles)"""
        out = (C.c_uint64 * 2)()
        self._ck(self._L.tsdf_fill_stats(self._c, out))
        return int(out[0]), int(out[1])

    def timer_reserve(self, name, n): self._ck(self._L.tsdf_timer_reserve(self._c, name.encode(), int(n)))

    def timer_begin(self, name): self._ck(self._L.tsdf_timer_begin(self._c, name.encode()))
    def timer_end(self, name): self._ck(self._L.tsdf_timer_end(self._c, name.encode()))
    def timer_end_after_fill(self, name): self._ck(self._L.tsdf_timer_end_after_fill(self._c, name.encode()))
    def set_stage_overlap(self, on): self._ck(self._L.tsdf_set_stage_overlap(self._c, int(bool(on))))

    def timer_samples(self, name, capacity=8192):
        out, n = np.zeros(capacity, np.float32), C.c_uint32()
        self._ck(self._L.tsdf_timer_samples(self._c, name.encode(), _fp(out), capacity, C.byref(n)))
        return out[:n.value].copy()

    def timer_spans(self, name, origin, capacity=8192):
        """(begin, end) in ms after the first begin of timer `origin`, for every invocation of `name` since the last reset"""
        b, e, n = np.zeros(capacity, np.float32), np.zeros(capacity, np.float32), C.c_uint32()
        self._ck(self._L.tsdf_timer_spans(self._c, name.encode(), origin.encode(), _fp(b), _fp(e), capacity, C.byref(n)))
        return b[:n.value].copy(), e[:n.value].copy()

    def timer_stats(self, name):
        n, ms = C.c_uint32(), C.c_float()
        self._ck(self._L.tsdf_timer_stats(self._c, name.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    # ------------------------------------------------------------------ native RCCL exchange (comm.cpp): one communicator per context
    @staticmethod
    def comm_unique_id():
        """128 bytes made by rank 0 (ncclGetUniqueId); the caller carries them to the other ranks"""
        L = load_library()
        buf = (C.c_uint8 * 128)()
        rc = L.tsdf_comm_unique_id(buf)
        if rc != 0:
            raise TsdfError(rc, "RCCL is not available in this process")
        return bytes(buf)

    def comm_init(self, unique_id, rank, world, dedicated_compositor=False):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._ck(self._L.tsdf_comm_init(self._c, buf, int(rank), int(world), 1 if dedicated_compositor else 0))

    def comm_destroy(self): self._ck(self._L.tsdf_comm_destroy(self._c))

    def broadcast_frame(self, root, scene=None, dev_ptrs=None):
        """on the root the frame as it arrived -- scene: host arrays, or dev_ptrs: (depth, quality, silhouette, colour) device pointers --, elsewhere nothing"""
        if dev_ptrs is not None:
            self._ck(self._L.tsdf_broadcast_frame(self._c, int(root), *[C.c_void_p(int(p)) for p in dev_ptrs]))
            return
        if scene is None:
            self._ck(self._L.tsdf_broadcast_frame(self._c, int(root), None, None, None, None))
            return
        col = np.ascontiguousarray(scene["color"], np.uint8)
        self._ck(self._L.tsdf_broadcast_frame(self._c, int(root), _fp(_f32(scene["depth"])), _fp(_f32(scene["quality"])), _fp(_f32(scene["silhouette"])),
                                              col.ctypes.data_as(C.POINTER(C.c_uint8))))

    def halo_exchange(self): self._ck(self._L.tsdf_halo_exchange(self._c))
    def composite_gather(self): self._ck(self._L.tsdf_composite_gather(self._c))

    def composite_finish(self):
        r = C.c_uint32()
        self._ck(self._L.tsdf_composite_finish(self._c, C.byref(r)))
        return bool(r.value)

    def comm_set_capacity_limits(self, min_records, max_records=0): self._ck(self._L.tsdf_comm_set_capacity_limits(self._c, int(min_records), int(max_records)))

    def comm_frame_status(self, frame):
        """True: frame `frame` of the native compact composite was built from truncated record lists and not repaired (tsdf_comm_frame_status)"""
        t = C.c_int32()
        self._ck(self._L.tsdf_comm_frame_status(self._c, C.c_uint64(int(frame)), C.byref(t)))
        return bool(t.value)

    def comm_stats(self):
        a, b = C.c_uint32(), C.c_uint32()
        self._ck(self._L.tsdf_comm_stats(self._c, C.byref(a), C.byref(b)))
        return {"regathers": a.value, "overflowed_frames": b.value}

    def halo_info(self):
        layers, nbytes = C.c_uint32(), C.c_uint64()
        self._ck(self._L.tsdf_halo_info(self._c, C.byref(layers), C.byref(nbytes)))
        return layers.value, nbytes.value

    def halo_pack_dev(self, lo_ptr, hi_ptr):
        self._ck(self._L.tsdf_halo_pack_dev(self._c, C.c_void_p(lo_ptr), C.c_void_p(hi_ptr)))

    def halo_unpack_dev(self, below_ptr, above_ptr):
        self._ck(self._L.tsdf_halo_unpack_dev(self._c, C.c_void_p(below_ptr), C.c_void_p(above_ptr)))

    def export_partial_dev(self, dst_ptr):
        self._ck(self._L.tsdf_export_partial_dev(self._c, C.c_void_p(dst_ptr)))

    def export_hits_dev(self, dst_ptr, capacity):
        self._ck(self._L.tsdf_export_hits_dev(self._c, C.c_void_p(dst_ptr), C.c_uint32(capacity)))

    def composite_hits_dev(self, gathered_ptr, n, stride_bytes):
        self._ck(self._L.tsdf_composite_hits_dev(self._c, C.c_void_p(gathered_ptr), C.c_uint32(n), C.c_uint64(stride_bytes)))

    def composite_dev(self, gathered_ptr, n):
        self._ck(self._L.tsdf_composite_dev(self._c, C.c_void_p(gathered_ptr), int(n)))
