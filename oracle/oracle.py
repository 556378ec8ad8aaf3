"""ctypes front-end of the CPU oracle (oracle/tsdf_oracle.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (rgbd-recon_amd/) never imports this module.
PARITY UNPINNED (see the header of tsdf_oracle.cpp).

Method names follow kinect::ReconIntegration (framework/reconstruction/recon_integration.hpp:38-64).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("RGBDR_ORACLE_LIB", os.path.join(_HERE, "libtsdf_oracle.so"))   # override: the sanitizer build (tests/test_oracle_sanitizers.py)


class OrcConfig(C.Structure):
    _fields_ = [("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3), ("voxel_size", C.c_float),
                ("res", C.c_uint32 * 3), ("brick_size", C.c_float * 3), ("limit", C.c_float),
                ("num_streams", C.c_uint32), ("depth_w", C.c_uint32), ("depth_h", C.c_uint32),
                ("color_w", C.c_uint32), ("color_h", C.c_uint32), ("view_w", C.c_uint32), ("view_h", C.c_uint32)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        L.orc_create.restype = C.c_void_p
        L.orc_update_occupied.restype = C.c_float
        L.orc_num_bricks.restype = C.c_uint32
        L.orc_num_occupied.restype = C.c_uint32
        L.orc_tsdf.restype = C.POINTER(C.c_float)
        L.orc_tex2d_nearest.restype = C.c_float
        _lib = L
    return _lib


def set_threads(n):
    """OpenMP threads of the oracle's parallel loops from now on (bench.py: 1-thread and all-cores CPU baselines); returns the previous maximum"""
    return int(lib().orc_set_threads(int(n)))


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


class OracleRecon:
    def __init__(self, scene, res=None, voxel_size=0.01, brick_size=0.1, limit=0.01, view=(1280, 720)):
        L = lib()
        cfg = OrcConfig()
        cfg.bbox_min[:] = [float(x) for x in scene["bbox_min"]]
        cfg.bbox_max[:] = [float(x) for x in scene["bbox_max"]]
        cfg.voxel_size = voxel_size
        cfg.res[:] = list(res) if res is not None else [0, 0, 0]
        bs = [brick_size] * 3 if np.isscalar(brick_size) else list(brick_size)
        assert all(b > 0 for b in bs)
        cfg.brick_size[:] = bs
        cfg.limit = limit
        cfg.num_streams = scene["n"]
        cfg.depth_w, cfg.depth_h = scene["width"], scene["height"]
        cfg.color_w, cfg.color_h = scene["color_width"], scene["color_height"]
        cfg.view_w, cfg.view_h = view
        self._L = L
        self._c = C.c_void_p(L.orc_create(C.byref(cfg)))
        self.scene = scene
        self.view = tuple(view)
        res3, rb3, b3, nl = (C.c_uint32 * 3)(), (C.c_uint32 * 3)(), (C.c_float * 3)(), C.c_uint32()
        L.orc_get_layout(self._c, res3, rb3, b3, C.byref(nl))
        self.res, self.res_bricks, self.brick_size, self.num_lods = tuple(res3), tuple(rb3), tuple(b3), nl.value
        self._keep = []
        for i in range(scene["n"]):
            inv, uv, xyz = _f32(scene["cv_xyz_inv"][i]), _f32(scene["cv_uv"][i]), _f32(scene["cv_xyz"][i])
            self._keep += [inv, uv, xyz]
            ri = (C.c_uint32 * 3)(*[int(x) for x in scene["inv_res"]])
            rl = (C.c_uint32 * 3)(*[int(x) for x in scene["lut_res"]])
            L.orc_set_calibration(self._c, i, _p(inv), ri, _p(uv), rl, _p(xyz), rl)
        self.upload_frame(scene)
        self.flags = dict(use_bricks=1, skip_space=1, fill_holes=1, min_voxels=10, shade_mode=0)
        self._apply_flags()
        self.stereo = dict(org=(0, 0), off=(0.0, 0.0), mask=0, clear=1)

    def __del__(self):
        if getattr(self, "_c", None):
            self._L.orc_destroy(self._c)
            self._c = None

    def upload_frame(self, scene):
        d, q, s, col = _f32(scene["depth"]), _f32(scene["quality"]), _f32(scene["silhouette"]), np.ascontiguousarray(scene["color"], np.uint8)
        self._frame = (d, q, s, col)
        self._L.orc_set_frame(self._c, _p(d), _p(q), _p(s), _p(col, C.c_uint8))

    def _apply_flags(self):
        f = self.flags
        self._L.orc_set_flags(self._c, f["use_bricks"], f["skip_space"], f["fill_holes"], f["min_voxels"], f["shade_mode"])

    # --- reference operator surface
    def setUseBricks(self, a): self.flags["use_bricks"] = int(a); self._apply_flags()
    def setSpaceSkip(self, a): self.flags["skip_space"] = int(a); self._apply_flags()
    def setColorFilling(self, a): self.flags["fill_holes"] = int(a); self._apply_flags()
    def setMinVoxelsPerBrick(self, n): self.flags["min_voxels"] = int(n); self._apply_flags()
    def setShadeMode(self, m): self.flags["shade_mode"] = int(m); self._apply_flags()
    def setTsdfLimit(self, v): self._L.orc_set_limit(self._c, C.c_float(v))

    def setVoxelSize(self, size):
        self._L.orc_set_voxel_size(self._c, C.c_float(size))
        res3, rb3, b3, nl = (C.c_uint32 * 3)(), (C.c_uint32 * 3)(), (C.c_float * 3)(), C.c_uint32()
        self._L.orc_get_layout(self._c, res3, rb3, b3, C.byref(nl))
        self.res, self.res_bricks, self.brick_size = tuple(res3), tuple(rb3), tuple(b3)

    # stereo modes of the client (kinect_client.cpp:616-669)
    def _apply_stereo(self):
        s = self.stereo
        self._L.orc_set_stereo(self._c, int(s["org"][0]), int(s["org"][1]), C.c_float(s["off"][0]), C.c_float(s["off"][1]), int(s["mask"]), int(s["clear"]))

    def setViewportOffset(self, x, y): self.stereo["off"] = (float(x), float(y)); self._apply_stereo()
    def setViewportOrigin(self, x, y): self.stereo["org"] = (int(x), int(y)); self._apply_stereo()
    def setColorMaskMode(self, m): self.stereo["mask"] = int(m); self._apply_stereo()
    def setFramebufferClear(self, clear_color): self.stereo["clear"] = int(bool(clear_color)); self._apply_stereo()
    def clearOccupiedBricks(self): self._L.orc_clear_occupied(self._c)
    def markBricks(self): self._L.orc_mark_bricks(self._c)
    def updateOccupiedBricks(self): return self._L.orc_update_occupied(self._c)
    def integrate(self): self._L.orc_integrate(self._c)

    # --- NetKinectArray side (framework/NetKinectArray.cpp): raw frame -> processTextures()
    def upload_raw_frame(self, scene):
        raw, col = _f32(scene["depth_raw"]), np.ascontiguousarray(scene["color"], np.uint8)
        self._raw = (raw, col)
        self._L.orc_set_raw_frame(self._c, _p(raw), _p(col, C.c_uint8))
        for i in range(scene["n"]):
            self._L.orc_set_depth_limits(self._c, i, C.c_float(float(scene["depth_limits"][0])), C.c_float(float(scene["depth_limits"][1])))
            self._L.orc_set_camera_position(self._c, i, _p(_f32(scene["camera_positions"][i])))

    # --- frame ingest (readLoop / readFromFiles / update(), NetKinectArray.cpp:482-529, :709-749, :225-236)
    def setDepthCompression(self, stream, compress, near, far):
        self._L.orc_set_depth_compression(self._c, stream, int(compress), C.c_float(near), C.c_float(far))

    def upload_wire_frame(self, message, color_format=0, depth_format=0):
        """message: bytes of one ZMQ message / one record per .stream file, per sensor [colour][depth]."""
        sc = self.scene
        n, w, h, cw, ch = sc["n"], sc["width"], sc["height"], sc["color_width"], sc["color_height"]
        ts, cols, deps = wire_split(message, n, *wire_sizes(w, h, cw, ch, color_format, depth_format))
        col = np.zeros((n, ch, cw, 3), np.uint8)
        raw = np.zeros((n, h, w), np.float32)
        for i in range(n):
            col[i] = cols[i].reshape(ch, cw, 3) if color_format == 0 else decode_dxt(cols[i], cw, ch, color_format)[..., :3]
            if depth_format == 0:
                raw[i] = deps[i].view(np.float32).reshape(h, w)
            else:   # GL_LUMINANCE from GL_UNSIGNED_BYTE: normalised, c / 255 (NetKinectArray.cpp:171)
                raw[i] = (deps[i].astype(np.float32) / np.float32(255.0)).reshape(h, w)
        self._raw = (np.ascontiguousarray(raw), np.ascontiguousarray(col))
        self._L.orc_set_raw_frame(self._c, _p(self._raw[0]), _p(self._raw[1], C.c_uint8))
        return ts

    def setPreprocess(self, filter_textures=True, processed_depth=True, refine=True):
        self._L.orc_set_preprocess_flags(self._c, int(filter_textures), int(processed_depth), int(refine))

    def processTextures(self): self._L.orc_process_textures(self._c)

    def preprocessed(self):
        n, h, w = self.scene["n"], self.scene["height"], self.scene["width"]
        out = dict(depth2=np.zeros((n, h, w), np.float32), depth_rg=np.zeros((n, h, w, 2), np.float32), lab=np.zeros((n, h, w, 3), np.float32),
                   depth_b=np.zeros((n, h, w, 2), np.float32), silhouette=np.zeros((n, h, w), np.float32),
                   normals=np.zeros((n, h, w, 3), np.float32), quality=np.zeros((n, h, w), np.float32))
        self._L.orc_get_preprocessed(self._c, _p(out["depth2"]), _p(out["depth_rg"]), _p(out["lab"]), _p(out["depth_b"]), _p(out["silhouette"]),
                                     _p(out["normals"]), _p(out["quality"]))
        return out

    def draw(self, mv, proj):
        self._L.orc_draw(self._c, _p(_f32(mv)), _p(_f32(proj)))

    def fillColors(self): self._L.orc_fill_colors(self._c)

    # --- kinect::ReconPoints (recon_points.cpp): the point back-end behind the same Reconstruction interface
    def upload_normals(self, normals):
        self._normals = _f32(normals)
        self._L.orc_set_normals(self._c, _p(self._normals))

    def drawPoints(self, mv, proj):
        self._L.orc_draw_points(self._c, _p(_f32(mv)), _p(_f32(proj)))

    # --- kinect::ReconTrigrid (recon_trigrid.cpp): the triangle-grid back-end
    def setMinLength(self, v): self._L.orc_set_min_length(self._c, C.c_float(v))

    def drawTrigrid(self, mv, proj):
        self._L.orc_draw_trigrid(self._c, _p(_f32(mv)), _p(_f32(proj)))

    def drawF(self, mv, proj):
        self.draw(mv, proj)
        if self.flags["fill_holes"]:
            self.fillColors()

    # --- downloads
    def numBricks(self): return self._L.orc_num_bricks(self._c)

    def counters(self):
        a = np.zeros(self.numBricks(), np.uint32)
        self._L.orc_get_counters(self._c, _p(a, C.c_uint32))
        return a

    def set_counters(self, a):
        a = np.ascontiguousarray(a, np.uint32)
        assert a.size == self.numBricks()
        self._L.orc_set_counters(self._c, _p(a, C.c_uint32))

    def occupied(self):
        a = np.zeros(self._L.orc_num_occupied(self._c), np.uint32)
        if a.size:
            self._L.orc_get_occupied(self._c, _p(a, C.c_uint32))
        return a

    def brick_ranges(self):
        a = np.zeros((self.numBricks(), 6), np.uint32)
        self._L.orc_get_brick_ranges(self._c, _p(a, C.c_uint32))
        return a

    def tsdf(self):
        n = self.res[0] * self.res[1] * self.res[2]
        return np.ctypeslib.as_array(self._L.orc_tsdf(self._c), shape=(n,)).reshape(self.res[2], self.res[1], self.res[0]).copy()

    def set_tsdf(self, v):
        v = _f32(v)
        assert v.size == self.res[0] * self.res[1] * self.res[2]
        self._L.orc_set_tsdf(self._c, _p(v))

    def view_images(self):
        w, h = self.view
        rgba, d, ns, pe = np.zeros((h, w, 4), np.float32), np.zeros((h, w), np.float32), np.zeros((h, w), np.float32), np.zeros((h, w, 4), np.float32)
        self._L.orc_get_view(self._c, _p(rgba), _p(d), _p(ns), _p(pe))
        return rgba, d, ns, pe

    def set_view_images(self, rgba, depth):
        self._L.orc_set_view(self._c, _p(_f32(rgba)), _p(_f32(depth)))

    def framebuffer(self):
        w, h = self.view
        rgba, d = np.zeros((h, w, 4), np.float32), np.zeros((h, w), np.float32)
        self._L.orc_get_framebuffer(self._c, _p(rgba), _p(d))
        return rgba, d

    def atlas(self):
        w, h = self.view
        aw = int(np.float32(w) * np.float32(1.5))
        rgba, d = np.zeros((h, aw, 4), np.float32), np.zeros((h, aw), np.float32)
        self._L.orc_get_atlas(self._c, _p(rgba), _p(d))
        return rgba, d

    def lod_tables(self):
        off, res = np.zeros((self.num_lods, 2), np.uint32), np.zeros((self.num_lods, 2), np.uint32)
        self._L.orc_get_lod_tables(self._c, _p(off, C.c_uint32), _p(res, C.c_uint32))
        return off, res

    def view_matrices(self, mv, proj):
        """(image_to_eye, NormalMatrix, CameraPos) of draw(), recon_integration.cpp:182-205."""
        m = self.view_matrices_all(mv, proj)
        return m["image_to_eye"], m["normal_matrix"], m["camera_pos"]

    def view_matrices_all(self, mv, proj):
        """The whole matrix block of draw() (+ vol_to_world, :66-72) -> dict like rgbd_recon_amd.view_matrices."""
        out = np.zeros(51, np.float32)
        self._L.orc_view_matrices(self._c, _p(_f32(mv)), _p(_f32(proj)), _p(out))
        return {"vol_to_world": out[:16].copy(), "image_to_eye": out[16:32].copy(), "normal_matrix": out[32:48].copy(), "camera_pos": out[48:].copy()}


# --- frame ingest helpers
def wire_sizes(w, h, cw, ch, color_format, depth_format):
    """(m_colorsize, m_depthsize), NetKinectArray::init, NetKinectArray.cpp:118-139."""
    cs = {0: cw * ch * 3, 1: cw * ch // 2, 5: cw * ch}[color_format]        # RGB8 / DXT1 (fastdxt w*h*4/8) / DXT5 (307200 at 640x480)
    ds = w * h * (1 if depth_format else 4)
    return cs, ds


def wire_split(message, n, cs, ds):
    """readLoop(), NetKinectArray.cpp:513-523: the timestamp is the first 8 bytes OF the first colour image (offset starts at 0)."""
    m = np.frombuffer(message, np.uint8)
    assert m.size == (cs + ds) * n
    ts = float(m[:8].view(np.float64)[0])
    cols = [m[i * (cs + ds): i * (cs + ds) + cs] for i in range(n)]
    deps = [m[i * (cs + ds) + cs: (i + 1) * (cs + ds)] for i in range(n)]
    return ts, cols, deps


def decode_dxt(blocks, w, h, fmt):
    b = np.ascontiguousarray(blocks, np.uint8)
    out = np.zeros((h, w, 4), np.uint8)
    lib().orc_decode_dxt(_p(b, C.c_uint8), w, h, fmt, _p(out, C.c_uint8))
    return out


# --- inverse calibration volumes
def frustum(cv_xyz):
    """cv_xyz [rz][ry][rx][3] -> (planes [6][4], camera position [3]); kinect::Frustum, frustum.cpp."""
    v = _f32(cv_xyz)
    res = (C.c_uint32 * 3)(v.shape[2], v.shape[1], v.shape[0])
    planes, cam = np.zeros((6, 4), np.float32), np.zeros(3, np.float32)
    lib().orc_frustum(_p(v), res, _p(planes), _p(cam))
    return planes, cam


def invert_calibration(cv_xyz, bbox_min, bbox_max, res_inv):
    """CalibrationInverter::calculateInverseVolumes for one sensor -> [rz][ry][rx][4] (brute-force 8-NN: small sizes only)."""
    v = _f32(cv_xyz)
    res = (C.c_uint32 * 3)(v.shape[2], v.shape[1], v.shape[0])
    ri = (C.c_uint32 * 3)(*[int(x) for x in res_inv])
    out = np.zeros((res_inv[2], res_inv[1], res_inv[0], 4), np.float32)
    lib().orc_invert_calibration(_p(v), res, _p(_f32(bbox_min)), _p(_f32(bbox_max)), ri, _p(out))
    return out


# --- sampling primitives (unit tests)
def tex3d(t, u, v, w):
    t = _f32(t)
    nz, ny, nx, nc = t.shape
    out = np.zeros(nc, np.float32)
    lib().orc_tex3d(_p(t), nc, (C.c_uint32 * 3)(nx, ny, nz), C.c_float(u), C.c_float(v), C.c_float(w), _p(out))
    return out


def tex2d_linear(t, layer, u, v):
    t = _f32(t)
    nl, h, w, nc = t.shape
    out = np.zeros(nc, np.float32)
    lib().orc_tex2d_linear(_p(t), nc, w, h, layer, C.c_float(u), C.c_float(v), _p(out))
    return out


def tex2d_nearest(t, layer, u, v, ch=0):
    t = _f32(t)
    nl, h, w, nc = t.shape
    return lib().orc_tex2d_nearest(_p(t), nc, w, h, layer, C.c_float(u), C.c_float(v), ch)
