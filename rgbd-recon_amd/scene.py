"""Synthetic calibrated RGB-D scene (harness input, not part of the hot path).

The reference ships no calibration or stream fixtures, so tests, bench.py and
smoke() all feed the path from this closed-form generator:

* N pinhole depth cameras on a circle (radius 2.5 m, height 1.1 m) looking at (0, 1.1, 0)
* forward LUT  ``cv_xyz``     RGB32F  (u, v, d_norm)   -> world xyz   (framework/calibration/CalibVolumes.cpp:132-144)
* colour LUT   ``cv_uv``      RG32F   identity on (u, v)
* inverse LUT  ``cv_xyz_inv`` RGBA32F unit-cube voxel -> (u, v, d_norm, 1), (-1,-1,-1,-1) outside the
  frustum (framework/calibration/calibration_inverter.cpp:87-101)
* raw sensor input for the pre-processing passes: ``depth_raw`` R32F metres (1 % random holes), ``color`` RGB8,
  ``camera_positions`` (CalibVolumes::getCameraPositions)
* per stream images in the formats NetKinectArray hands to the path (framework/NetKinectArray.cpp:159-188):
  depth RG32F (r = normalised depth, 0 = invalid), quality R32F, silhouette R32F, normals RGB32F, colour RGB8

All LUT volumes are x-fastest (calibration_volume.hpp:57-59).  Everything is computed in float64
and rounded once to float32.
"""
from __future__ import annotations

import numpy as np

BBOX_MIN = np.array([-1.0, 0.0, -1.0])   # source/kinect_client.cpp:206-207 default bbox
BBOX_MAX = np.array([1.0, 2.2, 1.0])
DEPTH_MIN, DEPTH_MAX = 0.5, 4.5          # calibration_inverter.cpp:113, pre_morph.fs:32-33

SPHERE_C = np.array([0.0, 1.1, 0.0])
SPHERE_R = 0.45
BOX_C = np.array([0.5, 0.4, 0.3])
BOX_H = 0.15


# ----------------------------------------------------------------------------- matrices
def look_at(eye, target, up=(0.0, 1.0, 0.0)) -> np.ndarray:
    """gluLookAt; returns a 4x4 (row, col) float64 matrix."""
    eye = np.asarray(eye, np.float64)
    f = np.asarray(target, np.float64) - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, np.asarray(up, np.float64))
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[:3, 3] = -m[:3, :3] @ eye
    return m


def perspective(fovy_deg, aspect, near, far) -> np.ndarray:
    """gluPerspective (source/kinect_client.cpp:100 uses 50 deg, 0.1, 200)."""
    f = 1.0 / np.tan(np.radians(fovy_deg) * 0.5)
    m = np.zeros((4, 4))
    m[0, 0] = f / aspect
    m[1, 1] = f
    m[2, 2] = (far + near) / (near - far)
    m[2, 3] = 2.0 * far * near / (near - far)
    m[3, 2] = -1.0
    return m


def gl_flat(m: np.ndarray) -> np.ndarray:
    """4x4 (row, col) -> 16 floats column-major, as glGetFloatv(GL_*_MATRIX) returns them."""
    return np.ascontiguousarray(m.T, dtype=np.float32).reshape(16)


def default_view(view_w=1280, view_h=720):
    """Benchmark view of SURVEY.md §8d: eye (0,1.1,3) -> (0,1.1,0), 50 deg, 0.1..200."""
    mv = look_at((0.0, 1.1, 3.0), (0.0, 1.1, 0.0))
    pr = perspective(50.0, view_w / float(view_h), 0.1, 200.0)
    return gl_flat(mv), gl_flat(pr)


# ----------------------------------------------------------------------------- cameras
class Camera:
    def __init__(self, k, n, width, height, focal):
        a = 2.0 * np.pi * k / n
        self.pos = np.array([2.5 * np.cos(a), 1.1, 2.5 * np.sin(a)])
        fwd = np.array([0.0, 1.1, 0.0]) - self.pos
        self.fwd = fwd / np.linalg.norm(fwd)
        r = np.cross(self.fwd, [0.0, 1.0, 0.0])
        self.right = r / np.linalg.norm(r)
        self.up = np.cross(self.right, self.fwd)
        self.w, self.h, self.f = width, height, focal
        self.cx, self.cy = width * 0.5, height * 0.5

    def unproject(self, u, v, d):
        """normalised (u, v) and metric z-depth d -> world xyz (broadcasting)."""
        a = (u * self.w - self.cx) / self.f
        b = (v * self.h - self.cy) / self.f
        return (self.pos + d[..., None] * (self.fwd + a[..., None] * self.right + b[..., None] * self.up))

    def project(self, x):
        """world xyz [...,3] -> (u, v, d)."""
        rel = x - self.pos
        d = rel @ self.fwd
        with np.errstate(divide="ignore", invalid="ignore"):
            a = (rel @ self.right) / d
            b = (rel @ self.up) / d
        return (self.cx + self.f * a) / self.w, (self.cy + self.f * b) / self.h, d


def _ray_sphere(o, d, centre):
    oc = o - centre
    b = np.einsum("...k,...k->...", oc, d)
    c = np.einsum("...k,...k->...", oc, oc) - SPHERE_R ** 2
    a = np.einsum("...k,...k->...", d, d)
    disc = b * b - a * c
    t = np.where(disc >= 0, (-b - np.sqrt(np.maximum(disc, 0))) / a, np.inf)
    return np.where(t > 0, t, np.inf)


def _ray_box(o, d, centre):
    lo, hi = centre - BOX_H, centre + BOX_H
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t0 = (lo - o) * inv
        t1 = (hi - o) * inv
    tn = np.minimum(t0, t1)
    tf = np.maximum(t0, t1)
    tmin = tn.max(-1)
    tmax = tf.min(-1)
    axis = tn.argmax(-1)
    hit = (tmin <= tmax) & (tmin > 0)
    return np.where(hit, tmin, np.inf), axis


def make_scene(n_streams=4, width=640, height=480, lut_res=128, inv_res=128, seed=1234,
               color_width=None, color_height=None, sphere_c=None, box_c=None):
    """Returns a dict of contiguous numpy arrays (see module docstring for formats).  sphere_c / box_c move the two
    objects (a second, different frame for the same calibration)."""
    sphere_c = SPHERE_C if sphere_c is None else np.asarray(sphere_c, np.float64)
    box_c = BOX_C if box_c is None else np.asarray(box_c, np.float64)
    rng = np.random.default_rng(seed)
    focal = 570.0 * width / 640.0
    cw = color_width or width
    ch = color_height or height
    cams = [Camera(k, n_streams, width, height, focal) for k in range(n_streams)]
    out = dict(n=n_streams, width=width, height=height, color_width=cw, color_height=ch,
               bbox_min=BBOX_MIN.astype(np.float32), bbox_max=BBOX_MAX.astype(np.float32),
               depth_limits=np.array([DEPTH_MIN, DEPTH_MAX], np.float32),
               lut_res=np.array([lut_res] * 3, np.uint32), inv_res=np.array([inv_res] * 3, np.uint32))

    # ---- LUTs (texel centres at (i + .5) / n, x fastest)
    c = (np.arange(lut_res) + 0.5) / lut_res
    dn, vv, uu = np.meshgrid(c, c, c, indexing="ij")             # [z=d][y=v][x=u]
    ci = (np.arange(inv_res) + 0.5) / inv_res
    wz, wy, wx = np.meshgrid(ci, ci, ci, indexing="ij")
    world = BBOX_MIN + np.stack([wx, wy, wz], -1) * (BBOX_MAX - BBOX_MIN)
    xyz, uvl, inv = [], [], []
    for cam in cams:
        xyz.append(cam.unproject(uu, vv, DEPTH_MIN + dn * (DEPTH_MAX - DEPTH_MIN)).reshape(-1, 3))
        uvl.append(np.stack([uu, vv], -1).reshape(-1, 2))
        u, v, d = cam.project(world)
        dnorm = (d - DEPTH_MIN) / (DEPTH_MAX - DEPTH_MIN)
        ok = (u >= 0) & (u <= 1) & (v >= 0) & (v <= 1) & (dnorm >= 0) & (dnorm <= 1)
        val = np.stack([u, v, dnorm, np.ones_like(u)], -1)
        val[~ok] = -1.0
        inv.append(val.reshape(-1, 4))
    out["cv_xyz"] = np.ascontiguousarray(np.stack(xyz), np.float32)
    out["cv_uv"] = np.ascontiguousarray(np.stack(uvl), np.float32)
    out["cv_xyz_inv"] = np.ascontiguousarray(np.stack(inv), np.float32)

    # ---- per-stream images
    py, px = np.meshgrid(np.arange(height) + 0.5, np.arange(width) + 0.5, indexing="ij")
    depth = np.zeros((n_streams, height, width, 2), np.float32)
    quality = np.zeros((n_streams, height, width), np.float32)
    silhouette = np.zeros((n_streams, height, width), np.float32)
    normals = np.zeros((n_streams, height, width, 3), np.float32)
    depth_raw = np.zeros((n_streams, height, width), np.float32)
    color = np.zeros((n_streams, ch, cw, 3), np.uint8)
    noise = rng.integers(0, 4, size=(n_streams, ch, cw, 3), dtype=np.uint8)
    drop = rng.random((n_streams, height, width)) < 0.01                      # sensor holes for the morph pass to close (drawn after `noise`: golden fixtures)
    for k, cam in enumerate(cams):
        a = (px - cam.cx) / cam.f
        b = (py - cam.cy) / cam.f
        dirs = cam.fwd + a[..., None] * cam.right + b[..., None] * cam.up      # z-depth parametrised
        o = np.broadcast_to(cam.pos, dirs.shape)
        ts = _ray_sphere(o, dirs, sphere_c)
        tb, axis = _ray_box(o, dirs, box_c)
        t = np.minimum(ts, tb)
        hit = np.isfinite(t)
        tt = np.where(hit, t, 0.0)
        p = o + tt[..., None] * dirs
        nrm_s = (p - sphere_c) / SPHERE_R
        nrm_b = np.zeros_like(p)
        sgn = -np.sign(np.take_along_axis(dirs, axis[..., None], -1))[..., 0]
        np.put_along_axis(nrm_b, axis[..., None], sgn[..., None], -1)
        nrm = np.where((ts <= tb)[..., None], nrm_s, nrm_b)
        inside = np.all((p >= BBOX_MIN) & (p <= BBOX_MAX), -1)
        dnorm = (tt - DEPTH_MIN) / (DEPTH_MAX - DEPTH_MIN)                     # t is z-depth here
        valid = hit & inside & (dnorm > 0) & (dnorm < 1)
        to_cam = -dirs / np.linalg.norm(dirs, axis=-1, keepdims=True)
        cos_t = np.clip(np.einsum("...k,...k->...", nrm, to_cam), 0.0, 1.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            q = np.clip(cos_t ** 2 / (6.5 * dnorm), 0.05, 4.0)                 # shape of pre_quality.fs:107-114
        depth_raw[k] = np.where(hit & ~drop[k], tt, 0.0)                              # metres along the optical axis, 0 = no return
        depth[k, ..., 0] = np.where(valid, dnorm, 0.0)
        quality[k] = np.where(valid, q, 0.0)
        silhouette[k] = valid.astype(np.float32)
        normals[k] = np.where(valid[..., None], nrm, 0.0)
        # colour camera == depth camera (cv_uv identity); resample to the colour resolution
        cyi = np.minimum((np.arange(ch) + 0.5) / ch * height, height - 1).astype(int)
        cxi = np.minimum((np.arange(cw) + 0.5) / cw * width, width - 1).astype(int)
        pc = p[cyi][:, cxi]
        vc = valid[cyi][:, cxi]
        cell = np.floor(pc / 0.1).astype(np.int64).sum(-1) & 1
        base = np.where(cell[..., None] == 1, np.array([200, 60, 40]), np.array([40, 90, 200]))
        base = base + (np.array([0, 8 * (k + 1), 0]))                          # per-stream tint
        col = np.where(vc[..., None], base, 16) + noise[k]
        color[k] = np.clip(col, 0, 255).astype(np.uint8)
    out.update(depth=depth, quality=quality, silhouette=silhouette, normals=normals, color=color, depth_raw=depth_raw,
               camera_positions=np.stack([c.pos for c in cams]).astype(np.float32))
    return out


# ---------------------------------------------------------------------- the SENDER side of the wire (synthetic input only)
# The reference's server (not part of this project) compresses colour with fastdxt and depth to 8 bit; these helpers
# produce well-formed messages for tests and bench.py.  They are plain range-fit encoders, not the reference's.
def _pack565(rgb):
    r, g, b = (rgb[..., 0].astype(np.uint32) >> 3), (rgb[..., 1].astype(np.uint32) >> 2), (rgb[..., 2].astype(np.uint32) >> 3)
    return (r << 11) | (g << 5) | b


def _expand565(v):
    r, g, b = (v >> 11) & 31, (v >> 5) & 63, v & 31
    return np.stack([(r << 3) | (r >> 2), (g << 2) | (g >> 4), (b << 3) | (b >> 2)], -1).astype(np.int32)


def encode_dxt1(rgb: np.ndarray) -> np.ndarray:
    """[h][w][3] uint8 (h, w multiples of 4) -> DXT1 blocks (8 bytes each, four-colour mode), row-major block order."""
    h, w = rgb.shape[:2]
    assert h % 4 == 0 and w % 4 == 0
    blk = rgb.reshape(h // 4, 4, w // 4, 4, 3).transpose(0, 2, 1, 3, 4).reshape(-1, 16, 3).astype(np.int32)
    lum = blk @ np.array([2, 5, 1])
    hi = np.take_along_axis(blk, lum.argmax(1)[:, None, None], 1)[:, 0]
    lo = np.take_along_axis(blk, lum.argmin(1)[:, None, None], 1)[:, 0]
    c0, c1 = _pack565(hi), _pack565(lo)
    swap = c0 < c1
    c0, c1 = np.where(swap, c1, c0), np.where(swap, c0, c1)
    e0, e1 = _expand565(c0), _expand565(c1)
    flat = c0 == c1                                                   # equal endpoints select the three-colour mode: index 0 everywhere
    pal = np.stack([e0, e1, (2 * e0 + e1) // 3, (e0 + 2 * e1) // 3], 1)                       # [B][4][3]
    err = ((blk[:, :, None, :] - pal[:, None, :, :]) ** 2).sum(-1)                             # [B][16][4]
    idx = np.where(flat[:, None], 0, err.argmin(-1)).astype(np.uint32)
    bits = (idx << (2 * np.arange(16, dtype=np.uint32))[None, :]).sum(1).astype(np.uint32)
    out = np.zeros((blk.shape[0], 2), np.uint32)
    out[:, 0] = c0 | (c1 << 16)
    out[:, 1] = bits
    return out.view(np.uint8).reshape(-1)


def encode_dxt5(rgba: np.ndarray) -> np.ndarray:
    """[h][w][4] uint8 -> DXT5 blocks (16 bytes: 8 alpha + the DXT1-style colour block)."""
    h, w = rgba.shape[:2]
    col = encode_dxt1(rgba[..., :3]).reshape(-1, 8)
    a = rgba[..., 3].reshape(h // 4, 4, w // 4, 4).transpose(0, 2, 1, 3).reshape(-1, 16).astype(np.int32)
    a0, a1 = a.max(1), a.min(1)                                       # a0 > a1: seven-step table; a0 == a1: five-step table, codes 0
    tab = np.stack([a0, a1] + [((7 - i) * a0 + i * a1) // 7 for i in range(1, 7)], 1)
    idx = np.where((a0 == a1)[:, None], 0, np.abs(a[:, :, None] - tab[:, None, :]).argmin(-1)).astype(np.uint64)
    bits = (idx << (3 * np.arange(16, dtype=np.uint64))[None, :]).sum(1).astype(np.uint64)
    ab = np.zeros((a.shape[0], 8), np.uint8)
    ab[:, 0], ab[:, 1] = a0, a1
    for k in range(6):
        ab[:, 2 + k] = (bits >> np.uint64(8 * k)) & np.uint64(0xFF)
    return np.concatenate([ab, col], 1).reshape(-1)


def compress_depth_u8(depth_m: np.ndarray, near: float, far: float) -> np.ndarray:
    """Inverse of uncompress() (glsl/pre_depth.fs:51-61): metres -> 8-bit code of the sqrt mapping; 0 = no measurement."""
    scale = np.float32(far - near)
    sn = scale / np.float32(255.0)
    t = (depth_m.astype(np.float32) - np.float32(near)) / scale - np.float32(0.15) * sn
    code = np.rint(np.sqrt(np.clip(t, 0.0, 1.0)) * 255.0)
    return np.where(depth_m > 0, np.clip(code, 5, 255), 0).astype(np.uint8)


def make_wire_message(scene, color_format=0, depth_format=0, timestamp=None, near=0.5, far=4.5) -> bytes:
    """One ZMQ message as NetKinectArray::readLoop expects it (framework/NetKinectArray.cpp:513-523): per sensor
    [colour][depth].  A timestamp, if given, overwrites the first 8 bytes as the reference's sender does."""
    parts = []
    for i in range(scene["n"]):
        col = np.ascontiguousarray(scene["color"][i], np.uint8)
        if color_format == 1:
            parts.append(encode_dxt1(col).tobytes())
        elif color_format == 5:
            a = np.full(col.shape[:2] + (1,), 255, np.uint8)
            a[::7, ::5] = 64                                           # some alpha structure to decode
            parts.append(encode_dxt5(np.concatenate([col, a], -1)).tobytes())
        else:
            parts.append(col.tobytes())
        d = np.ascontiguousarray(scene["depth_raw"][i], np.float32)
        parts.append(compress_depth_u8(d, near, far).tobytes() if depth_format else d.tobytes())
    msg = bytearray(b"".join(parts))
    if timestamp is not None:
        msg[:8] = np.float64(timestamp).tobytes()
    return bytes(msg)
