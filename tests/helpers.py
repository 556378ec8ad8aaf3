"""Hand-built miniature scenes for known-answer tests (no rendering, every value chosen by hand)."""
import numpy as np


def tiny_scene(pos_calib, depth, quality, silhouette, w=2, h=2, lut=2, color=None):
    """N streams; stream i maps EVERY voxel to pos_calib[i] = (u, v, z) (constant inverse LUT) and its
    images are constant: depth[i], quality[i], silhouette[i].  With constant textures every GL filter
    returns the constant exactly, so tsdf_integration.vs:28-55 can be evaluated by hand."""
    n = len(pos_calib)
    inv = np.zeros((n, lut ** 3, 4), np.float32)
    for i, p in enumerate(pos_calib):
        inv[i, :, :3] = p
        inv[i, :, 3] = 1.0
    c = (np.arange(lut) + 0.5) / lut
    dn, vv, uu = np.meshgrid(c, c, c, indexing="ij")
    uv = np.stack([uu, vv], -1).reshape(1, -1, 2).repeat(n, 0).astype(np.float32)
    xyz = np.stack([uu, vv, dn], -1).reshape(1, -1, 3).repeat(n, 0).astype(np.float32)    # world == (u, v, d) in a unit bbox
    d = np.zeros((n, h, w, 2), np.float32)
    q = np.zeros((n, h, w), np.float32)
    s = np.zeros((n, h, w), np.float32)
    for i in range(n):
        d[i, ..., 0] = depth[i]
        q[i] = quality[i]
        s[i] = silhouette[i]
    col = np.zeros((n, h, w, 3), np.uint8) if color is None else np.asarray(color, np.uint8)
    return dict(n=n, width=w, height=h, color_width=w, color_height=h,
                bbox_min=np.zeros(3, np.float32), bbox_max=np.ones(3, np.float32),
                depth_limits=np.array([0.5, 4.5], np.float32),
                lut_res=np.array([lut] * 3, np.uint32), inv_res=np.array([lut] * 3, np.uint32),
                cv_xyz=np.ascontiguousarray(xyz), cv_uv=np.ascontiguousarray(uv), cv_xyz_inv=inv,
                depth=d, quality=q, silhouette=s, normals=np.zeros((n, h, w, 3), np.float32), color=col)


def tsdf_close(a, b, limit, rtol=1e-3):
    """north_star tolerance: |a-b| <= 1e-3 * max(|a|, |b|, limit); NaN equals NaN."""
    with np.errstate(invalid="ignore"):
        return (np.abs(a - b) <= rtol * np.maximum(np.maximum(np.abs(a), np.abs(b)), limit)) | (np.isnan(a) & np.isnan(b))


def same(a, b):
    """bit-for-bit equality of float arrays, NaN equal to NaN (the HIP path restates the oracle's fp32 operation order)."""
    a, b = np.asarray(a), np.asarray(b)
    with np.errstate(invalid="ignore"):
        return (a == b) | (np.isnan(a) & np.isnan(b))


def assert_same(a, b, what=""):
    m = same(a, b)
    if not m.all():
        with np.errstate(invalid="ignore"):
            d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))[~m]
        raise AssertionError(f"{what}: {int((~m).sum())} of {m.size} values differ (largest difference {np.nanmax(d) if d.size else 0:.3g})")


def assert_close_abs(a, b, atol, what=""):
    with np.errstate(invalid="ignore"):
        m = (np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)) <= atol) | same(a, b)
    assert m.all(), f"{what}: {int((~m).sum())} of {m.size} values differ by more than {atol}"


POW_ATOL = 1e-6      # shade mode 1 (shading.glsl:32-69) goes through pow(): glibc powf on the CPU, the device library's on the GPU (<= 1 ulp apart)


def assert_frames_identical(hip, orc, what="", min_hits=1, colour_atol=0.0):
    """the rendered frame of both sides (after drawF): framebuffer colour and depth, every value (colour_atol only for a shade
    mode that calls pow())"""
    (fc, fd), (gc, gd) = hip.framebuffer(), orc.framebuffer()
    assert_same(fd, gd, f"{what} framebuffer depth")
    if colour_atol:
        assert_close_abs(fc, gc, colour_atol, f"{what} framebuffer colour")
    else:
        assert_same(fc, gc, f"{what} framebuffer colour")
    n = int((fd < 1).sum())
    assert n >= min_hits, f"{what}: only {n} covered pixels"
    return n
