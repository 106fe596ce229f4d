"""ctypes binding of the CPU oracle (oracle/_build/liborc.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the gpupathtracer_amd package.
"""
import ctypes
import os
import subprocess

import numpy as np

from gpupathtracer_amd.types import CAMERA_DTYPE, POLY_DTYPE, SPHERE_DTYPE

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "liborc.so")


class OrcParams(ctypes.Structure):
    _fields_ = [
        ("max_depth", ctypes.c_int),
        ("ray_bias", ctypes.c_float),
        ("min_ray_weight", ctypes.c_double),
        ("ground_elevation", ctypes.c_float),
        ("ground_enabled", ctypes.c_int),
        ("row_stride_is_width", ctypes.c_int),
    ]


HIT_DTYPE = np.dtype([("prim", np.int32), ("t", np.float32)])

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.orc_randhash.restype = ctypes.c_uint32
        L.orc_randhash.argtypes = [ctypes.c_uint32]
        L.orc_minstd_nth.restype = ctypes.c_uint32
        L.orc_minstd_nth.argtypes = [ctypes.c_uint32, ctypes.c_int]
        L.orc_uniform_stream.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
        L.orc_params_default.argtypes = [ctypes.POINTER(OrcParams)]
        L.orc_render_pass.restype = ctypes.c_int64
        L.orc_render_pass.argtypes = [
            ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
            ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
            ctypes.c_void_p, ctypes.POINTER(OrcParams),
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_render_pass_ex.restype = ctypes.c_int64
        L.orc_render_pass_ex.argtypes = L.orc_render_pass.argtypes + [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_camera_rays.argtypes = [
            ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
            ctypes.c_void_p, ctypes.POINTER(OrcParams), ctypes.c_void_p]
        L.orc_closest_triangle.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
            ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_tonemap_8bit.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.orc_num_threads.restype = ctypes.c_int
        L.orc_set_num_threads.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def params(max_depth=30, row_stride_is_width=0, ground_enabled=1):
    p = OrcParams()
    lib().orc_params_default(ctypes.byref(p))
    p.max_depth = max_depth
    p.row_stride_is_width = row_stride_is_width
    p.ground_enabled = ground_enabled
    return p


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def randhash(a):
    return int(lib().orc_randhash(ctypes.c_uint32(a & 0xFFFFFFFF)))


def minstd_nth(seed, n):
    return int(lib().orc_minstd_nth(ctypes.c_uint32(seed), n))


def uniform_stream(seed, n):
    out = np.empty(n, dtype=np.float32)
    lib().orc_uniform_stream(ctypes.c_uint32(seed & 0xFFFFFFFF), n, _ptr(out))
    return out


def render_pass(spheres, polys, camera, counter, p, pixel_ids=None, want_hits=False, want_events=False):
    """One pass of the reference algorithm.  Returns dict(rgb, live, bounces[, hits][, events, prims]);
    events / prims: [n, max_depth], the branch code (ORC_EV_*) and primitive of every bounce."""
    spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE) if spheres is not None else np.zeros(0, SPHERE_DTYPE)
    polys = np.ascontiguousarray(polys, dtype=POLY_DTYPE) if polys is not None else np.zeros(0, POLY_DTYPE)
    cam = np.array(camera, dtype=CAMERA_DTYPE).reshape(1)
    num_pixels = int(cam["resolution"][0][0]) * int(cam["resolution"][0][1])
    if pixel_ids is not None:
        pixel_ids = np.ascontiguousarray(pixel_ids, dtype=np.int32)
        n = pixel_ids.size
    else:
        n = num_pixels
    rgb = np.empty((n, 3), dtype=np.float32)
    live = np.zeros(p.max_depth, dtype=np.int64)
    hits = np.empty(n, dtype=HIT_DTYPE) if want_hits else None
    events = np.empty((n, p.max_depth), dtype=np.uint8) if want_events else None
    prims = np.empty((n, p.max_depth), dtype=np.int32) if want_events else None
    total = lib().orc_render_pass_ex(
        polys.size, _ptr(polys), spheres.size, _ptr(spheres), num_pixels, _ptr(pixel_ids), n,
        int(counter), _ptr(cam), ctypes.byref(p), _ptr(rgb), _ptr(live), _ptr(hits), _ptr(events), _ptr(prims))
    out = {"rgb": rgb, "live": live, "bounces": int(total)}
    if want_events:
        out["events"], out["prims"] = events, prims
    if want_hits:
        out["hits"] = hits
    return out


def camera_rays(camera, counter, p, pixel_ids=None):
    cam = np.array(camera, dtype=CAMERA_DTYPE).reshape(1)
    num_pixels = int(cam["resolution"][0][0]) * int(cam["resolution"][0][1])
    if pixel_ids is not None:
        pixel_ids = np.ascontiguousarray(pixel_ids, dtype=np.int32)
        n = pixel_ids.size
    else:
        n = num_pixels
    out = np.empty((n, 6), dtype=np.float32)
    lib().orc_camera_rays(num_pixels, _ptr(pixel_ids), n, int(counter), _ptr(cam), ctypes.byref(p), _ptr(out))
    return out


def closest_triangle(verts, rays, t_init=123456789.0):
    verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    tri = np.empty(rays.shape[0], dtype=np.int32)
    t = np.empty(rays.shape[0], dtype=np.float32)
    lib().orc_closest_triangle(_ptr(verts), verts.shape[0], _ptr(rays), rays.shape[0],
                               ctypes.c_float(t_init), _ptr(tri), _ptr(t))
    return tri, t


def tonemap_8bit(sum_rgb, passes):
    sum_rgb = np.ascontiguousarray(sum_rgb, dtype=np.float32).reshape(-1, 3)
    out = np.empty((sum_rgb.shape[0], 3), dtype=np.uint8)
    lib().orc_tonemap_8bit(_ptr(sum_rgb), sum_rgb.shape[0], int(passes), _ptr(out))
    return out


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def kd_build(verts, small_threshold=24, traversal_cost=3.0):
    """Oracle kd-tree builder (oracle/kdtree_oracle.c)."""
    L = lib()
    L.orc_kd_build.restype = ctypes.c_int
    L.orc_kd_build.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                               ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int),
                               ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int),
                               ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    L.orc_kd_free.argtypes = [ctypes.c_void_p]
    verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
    pn, pr = ctypes.c_void_p(), ctypes.c_void_p()
    nn, nr, md = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    bounds = np.zeros(6, dtype=np.float32)
    rc = L.orc_kd_build(_ptr(verts), verts.shape[0], int(small_threshold), ctypes.c_float(traversal_cost),
                        ctypes.byref(pn), ctypes.byref(nn), ctypes.byref(pr), ctypes.byref(nr),
                        _ptr(bounds), ctypes.byref(md))
    assert rc == 0
    nodes = np.ctypeslib.as_array(ctypes.cast(pn, ctypes.POINTER(ctypes.c_uint32)), shape=(nn.value, 2)).copy()
    refs = (np.ctypeslib.as_array(ctypes.cast(pr, ctypes.POINTER(ctypes.c_int32)), shape=(max(nr.value, 1),)).copy()[:nr.value])
    L.orc_kd_free(pn)
    L.orc_kd_free(pr)
    return {"nodes": nodes, "refs": refs, "bounds": bounds, "max_depth": md.value}
