"""Python mirror of the C-ABI (include/gpt.h): thin, typed wrappers over
libgpt_hip.so.  All rendering happens in the HIP library."""
import ctypes

import numpy as np

from . import _lib
from .types import (CAMERA_DTYPE, MATERIAL_DTYPE, POLY_DTYPE, RENDER_PARAMS_DTYPE, SPHERE_DTYPE)

OK = 0
PRIM_MISS = -1
PRIM_GROUND = -2
PRIM_SPHERE = 0x40000000


class GptError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("gpt status %d: %s" % (status, message))
        self.status = status


def _check(status):
    if status != OK:
        raise GptError(status, _lib.lib().gpt_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _cam(camera):
    return np.array(camera, dtype=CAMERA_DTYPE).reshape(1)


def render_params(**overrides):
    """gpt_render_params_default() plus keyword overrides."""
    p = np.zeros(1, dtype=RENDER_PARAMS_DTYPE)
    _lib.lib().gpt_render_params_default(_ptr(p))
    for k, v in overrides.items():
        p[k] = v
    return p


class CameraStruct(ctypes.Structure):
    _fields_ = [("raw", ctypes.c_uint64 * 8)]   # 64 bytes, 8-aligned like the CUDA struct


class Scene:
    """gpt_scene: device-resident geometry, kd-tree and path buffers."""

    def __init__(self, spheres, polys=None, params=None, device=0):
        L = _lib.lib()
        spheres = np.ascontiguousarray(spheres if spheres is not None else np.zeros(0, SPHERE_DTYPE),
                                       dtype=SPHERE_DTYPE)
        polys = np.ascontiguousarray(polys if polys is not None else np.zeros(0, POLY_DTYPE), dtype=POLY_DTYPE)
        self.params = params if params is not None else render_params()
        self._h = ctypes.c_void_p()
        _check(L.gpt_scene_create(_ptr(spheres), spheres.size, _ptr(polys), polys.size, _ptr(self.params),
                                  int(device), ctypes.byref(self._h)))
        self.num_pixels = 0
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib().gpt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def reset(self, num_pixels):
        _check(_lib.lib().gpt_scene_reset(self._h, int(num_pixels)))
        self.num_pixels = int(num_pixels)

    def _ensure(self, cam):
        n = int(cam["resolution"][0][0]) * int(cam["resolution"][0][1])
        if n != self.num_pixels:
            self.reset(n)
        return n

    def render_pass(self, camera, counter, want_pixels=True):
        cam = _cam(camera)
        n = self._ensure(cam)
        out = np.empty((n, 3), dtype=np.float32) if want_pixels else None
        _check(_lib.lib().gpt_render_pass(self._h, _ptr(cam), int(counter), _ptr(out)))
        return out

    def render_range(self, camera, counter_begin, counter_end, sync=True):
        cam = _cam(camera)
        self._ensure(cam)
        _check(_lib.lib().gpt_render_range(self._h, _ptr(cam), int(counter_begin), int(counter_end), 1 if sync else 0))

    def sync(self):
        _check(_lib.lib().gpt_scene_sync(self._h))

    def accum_download(self):
        out = np.empty((self.num_pixels, 3), dtype=np.float32)
        passes = ctypes.c_int()
        _check(_lib.lib().gpt_accum_download(self._h, _ptr(out), ctypes.byref(passes)))
        return out, passes.value

    def accum_upload(self, rgb, passes):
        rgb = np.ascontiguousarray(rgb, dtype=np.float32).reshape(self.num_pixels, 3)
        _check(_lib.lib().gpt_accum_upload(self._h, _ptr(rgb), int(passes)))

    def accum_device_ptr(self):
        p = ctypes.c_void_p()
        nbytes = ctypes.c_size_t()
        _check(_lib.lib().gpt_accum_device_ptr(self._h, ctypes.byref(p), ctypes.byref(nbytes)))
        return p.value, nbytes.value

    def set_profiling(self, time_kernels=False, count_kd_work=False):
        _check(_lib.lib().gpt_scene_set_profiling(self._h, int(time_kernels), int(count_kd_work)))

    def stats(self):
        st = _lib.GptStats()
        _check(_lib.lib().gpt_get_stats(self._h, ctypes.byref(st)))
        return {
            "path_bounces": int(st.path_bounces),
            "live_at_depth": np.array(st.live_at_depth[:], dtype=np.int64),
            "passes": int(st.passes),
            "kd_nodes_visited": int(st.kd_nodes_visited),
            "kd_tris_tested": int(st.kd_tris_tested),
            "hits": int(st.hits),
            "device_ms_trace": float(st.device_ms_trace),
            "device_ms_total": float(st.device_ms_total),
            "trace_launches": int(st.trace_launches),
            "kd_nodes_at_depth": np.array(st.kd_nodes_at_depth[:], dtype=np.int64),
            "kd_tris_at_depth": np.array(st.kd_tris_at_depth[:], dtype=np.int64),
            "wave_iters_node": int(st.wave_iters_node),
            "wave_iters_tri": int(st.wave_iters_tri),
            "busy_lane_rounds": int(st.busy_lane_rounds),
            "wave_rounds": int(st.wave_rounds),
        }

    def info(self):
        a, b, c, d = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _check(_lib.lib().gpt_scene_info(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d)))
        return {"kd_nodes": a.value, "kd_refs": b.value, "kd_depth": c.value, "passes_in_flight": d.value}

    def kd_small_threshold(self):
        """gpt_scene_kd_small_threshold: the SAH stage threshold the scene's tree was built with."""
        v = ctypes.c_int()
        _check(_lib.lib().gpt_scene_kd_small_threshold(self._h, ctypes.byref(v)))
        return v.value

    def primary_hits(self, camera, counter):
        cam = _cam(camera)
        n = self._ensure(cam)
        prim = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float32)
        _check(_lib.lib().gpt_debug_primary_hits(self._h, _ptr(cam), int(counter), _ptr(prim), _ptr(t)))
        return prim, t

    def render_events(self, camera, counter):
        """gpt_debug_render_events: (pixels[n,3], events[n,depth] uint8, prims[n,depth] int32) of one pass."""
        cam = _cam(camera)
        n = self._ensure(cam)
        depth = max(1, int(self.params["max_depth"][0]))
        out = np.empty((n, 3), dtype=np.float32)
        events = np.empty((n, depth), dtype=np.uint8)
        prims = np.empty((n, depth), dtype=np.int32)
        _check(_lib.lib().gpt_debug_render_events(self._h, _ptr(cam), int(counter), _ptr(events), _ptr(prims), _ptr(out)))
        return out, events, prims

    def dump_queue(self, camera, counter0, passes, depth, max_rays):
        """gpt_debug_dump_queue: (rays[n,6], queue length) of the path queue entering `depth`."""
        cam = _cam(camera)
        self._ensure(cam)
        rays = np.empty((int(max_rays), 6), dtype=np.float32)
        n, live = ctypes.c_int(), ctypes.c_uint()
        _check(_lib.lib().gpt_debug_dump_queue(self._h, _ptr(cam), int(counter0), int(passes), int(depth), _ptr(rays),
                                               int(max_rays), ctypes.byref(n), ctypes.byref(live)))
        return rays[:n.value], live.value

    def intersect(self, rays, use_kdtree=True):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        tri = np.empty(rays.shape[0], dtype=np.int32)
        t = np.empty(rays.shape[0], dtype=np.float32)
        _check(_lib.lib().gpt_debug_intersect(self._h, _ptr(rays), rays.shape[0], 1 if use_kdtree else 0,
                                              _ptr(tri), _ptr(t)))
        return tri, t


def rng_uniforms(seeds, n_draws, device=0):
    """gpt_debug_rng_uniforms: the device's uniform streams, [len(seeds), n_draws] float32."""
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    out = np.empty((seeds.size, int(n_draws)), dtype=np.float32)
    _check(_lib.lib().gpt_debug_rng_uniforms(int(device), _ptr(seeds), seeds.size, int(n_draws), _ptr(out)))
    return out


def debug_helpers(fn, words_in, words_out, device=0):
    """gpt_debug_helpers: helper function `fn` of the device code on n records of `words_in.shape[1]` words;
    returns [n, words_out] uint32."""
    words_in = np.ascontiguousarray(words_in, dtype=np.uint32)
    out = np.empty((words_in.shape[0], int(words_out)), dtype=np.uint32)
    _check(_lib.lib().gpt_debug_helpers(int(device), int(fn), words_in.shape[0], _ptr(words_in), _ptr(out)))
    return out


def shutdown():
    """gpt_shutdown: RCCL communicators and launchKernel's cache released."""
    _check(_lib.lib().gpt_shutdown())


def accum_allreduce(scenes):
    """gpt_accum_allreduce: one RCCL sum of the scenes' accumulated images (one scene per GPU)."""
    arr = (ctypes.c_void_p * len(scenes))(*[sc._h for sc in scenes])
    _check(_lib.lib().gpt_accum_allreduce(ctypes.cast(arr, _lib.c_void_pp), len(scenes)))


NOALLOC_MAX_RAYS = 16384


def intersect_noalloc(scene, d_rays, n, d_tri, d_t, use_kdtree=True):
    """gpt_intersect_noalloc: device pointers in and out, no allocation inside the call."""
    _check(_lib.lib().gpt_intersect_noalloc(scene._h, d_rays, int(n), 1 if use_kdtree else 0, d_tri, d_t))


def kdtree_build(verts, small_threshold=0, on_gpu=False, device=0):
    """kd-tree builder (host, or the same tree built on the GPU).  Returns dict(nodes[n,2] uint32, refs int32, bounds[6], max_depth)."""
    L = _lib.lib()
    verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
    h = ctypes.c_void_p()
    if on_gpu:
        _check(L.gpt_kdtree_build_gpu(_ptr(verts), verts.shape[0], int(small_threshold), int(device), ctypes.byref(h)))
    else:
        _check(L.gpt_kdtree_build(_ptr(verts), verts.shape[0], int(small_threshold), ctypes.byref(h)))
    try:
        nn, nr, md = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        bounds = np.zeros(6, dtype=np.float32)
        _check(L.gpt_kdtree_info(h, ctypes.byref(nn), ctypes.byref(nr), ctypes.byref(md), _ptr(bounds)))
        nodes = np.zeros((nn.value, 2), dtype=np.uint32)
        refs = np.zeros(nr.value, dtype=np.int32)
        _check(L.gpt_kdtree_copy(h, _ptr(nodes), _ptr(refs)))
    finally:
        L.gpt_kdtree_free(h)
    return {"nodes": nodes, "refs": refs, "bounds": bounds, "max_depth": md.value}


def obj_load(path, material=None):
    """gpt_obj_load: returns (polys[POLY_DTYPE], bbox[6])."""
    L = _lib.lib()
    mat = np.array(material, dtype=MATERIAL_DTYPE).reshape(1) if material is not None else None
    out = ctypes.c_void_p()
    n = ctypes.c_int()
    bbox = np.zeros(6, dtype=np.float32)
    _check(L.gpt_obj_load(path.encode(), _ptr(mat), ctypes.byref(out), ctypes.byref(n), _ptr(bbox)))
    try:
        buf = (ctypes.c_char * (n.value * POLY_DTYPE.itemsize)).from_address(out.value) if n.value else b""
        polys = np.frombuffer(bytes(buf), dtype=POLY_DTYPE).copy()
    finally:
        L.gpt_free(out)
    return polys, bbox


def default_scene():
    L = _lib.lib()
    s = np.zeros(L.gpt_default_scene_sphere_count(), dtype=SPHERE_DTYPE)
    L.gpt_default_scene(_ptr(s))
    return s


def material_preset(name):
    m = np.zeros(1, dtype=MATERIAL_DTYPE)
    _check(_lib.lib().gpt_material_preset(name.encode(), _ptr(m)))
    return m[0]


def scene_preset_names():
    L = _lib.lib()
    return [L.gpt_scene_preset_name(i).decode() for i in range(L.gpt_scene_preset_count())]


def scene_preset(name, mesh_dir=None):
    """gpt_scene_preset: (spheres, polys or None, view dict for orbit_camera) of a named scene."""
    L = _lib.lib()
    ps, pp = ctypes.c_void_p(), ctypes.c_void_p()
    ns, npoly = ctypes.c_int(), ctypes.c_int()
    view = np.zeros(8, dtype=np.float32)
    _check(L.gpt_scene_preset(name.encode(), mesh_dir.encode() if mesh_dir else None, ctypes.byref(ps), ctypes.byref(ns),
                              ctypes.byref(pp), ctypes.byref(npoly), _ptr(view)))
    try:
        spheres = np.frombuffer(bytes((ctypes.c_char * (ns.value * SPHERE_DTYPE.itemsize)).from_address(ps.value)),
                                dtype=SPHERE_DTYPE).copy() if ns.value else np.zeros(0, SPHERE_DTYPE)
        polys = np.frombuffer(bytes((ctypes.c_char * (npoly.value * POLY_DTYPE.itemsize)).from_address(pp.value)),
                              dtype=POLY_DTYPE).copy() if npoly.value else None
    finally:
        L.gpt_free(ps)
        L.gpt_free(pp)
    return spheres, polys, {"yaw": float(view[0]), "pitch": float(view[1]), "radius": float(view[2]), "aperture": float(view[3]),
                            "fovx": float(view[4]), "center": tuple(float(x) for x in view[5:8])}


def orbit_camera(width, height, fovx=45.0, yaw=0.0, pitch=0.3, radius=4.0, aperture=0.04, center=(0, 0, 0)):
    c = np.zeros(1, dtype=CAMERA_DTYPE)
    ctr = np.array(center, dtype=np.float32)
    _lib.lib().gpt_orbit_camera(width, height, fovx, yaw, pitch, radius, aperture, _ptr(ctr), _ptr(c))
    return c[0]


def default_camera(width, height):
    c = np.zeros(1, dtype=CAMERA_DTYPE)
    _lib.lib().gpt_default_camera(width, height, _ptr(c))
    return c[0]


def tonemap_8bit(sum_rgb, passes):
    sum_rgb = np.ascontiguousarray(sum_rgb, dtype=np.float32).reshape(-1, 3)
    out = np.empty((sum_rgb.shape[0], 3), dtype=np.uint8)
    _lib.lib().gpt_tonemap_8bit(_ptr(sum_rgb), sum_rgb.shape[0], int(passes), _ptr(out))
    return out


def write_ppm(path, sum_rgb, width, height, passes):
    sum_rgb = np.ascontiguousarray(sum_rgb, dtype=np.float32)
    _check(_lib.lib().gpt_write_ppm(path.encode(), _ptr(sum_rgb), int(width), int(height), int(passes)))


def write_pfm(path, sum_rgb, width, height, passes):
    sum_rgb = np.ascontiguousarray(sum_rgb, dtype=np.float32)
    _check(_lib.lib().gpt_write_pfm(path.encode(), _ptr(sum_rgb), int(width), int(height), int(passes)))


def write_exr(path, sum_rgb, width, height, passes):
    sum_rgb = np.ascontiguousarray(sum_rgb, dtype=np.float32)
    _check(_lib.lib().gpt_write_exr(path.encode(), _ptr(sum_rgb), int(width), int(height), int(passes)))


# ---- the reference-compatible entry point -------------------------------------
def device_alloc(nbytes):
    p = ctypes.c_void_p()
    _check(_lib.lib().gpt_device_alloc(ctypes.byref(p), int(nbytes)))
    return p


def device_upload(dst, host_array):
    host_array = np.ascontiguousarray(host_array)
    _check(_lib.lib().gpt_device_upload(dst, _ptr(host_array), host_array.nbytes))


def device_download(src, nbytes, dtype=np.uint8):
    out = np.empty(int(nbytes) // np.dtype(dtype).itemsize, dtype=dtype)
    _check(_lib.lib().gpt_device_download(_ptr(out), src, out.nbytes))
    return out


def device_free(p):
    _check(_lib.lib().gpt_device_free(p))


def launch_kernel(num_polys, dev_polys, num_spheres, dev_spheres, num_pixels, counter, camera):
    """launchKernel(numPolys, polys, numSpheres, spheres, numPixels, pixels, counter, renderCamera)
    with device pointers for polys/spheres, as PathTracer::render calls it (path_tracer.cpp:97)."""
    L = _lib.lib()
    fn = L.launchKernel
    fn.restype = None
    fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                   ctypes.c_int, CameraStruct]
    cam = CameraStruct.from_buffer_copy(_cam(camera).tobytes())
    pixels = np.empty((num_pixels, 3), dtype=np.float32)
    fn(int(num_polys), dev_polys, int(num_spheres), dev_spheres, int(num_pixels), _ptr(pixels), int(counter), cam)
    return pixels


def launch_kernel_params(params):
    _lib.lib().gpt_launch_kernel_params(_ptr(params) if params is not None else None)


def launch_kernel_static_scene(is_static):
    """gpt_launch_kernel_static_scene: non-zero = launchKernel stops re-reading the scene arrays on every call."""
    _lib.lib().gpt_launch_kernel_static_scene(int(is_static))
