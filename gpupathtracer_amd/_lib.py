"""Loader of the C-ABI library (include/gpt.h).  There is no fallback: if the
HIP library has not been built, importing the product path fails loudly."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPT_LIB_PATH", os.path.join(_HERE, "lib", "libgpt_hip.so"))

_lib = None

c_void_pp = ctypes.POINTER(ctypes.c_void_p)


class GptStats(ctypes.Structure):
    _fields_ = [
        ("path_bounces", ctypes.c_uint64),
        ("live_at_depth", ctypes.c_uint64 * 64),
        ("passes", ctypes.c_uint64),
        ("kd_nodes_visited", ctypes.c_uint64),
        ("kd_tris_tested", ctypes.c_uint64),
        ("hits", ctypes.c_uint64),
        ("device_ms_trace", ctypes.c_double),
        ("device_ms_total", ctypes.c_double),
        ("trace_launches", ctypes.c_uint64),
        ("kd_nodes_at_depth", ctypes.c_uint64 * 64),
        ("kd_tris_at_depth", ctypes.c_uint64 * 64),
        ("wave_iters_node", ctypes.c_uint64),
        ("wave_iters_tri", ctypes.c_uint64),
        ("busy_lane_rounds", ctypes.c_uint64),
        ("wave_rounds", ctypes.c_uint64),
    ]


# every symbol include/gpt.h declares: (restype, argtypes)
_V = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
SYMBOLS = {
    "gpt_last_error": (ctypes.c_char_p, []),
    "gpt_render_params_default": (None, [_V]),
    "gpt_scene_create": (_I, [_V, _I, _V, _I, _V, _I, c_void_pp]),
    "gpt_scene_destroy": (_I, [_V]),
    "gpt_scene_reset": (_I, [_V, _I]),
    "gpt_render_pass": (_I, [_V, _V, _I, _V]),
    "gpt_render_range": (_I, [_V, _V, _I, _I, _I]),
    "gpt_scene_sync": (_I, [_V]),
    "gpt_accum_device_ptr": (_I, [_V, c_void_pp, ctypes.POINTER(ctypes.c_size_t)]),
    "gpt_accum_download": (_I, [_V, _V, ctypes.POINTER(_I)]),
    "gpt_accum_upload": (_I, [_V, _V, _I]),
    "gpt_accum_allreduce": (_I, [c_void_pp, _I]),
    "gpt_shutdown": (_I, []),
    "gpt_scene_kd_small_threshold": (_I, [_V, ctypes.POINTER(_I)]),
    "gpt_get_stats": (_I, [_V, ctypes.POINTER(GptStats)]),
    "gpt_scene_set_profiling": (_I, [_V, _I, _I]),
    "gpt_scene_info": (_I, [_V, ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "gpt_debug_primary_hits": (_I, [_V, _V, _I, _V, _V]),
    "gpt_debug_intersect": (_I, [_V, _V, _I, _I, _V, _V]),
    "gpt_debug_render_events": (_I, [_V, _V, _I, _V, _V, _V]),
    "gpt_debug_dump_queue": (_I, [_V, _V, _I, _I, _I, _V, _I, ctypes.POINTER(_I), ctypes.POINTER(ctypes.c_uint)]),
    "gpt_debug_rng_uniforms": (_I, [_I, _V, _I, _I, _V]),
    "gpt_debug_helpers": (_I, [_I, _I, _I, _V, _V]),
    "gpt_intersect_noalloc": (_I, [_V, _V, _I, _I, _V, _V]),
    "gpt_kdtree_build": (_I, [_V, _I, _I, c_void_pp]),
    "gpt_kdtree_build_gpu": (_I, [_V, _I, _I, _I, c_void_pp]),
    "gpt_kdtree_info": (_I, [_V, ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_I), _V]),
    "gpt_kdtree_copy": (_I, [_V, _V, _V]),
    "gpt_kdtree_free": (None, [_V]),
    "launchKernel": (None, None),   # Camera is passed by value: bound in api.launch_kernel
    "gpt_launch_kernel_params": (None, [_V]),
    "gpt_launch_kernel_static_scene": (None, [_I]),
    "gpt_device_alloc": (_I, [c_void_pp, ctypes.c_size_t]),
    "gpt_device_upload": (_I, [_V, _V, ctypes.c_size_t]),
    "gpt_device_download": (_I, [_V, _V, ctypes.c_size_t]),
    "gpt_device_free": (_I, [_V]),
    "gpt_obj_load": (_I, [ctypes.c_char_p, _V, c_void_pp, ctypes.POINTER(_I), _V]),
    "gpt_free": (None, [_V]),
    "gpt_default_scene_sphere_count": (_I, []),
    "gpt_default_scene": (None, [_V]),
    "gpt_material_preset": (_I, [ctypes.c_char_p, _V]),
    "gpt_scene_preset_count": (_I, []),
    "gpt_scene_preset_name": (ctypes.c_char_p, [_I]),
    "gpt_scene_preset": (_I, [ctypes.c_char_p, ctypes.c_char_p, c_void_pp, ctypes.POINTER(_I), c_void_pp, ctypes.POINTER(_I), _V]),
    "gpt_orbit_camera": (None, [_F, _F, _F, _F, _F, _F, _F, _V, _V]),
    "gpt_default_camera": (None, [_F, _F, _V]),
    "gpt_tonemap_8bit": (None, [_V, _I, _I, _V]),
    "gpt_write_ppm": (_I, [ctypes.c_char_p, _V, _I, _I, _I]),
    "gpt_write_pfm": (_I, [ctypes.c_char_p, _V, _I, _I, _I]),
    "gpt_write_exr": (_I, [ctypes.c_char_p, _V, _I, _I, _I]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "gpupathtracer_amd: %s is missing -- build it with "
                "`make -C gpupathtracer_amd/csrc` (or __graft_entry__.build()); "
                "there is no CPU fallback" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)          # AttributeError if a declared symbol is not exported
            if args is not None:
                fn.restype = res
                fn.argtypes = args
        _lib = L
    return _lib
