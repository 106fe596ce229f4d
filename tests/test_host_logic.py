"""CPU: the library's per-path arithmetic (pt_device.h, the code the HIP kernels call),
compiled for the host by tests/host_harness, against the oracle.  Every expression is
evaluated in the type and order the reference's host build gives it, and sin / cos / log /
pow restate the C library's algorithms bit for bit, so the two agree in every bit of every
sample."""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as O
import scenes as S
from gpupathtracer_amd import meshes
from gpupathtracer_amd.types import CAMERA_DTYPE, POLY_DTYPE, RENDER_PARAMS_DTYPE

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def H():
    so = os.path.join(HERE, "host_harness", "libhh.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "host_harness")])
    L = ctypes.CDLL(so)
    L.hh_tree_build.restype = ctypes.c_void_p
    L.hh_tree_build.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float]
    L.hh_tree_free.argtypes = [ctypes.c_void_p]
    L.hh_render_pass.restype = ctypes.c_int64
    return L


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def params(depth, stride=0):
    r = np.zeros(1, RENDER_PARAMS_DTYPE)
    r["max_depth"] = depth
    r["ray_bias"] = 0.0002
    r["min_ray_weight"] = 0.00001
    r["ground_elevation"] = -0.8
    r["ground_enabled"] = 1
    r["row_stride_is_width"] = stride
    return r


def host_render(H, tree, polys, spheres, cam, counter, rp, ids):
    n = ids.size
    out = np.zeros((n, 3), np.float32)
    fp = np.zeros(n, np.int32)
    ft = np.zeros(n, np.float32)
    cam = np.array(cam, CAMERA_DTYPE).reshape(1)
    tot = H.hh_render_pass(ctypes.c_void_p(tree) if tree else None, polys.size, P(polys), spheres.size, P(spheres),
                           P(ids), n, counter, P(cam), P(rp), P(out), P(fp), P(ft))
    return out, tot, fp, ft


def test_camera_rays_pinhole_bit_exact(H):
    cam = S.orbit_camera(96, 64, aperture=0.0)
    ids = np.arange(96 * 64, dtype=np.int32)
    got = np.zeros((ids.size, 6), np.float32)
    c = np.array(cam, CAMERA_DTYPE).reshape(1)
    for stride in (0, 1):
        H.hh_camera_rays(P(ids), ids.size, 9, P(c), stride, P(got))
        ref = O.camera_rays(cam, 9, O.params(row_stride_is_width=stride))
        assert np.array_equal(got.view(np.int32), ref.view(np.int32))


@pytest.mark.parametrize("depth", [8, 30])
def test_default_scene_logic_parity(H, depth):
    w = 160
    sph = S.default_spheres()
    cam = S.orbit_camera(w, w)
    ids = np.arange(w * w, dtype=np.int32)
    nop = np.zeros(0, POLY_DTYPE)
    for counter in (0, 5):
        o = O.render_pass(sph, None, cam, counter, O.params(max_depth=depth), want_hits=True)
        got, tot, fp, ft = host_render(H, None, nop, sph, cam, counter, params(depth), ids)
        assert tot == o["bounces"]
        assert np.array_equal(fp, o["hits"]["prim"]) and np.array_equal(ft.view(np.int32), o["hits"]["t"].view(np.int32))
        assert np.array_equal(got.view(np.uint32), o["rgb"].view(np.uint32))


def test_mesh_scene_logic_parity_kd_and_brute(H):
    w = 96
    sph = S.spheres_with(["steel", "greenGlass", "marble", "ketchup", "something", "glass", "gold", "white",
                          "red", "orange", "purple", "lightBlue", "light", "green"])
    dv, dn = meshes.load_asset("diamond")
    dv = meshes.transform(dv, 0.8, (0.2, -2.2, 2.6))
    polys = meshes.polys_from_triangles(dv, dn, S.MATERIALS["glass"])
    cam = S.orbit_camera(w, w)
    ids = np.arange(w * w, dtype=np.int32)
    o = O.render_pass(sph, polys, cam, 1, O.params(max_depth=12), want_hits=True)
    tree = H.hh_tree_build(P(np.ascontiguousarray(dv)), dv.shape[0], 64, 1.0)
    try:
        a, ta, fpa, fta = host_render(H, None, polys, sph, cam, 1, params(12), ids)
        b, tb, fpb, ftb = host_render(H, tree, polys, sph, cam, 1, params(12), ids)
    finally:
        H.hh_tree_free(ctypes.c_void_p(tree))
    # kd and brute force are the same function
    assert np.array_equal(a, b) and ta == tb and np.array_equal(fpa, fpb) and np.array_equal(fta, ftb)
    # every material branch incl. absorbing and scattering media: the oracle's samples bit for bit
    assert ta == o["bounces"]
    assert np.array_equal(a.view(np.uint32), o["rgb"].view(np.uint32))
    assert ((fpa >= 0) & (fpa < polys.size)).sum() > 50


def test_sincos_is_the_c_librarys_bit_for_bit(H):
    """pt::sincos_f (what the kernels call for lens, diffuse and scatter directions) restates the C library's
    sinf / cosf algorithm: identical bits for EVERY float of the range the renderer uses, [0, 2 pi]."""
    H.hh_sincos_mismatches.restype = ctypes.c_longlong
    H.hh_sincos_mismatches.argtypes = [ctypes.c_uint32] * 3
    two_pi_bits = int(np.array([6.2831855], np.float32).view(np.uint32)[0])
    assert H.hh_sincos_mismatches(0, two_pi_bits, 1) == 0


def test_log_and_pow_are_the_c_librarys_bit_for_bit(H):
    """pt::log_f vs logf and pt::pow_e_f vs powf((float)E, y).  Every 5th of all 2^32 float bit patterns here (zero,
    subnormals, negatives, infinities and NaNs included), plus whole stretches around 1, 2^-31 and the overflow /
    underflow thresholds; the complete sweep of all 2^32 patterns (30 s) was run when the functions were written: none differ."""
    for fn in (H.hh_log_mismatches, H.hh_pow_e_mismatches):
        fn.restype = ctypes.c_longlong
        fn.argtypes = [ctypes.c_uint32] * 3
        assert fn(0, 0xFFFFFFFF, 5) == 0
    bits = lambda x: int(np.array([x], np.float32).view(np.uint32)[0])
    assert H.hh_log_mismatches(bits(0.99), bits(1.01), 1) == 0
    assert H.hh_log_mismatches(bits(2.0 ** -31), bits(2.0 ** -30), 1) == 0
    for lo, hi in ((87.0, 89.5), (-104.5, -86.5)):
        a, b = sorted((bits(lo), bits(hi)))
        assert H.hh_pow_e_mismatches(a, b, 1) == 0
