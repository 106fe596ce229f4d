"""The reference's own kd-tree fixture -- its benchmark's dragon meshes and parallel-ray set (cukd/test/ray_traversal.cu,
which has no assertions) -- as parity cases: traversal == the reference's brute-force loop (path_tracer_kernel.cu:536-557),
bit for bit, on the host build of the traversal (CPU) and through the C-ABI on the GPU."""
import ctypes

import numpy as np
import pytest

import cukd_fixture as F
import oracle_lib as O
from gpupathtracer_amd import api, meshes
from test_kdtree import harness  # noqa: F401  (fixture)


def test_fixture_meshes_and_rays():
    v4, _ = meshes.load_asset("dragon_res4")
    v3, _ = meshes.load_asset("dragon_res3")
    assert v4.shape == (11102, 9) and v3.shape == (47794, 9)          # face counts of the two PLY files
    r = F.rays()
    assert r.shape == (1024 * 864, 6)
    assert np.all(r[:, 2] == -5) and np.all(r[:, 3:] == (0, 0, 1))
    # the benchmark's x runs from -xmax upward, its y from ymax downward
    assert r[0, 0] == np.float32(-0.11) and r[0, 1] == np.float32(0.23)
    assert abs(float(r[1023, 0]) - 0.0998) < 1e-4 and abs(float(r[-1, 1]) - 0.03023) < 1e-4


def test_dragon_traversal_equals_brute_force_on_cpu(harness):  # noqa: F811
    v, _ = meshes.load_asset("dragon_res4")
    rays = np.ascontiguousarray(F.rays()[::37])                      # 23 912 of the benchmark's rays
    H = harness
    tree = H.hh_tree_build(v.ctypes.data_as(ctypes.c_void_p), v.shape[0], 64, 1.0)
    try:
        out = {}
        for mode in (0, 1):
            tri = np.zeros(len(rays), np.int32)
            t = np.zeros(len(rays), np.float32)
            H.hh_closest(ctypes.c_void_p(tree), rays.ctypes.data_as(ctypes.c_void_p), len(rays), ctypes.c_float(123456789.0),
                         mode, tri.ctypes.data_as(ctypes.c_void_p), t.ctypes.data_as(ctypes.c_void_p), None)
            out[mode] = (tri, t)
    finally:
        H.hh_tree_free(ctypes.c_void_p(tree))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1].view(np.int32), out[1][1].view(np.int32))
    to, tto = O.closest_triangle(v, rays[:3000])                     # the oracle's restatement of the reference loop
    assert np.array_equal(to, out[0][0][:3000]) and np.array_equal(tto.view(np.int32), out[0][1][:3000].view(np.int32))
    assert (out[0][0] >= 0).mean() > 0.3                             # the dragon fills a good part of the window


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dragon_res4", "dragon_res3"])
def test_dragon_benchmark_rays_kd_equals_brute_force(name):
    """All 884 736 rays of the benchmark on both meshes: tree built on the host and on the GPU (the same tree), traversal
    against the device's brute-force loop."""
    v, n = meshes.load_asset(name)
    polys = meshes.polys_from_triangles(v, n, api.material_preset("white"))
    rays = F.rays()
    with api.Scene(None, polys, api.render_params(max_depth=1)) as sc:
        tk, ttk = sc.intersect(rays, use_kdtree=True)
        tb, ttb = sc.intersect(rays, use_kdtree=False)
    assert np.array_equal(tk, tb) and np.array_equal(ttk.view(np.int32), ttb.view(np.int32))
    assert (tb >= 0).mean() > 0.3
    host = api.kdtree_build(v, on_gpu=False)
    dev = api.kdtree_build(v, on_gpu=True)
    assert np.array_equal(host["nodes"], dev["nodes"]) and np.array_equal(host["refs"], dev["refs"])
