"""kd-tree built on the GPU == the host builder's tree, word for word."""
import time
import numpy as np
import pytest

import scenes as S
from gpupathtracer_amd import api, meshes

pytestmark = pytest.mark.gpu


def _compare(v, small):
    t0 = time.perf_counter(); h = api.kdtree_build(v, small); th = time.perf_counter() - t0
    t0 = time.perf_counter(); g = api.kdtree_build(v, small, on_gpu=True); tg = time.perf_counter() - t0
    print("tris %d small %d: host %.3f s, gpu %.3f s, nodes %d refs %d depth %d" % (v.shape[0], small, th, tg, h["nodes"].shape[0], h["refs"].size, h["max_depth"]))
    assert g["nodes"].shape == h["nodes"].shape, (g["nodes"].shape, h["nodes"].shape)
    bad = np.nonzero((g["nodes"] != h["nodes"]).any(axis=1))[0]
    assert bad.size == 0, (bad[:10], g["nodes"][bad[:4]], h["nodes"][bad[:4]])
    assert np.array_equal(g["refs"], h["refs"])
    assert np.array_equal(g["bounds"].view(np.uint32), h["bounds"].view(np.uint32))
    assert g["max_depth"] == h["max_depth"]


@pytest.mark.parametrize("name,sub,small", [("box", 0, 0), ("cow", 0, 0), ("cow", 0, 4), ("monkey", 0, 64), ("cow", 2, 0), ("cow", 3, 0), ("cow", 4, 0)])
def test_gpu_builder_equals_host_builder(name, sub, small):
    v, n = meshes.load_asset(name)
    if sub:
        v, n = meshes.subdivide(v, n, sub)
    _compare(np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 9), small)


def test_gpu_builder_degenerate_inputs():
    rng = np.random.default_rng(3)
    # identical triangles (the median separates nothing), zero-area triangles, +-0 coordinates
    one = rng.random((1, 9), dtype=np.float32)
    v = np.concatenate([np.repeat(one, 200, axis=0), np.zeros((50, 9), np.float32), -np.zeros((50, 9), np.float32),
                        rng.random((3000, 9), dtype=np.float32) * 0.05 + rng.random((3000, 1), dtype=np.float32)])
    _compare(v.astype(np.float32), 0)
    soup = (rng.random((200000, 1, 3), dtype=np.float32) * 2 - 1) + (rng.random((200000, 3, 3), dtype=np.float32) - 0.5) * 0.02
    _compare(soup.reshape(-1, 9).astype(np.float32), 0)


def test_gpu_builder_ten_million_triangles():
    """The largest baseline configuration: the 10 M-triangle soup of bench.py."""
    v, _ = meshes.random_soup(10000000, seed=12345, extent=1.0, edge=0.01)
    _compare(np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 9), 0)


def test_sah_stage_threshold_is_chosen_per_scene():
    """kd_small_threshold 0 (the default): the trees for 12 and 96 are built and the one with the smaller expected
    traversal work is kept -- 12 for a surface mesh, 96 for a volume of scattered small triangles."""
    cow, _ = meshes.load_asset("cow")
    soup, _ = meshes.random_soup(60000, seed=2)
    for verts, want in ((cow, 12), (soup, 96)):
        polys = meshes.polys_from_triangles(verts, np.zeros((len(verts), 3), np.float32), S.MATERIALS["white"])
        sizes = {}
        for small in (0, 12, 96):
            with api.Scene(None, polys, api.render_params(max_depth=1, kd_small_threshold=small)) as sc:
                info = sc.info()
                sizes[small] = (info["kd_nodes"], info["kd_refs"], info["kd_depth"])
        assert sizes[12] != sizes[96]
        assert sizes[0] == sizes[want], (sizes, want)


def test_scene_create_falls_back_to_the_host_builder(monkeypatch, capfd):
    """gpt_scene_create indexes a mesh of this size on the GPU, twice (the per-scene choice of the SAH stage threshold builds
    two trees).  When the second of those builds fails, the host builder -- which builds the same tree word for word --
    takes over for it: same tree statistics, same choice, same hits as a scene created without the failure."""
    v, n = meshes.load_asset("cow")
    v, n = meshes.subdivide(v, n, 3)                                   # 371 456 triangles: above the GPU builder's threshold
    polys = meshes.polys_from_triangles(v, n, api.material_preset("glass"))
    rng = np.random.default_rng(3)
    lo, hi = v.reshape(-1, 3).min(axis=0), v.reshape(-1, 3).max(axis=0)
    o = (rng.random((20000, 3), dtype=np.float32) * 3 - 1) * (hi - lo) + lo
    d = (rng.random((20000, 3), dtype=np.float32) - 0.5)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    with api.Scene(None, polys, api.render_params(max_depth=1)) as sc:
        info0, small0 = sc.info(), sc.kd_small_threshold()
        tri0, t0 = sc.intersect(rays, use_kdtree=True)
    monkeypatch.setenv("GPT_TEST_FAIL_GPU_BUILD", "2")    # (counts the GPU builds made while the variable is set)
    with api.Scene(None, polys, api.render_params(max_depth=1)) as sc:
        info1, small1 = sc.info(), sc.kd_small_threshold()
        tri1, t1 = sc.intersect(rays, use_kdtree=True)
    err = capfd.readouterr().err
    assert "building the kd-tree on the host instead" in err
    assert info0 == info1 and small0 == small1
    assert np.array_equal(tri0, tri1) and np.array_equal(t0.view(np.int32), t1.view(np.int32))
