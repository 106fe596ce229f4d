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
