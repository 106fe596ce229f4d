"""CPU: pins that do not come from this repository's own reading of the reference.

* tests/golden/thrust_rng.npz -- outputs of the REAL thrust::default_random_engine and
  uniform_real_distribution<float>(0,1) (the image's rocThrust, host build), the reference's
  random-number dependency (path_tracer_kernel.cu:139-142, 492-493).  The oracle's generator and
  the generator the HIP kernels use (pt::Rng, here compiled for the host) must reproduce every
  value bit for bit.
* tests/golden/obj_<name>.npz -- what the REFERENCE's own OBJ loader (objcore, compiled from
  /root/reference without stand-ins) holds after loading each of its bundled meshes.  The
  library's loader and the bundled assets must agree with it exactly.

Both files are produced by tools/make_goldens.py (container only)."""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as O
from gpupathtracer_amd import api, meshes

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
MESHES = ("box", "box2", "cone", "diamond", "coil", "monkey", "cow")


@pytest.fixture(scope="module")
def rng_gold():
    return np.load(os.path.join(GOLD, "thrust_rng.npz"))


def test_thrust_fixture_holds_the_edge_cases(rng_gold):
    u, seeds = rng_gold["uniform"], rng_gold["seeds"]
    assert u.shape == (64, 4096) and seeds.shape == (64,) and int(rng_gold["minstd_10000th"]) == 399268537
    assert (u == 1.0).sum() >= 3 and (u == 0.0).sum() >= 1          # float(x - 1) rounds up to 2^31; x == 1
    assert u.min() >= 0.0 and u.max() <= 1.0
    # seeds 0 and m both become state 1; m + 5 is 5
    s = list(seeds)
    assert np.array_equal(u[s.index(0)], u[s.index(1)]) and np.array_equal(u[s.index(2147483647)], u[s.index(1)])


def test_oracle_rng_equals_thrust(rng_gold):
    for i, seed in enumerate(rng_gold["seeds"]):
        got = O.uniform_stream(int(seed), 4096)
        assert np.array_equal(got.view(np.uint32), rng_gold["uniform"][i].view(np.uint32)), int(seed)
        raw = [O.minstd_nth(int(seed), k + 1) for k in range(0, 64, 9)]
        assert raw == [int(x) for x in rng_gold["raw"][i][0:64:9]]


def test_kernel_rng_host_build_equals_thrust(rng_gold):
    so = os.path.join(HERE, "host_harness", "libhh.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "host_harness")])
    L = ctypes.CDLL(so)
    seeds = np.ascontiguousarray(rng_gold["seeds"], dtype=np.uint32)
    out = np.empty((seeds.size, 4096), np.float32)
    L.hh_rng_uniforms(seeds.ctypes.data_as(ctypes.c_void_p), int(seeds.size), 4096, out.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(out.view(np.uint32), rng_gold["uniform"].view(np.uint32))


def golden_polys(name):
    """The reference loader's data after PathTracer::prepMesh's conversion (path_tracer.cpp:58-67):
    first three vertices of every face, normal = the normal at the face's FIRST normal index."""
    z = np.load(os.path.join(GOLD, "obj_%s.npz" % name))
    fo, fi = z["face_offsets"], z["face_indices"]
    no, ni = z["facenormal_offsets"], z["facenormal_indices"]
    n_faces = fo.size - 1
    assert no.size - 1 == n_faces                       # every bundled mesh gives a normal list per face
    first = fo[:-1]
    verts = np.concatenate([z["points"][fi[first]], z["points"][fi[first + 1]], z["points"][fi[first + 2]]], axis=1)
    normals = z["normals"][ni[no[:-1]]]
    return verts.astype(np.float32), normals.astype(np.float32), z["bbox"]


@pytest.mark.parametrize("name", MESHES)
def test_bundled_assets_equal_the_reference_loaders_output(name):
    v, n, _ = golden_polys(name)
    av, an = meshes.load_asset(name)
    assert np.array_equal(av.view(np.uint32), v.view(np.uint32)) and np.array_equal(an.view(np.uint32), n.view(np.uint32))


@pytest.mark.parametrize("name", MESHES)
def test_obj_loader_equals_the_reference_loader(reference_root, name):
    polys, bbox = api.obj_load(os.path.join(reference_root, name + ".obj"))
    v, n, bb = golden_polys(name)
    got_v = np.concatenate([polys["p0"], polys["p1"], polys["p2"]], axis=1)
    assert np.array_equal(got_v.view(np.uint32), v.view(np.uint32))
    assert np.array_equal(polys["n"].view(np.uint32), n.view(np.uint32))
    assert np.array_equal(bbox.view(np.uint32), bb.view(np.uint32))


def test_loader_syntaxes_equal_the_reference_loader():
    """tests/golden/quirks_a.obj (this repository's file) went through the reference's loader when the
    goldens were made: both slash syntaxes, polygons (first three vertices kept), comments and groups."""
    polys, bbox = api.obj_load(os.path.join(GOLD, "quirks_a.obj"))
    v, n, bb = golden_polys("quirks_a")
    assert polys.size == 5
    got_v = np.concatenate([polys["p0"], polys["p1"], polys["p2"]], axis=1)
    assert np.array_equal(got_v.view(np.uint32), v.view(np.uint32))
    assert np.array_equal(polys["n"].view(np.uint32), n.view(np.uint32))
    assert np.array_equal(bbox.view(np.uint32), bb.view(np.uint32))


def test_faces_without_normals():
    """The reference's loader records no normal list for such faces (its prepMesh would then read out of
    range); vertices and bbox still agree, and the library supplies the geometric normal instead."""
    z = np.load(os.path.join(GOLD, "obj_quirks_b.npz"))
    assert z["facenormal_offsets"].size == 1 and z["normals"].size == 0
    polys, bbox = api.obj_load(os.path.join(GOLD, "quirks_b.obj"))
    first = z["face_offsets"][:-1]
    fi = z["face_indices"]
    want_v = np.concatenate([z["points"][fi[first]], z["points"][fi[first + 1]], z["points"][fi[first + 2]]], axis=1)
    got_v = np.concatenate([polys["p0"], polys["p1"], polys["p2"]], axis=1)
    assert np.array_equal(got_v, want_v) and np.array_equal(bbox, z["bbox"])
    e1, e2 = polys["p1"] - polys["p0"], polys["p2"] - polys["p0"]
    g = np.cross(e1.astype(np.float64), e2.astype(np.float64))
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    assert np.allclose(polys["n"], g, atol=1e-6)
