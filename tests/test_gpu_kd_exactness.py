"""GPU: what "the kd traversal returns the brute-force loop's hit" covers, as assertions.

(1) CAMERA AND BOUNCE RAYS -- the rays the renderer traces: every ray of the path queues entering depths 0, 1 and 2 of a
    1080p pass on the benchmark's scene (cow subdivided 4x, the tree the benchmark runs on), every primary ray of the
    diamond + coil and the box scenes, a strided subset of the 4K primary rays on the 10 M-triangle soup: ZERO rays on
    which triangle index or t bits differ from the device's brute-force loop (the reference's loop, :536-557).

(2) ADVERSARIAL RAYS -- aimed EXACTLY at mesh vertices from R = 1, 2, 4, 8 root-box extents away (and 10 000 for the monkey).
    Here the reference's loop itself stops being geometry: for a silhouette triangle next to the target vertex, seen
    edge-on (determinant ~1e-11), `tvec . pvec` and `d . qvec` round to exact zeros, the test passes with u = v = 0 and
    t is noise several millimetres in front of the surface; farther out it accepts triangles the ray passes many
    triangle sizes away from.  The loop tests every triangle with every ray, so it finds them; a traversal only tests the
    triangles of the cells the ray passes through, up to the first hit.  The supported bound, asserted here: at R <= 8
    extents at most 100 rays per million differ, and on EVERY such ray the loop's answer is such an artefact -- in exact
    arithmetic (float64 on the same float32 inputs) the ray misses the loop's triangle by more than 1e-3 of the triangle
    (barycentric units), or the distance the loop reports is off from the exact one by more than 1e-4 of itself (800 float
    ulps: an edge-on triangle that is grazed, with noise for a distance).  A ray on which the loop's triangle is hit at
    the distance it reports and the traversal misses it fails the test."""
import numpy as np
import pytest

import scenes as S
from kd_ray_families import aimed_at_vertices, families
from gpupathtracer_amd import api, meshes

pytestmark = pytest.mark.gpu

SUPPORTED_EXTENTS = (1.0, 2.0, 4.0, 8.0)
MAX_ARTEFACTS_PER_MILLION = 100


def differing(sc, rays):
    tk, ttk = sc.intersect(rays, use_kdtree=True)
    tb, ttb = sc.intersect(rays, use_kdtree=False)
    bad = np.flatnonzero((tk != tb) | (ttk.view(np.int32) != ttb.view(np.int32)))
    return bad, tk, ttk, tb, ttb


def exact_barycentrics(verts, rays):
    """Moller-Trumbore in float64 on the float32 inputs: (u, v, 1 - u - v, t) per (triangle, ray) pair."""
    a, b, c = [verts[:, 3 * k:3 * k + 3].astype(np.float64) for k in range(3)]
    o, d = rays[:, :3].astype(np.float64), rays[:, 3:].astype(np.float64)
    e1, e2 = b - a, c - a
    p = np.cross(d, e2)
    det = (e1 * p).sum(1)
    tv = o - a
    q = np.cross(tv, e1)
    with np.errstate(divide="ignore", invalid="ignore"):
        u, v, t = (tv * p).sum(1) / det, (d * q).sum(1) / det, (e2 * q).sum(1) / det
    return u, v, 1 - u - v, t


def assert_only_loop_artefacts(v, rays, bad, tk, tb, label, ttk, ttb):
    """Every differing ray: the loop reports a triangle (the traversal never finds a hit the loop lacks), and the ray
    misses that triangle by a clear margin in exact arithmetic."""
    if bad.size == 0:
        return
    assert (tb[bad] >= 0).all(), (label, "the traversal reports a hit where the loop has none")
    u, w, r, t = exact_barycentrics(v[tb[bad]], rays[bad])
    least = np.minimum(np.minimum(u, w), r)
    noise_t = ~(np.abs(ttb[bad].astype(np.float64) - t) <= 1e-4 * np.abs(t))
    keep = ~((least < -1e-3) | noise_t)
    real = bad[keep]
    detail = [(int(i), int(tk[i]), int(tb[i]), float(a), float(b), float(c), float(d), float(ttk[i]), float(ttb[i]))
              for i, a, b, c, d in zip(real[:5], u[keep], w[keep], r[keep], t[keep])]
    assert real.size == 0, (label, "the loop's triangle IS hit and the traversal missed it", detail)


@pytest.fixture(scope="module")
def cow_heavy():
    v, n = meshes.load_asset("cow")
    v, n = meshes.subdivide(v, n, 4)
    return v, meshes.polys_from_triangles(v, n, S.MATERIALS["glass"])


def test_camera_and_bounce_rays_of_the_benchmark_frame_equal_the_loop(cow_heavy):
    v, polys = cow_heavy
    w, h = 1920, 1080
    cam = S.orbit_camera(w, h)
    with api.Scene(S.default_spheres(), polys, api.render_params(max_depth=8)) as sc:
        sc.reset(w * h)
        for depth in (0, 1, 2):
            rays, live = sc.dump_queue(cam, 0, 1, depth, w * h)
            assert len(rays) == live and live >= (w * h if depth == 0 else 100000)
            bad, tk, _, tb, _ = differing(sc, rays)
            assert (tb >= 0).sum() > 50000
            assert bad.size == 0, (depth, bad.size, bad[:5])


def test_primary_rays_of_the_other_configurations_equal_the_loop():
    w, h = 1920, 1080
    cam = S.orbit_camera(w, h)
    dv, dn = meshes.load_asset("diamond")
    cv, cn = meshes.load_asset("coil")
    polys = np.concatenate([
        meshes.polys_from_triangles(meshes.transform(dv, 0.8, (0.2, -2.2, 2.6)), dn, S.MATERIALS["glass"]),
        meshes.polys_from_triangles(meshes.transform(cv, 0.25, (-1.2, -2.2, 1.2)), cn, S.MATERIALS["marble"])])
    with api.Scene(S.default_spheres(), polys, api.render_params(max_depth=8)) as sc:
        sc.reset(w * h)
        for depth in (0, 1, 3):
            rays, live = sc.dump_queue(cam, 0, 1, depth, w * h)
            bad, tk, _, tb, _ = differing(sc, rays)
            assert (tb >= 0).sum() > 1000 and bad.size == 0, (depth, bad.size)
    bv, bn = meshes.load_asset("box")
    cam = S.orbit_camera(512, 512)
    with api.Scene(S.default_spheres(), S.cornell_box(bv, bn), api.render_params(max_depth=8)) as sc:
        sc.reset(512 * 512)
        for depth in (0, 1, 2, 5):
            rays, live = sc.dump_queue(cam, 0, 1, depth, 512 * 512)
            bad, tk, _, tb, _ = differing(sc, rays)
            assert (tb >= 0).sum() > 1000 and bad.size == 0, (depth, bad.size)


def test_primary_rays_of_the_10m_soup_at_4k_equal_the_loop():
    w, h = 3840, 2160
    cam = S.orbit_camera(w, h)
    v, n = meshes.random_soup(10000000, seed=12345, extent=1.0, edge=0.01)
    v = meshes.transform(v, 0.75, (0.0, 0.1, 0.0))
    polys = meshes.polys_from_triangles(v, n, S.MATERIALS["white"])
    with api.Scene(S.default_spheres(), polys, api.render_params(max_depth=8, passes_in_flight=1)) as sc:
        sc.reset(w * h)
        for depth in (0, 1):
            rays, live = sc.dump_queue(cam, 0, 1, depth, w * h)
            rays = rays[depth::41]                                 # ~200 000 rays x 10 M triangles for the loop
            bad, tk, _, tb, _ = differing(sc, rays)
            assert (tb >= 0).sum() > 20000 and bad.size == 0, (depth, bad.size)


@pytest.mark.parametrize("subdiv", [3, 4])
def test_rays_aimed_at_vertices_differ_only_by_loop_artefacts(subdiv, cow_heavy):
    if subdiv == 4:
        v, polys = cow_heavy
    else:
        cv, cn = meshes.load_asset("cow")
        v, n = meshes.subdivide(cv, cn, subdiv)
        polys = meshes.polys_from_triangles(v, n, S.MATERIALS["glass"])
    rng = np.random.default_rng(5 + subdiv)
    m = 200000
    with api.Scene(None, polys, api.render_params(max_depth=1)) as sc:
        for R in SUPPORTED_EXTENTS:
            rays = aimed_at_vertices(v, rng, m, R)
            bad, tk, ttk, tb, ttb = differing(sc, rays)
            assert (tb >= 0).sum() > 0.9 * m
            assert bad.size <= MAX_ARTEFACTS_PER_MILLION * m // 1000000, (subdiv, R, bad.size)
            assert_only_loop_artefacts(v, rays, bad, tk, tb, (subdiv, R), ttk, ttb)


def test_monkey_from_10000_extents_differs_only_by_loop_artefacts():
    """The far-away family of the boundary-ray sets (origins 10 000 extents out, aimed at vertices) on the monkey, at three
    scales: beyond the supported distance, so differences may occur (measured: ~25 per 200 000) -- but only artefacts."""
    v, n = meshes.load_asset("monkey")
    rng = np.random.default_rng(2026)
    for scale, shift in ((1.0, 0.0), (1e-3, 0.0), (1e3, 5000.0)):
        vv = (v * np.float32(scale) + np.float32(shift)).astype(np.float32)
        polys = meshes.polys_from_triangles(vv, n, S.MATERIALS["glass"])
        rays = families(vv, rng, 160000)[7 * 20000:]
        with api.Scene(None, polys, api.render_params(max_depth=1)) as sc:
            bad, tk, ttk, tb, ttb = differing(sc, rays)
        assert bad.size <= 20, (scale, bad.size)
        assert_only_loop_artefacts(vv, rays, bad, tk, tb, ("monkey", scale), ttk, ttb)
