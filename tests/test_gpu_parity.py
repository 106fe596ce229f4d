"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU
oracle (oracle/pathtracer_oracle.c) on the same inputs.

Tolerances.  The stated bar for radiance is 1e-4 relative per sample.  These
tests assert more: everything -- RNG, hit primitive ids and distances, live-path
counts, the branch every bounce takes, and every sample's radiance -- equals the
oracle's bit for bit (tests/parity.py says why that is attainable).
"""
import numpy as np
import pytest

import oracle_lib as O
import parity
import scenes as S
from gpupathtracer_amd import api, meshes

pytestmark = pytest.mark.gpu


def check_against_oracle(sc, sph, polys, cam, counter, depth, stride_mode, ids=None):
    """One pass through gpt_debug_render_events vs the oracle: paths, then radiance."""
    got, ev, pr = sc.render_events(cam, counter)
    ref = O.render_pass(sph, polys, cam, counter, O.params(max_depth=depth, row_stride_is_width=stride_mode),
                        pixel_ids=ids, want_events=True)
    parity.check_pass(got, ev, pr, ref, ids=ids)
    return got, ref


def test_device_rng_equals_thrust():
    """The generator as the GPU runs it against the real Thrust classes' output (tests/golden/thrust_rng.npz,
    made from the image's rocThrust by tools/make_goldens.py): every uniform bit for bit, u == 1.0 included."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "thrust_rng.npz"))
    got = api.rng_uniforms(z["seeds"], 4096)
    assert np.array_equal(got.view(np.uint32), z["uniform"].view(np.uint32))
    assert (got == 1.0).sum() >= 3


@pytest.mark.parametrize("name", ["default", "materials", "meshes"])
def test_committed_vectors(name):
    """The HIP path against the committed vectors of tests/golden/oracle_*.npz (no oracle call): radiance bit for bit,
    the branch and primitive of every bounce, the live count of every depth."""
    import os
    import golden_scenes as G
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_%s.npz" % name))
    sph, polys, cam, depth, counters, stride_mode = G.scene(name)
    with api.Scene(sph, polys, api.render_params(max_depth=depth, row_stride_is_width=stride_mode)) as sc:
        for c in counters:
            got, ev, pr = sc.render_events(cam, c)
            assert np.array_equal(ev, z["events_%d" % c]) and np.array_equal(pr, z["prims_%d" % c])
            assert parity.bits_equal(got, z["rgb_%d" % c])
            assert np.array_equal((ev > 0).sum(axis=0), z["live_%d" % c])


def test_primary_hits_pinhole_exact():
    """No lens sampling -> no transcendental on the primary path: ids and t bit-exact."""
    w = h = 256
    cam = S.orbit_camera(w, h, aperture=0.0)
    sph = S.default_spheres()
    with api.Scene(sph, None, api.render_params(max_depth=4, row_stride_is_width=0)) as sc:
        prim, t = sc.primary_hits(cam, 5)
    ref = O.render_pass(sph, None, cam, 5, O.params(max_depth=1), want_hits=True)
    assert np.array_equal(prim, ref["hits"]["prim"])
    assert np.array_equal(t.view(np.int32), ref["hits"]["t"].view(np.int32))


def test_primary_hits_with_lens():
    w = h = 256
    cam = S.orbit_camera(w, h)
    sph = S.default_spheres()
    with api.Scene(sph, None, api.render_params(max_depth=4, row_stride_is_width=0)) as sc:
        prim, t = sc.primary_hits(cam, 0)
    ref = O.render_pass(sph, None, cam, 0, O.params(max_depth=1), want_hits=True)
    # the lens sample's sin / cos are the C library's bit for bit (pt::sincos_f), everything else is IEEE
    # float arithmetic in the same order: ids and t are exact with the lens as well
    assert np.array_equal(prim, ref["hits"]["prim"])
    assert np.array_equal(t.view(np.int32), ref["hits"]["t"].view(np.int32))


@pytest.mark.parametrize("kw", [dict(aperture=0.6), dict(aperture=0.3, radius=1.2), dict(fovx=120.0, pitch=1.2),
                                dict(aperture=0.04, radius=0.45, yaw=1.0), dict(aperture=2.0, radius=2.5)])
def test_primary_hits_with_per_pixel_sphere_masks(kw):
    """Primary rays test only the spheres of their pixel's mask (k_camera_masks).  Wide lenses, a camera between the
    spheres, one inside the lens' reach of a sphere, a wide field of view: ids and t still equal the full loop's, both
    with one pass per launch (a wave = 64 neighbouring pixels) and with 64 (a wave = one pixel)."""
    w, h = 192, 128
    cam = S.orbit_camera(w, h, **kw)
    sph = S.default_spheres()
    with api.Scene(sph, None, api.render_params(max_depth=1, row_stride_is_width=1, passes_in_flight=64)) as sc:
        prim, t = sc.primary_hits(cam, 3)
        sc.reset(w * h)
        sc.render_range(cam, 0, 64)
        img, passes = sc.accum_download()
    ref = O.render_pass(sph, None, cam, 3, O.params(max_depth=1, row_stride_is_width=1), want_hits=True)
    assert np.array_equal(prim, ref["hits"]["prim"])
    assert np.array_equal(t.view(np.int32), ref["hits"]["t"].view(np.int32))
    assert (prim != -1).any()
    # 64 passes in one launch: what one bounce contributes depends on the primary hit alone
    acc = np.zeros((w * h, 3), dtype=np.float32)
    for c in range(64):
        acc += O.render_pass(sph, None, cam, c, O.params(max_depth=1, row_stride_is_width=1))["rgb"]
    assert passes == 64 and parity.bits_equal(img, acc)


@pytest.mark.parametrize("aperture", [0.0, 0.04])
def test_sphere_masks_keep_sub_pixel_spheres_at_4k(aperture):
    """32 spheres of sub-pixel angular size, 20 to 120 units from the camera, at 3840 x 2160: the per-pixel masks
    (k_camera_masks) must keep every sphere a ray of the pixel grazes.  Checked against the full loop on the device
    itself: with a 33rd sphere (behind the camera, never hit) the masks are switched off and every ray tests every sphere."""
    w, h = 3840, 2160
    cam = S.orbit_camera(w, h, aperture=aperture)
    eye = np.array(cam["position"], np.float32).reshape(3)
    view = np.array(cam["view"], np.float32).reshape(3)
    rng = np.random.default_rng(11)
    sph = np.zeros(32, dtype=S.SPHERE_DTYPE)
    right = np.cross(view, (0, 1, 0)); right /= np.linalg.norm(right)
    up = np.cross(right, view)
    for i in range(32):
        dist = rng.uniform(20.0, 120.0)
        # inside the frustum (half-angles 22.5 and ~12.9 degrees)
        a, b = rng.uniform(-0.35, 0.35), rng.uniform(-0.2, 0.2)
        d = view + a * right + b * up
        sph[i]["position"] = eye + dist * d / np.linalg.norm(d)
        sph[i]["radius"] = dist * 2.0e-4 * rng.uniform(0.3, 0.9)        # a pixel subtends 2.0e-4 rad
        sph[i]["material"] = S.MATERIALS["red"]
    more = np.concatenate([sph, sph[:1]])
    more[32]["position"] = eye - 10.0 * view
    rp = api.render_params(max_depth=1, ground_enabled=0)
    with api.Scene(sph, None, rp) as sc:
        prim, t = sc.primary_hits(cam, 0)
    with api.Scene(more, None, rp) as sc:
        prim_full, t_full = sc.primary_hits(cam, 0)
    assert np.array_equal(prim, prim_full) and np.array_equal(t.view(np.int32), t_full.view(np.int32))
    assert (prim_full >= 0).sum() >= 8, "the fixture must have rays that hit the tiny spheres"


@pytest.mark.parametrize("depth", [8, 30])
def test_default_scene_pass(depth):
    """The reference's default scene (14 spheres, ground, sky), per-sample radiance."""
    w = h = 256
    cam = S.orbit_camera(w, h)
    sph = S.default_spheres()
    with api.Scene(sph, None, api.render_params(max_depth=depth, row_stride_is_width=0)) as sc:
        for counter in (0, 1, 7):
            got, ref = check_against_oracle(sc, sph, None, cam, counter, depth, 0)
            assert np.array_equal(got, sc.render_pass(cam, counter))     # the instrumented pass IS the pass
        st = sc.stats()
    assert st["passes"] == 6
    assert st["live_at_depth"][0] == 6 * w * h


def test_live_counts_match_oracle_pinhole():
    w = h = 128
    cam = S.orbit_camera(w, h, aperture=0.0)
    sph = S.default_spheres()
    depth = 12
    with api.Scene(sph, None, api.render_params(max_depth=depth, row_stride_is_width=0)) as sc:
        sc.render_pass(cam, 3, want_pixels=False)
        st = sc.stats()
    ref = O.render_pass(sph, None, cam, 3, O.params(max_depth=depth))
    live = st["live_at_depth"][:depth]
    assert np.array_equal(live, ref["live"]), (live, ref["live"])


def test_all_material_branches():
    """Every sphere gets a different preset so that reflection, refraction, absorption-only
    media, scattering media and pure diffuse are all exercised."""
    names = ["steel", "greenGlass", "marble", "ketchup", "something", "glass", "gold", "white",
             "red", "orange", "purple", "lightBlue", "light", "green"]
    sph = S.spheres_with(names)
    w = h = 192
    cam = S.orbit_camera(w, h)
    depth = 16
    with api.Scene(sph, None, api.render_params(max_depth=depth, row_stride_is_width=0)) as sc:
        for counter in (2, 3):
            _, ref = check_against_oracle(sc, sph, None, cam, counter, depth, 0)
            # every branch of traceRayKernel occurs: scatter, reflect, transmit, diffuse, miss, weight kill
            assert set(np.unique(ref["events"] & 7)) >= {1, 2, 3, 4, 5} and (ref["events"] & 8).any()


@pytest.mark.parametrize("ground", [0, 1])
def test_empty_scene(ground):
    """No spheres and no polys (numSpheres = numPolys = 0, what the reference's loops simply skip): with the ground switched
    off every ray ends in the sky at depth 0; with it, paths bounce between ground and sky.  Bit for bit the oracle's."""
    w = h = 64
    cam = S.orbit_camera(w, h)
    depth = 5
    with api.Scene(None, None, api.render_params(max_depth=depth, row_stride_is_width=0, ground_enabled=ground)) as sc:
        got, ev, pr = sc.render_events(cam, 9)
        st = sc.stats()
    ref = O.render_pass(None, None, cam, 9, O.params(max_depth=depth, ground_enabled=ground), want_events=True)
    parity.check_pass(got, ev, pr, ref)
    assert st["path_bounces"] == ref["bounces"]
    if not ground:
        assert st["path_bounces"] == w * h and (got > 0).all()


def test_paths_that_add_radiance_at_several_bounces():
    """Glowing diffuse spheres: a path adds emitted light at every diffuse bounce and the sky when it leaves, so its
    radiance slot is added to several times, in bounce order (accumulatedColors, path_tracer_kernel.cu:690, 712).  The
    kernel stores the first contribution to a slot and read-modify-writes the later ones; per-sample radiance must equal
    the oracle's in every bit, in one-pass launches and with many passes per launch."""
    sph = S.default_spheres().copy()
    for i in range(sph.size):
        m = sph[i]["material"]
        m["diffuseColor"] = (0.8, 0.7, 0.6)
        m["emittedColor"] = (0.25 * (i % 3), 0.125, 0.0625 * (i % 5))     # some exactly zero components, some spheres dark in red
        m["specularColor"] = (0, 0, 0)
        m["medium"]["refractiveIndex"] = 1.0                             # no specular lobe: every hit is a diffuse bounce
        m["hasTransmission"] = False
    w = h = 96
    cam = S.orbit_camera(w, h)
    depth = 6
    with api.Scene(sph, None, api.render_params(max_depth=depth, row_stride_is_width=0)) as sc:
        _, ref = check_against_oracle(sc, sph, None, cam, 5, depth, 0)
    assert ((ref["events"] & 7) == 4).sum(axis=1).max() >= 3, "the fixture needs paths with several diffuse bounces"
    P = 8
    with api.Scene(sph, None, api.render_params(max_depth=depth, row_stride_is_width=0, passes_in_flight=P)) as sc:
        sc.reset(w * h)
        sc.render_range(cam, 0, P)
        img, passes = sc.accum_download()
    acc = np.zeros((w * h, 3), np.float32)
    for c in range(P):
        acc += O.render_pass(sph, None, cam, c, O.params(max_depth=depth))["rgb"].reshape(w * h, 3)
    assert passes == P and np.array_equal(img.view(np.int32), acc.view(np.int32))


@pytest.mark.parametrize("use_kd", [0, 1])
def test_mesh_scene_pass(use_kd):
    """Glass diamond + subsurface coil + spheres: brute force (the reference's loop) and kd-tree."""
    w = h = 128
    cam = S.orbit_camera(w, h)
    sph = S.default_spheres()
    dv, dn = meshes.load_asset("diamond")
    dv = meshes.transform(dv, 0.8, (0.2, -2.2, 2.6))
    cv, cn = meshes.load_asset("coil")
    cv = meshes.transform(cv, 0.25, (-1.2, -2.2, 1.2))
    polys = np.concatenate([meshes.polys_from_triangles(dv, dn, S.MATERIALS["glass"]),
                            meshes.polys_from_triangles(cv, cn, S.MATERIALS["marble"])])
    depth = 8
    with api.Scene(sph, polys, api.render_params(max_depth=depth, row_stride_is_width=0, use_kdtree=use_kd)) as sc:
        prim, t = sc.primary_hits(cam, 1)
        check_against_oracle(sc, sph, polys, cam, 1, depth, 0)
    ref = O.render_pass(sph, polys, cam, 1, O.params(max_depth=depth), want_hits=True)
    assert np.array_equal(prim, ref["hits"]["prim"]) and np.array_equal(t.view(np.int32), ref["hits"]["t"].view(np.int32))
    assert ((prim >= 0) & (prim < polys.size)).sum() > 100      # the meshes are in view


def test_kdtree_equals_brute_force_on_device():
    """Closest-triangle queries: traversal must return the brute-force loop's answer, bit for bit."""
    v, n = meshes.load_asset("cow")
    v, n = meshes.subdivide(v, n, 1)                     # 23 216 triangles
    polys = meshes.polys_from_triangles(v, n, S.MATERIALS["glass"])
    rng = np.random.default_rng(7)
    m = 300000
    lo, hi = v.reshape(-1, 3).min(axis=0), v.reshape(-1, 3).max(axis=0)
    ctr, ext = (lo + hi) / 2, (hi - lo).max()
    o = ((rng.random((m, 3), dtype=np.float32) * 3 - 1.5) * ext + ctr).astype(np.float32)
    tgt = ((rng.random((m, 3), dtype=np.float32) - 0.5) * ext + ctr).astype(np.float32)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    # degenerate families: axis-parallel directions, origins on mesh vertices
    k = 20000
    rays[:k, 3:] = 0
    rays[np.arange(k), 3 + rng.integers(0, 3, k)] = rng.choice([-1.0, 1.0], k)
    rays[k:2 * k, :3] = v.reshape(-1, 3)[rng.integers(0, v.size // 3, k)]
    with api.Scene(None, polys, api.render_params(max_depth=1)) as sc:
        tk, ttk = sc.intersect(rays, use_kdtree=True)
        tb, ttb = sc.intersect(rays, use_kdtree=False)
    assert (tb >= 0).sum() > 50000
    assert np.array_equal(tk, tb)
    assert np.array_equal(ttk.view(np.int32), ttb.view(np.int32))
    # and the device brute force is the oracle's loop
    sub = slice(0, 4000)
    to, tto = O.closest_triangle(v, rays[sub])
    assert np.array_equal(to, tb[sub]) and np.array_equal(tto.view(np.int32), ttb[sub].view(np.int32))


@pytest.mark.parametrize("small", [2, 64, 4096])
def test_kd_kernel_leaf_sizes_and_duplicate_triangles(small):
    """The wave-cooperative triangle phase: leaves far larger than a 64-pair chunk (threshold 4096),
    tiny leaves, and a mesh in which every triangle exists twice (equal t: the lower index wins)."""
    v, n = meshes.load_asset("monkey")
    v2 = np.concatenate([v, v]).astype(np.float32)
    n2 = np.concatenate([n, n]).astype(np.float32)
    polys = meshes.polys_from_triangles(v2, n2, S.MATERIALS["glass"])
    rng = np.random.default_rng(11 + small)
    m = 120000
    pts = v.reshape(-1, 3)
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    ctr, ext = (lo + hi) / 2, (hi - lo).max()
    o = ((rng.random((m, 3), dtype=np.float32) * 3 - 1.5) * ext + ctr).astype(np.float32)
    tgt = pts[rng.integers(0, pts.shape[0], m)] + ((rng.random((m, 3), dtype=np.float32) - 0.5) * 0.05 * ext)
    d = (tgt - o).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    rays[:64, 3:] = np.nan                                  # rays the reference's loop can never hit with
    with api.Scene(None, polys, api.render_params(max_depth=1, kd_small_threshold=small)) as sc:
        tk, ttk = sc.intersect(rays, use_kdtree=True)
        tb, ttb = sc.intersect(rays, use_kdtree=False)
    hit = tb >= 0
    assert hit.sum() > m // 3 and (tb[:64] == -1).all()
    assert np.array_equal(tk, tb)
    assert np.array_equal(ttk.view(np.int32), ttb.view(np.int32))
    # duplicates: the winner is always the first copy
    assert (tb[hit] < polys.size // 2).all()
    sub = slice(0, 3000)
    to, tto = O.closest_triangle(v2, rays[sub])
    assert np.array_equal(to, tb[sub]) and np.array_equal(tto.view(np.int32), ttb[sub].view(np.int32))


@pytest.mark.parametrize("env", [{"GPT_LEAF_ORDER": "0"}, {"GPT_NO_LINE_FIT": "1"}])
def test_kd_kernel_alternative_leaf_layouts(env, monkeypatch):
    """The two other device layouts of the leaves' triangles -- an index list into per-triangle
    records (what a copy beyond 4 GiB falls back to) and packed runs that are not fitted to
    128-byte lines -- give the brute-force loop's hits as well."""
    for k, val in env.items():
        monkeypatch.setenv(k, val)
    v, n = meshes.load_asset("monkey")
    polys = meshes.polys_from_triangles(v, n, S.MATERIALS["glass"])
    rng = np.random.default_rng(5)
    m = 60000
    pts = v.reshape(-1, 3)
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    ctr, ext = (lo + hi) / 2, (hi - lo).max()
    o = ((rng.random((m, 3), dtype=np.float32) * 3 - 1.5) * ext + ctr).astype(np.float32)
    tgt = pts[rng.integers(0, pts.shape[0], m)] + ((rng.random((m, 3), dtype=np.float32) - 0.5) * 0.05 * ext)
    d = (tgt - o).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    with api.Scene(None, polys, api.render_params(max_depth=1)) as sc:
        tk, ttk = sc.intersect(rays, use_kdtree=True)
        tb, ttb = sc.intersect(rays, use_kdtree=False)
    assert (tb >= 0).sum() > m // 3
    assert np.array_equal(tk, tb)
    assert np.array_equal(ttk.view(np.int32), ttb.view(np.int32))


def test_image_accumulates_passes_in_order():
    """image += pass, float32, in counter order (path_tracer.cpp:100-103)."""
    w = h = 96
    cam = S.orbit_camera(w, h)
    sph = S.default_spheres()
    with api.Scene(sph, None, api.render_params(max_depth=6)) as sc:
        acc = np.zeros((w * h, 3), dtype=np.float32)
        for c in range(5):
            acc += sc.render_pass(cam, c)
        img, passes = sc.accum_download()
        assert passes == 5 and np.array_equal(img, acc)
        sc.reset(w * h)
        sc.render_range(cam, 0, 5)
        img2, passes2 = sc.accum_download()
        assert passes2 == 5 and np.array_equal(img2, acc)


def test_launch_kernel_drop_in():
    """The reference entry point with device-resident Sphere/Poly arrays."""
    w = h = 64
    cam = S.orbit_camera(w, h)
    sph = S.default_spheres()
    bv, bn = meshes.load_asset("box")
    bv = meshes.transform(bv, 0.3, (0.0, -0.5, 1.8))
    polys = meshes.polys_from_triangles(bv, bn, S.MATERIALS["glass"])
    d_s = api.device_alloc(sph.nbytes)
    d_p = api.device_alloc(polys.nbytes)
    api.device_upload(d_s, sph)
    api.device_upload(d_p, polys)
    try:
        api.launch_kernel_params(api.render_params(max_depth=8, row_stride_is_width=0))
        px0 = api.launch_kernel(polys.size, d_p, sph.size, d_s, w * h, 0, cam)
        px1 = api.launch_kernel(polys.size, d_p, sph.size, d_s, w * h, 1, cam)
        with api.Scene(sph, polys, api.render_params(max_depth=8, row_stride_is_width=0)) as sc:
            want0 = sc.render_pass(cam, 0)
            want1 = sc.render_pass(cam, 1)
        assert np.array_equal(px0, want0) and np.array_equal(px1, want1)
        # a run of consecutive counters is answered from batches traced ahead; every call still returns
        # exactly its own pass, also after the camera moves or the counter jumps
        cam2 = S.orbit_camera(w, h, yaw=0.4)
        calls = [(c, cam) for c in range(2, 14)] + [(14, cam2), (15, cam2), (16, cam2), (3, cam2), (17, cam), (18, cam)]
        got = [api.launch_kernel(polys.size, d_p, sph.size, d_s, w * h, c, k) for c, k in calls]
        with api.Scene(sph, polys, api.render_params(max_depth=8, row_stride_is_width=0)) as sc:
            for (c, k), px in zip(calls, got):
                assert np.array_equal(px, sc.render_pass(k, c)), c
        with api.Scene(sph, polys, api.render_params(max_depth=8, row_stride_is_width=0)) as sc:
            got, _ = check_against_oracle(sc, sph, polys, cam, 0, 8, 0)
            assert np.array_equal(px0, got)
        # a changed scene behind the same pointers is picked up
        sph2 = sph.copy()
        sph2[0]["radius"] = 0.5
        api.device_upload(d_s, sph2)
        px2 = api.launch_kernel(polys.size, d_p, sph.size, d_s, w * h, 0, cam)
        assert not np.array_equal(px2, px0)
        # ... also in the middle of a run of consecutive counters, whose passes were traced ahead from the old
        # contents: the very next call renders the new ones (the reference reads its arrays on every call)
        for c in range(1, 8):
            api.launch_kernel(polys.size, d_p, sph.size, d_s, w * h, c, cam)
        api.device_upload(d_s, sph)                          # back to the original spheres
        late = [api.launch_kernel(polys.size, d_p, sph.size, d_s, w * h, c, cam) for c in range(8, 8 + 5)]
        with api.Scene(sph, polys, api.render_params(max_depth=8, row_stride_is_width=0)) as sc:
            for k, px in enumerate(late):
                assert np.array_equal(px, sc.render_pass(cam, 8 + k)), k
        # a caller that declares the scene static is not re-read: stale by its own promise, then exact again
        api.launch_kernel_static_scene(1)
        try:
            api.device_upload(d_s, sph2)
            stale = api.launch_kernel(polys.size, d_p, sph.size, d_s, w * h, 13, cam)
            with api.Scene(sph, polys, api.render_params(max_depth=8, row_stride_is_width=0)) as sc:
                assert np.array_equal(stale, sc.render_pass(cam, 13))
        finally:
            api.launch_kernel_static_scene(0)
        fresh = api.launch_kernel(polys.size, d_p, sph.size, d_s, w * h, 14, cam)
        with api.Scene(sph2, polys, api.render_params(max_depth=8, row_stride_is_width=0)) as sc:
            assert np.array_equal(fresh, sc.render_pass(cam, 14))
    finally:
        api.launch_kernel_params(None)
        api.device_free(d_s)
        api.device_free(d_p)


def test_non_square_stride_modes():
    """Width-stride fix vs the reference's arithmetic: identical on squares, both match the oracle."""
    sph = S.default_spheres()
    cam = S.orbit_camera(160, 90)
    for mode in (0, 1):
        with api.Scene(sph, None, api.render_params(max_depth=6, row_stride_is_width=mode)) as sc:
            check_against_oracle(sc, sph, None, cam, 0, 6, mode)


def test_queue_dump_is_the_camera_rays_pixel_major():
    """gpt_debug_dump_queue at depth 0: slot i holds pass i % P of pixel i // P, and the rays are the oracle's (lens on)."""
    w, h, P = 64, 48, 4
    cam = S.orbit_camera(w, h)
    with api.Scene(S.default_spheres(), None, api.render_params(max_depth=4, passes_in_flight=P)) as sc:
        rays, live = sc.dump_queue(cam, 10, P, 0, w * h * P)
        later, live2 = sc.dump_queue(cam, 10, P, 2, w * h * P)
    assert live == w * h * P and rays.shape[0] == live
    for k in range(P):
        ref = O.camera_rays(cam, 10 + k, O.params(row_stride_is_width=1))
        assert np.array_equal(rays[k::P].view(np.uint32), ref.view(np.uint32))
    assert 0 < live2 < live and later.shape[0] == live2 and np.isfinite(later).all()


def test_depth_zero_renders_black():
    """No bounce at all: the pass is black and still counted (nothing initialises the pass radiance but the library)."""
    w = h = 64
    cam = S.orbit_camera(w, h)
    with api.Scene(S.default_spheres(), None, api.render_params(max_depth=0)) as sc:
        px = sc.render_pass(cam, 0)
        sc.render_range(cam, 1, 4)
        img, passes = sc.accum_download()
    assert not px.any() and not img.any() and passes == 4


def test_invalid_arguments_are_reported():
    sph = S.default_spheres()
    with pytest.raises(api.GptError):
        api.Scene(sph, None, api.render_params(max_depth=1000))
    with api.Scene(sph, None, api.render_params(max_depth=2)) as sc:
        sc.reset(64 * 64)
        cam = S.orbit_camera(32, 32)
        with pytest.raises(api.GptError):
            # resolution does not match the buffers
            from gpupathtracer_amd import _lib
            import ctypes
            c = np.array(cam, dtype=api.CAMERA_DTYPE).reshape(1)
            api._check(_lib.lib().gpt_render_pass(sc._h, c.ctypes.data_as(ctypes.c_void_p), 0, None))
