"""GPU: full-size configurations.  The oracle cannot render 2 M pixels x 1.5 M
triangles in test time, so full-size runs are checked (a) against the oracle on strided
pixel subsets (every pixel of a pass is independent of every other), (b) kd traversal
against the device's own brute-force loop on the frame's primary rays, and (c) through
size-independent properties: determinism, independence from the number of passes traced
per launch, image = sum of passes, pass-count bookkeeping."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import parity
import scenes as S
from gpupathtracer_amd import api, meshes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strided_ids(w, h, stride):
    ys, xs = np.meshgrid(np.arange(stride // 2, h, stride), np.arange(stride // 2, w, stride), indexing="ij")
    return (ys * w + xs).astype(np.int32).ravel()


def check_subset(sc, sph, polys, cam, counter, depth, ids):
    """One full-frame pass on the GPU; paths and radiance of the pixels `ids` against the oracle."""
    got, ev, pr = sc.render_events(cam, counter)
    ref = O.render_pass(sph, polys, cam, counter, O.params(max_depth=depth, row_stride_is_width=1), pixel_ids=ids,
                        want_events=True, want_hits=True)
    parity.check_pass(got, ev, pr, ref, ids=ids)
    return got, pr, ref


def test_spheres_1080p_against_oracle_subset():
    """Config: the reference's default scene at 1920x1080 (width-stride fix), depth 8."""
    w, h, depth = 1920, 1080, 8
    cam = S.orbit_camera(w, h)
    sph = S.default_spheres()
    ids = strided_ids(w, h, 13)
    with api.Scene(sph, None, api.render_params(max_depth=depth)) as sc:
        got, _, _ = check_subset(sc, sph, None, cam, 4, depth, ids)
        again = sc.render_pass(cam, 4)
        st = sc.stats()
    assert np.array_equal(got, again)                       # deterministic
    assert st["live_at_depth"][0] == 2 * w * h
    # bounces per path of the default scene at depth 8 (reference run: 2.69 at 512x512)
    assert 2.4 < st["path_bounces"] / (2 * w * h) < 3.0


def test_cow_heavy_1080p_kd_vs_brute_and_oracle():
    """Config (the benchmark's workload): cow subdivided 4x (1 485 824 triangles, glass) + default spheres,
    1080p, depth 8, on the tree the benchmark runs on."""
    w, h, depth = 1920, 1080, 8
    cam = S.orbit_camera(w, h)
    sph = S.default_spheres()
    v, n = meshes.load_asset("cow")
    v, n = meshes.subdivide(v, n, 4)
    assert v.shape[0] == 1485824
    polys = meshes.polys_from_triangles(v, n, S.MATERIALS["glass"])
    with api.Scene(sph, polys, api.render_params(max_depth=depth)) as sc:
        # traversal vs the brute-force loop on the device, bit-exact: primary rays of a strided pixel subset,
        # and rays between random points around the mesh (what bounces inside the glass look like)
        ids = strided_ids(w, h, 9)
        rays = O.camera_rays(cam, 0, O.params(row_stride_is_width=1), pixel_ids=ids)
        rng = np.random.default_rng(3)
        pts = v.reshape(-1, 3)
        lo, hi = pts.min(axis=0), pts.max(axis=0)
        m = 12000
        o = (lo + (hi - lo) * rng.random((m, 3), dtype=np.float32)).astype(np.float32)
        d = pts[rng.integers(0, pts.shape[0], m)] - o
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        rays = np.concatenate([rays, np.concatenate([o, d], axis=1)]).astype(np.float32)
        tk, ttk = sc.intersect(rays, use_kdtree=True)
        tb, ttb = sc.intersect(rays, use_kdtree=False)
        assert np.array_equal(tk, tb) and np.array_equal(ttk.view(np.int32), ttb.view(np.int32))
        assert (tb[:ids.size] >= 0).sum() > 500 and (tb[ids.size:] >= 0).sum() > m // 2
        # and the device's brute force is the oracle's loop
        to, tto = O.closest_triangle(v, rays[-300:])
        assert np.array_equal(to, tb[-300:]) and np.array_equal(tto.view(np.int32), ttb[-300:].view(np.int32))
        # paths, first-hit triangle indices and radiance of ~3 000 pixels against the oracle (brute force)
        ids = strided_ids(w, h, 26)
        got, pr, ref = check_subset(sc, sph, polys, cam, 0, depth, ids)
        info = sc.info()
    assert info["kd_depth"] <= 56 and info["kd_nodes"] > 1000
    on_mesh = (ref["hits"]["prim"] >= 0) & (ref["hits"]["prim"] < polys.size)
    assert on_mesh.sum() > 50
    assert np.array_equal(pr[ids][:, 0], ref["hits"]["prim"])


def test_diamond_coil_1080p_against_oracle_subset():
    """Config: glass diamond + subsurface (marble) coil moved into view, lens on, 1080p, depth 8 -- the long refractive and
    in-medium tails that stress the compaction: paths, radiance and the live count of every depth on a pixel subset."""
    w, h, depth = 1920, 1080, 8
    cam = S.orbit_camera(w, h)
    sph = S.default_spheres()
    dv, dn = meshes.load_asset("diamond")
    cv, cn = meshes.load_asset("coil")
    polys = np.concatenate([
        meshes.polys_from_triangles(meshes.transform(dv, 0.8, (0.2, -2.2, 2.6)), dn, S.MATERIALS["glass"]),
        meshes.polys_from_triangles(meshes.transform(cv, 0.25, (-1.2, -2.2, 1.2)), cn, S.MATERIALS["marble"])])
    ids = strided_ids(w, h, 11)
    with api.Scene(sph, polys, api.render_params(max_depth=depth)) as sc:
        for counter in (0, 3):
            got, pr, ref = check_subset(sc, sph, polys, cam, counter, depth, ids)
        _, ev, _ = sc.render_events(cam, 3)
    assert (ref["events"] & 7 == 1).sum() > 50                    # scattering inside the marble happens
    assert np.array_equal((ev[ids] > 0).sum(axis=0), ref["live"])  # live paths entering every depth
    assert ref["live"][-1] > 0.02 * ids.size                       # and a tail still alive at the last depth


def test_passes_per_launch_do_not_change_the_image():
    w, h = 640, 360
    cam = S.orbit_camera(w, h)
    sph = S.default_spheres()
    v, n = meshes.load_asset("monkey")
    v = meshes.transform(v, 0.5, (0.0, -0.2, 1.5))
    polys = meshes.polys_from_triangles(v, n, S.MATERIALS["something"])
    images = []
    for batch in (1, 5, 16):
        with api.Scene(sph, polys, api.render_params(max_depth=10, passes_in_flight=batch)) as sc:
            sc.reset(w * h)
            sc.render_range(cam, 3, 3 + 11)
            img, passes = sc.accum_download()
            st = sc.stats()
            assert passes == 11 and st["passes"] == 11
            images.append((img, st["path_bounces"], st["live_at_depth"].copy()))
    for img, pb, live in images[1:]:
        assert np.array_equal(img, images[0][0]) and pb == images[0][1] and np.array_equal(live, images[0][2])


def test_box_scene_512_64spp_rmse_against_oracle():
    """Config: Cornell-style room built from the reference's box.obj (tests/scenes.py: cornell_box) around the
    default spheres, 512x512, 64 spp, depth 8.  The image after 64 passes against the oracle's: both add the
    passes in counter order in float32."""
    w = h = 512
    spp, depth = 64, 8
    cam = S.orbit_camera(w, h)
    sph = S.default_spheres()
    bv, bn = meshes.load_asset("box")
    polys = S.cornell_box(bv, bn)
    with api.Scene(sph, polys, api.render_params(max_depth=depth)) as sc:
        got0, ev, pr = sc.render_events(cam, 0)
        ref0 = O.render_pass(sph, polys, cam, 0, O.params(max_depth=depth, row_stride_is_width=1), want_events=True)
        parity.check_pass(got0, ev, pr, ref0)
        assert ((pr[:, 0] >= 0) & (pr[:, 0] < polys.size)).mean() > 0.2    # the walls are in view of the primary rays
        sc.reset(w * h)
        sc.render_range(cam, 0, spp)
        img, passes = sc.accum_download()
    assert passes == spp
    ref = np.zeros((w * h, 3), np.float32)
    p = O.params(max_depth=depth, row_stride_is_width=1)
    for c in range(spp):
        ref += O.render_pass(sph, polys, cam, c, p)["rgb"]
    a, b = img / spp, ref / spp
    rmse = float(np.sqrt(np.mean((a.astype(np.float64) - b) ** 2)))
    # stated bound of the converged-image comparison: RMSE 2e-4 (Monte-Carlo noise of this image at 64 spp: ~0.05).
    # Every sample equals the oracle's bit for bit and both sides add the passes in the same order, so the
    # images are in fact identical, and so are their 8-bit tone-mapped versions.
    assert rmse < 2e-4, rmse
    assert parity.bits_equal(img, ref), "image differs from the oracle's in %d pixels, RMSE %g" % (int((img != ref).any(axis=1).sum()), rmse)
    assert np.array_equal(api.tonemap_8bit(img, spp), O.tonemap_8bit(ref, spp))


def test_headless_renderer_cli(tmp_path):
    exe = os.path.join(ROOT, "gpupathtracer_amd", "lib", "gpt_render")
    assert os.path.exists(exe), "gpt_render has not been built (make -C gpupathtracer_amd/csrc)"
    ppm = str(tmp_path / "o.ppm")
    pfm = str(tmp_path / "o.pfm")
    out = subprocess.run([exe, "--width", "96", "--height", "64", "--spp", "6", "--depth", "8", "--out", ppm,
                          "--pfm", pfm], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "Iterations: 6" in out.stdout
    cam = api.orbit_camera(96, 64)
    with api.Scene(api.default_scene(), None, api.render_params(max_depth=8)) as sc:
        sc.reset(96 * 64)
        sc.render_range(cam, 0, 6)
        img, passes = sc.accum_download()
    raw = open(ppm, "rb").read()
    header = b"P6\n96 64\n255\n"
    assert raw.startswith(header)
    assert raw[len(header):] == api.tonemap_8bit(img, 6).tobytes()


def test_headless_renderer_named_scene(tmp_path):
    """`gpt_render --scene Monkey --assets DIR`: a named scene through the CLI equals the same scene through the API."""
    from test_cabi import write_obj
    exe = os.path.join(ROOT, "gpupathtracer_amd", "lib", "gpt_render")
    v, n = meshes.load_asset("monkey")
    write_obj(str(tmp_path / "monkey.obj"), v, n)
    ppm = str(tmp_path / "m.ppm")
    out = subprocess.run([exe, "--width", "96", "--height", "64", "--spp", "4", "--depth", "8", "--out", ppm,
                          "--scene", "Monkey", "--assets", str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "Scene Monkey: 14 spheres, 1936 triangles" in out.stdout
    sph, polys, view = api.scene_preset("Monkey", str(tmp_path))
    cam = api.orbit_camera(96, 64, **view)
    with api.Scene(sph, polys, api.render_params(max_depth=8)) as sc:
        sc.reset(96 * 64)
        sc.render_range(cam, 0, 4)
        img, passes = sc.accum_download()
    raw = open(ppm, "rb").read()
    header = b"P6\n96 64\n255\n"
    assert raw.startswith(header) and raw[len(header):] == api.tonemap_8bit(img, 4).tobytes()


def test_synthetic_soup_10m_4k():
    """Config: synthetic soup of 10 M triangles at 3840x2160 (the benchmark's `soup` workload)."""
    w, h, depth = 3840, 2160, 8
    cam = S.orbit_camera(w, h)
    sph = S.default_spheres()
    v, n = meshes.random_soup(10000000, seed=12345, extent=1.0, edge=0.01)
    v = meshes.transform(v, 0.75, (0.0, 0.1, 0.0))
    polys = meshes.polys_from_triangles(v, n, S.MATERIALS["white"])
    with api.Scene(sph, polys, api.render_params(max_depth=depth, passes_in_flight=2)) as sc:
        ids = strided_ids(w, h, 24)
        rays = O.camera_rays(cam, 1, O.params(row_stride_is_width=1), pixel_ids=ids)
        tk, ttk = sc.intersect(rays, use_kdtree=True)
        to, tto = O.closest_triangle(v, rays[:600])
        assert np.array_equal(tk[:600], to) and np.array_equal(ttk[:600].view(np.int32), tto.view(np.int32))
        assert (tk >= 0).sum() > 1000
        sc.reset(w * h)
        sc.render_range(cam, 0, 2)
        img, passes = sc.accum_download()
        st = sc.stats()
        assert passes == 2 and st["live_at_depth"][0] == 2 * w * h
        assert np.isfinite(img).all() and img.min() >= 0 and 0.05 < img.mean() / 2 < 1.0
        # paths and radiance of ~1 000 pixels of one pass against the oracle (brute force over 10 M triangles)
        ids = strided_ids(w, h, 90)
        check_subset(sc, sph, polys, cam, 0, depth, ids)
