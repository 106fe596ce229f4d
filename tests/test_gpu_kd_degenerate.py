"""GPU: kd traversal == brute-force loop, bit for bit, on rays chosen to sit on the traversal's decision boundaries:
origins on vertices (i.e. on split planes), axis-parallel directions, directions with zero / denormal / tiny components,
rays aimed exactly at vertices and edge midpoints (equal-t ties between neighbouring triangles), against meshes whose
triangles lie in axis-aligned planes and an ordinary one, at three scales.  Origins 10 000 extents away are asserted equal
here for the two meshes of large triangles; for the monkey that family has its own test
(tests/test_gpu_kd_exactness.py: from that far the brute-force loop's own float arithmetic starts to accept small
triangles seen edge-on that the ray does not pass near, and the test certifies every differing ray as such a case)."""
import numpy as np
import pytest

import scenes as S
from kd_ray_families import families, grid_mesh
from gpupathtracer_amd import api, meshes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["box", "grid", "monkey"])
def test_kd_equals_brute_force_on_boundary_rays(name):
    v, n = grid_mesh() if name == "grid" else meshes.load_asset(name)
    rng = np.random.default_rng(2026)
    for scale, shift in ((1.0, 0.0), (1e-3, 0.0), (1e3, 5000.0)):
        vv = (v * np.float32(scale) + np.float32(shift)).astype(np.float32)
        polys = meshes.polys_from_triangles(vv, n, S.MATERIALS["glass"])
        rays = families(vv, rng, 160000)
        if name == "monkey":
            rays = rays[: 7 * (len(rays) // 8)]          # the far-away family: test_gpu_kd_exactness.py
        for small in (0, 2):
            kw = {"kd_small_threshold": small} if small else {}
            with api.Scene(None, polys, api.render_params(max_depth=1, **kw)) as sc:
                tk, ttk = sc.intersect(rays, use_kdtree=True)
                tb, ttb = sc.intersect(rays, use_kdtree=False)
            assert (tb >= 0).sum() > len(rays) // 5
            assert np.array_equal(tk, tb), (name, scale, small)
            assert np.array_equal(ttk.view(np.int32), ttb.view(np.int32)), (name, scale, small)
