"""GPU: gpt_intersect_noalloc, the closest-hit query that allocates nothing, against
gpt_debug_intersect (the entry point it is a variant of) through the C-ABI: same rays,
same triangle ids and the same t bits, on two calls in a row, the second with a smaller
input (the scratch array keeps the first call's contents)."""
import numpy as np
import pytest

import scenes as S
from gpupathtracer_amd import api, meshes

pytestmark = pytest.mark.gpu


def _rays(v, m, seed):
    rng = np.random.default_rng(seed)
    pts = v.reshape(-1, 3)
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    ctr, ext = (lo + hi) / 2, (hi - lo).max()
    o = ((rng.random((m, 3), dtype=np.float32) * 3 - 1.5) * ext + ctr).astype(np.float32)
    tgt = pts[rng.integers(0, pts.shape[0], m)] + ((rng.random((m, 3), dtype=np.float32) - 0.5) * 0.05 * ext)
    d = (tgt - o).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)


def test_noalloc_intersect_equals_debug_intersect_twice():
    v, n = meshes.load_asset("cow")
    polys = meshes.polys_from_triangles(v, n, S.MATERIALS["glass"])
    big, small = api.NOALLOC_MAX_RAYS, 1000
    d_rays = api.device_alloc(big * 24)
    d_tri = api.device_alloc(big * 4)
    d_t = api.device_alloc(big * 4)
    try:
        with api.Scene(None, polys, api.render_params(max_depth=1)) as sc:
            for m, seed, use_kd in ((big, 21, True), (small, 22, True), (big, 23, False), (777, 24, True)):
                rays = _rays(v, m, seed)
                rays[:3, 3:] = np.nan                       # rays that can hit nothing
                api.device_upload(d_rays, rays)
                # poison the outputs so that stale values of the previous call cannot pass
                api.device_upload(d_tri, np.full(big, 12345, np.int32))
                api.device_upload(d_t, np.full(big, -7.0, np.float32))
                api.intersect_noalloc(sc, d_rays, m, d_tri, d_t, use_kdtree=use_kd)
                tri = api.device_download(d_tri, big * 4, np.int32)
                t = api.device_download(d_t, big * 4, np.float32)
                want_tri, want_t = sc.intersect(rays, use_kdtree=use_kd)
                assert (want_tri >= 0).sum() > m // 3
                assert np.array_equal(tri[:m], want_tri)
                assert np.array_equal(t[:m].view(np.int32), want_t.view(np.int32))
                assert (tri[m:] == 12345).all() and (t[m:] == -7.0).all()      # nothing beyond n is written
            # limits are reported, not silently clipped
            with pytest.raises(api.GptError) as e:
                api.intersect_noalloc(sc, d_rays, big + 1, d_tri, d_t)
            assert e.value.status == 7
            api.intersect_noalloc(sc, d_rays, 0, d_tri, d_t)               # empty input: a no-op
        # a scene without triangles: every ray misses
        with api.Scene(S.default_spheres(), None, api.render_params(max_depth=1)) as sc:
            api.intersect_noalloc(sc, d_rays, small, d_tri, d_t)
            assert (api.device_download(d_tri, small * 4, np.int32) == -1).all()
    finally:
        api.device_free(d_rays)
        api.device_free(d_tri)
        api.device_free(d_t)
