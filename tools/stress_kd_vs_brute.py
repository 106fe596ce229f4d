"""One-off stress run on the GPU box: kd traversal vs the brute-force loop, bit for bit, on 1.5 M random rays
(a third of them starting on the surface) against cow.obj subdivided 3x (371 456 triangles)."""
import numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gpupathtracer_amd import api, meshes
v, n = meshes.load_asset("cow"); v, n = meshes.subdivide(v, n, 3)
polys = meshes.polys_from_triangles(v, n, api.material_preset("glass"))
rng = np.random.default_rng(123)
m = 1500000
pts = v.reshape(-1, 3); lo, hi = pts.min(0), pts.max(0); ctr, ext = (lo + hi) / 2, (hi - lo).max()
o = ((rng.random((m, 3), dtype=np.float32) * 2.4 - 1.2) * ext + ctr).astype(np.float32)
o[: m // 3] = pts[rng.integers(0, pts.shape[0], m // 3)]            # origins on the surface (secondary rays)
d = rng.normal(size=(m, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o, d], 1).astype(np.float32)
for small in (0, 8):
    with api.Scene(None, polys, api.render_params(max_depth=1, **({"kd_small_threshold": small} if small else {}))) as sc:
        tk, ttk = sc.intersect(rays, use_kdtree=True)
        tb, ttb = sc.intersect(rays, use_kdtree=False)
    print("small", small, "hits", int((tb >= 0).sum()), "tri equal", bool(np.array_equal(tk, tb)), "t equal", bool(np.array_equal(ttk.view(np.int32), ttb.view(np.int32))))
    assert np.array_equal(tk, tb) and np.array_equal(ttk.view(np.int32), ttb.view(np.int32))
print("stress ok")
