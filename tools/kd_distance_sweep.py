import numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gpupathtracer_amd import api, meshes
rng = np.random.default_rng(5)
cv, cn = meshes.load_asset("cow")
for name, (v, n) in {"cow x4": meshes.subdivide(cv, cn, 1), "monkey": meshes.load_asset("monkey"), "cow x64": meshes.subdivide(cv, cn, 3)}.items():
    polys = meshes.polys_from_triangles(v, n, api.material_preset("glass"))
    pts = v.reshape(-1, 3); lo, hi = pts.min(0), pts.max(0); ctr, ext = (lo + hi) / 2, float((hi - lo).max())
    with api.Scene(None, polys, api.render_params(max_depth=1)) as sc:
        for dist in (1, 10, 100, 1000, 10000, 100000):
            m = 200000
            u = rng.normal(size=(m, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
            o = (ctr + u * ext * dist).astype(np.float32)
            d = pts[rng.integers(0, len(pts), m)] - o
            d = (d / np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
            rays = np.concatenate([o, d], 1).astype(np.float32)
            tk, ttk = sc.intersect(rays, use_kdtree=True); tb, ttb = sc.intersect(rays, use_kdtree=False)
            bad = int(((tk != tb) | (ttk.view(np.int32) != ttb.view(np.int32))).sum())
            print("%-8s origin at %6g extents: %d of %d rays differ (%d hits)" % (name, dist, bad, m, int((tb >= 0).sum())), flush=True)
