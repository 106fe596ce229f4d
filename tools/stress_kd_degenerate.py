"""Stress run on the GPU box: kd traversal vs the brute-force loop, bit for bit, on ray families chosen to sit on the
traversal's decision boundaries -- origins on vertices (= on split planes), directions with zero, denormal and tiny
components, rays aimed exactly at vertices and edge midpoints (equal-t ties between neighbouring triangles), far-away
origins (family 7: 10 000 extents away; there the brute-force loop itself starts to report triangles the ray does not
pass near -- see tools/kd_distance_sweep.py -- so fine meshes are expected to differ in that family) -- against meshes whose triangles lie IN axis-aligned planes (box, a grid of quads) and ordinary ones, at scales
1e-3, 1 and 1e3 with a large offset."""
import numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gpupathtracer_amd import api, meshes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from kd_ray_families import grid_mesh, families


rng = np.random.default_rng(2026)
cases = {"box": meshes.load_asset("box"), "grid": grid_mesh(), "monkey": meshes.load_asset("monkey")}
cv, cn = meshes.load_asset("cow"); cases["cow x4"] = meshes.subdivide(cv, cn, 1)
bad = 0
for name, (v, n) in cases.items():
    for scale, shift in ((1.0, 0.0), (1e-3, 0.0), (1e3, 5000.0)):
        vv = (v * np.float32(scale) + np.float32(shift)).astype(np.float32)
        polys = meshes.polys_from_triangles(vv, n, api.material_preset("glass"))
        rays = families(vv, rng, 400000 if len(vv) < 30000 else 200000)
        if name not in ("box", "grid") and not os.environ.get("STRESS_FAR"):
            rays = rays[: 7 * (len(rays) // 8)]      # family 7 only against large triangles (see the docstring)
        for small in (0, 2):
            with api.Scene(None, polys, api.render_params(max_depth=1, **({"kd_small_threshold": small} if small else {}))) as sc:
                tk, ttk = sc.intersect(rays, use_kdtree=True)
                tb, ttb = sc.intersect(rays, use_kdtree=False)
            ok = np.array_equal(tk, tb) and np.array_equal(ttk.view(np.int32), ttb.view(np.int32))
            diff = np.nonzero((tk != tb) | (ttk.view(np.int32) != ttb.view(np.int32)))[0]
            print("%-7s scale %-6g small %d: %d rays, %d hits, %s%s" % (name, scale, small, len(rays), int((tb >= 0).sum()), "equal" if ok else "DIFFER",
                  "" if ok else " at %s (families %s)" % (diff[:5], sorted(set((diff // (len(rays) // 8)).tolist())))), flush=True)
            if not ok and os.environ.get("STRESS_VERBOSE"):
                for i in diff[:4]:
                    print("   ray", i, rays[i].tolist(), "kd", int(tk[i]), float(ttk[i]), "brute", int(tb[i]), float(ttb[i]), flush=True)
            bad += 0 if ok else 1
print("stress ok" if bad == 0 else "stress FAILED: %d cases" % bad)
sys.exit(1 if bad else 0)
