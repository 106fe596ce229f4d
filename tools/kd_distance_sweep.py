"""Where does the kd traversal stop being equal to the brute-force loop?  Rays aimed at mesh vertices from a sphere of
radius R root-box extents around the mesh: 200 000 per distance, traced both ways on the device.  The differences at
large R are the brute-force loop's own: for a small triangle seen almost edge-on from far away its float arithmetic loses
all digits of the barycentric coordinates and now and then accepts a triangle the ray does not pass near; a traversal
never tests that triangle.  tests/test_gpu_kd_exactness.py asserts 0 differences up to the supported distance."""
import numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from gpupathtracer_amd import api, meshes
from kd_ray_families import aimed_at_vertices
rng = np.random.default_rng(5)
cv, cn = meshes.load_asset("cow")
which = sys.argv[1:] or ["cow x4", "monkey", "cow x64", "cow x256"]
dists = [float(x) for x in os.environ.get("DISTS", "1 2 4 8 16 32 100 1000 10000").split()]
sets = {"cow x4": lambda: meshes.subdivide(cv, cn, 1), "monkey": lambda: meshes.load_asset("monkey"),
        "cow x64": lambda: meshes.subdivide(cv, cn, 3), "cow x256": lambda: meshes.subdivide(cv, cn, 4)}
for name in which:
    v, n = sets[name]()
    polys = meshes.polys_from_triangles(v, n, api.material_preset("glass"))
    with api.Scene(None, polys, api.render_params(max_depth=1)) as sc:
        for dist in dists:
            rays = aimed_at_vertices(v, rng, 200000, dist)
            tk, ttk = sc.intersect(rays, use_kdtree=True); tb, ttb = sc.intersect(rays, use_kdtree=False)
            bad = int(((tk != tb) | (ttk.view(np.int32) != ttb.view(np.int32))).sum())
            print("%-8s origin at %6g extents: %d of %d rays differ (%d hits)" % (name, dist, bad, len(rays), int((tb >= 0).sum())), flush=True)
