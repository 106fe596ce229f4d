"""How often does the kd traversal's closest hit differ from the brute-force loop's on a real frame?  The primary rays of
one 1080p pass of the bench scene (cow subdivided SUBDIV times + the 14 spheres) are dumped from the path queue and
traced both ways.  The brute-force loop tests EVERY triangle with every ray; for a triangle seen almost edge-on from far
away its float arithmetic loses all digits of the barycentric coordinates and now and then accepts a triangle the ray
does not pass anywhere near.  A traversal never tests such a triangle, so on those rays -- and only there -- it returns
the geometrically right hit instead of the reference's artefact."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpupathtracer_amd import api
import bench
subdiv = int(os.environ.get("SUBDIV", "4"))
w, h = 1920, 1080
spheres, polys = bench.build_workload("cow_heavy", subdiv)
cam = api.default_camera(w, h)
with api.Scene(spheres, polys, api.render_params(max_depth=8)) as sc:
    sc.reset(w * h)
    for depth in (0, 1, 2):
        r, live = sc.dump_queue(cam, 0, 1, depth, w * h)
        n = len(r)
        tk, ttk = sc.intersect(r, use_kdtree=True)
        tb, ttb = sc.intersect(r, use_kdtree=False)
        bad = np.nonzero((tk != tb) | (ttk.view(np.int32) != ttb.view(np.int32)))[0]
        closer = int((ttb[bad] < ttk[bad]).sum())
        print("depth %d: %d rays, %d triangle hits, kd differs from the brute-force loop on %d (%.2e of rays; brute-force hit closer on %d)"
              % (depth, n, int((tb >= 0).sum()), len(bad), len(bad) / max(1, n), closer), flush=True)
