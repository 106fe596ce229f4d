"""Experiment: how much of the traversal's walk is the first descent?  For the rays entering depths 1..7 of a 1080p pass
of the bench scene: the share whose origin lies inside the tree's root box, the tree depth of the leaf that holds the origin
(= the node steps a root-to-leaf descent costs such a ray before its first triangle test), and next to it the node visits per
ray the instrumented kernel counts for the whole queue."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpupathtracer_amd import api
import bench
subdiv = int(os.environ.get("SUBDIV", "4"))
w, h = 1920, 1080
spheres, polys = bench.build_workload(os.environ.get("WORKLOAD", "cow_heavy"), subdiv)
cam = api.default_camera(w, h)
verts = np.concatenate([polys["p0"], polys["p1"], polys["p2"]], axis=1)
tree = api.kdtree_build(verts, small_threshold=int(os.environ.get("KD_SMALL", "12")), on_gpu=True)
nodes, bounds = tree["nodes"], tree["bounds"]
with api.Scene(spheres, polys, api.render_params(max_depth=8)) as sc:
    sc.reset(w * h)
    tot_desc = tot_rays = 0
    for depth in (1, 2, 3, 5, 7):
        rays, live = sc.dump_queue(cam, 0, 1, depth, w * h)
        n_all = len(rays)
        if len(rays) > 400000: rays = rays[:: len(rays) // 400000 + 1]
        o = rays[:, :3].astype(np.float64)
        inside = ((o >= bounds[:3]) & (o <= bounds[3:])).all(1)
        node = np.zeros(len(o), np.int64)
        alive = inside.copy()
        leaf_depth = np.zeros(len(o), np.int64)
        for L in range(1, 64):
            w0, w1 = nodes[node, 0], nodes[node, 1]
            axis = (w0 & 3).astype(np.int64)
            inner = alive & (axis != 3)
            if not inner.any(): break
            split = w1.view(np.float32).astype(np.float64)
            left = (w0 >> 2).astype(np.int64)
            idx = np.flatnonzero(inner)
            go_right = o[idx, axis[idx]] >= split[idx]
            node[idx] = left[idx] + go_right
            leaf_depth[idx] = L
            alive = inner
        ld = leaf_depth[inside]
        print("depth %d: %d rays, %.2f origin in root box; depth of the origin's leaf: mean %.1f median %d p90 %d; descent steps per ray of the queue %.1f"
              % (depth, n_all, inside.mean(), ld.mean(), np.median(ld), np.percentile(ld, 90), ld.sum() / len(o)), flush=True)
