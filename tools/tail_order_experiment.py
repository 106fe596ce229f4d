"""GPU experiment: what would the traversal kernel gain if the rays that start inside the mesh's box -- the long
traversals -- came FIRST in every part of the queue (a launch ends in a tail of its longest rays; started early they would
finish under the short ones)?  Takes the real queue entering depth D of a cow_heavy batch and traces it with
gpt_debug_intersect (the production kernel) in queue order, with the inside rays first in each of the kernel's eight
queue parts, and with them last.  Kernel times come from a rocprofv3 --kernel-trace run around this script
(k_intersect_kd<false, true> launches in the order printed here)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gpupathtracer_amd import api
import bench

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
M = int(sys.argv[2]) if len(sys.argv) > 2 else 16000000
P = int(sys.argv[3]) if len(sys.argv) > 3 else 64
spheres, polys = bench.build_workload("cow_heavy", 4)
cam = api.default_camera(1920, 1080)
sc = api.Scene(spheres, polys, api.render_params(max_depth=8, passes_in_flight=P))
sc.reset(1920 * 1080)
rays, live = sc.dump_queue(cam, 0, P, depth, M)
verts = np.concatenate([polys["p0"], polys["p1"], polys["p2"]], axis=0)
lo, hi = verts.min(axis=0), verts.max(axis=0)
inside = ((rays[:, :3] >= lo) & (rays[:, :3] <= hi)).all(axis=1)
print("depth %d: queue %d rays, took %d, %.2f start inside the mesh's box" % (depth, live, rays.shape[0], inside.mean()), flush=True)


def per_part(first, second, parts=8):
    a, b = np.flatnonzero(first), np.flatnonzero(second)
    out = []
    for p in range(parts):
        out.append(a[len(a) * p // parts: len(a) * (p + 1) // parts])
        out.append(b[len(b) * p // parts: len(b) * (p + 1) // parts])
    return np.concatenate(out)


def run(label, r):
    t0 = time.perf_counter()
    tri, t = sc.intersect(r, use_kdtree=True)
    print("%-40s hits %d wall %.1f ms (incl. copies)" % (label, int((tri >= 0).sum()), (time.perf_counter() - t0) * 1e3), flush=True)
    return tri


base = run("queue order", rays)
run("queue order (again)", rays)
order = per_part(inside, ~inside)
got = run("inside first in each part", rays[order])
assert np.array_equal(got, base[order])
run("inside last in each part", rays[per_part(~inside, inside)])
run("inside first, whole queue", rays[np.concatenate([np.flatnonzero(inside), np.flatnonzero(~inside)])])
