"""GPU experiment: how much faster would the kd traversal kernel run on a deep-bounce queue whose rays were sorted
by the cell of their origin?  Takes the real queue entering depth D of a cow_heavy batch (first M rays, queue order),
and traces it with gpt_debug_intersect (the production kernel) as is, sorted by Morton code of the origin (grid of
2^B cells per axis), and shuffled.  Kernel times come from a rocprofv3 --kernel-trace run around this script."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gpupathtracer_amd import api
import bench

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4000000
P = int(sys.argv[3]) if len(sys.argv) > 3 else 64
spheres, polys = bench.build_workload("cow_heavy", 4)
cam = api.default_camera(1920, 1080)
sc = api.Scene(spheres, polys, api.render_params(max_depth=8, passes_in_flight=P))
sc.reset(1920 * 1080)
rays, live = sc.dump_queue(cam, 0, P, depth, M)
print("depth %d: queue %d rays, took %d" % (depth, live, rays.shape[0]), flush=True)


def part1by2(x):
    x = x.astype(np.uint64) & 0x3ff
    x = (x | (x << 16)) & 0x30000ff
    x = (x | (x << 8)) & 0x300f00f
    x = (x | (x << 4)) & 0x30c30c3
    x = (x | (x << 2)) & 0x9249249
    return x


def morton(o, bits):
    lo, hi = o.min(axis=0), o.max(axis=0)
    q = np.clip(((o - lo) / np.maximum(hi - lo, 1e-9) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    return part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)


def run(label, r):
    t0 = time.perf_counter()
    tri, t = sc.intersect(r, use_kdtree=True)
    dt = time.perf_counter() - t0
    print("%-28s hits %d wall %.1f ms (incl. copies)" % (label, int((tri >= 0).sum()), dt * 1e3), flush=True)
    return tri, t


base = run("queue order", rays)
for bits in (6,):
    order = np.argsort(morton(rays[:, :3], bits), kind="stable")
    got = run("sorted by origin cell 2^%d" % bits, rays[order])
    assert np.array_equal(got[0], base[0][order])
# origin cell, then direction octant
octant = (rays[:, 3] > 0).astype(np.uint64) | ((rays[:, 4] > 0).astype(np.uint64) << 1) | ((rays[:, 5] > 0).astype(np.uint64) << 2)
order = np.argsort((morton(rays[:, :3], 6) << np.uint64(3)) | octant, kind="stable")
run("origin cell 2^6 + octant", rays[order])
rng = np.random.default_rng(1)
run("shuffled", rays[rng.permutation(rays.shape[0])])
