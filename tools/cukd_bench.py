"""The reference's kd-tree benchmark (cukd/test/ray_traversal.cu:116-189: build the tree once and then 10 timed times,
traverse 1024 x 864 parallel rays 100 times, print the averages) on this library: the dragon meshes that ship with it,
its ray set (tests/cukd_fixture.py), the host and the GPU builder, the production traversal kernel through
gpt_debug_intersect.  Kernel times of the traversal come from a rocprofv3 --kernel-trace run around this script
(tools/cukd_bench_run.sh); the wall time printed here includes the host-side packing and copies of every call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gpupathtracer_amd import api, meshes
import cukd_fixture as F

name = sys.argv[1] if len(sys.argv) > 1 else "dragon_res3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
v, n = meshes.load_asset(name)
print("Creating k-d tree for %d triangles" % v.shape[0], flush=True)
for on_gpu in (False, True):
    api.kdtree_build(v, on_gpu=on_gpu)                      # (the benchmark's untimed first build)
    t0 = time.perf_counter()
    for _ in range(10):
        tree = api.kdtree_build(v, on_gpu=on_gpu)
    dt = (time.perf_counter() - t0) / 10
    print("tree creation time average (%s builder): %.2f ms  (%d nodes, %d references, depth %d)"
          % ("GPU" if on_gpu else "host", dt * 1e3, tree["nodes"].shape[0], tree["refs"].size, tree["max_depth"]), flush=True)
rays = F.rays()
polys = meshes.polys_from_triangles(v, n, api.material_preset("white"))
with api.Scene(None, polys, api.render_params(max_depth=1)) as sc:
    tri, t = sc.intersect(rays, use_kdtree=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        sc.intersect(rays, use_kdtree=True)
    dt = (time.perf_counter() - t0) / reps
print("%d rays, %d hit; %d repetitions: %.2f ms per call incl. host packing and copies" % (len(rays), int((tri >= 0).sum()), reps, dt * 1e3), flush=True)
