"""Experiment: do two half-size batches on two streams (two Scene objects here) fill each other's
launch tails?  Compares one scene tracing 512 passes per launch with two scenes tracing 256 each,
enqueued alternately and running concurrently."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpupathtracer_amd import api
import bench
w, h = 1920, 1080
spheres, polys = bench.build_workload("cow_heavy", 4)
cam = api.default_camera(w, h)
N = 2048

def run_single(batch):
    sc = api.Scene(spheres, polys, api.render_params(max_depth=8, passes_in_flight=batch))
    sc.reset(w * h)
    sc.render_range(cam, 0, batch)          # warm-up
    t0 = time.perf_counter()
    sc.render_range(cam, 0, N)
    dt = time.perf_counter() - t0
    sc.close() if hasattr(sc, "close") else None
    return dt

def run_two(batch):
    a = api.Scene(spheres, polys, api.render_params(max_depth=8, passes_in_flight=batch))
    b = api.Scene(spheres, polys, api.render_params(max_depth=8, passes_in_flight=batch))
    for s in (a, b):
        s.reset(w * h)
        s.render_range(cam, 0, batch)
    t0 = time.perf_counter()
    for c in range(0, N, 2 * batch):
        a.render_range(cam, c, c + batch, sync=False)
        b.render_range(cam, c + batch, c + 2 * batch, sync=False)
    a.sync(); b.sync()
    return time.perf_counter() - t0

for label, fn, arg in (("one scene, 512 per launch", run_single, 512), ("two scenes, 256 per launch each", run_two, 256),
                       ("one scene, 256 per launch", run_single, 256)):
    dt = fn(arg)
    print("%-34s %.3f ms per pass" % (label, dt / N * 1e3), flush=True)
