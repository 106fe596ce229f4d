"""Times the reference's entry point launchKernel (device-resident Poly / Sphere arrays, one pass per call,
host pixel buffer) on the bench workload: a run of consecutive counters with an unchanged camera."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gpupathtracer_amd import api
import bench
w, h = 1920, 1080
spheres, polys = bench.build_workload("cow_heavy", int(os.environ.get("SUBDIV", "4")))
cam = api.default_camera(w, h)
d_s = api.device_alloc(spheres.nbytes); api.device_upload(d_s, spheres)
d_p = api.device_alloc(polys.nbytes); api.device_upload(d_p, polys)
lookahead = int(sys.argv[1]) if len(sys.argv) > 1 else 0
api.launch_kernel_params(api.render_params(max_depth=8, passes_in_flight=lookahead))
t0 = time.perf_counter()
api.launch_kernel(polys.size, d_p, spheres.size, d_s, w * h, 0, cam)
print("first call (scene upload + kd build): %.2f s" % (time.perf_counter() - t0))
n = 192
t0 = time.perf_counter()
for c in range(1, 1 + n):
    api.launch_kernel(polys.size, d_p, spheres.size, d_s, w * h, c, cam)
dt = time.perf_counter() - t0
print("passes_in_flight %d: %d calls, %.2f ms per call (incl. the 24.9 MB device-to-host copy)" % (lookahead, n, dt / n * 1e3))
times = []
for c in range(1 + n, 1 + 2 * n):
    t1 = time.perf_counter()
    api.launch_kernel(polys.size, d_p, spheres.size, d_s, w * h, c, cam)
    times.append((time.perf_counter() - t1) * 1e3)
times = np.array(times)
print("per call: mean %.2f ms, median %.2f, 95th percentile %.2f, max %.2f" % (times.mean(), np.median(times), np.percentile(times, 95), times.max()))
# the same with the caller's promise that the arrays do not change (no per-call checksum of the 178 MB of polys)
api.launch_kernel_static_scene(1)
times = []
for c in range(1 + 2 * n, 1 + 3 * n):
    t1 = time.perf_counter()
    api.launch_kernel(polys.size, d_p, spheres.size, d_s, w * h, c, cam)
    times.append((time.perf_counter() - t1) * 1e3)
times = np.array(times)
print("declared static: mean %.2f ms, median %.2f, 95th percentile %.2f, max %.2f" % (times.mean(), np.median(times), np.percentile(times, 95), times.max()))
api.launch_kernel_static_scene(0)
api.launch_kernel_params(None)
