"""Per-depth picture of the bench workload: live paths, kd nodes / triangles per path, and
(with rocprofv3's kernel trace around it) per-depth kernel times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gpupathtracer_amd import api
import bench
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
spheres, polys = bench.build_workload("cow_heavy", int(os.environ.get("SUBDIV", "4")))
cam = api.default_camera(1920, 1080)
sc = api.Scene(spheres, polys, api.render_params(max_depth=8, passes_in_flight=batch))
sc.reset(1920 * 1080)
sc.set_profiling(False, True)
sc.render_range(cam, 0, batch)
st = sc.stats()
print("info", sc.info())
for d in range(8):
    live = st["live_at_depth"][d]
    print("depth %d live %9d nodes/path %6.1f tris/path %6.1f" % (d, live, st["kd_nodes_at_depth"][d] / max(1, live), st["kd_tris_at_depth"][d] / max(1, live)))
print("node loop: lane visits %d, wave iterations %d -> lanes/iteration %.1f of 64" % (st["kd_nodes_visited"], st["wave_iters_node"], st["kd_nodes_visited"] / max(1, st["wave_iters_node"])))
print("tri loop: lane tests %d, wave iterations %d -> lanes/iteration %.1f of 64" % (st["kd_tris_tested"], st["wave_iters_tri"], st["kd_tris_tested"] / max(1, st["wave_iters_tri"])))
print("advance rounds: lane-level %d, wave-level %d -> busy lanes per round %.1f" % (st["busy_lane_rounds"], st["wave_rounds"], st["busy_lane_rounds"] / max(1, st["wave_rounds"])))
sc.set_profiling(True, False)
sc.reset(1920 * 1080)
for rep in range(2):
    t0 = time.perf_counter(); sc.render_range(cam, 0, batch); dt = time.perf_counter() - t0
    print("batch of %d: %.2f ms/pass" % (batch, dt / batch * 1e3))
