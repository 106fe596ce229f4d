#!/usr/bin/env python3
"""bench.py -- throughput of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one frame of the workload at the metric's sample count: 1024
sample-per-pixel passes (--spp-per-step; a pass is launchKernel's unit of
work) over all pixels, traced 512 passes per kernel launch.

N > 1: one process per GPU over RCCL.  Started without a launcher (no
WORLD_SIZE in the environment) the script starts its N ranks itself, as a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...`, relays the
child's JSON line and exits with its code; started by a launcher it is one of
the ranks.  A world size other than --gpus is an error.
  --scaling weak   (default) every rank renders its own K steps -- a disjoint
                   range of pass counters, i.e. independently seeded samples;
  --scaling strong the K x 1024 passes of the job are split across the ranks
                   (rank r renders a contiguous 1/N of the pass counters);
either way the float4 (r, g, b, passes) images are sum-reduced once with RCCL at
the end of the timed region.

Workload (the one the headline metric is quoted on): the reference scene as launchKernel would
see it with its mesh path enabled on a heavy cow: the 14 default spheres,
ground and sky (path_tracer.cpp:116-271), plus `cow_heavy` = cow.obj after 4
rounds of 1->4 midpoint subdivision (1 485 824 triangles, all glass as
path_tracer.cpp:275-278 assigns), default orbit camera, 1920x1080, depth 8,
kd-tree traversal.  Inputs (scene, tree, buffers) are resident in HBM before
the timed region starts.

Rank 0 prints one JSON line; `roofline` describes the dominant kernel
(k_intersect_kd, the kd-tree traversal) and `cpu_baseline` the CPU oracle timed on
a bounded sample of the same workload (outside the timed region).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak
T_START = time.perf_counter()
TRIS = [10000000]


def build_workload(name, subdiv):
    from gpupathtracer_amd import api, meshes
    spheres = api.default_scene()
    if name == "cow_heavy":
        v, n = meshes.load_asset("cow")
        v, n = meshes.subdivide(v, n, subdiv)
        polys = meshes.polys_from_triangles(v, n, api.material_preset("glass"))
    elif name == "spheres":
        polys = None
    elif name == "diamond_coil":
        # glass diamond + marble (subsurface) coil moved into view, lens on: long refractive / medium tails
        dv, dn = meshes.load_asset("diamond")
        cv, cn = meshes.load_asset("coil")
        polys = np.concatenate([
            meshes.polys_from_triangles(meshes.transform(dv, 0.8, (0.2, -2.2, 2.6)), dn, api.material_preset("glass")),
            meshes.polys_from_triangles(meshes.transform(cv, 0.25, (-1.2, -2.2, 1.2)), cn, api.material_preset("marble"))])
    elif name == "soup":
        # synthetic mesh: 10 M (or --tris) small triangles uniformly in [-1,1]^3, lifted above the ground
        v, n = meshes.random_soup(TRIS[0], seed=12345, extent=1.0, edge=0.01)
        polys = meshes.polys_from_triangles(meshes.transform(v, 0.75, (0.0, 0.1, 0.0)), n, api.material_preset("white"))
    else:
        raise SystemExit("unknown workload " + name)
    return spheres, polys


def cpu_baseline(spheres, polys, cam, depth, target_seconds=40.0):
    """The oracle (reference algorithm, brute-force triangle loop) on a strided pixel subset."""
    import oracle_lib as O
    w, h = int(cam["resolution"][0]), int(cam["resolution"][1])
    p = O.params(max_depth=depth, row_stride_is_width=1)
    threads = O.num_threads()
    # calibrate on a tiny subset, then size the sample for ~target_seconds
    def subset(stride):
        ys, xs = np.meshgrid(np.arange(stride // 2, h, stride), np.arange(stride // 2, w, stride), indexing="ij")
        return (ys * w + xs).astype(np.int32).ravel()
    ids = subset(120)
    t0 = time.perf_counter()
    r = O.render_pass(spheres, polys, cam, 0, p, pixel_ids=ids)
    dt = time.perf_counter() - t0
    per_px = dt / max(1, ids.size)
    want_px = max(ids.size, int(target_seconds / max(per_px, 1e-9)))
    stride = max(4, int(np.sqrt(w * h / want_px)))
    ids = subset(stride)
    t0 = time.perf_counter()
    r = O.render_pass(spheres, polys, cam, 0, p, pixel_ids=ids)
    dt = time.perf_counter() - t0
    return {
        "value": r["bounces"] / dt / 1e6,
        "unit": "Mpath-bounces/s",
        "cores": threads,
        "kind": "port",
        "sample": "pass counter 0, every %dth pixel in x and y of the %dx%d frame (%d paths, %d path-bounces, "
                  "%.1f s), reference algorithm with its brute-force triangle loop"
                  % (stride, w, h, ids.size, r["bounces"], dt),
    }


def cpu_kd_figure(spheres, polys, cam, depth, kd_small, target_seconds=8.0):
    """Context for the GPU / CPU ratio, NOT the reference's algorithm: the library's own per-path arithmetic and kd
    traversal (pt_device.h / kd_traverse.h compiled for the host by tests/host_harness, one lane at a time, OpenMP over
    pixels) on a strided pixel subset of the same pass.  Returns None when the harness has not been built."""
    import ctypes
    from gpupathtracer_amd.types import CAMERA_DTYPE, POLY_DTYPE, RENDER_PARAMS_DTYPE
    so = os.path.join(ROOT, "tests", "host_harness", "libhh.so")
    if polys is None or not os.path.exists(so):
        return None
    H = ctypes.CDLL(so)
    H.hh_tree_build.restype = ctypes.c_void_p
    H.hh_tree_build.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float]
    H.hh_tree_free.argtypes = [ctypes.c_void_p]
    H.hh_render_pass.restype = ctypes.c_int64
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
    w, h = int(cam["resolution"][0]), int(cam["resolution"][1])
    verts = np.ascontiguousarray(np.concatenate([polys["p0"], polys["p1"], polys["p2"]], axis=1), dtype=np.float32)
    rp = np.zeros(1, RENDER_PARAMS_DTYPE)
    rp["max_depth"], rp["ray_bias"], rp["min_ray_weight"] = depth, 0.0002, 0.00001
    rp["ground_elevation"], rp["ground_enabled"], rp["row_stride_is_width"] = -0.8, 1, 1
    camv = np.array(cam, CAMERA_DTYPE).reshape(1)
    polys_c = np.ascontiguousarray(polys, dtype=POLY_DTYPE)
    tree = H.hh_tree_build(P(verts), verts.shape[0], int(kd_small), ctypes.c_float(3.0))   # the tree of the GPU run
    try:
        def run(stride):
            ys, xs = np.meshgrid(np.arange(stride // 2, h, stride), np.arange(stride // 2, w, stride), indexing="ij")
            ids = (ys * w + xs).astype(np.int32).ravel()
            out = np.zeros((ids.size, 3), np.float32)
            fp, ft = np.zeros(ids.size, np.int32), np.zeros(ids.size, np.float32)
            t0 = time.perf_counter()
            tot = H.hh_render_pass(ctypes.c_void_p(tree), polys_c.size, P(polys_c), spheres.size, P(spheres), P(ids), ids.size, 0,
                                   P(camv), P(rp), P(out), P(fp), P(ft))
            return tot, time.perf_counter() - t0, ids.size
        tot, dt, n = run(16)
        stride = max(1, int(16 * np.sqrt(min(1.0, dt / target_seconds))))
        if stride < 16:
            tot, dt, n = run(stride)
    finally:
        H.hh_tree_free(ctypes.c_void_p(tree))
    return {"value": tot / dt / 1e6, "unit": "Mpath-bounces/s", "cores": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)),
            "kind": "port", "what": "NOT the reference's algorithm: this library's kd traversal and shading arithmetic built for the host "
            "(tests/host_harness), OpenMP over pixels, on the tree of the GPU run (SAH stage from %d triangles)" % int(kd_small),
            "sample": "pass counter 0, every %dth pixel in x and y (%d paths, %d path-bounces, %.1f s)" % (stride, n, tot, dt)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=1024, help="sample-per-pixel passes per step")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--workload", default="cow_heavy")
    ap.add_argument("--subdiv", type=int, default=4)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--tris", type=int, default=10000000, help="triangle count of the soup workload")
    ap.add_argument("--batch", type=int, default=0, help="passes traced together per launch (0: what fits half of the free device memory, at most 512)")
    ap.add_argument("--kd-small", type=int, default=0, help="kd builder: node size at which the SAH stage takes over (0: library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spheres", action="store_true", help="ablation: drop the 14 spheres")
    ap.add_argument("--traffic-file", default=None,
                    help="stored rocprofv3 --pmc counters of this workload (default: profiles/traffic_<workload>.json)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse the launch and the sharding without a GPU: gloo ranks, no rendering (CPU tests)")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args, argv):
    """--gpus N > 1 without a launcher: start the N ranks as a fresh child process tree (nothing in
    this process has touched the GPU), relay the child's output, return its exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONUNBUFFERED"] = "1"
    print("bench.py: starting %d ranks (torch.distributed.run, RCCL)" % args.gpus, file=sys.stderr, flush=True)
    # the ranks' stderr (their progress lines) is inherited and appears as it is written; stdout is read line by
    # line: anything but rank 0's JSON line is passed on at once, so a long run is never silent
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    lines = []
    for ln in proc.stdout:
        ln = ln.rstrip("\n")
        if ln.startswith("{"):
            lines.append(ln)
        else:
            print(ln, file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc == 0:
        if len(lines) != 1:
            print("bench.py: expected one JSON line from rank 0, got %d" % len(lines), file=sys.stderr)
            return 1
        got = json.loads(lines[0])
        if got.get("n_gpus") != args.gpus or (got.get("rccl") or {}).get("world_size") != args.gpus:
            print("bench.py: the ranks report n_gpus=%s, RCCL world size %s, not %d"
                  % (got.get("n_gpus"), (got.get("rccl") or {}).get("world_size"), args.gpus), file=sys.stderr)
            return 1
        print(lines[0], flush=True)
    return rc


def progress(rank, world, what):
    """One line per rank on stderr at the milestones of a run (a multi-GPU run is never silent for long)."""
    print("bench.py: rank %d/%d %s (t = %.1f s)" % (rank, world, what, time.perf_counter() - T_START), file=sys.stderr, flush=True)


def rank_passes(args, rank, world):
    """(warm-up range, timed range) of pass counters of `rank`; ranges of different ranks never overlap."""
    from gpupathtracer_amd import sharding
    Kp, Wp = args.steps * args.spp_per_step, args.warmup * args.spp_per_step
    if args.scaling == "strong":
        # the job's Kp passes are split; warm-up counters lie beyond them
        tb, te = sharding.pass_range(Kp, rank, world)
        wb, we = sharding.pass_range(Wp, rank, world)
        return (Kp + wb, Kp + we), (tb, te)
    base, _ = sharding.weak_range(Kp + Wp, rank)
    return (base, base + Wp), (base + Wp, base + Wp + Kp)


def dry_run(args, rank, world, emit_line):
    """The launch path without a GPU: rendezvous over gloo, the ranks' pass ranges, one sum-reduce."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    (wb, we), (tb, te) = rank_passes(args, rank, world)
    t = torch.tensor([float(te - tb), float(tb), float(te), 1.0], dtype=torch.float64)
    gathered = [torch.zeros_like(t) for _ in range(world)]
    if world > 1:
        dist.all_gather(gathered, t)
    else:
        gathered = [t]
    if rank == 0:
        emit_line(json.dumps({"metric": "dry-run", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "scaling": args.scaling, "passes": [int(g[0]) for g in gathered],
                          "ranges": [[int(g[1]), int(g[2])] for g in gathered],
                          # (the group the collective layer formed: gloo here, RCCL in a real run)
                          "rccl": {"world_size": dist.get_world_size() if world > 1 else 1,
                                   "backend": dist.get_backend() if world > 1 else None}}))
    if world > 1:
        dist.destroy_process_group()


def load_traffic(path, match, kernel):
    """Stored rocprofv3 --pmc figures of `kernel` (tools/collect_profiles.py) -- only when that collection was made on
    the configuration of this run: (HBM-side bytes per path-bounce, counters dict, source note)."""
    if not os.path.exists(path):
        return None, None, "no stored counters (%s)" % os.path.relpath(path, ROOT)
    try:
        tj = json.load(open(path))
    except Exception:
        return None, None, "unreadable " + os.path.basename(path)
    cfg = tj.get("config")
    if cfg != match:
        return None, None, "stored counters are for another configuration (%s)" % json.dumps(cfg)
    k = (tj.get("kernels") or {}).get(kernel)
    if not k or not k.get("hbm_bytes_per_launch") or not k.get("path_bounces_per_launch"):
        return None, None, "incomplete " + os.path.basename(path)
    counters = {n: k[n] for n in ("valu_busy", "valu_lane_utilisation", "l2_hit_rate", "wave_cycles_waiting") if n in k}
    return k["hbm_bytes_per_launch"] / k["path_bounces_per_launch"], counters, \
        "stored: %s (%s), not measured in this run" % (os.path.relpath(path, ROOT), tj.get("source", "rocprofv3 --pmc"))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(spawn_ranks(args, argv))          # before anything imports torch or touches the GPU

    # Only the JSON line goes to stdout: file descriptor 1 is pointed at stderr for the rest of the process (libraries
    # such as RCCL print banners to it from C code), the line itself is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit_line(text):
        os.write(json_fd, (text + "\n").encode())

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if args.dry_run:
        return dry_run(args, rank, world, emit_line)

    import torch
    import torch.distributed as dist
    from gpupathtracer_amd import api, sharding

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("rank %d has no GPU: %d visible" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    # GPT_FORCE_DIST=1 exercises the RCCL path with a single rank (one-GPU boxes)
    distributed = world > 1 or os.environ.get("GPT_FORCE_DIST") == "1"
    if distributed:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    progress(rank, world, "on device %d, RCCL group %s" % (local_rank, "up" if distributed else "not used"))
    TRIS[0] = args.tris
    spheres, polys = build_workload(args.workload, args.subdiv)
    if args.no_spheres:
        spheres = spheres[:0]
    cam = api.default_camera(args.width, args.height)
    n_pix = args.width * args.height
    K, W = args.steps, args.warmup

    params = api.render_params(max_depth=args.depth, row_stride_is_width=1, use_kdtree=1,
                               passes_in_flight=args.batch, **({"kd_small_threshold": args.kd_small} if args.kd_small > 0 else {}))
    t_build = time.perf_counter()
    scene = api.Scene(spheres, polys, params, device=local_rank)
    t_build = time.perf_counter() - t_build
    scene.reset(n_pix)

    def barrier():
        if distributed:
            dist.barrier()
        scene.sync()
        torch.cuda.synchronize()

    (wb, we), (tb, te) = rank_passes(args, rank, world)
    my_passes = te - tb

    # ---- untimed: kd work per path-bounce (instrumented kernel), then warm-up ----
    scene.set_profiling(time_kernels=False, count_kd_work=True)
    scene.render_range(cam, tb, tb + 1)
    st = scene.stats()
    pb1 = max(1, st["path_bounces"])
    nodes_per_pb = st["kd_nodes_visited"] / pb1
    tris_per_pb = st["kd_tris_tested"] / pb1
    hit_frac = st["hits"] / pb1
    scene.set_profiling(time_kernels=False, count_kd_work=False)
    progress(rank, world, "scene ready (%.1f s), warm-up of %d passes starts" % (t_build, we - wb))
    scene.render_range(cam, wb, we)
    scene.reset(n_pix)
    progress(rank, world, "warm-up done, timing passes %d..%d" % (tb, te))
    scene.set_profiling(time_kernels=True, count_kd_work=False)

    image = sharding.scene_image_tensor(scene, torch.device("cuda", local_rank))

    # ---- timed region: exactly K steps + the sum-reduce of the images ----
    barrier()
    t0 = time.perf_counter()
    scene.render_range(cam, tb, te, sync=True)
    sharding.sum_images(image, force=distributed)
    barrier()
    elapsed = time.perf_counter() - t0

    progress(rank, world, "timed region done: %.2f s" % elapsed)
    st = scene.stats()
    t = torch.tensor([elapsed, float(st["path_bounces"]), float(my_passes)], dtype=torch.float64, device="cuda")
    if distributed:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tmin = t.clone()
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed_max, total_pb, total_passes = float(tmax[0]), float(tsum[1]), int(round(float(tsum[2])))
        elapsed_min = float(tmin[0])
    else:
        elapsed_max, total_pb, total_passes = elapsed, float(st["path_bounces"]), my_passes
        elapsed_min = elapsed
    # every pass of the job is in the reduced image exactly once (its 4th channel counts them)
    passes_in_image = int(round(float(image[3].item())))
    if passes_in_image != total_passes:
        raise SystemExit("reduced image holds %d passes, the ranks rendered %d" % (passes_in_image, total_passes))

    if rank == 0:
        n_spheres = int(spheres.size)
        # Algorithmic bytes per path-bounce: B = B_state + B_geom, with
        #   B_state = 124 (path state read + written once), B_geom = 8 N_nodes + 40 N_tris + 16 N_spheres + 72 [hit].
        # The terms the kd intersect kernel itself executes are the ray it reads (24), the hit slot it
        # reads and may rewrite (8) and the tree terms; ground / sphere tests, state and materials
        # belong to k_camera / k_shade.  `achieved` prices the dominant kernel with ITS terms;
        # `pass_level` prices the whole path-bounce against the whole device time.
        bytes_per_pb = 124.0 + 8.0 * nodes_per_pb + 40.0 * tris_per_pb + 16.0 * n_spheres + 72.0 * hit_frac
        kd_bytes_per_pb = 24.0 + 8.0 + 8.0 * nodes_per_pb + 40.0 * tris_per_pb
        launches = max(1, st["trace_launches"])
        pb_per_launch = st["path_bounces"] / launches
        trace_share = st["device_ms_trace"] / max(1e-9, st["device_ms_total"])
        info = scene.info()
        batch = info.pop("passes_in_flight")
        if trace_share >= 0.5:
            kernel, kernel_bytes = "k_intersect_kd", kd_bytes_per_pb
            avg_launch_s = st["device_ms_trace"] / 1e3 / launches
        else:   # no (or cheap) triangle work: the shading kernel dominates; it moves everything but the tree terms
            # (the spheres' 16 bytes each are read by every path-bounce but live in the caches: left out here)
            kernel, kernel_bytes = "k_shade", 124.0 + 72.0 * hit_frac
            avg_launch_s = (st["device_ms_total"] - st["device_ms_trace"]) / 1e3 / launches
        achieved = kernel_bytes * pb_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        pass_level = bytes_per_pb * st["path_bounces"] / (st["device_ms_total"] / 1e3) / 1e9 if st["device_ms_total"] > 0 else 0.0
        traffic, traffic_source = None, "not collected for this kernel"
        # what a stored counter collection must have been made on to speak for this run
        counter_config = {"workload": args.workload, "subdiv": args.subdiv if args.workload == "cow_heavy" else None,
                          "tris": 0 if polys is None else int(polys.size), "width": args.width, "height": args.height,
                          "depth": args.depth, "passes_per_launch": batch, "kd": info}
        traffic_file = args.traffic_file or os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        per_pb, counters, traffic_source = load_traffic(traffic_file, counter_config, kernel)
        if per_pb is not None:
            traffic = per_pb * pb_per_launch
        # A workload whose algorithmic bytes exceed what HBM could deliver in the measured time is served
        # from the caches: its bound is the vector ALU, and an HBM fraction would say nothing.
        cache_resident = achieved > HBM_PEAK_GBS
        roofline = {
            "kernel": kernel,
            "bound": "valu" if cache_resident else "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": None if cache_resident else achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "counter_config": counter_config,
            # measured HBM-side bytes over the kernel's time over the peak (needs stored counters of this configuration)
            "hbm_frac": (traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS) if (traffic and avg_launch_s > 0) else None,
            # the kernel's stored SQ / TCC figures: VALU issue-slot occupancy, lane activity, L2 hit rate, waiting share
            "counters": counters,
            "algorithmic_gbps": achieved,
            "bytes_per_path_bounce": kernel_bytes,
            "pass_level": {"bytes_per_path_bounce": bytes_per_pb, "algorithmic_gbps": pass_level,
                           "note": "whole path-bounce (all kernels, sphere reads included) over whole device time; not an HBM fraction"},
            "kd_nodes_per_path_bounce": nodes_per_pb,
            "tris_per_path_bounce": tris_per_pb,
            "hit_fraction": hit_frac,
            "avg_launch_ms": avg_launch_s * 1e3,
            "launches": launches,
            "path_bounces_per_launch": pb_per_launch,
            "intersect_share_of_pass": trace_share,
        }
        if cache_resident:
            roofline["note"] = ("algorithmic bytes exceed what HBM could deliver in that time: the scene is cache resident and "
                                "the kernel VALU-bound; its VALU occupancy and lane activity are in `counters`")
        out = {
            "metric": "Mpath-bounces/s",
            "value": total_pb / elapsed_max / 1e6,
            "unit": "Mpath-bounces/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed_max / K * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "ms_per_pass": elapsed_max / max(1, my_passes) * 1e3,
            "spp_per_s": total_passes / elapsed_max,
            "timed_region_s": elapsed_max,
            "config": {
                "workload": "%s: %s (%d triangles) + the reference's 14 spheres, "
                            "ground and sky; default orbit camera %dx%d, depth %d, %d spp per step, kd-tree"
                            % (args.workload,
                               {"cow_heavy": "cow.obj subdivided %dx, glass" % args.subdiv,
                                "spheres": "no mesh", "diamond_coil": "glass diamond + marble coil",
                                "soup": "uniform random small white triangles"}[args.workload],
                               0 if polys is None else polys.size, args.width, args.height, args.depth, args.spp_per_step),
                "passes_total": total_passes,
                "passes_per_gpu": my_passes,
                "passes_per_launch": batch,
                "parallelism": "spp-sharded x%d (%s), one RCCL sum-reduce of the float4 image" % (world, args.scaling),
                "bounces_per_path": st["path_bounces"] / max(1, my_passes * n_pix),
                "kd": dict(info, build_and_upload_s=round(t_build, 2)),
            },
            "roofline": roofline,
            # what the collective layer itself saw: the group RCCL formed, and the one sum-reduce of the float4 images
            "rccl": {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                     "reduced_bytes": int(image.numel() * image.element_size()), "collectives_in_timed_region": 1} if distributed
                    else {"world_size": 1, "backend": None, "reduced_bytes": 0, "collectives_in_timed_region": 0},
            # load balance: the timed region of the fastest and the slowest rank (value uses the slowest)
            "rank_elapsed_s": {"min": elapsed_min, "max": elapsed_max},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(spheres, polys if polys is not None else None, cam, args.depth)
            kd_fig = cpu_kd_figure(spheres, polys, cam, args.depth, scene.kd_small_threshold())
            if kd_fig is not None:
                out["cpu_kd_traversal"] = kd_fig
        emit_line(json.dumps(out))
    scene.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
