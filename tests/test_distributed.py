"""CPU, world_size 2 over gloo: the multi-GPU path of bench.py -- disjoint pass ranges per
rank, one sum-reduce of the (r,g,b,passes) images -- with the per-rank renders supplied by
the oracle (the HIP library needs a GPU)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpupathtracer_amd import sharding

HERE = os.path.dirname(os.path.abspath(__file__))


def test_pass_ranges_partition_the_counters():
    for total in (0, 1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            covered = []
            for r in range(world):
                b, e = sharding.pass_range(total, r, world)
                assert 0 <= b <= e <= total
                covered += list(range(b, e))
            assert covered == list(range(total))
            sizes = [sharding.pass_range(total, r, world) for r in range(world)]
            assert max(e - b for b, e in sizes) - min(e - b for b, e in sizes) <= 1
    assert sharding.weak_range(10, 3) == (30, 40)
    assert sharding.weak_range(10, 0, offset=5) == (5, 15)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_passes, w, h, depth, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import oracle_lib as O
    import scenes as S
    O.set_num_threads(2)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sph = S.default_spheres()
        cam = S.orbit_camera(w, h)
        p = O.params(max_depth=depth, row_stride_is_width=1)
        begin, end = sharding.pass_range(total_passes, rank, world)
        img = np.zeros((w * h, 4), dtype=np.float32)          # (r, g, b, passes) like the device image
        for c in range(begin, end):
            img[:, :3] += O.render_pass(sph, None, cam, c, p)["rgb"]
            img[:, 3] += 1
        t = torch.from_numpy(img.reshape(-1))
        sharding.sum_images(t)
        if rank == 0:
            np.save(os.path.join(out_dir, "sum.npy"), t.numpy().reshape(-1, 4))
    finally:
        dist.destroy_process_group()


def test_two_rank_sum_equals_single_process_sum(tmp_path):
    import oracle_lib as O
    import scenes as S
    world, total, w, h, depth = 2, 5, 48, 32, 6
    mp.spawn(_worker, args=(world, _free_port(), total, w, h, depth, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "sum.npy"))
    sph = S.default_spheres()
    cam = S.orbit_camera(w, h)
    p = O.params(max_depth=depth, row_stride_is_width=1)
    want = np.zeros((w * h, 3), dtype=np.float32)
    for c in range(total):
        want += O.render_pass(sph, None, cam, c, p)["rgb"]
    assert np.all(got[:, 3] == total)                               # every pass counted exactly once
    # same samples, different float summation order (2+3 passes vs 5 in sequence)
    assert np.allclose(got[:, :3], want, rtol=1e-6, atol=1e-7)


def test_sum_images_is_a_no_op_without_a_process_group():
    t = torch.arange(8, dtype=torch.float32)
    assert torch.equal(sharding.sum_images(t.clone()), t)


def _run_bench(args, env_extra=None):
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True,
                         text=True, timeout=300)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out.returncode, [json.loads(ln) for ln in lines], out.stderr


def test_bench_starts_its_own_ranks():
    """`bench.py --gpus 2` without a launcher starts two ranks itself (a fresh child under
    torch.distributed.run); the dry run rehearses rendezvous and sharding over gloo without a GPU."""
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--spp-per-step", "7", "--dry-run"])
    assert rc == 0, err
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["scaling"] == "weak"
    (b0, e0), (b1, e1) = lines[0]["ranges"]
    assert e0 - b0 == e1 - b1 == 21 and (e0 <= b1 or e1 <= b0)       # weak: 3 steps x 7 passes each, disjoint counters
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--spp-per-step", "7", "--dry-run",
                                 "--scaling", "strong"])
    assert rc == 0, err
    assert lines[0]["ranges"] == [[0, 11], [11, 21]] and lines[0]["scaling"] == "strong"   # 21 passes split


def test_bench_rejects_a_world_size_other_than_gpus():
    rc, lines, err = _run_bench(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and not lines and "--gpus 2" in err
    rc, lines, err = _run_bench(["--gpus", "1", "--dry-run"])
    assert rc == 0 and lines[0]["n_gpus"] == 1
