"""GPU: the multi-GPU exchange step on the one GPU a test box has -- a communicator of one rank.
The reduce of a single image must leave it bit for bit as rendered, through both routes:
RCCL under torch.distributed (what bench.py uses) and RCCL called by the library itself
(gpt_accum_allreduce, what `gpt_render --gpus N` uses)."""
import os
import socket
import subprocess

import numpy as np
import pytest

import scenes as S
from gpupathtracer_amd import api, meshes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene():
    v, n = meshes.load_asset("monkey")
    v = meshes.transform(v, 0.5, (0.0, -0.2, 1.5))
    return S.default_spheres(), meshes.polys_from_triangles(v, n, S.MATERIALS["glass"])


def test_one_rank_nccl_sum_through_sharding_leaves_the_image_unchanged():
    import torch
    import torch.distributed as dist
    from gpupathtracer_amd import sharding
    w, h = 320, 180
    cam = S.orbit_camera(w, h)
    sph, polys = _scene()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        with api.Scene(sph, polys, api.render_params(max_depth=8)) as sc:
            sc.reset(w * h)
            sc.render_range(cam, 0, 6)
            want, passes = sc.accum_download()
            image = sharding.scene_image_tensor(sc, torch.device("cuda", 0))
            assert image.numel() == w * h * 4
            sharding.sum_images(image, force=True)            # RCCL all-reduce over a group of one
            torch.cuda.synchronize()
            got, passes2 = sc.accum_download()
        assert passes == passes2 == 6 and np.array_equal(got, want)
    finally:
        dist.destroy_process_group()


def test_library_rccl_allreduce_single_scene():
    w, h = 320, 180
    cam = S.orbit_camera(w, h)
    sph, polys = _scene()
    with api.Scene(sph, polys, api.render_params(max_depth=8)) as sc:
        sc.reset(w * h)
        sc.render_range(cam, 0, 6, sync=False)                # the reduce queues behind the rendering
        api.accum_allreduce([sc])
        got, passes = sc.accum_download()
        sc.reset(w * h)
        sc.render_range(cam, 0, 6)
        want, _ = sc.accum_download()
        assert passes == 6 and np.array_equal(got, want)
        # gpt_shutdown destroys the cached communicator; the next reduce sets one up again
        api.shutdown()
        api.accum_allreduce([sc])
        again, passes = sc.accum_download()
        assert passes == 6 and np.array_equal(again, want)
        with api.Scene(sph, None, api.render_params(max_depth=2)) as other:
            other.reset(w * h)
            with pytest.raises(api.GptError):
                api.accum_allreduce([sc, other])              # two scenes on one device


def test_gpt_render_gpus_flag(tmp_path):
    """`gpt_render --gpus 1` goes through the threaded, RCCL-reduced route and writes the same image."""
    exe = os.path.join(ROOT, "gpupathtracer_amd", "lib", "gpt_render")
    outs = []
    for extra in ([], ["--gpus", "1"]):
        ppm = str(tmp_path / ("o%d.ppm" % len(outs)))
        out = subprocess.run([exe, "--width", "96", "--height", "64", "--spp", "6", "--depth", "8", "--out", ppm] + extra,
                             capture_output=True, text=True, timeout=180)
        assert out.returncode == 0, out.stderr
        assert "Iterations: 6" in out.stdout
        outs.append(open(ppm, "rb").read())
    assert outs[0] == outs[1]
