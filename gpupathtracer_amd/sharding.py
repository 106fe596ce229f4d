"""Samples-per-pixel sharding across the GPUs of one node.

Every sample is a pure function of (pass counter, pixel, depth) through the
seed formulas (path_tracer_kernel.cu:139, :492), so passes partition freely:
rank r renders its own contiguous range of pass counters into its own
device-resident float4 image, and the images are summed once (RCCL all-reduce
over xGMI; gloo on CPU in the tests).  The reference sums passes sequentially
on the host (path_tracer.cpp:100-102); the sharded sum differs from that only
in float summation order.
"""
import torch
import torch.distributed as dist


def pass_range(total_passes, rank, world):
    """Contiguous split of counters [0, total_passes): (begin, end) of `rank`."""
    base, extra = divmod(int(total_passes), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def weak_range(passes_per_rank, rank, offset=0):
    """Weak scaling: every rank renders `passes_per_rank` passes of its own."""
    begin = offset + rank * int(passes_per_rank)
    return begin, begin + int(passes_per_rank)


def sum_images(image, group=None, force=False):
    """In-place sum of the per-rank (r, g, b, passes) images over all ranks.  A group of one rank has
    nothing to add and is skipped unless `force` asks for the collective anyway (to exercise the RCCL
    path on a one-GPU box)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        dist.all_reduce(image, op=dist.ReduceOp.SUM, group=group)
    return image


class DeviceImage:
    """__cuda_array_interface__ view of a scene's accumulation image so torch can reduce it in place."""

    def __init__(self, ptr, n_floats):
        self.__cuda_array_interface__ = {"shape": (int(n_floats),), "typestr": "<f4",
                                         "data": (int(ptr), False), "version": 2}


def scene_image_tensor(scene, device):
    ptr, nbytes = scene.accum_device_ptr()
    return torch.as_tensor(DeviceImage(ptr, nbytes // 4), device=device)
