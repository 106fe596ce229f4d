"""gpupathtracer_amd -- MI355X-native implementation of GPUPathTracer's
path_tracer_kernel hot path (launchKernel), behind a C-ABI (include/gpt.h).

The package is a thin host-side mirror of that C-ABI; all rendering runs in
libgpt_hip.so (hand-written gfx950 HIP kernels).  Nothing here falls back to a
CPU implementation.
"""
from . import types  # noqa: F401

__all__ = ["types", "api", "meshes"]
