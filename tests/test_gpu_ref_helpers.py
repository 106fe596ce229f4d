"""GPU: the device's per-path helper functions against outputs of the reference's own text
(tests/golden/ref_helpers.npz; see tests/test_ref_helpers.py for what the file is), through the
C-ABI entry gpt_debug_helpers -- a kernel that evaluates the very inline functions the render
kernels use (pt_device.h) -- bit for bit, and the traversal kernel's own form of the triangle test
(no early return, two-comparison acceptance) against FindTriangleIntersect's hit flag and t."""
import os

import numpy as np
import pytest

import ref_helper_inputs as R
from gpupathtracer_amd import api
from test_ref_helpers import inputs_of, mismatches

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "ref_helpers.npz"))


@pytest.mark.parametrize("name", R.FUNCTIONS)
def test_gpu_functions_equal_reference_text(name, gold):
    words = inputs_of(name, gold)
    out = api.debug_helpers(R.FUNCTIONS.index(name), words, R.LAYOUT[name][1])
    bad = mismatches(name, out, gold["out_" + name], words)
    assert bad.size == 0, (name, bad[:5], out[bad[:5]], gold["out_" + name][bad[:5]])


def test_traversal_kernel_triangle_test_equals_reference_text(gold):
    """k_intersect_kd accepts a triangle iff FindTriangleIntersect reports a hit and 0 < t < bestT (the caller's rule at
    path_tracer_kernel.cu:544; bestT starts at 123456789, :495), and then holds the same t bits.  (Without its early
    return a zero determinant yields infinite or NaN quotients; those never beat a finite bestT.)"""
    words = inputs_of("triangle", gold)
    ref = gold["out_triangle"]
    out = api.debug_helpers(len(R.FUNCTIONS), words, 2)
    ref_t = ref[:, 1].view(np.float32)
    written = ref[:, 1] != 0x7FC0DEAD
    want = (ref[:, 0] == 1) & written & (ref_t > 0) & (ref_t < np.float32(123456789))
    bad = np.flatnonzero((out[:, 0] == 1) != want)
    assert bad.size == 0, (bad[:5], out[bad[:5]], ref[bad[:5]], words[bad[:5]].view(np.float32))
    assert np.array_equal(out[want, 1], ref[want, 1])
    assert want.sum() > 10000
