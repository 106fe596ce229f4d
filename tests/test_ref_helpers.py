"""CPU: the per-path helper functions against outputs of the REFERENCE'S OWN TEXT.

tests/golden/ref_helpers.npz holds, for 102 400 inputs per function, what the reference's
path_tracer_kernel.cu lines 75-84 and 194-475 return -- randhash, positionAlongRay,
findGroundPlaneIntersection, FindTriangleIntersect, randomCosineWeightedDirectionInHemisphere,
randomDirectionInSphere, computeReflectionDirection, computeTransmissionDirection, computeFresnel,
computeBackgroundColor, computeTransmission -- compiled unmodified with g++ against the genuine CUDA
toolkit headers of the image's Triton wheel and the reference's own cutil_math.h
(oracle/ref_build/helpers_dump.cpp; produced by tools/make_goldens.py, container only).
findSphereIntersection is not in the file: under a host compiler its min / max calls bind to
cutil_math.h's host int min(int, int) and truncate.

The oracle's restatements and the library's device functions (pt_device.h, here compiled for the host)
must reproduce every output word; tests/test_gpu_ref_helpers.py repeats the comparison on the GPU.
Two NaNs count as equal (their payloads are not defined by the source)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import ref_helper_inputs as R

HERE = os.path.dirname(os.path.abspath(__file__))
UNSET = 0xFFFFFFFF


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "ref_helpers.npz"))


def inputs_of(name, gold):
    words = R.make(name)
    assert R.digest(words) == str(gold["sha_" + name]), "inputs differ from those the reference was run on"
    return words


def mismatches(name, got, ref, words):
    """Rows whose outputs differ from the reference's in any word the implementation computes."""
    got, ref = np.asarray(got, np.uint32), np.asarray(ref, np.uint32)
    same = (got == ref) | (got == UNSET)
    if name != "randhash":
        gf, rf = got.view(np.float32), ref.view(np.float32)
        same |= (gf != gf) & (rf != rf)
    if name == "triangle":
        # the reference returns before writing t and the barycentrics when the determinant is 0
        unwritten = ref[:, 1] == 0x7FC0DEAD
        same[unwritten, 1:] = True
    bad = np.flatnonzero(~same.all(axis=1))
    return bad


def test_fixture_covers_the_branches(gold):
    tri = gold["out_triangle"]
    hit = tri[:, 0] == 1
    assert 0.15 < hit.mean() < 0.8 and (tri[:, 1] == 0x7FC0DEAD).sum() > 1000          # hits, misses, zero determinants
    tr = gold["out_transmit"].view(np.float32)
    tir = (tr == 0).all(axis=1)
    assert 0.02 < tir.mean() < 0.5                                                      # total internal reflection
    fr = gold["out_surface"].view(np.float32)[:, 6]
    assert (fr == 1).sum() > 1000 and ((fr > 0) & (fr < 1)).sum() > 50000
    ff = gold["out_fresnel"].view(np.float32)[:, 0]
    assert (ff == 1).sum() > 10000 and ((ff >= 0) & (ff < 1)).sum() > 10000
    g = gold["out_ground"].view(np.float32)
    assert (g[:, 0] == -1).sum() > 1000 and (g[:, 0] > 0).sum() > 10000
    assert int(gold["out_randhash"][0, 0]) == 1800329511 and int(gold["out_randhash"][1, 0]) == 3028713910
    tm = gold["out_transmission"].view(np.float32)
    assert (tm == 0).any() and (tm == 1).any() and np.isnan(tm).any()                   # underflow, sigma 0, 0 * inf


@pytest.mark.parametrize("name", R.FUNCTIONS)
def test_oracle_equals_reference_text(name, gold):
    words = inputs_of(name, gold)
    L = O.lib()
    out = np.zeros((len(words), R.LAYOUT[name][1]), np.uint32)
    assert L.orc_helper_eval(name.encode(), len(words), words.ctypes.data_as(ctypes.c_void_p),
                             out.ctypes.data_as(ctypes.c_void_p)) == 0
    bad = mismatches(name, out, gold["out_" + name], words)
    assert bad.size == 0, (name, bad[:5], out[bad[:5]], gold["out_" + name][bad[:5]])
    assert not (out == UNSET).all(axis=0).any()          # the oracle computes every output word


@pytest.mark.parametrize("name", R.FUNCTIONS)
def test_device_functions_host_build_equal_reference_text(name, gold):
    so = os.path.join(HERE, "host_harness", "libhh.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "host_harness")])
    L = ctypes.CDLL(so)
    words = inputs_of(name, gold)
    out = np.zeros((len(words), R.LAYOUT[name][1]), np.uint32)
    assert L.hh_helper_eval(R.FUNCTIONS.index(name), len(words), words.ctypes.data_as(ctypes.c_void_p),
                            out.ctypes.data_as(ctypes.c_void_p)) == 0
    bad = mismatches(name, out, gold["out_" + name], words)
    assert bad.size == 0, (name, bad[:5], out[bad[:5]], gold["out_" + name][bad[:5]])
    assert (out != UNSET).any(axis=0).sum() >= {"ground": 1, "triangle": 2, "fresnel": 1, "surface": 7}.get(name, R.LAYOUT[name][1])
