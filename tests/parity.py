"""Shared checks of the GPU parity tests: one pass of the HIP path against the oracle.

The stated bar for radiance is 1e-4 relative per sample.  The HIP path meets a stricter one: every
expression is evaluated in the type and order a host build of the reference gives it, without fused
multiply-add, with IEEE division and square root, and sin / cos / log / pow restate the host C
library's algorithms bit for bit (pt_device.h) -- so every sample of every pass must equal the oracle's
IN EVERY BIT, and so must the record of what every bounce of every path did (branch taken, primitive
hit).  The checks below assert exactly that; when they fail they say how far off the result is in
the terms of the stated bar (paths that took another branch, worst relative error on identical paths)."""
import numpy as np

REL_TOL = 1e-4          # the stated bar; the assertion itself is equality


def rel_err(got, ref):
    return (np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)).max(axis=1)


def same_paths(ev_got, prim_got, ev_ref, prim_ref):
    """Per sample: identical branch codes and primitives at every depth."""
    return (ev_got == ev_ref).all(axis=1) & (prim_got == prim_ref).all(axis=1)


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def check_pass(got, ev_got, prim_got, ref, ids=None):
    """`ref`: oracle render_pass(..., want_events=True) result; `ids`: the pixel subset it covers."""
    if ids is not None:
        got, ev_got, prim_got = got[ids], ev_got[ids], prim_got[ids]
    same = same_paths(ev_got, prim_got, ref["events"], ref["prims"])
    r = rel_err(got, ref["rgb"])
    report = {"flips": float((~same).mean()), "worst_same": float(r[same].max()) if same.any() else 0.0,
              "beyond_tol": float((r > REL_TOL).mean()),
              "samples_differing_in_any_bit": int((got.view(np.uint32) != ref["rgb"].view(np.uint32)).any(axis=1).sum())}
    assert same.all(), "paths that took another branch or hit another primitive: %s" % report
    assert bits_equal(got, ref["rgb"]), "radiance differs from the oracle's: %s" % report
    return report
