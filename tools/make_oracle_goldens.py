"""Regenerates tests/golden/oracle_*.npz: outputs of THIS repository's CPU oracle (oracle/pathtracer_oracle.c) on three small
scenes -- per-sample radiance, the branch and primitive of every bounce, live counts.  They do not pin the oracle to the
reference (nothing can: see tests/golden/README.md); they freeze the oracle against accidental change and give the GPU
tests committed vectors to compare the HIP path with, bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle_lib as O  # noqa: E402
import golden_scenes as G  # noqa: E402

for name in G.NAMES:
    sph, polys, cam, depth, counters, stride_mode = G.scene(name)
    out = {}
    for c in counters:
        r = O.render_pass(sph, polys, cam, c, O.params(max_depth=depth, row_stride_is_width=stride_mode), want_events=True)
        out["rgb_%d" % c] = r["rgb"]
        out["events_%d" % c] = r["events"]
        out["prims_%d" % c] = r["prims"]
        out["live_%d" % c] = r["live"]
    path = os.path.join(ROOT, "tests", "golden", "oracle_%s.npz" % name)
    np.savez_compressed(path, **out)
    print(name, os.path.getsize(path), "bytes")
