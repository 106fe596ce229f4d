"""Regenerates gpupathtracer_amd/assets/*.npz from the reference's OBJ files
(container only: needs /root/reference).  Each asset holds what the library's
OBJ loader (gpt_obj_load) produces for that file: verts[n,9] = Poly.p0/p1/p2
and normals[n,3] = Poly.n, float32 -- mesh DATA, not source."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpupathtracer_amd import api  # noqa: E402

REF = "/root/reference/GPUPathTracer"
OUT = os.path.join(ROOT, "gpupathtracer_amd", "assets")
os.makedirs(OUT, exist_ok=True)
for name in ("box", "box2", "cone", "diamond", "coil", "monkey", "cow"):
    polys, bbox = api.obj_load(os.path.join(REF, name + ".obj"))
    verts = np.concatenate([polys["p0"], polys["p1"], polys["p2"]], axis=1).astype(np.float32)
    normals = polys["n"].astype(np.float32)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), verts=verts, normals=normals, bbox=bbox)
    print(name, verts.shape[0], bbox)
