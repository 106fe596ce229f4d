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


def load_ply_ascii(path):
    """The PLY subset of the reference's kd-tree benchmark fixture (cukd/test/ray_traversal.cu:11-68 reads the same files):
    ascii, `element vertex N` with x y z first, `element face M` as `3 i j k` lists."""
    with open(path) as f:
        n_v = n_f = 0
        for line in f:
            t = line.split()
            if t[:2] == ["element", "vertex"]: n_v = int(t[2])
            if t[:2] == ["element", "face"]: n_f = int(t[2])
            if t and t[0] == "end_header": break
        v = np.array([f.readline().split()[:3] for _ in range(n_v)], dtype=np.float32)
        idx = np.array([f.readline().split()[1:4] for _ in range(n_f)], dtype=np.int64)
    return v, idx


# the Stanford dragon decimations shipped with the reference's kd-tree library as its benchmark input
for name in ("dragon_vrip_res4", "dragon_vrip_res3"):
    v, idx = load_ply_ascii(os.path.join(REF, "GPUPathTracer", "cukd", "test", "dragon", name + ".ply"))
    verts = np.concatenate([v[idx[:, 0]], v[idx[:, 1]], v[idx[:, 2]]], axis=1).astype(np.float32)
    e1, e2 = verts[:, 3:6] - verts[:, 0:3], verts[:, 6:9] - verts[:, 0:3]
    nrm = np.cross(e1, e2)
    ln = np.linalg.norm(nrm, axis=1, keepdims=True)
    normals = np.where(ln > 0, nrm / np.maximum(ln, 1e-30), np.array([[0, 1, 0]], np.float32)).astype(np.float32)
    bbox = np.concatenate([v.min(axis=0), v.max(axis=0)]).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, name.replace("_vrip", "") + ".npz"), verts=verts, normals=normals, bbox=bbox)
    print(name, verts.shape[0], bbox)
