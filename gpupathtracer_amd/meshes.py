"""Triangle-mesh helpers for the harness side of the C-ABI: bundled assets,
deterministic 1->4 midpoint subdivision (the stand-in for the reference's
missing cow_heavy.obj) and Poly-array construction as PathTracer::prepMesh +
createDeviceData do it (path_tracer.cpp:58-67, 275-278)."""
import os

import numpy as np

from .types import POLY_DTYPE

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
f32 = np.float32


def load_asset(name):
    """(verts[n,9], normals[n,3]) of a bundled mesh: the Poly.p0/p1/p2 and Poly.n
    values the OBJ loader produces for the reference's `<name>.obj`."""
    z = np.load(os.path.join(_ASSETS, name + ".npz"))
    return z["verts"].astype(f32), z["normals"].astype(f32)


def subdivide(verts, normals, levels):
    """`levels` rounds of 1->4 midpoint subdivision.  Midpoints are (a+b)*0.5 in
    float32 (commutative, so shared edges stay watertight); children inherit the
    parent's flat normal.  Child order: (a,ab,ca) (ab,b,bc) (ca,bc,c) (ab,bc,ca)."""
    v = np.asarray(verts, dtype=f32).reshape(-1, 3, 3)
    n = np.asarray(normals, dtype=f32).reshape(-1, 3)
    for _ in range(levels):
        a, b, c = v[:, 0], v[:, 1], v[:, 2]
        ab = ((a + b) * f32(0.5)).astype(f32)
        bc = ((b + c) * f32(0.5)).astype(f32)
        ca = ((c + a) * f32(0.5)).astype(f32)
        kids = np.stack([
            np.stack([a, ab, ca], axis=1),
            np.stack([ab, b, bc], axis=1),
            np.stack([ca, bc, c], axis=1),
            np.stack([ab, bc, ca], axis=1),
        ], axis=1)                                  # [n, 4, 3, 3]
        v = kids.reshape(-1, 3, 3)
        n = np.repeat(n, 4, axis=0)
    return v.reshape(-1, 9).astype(f32), n.astype(f32)


def transform(verts, scale=1.0, translate=(0.0, 0.0, 0.0)):
    v = np.asarray(verts, dtype=f32).reshape(-1, 3, 3)
    return ((v * f32(scale)).astype(f32) + np.asarray(translate, dtype=f32)).astype(f32).reshape(-1, 9)


def polys_from_triangles(verts, normals, material):
    verts = np.asarray(verts, dtype=f32).reshape(-1, 9)
    p = np.zeros(verts.shape[0], dtype=POLY_DTYPE)
    p["position"] = (0.0, -0.8, 0.0)
    p["p0"] = verts[:, 0:3]
    p["p1"] = verts[:, 3:6]
    p["p2"] = verts[:, 6:9]
    p["n"] = np.asarray(normals, dtype=f32).reshape(-1, 3)
    p["material"] = material
    return p


def random_soup(n_tris, seed=12345, extent=1.0, edge=0.01):
    """Deterministic synthetic mesh: n small triangles uniformly placed in
    [-extent, extent]^3 with edge length ~`edge` (numpy PCG64 stream `seed`)."""
    rng = np.random.default_rng(seed)
    c = (rng.random((n_tris, 1, 3), dtype=f32) * 2 - 1) * f32(extent)
    d = (rng.random((n_tris, 3, 3), dtype=f32) - f32(0.5)) * f32(edge)
    v = (c + d).astype(f32)
    a = v[:, 1] - v[:, 0]
    b = v[:, 2] - v[:, 0]
    nrm = np.cross(a, b).astype(f32)
    ln = np.sqrt((nrm * nrm).sum(axis=1)).astype(f32)
    ln[ln == 0] = 1
    return v.reshape(-1, 9), (nrm / ln[:, None]).astype(f32)
