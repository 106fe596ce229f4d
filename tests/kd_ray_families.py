"""Ray families that sit on the kd traversal's decision boundaries, and a mesh whose triangles lie in axis-aligned
planes (every one of them in a candidate split plane).  Used by tests/test_gpu_kd_degenerate.py and
tools/stress_kd_degenerate.py."""
import numpy as np


def grid_mesh(n=24):
    """n x n quads in the planes y = 0 and x = 0 (two triangles each): every triangle lies in a candidate split plane."""
    t = []
    for i in range(n):
        for j in range(n):
            a, b, c, d = (i, 0, j), (i + 1, 0, j), (i + 1, 0, j + 1), (i, 0, j + 1)
            t += [a, b, c, a, c, d]
            a, b, c, d = (0, i, j), (0, i + 1, j), (0, i + 1, j + 1), (0, i, j + 1)
            t += [a, b, c, a, c, d]
    v = (np.array(t, np.float32).reshape(-1, 9) / n * 2 - 1).astype(np.float32)
    nrm = np.zeros((len(v), 3), np.float32); nrm[:, 1] = 1
    return v, nrm


def families(v, rng, m):
    pts = v.reshape(-1, 3)
    lo, hi = pts.min(0), pts.max(0); ctr, ext = (lo + hi) / 2, float((hi - lo).max())
    k = m // 8
    o = ((rng.random((m, 3), dtype=np.float32) * 2.4 - 1.2) * ext + ctr).astype(np.float32)
    d = rng.normal(size=(m, 3)).astype(np.float32)
    # 0: random.  1: origins on vertices.  2: axis-parallel.  3: two tiny components.  4: aimed at vertices.
    # 5: aimed at edge midpoints.  6: origins on vertices AND axis-parallel.  7: far-away origins aimed at the mesh.
    o[k:2 * k] = pts[rng.integers(0, len(pts), k)]
    d[2 * k:3 * k] = 0; d[np.arange(2 * k, 3 * k), rng.integers(0, 3, k)] = rng.choice([-1.0, 1.0], k)
    tiny = rng.choice(np.array([0.0, -0.0, 1e-45, -1e-45, 1e-38, 1e-30, -1e-20, 1e-12, 1e-8], np.float32), (k, 3))
    keep = rng.integers(0, 3, k); tiny[np.arange(k), keep] = rng.choice([-1.0, 1.0], k); d[3 * k:4 * k] = tiny
    d[4 * k:5 * k] = pts[rng.integers(0, len(pts), k)] - o[4 * k:5 * k]
    tri = v[rng.integers(0, len(v), k)].reshape(k, 3, 3)
    d[5 * k:6 * k] = (tri[:, 0] + tri[:, 1]) * 0.5 - o[5 * k:6 * k]
    o[6 * k:7 * k] = pts[rng.integers(0, len(pts), k)]
    d[6 * k:7 * k] = 0; d[np.arange(6 * k, 7 * k), rng.integers(0, 3, k)] = rng.choice([-1.0, 1.0], k)
    o[7 * k:] = (ctr + rng.normal(size=(m - 7 * k, 3)) * ext * 1e4).astype(np.float32)
    d[7 * k:] = pts[rng.integers(0, len(pts), m - 7 * k)] - o[7 * k:]
    nrm = np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True); nrm[nrm == 0] = 1
    keepraw = np.zeros(m, bool); keepraw[3 * k:4 * k] = True      # the tiny-component family stays as it is
    d = np.where(keepraw[:, None], d, (d / nrm).astype(np.float32)).astype(np.float32)
    return np.concatenate([o, d], 1).astype(np.float32)


def aimed_at_vertices(v, rng, m, distance):
    """m rays from a sphere of radius `distance` root-box extents around the mesh's centre, each aimed exactly at a vertex."""
    pts = v.reshape(-1, 3)
    lo, hi = pts.min(0), pts.max(0); ctr, ext = (lo + hi) / 2, float((hi - lo).max())
    u = rng.normal(size=(m, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    o = (ctr + u * ext * distance).astype(np.float32)
    d = pts[rng.integers(0, len(pts), m)] - o
    d = (d / np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
    return np.concatenate([o, d], 1).astype(np.float32)
