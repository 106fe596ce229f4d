"""Experiment: how many spheres does a secondary ray still have to test when a table indexed by (origin cell, direction
bin) names the spheres a ray from that cell in that direction can possibly hit?  Rays: the path queues entering depths
1..4 of a 1080p pass.  Printed: mean candidates per ray and the mean over waves (64 consecutive queue entries) of the
largest count in the wave (what a per-lane loop costs), against the 14 tests of the full loop."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpupathtracer_amd import api
import bench


def build_table(centres, radii, lo, hi, G, D):
    """mask[cell, bin]: conservative set of spheres hit by some ray with origin in the cell and direction in the bin."""
    n = len(centres)
    cell = (hi - lo) / G
    rho = 0.5 * np.linalg.norm(cell)
    # bins: cube faces (axis, sign) x D x D patches; bounding cone of each patch
    axes, halfangles = [], []
    for face in range(6):
        ax, sg = face >> 1, 1.0 if (face & 1) == 0 else -1.0
        for iu in range(D):
            for iv in range(D):
                u0, u1 = -1 + 2 * iu / D, -1 + 2 * (iu + 1) / D
                v0, v1 = -1 + 2 * iv / D, -1 + 2 * (iv + 1) / D
                def vec(u, v):
                    d = np.zeros(3); d[ax] = sg; d[(ax + 1) % 3] = u; d[(ax + 2) % 3] = v
                    return d / np.linalg.norm(d)
                a = vec((u0 + u1) / 2, (v0 + v1) / 2)
                th = max(np.arccos(np.clip(a @ vec(u, v), -1, 1)) for u in (u0, u1) for v in (v0, v1))
                axes.append(a); halfangles.append(th + 2e-3)
    axes, halfangles = np.array(axes), np.array(halfangles)
    idx = np.stack(np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing="ij"), -1).reshape(-1, 3)
    pc = lo + (idx + 0.5) * cell
    mask = np.zeros((G ** 3, len(axes)), np.uint32)
    for s in range(n):
        w = centres[s] - pc
        dist = np.linalg.norm(w, axis=1)
        Reff = np.sqrt(radii[s] ** 2 + 2e-6 * (dist + rho) ** 2) * 1.001 + rho
        inside = dist <= Reff
        alpha = np.arcsin(np.clip(Reff / np.maximum(dist, 1e-30), 0, 1))
        cosang = np.clip((w / np.maximum(dist, 1e-30)[:, None]) @ axes.T, -1, 1)
        keep = inside[:, None] | (np.arccos(cosang) <= halfangles[None, :] + alpha[:, None])
        mask |= keep.astype(np.uint32) << np.uint32(s)
    return mask


def lookup(mask, lo, hi, G, D, o, d, n_spheres):
    cell = (hi - lo) / G
    ci = np.floor((o - lo) / cell).astype(np.int64)
    inside = ((ci >= 0) & (ci < G)).all(1)
    ci = np.clip(ci, 0, G - 1)
    ad = np.abs(d)
    ax = np.argmax(ad, axis=1)
    k = np.arange(len(d))
    sg = d[k, ax] < 0
    u = d[k, (ax + 1) % 3] / np.maximum(ad[k, ax], 1e-30)
    v = d[k, (ax + 2) % 3] / np.maximum(ad[k, ax], 1e-30)
    iu = np.clip(np.floor((u + 1) / 2 * D).astype(np.int64), 0, D - 1)
    iv = np.clip(np.floor((v + 1) / 2 * D).astype(np.int64), 0, D - 1)
    b = ((ax * 2 + sg) * D + iu) * D + iv
    m = mask[(ci[:, 0] * G + ci[:, 1]) * G + ci[:, 2], b]
    return np.where(inside, m, np.uint32((1 << n_spheres) - 1)), inside


w, h = 1920, 1080
for workload in ("cow_heavy", "spheres"):
    spheres, polys = bench.build_workload(workload, 4)
    cam = api.default_camera(w, h)
    c = spheres["position"].astype(np.float64); r = spheres["radius"].astype(np.float64)
    lo, hi = (c - r[:, None]).min(0), (c + r[:, None]).max(0)
    if polys is not None:
        pts = np.concatenate([polys["p0"], polys["p1"], polys["p2"]]).astype(np.float64)
        lo, hi = np.minimum(lo, pts.min(0)), np.maximum(hi, pts.max(0))
    lo[1] = min(lo[1], -0.8)
    pad = 0.1 * (hi - lo).max(); lo, hi = lo - pad, hi + pad
    with api.Scene(spheres, polys, api.render_params(max_depth=8)) as sc:
        sc.reset(w * h)
        queues = {depth: sc.dump_queue(cam, 0, 1, depth, w * h)[0] for depth in (1, 2, 4)}
    for G, D in ((8, 4), (16, 4), (16, 8), (32, 8)):
        mask = build_table(c, r, lo, hi, G, D)
        line = []
        for depth, rays in queues.items():
            m, inside = lookup(mask, lo, hi, G, D, rays[:, :3].astype(np.float64), rays[:, 3:].astype(np.float64), len(c))
            pc = np.array([bin(int(x)).count("1") for x in np.unique(m)]); pcmap = dict(zip(np.unique(m).tolist(), pc.tolist()))
            cnt = np.vectorize(pcmap.get)(m)
            nw = len(cnt) // 64
            wmax = cnt[: nw * 64].reshape(nw, 64).max(1)
            line.append("depth %d: in-grid %.2f mean %.2f wave-max %.2f" % (depth, inside.mean(), cnt.mean(), wmax.mean()))
        print("%s G=%d D=%d (table %.0f KB): %s" % (workload, G, D, mask.size * 4 / 1024, "; ".join(line)), flush=True)
