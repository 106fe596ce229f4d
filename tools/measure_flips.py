"""GPU: measures, for the scenes of the parity tests, the fraction of samples whose path differs
between the HIP path and the oracle, and the worst relative radiance error among samples with
identical paths.  The bounds in tests/ are set from this output."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle_lib as O  # noqa: E402
import parity  # noqa: E402
import scenes as S  # noqa: E402
from gpupathtracer_amd import api, meshes  # noqa: E402


def strided_ids(w, h, stride):
    ys, xs = np.meshgrid(np.arange(stride // 2, h, stride), np.arange(stride // 2, w, stride), indexing="ij")
    return (ys * w + xs).astype(np.int32).ravel()


def measure(label, sph, polys, cam, depth, counters, stride_mode=1, ids=None, **kw):
    w, h = int(cam["resolution"][0]), int(cam["resolution"][1])
    with api.Scene(sph, polys, api.render_params(max_depth=depth, row_stride_is_width=stride_mode, **kw)) as sc:
        for c in counters:
            got, ev, pr = sc.render_events(cam, c)
            ref = O.render_pass(sph, polys, cam, c, O.params(max_depth=depth, row_stride_is_width=stride_mode),
                                pixel_ids=ids, want_events=True)
            g, e, p = (got, ev, pr) if ids is None else (got[ids], ev[ids], pr[ids])
            same = parity.same_paths(e, p, ref["events"], ref["prims"])
            r = parity.rel_err(g, ref["rgb"])
            first = np.argmax((e != ref["events"]) | (p != ref["prims"]), axis=1)[~same]
            print("%-28s c=%d n=%d flips=%.3e beyond_tol=%.3e beyond_tol_same=%d worst_same=%.3e first_diff_depth_hist=%s"
                  % (label, c, g.shape[0], (~same).mean(), (r > 1e-4).mean(), int((r[same] > 1e-4).sum()),
                     r[same].max() if same.any() else 0, np.bincount(first, minlength=1)[:10].tolist()), flush=True)
            if (r[same] > 1e-4).any():
                k = np.flatnonzero(same & (r > 1e-4))[:5]
                for i in k:
                    print("   sample", i, "got", g[i], "ref", ref["rgb"][i], "events", e[i][:12].tolist(), flush=True)


sph = S.default_spheres()
which = set(sys.argv[1:])
def on(name): return not which or name in which

if on("default"):
    for depth in (8, 30):
        measure("default 256^2 d%d" % depth, sph, None, S.orbit_camera(256, 256), depth, (0, 1, 7), stride_mode=0)
if on("materials"):
    names = ["steel", "greenGlass", "marble", "ketchup", "something", "glass", "gold", "white",
             "red", "orange", "purple", "lightBlue", "light", "green"]
    measure("all materials 192^2 d16", S.spheres_with(names), None, S.orbit_camera(192, 192), 16, (2, 3), stride_mode=0)
if on("mesh"):
    dv, dn = meshes.load_asset("diamond"); dv = meshes.transform(dv, 0.8, (0.2, -2.2, 2.6))
    cv, cn = meshes.load_asset("coil"); cv = meshes.transform(cv, 0.25, (-1.2, -2.2, 1.2))
    polys = np.concatenate([meshes.polys_from_triangles(dv, dn, S.MATERIALS["glass"]),
                            meshes.polys_from_triangles(cv, cn, S.MATERIALS["marble"])])
    for kd in (0, 1):
        measure("diamond+coil 128^2 kd=%d" % kd, sph, polys, S.orbit_camera(128, 128), 8, (1, 2), stride_mode=0, use_kdtree=kd)
if on("nonsquare"):
    for mode in (0, 1):
        measure("160x90 stride mode %d" % mode, sph, None, S.orbit_camera(160, 90), 6, (0, 1), stride_mode=mode)
if on("box"):
    bv, bn = meshes.load_asset("box"); bv = meshes.transform(bv, 0.3, (0.0, -0.5, 1.8))
    measure("glass box 64^2", sph, meshes.polys_from_triangles(bv, bn, S.MATERIALS["glass"]), S.orbit_camera(64, 64), 8, (0, 1, 2, 3), stride_mode=0)
if on("spheres1080"):
    w, h = 1920, 1080
    measure("spheres 1080p subset", sph, None, S.orbit_camera(w, h), 8, (4,), ids=strided_ids(w, h, 13))
if on("cow"):
    w, h = 1920, 1080
    for sub, stride in ((3, 40), (4, 40)):
        v, n = meshes.load_asset("cow"); v, n = meshes.subdivide(v, n, sub)
        polys = meshes.polys_from_triangles(v, n, S.MATERIALS["glass"])
        measure("cow subdiv %d 1080p subset" % sub, sph, polys, S.orbit_camera(w, h), 8, (0,), ids=strided_ids(w, h, stride))
if on("soup"):
    w, h = 3840, 2160
    v, n = meshes.random_soup(2000000, seed=12345, extent=1.0, edge=0.01); v = meshes.transform(v, 0.75, (0.0, 0.1, 0.0))
    polys = meshes.polys_from_triangles(v, n, S.MATERIALS["white"])
    measure("soup 2M 4K subset", sph, polys, S.orbit_camera(w, h), 8, (0,), ids=strided_ids(w, h, 90), passes_in_flight=1)
