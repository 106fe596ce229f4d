"""Inputs of the helper-function pin (tests/golden/ref_helpers.npz), shared by the generator
(tools/make_goldens.py, container only) and by the tests that compare the oracle, the host build of
pt_device.h and the GPU against the stored outputs of the reference's own text.

Only OUTPUTS are stored; the inputs are recomputed here from a counter-based integer hash
(splitmix64 over uint64 arrays) and IEEE float32 +, *, /, sqrt -- nothing that depends on a random
number library's stream -- and the golden file keeps a checksum of every input block, so a test
knows that it fed the same words the reference was fed.

The inputs are drawn the way the renderer forms the arguments: unit directions, unit and nearly-unit
normals (Poly.n is used as read from the file), refractive-index pairs of the material presets in
both orders (total internal reflection included), xi = float(x - 1) / 2^31 with xi == 0 and
xi == 1.0f, sigma * distance products from 0 to overflow.
Record layouts (32-bit words) are those of oracle/ref_build/helpers_dump.cpp."""
import hashlib

import numpy as np

N = 102400          # records per function

# name -> (words in, words out)
LAYOUT = {
    "randhash": (1, 1), "position": (7, 3), "ground": (7, 4), "triangle": (15, 4), "cosine": (5, 3),
    "sphere_dir": (2, 3), "reflect": (6, 3), "transmit": (8, 3), "fresnel": (14, 2), "surface": (8, 8),
    "background": (3, 3), "transmission": (4, 3),
}
FUNCTIONS = tuple(LAYOUT)

AIR = np.float32(1.000293)
IORS = np.array([1.000293, 1.491, 2.42, 1.62, 1.486, 1.333, 1.35, 1.2, 1000.0], np.float32)
SIGMAS = np.array([[1.0, 0.01, 1.0], [0.6, 0.6, 0.6], [0.9, 0.3, 0.02], [0.02, 5.1, 5.7], [0.0, 0.0, 0.0]], np.float32)


def _mix(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return x


def bits(stream, n, cols=1):
    """n x cols uint32 words of stream `stream`."""
    with np.errstate(over="ignore"):
        idx = np.arange(n * cols, dtype=np.uint64) + (np.uint64(stream) << np.uint64(40))
        return (_mix(_mix(idx)) >> np.uint64(32)).astype(np.uint32).reshape(n, cols)


def unit(stream, n, cols=1):
    """floats in [0, 1) with 24 random bits."""
    return ((bits(stream, n, cols) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def sym(stream, n, cols=1):
    return (unit(stream, n, cols) * np.float32(2) - np.float32(1)).astype(np.float32)


def xi(stream, n):
    """Uniforms as the renderer's generator forms them: float(x - 1) / 2^31, x in [1, 2^31 - 2];
    the first entries are the edge values 0, the largest below 1, and 1.0f (x - 1 >= 2^31 - 64)."""
    x = bits(stream, n)[:, 0] % np.uint32(2147483646)
    u = (x.astype(np.float32) * np.float32(2.0 ** -31)).astype(np.float32)
    u[:6] = np.array([0.0, 1.0, np.float32(2147483520) * np.float32(2.0 ** -31), 2.0 ** -31, 0.5, 0.25], np.float32)
    return u


def normalize(v):
    v = v.astype(np.float32)
    d = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]).astype(np.float32)
    d = (d + v[:, 2] * v[:, 2]).astype(np.float32)
    inv = (np.float32(1) / np.sqrt(d, dtype=np.float32)).astype(np.float32)
    return (v * inv[:, None]).astype(np.float32)


def directions(stream, n):
    """Unit vectors; every 16th has one or two components zeroed first (axis-parallel rays)."""
    v = sym(stream, n, 3)
    v[np.abs(v).sum(axis=1) == 0] = np.float32(1)
    k = np.arange(n)
    z = (k % 16) == 0
    v[z, (k[z] // 16) % 3] = 0
    z2 = (k % 64) == 32
    v[z2, (k[z2] // 64) % 3] = 0
    v[np.abs(v).sum(axis=1) == 0] = np.float32(1)
    return normalize(v)


def normals(stream, n):
    """Unit normals; a quarter scaled to lengths in [0.9, 1.1] like normals read from a file, and every
    32nd an axis direction (the ground plane's (0, 1, 0) among them)."""
    nrm = directions(stream, n)
    k = np.arange(n)
    q = (k % 4) == 1
    nrm[q] = (nrm[q] * (np.float32(0.9) + np.float32(0.2) * unit(stream + 1, n)[q])).astype(np.float32)
    ax = (k % 32) == 2
    e = np.zeros((n, 3), np.float32)
    e[k, (k // 32) % 3] = np.where((k // 96) % 2 == 0, 1.0, -1.0)
    nrm[ax] = e[ax]
    return nrm


def ior_pairs(stream, n):
    """(incident, transmitted): air -> material and material -> air, half and half."""
    m = IORS[bits(stream, n)[:, 0] % np.uint32(len(IORS))]
    flip = (bits(stream + 1, n)[:, 0] & np.uint32(1)).astype(bool)
    n1 = np.where(flip, m, AIR).astype(np.float32)
    n2 = np.where(flip, AIR, m).astype(np.float32)
    return n1, n2


def make(name, n=N):
    """The n x LAYOUT[name][0] input block of one function, as uint32 words."""
    s = 100 * (FUNCTIONS.index(name) + 1)
    if name == "randhash":
        a = bits(s, n)[:, 0].copy()
        a[:8] = [0, 1, 2, 0xFFFFFFFF, 0x7FFFFFFF, 2073599, 8294399, 30]
        a[8:8 + 4096] = np.arange(4096, dtype=np.uint32)           # pass counters, depths, small pixel indices
        return a.reshape(n, 1)
    if name == "position":
        o = (sym(s, n, 3) * np.float32(4)).astype(np.float32)
        d = directions(s + 1, n)
        t = (unit(s + 2, n)[:, 0] * np.float32(10)).astype(np.float32)
        t[:3] = [0.0, 123456789.0, np.inf]
        out = np.concatenate([o, d, t[:, None]], axis=1)
    elif name == "ground":
        o = (sym(s, n, 3) * np.float32(4)).astype(np.float32)
        d = directions(s + 1, n)
        k = np.arange(n)
        d[(k % 50) == 7, 1] = 0                                     # parallel to the plane: no intersection
        d[(k % 50) == 8, 1] = np.float32(1e-30)
        d[(k % 50) == 9, 1] = np.float32(-1e-30)
        o[(k % 50) == 10, 1] = np.float32(-0.8)                     # origin on the plane
        elev = np.full((n, 1), np.float32(-0.8))
        elev[(k % 10) == 3] = sym(s + 2, n)[(k % 10) == 3]
        out = np.concatenate([elev, o, d], axis=1)
    elif name == "triangle":
        # triangles of the sizes meshes hold, rays aimed at a point near the triangle (about half of them hit)
        a = sym(s, n, 3)
        size = np.ldexp(np.float32(1), -(bits(s + 1, n)[:, 0] % np.uint32(9)).astype(np.int32)).astype(np.float32)[:, None]
        b = (a + sym(s + 2, n, 3) * size).astype(np.float32)
        c = (a + sym(s + 3, n, 3) * size).astype(np.float32)
        w = unit(s + 4, n, 2)
        w = (w * np.float32(1.6) - np.float32(0.3)).astype(np.float32)                     # barycentrics in [-0.3, 1.3]
        target = (a + (b - a) * w[:, :1] + (c - a) * w[:, 1:]).astype(np.float32)
        o = (sym(s + 5, n, 3) * np.float32(4)).astype(np.float32)
        k = np.arange(n)
        on_vertex = (k % 20) == 5                                    # rays that start on a vertex of the triangle
        o[on_vertex] = a[on_vertex]
        d = normalize((target - o).astype(np.float32) + np.float32(1e-30))
        in_plane = (k % 20) == 6                                     # direction along an edge: determinant 0 or nearly
        d[in_plane] = normalize((b - a)[in_plane] + np.float32(1e-30))
        degenerate = (k % 20) == 7                                   # zero-area triangle
        c[degenerate] = b[degenerate]
        behind = (k % 20) == 8                                       # triangle behind the origin
        d[behind] = -d[behind]
        out = np.concatenate([a, b, c, o, d], axis=1)
    elif name == "cosine":
        out = np.concatenate([normals(s, n), xi(s + 2, n)[:, None], xi(s + 3, n)[::-1][:, None]], axis=1)
    elif name == "sphere_dir":
        out = np.stack([xi(s, n), xi(s + 1, n)[::-1]], axis=1)
    elif name in ("reflect", "transmit", "surface", "fresnel"):
        nrm = normals(s, n)
        inc = directions(s + 2, n)
        k = np.arange(n)
        # the renderer flips the normal towards the incident direction (:638-648); a tenth stay unflipped
        flip = ((nrm * inc).sum(axis=1) < 0) & ((k % 10) != 0)
        nrm[flip] = -nrm[flip]
        graze = (k % 8) == 3                                         # grazing incidence: total internal reflection from inside
        inc[graze] = normalize((inc[graze] - nrm[graze] * ((nrm[graze] * inc[graze]).sum(axis=1) * np.float32(0.97))[:, None]).astype(np.float32) + np.float32(1e-30))
        perp = (k % 64) == 5                                         # normal incidence
        inc[perp] = nrm[perp]
        n1, n2 = ior_pairs(s + 3, n)
        if name == "reflect":
            out = np.concatenate([nrm, inc], axis=1)
        elif name in ("transmit", "surface"):
            out = np.concatenate([nrm, inc, n1[:, None], n2[:, None]], axis=1)
        else:
            # computeFresnel on caller-given directions: R unused by it, T arbitrary -- zero (what a total internal
            # reflection leaves), short (around the 0.12345 test), on the normal's side, on the far side
            R = directions(s + 5, n)
            T = directions(s + 6, n)
            far = (T * nrm).sum(axis=1) > 0
            T[far & ((k % 3) != 0)] = -T[far & ((k % 3) != 0)]
            scale = np.ones(n, np.float32)
            scale[(k % 16) == 1] = 0
            scale[(k % 16) == 2] = np.float32(0.12345)
            scale[(k % 16) == 3] = np.float32(0.1234)
            scale[(k % 16) == 4] = np.float32(0.1235)
            T = (T * scale[:, None]).astype(np.float32)
            out = np.concatenate([nrm, inc, n1[:, None], n2[:, None], R, T], axis=1)
    elif name == "background":
        out = directions(s, n)
    elif name == "transmission":
        sig = SIGMAS[bits(s, n)[:, 0] % np.uint32(len(SIGMAS))]
        k = np.arange(n)
        # distances over the magnitudes scatter and hit distances take (2^-12 .. 2^4), and the extremes
        dist = np.ldexp(unit(s + 1, n)[:, 0], (bits(s + 2, n)[:, 0] % np.uint32(17)).astype(np.int32) - 12).astype(np.float32)
        dist[:8] = [0.0, 1e-30, 1e-10, 123456789.0, 1e10, 1e30, 3e38, np.inf]
        sig[6:8] = SIGMAS[[0, 4]]                                   # 0.01 * 3e38 overflows; 0 * inf is a NaN
        out = np.concatenate([sig, dist[:, None]], axis=1)
    else:
        raise KeyError(name)
    out = np.ascontiguousarray(out, dtype=np.float32)
    assert out.shape == (n, LAYOUT[name][0]), (name, out.shape)
    return out.view(np.uint32)


def digest(words):
    return hashlib.sha256(np.ascontiguousarray(words, dtype=np.uint32).tobytes()).hexdigest()
