"""Test-side scene tables: an independent Python restatement of the reference's
hard-coded scene (path_tracer.cpp:116-278) and orbit camera
(interactive_camera.cpp:15-25,66-87; display.cpp:78-85).  Used to feed BOTH the
oracle and the HIP library with the same inputs, and to cross-check the
library's own gpt_default_scene()/gpt_orbit_camera().
"""
import ctypes
import math

import numpy as np

from gpupathtracer_amd.types import (CAMERA_DTYPE, MATERIAL_DTYPE, POLY_DTYPE, SPHERE_DTYPE,
                                     default_material, make_camera)

_libm = ctypes.CDLL("libm.so.6")
_libm.sinf.restype = ctypes.c_float
_libm.sinf.argtypes = [ctypes.c_float]
_libm.cosf.restype = ctypes.c_float
_libm.cosf.argtypes = [ctypes.c_float]

f32 = np.float32


def _mat(diffuse=None, specular=None, emitted=None, ior=None, transmits=False,
         absorption=None, scattering=None):
    m = default_material()
    if diffuse is not None:
        m["diffuseColor"] = diffuse
    if specular is not None:
        m["specularColor"] = specular
    if emitted is not None:
        m["emittedColor"] = emitted
    if ior is not None:
        m["medium"]["refractiveIndex"] = ior
    m["hasTransmission"] = 1 if transmits else 0
    if absorption is not None:
        m["medium"]["absorptionAndScatteringProperties"]["absorptionCoefficient"] = absorption
    if scattering is not None:
        m["medium"]["absorptionAndScatteringProperties"]["reducedScatteringCoefficient"] = scattering
    return m


# path_tracer.cpp:124-215.  `purple` ends up with a black diffuse colour because
# of the assignment at :150.
MATERIALS = {
    "red": _mat(diffuse=(0.87, 0.15, 0.15), specular=(1, 1, 1), ior=1.491),
    "green": _mat(diffuse=(0.15, 0.87, 0.15), specular=(1, 1, 1), ior=1.491),
    "orange": _mat(diffuse=(0.93, 0.33, 0.04), specular=(1, 1, 1), ior=1.491),
    "purple": _mat(diffuse=(0, 0, 0), specular=(1, 1, 1), ior=1.491),
    "glass": _mat(specular=(1, 1, 1), ior=2.42, transmits=True),
    "greenGlass": _mat(specular=(1, 1, 1), ior=1.62, transmits=True, absorption=(1.0, 0.01, 1.0)),
    "marble": _mat(specular=(1, 1, 1), ior=1.486, transmits=True, absorption=(0.6, 0.6, 0.6),
                   scattering=8.0),
    "something": _mat(specular=(1, 1, 1), ior=1.333, transmits=True, absorption=(0.9, 0.3, 0.02),
                      scattering=2.0),
    "ketchup": _mat(specular=(1, 1, 1), ior=1.350, transmits=True, absorption=(0.02, 5.1, 5.7),
                    scattering=9.0),
    "white": _mat(diffuse=(0.9, 0.9, 0.9)),
    "lightBlue": _mat(diffuse=(0.4, 0.6, 0.8), specular=(1, 1, 1), ior=1.2),
    "gold": _mat(specular=(0.869, 0.621, 0.027), ior=1000.0),
    "steel": _mat(specular=(0.89, 0.89, 0.89), ior=1000.0),
    "light": _mat(emitted=(13, 13, 11)),
}

# path_tracer.cpp:217-271
_SPHERES = [
    ((-0.9, 0, -0.9), 0.8, "steel"),
    ((0.9, -0.5, 1.3), 0.3, "greenGlass"),
    ((-0.5, -0.4, 1.0), 0.4, "lightBlue"),
    ((-1.0, -0.7, 1.2), 0.1, "lightBlue"),
    ((-0.5, -0.7, 1.7), 0.1, "lightBlue"),
    ((0.3, -0.7, 1.4), 0.1, "lightBlue"),
    ((-0.1, -0.7, 0.1), 0.1, "lightBlue"),
    ((0.2, -0.55, 0.7), 0.25, "lightBlue"),
    ((0.8, 0, -0.4), 0.8, "green"),
    ((0.8, 1.2, -0.4), 0.4, "green"),
    ((0.8, 1.8, -0.4), 0.2, "green"),
    ((0.8, 2.1, -0.4), 0.1, "green"),
    ((0.8, 2.25, -0.4), 0.05, "green"),
    ((0.8, 2.325, -0.4), 0.025, "green"),
]


def default_spheres():
    s = np.zeros(len(_SPHERES), dtype=SPHERE_DTYPE)
    for i, (pos, r, mat) in enumerate(_SPHERES):
        s[i]["position"] = pos
        s[i]["radius"] = r
        s[i]["material"] = MATERIALS[mat]
    return s


def spheres_with(materials):
    """The default sphere geometry with a different material per sphere (list of names)."""
    s = default_spheres()
    for i, name in enumerate(materials):
        s[i]["material"] = MATERIALS[name]
    return s


def orbit_camera(width, height, fovx=45.0, yaw=0.0, pitch=0.3, radius=4.0, aperture=0.04,
                 center=(0.0, 0.0, 0.0)):
    """InteractiveCamera(...) -> setResolution -> setFOVX -> buildRenderCamera."""
    pi_f = f32(3.1415926535897932384626422832795028841971)  # BasicMath::PI is a float
    yaw, pitch, radius = f32(yaw), f32(pitch), f32(radius)
    # degreesToRadians (basic_math.cpp:33-36): double arithmetic, float result
    rad = f32(float(f32(fovx)) / 180.0 * float(pi_f))
    ratio = f32(f32(height) / f32(width))
    # setFOVX (interactive_camera.cpp:66-70)
    inner = f32(math.atan(math.tan(float(rad) * 0.5) * float(ratio)) * 2.0)
    fovy = f32(float(inner) * 180.0 / float(pi_f))  # radiansToDegrees
    sx = f32(_libm.sinf(yaw)) * f32(_libm.cosf(pitch))
    sy = f32(_libm.sinf(pitch))
    sz = f32(_libm.cosf(yaw)) * f32(_libm.cosf(pitch))
    d = np.array([sx, sy, sz], dtype=f32)
    c = np.array(center, dtype=f32)
    eye = (c + d * radius).astype(f32)
    return make_camera((width, height), eye, -d, (0, 1, 0), (f32(fovx), fovy), f32(aperture), radius)


def polys_from_triangles(verts, normals, material):
    """Poly array as PathTracer::prepMesh + createDeviceData build it
    (path_tracer.cpp:58-67, 275-278)."""
    verts = np.asarray(verts, dtype=f32).reshape(-1, 9)
    p = np.zeros(verts.shape[0], dtype=POLY_DTYPE)
    p["position"] = (0.0, -0.8, 0.0)
    p["p0"] = verts[:, 0:3]
    p["p1"] = verts[:, 3:6]
    p["p2"] = verts[:, 6:9]
    p["n"] = np.asarray(normals, dtype=f32).reshape(-1, 3)
    p["material"] = material
    return p


def geometric_normals(verts):
    """normalize(cross(p1-p0, p2-p0)) in float32 (obj.cpp:87-89 style)."""
    v = np.asarray(verts, dtype=f32).reshape(-1, 3, 3)
    a = v[:, 1] - v[:, 0]
    b = v[:, 2] - v[:, 0]
    n = np.cross(a, b).astype(f32)
    ln = np.sqrt((n * n).sum(axis=1, dtype=f32)).astype(f32)
    ln[ln == 0] = 1
    return (n / ln[:, None]).astype(f32)


def cornell_box(box_verts, box_normals):
    """A Cornell-style room from the reference's box.obj (a cube of +-0.8): scaled x3 and lifted so that its
    floor lies in the ground plane y = -0.8; the front (+z) wall is left out so that the default orbit
    camera, which stands outside at z = 3.8, looks in, and so is the floor (the kernel's hard-coded ground
    plane is the floor).  Left wall `red`, right wall `green`, back wall `white`, ceiling `light`
    (the only emitter).  box.obj's faces come in pairs: front, top, back, bottom, right, left."""
    v = np.asarray(box_verts, dtype=f32).reshape(-1, 3, 3)
    v = (v * f32(3.0) + np.array([0.0, 1.6, 0.0], dtype=f32)).astype(f32).reshape(-1, 9)
    n = np.asarray(box_normals, dtype=f32).reshape(-1, 3)
    walls = [(2, "light"), (3, "light"), (4, "white"), (5, "white"), (8, "green"), (9, "green"), (10, "red"), (11, "red")]
    polys = np.concatenate([polys_from_triangles(v[i:i + 1], n[i:i + 1], MATERIALS[m]) for i, m in walls])
    return polys
