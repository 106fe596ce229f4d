"""The three small scenes behind tests/golden/oracle_*.npz (made by tools/make_oracle_goldens.py)."""
import numpy as np

import scenes as S
from gpupathtracer_amd import meshes

NAMES = ("default", "materials", "meshes")


def scene(name):
    """(spheres, polys, camera, depth, pass counters, row_stride_is_width)"""
    if name == "default":          # the reference's default scene, reference pixel arithmetic, its 30 bounces
        return S.default_spheres(), None, S.orbit_camera(64, 64), 30, (0, 1), 0
    if name == "materials":        # every material preset on a sphere: all branches incl. absorbing and scattering media
        names = ["steel", "greenGlass", "marble", "ketchup", "something", "glass", "gold", "white",
                 "red", "orange", "purple", "lightBlue", "light", "green"]
        return S.spheres_with(names), None, S.orbit_camera(64, 48), 16, (2,), 1
    if name == "meshes":           # glass diamond + marble coil + spheres
        dv, dn = meshes.load_asset("diamond")
        cv, cn = meshes.load_asset("coil")
        polys = np.concatenate([
            meshes.polys_from_triangles(meshes.transform(dv, 0.8, (0.2, -2.2, 2.6)), dn, S.MATERIALS["glass"]),
            meshes.polys_from_triangles(meshes.transform(cv, 0.25, (-1.2, -2.2, 1.2)), cn, S.MATERIALS["marble"])])
        return S.default_spheres(), polys, S.orbit_camera(64, 64), 8, (1,), 0
    raise KeyError(name)
