"""numpy views of the reference's scene PODs (layouts of include/gpt.h).

Offsets follow the CUDA struct layouts of the reference headers:
camera.h:6-14, sphere.h:7-14, poly.h:7-17, material.h:8-16, medium.h:8-13,
absorption_and_scattering_properties.h:7-12, ray.h:6-11.
"""
import numpy as np

F3 = (np.float32, (3,))

AS_DTYPE = np.dtype({
    "names": ["absorptionCoefficient", "reducedScatteringCoefficient"],
    "formats": [F3, np.float32],
    "offsets": [0, 12],
    "itemsize": 16,
})

MEDIUM_DTYPE = np.dtype({
    "names": ["refractiveIndex", "absorptionAndScatteringProperties"],
    "formats": [np.float32, AS_DTYPE],
    "offsets": [0, 4],
    "itemsize": 20,
})

MATERIAL_DTYPE = np.dtype({
    "names": ["diffuseColor", "emittedColor", "specularColor", "hasTransmission", "medium"],
    "formats": [F3, F3, F3, np.uint8, MEDIUM_DTYPE],
    "offsets": [0, 12, 24, 36, 40],
    "itemsize": 60,
})

SPHERE_DTYPE = np.dtype({
    "names": ["position", "radius", "material"],
    "formats": [F3, np.float32, MATERIAL_DTYPE],
    "offsets": [0, 12, 16],
    "itemsize": 76,
})

POLY_DTYPE = np.dtype({
    "names": ["position", "p0", "p1", "p2", "n", "material"],
    "formats": [F3, F3, F3, F3, F3, MATERIAL_DTYPE],
    "offsets": [0, 12, 24, 36, 48, 60],
    "itemsize": 120,
})

RAY_DTYPE = np.dtype({
    "names": ["origin", "direction"],
    "formats": [F3, F3],
    "offsets": [0, 12],
    "itemsize": 24,
})

CAMERA_DTYPE = np.dtype({
    "names": ["resolution", "position", "view", "up", "fov", "apertureRadius", "focalDistance"],
    "formats": [(np.float32, (2,)), F3, F3, F3, (np.float32, (2,)), np.float32, np.float32],
    "offsets": [0, 8, 20, 32, 48, 56, 60],
    "itemsize": 64,
})

# gpt_render_params (include/gpt.h)
RENDER_PARAMS_DTYPE = np.dtype({
    "names": ["max_depth", "ray_bias", "min_ray_weight", "ground_elevation", "ground_enabled",
              "row_stride_is_width", "use_kdtree", "kd_small_threshold", "passes_in_flight", "reserved"],
    "formats": [np.int32, np.float32, np.float64, np.float32, np.int32, np.int32, np.int32,
                np.int32, np.int32, (np.int32, (6,))],
    "offsets": [0, 4, 8, 16, 20, 24, 28, 32, 36, 40],
    "itemsize": 64,
})

AIR_IOR = np.float32(1.000293)  # medium.h:17


def default_material():
    """SET_DEFAULT_MATERIAL_PROPERTIES (material.h:19-26): all black, no
    transmission, medium = air."""
    m = np.zeros((), dtype=MATERIAL_DTYPE)
    m["medium"]["refractiveIndex"] = AIR_IOR
    return m


def make_camera(resolution, position, view, up, fov, aperture_radius, focal_distance):
    c = np.zeros((), dtype=CAMERA_DTYPE)
    c["resolution"] = resolution
    c["position"] = position
    c["view"] = view
    c["up"] = up
    c["fov"] = fov
    c["apertureRadius"] = aperture_radius
    c["focalDistance"] = focal_distance
    return c
