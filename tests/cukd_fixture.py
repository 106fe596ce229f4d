"""The ray set of the reference's kd-tree benchmark (cukd/test/ray_traversal.cu:102-114, 153-159): width x height
parallel rays along +z from z = -5 over the Stanford dragon decimations that ship beside it
(cukd/test/dragon/dragon_vrip_res3.ply / _res4.ply; here gpupathtracer_amd/assets/dragon_res3.npz / dragon_res4.npz,
made by tools/make_assets.py).  The x coordinate is the benchmark's own: (xmax - xmin) / width * j - xmax."""
import numpy as np

WIDTH, HEIGHT = 1024, 864
XMIN, XMAX, YMIN, YMAX = np.float32(-0.1), np.float32(0.11), np.float32(0.03), np.float32(0.23)


def rays(width=WIDTH, height=HEIGHT):
    j = np.arange(width, dtype=np.float32)
    i = np.arange(height, dtype=np.float32)
    x = (XMAX - XMIN) / np.float32(width) * j - XMAX                 # float arithmetic, left to right, as the C++
    y = -(YMAX - YMIN) / np.float32(height) * i + YMAX
    r = np.zeros((height, width, 6), np.float32)
    r[:, :, 0] = x[None, :]
    r[:, :, 1] = y[:, None]
    r[:, :, 2] = -5.0
    r[:, :, 5] = 1.0
    return r.reshape(-1, 6)
