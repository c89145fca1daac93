import numpy as np


def convert_heightfield_to_trimesh(height_field_raw, horizontal_scale, vertical_scale, slope_threshold=None):
    """Grid -> (vertices, triangles); only shapes matter for the golden generator."""
    hf = np.asarray(height_field_raw)
    nr, nc = hf.shape
    yy, xx = np.meshgrid(np.arange(nc) * horizontal_scale, np.arange(nr) * horizontal_scale)
    v = np.stack([xx.ravel(), yy.ravel(), hf.ravel() * vertical_scale], 1).astype(np.float32)
    t = np.zeros((2 * (nr - 1) * (nc - 1), 3), dtype=np.uint32)
    return v, t


class SubTerrain:
    pass
