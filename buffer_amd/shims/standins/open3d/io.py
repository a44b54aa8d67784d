import numpy as np

from .geometry import PointCloud


def read_point_cloud(filename, format="auto", remove_nan_points=True, remove_infinite_points=True, print_progress=False):
    """PLY vertex positions -> PointCloud (utils/tools.py:6-7).  A missing/unreadable file gives an EMPTY cloud and a
    warning, like open3d; the reference only reads x, y, z."""
    from buffer_amd.threedmatch import read_ply
    pc = PointCloud()
    try:
        pts = read_ply(filename).astype(np.float64)
    except (OSError, ValueError) as e:
        print(f"[Open3D WARNING] Read PLY failed: unable to open file: {filename} ({e})")
        return pc
    keep = np.ones(len(pts), bool)
    if remove_nan_points:
        keep &= ~np.isnan(pts).any(1)
    if remove_infinite_points:
        keep &= ~np.isinf(pts).any(1)
    pc.points = pts[keep]
    return pc
