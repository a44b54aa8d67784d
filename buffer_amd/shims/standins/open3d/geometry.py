import numpy as np

from .utility import Vector3dVector


class KDTreeSearchParamKNN:
    def __init__(self, knn=30):
        self.knn = int(knn)


class KDTreeSearchParamHybrid:
    def __init__(self, radius, max_nn):
        self.radius, self.max_nn = float(radius), int(max_nn)


def _device():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("open3d stand-in (buffer_amd): this call runs on a HIP device and none is visible")
    return torch.device("cuda", torch.cuda.current_device())


class PointCloud:
    def __init__(self, points=None):
        self._points = Vector3dVector()
        self._normals = Vector3dVector()
        self._colors = Vector3dVector()
        if points is not None:
            self.points = points

    points = property(lambda s: s._points, lambda s, v: setattr(s, "_points", Vector3dVector(v)))
    normals = property(lambda s: s._normals, lambda s, v: setattr(s, "_normals", Vector3dVector(v)))
    colors = property(lambda s: s._colors, lambda s, v: setattr(s, "_colors", Vector3dVector(v)))

    def has_points(self):
        return len(self._points) > 0

    def has_normals(self):
        return len(self._normals) > 0 and len(self._normals) == len(self._points)

    def has_colors(self):
        return len(self._colors) > 0 and len(self._colors) == len(self._points)

    def paint_uniform_color(self, color):
        self._colors = Vector3dVector(np.repeat(np.asarray(color, np.float64).reshape(1, 3), len(self._points), 0))
        return self

    def transform(self, T):
        T = np.asarray(T, np.float64)
        self._points = Vector3dVector(np.asarray(self._points) @ T[:3, :3].T + T[:3, 3])
        if self.has_normals():
            self._normals = Vector3dVector(np.asarray(self._normals) @ T[:3, :3].T)
        return self

    def voxel_down_sample(self, voxel_size):
        """-> new PointCloud of voxel means (csrc/preprocess.hip buf_voxel_downsample; fp64 means like open3d's)."""
        import torch
        from buffer_amd import preprocess
        dev = _device()
        pts = torch.from_numpy(np.asarray(self._points)).to(dev)
        out = PointCloud()
        if self.has_normals():
            m, nm = preprocess.voxel_down_sample(pts, voxel_size, normals=torch.from_numpy(np.asarray(self._normals)).to(dev))
            out._normals = Vector3dVector(nm.cpu().numpy())
        else:
            m = preprocess.voxel_down_sample(pts, voxel_size)
        out._points = Vector3dVector(m.cpu().numpy())
        if self.has_colors():                                # colours are display only in the reference: uniform paint kept
            out._colors = Vector3dVector(np.repeat(np.asarray(self._colors)[:1], len(out._points), 0))
        return out

    def estimate_normals(self, search_param=None, fast_normal_computation=True):
        """30-NN PCA normals by default (csrc/preprocess.hip buf_knn_normals); orientation is left as computed."""
        import torch
        from buffer_amd import preprocess
        if isinstance(search_param, KDTreeSearchParamHybrid):
            raise NotImplementedError("open3d stand-in: only KDTreeSearchParamKNN neighbourhoods are provided")
        knn = search_param.knn if search_param is not None else 30
        pts = torch.from_numpy(np.asarray(self._points, np.float32)).to(_device())
        self._normals = Vector3dVector(preprocess.estimate_normals(pts, knn=knn, orient=False).cpu().numpy())
        return self

    def orient_normals_towards_camera_location(self, camera_location=(0.0, 0.0, 0.0)):
        if not self.has_normals():
            raise RuntimeError("[Open3D Error] No normals in the PointCloud. Call EstimateNormals() first.")
        p, n = np.asarray(self._points), np.asarray(self._normals).copy()
        cam = np.asarray(camera_location, np.float64).reshape(1, 3)
        zero = np.linalg.norm(n, axis=1) == 0.0                 # open3d: a zero normal becomes the view direction
        view = cam - p
        n[zero] = view[zero] / np.maximum(np.linalg.norm(view[zero], axis=1, keepdims=True), 1e-300)
        flip = (n * view).sum(1) < 0.0
        n[flip] *= -1.0
        self._normals = Vector3dVector(n)
        return self
