import numpy as np


class Vector3dVector(np.ndarray):
    """f64[n,3]; np.asarray(v) / np.array(v) give the plain array, as with open3d's buffer protocol."""

    def __new__(cls, data=()):
        a = np.array(data, dtype=np.float64)
        if a.size == 0:
            a = a.reshape(0, 3)
        if a.ndim != 2 or a.shape[1] != 3:
            raise RuntimeError(f"Vector3dVector: expected an (n, 3) array, got shape {a.shape}")
        return np.ascontiguousarray(a).view(cls)


class Vector2iVector(np.ndarray):
    """int32[n,2] (correspondence sets)."""

    def __new__(cls, data=()):
        a = np.array(data)
        if a.size == 0:
            a = a.reshape(0, 2)
        if a.ndim != 2 or a.shape[1] != 2:
            raise RuntimeError(f"Vector2iVector: expected an (n, 2) array, got shape {a.shape}")
        return np.ascontiguousarray(a.astype(np.int32)).view(cls)
