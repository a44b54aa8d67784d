"""cpp_wrappers.cpp_subsampling.grid_subsampling -- drop-in for the CPython module built from
cpp_wrappers/cpp_subsampling/wrapper.cpp (:62-333 subsample_batch, :338-566 subsample).

Same call conventions: positional points (and batches), everything else keyword-only; array-likes are
coerced to C-contiguous float32 / int32; new numpy arrays are returned; failures raise RuntimeError.
Rows come back per batch element in ascending voxel-key order (the reference: libstdc++
unordered_map order) -- same multiset, bit for bit.  `classes` (label voting) is not supported.
"""
import numpy as np
import torch

from buffer_amd import ops

_METHODS = ("barycenters", "voxelcenters")


def _f32(a, what):
    try:
        if isinstance(a, torch.Tensor):
            a = a.detach().cpu().numpy()
        return np.ascontiguousarray(a, dtype=np.float32)
    except Exception:
        raise RuntimeError(f"Error converting input {what} to numpy arrays of type float32")


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("grid_subsampling: no HIP device (buffer_amd has no CPU path)")
    return torch.device("cuda", torch.cuda.current_device())


def subsample_batch(points, batches, *, features=None, classes=None, sampleDl=0.1, method="barycenters", max_p=0,
                    verbose=0):
    if method not in _METHODS:                                             # wrapper.cpp:92-96 (validated, then ignored)
        raise RuntimeError('Error parsing method. Valid method names are "barycenters" and "voxelcenters" ')
    if classes is not None:
        raise RuntimeError("grid_subsampling (MI355X): label voting (classes=) is not supported")
    pts = _f32(points, "points")
    try:
        b = np.ascontiguousarray(batches.detach().cpu().numpy() if isinstance(batches, torch.Tensor) else batches,
                                 dtype=np.int32)
    except Exception:
        raise RuntimeError("Error converting input batches to numpy arrays of type int32")
    if pts.ndim != 2 or pts.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : points.shape is not (N, 3)")
    if b.ndim > 1:
        raise RuntimeError("Wrong dimensions : batches.shape is not (B,) ")
    feats = None
    if features is not None:
        feats = _f32(features, "features")
        if feats.ndim != 2:
            raise RuntimeError("Wrong dimensions : features.shape is not (N, d)")
        if feats.shape[0] != pts.shape[0]:
            raise RuntimeError("Wrong dimensions : features.shape is not (N, d)")
    dev = _device()
    try:
        res = ops.grid_subsample_batch(torch.from_numpy(pts).to(dev), b, float(sampleDl), int(max_p),
                                       features=None if feats is None else torch.from_numpy(feats).to(dev))
    except ops._lib.BufferHipError as e:
        raise RuntimeError(str(e))
    if res[0].shape[0] < 1:                                                # wrapper.cpp:266-270
        raise RuntimeError("Error")
    out = (res[0].cpu().numpy(), np.asarray(res[1], np.int32))
    if feats is not None:
        out = out + (res[2].cpu().numpy(),)
    return out


def subsample(points, *, features=None, classes=None, sampleDl=0.1, method="barycenters", verbose=0):
    pts = _f32(points, "points")
    r = subsample_batch(pts, np.array([pts.shape[0]], np.int32), features=features, classes=classes, sampleDl=sampleDl,
                        method=method, verbose=verbose)
    return r[0] if features is None else (r[0], r[2])
