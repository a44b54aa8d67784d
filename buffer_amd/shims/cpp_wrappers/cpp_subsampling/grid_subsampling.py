"""cpp_wrappers.cpp_subsampling.grid_subsampling -- drop-in for the CPython module built from
cpp_wrappers/cpp_subsampling/wrapper.cpp (:62-333 subsample_batch, :338-566 subsample).

Same call conventions: positional points (and batches), everything else keyword-only; array-likes are
coerced to C-contiguous float32 / int32; new numpy arrays are returned; failures raise RuntimeError.
Rows come back per batch element in ascending voxel-key order (the reference: libstdc++
unordered_map order) -- same multiset, bit for bit.  `classes` (majority label per voxel and label column,
grid_subsampling.cpp:97-103) rides on the feature path: every label column is one-hot encoded, the kernel's per-voxel feature
means are then the label frequencies, and the arg-max is the majority label.  Ties go to the smallest label (the reference:
whatever its unordered_map iterates first).
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
    labels, onehots, vocab = None, [], []
    if classes is not None:
        try:
            labels = np.ascontiguousarray(classes.detach().cpu().numpy() if isinstance(classes, torch.Tensor) else classes,
                                          dtype=np.int32)
        except Exception:
            raise RuntimeError("Error converting input classes to numpy arrays of type int32")
        if labels.ndim == 1:
            labels = labels[:, None]
        if labels.ndim != 2 or labels.shape[0] != pts.shape[0]:
            raise RuntimeError("Wrong dimensions : classes.shape is not (N,) or (N, d)")
        for d in range(labels.shape[1]):                                   # one-hot per label column, classes ascending
            u, inv = np.unique(labels[:, d], return_inverse=True)
            vocab.append(u)
            oh = np.zeros((labels.shape[0], len(u)), np.float32)
            oh[np.arange(labels.shape[0]), inv] = 1.0
            onehots.append(oh)
    allf = ([feats] if feats is not None else []) + onehots
    dev = _device()
    try:
        res = ops.grid_subsample_batch(torch.from_numpy(pts).to(dev), b, float(sampleDl), int(max_p),
                                       features=torch.from_numpy(np.concatenate(allf, 1)).to(dev) if allf else None)
    except ops._lib.BufferHipError as e:
        raise RuntimeError(str(e))
    if res[0].shape[0] < 1:                                                # wrapper.cpp:266-270
        raise RuntimeError("Error")
    out = (res[0].cpu().numpy(), np.asarray(res[1], np.int32))
    means = res[2].cpu().numpy() if allf else None
    col = 0
    if feats is not None:
        out = out + (means[:, :feats.shape[1]],)
        col = feats.shape[1]
    if labels is not None:
        voted = np.empty((means.shape[0], labels.shape[1]), np.int32)
        for d, u in enumerate(vocab):                                      # frequencies -> majority label (first maximum)
            voted[:, d] = u[np.argmax(means[:, col:col + len(u)], axis=1)]
            col += len(u)
        out = out + (voted,)
    return out


def subsample(points, *, features=None, classes=None, sampleDl=0.1, method="barycenters", verbose=0):
    pts = _f32(points, "points")
    r = subsample_batch(pts, np.array([pts.shape[0]], np.int32), features=features, classes=classes, sampleDl=sampleDl,
                        method=method, verbose=verbose)
    rest = r[2:]                                                           # wrapper.cpp:540-566: points[, features][, classes]
    return r[0] if not rest else (r[0],) + tuple(rest)
