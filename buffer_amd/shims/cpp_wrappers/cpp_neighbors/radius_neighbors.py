"""cpp_wrappers.cpp_neighbors.radius_neighbors -- drop-in for the CPython module built from
cpp_wrappers/cpp_neighbors/wrapper.cpp:58-238 (batch_query -> batch_nanoflann_neighbors).

Returns a new int32 ndarray [Nq, max_count]: rows ascending by squared distance (ties by support
index), padded with the total support count; RuntimeError on bad shapes or when no neighbour exists."""
import numpy as np
import torch

from buffer_amd import ops


def _arr(a, dtype, msg):
    try:
        if isinstance(a, torch.Tensor):
            a = a.detach().cpu().numpy()
        return np.ascontiguousarray(a, dtype=dtype)
    except Exception:
        raise RuntimeError(msg)


def batch_query(queries, supports, q_batches, s_batches, *, radius=0.1):
    q = _arr(queries, np.float32, "Error converting query points to numpy arrays of type float32")
    s = _arr(supports, np.float32, "Error converting support points to numpy arrays of type float32")
    qb = _arr(q_batches, np.int32, "Error converting query batches to numpy arrays of type int32")
    sb = _arr(s_batches, np.int32, "Error converting support batches to numpy arrays of type int32")
    if q.ndim != 2 or q.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : query.shape is not (N, 3)")
    if s.ndim != 2 or s.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : support.shape is not (N, 3)")
    if qb.ndim > 1:
        raise RuntimeError("Wrong dimensions : queries_batches.shape is not (B,) ")
    if sb.ndim > 1:
        raise RuntimeError("Wrong dimensions : supports_batches.shape is not (B,) ")
    if qb.shape[0] != sb.shape[0]:
        raise RuntimeError("Wrong number of batch elements: different for queries and supports ")
    if not torch.cuda.is_available():
        raise RuntimeError("radius_neighbors: no HIP device (buffer_amd has no CPU path)")
    dev = torch.device("cuda", torch.cuda.current_device())
    try:
        out = ops.radius_neighbors(torch.from_numpy(q).to(dev), torch.from_numpy(s).to(dev), qb, sb, float(radius))
    except ops._lib.BufferHipError as e:
        raise RuntimeError(str(e))
    if out.numel() < 1:                                                     # wrapper.cpp:201-205
        raise RuntimeError("Error")
    return out.cpu().numpy()
