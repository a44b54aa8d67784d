"""Device operators: torch tensors in HBM -> libbuffer_hip.so kernels on the current HIP stream.

PyTorch is used for device memory and stream handles only; every operator below is a hand-written
gfx950 kernel behind the C ABI of include/buffer_hip.h.  Nothing here falls back to the CPU.
"""
import ctypes as C
import numpy as np
import torch

from . import _lib
from ._lib import buf_grid_t, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.BufferHipError(f"{what}: expected a tensor in device memory (buffer_amd has no CPU path)")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _host_i32(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.int32).reshape(-1)


def _hptr(a):
    return C.c_void_p(a.ctypes.data)


class CellGrid:
    """Uniform cell grid over stacked support clouds (A2 build half: buf_grid_build)."""

    def __init__(self, supports, s_lengths, radius, cells_per_elem=0):
        L = _lib.lib()
        self.supports = _dev(supports, torch.float32, "CellGrid.supports")
        if self.supports.dim() != 2 or self.supports.shape[1] != 3:
            raise _lib.BufferHipError("Wrong dimensions : support.shape is not (N, 3)")
        self.s_lengths = _host_i32(s_lengths)
        self.ns = int(self.supports.shape[0])
        self.nb = int(self.s_lengths.shape[0])
        self.radius = float(radius)
        if cells_per_elem <= 0:
            cells_per_elem = L.buf_grid_default_cells(self.ns, self.nb)
        nbytes = L.buf_grid_ws_bytes(self.ns, self.nb, cells_per_elem)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.supports.device)
        self.g = buf_grid_t()
        check(L.buf_grid_build(C.byref(self.g), _ptr(self.supports), self.ns, _hptr(self.s_lengths), self.nb,
                               self.radius, cells_per_elem, _ptr(self.ws), nbytes, _stream()), "buf_grid_build")

    @property
    def order(self):
        """int32[ns]: global support index in cell order (a spatially coherent processing order)."""
        n = max(self.ns, 1)
        off = self.g.order - self.ws.data_ptr()
        return self.ws[off:off + 4 * n].view(torch.int32)[:self.ns]

    def query(self, queries, q_lengths, k, radius=None, q_order=None, counts=False, max_count=None):
        """-> int32[nq,k] (+ int32[nq] untruncated counts).  Rows ascending by (d2, index), padded
        with ns.  max_count: optional int32[1] device tensor that is atomically max-ed."""
        L = _lib.lib()
        queries = _dev(queries, torch.float32, "CellGrid.query")
        if queries.dim() != 2 or queries.shape[1] != 3:
            raise _lib.BufferHipError("Wrong dimensions : query.shape is not (N, 3)")
        q_lengths = _host_i32(q_lengths)
        if q_lengths.shape[0] != self.nb:
            raise _lib.BufferHipError("Wrong number of batch elements: different for queries and supports ")
        nq = int(queries.shape[0])
        out = torch.empty((nq, k), dtype=torch.int32, device=queries.device)
        cnt = torch.empty((nq,), dtype=torch.int32, device=queries.device) if counts else None
        if q_order is not None:
            q_order = _dev(q_order, torch.int32, "q_order")
        r = self.radius if radius is None else float(radius)
        check(L.buf_grid_query(C.byref(self.g), _ptr(queries), nq, _hptr(q_lengths), _ptr(q_order), r, int(k),
                               _ptr(out), _ptr(cnt), _ptr(max_count), _stream()), "buf_grid_query")
        return (out, cnt) if counts else out


def radius_neighbors(queries, supports, q_lengths, s_lengths, radius, k=None):
    """batch_query semantics on device: int32[nq, max_count] when k is None (two passes: count, fill)."""
    grid = CellGrid(supports, s_lengths, radius)
    if k is None:
        mc = torch.zeros(1, dtype=torch.int32, device=grid.supports.device)
        grid.query(queries, q_lengths, 0, max_count=mc)
        k = int(mc.item())
    return grid.query(queries, q_lengths, k)


def grid_subsample_batch(points, lengths, dl, max_p=0, max_cells=0):
    """subsample_batch on device -> (f32[M,3] device tensor, int32[nb] numpy lengths).
    Rows per element in ascending voxel-key order."""
    L = _lib.lib()
    points = _dev(points, torch.float32, "grid_subsample_batch.points")
    if points.dim() != 2 or points.shape[1] != 3:
        raise _lib.BufferHipError("Wrong dimensions : points.shape is not (N, 3)")
    lengths = _host_i32(lengths)
    n, nb = int(points.shape[0]), int(lengths.shape[0])
    if max_cells <= 0:
        max_cells = max(1 << 22, 64 * n)
    nbytes = L.buf_grid_subsample_ws_bytes(n, nb, max_cells)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=points.device)
    out = torch.empty((max(n, 1), 3), dtype=torch.float32, device=points.device)
    out_b = np.zeros(nb, np.int32)
    m = C.c_int(0)
    check(L.buf_grid_subsample_batch(_ptr(points), n, _hptr(lengths), nb, float(dl), int(max_p), _ptr(out),
                                     _hptr(out_b), C.byref(m), max_cells, _ptr(ws), nbytes, _stream()),
          "buf_grid_subsample_batch")
    return out[:m.value], out_b
