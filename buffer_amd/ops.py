"""Device operators: torch tensors in HBM -> libbuffer_hip.so kernels on the current HIP stream.

PyTorch is used for device memory and stream handles only; every operator below is a hand-written
gfx950 kernel behind the C ABI of include/buffer_hip.h.  Nothing here falls back to the CPU.
"""
import ctypes as C
import time
import os
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
            f = int(os.environ.get('BUF_GRID_CELLS_PER_POINT', 0))          # development switch: table cells per support point (library default 16)
            if f > 0:
                per = (self.ns + self.nb - 1) // max(self.nb, 1)
                cells_per_elem = min(cells_per_elem, f * per + 65536)
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
        todo = torch.empty((max(nq, 1),), dtype=torch.int32, device=queries.device) if (k > 0 or q_order is not None) else None
        check(L.buf_grid_query(C.byref(self.g), _ptr(queries), nq, _hptr(q_lengths), _ptr(q_order), r, int(k),
                               _ptr(out), _ptr(cnt), _ptr(max_count), _ptr(todo), _stream()), "buf_grid_query")
        return (out, cnt) if counts else out


def radius_neighbors(queries, supports, q_lengths, s_lengths, radius, k=None):
    """batch_query semantics on device: int32[nq, max_count] when k is None (two passes: count, fill)."""
    grid = CellGrid(supports, s_lengths, radius)
    if k is None:
        mc = torch.zeros(1, dtype=torch.int32, device=grid.supports.device)
        grid.query(queries, q_lengths, 0, max_count=mc)
        k = int(mc.item())
    return grid.query(queries, q_lengths, k)


HOST_WAIT_S = [0.0]        # diagnostics: seconds callers spent blocked in calls that end with a host round trip (subsample row counts)


def grid_subsample_batch(points, lengths, dl, max_p=0, max_cells=0, features=None):
    """subsample_batch on device -> (f32[M,3] device tensor, int32[nb] numpy lengths[, f32[M,fd] feature means]).
    Rows per element in ascending voxel-key order."""
    L = _lib.lib()
    points = _dev(points, torch.float32, "grid_subsample_batch.points")
    if points.dim() != 2 or points.shape[1] != 3:
        raise _lib.BufferHipError("Wrong dimensions : points.shape is not (N, 3)")
    lengths = _host_i32(lengths)
    n, nb = int(points.shape[0]), int(lengths.shape[0])
    if max_cells <= 0:
        # bucket table of the counting sort: the result does not depend on it (a bucket holds consecutive voxel keys, ranked by key),
        # the memset + two scan passes over it do cost (64 n cells were 180 MB per call at 32 pairs: most of A1's time)
        max_cells = max(1 << 16, int(os.environ.get('BUF_VOX_CELLS_PER_POINT', 4)) * n, 2 * nb)
    fd = 0
    out_f = None
    if features is not None:
        features = _dev(features, torch.float32, "grid_subsample_batch.features")
        fd = int(features.shape[1])
        out_f = torch.empty((max(n, 1), fd), dtype=torch.float32, device=points.device)
    nbytes = L.buf_grid_subsample_ws_bytes(n, nb, max_cells, fd)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=points.device)
    out = torch.empty((max(n, 1), 3), dtype=torch.float32, device=points.device)
    out_b = np.zeros(nb, np.int32)
    m = C.c_int(0)
    t0 = time.perf_counter()
    check(L.buf_grid_subsample_batch(_ptr(points), n, _hptr(lengths), nb, float(dl), int(max_p), _ptr(features), fd,
                                     _ptr(out), _ptr(out_f), _hptr(out_b), C.byref(m), max_cells, _ptr(ws), nbytes,
                                     _stream()), "buf_grid_subsample_batch")
    HOST_WAIT_S[0] += time.perf_counter() - t0          # the call ends with a stream synchronise (rows per element go to the host)
    if features is not None:
        return out[:m.value], out_b, out_f[:m.value]
    return out[:m.value], out_b


# ----------------------------------------------------------------------------- pointnet2 / knn / svd
def furthest_point_sample(xyz, npoint):
    """xyz f32[B,N,3] -> int32[B,npoint]."""
    L = _lib.lib()
    xyz = _dev(xyz, torch.float32, "furthest_point_sample")
    b, n, _ = xyz.shape
    out = torch.empty((b, npoint), dtype=torch.int32, device=xyz.device)
    nbytes = L.buf_fps_ws_bytes(b, n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=xyz.device)
    check(L.buf_fps(_ptr(xyz), b, n, int(npoint), _ptr(out), _ptr(ws), nbytes, _stream()), "buf_fps")
    return out


def furthest_point_sample_ragged(xyz, lengths, npoint):
    """clouds of different size stacked in xyz f32[sum(n),3] -> int32[b,npoint] (indices local to each cloud);
    all clouds are sampled concurrently, one workgroup each."""
    L = _lib.lib()
    xyz = _dev(xyz, torch.float32, "furthest_point_sample_ragged")
    lengths = _host_i32(lengths)
    b = int(lengths.shape[0])
    out = torch.empty((b, npoint), dtype=torch.int32, device=xyz.device)
    nbytes = L.buf_fps_ws_bytes(1, int(xyz.shape[0]))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=xyz.device)
    check(L.buf_fps_ragged(_ptr(xyz), _hptr(lengths), b, int(npoint), _ptr(out), _ptr(ws), nbytes, _stream()),
          "buf_fps_ragged")
    return out


def gather_operation(feat, idx):
    """feat f32[B,C,N], idx int32[B,M] -> f32[B,C,M]."""
    L = _lib.lib()
    feat, idx = _dev(feat, torch.float32, "gather_operation"), _dev(idx, torch.int32, "gather_operation")
    b, c, n = feat.shape
    m = idx.shape[1]
    out = torch.empty((b, c, m), dtype=torch.float32, device=feat.device)
    check(L.buf_gather(_ptr(feat), _ptr(idx), b, c, n, m, _ptr(out), _stream()), "buf_gather")
    return out


def grouping_operation(feat, idx):
    """feat f32[B,C,N], idx int32[B,M,S] -> f32[B,C,M,S]."""
    L = _lib.lib()
    feat, idx = _dev(feat, torch.float32, "grouping_operation"), _dev(idx, torch.int32, "grouping_operation")
    b, c, n = feat.shape
    _, m, s = idx.shape
    out = torch.empty((b, c, m, s), dtype=torch.float32, device=feat.device)
    check(L.buf_group(_ptr(feat), _ptr(idx), b, c, n, m, s, _ptr(out), _stream()), "buf_group")
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    """xyz f32[B,N,3], new_xyz f32[B,M,3] -> int32[B,M,nsample]."""
    L = _lib.lib()
    xyz, new_xyz = _dev(xyz, torch.float32, "ball_query"), _dev(new_xyz, torch.float32, "ball_query")
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    out = torch.zeros((b, m, nsample), dtype=torch.int32, device=xyz.device)
    check(L.buf_ball_query(_ptr(xyz), _ptr(new_xyz), b, n, m, float(radius), int(nsample), _ptr(out), _stream()),
          "buf_ball_query")
    return out


def three_nn(unknown, known):
    L = _lib.lib()
    unknown, known = _dev(unknown, torch.float32, "three_nn"), _dev(known, torch.float32, "three_nn")
    b, n, _ = unknown.shape
    m = known.shape[1]
    dist = torch.empty((b, n, 3), dtype=torch.float32, device=unknown.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=unknown.device)
    check(L.buf_three_nn(_ptr(unknown), _ptr(known), b, n, m, _ptr(dist), _ptr(idx), _stream()), "buf_three_nn")
    return dist, idx


def select_patches(pts, kpts, radius, nsample, out=None):
    """pts f32[N,3] (already permuted), kpts f32[P,3] -> f32[P,nsample,3] (optionally into a contiguous `out` view)."""
    L = _lib.lib()
    pts, kpts = _dev(pts, torch.float32, "select_patches"), _dev(kpts, torch.float32, "select_patches")
    if out is None:
        out = torch.empty((kpts.shape[0], nsample, 3), dtype=torch.float32, device=pts.device)
    elif not out.is_contiguous() or tuple(out.shape) != (kpts.shape[0], nsample, 3):
        raise _lib.BufferHipError("select_patches: out must be a contiguous [P,nsample,3] tensor")
    check(L.buf_select_patches(_ptr(pts), _ptr(kpts), pts.shape[0], kpts.shape[0], float(radius), int(nsample),
                               _ptr(out), _stream()), "buf_select_patches")
    return out


_MASK64 = (1 << 64) - 1


def perm_key(seed, cloud):
    """64-bit key of the keyed permutation of cloud `cloud` (0 = src, 1 = tgt) of the pair registered with `seed`."""
    x = ((int(seed) * 2 + int(cloud)) * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03) & _MASK64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _MASK64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _MASK64
    return x ^ (x >> 31)


def permute_clouds(clouds, keys):
    """clouds: list of f32[n_c,3] device tensors -> (f32[sum n_c,3] stacked and shuffled cloud by cloud, int32[nc] lengths).
    One launch; the permutation of a cloud depends on its key and length only (patch_embedder.py:97-98's randperm)."""
    L = _lib.lib()
    clouds = [_dev(c, torch.float32, "permute_clouds") for c in clouds]
    lens = np.array([c.shape[0] for c in clouds], np.int32)
    out = torch.empty((int(lens.sum()), 3), dtype=torch.float32, device=clouds[0].device)
    nc = len(clouds)
    ptrs = (C.c_void_p * nc)(*[c.data_ptr() for c in clouds])
    ks = (C.c_ulonglong * nc)(*[int(k) & _MASK64 for k in keys])
    check(L.buf_permute_clouds(ptrs, _hptr(lens), ks, nc, _ptr(out), _stream()), "buf_permute_clouds")
    return out, lens


def select_patches_batched(sup, lengths, kpts, m, radius, nsample, out=None):
    """MiniSpinNet.select_patches for every cloud of a step in one launch: sup f32[sum n_c,3] (the permuted support
    clouds stacked), kpts f32[nc*m,3] (m keypoints per cloud) -> f32[nc*m, nsample, 3]."""
    L = _lib.lib()
    sup = _dev(sup, torch.float32, "select_patches_batched.sup")
    kpts = _dev(kpts, torch.float32, "select_patches_batched.kpts")
    lengths = _host_i32(lengths)
    nc = int(lengths.shape[0])
    if kpts.shape[0] != nc * m:
        raise _lib.BufferHipError(f"select_patches_batched: {kpts.shape[0]} keypoints for {nc} clouds x {m}")
    if out is None:
        out = torch.empty((nc * m, nsample, 3), dtype=torch.float32, device=sup.device)
    nbytes = L.buf_select_patches_batched_ws_bytes(int(sup.shape[0]), nc)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=sup.device)
    check(L.buf_select_patches_batched(_ptr(sup), _hptr(lengths), nc, _ptr(kpts), int(m), float(radius), int(nsample), _ptr(out),
                                       _ptr(ws), nbytes, _stream()), "buf_select_patches_batched")
    return out


def compact_greater(x, threshold):
    """ascending int32 indices i with x[i] > threshold (x f32[n] or [n,1]); one host read-back for the count."""
    L = _lib.lib()
    x = _dev(x, torch.float32, "compact_greater").reshape(-1)
    n = int(x.shape[0])
    idx = torch.empty((max(n, 1),), dtype=torch.int32, device=x.device)
    cnt = torch.zeros((1,), dtype=torch.int32, device=x.device)
    nbytes = L.buf_compact_ws_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    check(L.buf_compact_greater(_ptr(x), 1, n, float(threshold), _ptr(idx), _ptr(cnt), _ptr(ws), nbytes, _stream()),
          "buf_compact_greater")
    return idx[:int(cnt.item())]


def knn(ref, query, k):
    """ref f32[B,N,D], query f32[B,Q,D] -> (dist f32[B,Q,k], idx int64[B,Q,k])."""
    L = _lib.lib()
    ref, query = _dev(ref, torch.float32, "knn"), _dev(query, torch.float32, "knn")
    b, n, d = ref.shape
    q = query.shape[1]
    dist = torch.empty((b, q, k), dtype=torch.float32, device=ref.device)
    idx = torch.empty((b, q, k), dtype=torch.int64, device=ref.device)
    nbytes = L.buf_knn1_ws_bytes(b, n, q) if (int(k) == 1 and d == 32 and n > 0) else L.buf_knn_ws_bytes(b, q, int(k))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=ref.device)
    check(L.buf_knn(_ptr(ref), _ptr(query), b, n, q, d, int(k), _ptr(dist), _ptr(idx), _ptr(ws), nbytes, _stream()),
          "buf_knn")
    return dist, idx


def svd3x3(a):
    """a f32[B,3,3] -> (U, S, V), a = U diag(S) V^T."""
    L = _lib.lib()
    a = _dev(a, torch.float32, "svd3x3")
    n = a.shape[0]
    u, v = torch.empty_like(a), torch.empty_like(a)
    s = torch.empty((n, 3), dtype=torch.float32, device=a.device)
    check(L.buf_svd3x3_batched(_ptr(a), n, _ptr(u), _ptr(s), _ptr(v), _stream()), "buf_svd3x3_batched")
    return u, s, v


# ----------------------------------------------------------------------------- VN blocks
class VnLayer:
    """Device copy of one VNLinearLeakyReLU (map_to_feat / map_to_dir / VN batch-norm folded)."""

    def __init__(self, W, prefix, device, slope=0.2, linear_only=False):
        if linear_only:
            self.wf = torch.as_tensor(W[prefix + '.weight'], dtype=torch.float32, device=device).contiguous()
            self.wd = self.bsc = self.bsh = None
        else:
            self.wf = torch.as_tensor(W[prefix + '.map_to_feat.weight'], dtype=torch.float32, device=device).contiguous()
            self.wd = torch.as_tensor(W[prefix + '.map_to_dir.weight'], dtype=torch.float32, device=device).contiguous()
            self.bsc = self.bsh = None
            if self.wf.shape[0] != 1:                                  # vn_layers.py:123
                g = np.asarray(W[prefix + '.batchnorm.bn.weight'], np.float64)
                b = np.asarray(W[prefix + '.batchnorm.bn.bias'], np.float64)
                m = np.asarray(W[prefix + '.batchnorm.bn.running_mean'], np.float64)
                v = np.asarray(W[prefix + '.batchnorm.bn.running_var'], np.float64)
                sc = g / np.sqrt(v + 1e-5)
                self.bsc = torch.tensor(sc, dtype=torch.float32, device=device)
                self.bsh = torch.tensor(b - m * sc, dtype=torch.float32, device=device)
        self.cout, self.cin = int(self.wf.shape[0]), int(self.wf.shape[1])
        self.slope = float(slope)


PF_BYTES = [0.0]           # diagnostics (bench.py): bytes of the hoisted form's PF table written + re-read by the calls so far (48 * cout * ns each)


def vn_gather_block(layer, q_pts, s_pts, feats, idx, mode, scale=1.0):
    """VNNBlock / conv half of VNNResnetBlock -> f32[nq, 3*cout]."""
    L = _lib.lib()
    nq, k = idx.shape
    ns = s_pts.shape[0]
    cin = feats.shape[1] // 3
    out = torch.empty((nq, 3 * layer.cout), dtype=torch.float32, device=feats.device)
    if int(mode) == 1 and not os.environ.get('BUF_VN_GATHER_DIRECT'):
        # the channel contraction once per support point instead of once per neighbour slot (csrc/vn.hip, round 4)
        wsb = L.buf_vn_gather_pre_ws_bytes(ns, layer.cout)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=feats.device)
        PF_BYTES[0] += 48.0 * layer.cout * ns               # PF[j] = [Wf f_j | Wd f_j]: 2 * cout vectors of 12 B per support row, written once and read back
        check(L.buf_vn_gather_block_pre(_ptr(q_pts), _ptr(s_pts), _ptr(feats), _ptr(idx), nq, ns, k, cin, layer.cout, float(scale),
                                        _ptr(layer.wf), _ptr(layer.wd), _ptr(layer.bsc), _ptr(layer.bsh), layer.slope, _ptr(out),
                                        _ptr(ws), wsb, _stream()), "buf_vn_gather_block_pre")
        return out
    check(L.buf_vn_gather_block(_ptr(q_pts), _ptr(s_pts), _ptr(feats), _ptr(idx), nq, ns, k, cin, layer.cout, int(mode),
                                float(scale), _ptr(layer.wf), _ptr(layer.wd), _ptr(layer.bsc), _ptr(layer.bsh),
                                layer.slope, _ptr(out), _stream()), "buf_vn_gather_block")
    return out


def vn_pointwise(layer, b, a=None, ind_a=None, residual=None):
    """VN layer on concat(a[ind_a[:,0]], b) (+ residual) -> f32[n, 3*cout]."""
    L = _lib.lib()
    n = b.shape[0] if b is not None else (ind_a.shape[0] if ind_a is not None else a.shape[0])
    ca = a.shape[1] // 3 if a is not None else 0
    cb = b.shape[1] // 3 if b is not None else 0
    dev = (b if b is not None else a).device
    out = torch.empty((n, 3 * layer.cout), dtype=torch.float32, device=dev)
    stride = ind_a.shape[1] if ind_a is not None else 0
    check(L.buf_vn_pointwise(_ptr(a), _ptr(ind_a), stride, a.shape[0] if a is not None else 0, ca, _ptr(b), cb, n,
                             layer.cout, _ptr(layer.wf), _ptr(layer.wd), _ptr(layer.bsc), _ptr(layer.bsh), layer.slope,
                             _ptr(residual), _ptr(out), _stream()), "buf_vn_pointwise")
    return out


def gather_max(feats, idx):
    L = _lib.lib()
    nq, k = idx.shape
    out = torch.empty((nq, feats.shape[1]), dtype=torch.float32, device=feats.device)
    check(L.buf_gather_max(_ptr(feats), _ptr(idx), nq, feats.shape[0], k, feats.shape[1], _ptr(out), _stream()),
          "buf_gather_max")
    return out


def vn_std(x, z):
    L = _lib.lib()
    n, c = x.shape[0], x.shape[1] // 3
    out = torch.empty((n, 3 * c), dtype=torch.float32, device=x.device)
    check(L.buf_vn_std(_ptr(x), _ptr(z), n, c, _ptr(out), _stream()), "buf_vn_std")
    return out


# ----------------------------------------------------------------------------- patch voxelisation
def patch_voxelize(patches, axis, des_r, centres, azi_cs, voxel_r, nsample, mlp_w, mlp_b, bn_scale, bn_shift,
                   azi_n=20, want_patches=False):
    """patches f32[P,S,3] -> (x f32[P,16,ncentres], R f32[P,3,3], rand_axis f32[P,3], patches_norm|None)."""
    L = _lib.lib()
    P, S, _ = patches.shape
    nc = centres.shape[0]
    dev = patches.device
    x = torch.empty((P, 16, nc), dtype=torch.float32, device=dev)
    R = torch.empty((P, 3, 3), dtype=torch.float32, device=dev)
    ra = torch.empty((P, 3), dtype=torch.float32, device=dev)
    pn = torch.empty((P, S, 3), dtype=torch.float32, device=dev) if want_patches else None
    hw = [np.ascontiguousarray(a, dtype=np.float32) for a in (mlp_w, mlp_b, bn_scale, bn_shift)]
    wsb = L.buf_patch_voxelize_ws_bytes(nc)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    check(L.buf_patch_voxelize(_ptr(patches), _ptr(axis), P, S, float(des_r), _ptr(centres), nc, int(azi_n),
                               _ptr(azi_cs), float(voxel_r), int(nsample), _hptr(hw[0]), _hptr(hw[1]), _hptr(hw[2]),
                               _hptr(hw[3]), _ptr(x), _ptr(R), _ptr(ra), _ptr(pn), _ptr(ws), wsb, _stream()), "buf_patch_voxelize")
    return x, R, ra, pn


def row_linear(x, w, b, activation=None):
    """Conv1d(kernel 1) of the score heads: x f32[n,cin] @ w[cout,cin].T + b, then None | 'sigmoid' | 'softplus'."""
    L = _lib.lib()
    x = _dev(x, torch.float32, "row_linear")
    n, cin = int(x.shape[0]), int(x.shape[1])
    cout = int(w.shape[0])
    out = torch.empty((n, cout), dtype=torch.float32, device=x.device)
    act = {None: 0, 'sigmoid': 1, 'softplus': 2}[activation]
    check(L.buf_row_linear(_ptr(x), n, cin, cout, _ptr(w.contiguous()), _ptr(b.contiguous()), act, _ptr(out), _stream()),
          "buf_row_linear")
    return out


def score_head(x, seg_lengths, vn1, vn2, lin, w, b, final, eps=1e-5):
    """One score head (VNStdFeature + Conv1d / InstanceNorm1d stack, point_learner.py:128-136,163-171) in 7 launches:
    x f32[n,30], seg_lengths int[nseg] (host: rows per pair), vn1 / vn2 / lin VnLayer, w / b the three Conv1d layers -> f32[n,1]."""
    L = _lib.lib()
    x = _dev(x, torch.float32, "score_head")
    n = int(x.shape[0])
    lens = _host_i32(seg_lengths)
    nseg = int(lens.shape[0])
    out = torch.empty((n, 1), dtype=torch.float32, device=x.device)
    nbytes = L.buf_score_head_ws_bytes(n, nseg)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    act = {None: 0, 'sigmoid': 1, 'softplus': 2}[final]
    check(L.buf_score_head(_ptr(x), n, _hptr(lens), nseg, _ptr(vn1.wf), _ptr(vn1.wd), _ptr(vn1.bsc), _ptr(vn1.bsh), vn1.slope,
                           _ptr(vn2.wf), _ptr(vn2.wd), _ptr(vn2.bsc), _ptr(vn2.bsh), vn2.slope, _ptr(lin.wf),
                           _ptr(w[0]), _ptr(b[0]), _ptr(w[1]), _ptr(b[1]), int(w[1].shape[0]), _ptr(w[2]), _ptr(b[2]), act, float(eps),
                           _ptr(out), _ptr(ws), nbytes, _stream()), "buf_score_head")
    return out


def score_head_supported(x_width, vn1, vn2, lin, w):
    """the fused head exists for the released widths only (10 -> 10 -> 5 -> 3 vector channels, Conv1d 30 -> 20 -> c <= 10 -> 1)"""
    return (x_width == 30 and tuple(vn1.wf.shape) == (10, 10) and tuple(vn2.wf.shape) == (5, 10) and tuple(lin.wf.shape) == (3, 5)
            and vn1.wd is not None and vn2.wd is not None and tuple(w[0].shape) == (20, 30) and w[1].shape[1] == 20 and w[1].shape[0] <= 10
            and tuple(w[2].shape) == (1, int(w[1].shape[0])))


def segment_instance_norm(x, seg_lengths, eps=1e-5):
    """InstanceNorm1d (biased variance) over contiguous row segments: x f32[n,c], seg_lengths int[nseg] (host) -> f32[n,c]."""
    L = _lib.lib()
    x = _dev(x, torch.float32, "segment_instance_norm")
    n, c = int(x.shape[0]), int(x.shape[1])
    lens = _host_i32(seg_lengths)
    nseg = int(lens.shape[0])
    out = torch.empty_like(x)
    nbytes = L.buf_segment_instance_norm_ws_bytes(nseg, c)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    check(L.buf_segment_instance_norm(_ptr(x), n, c, _hptr(lens), nseg, float(eps), _ptr(out), _ptr(ws), nbytes, _stream()),
          "buf_segment_instance_norm")
    return out


# ----------------------------------------------------------------------------- pose recovery
def hypotheses_score(ind, ss_kpts, tt_kpts, ss_R, tt_R, azi_n=20, inlier_th=1 / 3):
    L = _lib.lib()
    m = ind.shape[0]
    dev = ind.device
    R = torch.empty((m, 3, 3), dtype=torch.float32, device=dev)
    t = torch.empty((m, 3), dtype=torch.float32, device=dev)
    num = torch.empty((m,), dtype=torch.int32, device=dev)
    best = torch.zeros((1,), dtype=torch.int32, device=dev)
    mask = torch.zeros((m,), dtype=torch.uint8, device=dev)
    check(L.buf_hypotheses_score(_ptr(ind.contiguous()), _ptr(ss_kpts.contiguous()), _ptr(tt_kpts.contiguous()),
                                 _ptr(ss_R.contiguous()), _ptr(tt_R.contiguous()), m, int(azi_n), float(inlier_th),
                                 _ptr(R), _ptr(t), _ptr(num), _ptr(best), _ptr(mask), _stream()),
          "buf_hypotheses_score")
    return R, t, num, best, mask


def ransac_kabsch(src, tgt, corr, nhyp=4096, seed=0, max_dist=0.10, edge_similarity=0.8):
    """-> (T f32[4,4], info int32[2])."""
    L = _lib.lib()
    dev = src.device
    corr = _dev(corr, torch.int32, "ransac_kabsch")
    T = torch.empty((4, 4), dtype=torch.float32, device=dev)
    info = torch.zeros((2,), dtype=torch.int32, device=dev)
    nbytes = L.buf_ransac_ws_bytes(int(nhyp))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(L.buf_ransac_kabsch(_ptr(src.contiguous()), _ptr(tgt.contiguous()), _ptr(corr), corr.shape[0], int(nhyp),
                              int(seed), float(max_dist), float(edge_similarity), _ptr(T), _ptr(info), _ptr(ws), nbytes,
                              _stream()), "buf_ransac_kabsch")
    return T, info


def ransac_kabsch_masked(src, tgt, mask, nhyp=4096, seed=0, max_dist=0.10, edge_similarity=0.8):
    """ransac_kabsch on the correspondences with mask != 0 (uint8[m]); nothing returns to the host.
    -> (T f32[4,4], info int32[2])."""
    L = _lib.lib()
    dev = src.device
    mask = _dev(mask, torch.uint8, "ransac_kabsch_masked")
    m = int(mask.shape[0])
    T = torch.empty((4, 4), dtype=torch.float32, device=dev)
    info = torch.zeros((2,), dtype=torch.int32, device=dev)
    nbytes = L.buf_ransac_masked_ws_bytes(m, int(nhyp))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(L.buf_ransac_kabsch_masked(_ptr(src.contiguous()), _ptr(tgt.contiguous()), _ptr(mask), m, int(nhyp), int(seed),
                                     float(max_dist), float(edge_similarity), _ptr(T), _ptr(info), _ptr(ws), nbytes, _stream()),
          "buf_ransac_kabsch_masked")
    return T, info


def post_refine(T_init, src, tgt, thr=0.10, iters=20):
    L = _lib.lib()
    dev = src.device
    T_init = _dev(T_init, torch.float32, "post_refine").reshape(4, 4)
    T = torch.empty((4, 4), dtype=torch.float32, device=dev)
    info = torch.zeros((2,), dtype=torch.int32, device=dev)
    check(L.buf_post_refine(_ptr(T_init), _ptr(src.contiguous()), _ptr(tgt.contiguous()), src.shape[0], float(thr),
                            int(iters), _ptr(T), _ptr(info), _stream()), "buf_post_refine")
    return T, info


def recover_poses_batched(ind, ss_kpts, tt_kpts, ss_R, tt_R, seg_lengths, seeds, cfg):
    """hypotheses + scoring + RANSAC + refinement of every pair of a step in one set of launches: the matches of the pairs are
    stacked, pair p owns seg_lengths[p] consecutive rows -> poses f32[nb,4,4] (identity for pairs with fewer than 3 matches)."""
    L = _lib.lib()
    seg = _host_i32(seg_lengths)
    nb, M = int(seg.shape[0]), int(seg.sum())
    dev = ss_kpts.device
    poses = torch.empty((nb, 4, 4), dtype=torch.float32, device=dev)
    if nb == 0:
        return poses
    nbytes = L.buf_recover_poses_ws_bytes(M, nb, int(cfg.ransac_hypotheses))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    sd = (C.c_ulonglong * nb)(*[int(x) & _MASK64 for x in seeds])
    check(L.buf_recover_poses_batched(_ptr(ind.contiguous()), _ptr(ss_kpts.contiguous()), _ptr(tt_kpts.contiguous()),
                                      _ptr(ss_R.contiguous()), _ptr(tt_R.contiguous()), _hptr(seg), nb, sd, int(cfg.azi_n),
                                      float(cfg.inlier_th), int(cfg.ransac_hypotheses), float(cfg.dist_th), float(cfg.similar_th),
                                      float(cfg.refine_threshold), 20 if cfg.pose_refine else 0, _ptr(poses), _ptr(ws), nbytes,
                                      _stream()), "buf_recover_poses_batched")
    return poses


# ----------------------------------------------------------------------------- fused descriptor CNN
def mfma_tile_weights(wt, lk_major=False):
    """[K, Cout] (K, Cout multiples of 16) -> the B-operand tiling of the fused MFMA kernels: blocks [K/16][Cout/16] of
    256 floats laid out [lk][li][p], so that a lane's four k-steps are one 16-byte load.  K-row of (g, p, lk):
    16g + 4p + lk (k-step p takes channels 4p..4p+3: csrc/convnet.hip), or with lk_major 16g + 4lk + p (k-step p takes
    channel 4lk + p of every lane quarter, the order of csrc/costnet.hip's 16-byte A-fragment loads)."""
    K, cout = wt.shape
    assert K % 16 == 0 and cout % 16 == 0
    t = wt.reshape(K // 16, 4, 4, cout // 16, 16)            # [g, p, lk, n, li]  or  [g, lk, p, n, li]
    perm = (0, 3, 1, 4, 2) if lk_major else (0, 3, 2, 4, 1)
    return np.ascontiguousarray(np.transpose(t, perm), dtype=np.float32).reshape(-1)              # [g, n, lk, li, p]


def winograd_tile_weights(w, ng=None, blocks=4):
    """[Cout, Cin, 3, 3] -> the F(2x2, 3x3) filter transform U = G g G^T (fp64, rounded once to fp32) in the A-operand tiling
    of csrc/convnet_wg.hip: [N-group][i][k-step][n2][lk][li][j] = U[i][j][16 (NG g + n2) + li][4 ks + lk], N-groups of NG = 2
    N-tiles for 64 / 128 output channels and 1 otherwise.  16 * Cout * Cin floats (buf_winograd_tile_weights is the same function
    on the C side).  ng / blocks: the general form (buf_winograd_tile_filters); blocks = 5 appends g[1] G^T = U_1 - U_2."""
    cout, cin = w.shape[0], w.shape[1]
    assert cin % 4 == 0 and cout % 16 == 0
    if ng is None:
        ng = _lib.lib().buf_winograd_group(cin, cout)              # N-tiles per wavefront for these widths: the kernel's rule
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    G5 = np.concatenate([G, [[0, 1, 0]]])                                              # row 4: the filter's middle row as it is
    U = np.einsum('ia,ocab,jb->ijoc', G5, np.asarray(w, np.float64), G)                # [i, j, Cout, Cin]
    U = U[:blocks].reshape(blocks, 4, cout // (16 * ng), ng, 16, cin // 4, 4)          # [i, j, g, n2, li, ks, lk]
    return np.ascontiguousarray(np.transpose(U, (2, 0, 5, 3, 6, 4, 1)), dtype=np.float32).reshape(-1)


class CylindricalNet:
    """Device weights of Cylindrical_Net for csrc/convnet_wg.hip: per layer U = G g G^T in the kernel's tiling, biases."""

    def __init__(self, layers, device):
        """layers: list of 8 (w [Cout,Cin,3,3] np.float32 with BN folded, b [Cout], relu)"""
        self.wt, self.bias, self.cin, self.cout, self.relu = [], [], [], [], []
        self.entry = "buf_cylindrical_net_wg"
        for w, b, relu in layers:
            cout, cin = w.shape[0], w.shape[1]
            self.wt.append(torch.from_numpy(winograd_tile_weights(w)).to(device))
            self.bias.append(torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)).to(device))
            self.cin.append(cin); self.cout.append(cout); self.relu.append(1 if relu else 0)
        n = len(layers)
        self._wp = (C.c_void_p * n)(*[t.data_ptr() for t in self.wt])
        self._bp = (C.c_void_p * n)(*[t.data_ptr() for t in self.bias])
        self._ci = (C.c_int * n)(*self.cin)
        self._co = (C.c_int * n)(*self.cout)
        self._re = (C.c_int * n)(*self.relu)

    def __call__(self, x):
        """x f32[P,16,420] (or [P,48,140]) -> f32[P,32,7,20]"""
        L = _lib.lib()
        x = x.contiguous()
        P = x.shape[0]
        y = torch.empty((P, self.cout[-1], 7, 20), dtype=torch.float32, device=x.device)
        check(L.buf_cylindrical_net_wg(_ptr(x), P, self._wp, self._bp, self._ci, self._co, self._re, _ptr(y), _stream()), self.entry)
        return y


def split_tile_filters(w):
    """[Cout, Cin, 3, 3] fp32 -> the two f16 planes of csrc/convnet_h3.hip (hi = f16(w), lo' = f16((w - hi) 2^11)) in the kernel's
    tiling, as uint16 bit patterns (buf_split_tile_filters; host only)."""
    L = _lib.lib()
    w = np.ascontiguousarray(w, dtype=np.float32)
    cout, cin = w.shape[0], w.shape[1]
    out = np.empty(L.buf_split_filter_count(cout, cin), dtype=np.uint16)
    check(L.buf_split_tile_filters(w.ctypes.data, cout, cin, out.ctypes.data), "buf_split_tile_filters")
    return out


class CylindricalNetSplit:
    """Cylindrical_Net on the f16 matrix pipe with fp32-equivalent arithmetic (csrc/convnet_h3.hip, "split-f16"): opt-in next to
    CylindricalNet.  Same call, same [P,32,7,20] fp32 result to fp32 round-off.

    Safe by construction (round 6, csrc/split_safe.hip): the kernel flags every patch whose input or hidden activation leaves the f16
    range (|v| >= 65504, NaN included) and the fp32 kernel of CylindricalNet re-runs exactly those patches in the same stream, masked by
    the flags -- no host round trip, no exception; `last_flags` (int32[P], device) marks the patches of the last call that took the fp32
    kernel.  For widths the fp32 kernel is not built for (the released network's are) the old contract stays: `check_range()` raises if
    the status word was set."""

    def __init__(self, layers, device):
        self.wt, self.bias, self.cin, self.cout, self.relu = [], [], [], [], []
        self.entry = "buf_cylindrical_net_split"
        for w, b, relu in layers:
            cout, cin = w.shape[0], w.shape[1]
            self.wt.append(torch.from_numpy(split_tile_filters(w).view(np.int16)).to(device))
            self.bias.append(torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)).to(device))
            self.cin.append(cin); self.cout.append(cout); self.relu.append(1 if relu else 0)
        n = len(layers)
        self.status = torch.zeros(1, dtype=torch.int32, device=device)
        self._wp = (C.c_void_p * n)(*[t.data_ptr() for t in self.wt])
        self._bp = (C.c_void_p * n)(*[t.data_ptr() for t in self.bias])
        self._ci = (C.c_int * n)(*self.cin)
        self._co = (C.c_int * n)(*self.cout)
        self._re = (C.c_int * n)(*self.relu)
        self.last_flags = None
        # the fp32 re-run needs the Winograd tiling of the same filters; widths it is not built for keep the raising contract
        self.safe = n == 8 and _lib.lib().buf_cylindrical_net_wg_supports(self._ci, self._co) == 0 and not os.environ.get('BUF_SPLIT_UNSAFE')
        if self.safe:
            self.wt_wg = [torch.from_numpy(winograd_tile_weights(w)).to(device) for w, _, _ in layers]
            self._wwp = (C.c_void_p * n)(*[t.data_ptr() for t in self.wt_wg])

    def _safe_call(self, x, head):
        L = _lib.lib()
        x = x.contiguous()
        P = x.shape[0]
        out = torch.empty((P, self.cout[-1], 7, 20), dtype=torch.float32, device=x.device)
        desc = torch.empty((P, 32), dtype=torch.float32, device=x.device) if head is not None else None
        self.last_flags = torch.empty((max(P, 1),), dtype=torch.int32, device=x.device)
        check(L.buf_cylindrical_net_split_safe(_ptr(x), P, self._wp, self._wwp, self._bp, self._ci, self._co, self._re,
                                               _ptr(head.params) if head is not None else None, _ptr(out), _ptr(desc), _ptr(self.status),
                                               _ptr(self.last_flags), _stream()), "buf_cylindrical_net_split_safe")
        return (desc, out) if head is not None else out

    def __call__(self, x):
        """x f32[P,16,420] (or [P,48,140]) -> f32[P,32,7,20]"""
        if self.safe:
            return self._safe_call(x, None)
        L = _lib.lib()
        x = x.contiguous()
        P = x.shape[0]
        y = torch.empty((P, self.cout[-1], 7, 20), dtype=torch.float32, device=x.device)
        check(L.buf_cylindrical_net_split(_ptr(x), P, self._wp, self._bp, self._ci, self._co, self._re, _ptr(y), _ptr(self.status), _stream()),
              self.entry)
        return y

    def with_head(self, x, head):
        """x f32[P,16,420] -> (desc f32[P,32], equi f32[P,32,7,20]) with DescriptorHead `head` fused behind the last layer: the
        [32][140] map never leaves LDS; bit-identical to head(self(x))."""
        if self.safe:
            return self._safe_call(x, head)
        L = _lib.lib()
        x = x.contiguous()
        P = x.shape[0]
        desc = torch.empty((P, 32), dtype=torch.float32, device=x.device)
        equi = torch.empty((P, 32, 7, 20), dtype=torch.float32, device=x.device)
        check(L.buf_cylindrical_net_split_head(_ptr(x), P, self._wp, self._bp, self._ci, self._co, self._re, _ptr(head.params),
                                               _ptr(desc), _ptr(equi), _ptr(self.status), _stream()), "buf_cylindrical_net_split_head")
        return desc, equi

    def range_fallbacks(self):
        """patches of the last call that left the f16 range and were recomputed by the fp32 kernel (reads the flags: synchronises)"""
        return 0 if self.last_flags is None else int((self.last_flags != 0).sum().item())

    def check_range(self):
        """Safe form: nothing to raise (flagged patches were recomputed in fp32); clears the informational status word.  Otherwise
        (widths without an fp32 kernel): raises FloatingPointError if an activation left the f16 range."""
        if int(self.status.item()) != 0:
            self.status.zero_()
            if not self.safe:
                raise FloatingPointError("buf_cylindrical_net_split: an activation left the f16 range (|v| >= 65504); use the fp32 kernel")


class DescriptorHead:
    """Attention pooling + normalisation head (patch_embedder.py:66-72,81-84) for csrc/convnet.hip k_desc_head."""

    def __init__(self, w0, b0, w3, b3, device):
        """w0 [16,32], b0 [16], w3 [16], b3 [1]: np.float32, BatchNorms folded -> one device parameter block"""
        f = lambda a, n: np.asarray(a, dtype=np.float32).reshape(n)
        self.params = torch.from_numpy(np.concatenate([f(w0, 512), f(b0, 16), f(w3, 16), f(b3, 1)])).to(device)

    def __call__(self, y):
        """y f32[P,32,7,20] -> desc f32[P,32], equi f32[P,32,7,20]"""
        L = _lib.lib()
        y = y.contiguous()
        P = y.shape[0]
        assert y.dtype == torch.float32 and y.numel() == P * 32 * 140
        desc = torch.empty((P, 32), dtype=torch.float32, device=y.device)
        equi = torch.empty((P, 32, 7, 20), dtype=torch.float32, device=y.device)
        check(L.buf_descriptor_head(_ptr(y), P, _ptr(self.params), _ptr(desc), _ptr(equi), _stream()), "buf_descriptor_head")
        return desc, equi


def split_tile_gemm(w, nt):
    """[Cout, Cin, ntaps] fp32 -> the two f16 planes (hi, lo') in the tiling of csrc/costnet_h3.hip's ch_gemm, groups of nt
    16-output tiles, as uint16 bit patterns (buf_split_tile_gemm; host only)."""
    L = _lib.lib()
    w = np.ascontiguousarray(w, dtype=np.float32)
    cout, cin, ntaps = w.shape
    out = np.empty(L.buf_split_gemm_count(cout, cin, ntaps, nt), dtype=np.uint16)
    check(L.buf_split_tile_gemm(w.ctypes.data, cout, cin, ntaps, nt, out.ctypes.data), "buf_split_tile_gemm")
    return out


def separate_cost_layer0(w):
    """Layer 0 of CostNet is linear in cost[c][n][k][l] = S[c][k][(l-n) mod 20] - T[c][k][l] (models/BUFFER.py:49-60), so its
    3x3x3 kernel w [32,32,3(dn),3(dk),3(dl)] separates exactly into a kernel on S indexed by e = dl - dn (the S-term of an
    output depends on (k', (l'-n') mod 20) only) and one on T summed over dn (the T-term does not depend on n'):
      Ws[(dk*5 + e+2)*32 + c][o] = sum_{dl-dn=e} w[o][c][dn][dk][dl],   Wt[(dk*3 + dl)*32 + c][o] = sum_dn w[o][c][dn][dk][dl]
    (fp64 sums, fp32 storage) -> (Ws f32[480,32], Wt f32[288,32])."""
    w = np.asarray(w, np.float64)
    cout, cin = w.shape[0], w.shape[1]
    ws = np.zeros((3, 5, cin, cout))
    for dn in range(3):
        for dl in range(3):
            ws[:, dl - dn + 2] += np.transpose(w[:, :, dn, :, dl], (2, 1, 0))        # [dk, c, o]
    wt = np.transpose(w.sum(2), (2, 3, 1, 0))                                       # [dk, dl, c, o]
    return ws.reshape(-1, cout).astype(np.float32), wt.reshape(-1, cout).astype(np.float32)


class CostVolumeNet:
    """Device weights of CostNet re-laid for csrc/costnet.hip: layer 0 in its separated form (separate_cost_layer0: Ws then
    Wt in one buffer), layers 6..9 as Wt[((dn*KH+dk)*KW+dl)*Cin + c][Cout], MFMA-tiled; layers 1..5 as U = G g G^T in the
    Winograd tiling (winograd_tile_weights; groups per buf_cost_winograd_group; layer 1 with its three k planes as channels)."""

    def __init__(self, layers, device):
        """layers: 10 x (w [Cout,Cin,KD,KH,KW] np.float32 with BN folded, b [Cout])"""
        assert len(layers) == 10
        self.wt, self.bias = [], []
        for i, (w, b) in enumerate(layers):
            cout, cin = w.shape[0], w.shape[1]
            b = np.asarray(b, np.float32)
            if i == 0:
                assert tuple(w.shape) == (32, 32, 3, 3, 3), w.shape
                tiled = np.concatenate([mfma_tile_weights(m, lk_major=True) for m in separate_cost_layer0(w)])
                self.wt.append(torch.from_numpy(tiled).to(device))
                self.bias.append(torch.from_numpy(np.ascontiguousarray(b)).to(device))
                continue
            if i == 1 and _lib.lib().buf_cost_winograd_group(1):
                # layer 1 collapses k' (3 -> 1): a 3 x 3 correlation over (n', l') with the three k' planes as input channels
                assert tuple(w.shape) == (64, 32, 3, 3, 3), w.shape
                w2d = np.ascontiguousarray(np.transpose(w, (0, 3, 1, 2, 4)).reshape(64, 96, 3, 3))     # [o][(dk, c)][dn][dl]
                self.wt.append(torch.from_numpy(winograd_tile_weights(w2d, ng=_lib.lib().buf_cost_winograd_group(1))).to(device))
                self.bias.append(torch.from_numpy(np.ascontiguousarray(b)).to(device))
                continue
            if 2 <= i <= 5:                               # the (3,1,3) layers 16 -> 14 -> 12 -> 10 -> 8 run in the Winograd domain
                assert tuple(w.shape[2:]) == (3, 1, 3), w.shape
                tiled = winograd_tile_weights(np.ascontiguousarray(w[:, :, :, 0, :]), ng=_lib.lib().buf_cost_winograd_group(i))
                self.wt.append(torch.from_numpy(tiled).to(device))
                self.bias.append(torch.from_numpy(np.ascontiguousarray(b)).to(device))
                continue
            wt = np.transpose(w, (2, 3, 4, 1, 0)).reshape(-1, cout)
            if i == 9:                                    # 20 logits -> two full 16-column tiles
                wt = np.concatenate([wt, np.zeros((wt.shape[0], 32 - cout), np.float32)], 1)
                b = np.concatenate([b, np.zeros(32 - cout, np.float32)])
            self.wt.append(torch.from_numpy(mfma_tile_weights(np.ascontiguousarray(wt, dtype=np.float32), lk_major=True)).to(device))
            self.bias.append(torch.from_numpy(np.ascontiguousarray(b)).to(device))
        self._wp = (C.c_void_p * 10)(*[t.data_ptr() for t in self.wt])
        self._bp = (C.c_void_p * 10)(*[t.data_ptr() for t in self.bias])

    def __call__(self, s_eq, t_eq):
        """s_eq, t_eq f32[M,32,5,20] -> f32[M]"""
        L = _lib.lib()
        s_eq, t_eq = s_eq.contiguous(), t_eq.contiguous()
        m = s_eq.shape[0]
        out = torch.empty((m,), dtype=torch.float32, device=s_eq.device)
        check(L.buf_cost_volume_net(_ptr(s_eq), _ptr(t_eq), m, self._wp, self._bp, _ptr(out), _stream()),
              "buf_cost_volume_net")
        return out

    def gathered(self, equi, s_rows, t_rows):
        """equi f32[rows,32,7,20] (full maps of all keypoints), s_rows / t_rows int64[M] -> f32[M]: the row gather and the
        elevation slice 1..5 (BUFFER.py:291-292) happen inside the kernel."""
        L = _lib.lib()
        equi = _dev(equi, torch.float32, "CostVolumeNet.gathered")
        if equi.dim() != 4 or tuple(equi.shape[1:]) != (32, 7, 20):
            raise _lib.BufferHipError(f"CostVolumeNet.gathered: expected [rows,32,7,20] maps, got {tuple(equi.shape)}")
        s_rows, t_rows = _dev(s_rows, torch.int64, "s_rows"), _dev(t_rows, torch.int64, "t_rows")
        m = int(s_rows.shape[0])
        out = torch.empty((m,), dtype=torch.float32, device=equi.device)
        check(L.buf_cost_volume_net_gather(_ptr(equi), 7, _ptr(s_rows), _ptr(t_rows), m, self._wp, self._bp, _ptr(out), _stream()),
              "buf_cost_volume_net_gather")
        return out


class CostVolumeNetSplit:
    """CostNet on the f16 matrix pipe with fp32-equivalent arithmetic (csrc/costnet_h3.hip): opt-in next to CostVolumeNet, same calls.
    Layer 0 in its separated form (separate_cost_layer0), layer 1 with its three k' planes as channels, every layer direct form."""

    def __init__(self, layers, device):
        """layers: 10 x (w [Cout,Cin,KD,KH,KW] np.float32 with BN folded, b [Cout])"""
        assert len(layers) == 10
        mats = []
        w0 = layers[0][0]
        assert tuple(w0.shape) == (32, 32, 3, 3, 3), w0.shape
        ws, wt = separate_cost_layer0(w0)
        mats.append((np.transpose(ws.reshape(15, 32, 32), (2, 1, 0)), 2))                      # [o][c][dk*5 + e+2]
        mats.append((np.transpose(wt.reshape(9, 32, 32), (2, 1, 0)), 2))                       # [o][c][dk*3 + dl]
        w1 = layers[1][0]
        assert tuple(w1.shape) == (64, 32, 3, 3, 3), w1.shape
        mats.append((np.transpose(w1, (0, 3, 1, 2, 4)).reshape(64, 96, 9), 2))                 # [o][dk*32 + c][dn*3 + dl]
        for i in range(2, 9):
            w = layers[i][0]
            assert tuple(w.shape[2:]) == (3, 1, 3), w.shape
            mats.append((np.ascontiguousarray(w[:, :, :, 0, :]).reshape(w.shape[0], w.shape[1], 9), 2 if i < 7 else 1))
        w9 = layers[9][0]
        assert tuple(w9.shape) == (20, 32, 2, 1, 2), w9.shape
        mats.append((np.ascontiguousarray(w9[:, :, :, 0, :]).reshape(20, 32, 4), 1))
        self.wt = [torch.from_numpy(split_tile_gemm(m, nt).view(np.int16)).to(device) for m, nt in mats]
        self.bias = []
        for i, (_, b) in enumerate(layers):
            b = np.asarray(b, np.float32)
            if i == 9:
                b = np.concatenate([b, np.zeros(32 - b.shape[0], np.float32)])
            self.bias.append(torch.from_numpy(np.ascontiguousarray(b)).to(device))
        self.status = torch.zeros(1, dtype=torch.int32, device=device)
        self._wp = (C.c_void_p * 11)(*[t.data_ptr() for t in self.wt])
        self._bp = (C.c_void_p * 10)(*[t.data_ptr() for t in self.bias])
        # safe by construction (csrc/split_safe.hip): the fp32 kernel's weights ride along and re-run the matches the split kernel flags
        self.safe = not os.environ.get('BUF_SPLIT_UNSAFE')
        self.f32 = CostVolumeNet(layers, device) if self.safe else None
        self.last_flags = None

    def _safe(self, s_eq, t_eq, s_rows, t_rows, m, dev):
        L = _lib.lib()
        out = torch.empty((m,), dtype=torch.float32, device=dev)
        self.last_flags = torch.empty((max(m, 1),), dtype=torch.int32, device=dev)
        check(L.buf_cost_volume_net_split_safe(_ptr(s_eq), _ptr(t_eq), 7, _ptr(s_rows), _ptr(t_rows), m, self._wp, self._bp, self.f32._wp,
                                               self.f32._bp, _ptr(out), _ptr(self.status), _ptr(self.last_flags), _stream()),
              "buf_cost_volume_net_split_safe")
        return out

    def __call__(self, s_eq, t_eq):
        """s_eq, t_eq f32[M,32,5,20] -> f32[M]"""
        L = _lib.lib()
        s_eq, t_eq = s_eq.contiguous(), t_eq.contiguous()
        m = s_eq.shape[0]
        if self.safe:
            return self._safe(s_eq, t_eq, None, None, m, s_eq.device)
        out = torch.empty((m,), dtype=torch.float32, device=s_eq.device)
        check(L.buf_cost_volume_net_split(_ptr(s_eq), _ptr(t_eq), m, self._wp, self._bp, _ptr(out), _ptr(self.status), _stream()),
              "buf_cost_volume_net_split")
        return out

    def gathered(self, equi, s_rows, t_rows):
        L = _lib.lib()
        equi = _dev(equi, torch.float32, "CostVolumeNetSplit.gathered")
        if equi.dim() != 4 or tuple(equi.shape[1:]) != (32, 7, 20):
            raise _lib.BufferHipError(f"CostVolumeNetSplit.gathered: expected [rows,32,7,20] maps, got {tuple(equi.shape)}")
        s_rows, t_rows = _dev(s_rows, torch.int64, "s_rows"), _dev(t_rows, torch.int64, "t_rows")
        m = int(s_rows.shape[0])
        if self.safe:
            return self._safe(equi, equi, s_rows, t_rows, m, equi.device)
        out = torch.empty((m,), dtype=torch.float32, device=equi.device)
        check(L.buf_cost_volume_net_split_gather(_ptr(equi), 7, _ptr(s_rows), _ptr(t_rows), m, self._wp, self._bp, _ptr(out),
                                                 _ptr(self.status), _stream()), "buf_cost_volume_net_split_gather")
        return out

    def range_fallbacks(self):
        """matches of the last call that left the f16 range and were recomputed by the fp32 kernel (synchronises)"""
        return 0 if self.last_flags is None else int((self.last_flags != 0).sum().item())

    def check_range(self):
        """Safe form: nothing to raise (flagged matches were recomputed in fp32); clears the informational status word."""
        if int(self.status.item()) != 0:
            self.status.zero_()
            if not self.safe:
                raise FloatingPointError("buf_cost_volume_net_split: a value left the f16 range (|v| >= 65504); use the fp32 kernel")
