"""N1 -- the open3d pre-processing of the reference's datasets on device
(ThreeDMatch/dataset.py:89-153, KITTI/dataset.py): `voxel_down_sample`, `estimate_normals` (30-NN),
`orient_normals_towards_camera_location`, behind csrc/preprocess.hip.  open3d (0.13.0, README.md:28) is absent from
/root/reference and from this image: parity unpinned, algorithms restated from its published sources."""
import ctypes as C

import numpy as np
import torch

from . import _lib, ops
from ._lib import check
from .ops import _dev, _ptr, _stream


_SHUFFLE_SALT = 0x5851F42D4C957F2D


def voxel_down_sample(points, voxel_size, normals=None, max_cells=0):
    """open3d PointCloud.voxel_down_sample: points f32|f64[n,3] (device) -> f64[m,3] voxel means
    (, f64[m,3] mean normals), rows in ascending voxel-key order."""
    L = _lib.lib()
    if not isinstance(points, torch.Tensor) or not points.is_cuda:
        raise _lib.BufferHipError("voxel_down_sample: expected a tensor in device memory (buffer_amd has no CPU path)")
    if points.dim() != 2 or points.shape[1] != 3:
        raise _lib.BufferHipError("voxel_down_sample: points.shape is not (N, 3)")
    if not voxel_size > 0:
        raise _lib.BufferHipError("voxel_down_sample: voxel_size <= 0.")          # open3d's message
    dt = torch.float64 if points.dtype == torch.float64 else torch.float32
    points = points.to(dt).contiguous()
    if normals is not None:
        normals = _dev(normals, dt, "voxel_down_sample.normals")
    n = int(points.shape[0])
    if max_cells <= 0:
        max_cells = max(1 << 16, 4 * n)          # bucket table of the counting sort: the result does not depend on it, the scan over it costs
    nbytes = L.buf_voxel_downsample_ws_bytes(n, max_cells)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=points.device)
    out = torch.empty((max(n, 1), 3), dtype=torch.float64, device=points.device)
    out_n = torch.empty((max(n, 1), 3), dtype=torch.float64, device=points.device) if normals is not None else None
    m = C.c_int(0)
    check(L.buf_voxel_downsample(_ptr(points), _ptr(normals), 1 if dt == torch.float64 else 0, n, float(voxel_size),
                                 _ptr(out), _ptr(out_n), C.byref(m), max_cells, _ptr(ws), nbytes, _stream()),
          "buf_voxel_downsample")
    if normals is not None:
        return out[:m.value], out_n[:m.value]
    return out[:m.value]


def voxel_down_sample_batch(points, lengths, voxel_size, max_cells=0):
    """voxel_down_sample for several clouds stacked in `points` (f32|f64[sum n_c,3], lengths int[nc]) in one set of launches and
    one host round trip -> (f64[sum m_c,3] voxel means cloud after cloud, int32[nc] row counts); cloud by cloud the rows are
    those of voxel_down_sample."""
    L = _lib.lib()
    if not isinstance(points, torch.Tensor) or not points.is_cuda:
        raise _lib.BufferHipError("voxel_down_sample_batch: expected a tensor in device memory (buffer_amd has no CPU path)")
    if not voxel_size > 0:
        raise _lib.BufferHipError("voxel_down_sample: voxel_size <= 0.")
    dt = torch.float64 if points.dtype == torch.float64 else torch.float32
    points = points.to(dt).contiguous()
    lens = np.ascontiguousarray(lengths, dtype=np.int32)
    n, nb = int(points.shape[0]), int(lens.shape[0])
    if int(lens.sum()) != n or nb == 0:
        raise _lib.BufferHipError("voxel_down_sample_batch: lengths do not sum to the number of points")
    if max_cells <= 0:
        max_cells = max(1 << 16, 4 * n, 2 * nb)
    nbytes = L.buf_voxel_downsample_batch_ws_bytes(n, nb, max_cells)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=points.device)
    out = torch.empty((max(n, 1), 3), dtype=torch.float64, device=points.device)
    out_lens = np.zeros(nb, np.int32)
    check(L.buf_voxel_downsample_batch(_ptr(points), 1 if dt == torch.float64 else 0, n, lens.ctypes.data_as(C.c_void_p), nb,
                                       float(voxel_size), _ptr(out), out_lens.ctypes.data_as(C.c_void_p), max_cells, _ptr(ws), nbytes,
                                       _stream()), "buf_voxel_downsample_batch")
    return out[:int(out_lens.sum())], out_lens


def estimate_normals(points, knn=30, camera=(0.0, 0.0, 0.0), orient=True, radius=None, ncand=40, grow=1.35, lengths=None):
    """open3d estimate_normals(KDTreeSearchParamKNN(knn)) [+ orient_normals_towards_camera_location(camera)]:
    points f32[n,3] (device) -> unit normals f32[n,3].  `lengths` (optional): the rows are several clouds stacked
    (neighbours are searched inside a point's own cloud) -- one set of launches for all of them; the normals equal those of
    cloud-by-cloud calls (each point ends with its exact knn nearest neighbours either way).

    k-NN through the cell grid of the radius search: candidates = the `ncand` nearest inside a ball (fp32, sorted), re-ranked
    in fp64 by the kernel; rows whose ball held fewer than min(knn, n) points are retried with a `grow` times larger
    radius (small steps keep the balls under the 64 candidates the wave-per-query radius kernel handles in one pass)."""
    L = _lib.lib()
    pts = _dev(points, torch.float32, "estimate_normals")
    n = int(pts.shape[0])
    out = torch.zeros((n, 3), dtype=torch.float32, device=pts.device)
    if n == 0:
        return out
    lens = np.array([n], np.int32) if lengths is None else np.asarray(lengths, np.int32)
    if int(lens.sum()) != n:
        raise _lib.BufferHipError("estimate_normals: lengths do not sum to the number of points")
    if len(lens) > 1 and int(lens[lens > 0].min()) <= knn:       # a cloud smaller than the neighbourhood: one by one
        lo = 0
        for m in lens:
            out[lo:lo + m] = estimate_normals(pts[lo:lo + m], knn, camera, orient, radius, ncand, grow)
            lo += int(m)
        return out
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ncand = max(int(ncand), int(knn))
    if radius is None:
        # pilot: distance to the knn-th neighbour for a few hundred queries of the first cloud (plumbing; sets only the first radius)
        n0 = int(lens[0]) if int(lens[0]) > knn else n
        g = torch.Generator(device=pts.device).manual_seed(0)
        sel = torch.randperm(n0, generator=g, device=pts.device)[:256]
        d = torch.cdist(pts[:n0][sel].double(), pts[:n0].double())
        kth = torch.topk(d, min(int(knn), n0), dim=1, largest=False).values[:, -1]
        radius = float(kth.median().item()) * 1.1 + 1e-9
    cam = (C.c_double * 3)(*[float(c) for c in camera])
    todo = None
    q_lens = lens
    r = float(radius)
    for _ in range(200):
        grid = ops.CellGrid(pts, lens, r)
        q = pts if todo is None else pts[todo.long()].contiguous()
        nq = int(q.shape[0])
        cand = grid.query(q, q_lens, ncand)
        deficient = torch.empty((nq,), dtype=torch.uint8, device=pts.device)
        check(L.buf_knn_normals(_ptr(pts), n, _ptr(todo), nq, _ptr(cand), ncand, int(knn), cam, 1 if orient else 0,
                                _ptr(out), _ptr(deficient), _stream()), "buf_knn_normals")
        bad = torch.nonzero(deficient).reshape(-1).to(torch.int32)
        if bad.numel() == 0:
            return out
        todo = bad if todo is None else todo[bad.long()].contiguous()
        if len(lens) > 1:                                        # retried rows per cloud (ascending indices)
            t_host = todo.cpu().numpy()
            q_lens = np.diff(np.searchsorted(t_host, offs)).astype(np.int32)
        else:
            q_lens = np.array([todo.shape[0]], np.int32)
        r *= grow
    raise _lib.BufferHipError("estimate_normals: neighbourhood search did not converge")


def prepare_fragments(raws, downsample, voxel_size_0, max_num_pts=30000, seeds=None, with_normals=True):
    """The test-split branch of ThreeDMatchDataset.__getitem__ (dataset.py:93-95,125-153) for SEVERAL fragments:
    raws: list of f32[n,3] device tensors -> list of dict(fds_pts f32[N,3] shuffled, sds_pts f32[M,6] = shuffled
    second-level points + normals).  The two voxel levels and the shuffles run per fragment (second level on the first
    level's fp64 means, like open3d's chained calls; fragment i shuffles with the keyed permutation of seeds[i]); the 30-NN normals of
    all fragments are estimated in ONE stacked pass.  Fragment by fragment the result is that of prepare_fragment."""
    seeds = list(range(len(raws))) if seeds is None else list(seeds)
    out, sds_all = [], []
    if not raws:
        return out
    # both voxel levels for all fragments at once (second level on the first level's fp64 means: chained open3d calls)
    fds_all, fds_lens = voxel_down_sample_batch(torch.cat(raws) if len(raws) > 1 else raws[0], [int(r.shape[0]) for r in raws], downsample)
    sds_all64, sds_lens = voxel_down_sample_batch(fds_all, fds_lens, voxel_size_0)
    fds32, sds32_all = fds_all.float(), sds_all64.float()
    fo = np.concatenate([[0], np.cumsum(fds_lens)])
    so = np.concatenate([[0], np.cumsum(sds_lens)])
    levels, keys = [], []
    for i, seed in enumerate(seeds):
        levels += [fds32[fo[i]:fo[i + 1]], sds32_all[so[i]:so[i + 1]]]
        # salted: the pipeline permutes the support clouds of pair `seed` with perm_key(seed, 0 / 1) itself
        keys += [ops.perm_key(seed, 0) ^ _SHUFFLE_SALT, ops.perm_key(seed, 1) ^ _SHUFFLE_SALT]
    # np.random.shuffle of both levels (dataset.py:95,112): a keyed pseudo-random permutation per cloud, ONE launch for all of
    # them (torch.randperm is a radix sort per call)
    stacked, lens = ops.permute_clouds(levels, keys)
    offs = np.concatenate([[0], np.cumsum(lens)])
    for i, seed in enumerate(seeds):
        fds32 = stacked[offs[2 * i]:offs[2 * i + 1]]
        sds32 = stacked[offs[2 * i + 1]:offs[2 * i + 2]]
        if sds32.shape[0] > max_num_pts:                                                       # dataset.py:131-137
            g = torch.Generator(device=sds32.device).manual_seed(int(seed))
            sds32 = sds32[torch.randperm(sds32.shape[0], generator=g, device=sds32.device)[:max_num_pts]]
        out.append(dict(fds_pts=fds32.contiguous()))
        sds_all.append(sds32.contiguous())
    if with_normals and sds_all:
        lens = [int(s.shape[0]) for s in sds_all]
        nrm = estimate_normals(torch.cat(sds_all), lengths=lens)
        lo = 0
        for i, s in enumerate(sds_all):
            sds_all[i] = torch.cat([s, nrm[lo:lo + lens[i]]], dim=1).contiguous()
            lo += lens[i]
    for o, s in zip(out, sds_all):
        o['sds_pts'] = s
    return out


def prepare_fragment(raw_points, downsample, voxel_size_0, max_num_pts=30000, seed=0, with_normals=True):
    """One fragment of prepare_fragments: raw f32[n,3] -> dict(fds_pts f32[N,3] shuffled, sds_pts f32[M,6])."""
    return prepare_fragments([raw_points], downsample, voxel_size_0, max_num_pts, [seed], with_normals)[0]
