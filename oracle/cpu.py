"""TEST INFRASTRUCTURE ONLY -- ctypes/numpy front-end to oracle/liboracle.so (our
plain-C restatement) and, when built, oracle/_ref/libbuffer_ref.so (the
reference's own C++ cores compiled in place from /root/reference).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (buffer_amd) never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BUF_ORACLE_SO / BUF_ORACLE_REF_SO: the sanitizer builds (make -C oracle asan) under tests/test_sanitizers_cpu.py
_ORACLE_SO = os.environ.get("BUF_ORACLE_SO") or os.path.join(_HERE, "liboracle.so")
_REF_SO = os.environ.get("BUF_ORACLE_REF_SO") or os.path.join(_HERE, "_ref", "libbuffer_ref.so")

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def build(ref=True):
    """Compile liboracle.so (always) and _ref (only where /root/reference is mounted)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if ref and os.path.isdir("/root/reference/cpp_wrappers"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_ref"])


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_ORACLE_SO):
            build(ref=False)
        _lib = C.CDLL(_ORACLE_SO)
        _lib.orc_grid_subsample_batch.restype = C.c_int
        _lib.orc_radius_neighbors.restype = C.c_int
    return _lib


def have_ref():
    return os.path.exists(_REF_SO)


def ref():
    global _ref
    if _ref is None:
        if not have_ref():
            raise RuntimeError("oracle/_ref/libbuffer_ref.so not built (run `make -C oracle _ref` "
                               "where /root/reference is mounted)")
        _ref = C.CDLL(_REF_SO)
        _ref.ref_batch_query.restype = C.c_int
        _ref.ref_subsample_batch.restype = C.c_int
    return _ref


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t)


# ----------------------------------------------------------------------------- A1
def grid_subsample_batch(points, batches, dl, max_p=0, return_keys=False):
    points, batches = _f32(points), _i32(batches)
    n, nb = points.shape[0], batches.shape[0]
    out = np.empty((max(n, 1), 3), np.float32)
    ob = np.empty(nb, np.int32)
    keys = np.empty(max(n, 1), np.uint64)
    m = lib().orc_grid_subsample_batch(_p(points, _fp), n, _p(batches, _ip), nb, C.c_float(dl),
                                       int(max_p), _p(out, _fp), _p(ob, _ip),
                                       keys.ctypes.data_as(C.POINTER(C.c_uint64)))
    if m < 0:
        raise RuntimeError("orc_grid_subsample_batch failed")
    if return_keys:
        return out[:m].copy(), ob, keys[:m].copy()
    return out[:m].copy(), ob


def ref_grid_subsample_batch(points, batches, dl, max_p=0):
    points, batches = _f32(points), _i32(batches)
    n, nb = points.shape[0], batches.shape[0]
    ob = np.empty(nb, np.int32)
    ptr = _fp()
    m = ref().ref_subsample_batch(_p(points, _fp), n, _p(batches, _ip), nb, C.c_float(dl), int(max_p),
                                  C.byref(ptr), _p(ob, _ip))
    out = np.ctypeslib.as_array(ptr, shape=(max(m, 1), 3))[:m].copy()
    ref().ref_free(ptr)
    return out, ob


# ----------------------------------------------------------------------------- A2
def _radius(fn, free, queries, supports, q_batches, s_batches, radius):
    q, s = _f32(queries), _f32(supports)
    qb, sb = _i32(q_batches), _i32(s_batches)
    ptr = _ip()
    mc = fn(_p(q, _fp), q.shape[0], _p(s, _fp), s.shape[0], _p(qb, _ip), _p(sb, _ip), qb.shape[0],
            C.c_float(radius), C.byref(ptr))
    if mc > 0 and q.shape[0] > 0:
        out = np.ctypeslib.as_array(ptr, shape=(q.shape[0], mc)).copy()
    else:
        out = np.zeros((q.shape[0], 0), np.int32)
    free(ptr)
    return out


def radius_neighbors(queries, supports, q_batches, s_batches, radius):
    return _radius(lib().orc_radius_neighbors, lib().orc_free, queries, supports, q_batches, s_batches, radius)


def ref_radius_neighbors(queries, supports, q_batches, s_batches, radius):
    return _radius(ref().ref_batch_query, ref().ref_free, queries, supports, q_batches, s_batches, radius)


# ------------------------------------------------------------- external-op restatements
def fps(xyz, m):
    xyz = _f32(xyz)
    b, n, _ = xyz.shape
    idx = np.zeros((b, m), np.int32)
    lib().orc_fps(_p(xyz, _fp), b, n, int(m), _p(idx, _ip))
    return idx


def ball_query(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    lib().orc_ball_query(_p(xyz, _fp), _p(new_xyz, _fp), b, n, m, C.c_float(radius), int(nsample), _p(idx, _ip))
    return idx


def three_nn(unknown, known):
    unknown, known = _f32(unknown), _f32(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    dist = np.zeros((b, n, 3), np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    lib().orc_three_nn(_p(unknown, _fp), _p(known, _fp), b, n, m, _p(dist, _fp), _p(idx, _ip))
    return dist, idx


def knn(ref_pts, query, k):
    ref_pts, query = _f32(ref_pts), _f32(query)
    b, n, d = ref_pts.shape
    q = query.shape[1]
    dist = np.zeros((b, q, k), np.float32)
    idx = np.zeros((b, q, k), np.int64)
    lib().orc_knn(_p(ref_pts, _fp), _p(query, _fp), b, n, q, d, int(k), _p(dist, _fp),
                  idx.ctypes.data_as(C.POINTER(C.c_int64)))
    return dist, idx


def gather_operation(feat, idx):
    """feat[B,C,N], idx[B,M] -> [B,C,M]  (plain gather, pointnet2_ops)."""
    feat, idx = np.asarray(feat), np.asarray(idx)
    return np.take_along_axis(feat, idx[:, None, :].astype(np.int64).repeat(feat.shape[1], 1), axis=2)


def grouping_operation(feat, idx):
    """feat[B,C,N], idx[B,M,S] -> [B,C,M,S]."""
    feat, idx = np.asarray(feat), np.asarray(idx)
    b, c, n = feat.shape
    _, m, s = idx.shape
    flat = idx.reshape(b, 1, m * s).astype(np.int64).repeat(c, 1)
    return np.take_along_axis(feat, flat, axis=2).reshape(b, c, m, s)


# ---- open3d 0.13.0 pre-processing (parity unpinned; restated from open3d's published sources) ----
_dp = C.POINTER(C.c_double)


def o3d_voxel_down_sample(points, voxel_size, normals=None):
    """points f64[n,3] -> f64[m,3] voxel means (, mean normals), ascending voxel key."""
    pts = np.ascontiguousarray(points, dtype=np.float64)
    n = pts.shape[0]
    nrm = None if normals is None else np.ascontiguousarray(normals, dtype=np.float64)
    out = np.zeros((max(n, 1), 3), np.float64)
    out_n = np.zeros((max(n, 1), 3), np.float64) if nrm is not None else None
    f = lib().orc_o3d_voxel_downsample
    f.restype = C.c_int
    m = f(_p(pts, _dp), None if nrm is None else _p(nrm, _dp), int(n), C.c_double(voxel_size), _p(out, _dp),
          None if out_n is None else _p(out_n, _dp))
    if m < 0:
        raise ValueError("voxel_size <= 0.")
    return (out[:m], out_n[:m]) if nrm is not None else out[:m]


def o3d_estimate_normals(points, knn=30, camera=(0.0, 0.0, 0.0), orient=True):
    pts = _f32(points)
    n = pts.shape[0]
    out = np.zeros((n, 3), np.float32)
    cam = np.ascontiguousarray(camera, dtype=np.float64)
    lib().orc_o3d_estimate_normals(_p(pts, _fp), int(n), int(knn), _p(cam, _dp), 1 if orient else 0, _p(out, _fp))
    return out


def o3d_fast_eigen3x3(cov6):
    cov = np.ascontiguousarray(cov6, dtype=np.float64)
    out = np.zeros(3, np.float64)
    lib().orc_o3d_fast_eigen3x3(_p(cov, _dp), _p(out, _dp))
    return out
