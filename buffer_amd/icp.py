"""Point-to-point ICP on the device: the open3d `registration_icp` call that refines KITTI's odometry ground truth
(KITTI/dataset.py:95-117: threshold 0.20 m, identity init, <= 200 iterations, relative fitness / RMSE 1e-6).

Per iteration: nearest target point inside the correspondence distance for every transformed source point (A2 cell grid,
column 0 of the distance-sorted neighbour row: csrc/radius.hip), fitness = matched / source points, RMSE over the matches,
then the rigid update from the matches (centred 3x3 cross-covariance in fp64, SVD with the det correction) composed onto
the running transform.  open3d's loop, restated; open3d itself is absent here (parity unpinned)."""
import numpy as np
import torch

from . import ops


def _kabsch(p, q):
    """rigid T (4x4 f64, numpy) minimising sum |R p + t - q|^2; p, q f64[n,3] device tensors."""
    pc, qc = p.mean(0), q.mean(0)
    H = ((p - pc).T @ (q - qc)).cpu().numpy()
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T)) or 1.0])
    R = Vt.T @ D @ U.T
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = qc.cpu().numpy() - R @ pc.cpu().numpy()
    return T


def icp_point_to_point(src, tgt, max_dist, init=None, max_iteration=30, relative_fitness=1e-6, relative_rmse=1e-6):
    """src f32[n,3], tgt f32[m,3] (device) -> (T f64[4,4] numpy src->tgt, fitness, inlier_rmse, corr int32[k,2] numpy)."""
    if not (src.is_cuda and tgt.is_cuda):
        raise RuntimeError("icp_point_to_point: expected tensors in device memory (buffer_amd has no CPU path)")
    dev = src.device
    T = np.eye(4) if init is None else np.asarray(init, np.float64).copy()
    n, m = int(src.shape[0]), int(tgt.shape[0])
    empty = np.zeros((0, 2), np.int32)
    if n == 0 or m == 0:
        return T, 0.0, 0.0, empty
    grid = ops.CellGrid(tgt.float().contiguous(), [m], float(max_dist))
    src64, tgt64 = src.double(), tgt.double()

    def correspond(Tn):
        Tt = torch.from_numpy(Tn).to(dev)
        moved = src64 @ Tt[:3, :3].T + Tt[:3, 3]
        nn = grid.query(moved.float().contiguous(), [n], 1)[:, 0].long()
        hit = torch.nonzero(nn < m).flatten()
        if hit.numel() == 0:
            return moved, hit, nn, 0.0, 0.0
        d2 = ((moved[hit] - tgt64[nn[hit]]) ** 2).sum(1)
        return moved, hit, nn, hit.numel() / n, float(torch.sqrt(d2.mean()).item())

    moved, hit, nn, fit, rmse = correspond(T)
    for _ in range(int(max_iteration)):
        if hit.numel() < 3:
            break
        T = _kabsch(moved[hit], tgt64[nn[hit]]) @ T
        prev_fit, prev_rmse = fit, rmse
        moved, hit, nn, fit, rmse = correspond(T)
        if abs(prev_fit - fit) < relative_fitness and abs(prev_rmse - rmse) < relative_rmse:
            break
    corr = torch.stack([hit, nn[hit]], 1).to(torch.int32).cpu().numpy() if hit.numel() else empty
    return T, float(fit), float(rmse), corr
