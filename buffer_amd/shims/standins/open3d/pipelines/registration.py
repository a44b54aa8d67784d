import numpy as np


class TransformationEstimationPointToPoint:
    def __init__(self, with_scaling=False):
        self.with_scaling = bool(with_scaling)


class CorrespondenceCheckerBasedOnEdgeLength:
    def __init__(self, similarity_threshold=0.9):
        self.similarity_threshold = float(similarity_threshold)


class CorrespondenceCheckerBasedOnDistance:
    def __init__(self, distance_threshold):
        self.distance_threshold = float(distance_threshold)


class RANSACConvergenceCriteria:
    def __init__(self, max_iteration=100000, confidence=0.999):
        self.max_iteration, self.confidence = int(max_iteration), float(confidence)


class ICPConvergenceCriteria:
    def __init__(self, relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30):
        self.relative_fitness, self.relative_rmse, self.max_iteration = float(relative_fitness), float(relative_rmse), int(max_iteration)


class RegistrationResult:
    def __init__(self, transformation=None, fitness=0.0, inlier_rmse=0.0, correspondence_set=None):
        self.transformation = np.eye(4) if transformation is None else np.asarray(transformation, np.float64)
        self.fitness, self.inlier_rmse = float(fitness), float(inlier_rmse)
        self.correspondence_set = np.zeros((0, 2), np.int32) if correspondence_set is None else correspondence_set

    def __repr__(self):
        return (f"RegistrationResult with fitness={self.fitness:e}, inlier_rmse={self.inlier_rmse:e}, "
                f"and correspondence_set size of {len(self.correspondence_set)}")


def _device():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("open3d stand-in (buffer_amd): this call runs on a HIP device and none is visible")
    return torch.device("cuda", torch.cuda.current_device())


def registration_ransac_based_on_correspondence(source, target, corres, max_correspondence_distance,
                                                estimation_method=None, ransac_n=3, checkers=(), criteria=None, seed=0):
    """models/BUFFER.py:318-326 -> buf_ransac_kabsch (csrc/registration.hip): a fixed budget of seeded 3-point
    hypotheses, each pre-checked by the edge-length and distance checkers, Kabsch, ranked by inlier count then RMSE
    within `max_correspondence_distance`.  `criteria.max_iteration` caps the budget (default budget 4096);
    `confidence` early termination does not apply to a batch that is evaluated at once."""
    import torch
    from buffer_amd import ops
    from buffer_amd.config import THREEDMATCH
    est = estimation_method or TransformationEstimationPointToPoint(False)
    if est.with_scaling or int(ransac_n) != 3:
        raise NotImplementedError("open3d stand-in: rigid point-to-point estimation from 3-point samples only")
    edge, dist = 0.0, float(max_correspondence_distance)       # no checker = nothing rejected
    for c in checkers:
        if isinstance(c, CorrespondenceCheckerBasedOnEdgeLength):
            edge = c.similarity_threshold
        elif isinstance(c, CorrespondenceCheckerBasedOnDistance):
            if abs(c.distance_threshold - dist) > 1e-12:          # the kernel checks samples and scores candidates with ONE distance
                raise NotImplementedError("open3d stand-in: the distance checker must use max_correspondence_distance")
        else:
            raise NotImplementedError(f"open3d stand-in: checker {type(c).__name__}")
    nhyp = THREEDMATCH.ransac_hypotheses
    if criteria is not None:
        nhyp = max(1, min(nhyp, criteria.max_iteration))
    dev = _device()
    src = torch.from_numpy(np.asarray(source.points, np.float32)).to(dev)
    tgt = torch.from_numpy(np.asarray(target.points, np.float32)).to(dev)
    corr = np.asarray(corres, np.int32).reshape(-1, 2)
    if len(corr) < 3:
        return RegistrationResult()
    T, info = ops.ransac_kabsch(src, tgt, torch.from_numpy(corr).to(dev), nhyp=nhyp, seed=seed,
                                max_dist=float(max_correspondence_distance), edge_similarity=edge)
    T = T.cpu().numpy().astype(np.float64)
    p = np.asarray(source.points)[corr[:, 0]] @ T[:3, :3].T + T[:3, 3]
    d = np.linalg.norm(p - np.asarray(target.points)[corr[:, 1]], axis=1)
    inl = d < max_correspondence_distance
    return RegistrationResult(T, inl.mean() if len(inl) else 0.0, float(np.sqrt((d[inl] ** 2).mean())) if inl.any() else 0.0,
                              corr[inl])


def registration_icp(source, target, max_correspondence_distance, init=None, estimation_method=None, criteria=None):
    """KITTI/dataset.py:104-107: point-to-point ICP on the device (buffer_amd/icp.py)."""
    import torch
    from buffer_amd import icp
    est = estimation_method or TransformationEstimationPointToPoint(False)
    if est.with_scaling:
        raise NotImplementedError("open3d stand-in: rigid point-to-point ICP only")
    cr = criteria or ICPConvergenceCriteria()
    dev = _device()
    src = torch.from_numpy(np.asarray(source.points, np.float32)).to(dev)
    tgt = torch.from_numpy(np.asarray(target.points, np.float32)).to(dev)
    T, fit, rmse, corr = icp.icp_point_to_point(src, tgt, float(max_correspondence_distance),
                                                np.eye(4) if init is None else np.asarray(init, np.float64),
                                                cr.max_iteration, cr.relative_fitness, cr.relative_rmse)
    return RegistrationResult(T, fit, rmse, corr)
