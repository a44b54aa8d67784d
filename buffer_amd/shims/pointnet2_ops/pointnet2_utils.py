"""pointnet2_ops.pointnet2_utils -- the five operators BUFFER names (README.md:31; call sites
models/BUFFER.py:266-271, models/patch_embedder.py:100-104, utils/common.py:442-455), inference only.
Inputs must be contiguous float32 / int32 tensors in device memory; outputs are allocated on the same
device and produced on the current stream (the upstream autograd.Function.apply aliases)."""
import torch

from buffer_amd import ops


def _check(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(f"{name} must be a CUDA tensor")


def furthest_point_sample(xyz, npoint):
    _check(xyz, "xyz")
    return ops.furthest_point_sample(xyz, npoint)


def gather_operation(features, idx):
    _check(features, "features")
    return ops.gather_operation(features, idx)


def ball_query(radius, nsample, xyz, new_xyz):
    _check(xyz, "xyz")
    return ops.ball_query(radius, nsample, xyz, new_xyz)


def grouping_operation(features, idx):
    _check(features, "features")
    return ops.grouping_operation(features, idx)


def three_nn(unknown, known):
    _check(unknown, "unknown")
    return ops.three_nn(unknown, known)
