"""torch_batch_svd.svd (README.md:35; call site utils/common.py:715): batched SVD of 3x3 matrices,
A = U diag(S) V^T with S descending (signs are free, callers disambiguate)."""
import torch

from buffer_amd import ops


def svd(a):
    if a.dim() != 3 or a.shape[1] != 3 or a.shape[2] != 3:
        raise RuntimeError("torch_batch_svd (MI355X): only [B,3,3] inputs are supported (BUFFER's cal_Z_axis)")
    if not a.is_cuda:
        raise RuntimeError("input must be a CUDA tensor")
    return ops.svd3x3(a)
