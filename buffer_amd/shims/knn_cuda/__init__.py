"""knn_cuda.KNN (KNN_CUDA 0.2, README.md:32; call sites models/BUFFER.py:347,352,374): brute-force
k-NN, Euclidean distances ascending, 0-based int64 indices."""
import torch

from buffer_amd import ops

__version__ = "0.2"


class KNN(torch.nn.Module):
    def __init__(self, k, transpose_mode=False):
        super().__init__()
        self.k = k
        self._t = transpose_mode

    def forward(self, ref, query):
        assert ref.size(0) == query.size(0), "ref.shape={} != query.shape={}".format(ref.shape, query.shape)
        with torch.no_grad():
            if not self._t:                      # [B, D, N] layout -> [B, N, D]
                ref, query = ref.transpose(1, 2), query.transpose(1, 2)
            d, i = ops.knn(ref.contiguous(), query.contiguous(), self.k)
            if not self._t:
                d, i = d.transpose(1, 2).contiguous(), i.transpose(1, 2).contiguous()
        return d, i


def knn(ref, query, k):
    """functional form on [B, D, N] inputs -> ([B, k, Q], [B, k, Q])"""
    return KNN(k, transpose_mode=False)(ref, query)
