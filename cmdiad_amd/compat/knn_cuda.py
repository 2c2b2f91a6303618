"""``knn_cuda.KNN`` (unlimblue/KNN_CUDA 0.2) on cmdiad_knn_group: brute-force squared-L2 k nearest neighbours, ascending,
int64 indices.  transpose_mode=True takes ref [B,N,C] / query [B,M,C] and returns (dist [B,M,k], idx [B,M,k]) -- the only
mode the reference uses (models/models.py:86,100; C = 3, k <= 128)."""
import torch
from torch import nn

from .. import ops


class KNN(nn.Module):
    def __init__(self, k, transpose_mode=False):
        super().__init__()
        self.k, self.transpose_mode = k, transpose_mode

    def forward(self, ref, query):
        if not self.transpose_mode:
            ref, query = ref.transpose(1, 2), query.transpose(1, 2)
        if ref.shape[-1] != 3:
            raise NotImplementedError("cmdiad_amd's KNN covers 3-d points (the reference's only use)")
        ref, query = ref.detach().float().contiguous(), query.detach().float().contiguous()
        idx, nb = ops.knn_group(ref, query, self.k)          # nb = neighbour - query
        dist = nb.pow(2).sum(-1).sqrt()                       # KNN_CUDA returns Euclidean distances
        if not self.transpose_mode:
            dist, idx = dist.transpose(1, 2), idx.transpose(1, 2)
        return dist, idx
