"""``pointnet2_ops.pointnet2_utils`` (erikwijmans/Pointnet2_PyTorch) surface on the HIP kernels.

Same call signatures and result dtypes/layouts as the CUDA wheel: furthest_point_sample -> int32 [B,npoint];
gather_operation(features [B,C,N], idx int32 [B,npoint]) -> [B,C,npoint]; ball_query(radius, nsample, xyz, new_xyz) ->
int32 [B,npoint,nsample]; grouping_operation(features [B,C,N], idx [B,npoint,nsample]) -> [B,C,npoint,nsample];
QueryAndGroup / GroupAll modules.  Inference only (the reference runs them under no_grad; no backward is provided)."""
import torch
from torch import nn

from .. import ops


def _f32c(t):
    return t.detach().float().contiguous()


def furthest_point_sample(xyz, npoint):
    """xyz [B,N,3] -> int32 [B,npoint] (first index 0; points with |p|^2 <= 1e-3 never selected)."""
    return ops.fps(_f32c(xyz), int(npoint))[0]


def gather_operation(features, idx):
    return ops.gather_points(_f32c(features), idx.to(torch.int32).contiguous())


def grouping_operation(features, idx):
    return ops.gather_points(_f32c(features), idx.to(torch.int32).contiguous())


def ball_query(radius, nsample, xyz, new_xyz):
    return ops.ball_query(radius, nsample, _f32c(xyz), _f32c(new_xyz))


class QueryAndGroup(nn.Module):
    """Groups with a ball query: (xyz [B,N,3], new_xyz [B,npoint,3], features [B,C,N] | None) -> [B,3+C,npoint,nsample]."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)  # [B,3,npoint,nsample]
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return grouped_xyz
        grouped = grouping_operation(features, idx)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped


class GroupAll(nn.Module):
    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped = features.unsqueeze(2)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
