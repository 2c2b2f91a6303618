"""Drop-in for ``HallucinationCrossModalityNetwork`` of the reference's
``models/hallucination_network.py:18-69`` (the feature-to-feature distillation MLPs).

Same constructor, parameter names (``xyz_norm.*``, ``xyz_mlp.mlp_module.0.fc{1,2,3}.*``, ``rgb_norm.*``,
``rgb_mlp.mlp_module.0.fc{1,2,3}.*``), ``hallucination_generation`` and ``forward`` signatures; checkpoints
written by either implementation load in the other.  The arithmetic runs in the HIP kernels:
inference through cmdiad_amd.runtime.PackedHallucination, training (forward + loss + backward) through
cmdiad_amd.train, exposed to autograd so ``loss.backward()`` / ``torch.optim.Adam`` in
hallucination_network_pretrain.py keep working unchanged.

The conv / feature-to-input heads of the reference file (lines 72-220) are out of scope.
"""
import torch
import torch.nn as nn

from .. import runtime
from ..utils.utils import MlpModule


class HallucinationCrossModalityNetwork(nn.Module):
    def __init__(self, args, xyz_dim, rgb_dim, hidden_ratio=2.5, mlp_depth=1):
        super().__init__()
        if mlp_depth != 1:
            raise NotImplementedError("cmdiad_amd implements mlp_depth=1 (the reference default)")
        self.args = args
        self.xyz_dim, self.rgb_dim = xyz_dim, rgb_dim
        self.xyz_norm = nn.LayerNorm(xyz_dim)
        self.xyz_mlp = MlpModule(in_features=xyz_dim, hidden_features=int(xyz_dim * hidden_ratio),
                                 out_features=self.rgb_dim, act_layer=nn.GELU, mlp_depth=mlp_depth)
        self.rgb_norm = nn.LayerNorm(rgb_dim)
        self.rgb_mlp = MlpModule(in_features=rgb_dim, hidden_features=int(rgb_dim * hidden_ratio),
                                 out_features=self.xyz_dim, act_layer=nn.GELU, mlp_depth=mlp_depth)
        self._packed = None

    def _pack(self):
        ver = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is None or self._packed[0] != ver:
            dev = next(self.parameters()).device
            self._packed = (ver, runtime.PackedHallucination(self.state_dict(), device=dev))
        return self._packed[1]

    def hallucination_generation(self, xyz_feature=None, rgb_feature=None, out_type='Train'):
        """models/hallucination_network.py:34-45.  [B,T,D] -> [B,T,D'] (CUDA, fp32)."""
        pk = self._pack()
        dev = next(self.parameters()).device
        if out_type == 'train':
            return (pk.generate(rgb_feature.to(dev).float(), 'rgb'), pk.generate(xyz_feature.to(dev).float(), 'xyz'))
        elif out_type == 'xyz':
            return pk.generate(rgb_feature.to(dev).float(), 'rgb')
        elif out_type == 'rgb':
            return pk.generate(xyz_feature.to(dev).float(), 'xyz')

    def forward(self, xyz_feature, rgb_feature, sigmoid, dist_method='cos_dist'):
        """models/hallucination_network.py:47-69 -> (loss_xyz, loss_rgb), differentiable w.r.t. the
        parameters (autograd.Function over the HIP forward/backward kernels)."""
        from .. import train
        assert len(xyz_feature.shape) == 3 and len(rgb_feature.shape) == 3
        assert xyz_feature.shape[2] == self.xyz_dim and rgb_feature.shape[2] == self.rgb_dim
        loss_xyz = train.direction_loss(self, 'rgb', rgb_feature, xyz_feature, dist_method)  # rgb -> hallucinated xyz
        loss_rgb = train.direction_loss(self, 'xyz', xyz_feature, rgb_feature, dist_method)  # xyz -> hallucinated rgb
        return loss_xyz, loss_rgb
