"""Drop-in for ``HallucinationCrossModalityNetwork`` of the reference's
``models/hallucination_network.py:18-69`` (the feature-to-feature distillation MLPs).

Same constructor, parameter names (``xyz_norm.*``, ``xyz_mlp.mlp_module.0.fc{1,2,3}.*``, ``rgb_norm.*``,
``rgb_mlp.mlp_module.0.fc{1,2,3}.*``), ``hallucination_generation`` and ``forward`` signatures; checkpoints
written by either implementation load in the other.  The arithmetic runs in the HIP kernels:
inference through cmdiad_amd.runtime.PackedHallucination, training (forward + loss + backward) through
cmdiad_amd.train, exposed to autograd so ``loss.backward()`` / ``torch.optim.Adam`` in
hallucination_network_pretrain.py keep working unchanged.

The other heads of the reference file -- HallucinationCrossModalityConv (72-143), HallucinationRGBFeatureToXYZInputMLP
(146-182), HallucinationFeatureToInputConv (185-220) -- (SURVEY 8f row f4): same constructors, state_dict keys and
``hallucination_generation`` / ``forward`` signatures.  INFERENCE (``eval()`` or ``no_grad``) runs eval-mode arithmetic
(BatchNorm running statistics folded into the convolution weights) on the implicit-GEMM convolution kernel
(cmdiad_conv2d_nhwc_bf16) and cmdiad_upsample_bicubic.  TRAINING (``train()`` with gradients enabled:
hallucination_network_pretrain.py:106-147): all three train on the hand-written paths of cmdiad_amd/conv_train.py (batch-statistics
BatchNorm, bf16 MFMA convolutions / GEMMs forward, data gradient and weight gradient, bicubic adjoint; exposed to autograd so the
reference's loop runs unchanged); CMDIAD_CONV_TRAIN=torch selects the modules' own torch layers (fp32, MIOpen / rocBLAS) as the A/B
reference.  tests/test_gpu_heads.py checks a three-step Adam loss curve of each head against the reference's own (golden G12);
tests/test_gpu_conv_train.py the hand-written paths against torch autograd.
"""
import os

import torch
import torch.nn as nn

from .. import runtime
from ..utils.utils import MlpModule


def feature_reshape(feature):
    """[B, HW, C] token matrix -> [B, C, H, W] (hallucination_network.py:6-9)."""
    side = int(round(feature.shape[1] ** 0.5))
    return feature.transpose(1, 2).reshape(feature.shape[0], feature.shape[2], side, side)


def feature_reshape_back(feature):
    """[B, C, H, W] -> [B, HW, C] (hallucination_network.py:12-15)."""
    return feature.reshape(feature.shape[0], feature.shape[1], -1).transpose(1, 2)


class _PackedHead(nn.Module):
    """Caches the device-side packing of an inference head and refreshes it when a parameter changes."""
    _packer = None

    def _pack(self):
        flat = self.__dict__.get("_cmdiad_flat")  # the module walk costs ms; re-collected every 64 calls (models._param_version)
        if flat is None or flat[1] <= 0:
            flat = [list(self.parameters()) + list(self.buffers()), 64]
            self.__dict__["_cmdiad_flat"] = flat
        flat[1] -= 1
        tensors = flat[0]
        ver = tuple((t.data_ptr(), t._version) for t in tensors)
        cached = self.__dict__.get("_cmdiad_packed")
        if cached is None or cached[0] != ver:
            dev = tensors[0].device
            if dev.type != "cuda":
                raise RuntimeError(f"{type(self).__name__}: move the module to the GPU first (cmdiad_amd has no CPU path)")
            cached = (ver, type(self)._packer(self.state_dict(), dev))
            self.__dict__["_cmdiad_packed"] = cached
        return cached[1]

    def _autograd(self):
        """True when forward() has to build a graph: train() mode with gradients enabled (the pretraining loop)."""
        return self.training and torch.is_grad_enabled()

    def _device(self):
        return next(self.parameters()).device

    @staticmethod
    def _mean_row_norm(a, b, dim):
        d = torch.linalg.norm(a - b, dim=dim)
        return torch.sum(d) / d.shape[0]


class HallucinationCrossModalityNetwork(nn.Module):
    def __init__(self, args, xyz_dim, rgb_dim, hidden_ratio=2.5, mlp_depth=1):
        super().__init__()
        if mlp_depth < 1:
            raise ValueError("mlp_depth must be >= 1")
        if mlp_depth > 1 and xyz_dim != rgb_dim:
            raise ValueError("mlp_depth > 1 chains blocks of in_features -> out_features (utils/utils.py:103-115): "
                             "the two modalities must have the same width")
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
        dev = self.xyz_norm.weight.device
        # one contiguous fp32 copy per modality (the trainer hands in slices of a [B,T,1536] batch): each is the input of one
        # direction and the target of the other
        xyz_feature = xyz_feature.to(dev).float().contiguous()
        rgb_feature = rgb_feature.to(dev).float().contiguous()
        loss_xyz = train.direction_loss(self, 'rgb', rgb_feature, xyz_feature, dist_method)  # rgb -> hallucinated xyz
        loss_rgb = train.direction_loss(self, 'xyz', xyz_feature, rgb_feature, dist_method)  # xyz -> hallucinated rgb
        return loss_xyz, loss_rgb


class HallucinationCrossModalityConv(_PackedHead):
    """hallucination_network.py:72-143: per direction four 3x3 convolutions 768 -> 768 on the 56 x 56 feature map, the
    first three followed by BatchNorm + ReLU."""
    _packer = staticmethod(lambda sd, dev: runtime.PackedConvFtoF(sd, device=dev))

    def __init__(self, args, xyz_dim, rgb_dim):
        super().__init__()
        self.args, self.xyz_dim, self.rgb_dim = args, xyz_dim, rgb_dim

        def tower(cin):
            layers = []
            for i in range(4):
                layers.append(nn.Conv2d(cin if i == 0 else 768, 768, kernel_size=(3, 3), stride=(1, 1), padding=1, bias=False))
                if i < 3:
                    layers += [nn.BatchNorm2d(768), nn.ReLU()]
            return nn.Sequential(*layers)

        self.xyz_conv = tower(xyz_dim)
        self.rgb_conv = tower(rgb_dim)
        self.sig = nn.Sigmoid()

    def hallucination_generation(self, xyz_feature=None, rgb_feature=None, out_type='train'):
        """[B,3136,768] tokens in, tokens out (hallucination_network.py:113-131); 'xyz' = hallucinated xyz features from
        the rgb features, 'rgb' the reverse, 'train' = (xyz_hallucination, rgb_hallucination).  The unused feature may be
        omitted (the reference's signature makes both positional, so its own keyword call sites,
        multiple_features.py:335,351, raise TypeError with this head; the defaults are a superset)."""
        pk = self._pack()
        if out_type == 'train':
            return pk.generate(rgb_feature, 'rgb'), pk.generate(xyz_feature, 'xyz')
        elif out_type == 'xyz':
            return pk.generate(rgb_feature, 'rgb')
        elif out_type == 'rgb':
            return pk.generate(xyz_feature, 'xyz')

    def forward(self, xyz_feature, rgb_feature, sigmoid, dist_method):
        """hallucination_network.py:133-147 -> (distance_to_xyz_real, distance_to_rgb_real)."""
        if self._autograd() and os.environ.get("CMDIAD_CONV_TRAIN", "hip") == "hip":
            # hand-written forward + backward (cmdiad_amd/conv_train.py): batch-statistics BatchNorm, bf16 MFMA convolutions,
            # exposed to autograd so loss.backward() / Adam of the reference's loop work unchanged
            from .. import conv_train
            assert xyz_feature.shape[1:] == (3136, self.xyz_dim) and rgb_feature.shape[1:] == (3136, self.rgb_dim)
            return (conv_train.tower_loss(self.rgb_conv, rgb_feature, xyz_feature, sigmoid is True),
                    conv_train.tower_loss(self.xyz_conv, xyz_feature, rgb_feature, sigmoid is True))
        if self._autograd():   # CMDIAD_CONV_TRAIN=torch: the towers' own torch layers (MIOpen) -- the A/B reference of the above
            dev = self._device()
            xyz_feature, rgb_feature = xyz_feature.to(dev).float(), rgb_feature.to(dev).float()
            xyz_h = feature_reshape_back(self.rgb_conv(feature_reshape(rgb_feature)))
            rgb_h = feature_reshape_back(self.xyz_conv(feature_reshape(xyz_feature)))
            return self._losses(xyz_h, rgb_h, xyz_feature, rgb_feature, sigmoid)
        with torch.no_grad():
            xyz_h, rgb_h = self.hallucination_generation(xyz_feature, rgb_feature, 'train')
            return self._losses(xyz_h, rgb_h, xyz_feature.to(xyz_h.device), rgb_feature.to(rgb_h.device), sigmoid)

    def _losses(self, xyz_h, rgb_h, xyz_feature, rgb_feature, sigmoid):
        assert tuple(xyz_h.shape[1:]) == (3136, 768)
        if sigmoid is True:
            return (self._mean_row_norm(self.sig(xyz_h), self.sig(xyz_feature), 2),
                    self._mean_row_norm(self.sig(rgb_h), self.sig(rgb_feature), 2))
        return self._mean_row_norm(xyz_h, xyz_feature, 2), self._mean_row_norm(rgb_h, rgb_feature, 2)


class HallucinationRGBFeatureToXYZInputMLP(_PackedHead):
    """hallucination_network.py:146-182: LayerNorm -> 768 -> 1152 -> 384 -> 96 -> 3 (or 1 with --estimate_depth) with GELU
    between, then bicubic 56 -> 224."""
    _packer = staticmethod(lambda sd, dev: runtime.PackedFtoIMLP(sd, device=dev))

    def __init__(self, args, rgb_dim):
        super().__init__()
        self.args = args
        out_dim = 1 if getattr(args, "estimate_depth", False) else 3
        self.rgb_dim = rgb_dim
        self.rgb_norm = nn.LayerNorm(rgb_dim)
        widths = (rgb_dim, 1152, 384, 96, out_dim)
        layers = []
        for i in range(4):
            layers.append(nn.Linear(widths[i], widths[i + 1]))
            if i < 3:
                layers.append(nn.GELU())
        self.mlp = nn.Sequential(*layers)

    def hallucination_generation(self, x):
        """[B,3136,768] -> [B,out_dim,224,224] f32 on the GPU."""
        return self._pack().generate(x)

    def forward(self, rgb_feature, xyz):
        """hallucination_network.py:174-182."""
        rgb_feature = rgb_feature.reshape(rgb_feature.shape[0], rgb_feature.shape[1], -1)
        if self._autograd() and os.environ.get("CMDIAD_CONV_TRAIN", "hip") == "hip":   # hand-written forward + backward
            from .. import conv_train
            assert rgb_feature.shape[1:] == (3136, self.rgb_dim) and xyz.shape[2:] == (224, 224)
            return conv_train.ftoi_mlp_loss(self, rgb_feature, xyz)
        if self._autograd():   # CMDIAD_CONV_TRAIN=torch: the module's own torch layers, the A/B reference of the above
            dev = self._device()
            x = self.mlp(self.rgb_norm(rgb_feature.to(dev).float())).transpose(1, 2)
            h = nn.functional.interpolate(x.reshape(x.shape[0], x.shape[1], 56, 56), size=(224, 224), mode='bicubic')
            return self._mean_row_norm(h, xyz.to(dev), 1)
        with torch.no_grad():
            h = self.hallucination_generation(rgb_feature)
            return self._mean_row_norm(h, xyz.to(h.device), 1)


class HallucinationFeatureToInputConv(_PackedHead):
    """hallucination_network.py:185-220: conv 768 -> 384 at 56 x 56, bicubic to 224 x 224, conv 384 -> 96 -> 32 -> 3 with
    ReLU between (``norm`` exists in the state_dict but the reference's forward never applies it)."""
    _packer = staticmethod(lambda sd, dev: runtime.PackedFtoIConv(sd, device=dev))

    def __init__(self, args=None, dim=768):
        super().__init__()
        self.args, self.dim = args, dim
        self.norm = nn.LayerNorm(dim)
        widths = (dim, 384, 96, 32, 3)
        for i in range(4):
            setattr(self, f"conv{i + 1}", nn.Conv2d(widths[i], widths[i + 1], kernel_size=(3, 3), stride=(1, 1), padding=1))
        self.act = nn.ReLU(inplace=True)

    def hallucination_generation(self, feature):
        """[B,3136,768] -> [B,3,224,224] f32 on the GPU."""
        return self._pack().generate(feature)

    def forward(self, feature, img):
        """hallucination_network.py:211-220."""
        if self._autograd() and os.environ.get("CMDIAD_CONV_TRAIN", "hip") == "hip":   # hand-written forward + backward
            from .. import conv_train
            assert feature.shape[1:] == (3136, self.dim) and img.shape[1:] == (3, 224, 224)
            return conv_train.ftoi_conv_loss(self, feature, img)
        if self._autograd():   # CMDIAD_CONV_TRAIN=torch: the module's own torch layers, the A/B reference of the above
            dev = self._device()
            f = feature.to(dev).float().transpose(1, 2)
            h = self.conv1(f.reshape(f.shape[0], f.shape[1], 56, 56))
            h = nn.functional.interpolate(h, size=(224, 224), mode='bicubic')
            h = self.conv4(torch.relu(self.conv3(torch.relu(self.conv2(h)))))
            assert h.shape[1:] == (3, 224, 224) and img.shape[1:] == (3, 224, 224)
            return self._mean_row_norm(h, img.to(dev), 1)
        with torch.no_grad():
            h = self.hallucination_generation(feature)
            assert h.shape[1:] == (3, 224, 224) and img.shape[1:] == (3, 224, 224)
            return self._mean_row_norm(h, img.to(h.device), 1)
