"""Drop-in for ``HRNet`` of the reference's ``models/hrnet.py`` -- the input-to-feature distillation head
(``--use_hrnet``, README "ItoF"): what the reference's constructor actually builds and its ``hallucination_generation``
actually runs (hrnet.py:146-177, 251-262) is a ResNet-style trunk, not a multi-resolution HRNet: two stride-2 3x3
stem convolutions (3 -> 64 -> 128, 224 -> 56), twelve Bottlenecks (layer1-3; layer4 is constructed, so it is in the
state_dict, but never called), and a 1x1 ``final_layer`` (c -> 768; the trunk is 512 wide, so c must be 512).  The
StageModule / BasicBlock / transition code of that file is commented out or never instantiated and is not reproduced.

SURVEY 8f row f4: same constructor, state_dict keys and method signatures.  Inference (``eval()`` / ``no_grad``): eval-mode
arithmetic with BatchNorm folded, on cmdiad_conv_stem + cmdiad_conv2d_nhwc_bf16 (cmdiad_amd.runtime.PackedHRNet).  Training
(``train()`` with gradients: --train_method *InputTo*FeatureHRNET): the hand-written forward + backward of
cmdiad_amd/conv_train.py from batch 16 up (the reference's default batch is 64; 1.4x the torch layers at batch 32), the module's own
torch layers on the GPU (fp32, batch-statistics BatchNorm, autograd; MIOpen kernels) below that, where the hand-written path's
~500 launches per step are not hidden yet; CMDIAD_HRNET_TRAIN=hip / torch forces one.  Both follow golden G12
(tests/test_gpu_heads.py) and agree with each other (tests/test_gpu_conv_train.py).
"""
import os

import torch
from torch import nn

from .. import runtime
from .hallucination_network import _PackedHead, feature_reshape_back  # noqa: F401  (re-exported like the reference)


class Bottleneck(nn.Module):
    """hrnet.py:8-43: 1x1 (in -> planes), 3x3 (planes -> planes), 1x1 (planes -> 4 planes), each + BatchNorm; ReLU after
    the first two and after the residual sum.  Parameter container: the arithmetic runs in PackedHRNet."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, bn_momentum=0.1):
        super().__init__()
        shapes = ((inplanes, planes, 1, 1, 0), (planes, planes, 3, stride, 1), (planes, planes * self.expansion, 1, 1, 0))
        for i, (cin, cout, k, s, p) in enumerate(shapes, 1):
            setattr(self, f"conv{i}", nn.Conv2d(cin, cout, kernel_size=k, stride=s, padding=p, bias=False))
            setattr(self, f"bn{i}", nn.BatchNorm2d(cout, momentum=bn_momentum))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        """hrnet.py:23-43 (training path of HRNet.forward; inference never calls it)."""
        out = torch.relu(self.bn1(self.conv1(x)))
        out = torch.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return torch.relu(out + (x if self.downsample is None else self.downsample(x)))


class HRNet(_PackedHead):
    _packer = staticmethod(lambda sd, dev: runtime.PackedHRNet(sd, device=dev))

    def __init__(self, c=48, nof_joints=17, bn_momentum=0.1):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=(3, 3), stride=(2, 2), padding=(1, 1), bias=False)
        self.bn1 = nn.BatchNorm2d(64, eps=1e-05, momentum=bn_momentum, affine=True, track_running_stats=True)
        self.conv2 = nn.Conv2d(64, 128, kernel_size=(3, 3), stride=(2, 2), padding=(1, 1), bias=False)
        self.bn2 = nn.BatchNorm2d(128, eps=1e-05, momentum=bn_momentum, affine=True, track_running_stats=True)
        self.relu = nn.ReLU(inplace=True)
        widen = nn.Sequential(nn.Conv2d(128, 512, kernel_size=(1, 1), stride=(1, 1), bias=False),
                              nn.BatchNorm2d(512, eps=1e-05, momentum=bn_momentum, affine=True, track_running_stats=True))
        for n in range(1, 5):
            blocks = [Bottleneck(128 if (n == 1 and i == 0) else 512, 128, downsample=widen if (n == 1 and i == 0) else None)
                      for i in range(4)]
            setattr(self, f"layer{n}", nn.Sequential(*blocks))
        self.final_layer = nn.Conv2d(c, nof_joints, kernel_size=(1, 1), stride=(1, 1))

    def hallucination_tokens(self, x):
        """[B,3,224,224] -> [B,3136,nof_joints] f32 tokens (what every caller reshapes the reference's output into)."""
        return self._pack().generate(x)

    def hallucination_generation(self, x):
        """hrnet.py:251-288: [B,3,224,224] -> [B,nof_joints,56,56] (a view of the token matrix)."""
        t = self.hallucination_tokens(x)
        return t.view(t.shape[0], 56, 56, t.shape[2]).permute(0, 3, 1, 2)

    def forward(self, img, feature):
        """hrnet.py:290-299."""
        # hand-written forward + backward (cmdiad_amd/conv_train.py), replayed as one HIP graph from the third step on: 5.6 / 6.2 /
        # 7.2 / 9.1 / 12.9 / 21.3 ms at batch 1 / 2 / 4 / 8 / 16 / 32 against 7.7 / 7.9 / 7.5 / 11.9 / 21.0 / 39.9 ms on the module's
        # torch layers (MIOpen / rocBLAS; profiles/r4_notes.md section 17).  CMDIAD_HRNET_TRAIN = auto (default: the hand-written
        # path) | hip | torch (the module's own layers, kept for A/B runs).
        mode = os.environ.get("CMDIAD_HRNET_TRAIN", "auto")
        if mode == "auto":
            mode = "hip"
        if self._autograd() and mode == "hip":
            from .. import conv_train
            assert tuple(img.shape[1:]) == (3, 224, 224) and tuple(feature.shape[1:]) == (3136, self.final_layer.out_channels)
            return conv_train.hrnet_loss(self, img, feature)
        if self._autograd():   # CMDIAD_HRNET_TRAIN=torch: the module's own torch layers (MIOpen / rocBLAS, autograd)
            dev = self._device()
            x = torch.relu(self.bn1(self.conv1(img.to(dev).float())))
            x = torch.relu(self.bn2(self.conv2(x)))
            x = self.final_layer(self.layer3(self.layer2(self.layer1(x))))
            assert tuple(x.shape[1:]) == (768, 56, 56) and tuple(feature.shape[1:]) == (3136, 768)
            return self._mean_row_norm(feature_reshape_back(x), feature.to(dev), 2)
        with torch.no_grad():
            h = self.hallucination_tokens(img)
            assert tuple(h.shape[1:]) == (3136, 768) and tuple(feature.shape[1:]) == (3136, 768)
            return self._mean_row_norm(h, feature.to(h.device), 2)
