"""Host-side helpers with the reference's names (reference utils/utils.py).

* set_seeds / set_multithreading ........ utils/utils.py:11-31
* KNNGaussianBlur ....................... utils/utils.py:71-83: the 8-bit quantisation (ToPILImage = mul(255).byte(),
  ToTensor = /255) and Pillow's GaussianBlur run on the device in cmdiad_blur8_maps, bit-exact
* MlpBlock / MlpModule .................. utils/utils.py:86-115: parameter containers with the reference's
  state_dict keys (fc1/fc2/fc3 under mlp_module.<i>); their arithmetic runs in the HIP kernels
  (cmdiad_amd.runtime / cmdiad_amd.train), not in torch.
* save_model / load_model ............... utils/utils.py:34-68
"""
import os
import random
from pathlib import Path

import numpy as np
import torch
from torch import nn


def set_seeds(seed: int = 0) -> None:
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def set_multithreading(cpu_num: int = 8) -> None:
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "VECLIB_MAXIMUM_THREADS",
                "NUMEXPR_NUM_THREADS"):
        os.environ[var] = str(cpu_num)
    torch.set_num_threads(cpu_num)


class KNNGaussianBlur(torch.nn.Module):
    """utils/utils.py:71-83.  The reference quantises the map to 8 bits and runs Pillow's GaussianBlur on the host;
    here the identical integer arithmetic runs in the HIP kernel cmdiad_blur8_maps (bit-exact with Pillow, see
    oracle orc_pil_gaussian_blur_u8 and tests/test_gpu_kernels.py::test_blur8_maps_bit_exact)."""

    def __init__(self, radius: int = 4):
        super().__init__()
        self.radius = radius

    def __call__(self, img):
        """img [1,1,H,W] -> [1,H,W] CPU f32 (the reference's return placement)."""
        from .. import ops
        dev = img.device if img.is_cuda else torch.device("cuda")
        maps = img.detach().to(dev, torch.float32).reshape(1, img.shape[-2], img.shape[-1]).contiguous()
        return ops.blur8_maps(maps, float(self.radius)).cpu()


class MlpBlock(nn.Module):
    def __init__(self, in_features, hidden_features, out_features, act_layer=nn.GELU):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, hidden_features)
        self.fc3 = nn.Linear(hidden_features, out_features)


class MlpModule(nn.Module):
    def __init__(self, in_features, hidden_features, out_features=None, act_layer=nn.GELU, mlp_depth=1):
        super().__init__()
        out_features = in_features if out_features is None else out_features
        self.mlp_module = nn.ModuleList(
            [MlpBlock(in_features, hidden_features, out_features, act_layer) for _ in range(mlp_depth)])


def load_model(args, model, optimizer, loss_scaler=None):
    if not getattr(args, "resume", None):
        return
    checkpoint = torch.load(args.resume, map_location="cpu")
    model.load_state_dict(checkpoint["model"])
    print("Resume checkpoint %s" % args.resume)
    # the reference reads args.train_stage, which no parser defines: treat "absent" as first stage
    if getattr(args, "train_stage", "first") != "second":
        if "optimizer" in checkpoint and "epoch" in checkpoint and not getattr(args, "eval", False):
            optimizer.load_state_dict(checkpoint["optimizer"])
            args.start_epoch = checkpoint["epoch"] + 1
            if loss_scaler is not None and "scaler" in checkpoint:
                loss_scaler.load_state_dict(checkpoint["scaler"])


def save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler, without_opt=True):
    path = Path(args.output_dir) / ("checkpoint-%s.pth" % str(epoch))
    to_save = {"model": model_without_ddp.state_dict(), "epoch": epoch, "args": args}
    if loss_scaler is not None:
        to_save.update(optimizer=optimizer.state_dict(), scaler=loss_scaler.state_dict())
    elif not without_opt:
        to_save.update(optimizer=optimizer.state_dict())
    torch.save(to_save, path)


class Interpolate(torch.nn.Module):
    """utils/utils.py:118-127: F.interpolate(size, mode, align_corners=False) as a layer (no caller in the reference)."""

    def __init__(self, size, mode):
        super().__init__()
        self.size, self.mode = size, mode

    def forward(self, x):
        return torch.nn.functional.interpolate(x, size=self.size, mode=self.mode, align_corners=False)
