"""Host-side helpers with the reference's names (reference utils/utils.py).

* set_seeds / set_multithreading ........ utils/utils.py:11-31
* KNNGaussianBlur ....................... utils/utils.py:71-83 (8-bit PIL blur kept on the host, SURVEY F8;
  torchvision is not a dependency here: ToPILImage/ToTensor are restated as mul(255).byte() / div(255))
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
from PIL import Image, ImageFilter
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
    def __init__(self, radius: int = 4):
        super().__init__()
        self.radius = radius
        self.blur_kernel = ImageFilter.GaussianBlur(radius=radius)

    def __call__(self, img):
        """img [1,1,H,W] (any device) -> [1,H,W] CPU f32: normalise by max, quantise to 8 bits, PIL blur."""
        img = img.detach().to("cpu", torch.float32)
        map_max = img.max()
        u8 = (img[0] / map_max).mul(255).byte().squeeze(0).numpy()
        blurred = Image.fromarray(u8, mode="L").filter(self.blur_kernel)
        return torch.from_numpy(np.asarray(blurred, dtype=np.uint8).copy()).float().div(255).unsqueeze(0) * map_max


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
