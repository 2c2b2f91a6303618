"""Synthetic MVTec-3D-shaped inputs: there is no dataset offline, so tests,
goldens and bench.py all draw their inputs from these seeded generators."""
import math

import torch


def synth_cloud(seed, frac=0.5, size=224, texture=0.0):
    """Organised point cloud [1,3,size,size] f32 in metres: regular x,y grid over +-0.10 m with
    1e-4 jitter, z = 0.50 + 0.03*bump + jitter, background zeroed outside a centred ellipse
    covering ``frac`` of the image (so |p|^2 >> 1e-3 and FPS's skip rule never fires).
    ``texture`` (metres) adds a seed-dependent centimetre-scale relief so that local neighbourhoods -- and
    hence the Point-MAE features of different patches -- differ (a smooth surface gives near-duplicate features)."""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.linspace(-0.1, 0.1, size), torch.linspace(-0.1, 0.1, size), indexing="ij")
    x = xs + 1e-4 * torch.randn(size, size, generator=g)
    y = ys + 1e-4 * torch.randn(size, size, generator=g)
    z = 0.5 + 0.03 * torch.exp(-((xs / 0.05) ** 2 + (ys / 0.07) ** 2)) + 1e-4 * torch.randn(size, size, generator=g)
    if texture:
        ph = torch.rand(4, generator=g) * 6.2831853
        z = z + texture * (torch.sin(xs * (6.2831853 / 0.013) + ph[0]) * torch.cos(ys * (6.2831853 / 0.017) + ph[1])
                           + 0.5 * torch.sin((xs + ys) * (6.2831853 / 0.007) + ph[2]) * torch.cos((xs - ys) * (6.2831853 / 0.023) + ph[3]))
    a = 0.1 * (frac * 4 / math.pi) ** 0.5
    mask = ((xs / a) ** 2 + (ys / (a * 0.85)) ** 2) <= 1.0
    pc = torch.stack([x, y, z], 0) * mask
    return pc.unsqueeze(0).float()


def synth_cloud_fixed_n(seed, n=24576, size=224):
    """Same surface, but exactly ``n`` foreground pixels (the fixed-N regime of the batch-32
    bench): the ``n`` pixels closest to the image centre in the elliptical metric are kept."""
    pc = synth_cloud(seed, frac=1.5, size=size)
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, size), torch.linspace(-1, 1, size), indexing="ij")
    r = (xs ** 2 + (ys / 0.85) ** 2).flatten()
    keep = torch.zeros(size * size, dtype=torch.bool)
    keep[torch.argsort(r, stable=True)[:n]] = True
    return pc * keep.view(1, 1, size, size)


def synth_rgb(seed, size=224):
    """[1,3,size,size] N(0,1): already at ImageNet-normalised scale (dataset.py:62-65)."""
    g = torch.Generator().manual_seed(1234 + seed)
    return torch.randn(1, 3, size, size, generator=g)


def synth_bank(rows, dim=768, seed=4321):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(rows, dim, generator=g)


def synth_labeled_sample(seed, anomalous=False, frac=None, size=224, texture=0.004, dent=0.015, shift=4.0, rough=0.0, side=20):
    """One sample of a synthetic anomaly-detection class -> (rgb [1,3,S,S], organised cloud [1,3,S,S], mask [1,S,S]).
    Normal samples: the textured surface above + N(0,1) image.  Anomalous ones carry a 20 x 20-pixel dent of ``dent``
    metres in z and a colour shift of ``shift`` sigma over the same pixels (SURVEY 8d plants 5 mm / 2 sigma; on iid-noise
    images with a 4 mm relief that leaves the image-level scores of normal and anomalous samples interleaved, so the
    defaults give the ranking a margin -- AUROC parity then measures the scorer, not luck).  ``rough`` (metres) adds seeded
    per-pixel noise inside the ``side`` x ``side`` defect: a flat dent changes the local geometry only along its rim, a rough one
    everywhere inside it."""
    pc = synth_cloud(seed, frac if frac is not None else (0.40 + 0.03 * (seed % 4)), size=size, texture=texture)
    rgb = synth_rgb(seed, size=size)
    mask = torch.zeros(1, size, size)
    if anomalous:
        y0, x0 = 70 + 9 * (seed % 7), 80 + 7 * (seed % 5)
        fg = (pc[0, 2, y0:y0 + side, x0:x0 + side] != 0)
        pc[0, 2, y0:y0 + side, x0:x0 + side] -= dent * fg
        if rough:
            g = torch.Generator().manual_seed(77_000 + seed)
            pc[0, 2, y0:y0 + side, x0:x0 + side] += rough * torch.randn(side, side, generator=g) * fg
        rgb[0, :, y0:y0 + side, x0:x0 + side] += shift
        mask[0, y0:y0 + side, x0:x0 + side] = 1
    return rgb, pc, mask


def sharpen_pointmae(sd, conv_gain=400.0, qk_gain=36.0):
    """Synthetic Point-MAE weights whose features DISCRIMINATE between patches (there are no checkpoints offline).  With
    O(1)-activation random weights the 8 mm neighbourhood coordinates vanish against the biases and random-init attention
    is uniform (every token receives the mean of V), so all 3136 xyz patch features of a sample are near-duplicates: their
    nearest-neighbour distances (~0.02 of |f| ~ 25) sit below the error of ANY 16-bit feature extractor.  A first-convolution
    gain (coordinates become O(1)) and sharper attention logits (q and k rows scaled) lift the patch-to-patch distances to
    ~4, ten times the bf16 error, so the xyz modality carries signal end to end."""
    sd = dict(sd)
    sd["encoder.first_conv.0.weight"] = sd["encoder.first_conv.0.weight"] * conv_gain
    for k in list(sd):
        if k.endswith("attn.qkv.weight"):
            w = sd[k].clone()
            w[: 2 * w.shape[0] // 3] *= qk_gain ** 0.5   # q and k rows: logits scale by qk_gain
            sd[k] = w
    return sd


class SyntheticClass:
    """A seeded stand-in for one MVTec-3D class directory (there is no dataset offline): ``train()`` yields what the
    reference's train loader yields -- ``(sample, label)`` with sample = (img, organised cloud, depth stand-in), dataset.py
    -- and ``test()`` what its test loader yields -- ``(sample, mask, label, rgb_path)``.  Three of every ten test samples
    are anomalous.  Every class draws from its own seed range, so classes are independent of each other and a
    class is the same whichever rank evaluates it."""

    def __init__(self, name, n_train, n_test, index=0, anomalous=lambda i: i % 10 in (0, 3, 7), defect=None, severity=1.0):
        """``severity`` scales the planted defect (depth of the dent and its roughness).  At 1.0 every anomalous sample scores far
        above every normal one (I-AUROC 1.000 on the oracle: a saturated metric that cannot show a regression); at 0.35 the
        oracle's image scores of normal and anomalous samples interleave (I-AUROC 0.88 on both test classes of
        tests/test_gpu_evaluate.py), so the ranking metrics are sensitive to the scorer."""
        self.name, self.n_train, self.n_test, self.index, self._anom = name, int(n_train), int(n_test), int(index), anomalous
        self.defect = dict(dent=0.02 * severity, rough=0.008 * severity, side=28) if defect is None else dict(defect)

    def _seed(self, split, i):
        return 100_000 * (self.index + 1) + (0 if split == "train" else 50_000) + i

    def train(self):
        for i in range(self.n_train):
            rgb, pc, _ = synth_labeled_sample(self._seed("train", i))
            yield (rgb, pc, pc), 0

    def test(self):
        import numpy as np
        for i in range(self.n_test):
            an = bool(self._anom(i))
            rgb, pc, mask = synth_labeled_sample(self._seed("test", i), anomalous=an, **self.defect)
            yield (rgb, pc, pc), mask, np.array([int(an)]), [f"{self.name}/test/{i:03d}.png"]
