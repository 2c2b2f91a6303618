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
