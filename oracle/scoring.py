"""Torch-CPU restatement of the Features-level arithmetic (patching, kNN scoring, blur).

TEST INFRASTRUCTURE ONLY.  Each function cites the reference lines it follows; the
composition (which torch op is called on what) is the reference's own, so this is also
the "reference CPU path" that bench.py times as ``cpu_baseline`` (kind "port").
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image, ImageFilter


def unorganize_no_zeros(organized_pc):
    """feature_extractors/multiple_features.py:10-25.  organized_pc [1,3,H,W] ->
    (pc [1,3,N] f32, nonzero_indices [N] int64): keep pixels whose x, y and z are all != 0."""
    pc = organized_pc.squeeze(0).permute(1, 2, 0).reshape(-1, 3).numpy()
    nz = np.nonzero(np.all(pc != 0, axis=1))[0]
    return torch.tensor(pc[nz, :]).unsqueeze(0).permute(0, 2, 1).contiguous(), nz


def interpolating_points(xyz1, xyz2, points2):
    """models/pointnet2_utils.py:45-75.  xyz1 [B,3,N], xyz2 [B,3,S], points2 [B,D,S] -> [B,D,N]."""
    a = xyz1.permute(0, 2, 1)
    b = xyz2.permute(0, 2, 1)
    f = points2.permute(0, 2, 1)
    d = -2 * torch.matmul(a, b.permute(0, 2, 1))
    d += torch.sum(a ** 2, -1).unsqueeze(-1)
    d += torch.sum(b ** 2, -1).unsqueeze(1)
    d, idx = d.sort(dim=-1)  # the reference sorts all S distances (:66) and keeps three (:67)
    d, idx = d[:, :, :3], idx[:, :, :3]
    r = 1.0 / (d + 1e-8)
    w = r / r.sum(dim=2, keepdim=True)
    B, N, _ = a.shape
    g = f[torch.arange(B).view(B, 1, 1), idx]  # [B,N,3,D]
    return (g * w.unsqueeze(-1)).sum(dim=2).permute(0, 2, 1)


def get_rgb_patch(rgb_feature_map):
    """features.py:160-167.  [1,768,28,28] -> ([784,768], [3136,768]); adaptive 28->56 is exact
    nearest-neighbour 2x replication."""
    C = rgb_feature_map.shape[1]
    p = rgb_feature_map.reshape(C, -1).T
    s = int(math.sqrt(p.shape[0]))
    p2 = F.adaptive_avg_pool2d(p.permute(1, 0).reshape(-1, s, s), (56, 56)).reshape(C, -1).T
    return p, p2


def get_xyz_patch(interpolated_pc, nonzero_indices, size=224, out=56):
    """features.py:169-184.  interpolated_pc [1,D,N] -> [out*out, D]."""
    D = interpolated_pc.shape[1]
    full = torch.zeros((1, D, size * size), dtype=interpolated_pc.dtype)
    full[:, :, torch.as_tensor(nonzero_indices)] = interpolated_pc
    full = full.view(1, D, size, size)
    pooled = F.adaptive_avg_pool2d(F.avg_pool2d(full, 3, stride=1), (out, out))
    return pooled.reshape(D, -1).T


def knn_gaussian_blur(img, radius=4):
    """utils/utils.py:71-83.  img [1,1,H,W] f32 -> [1,H,W].  torchvision's ToPILImage on a float
    tensor is ``mul(255).byte()`` -> mode 'L' and ToTensor is ``/255`` [external, torchvision is
    not vendored]: the map is quantised to 8 bits around PIL's GaussianBlur."""
    mx = img.max()
    u8 = (img[0] / mx).mul(255).byte().squeeze(0).numpy()
    blurred = Image.fromarray(u8, mode="L").filter(ImageFilter.GaussianBlur(radius=radius))
    return torch.from_numpy(np.asarray(blurred, dtype=np.uint8).copy()).float().div(255).unsqueeze(0) * mx


def single_s_s_map(patch, dist, bank, dims, gt_size=224, n_reweight=3, blur=True):
    """features.py:225-297 (compute_single_s_s_map) for one modality.
    patch [Q,D], dist [Q,Nb] = cdist(patch, bank), bank [Nb,D].
    Returns dict with every intermediate so parity tests can localise a mismatch."""
    min_val, min_idx = torch.min(dist, dim=1)
    s_idx = torch.argmax(min_val)
    s_star = torch.max(min_val)
    m_test = patch[s_idx].unsqueeze(0)
    m_star = bank[min_idx[s_idx]].unsqueeze(0)
    w_dist = torch.cdist(m_star, bank)
    _, nn_idx = torch.topk(w_dist, k=n_reweight, largest=False)
    m_star_knn = torch.linalg.norm(m_test - bank[nn_idx[0, 1:]], dim=1)
    D = torch.sqrt(torch.tensor(patch.shape[1]))
    w = 1 - (torch.exp(s_star / D) / (torch.sum(torch.exp(m_star_knn / D))))
    s = w * s_star
    s_map_pre = F.interpolate(min_val.view(1, 1, *dims), size=(gt_size, gt_size), mode="bilinear")
    out = dict(min_val=min_val, min_idx=min_idx, s_idx=s_idx, s_star=s_star, nn_idx=nn_idx[0],
               m_star_knn=m_star_knn, w=w, s=s, s_map_pre=s_map_pre[0])
    out["s_map"] = knn_gaussian_blur(s_map_pre) if blur else s_map_pre[0]
    return out


def score_modality(patch, bank, mean, std, blur=True):
    """normalise (multiple_features.py:976-977) + calculate_dist (features.py:186-190) + s/s_map."""
    patch = (patch - mean) / std
    dist = torch.cdist(patch, bank)
    side = int(math.sqrt(patch.shape[0]))
    return single_s_s_map(patch, dist, bank, (side, side), blur=blur)


def coreset_idx_randomp(z_lib, n, eps=0.9, random_state=None):
    """features.py:360-425 get_coreset_idx_randomp with dist_method_coreset='l2', coreset_dtype='FP16', restated for the
    CPU (the reference hard-codes .to("cuda"), features.py:397-399): sparse random projection, then greedy k-centre
    selection -- distances of the half-precision rows to the last pick (difference rounded to half, norm accumulated in
    float, result rounded to half), running minimum, FIRST arg-max, picked entry zeroed.  Pinned by
    tests/golden/g9_coreset.npz (the reference's own function run with the device string redirected)."""
    from sklearn import random_projection
    transformer = random_projection.SparseRandomProjection(eps=eps, random_state=random_state)
    z = torch.tensor(transformer.fit_transform(z_lib.numpy()))
    last = z[0:1]
    min_d = torch.linalg.norm(z - last, dim=1, keepdims=True).half()
    zh, last = z.half(), last.half()
    sel = [0]
    for _ in range(n - 1):
        d = torch.linalg.norm(zh - last, dim=1, keepdims=True)
        min_d = torch.minimum(d, min_d)
        i = int(torch.argmax(min_d))
        last = zh[i:i + 1]
        min_d[i] = 0
        sel.append(i)
    return torch.tensor(sel)
