"""CPU end-to-end path composed the way the reference composes it (TEST INFRASTRUCTURE ONLY; also the
``cpu_baseline`` of bench.py, kind "port"): torch-CPU for the networks, ``torch.cdist``, pooling and
scoring exactly as feature_extractors/features.py and multiple_features.py call them, the C oracle
for FPS / kNN grouping (the reference has no CPU implementation of those: models/models.py:5-6
hard-imports the CUDA packages).

Protocol mirrored: DoubleRGBPointFeatures (multiple_features.py:800-1015) = "DINO+Point_MAE".
"""
import math
import time

import numpy as np
import torch

from . import kernels as ok
from . import nets, scoring


class CpuExtractor:
    def __init__(self, sd_vit, sd_pm, num_group=1024, group_size=128):
        self.sd_vit, self.sd_pm, self.G, self.M = sd_vit, sd_pm, num_group, group_size
        self.timing = {}

    def _t(self, key, t0):
        self.timing[key] = self.timing.get(key, 0.0) + time.perf_counter() - t0

    def __call__(self, rgb, organized_pc):
        """rgb [1,3,S,S], organized_pc [1,3,S,S] -> (rgb_patch [784,768], xyz_patch [3136,768])."""
        with torch.no_grad():
            t0 = time.perf_counter()
            pc, nz = scoring.unorganize_no_zeros(organized_pc)                    # a1
            self._t("unorganize", t0); t0 = time.perf_counter()
            fmap = nets.vit_forward(self.sd_vit, rgb)                              # a2
            self._t("vit", t0); t0 = time.perf_counter()
            xyz = np.ascontiguousarray(pc[0].T.numpy())[None]
            cidx, cen = ok.fps(xyz, self.G)                                        # a3
            self._t("fps", t0); t0 = time.perf_counter()
            _, nb = ok.knn_group(xyz, cen, self.M)                                 # a4
            self._t("knn_group", t0); t0 = time.perf_counter()
            tok = nets.pointmae_encoder(self.sd_pm, torch.from_numpy(nb))          # a5
            self._t("encoder", t0); t0 = time.perf_counter()
            center = torch.from_numpy(cen)
            feats = nets.pointmae_transformer(self.sd_pm, tok, center)             # a6  [1,768,G]
            self._t("pmae_transformer", t0); t0 = time.perf_counter()
            interp = scoring.interpolating_points(pc, center.permute(0, 2, 1), feats)  # a7
            self._t("interp", t0); t0 = time.perf_counter()
            xyz_patch = scoring.get_xyz_patch(interp, nz)                          # a9
            self._t("xyz_pool", t0); t0 = time.perf_counter()
            rgb_patch, _ = scoring.get_rgb_patch(fmap)                             # a10
            self._t("rgb_patch", t0)
        return rgb_patch.contiguous(), xyz_patch.contiguous()


class CpuDoubleRGBPoint:
    """fit (banks + cross-wired statistics, optional greedy coreset) and predict (pre-OCSVM scores)."""

    def __init__(self, extractor, lambdas=(1.0, 1.0, 0.1, 0.1), f_coreset=1.0, coreset_eps=0.9, random_state=None):
        self.ex = extractor
        self.f_coreset, self.coreset_eps, self.random_state = f_coreset, coreset_eps, random_state
        self.xyz_s_l, self.xyz_m_l, self.rgb_s_l, self.rgb_m_l = lambdas
        self.timing = {}

    def fit(self, samples, coreset_override=None):
        """coreset_override = (xyz_idx, rgb_idx): use these selections instead of running the greedy coreset (the
        selection is chaotic in the last ulp of its input; tests pin it separately, g9_coreset)."""
        rp, xp = zip(*[self.ex(r, p) for r, p in samples])
        xyz_lib, rgb_lib = torch.cat(xp, 0), torch.cat(rp, 0)
        # multiple_features.py:877-880 (cross-wired)
        self.xyz_mean = self.rgb_mean = torch.mean(xyz_lib)
        self.xyz_std = self.rgb_std = torch.std(rgb_lib)
        self.xyz_lib = (xyz_lib - self.xyz_mean) / self.xyz_std
        self.rgb_lib = (rgb_lib - self.rgb_mean) / self.rgb_std
        if self.f_coreset < 1:  # multiple_features.py:885-895
            self.coreset_idx = {}
            for k, name in enumerate(("xyz_lib", "rgb_lib")):
                lib = getattr(self, name)
                idx = scoring.coreset_idx_randomp(lib, int(self.f_coreset * lib.shape[0]), self.coreset_eps, self.random_state)
                self.coreset_idx[name] = idx
                setattr(self, name, lib[idx if coreset_override is None else torch.as_tensor(coreset_override[k]).long()])
        return list(zip(rp, xp))

    def set_banks(self, xyz_lib, rgb_lib, xyz_mean, xyz_std, rgb_mean, rgb_std):
        self.xyz_lib, self.rgb_lib = xyz_lib, rgb_lib
        self.xyz_mean, self.xyz_std, self.rgb_mean, self.rgb_std = xyz_mean, xyz_std, rgb_mean, rgb_std

    def score(self, rgb_patch, xyz_patch, blur=True):
        t0 = time.perf_counter()
        rx = scoring.score_modality(xyz_patch, self.xyz_lib, self.xyz_mean, self.xyz_std, blur=blur)
        self.timing["score_xyz"] = self.timing.get("score_xyz", 0.0) + time.perf_counter() - t0
        t0 = time.perf_counter()
        rr = scoring.score_modality(rgb_patch, self.rgb_lib, self.rgb_mean, self.rgb_std, blur=blur)
        self.timing["score_rgb"] = self.timing.get("score_rgb", 0.0) + time.perf_counter() - t0
        s = torch.tensor([[self.xyz_s_l * rx["s"], self.rgb_s_l * rr["s"]]])
        s_map = torch.cat([self.xyz_m_l * rx["s_map"], self.rgb_m_l * rr["s_map"]], 0).reshape(2, -1).permute(1, 0)
        return s, s_map, rx, rr

    def predict(self, rgb, organized_pc, blur=True):
        rp, xp = self.ex(rgb, organized_pc)
        return self.score(rp, xp, blur=blur)
